import cProfile, pstats, io, os, sys, time, contextlib, tempfile
os.environ["HICMI_PART2_WORKERS"] = "1"
sys.path.insert(0, os.getcwd())
import torch
from hic_genome_assembler_amd import _lib, synth
from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
from hic_genome_assembler_amd.hostio import Bin
import bench
n = 16000
dev = torch.device("cuda", 0)
lay = synth.make_layout(n, seed=1)
contacts = synth.dense_contacts_torch(lay, dev, seed=1, sinkhorn_iters=12)
torch.cuda.synchronize()
work = tempfile.mkdtemp()
sizes = os.path.join(work, "s.sizes"); bench.write_sizes(lay, sizes)
f = lambda k: os.path.join(work, k)
ctx = _lib.Context(0)
bins = bench.make_bins(lay, Bin)
def step(prof=None):
    ctx.set_contacts_device(contacts.data_ptr(), n, keepalive=contacts)
    dm = p1.DeviceMatrix(ctx)
    with contextlib.redirect_stdout(io.StringIO()):
        p1.runResident(dm, list(bins), sizes, f("d.txt"), f("b.txt"), f("a.txt"), f("c.txt"), 5, 0.0, .05)
        if prof: prof.enable()
        p2.runResident(p2.GenomeMatrix(ctx), dm.kept_bins, f("c.txt"), f("o.txt"), f("p.txt"), 6, 5, lay.resolution)
        if prof: prof.disable()
step(); step()
pr = cProfile.Profile()
t0 = time.perf_counter(); step(pr); ctx.synchronize(); print("step ms", (time.perf_counter() - t0) * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
