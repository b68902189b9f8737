"""One map over two ranks ON THE GPU: both processes open libhicmi contexts on cuda:0 (gloo carries the
all-gathers, as RCCL does on a multi-GPU node).  Part 1's row-independent stages are sharded by rows
(hicmi_set_row_shard: each rank sums, sorts and scans only its rows; one all-gather per vector), UPGMA
runs on both, Part 2's chromosomes are dealt to the ranks (orderGenome shard=) and the files must be
the reference's."""
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

import golden_cases as gc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp_root, name, backend="gloo"):
    import contextlib
    import io
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HICMI_NO_PLOTS="1",
                      HICMI_PRESORT_FROM="200")           # (the pre-sort beside the chain also on these small maps)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import golden_cases as cases
    from hic_genome_assembler_amd import _lib, dist, orderGenome as p2, scaffoldToChromosomes as p1
    from hic_genome_assembler_amd.hostio import initiateLoci
    _lib.load()                                                   # the HIP library or nothing
    device = 0
    if backend == "nccl":                                         # one GPU per rank: RCCL carries the all-gathers as device tensors
        import torch
        device = rank
        torch.cuda.set_device(device)
        dist.init("nccl", device=torch.device("cuda", device))
    else:
        dist.init("gloo")
    import numpy as np
    seen = {"scans": 0, "foreign_nonzero": 0}
    inner_cut, inner_filter = _lib.Context.cut_scan, _lib.Context.filter_scan

    def spy_cut(self, start, M, psig, want_x=False):
        sig = inner_cut(self, start, M, psig, want_x)
        rows = np.arange(start, start + len(sig))
        seen["scans"] += 1
        seen["foreign_nonzero"] += int(np.count_nonzero(sig[rows % world != rank]))
        return sig

    def spy_filter(self, start, c, n_rows, M, psig, want_x=False):
        sig = inner_filter(self, start, c, n_rows, M, psig, want_x)
        rows = np.arange(start, start + len(sig))
        seen["scans"] += 1
        seen["foreign_nonzero"] += int(np.count_nonzero(sig[rows % world != rank]))
        return sig
    inner_rank = _lib.Context.rank_matrix

    def spy_rank(self, order):
        inner_rank(self, order)
        seen["presort"] = self.presort_state()[0]
    _lib.Context.cut_scan, _lib.Context.filter_scan, _lib.Context.rank_matrix = spy_cut, spy_filter, spy_rank
    tmp = os.path.join(tmp_root, "r%d" % rank)
    os.makedirs(tmp)
    spec = cases.load_case(name)[0]
    paths = cases.write_case_files(name, tmp)
    f = lambda k: os.path.join(tmp, k)  # noqa: E731
    with contextlib.redirect_stdout(io.StringIO()):
        p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("a.png"), f("b.png"),
                       f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                       True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000, device=device, shard=(rank, world))
        chroms = p2.readChromsFromFile(f("chromosomeGroups.txt"))
        binDict = p2.readGroupingsToValidBins(f("chromosomeGroups.txt"))
        binList = initiateLoci(paths["hicProBedFile"], paths["hicProBiasFile"], binID_dict=binDict)
        adj = p2.buildAdjacencyMatrix(paths["hicProMatrixFile"], binList, device=device)
        try:
            ordered = p2.runResident(adj, binList, f("chromosomeGroups.txt"), f("chromosomeOrders.txt"),
                                     f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000,
                                     shard=(rank, world))
        finally:
            adj.ctx.close()
    assert len(ordered) == len(chroms)
    assert seen["scans"] > 10 and seen["foreign_nonzero"] == 0    # this rank only ever flagged its own rows
    assert seen["presort"] == 1                                   # all rows pre-sorted beside the chain, the own ones re-addressed
    part1 = {fn: open(f(fn)).read() for fn in ("dendrogramOrder.txt", "binGroups.txt", "assessment.txt", "chromosomeGroups.txt")}
    for fn, text in part1.items():
        assert text == cases.golden_text(name, fn), (rank, fn)   # every rank wrote the reference's Part 1 files
    mine = p2.chromosomesOfRank(chroms, rank, world)
    deals = dist.gather_results({rank: mine})
    if rank == 0:
        out = {fn: open(f(fn)).read() for fn in ("chromosomeGroups.txt", "chromosomeOrders.txt", "plotOrder.txt")}
        with open(os.path.join(tmp_root, "sharded.json"), "w") as fh:
            json.dump({"files": out, "deals": {str(k): v for k, v in deals.items()}}, fh)
    else:
        assert not os.path.exists(f("chromosomeOrders.txt"))
    dist.barrier()


@pytest.mark.parametrize("name", ["n600", "n2000"])
def test_one_map_sharded_over_two_ranks_on_the_gpu(tmp_path, name):
    port = 29700 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path), name), nprocs=2, join=True)
    with open(tmp_path / "sharded.json") as fh:
        got = json.load(fh)
    for fn, text in got["files"].items():
        assert text == gc.golden_text(name, fn), fn
    a, b = got["deals"]["0"], got["deals"]["1"]
    assert a and b and not set(a) & set(b)


def test_one_map_sharded_over_two_gpus_with_rccl(tmp_path):
    """The same flow with one GPU per rank and backend "nccl" (= RCCL): dist.gather_owned's all_gather_into_tensor runs on
    DEVICE tensors over xGMI, the time reduction and the barrier are RCCL collectives.  Needs two visible GPUs: the 1-GPU
    boxes this suite usually runs on skip it; the first multi-GPU box exercises it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL all_gather_into_tensor on device tensors)")
    name = "n600"
    port = 29800 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path), name, "nccl"), nprocs=2, join=True)
    with open(tmp_path / "sharded.json") as fh:
        got = json.load(fh)
    for fn, text in got["files"].items():
        assert text == gc.golden_text(name, fn), fn
