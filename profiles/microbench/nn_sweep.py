"""nn-chain time per map for several sizes and HICMI_NNCHAIN_* settings (each setting in this process: env is read per call)."""
import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hic_genome_assembler_amd import _lib as hic, synth
sizes = [int(a) for a in sys.argv[1].split(",")]
settings = [dict(kv.split("=") for kv in s.split("+") if kv) for s in sys.argv[2].split(",")]
dev = torch.device("cuda", 0)
for n in sizes:
    lay = synth.make_layout(n, seed=1)
    ct = synth.dense_contacts_torch(lay, dev, seed=1, sinkhorn_iters=8)
    torch.cuda.synchronize()
    with hic.Context(0) as ctx:
        ctx.set_contacts_device(ct.data_ptr(), n, keepalive=ct)
        ref = None
        for st in settings:
            for k in list(os.environ):
                if k.startswith("HICMI_NNCHAIN_"): del os.environ[k]
            for k, v in st.items(): os.environ["HICMI_NNCHAIN_" + k] = v  # e.g. W1_COLS=256
            os.environ["HICMI_NO_PRESORT"] = "1"
            ctx.upgma()
            ctx.timing_enable(2); ctx.timing_reset()
            reps = 3 if n <= 8000 else 2
            for _ in range(reps): ctx.upgma()
            t = ctx.timing()["nnchain"]; stats = ctx.nnchain_stats()
            z = ctx.raw_merges()
            if ref is None: ref = z
            print(n, st, "nnchain ms %.2f  us/merge %.2f  scans/merge %.2f  same=%s" % (t["ms"] / reps, 1e3 * t["ms"] / reps / (n - 1), stats["scans"] / stats["merges"], np.array_equal(z, ref)), flush=True)
    del ct; torch.cuda.empty_cache()
