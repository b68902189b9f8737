#!/usr/bin/env python3
"""Benchmark of the Hi-C hot path on MI355X: one "step" = -part1 -part2 on a synthetic
ICE-balanced contact map that is already resident in HBM (BASELINE.json metric: Part1+Part2
wall-clock and bins/s).

    python bench.py --gpus 1 --steps 3 --warmup 1            # 16,000-bin map (BASELINE configs[2])
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task prompt) with two extra objects:
  roofline     - the kernel family that took the most device time in the timed region, its
                 algorithmic bytes per launch / average launch duration (HIP events recorded on the
                 library's own stream, hicmi_timing_*), against the 8 TB/s HBM peak;
  cpu_baseline - the CPU oracle (a port of the reference's NumPy/SciPy path, oracle/) timed on this
                 box's host cores on a smaller map of the same generator and settings.
N > 1: every rank processes its own map (independent genomes, no data-path collective): weak scaling.
"""
import argparse
import contextlib
import glob
import io
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def make_bins(lay, Bin):
    ids = lay.bin_ids
    return [Bin(int(ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], int(lay.start[k]), int(lay.stop[k]), 1.0, 0.)
            for k in range(lay.n_bins)]


def write_sizes(lay, path):
    with open(path, "w") as fh:
        for name, size in zip(lay.scaffold_names, lay.scaffold_sizes_bp):
            fh.write("%s\t%d\n" % (name, size))


def cpu_baseline(sample_bins, n_scaffolds, scan_scaffolds, work):
    """Time the CPU oracle on a bounded sample (resident matrix in, files out - same boundary as the GPU step)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hic_oracle as orc
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(sample_bins, seed=1)
    c = synth.dense_contacts(lay, seed=1, sinkhorn_iters=12)
    bins = make_bins(lay, orc.Bin)
    sizes = os.path.join(work, "cpu.sizes")
    write_sizes(lay, sizes)
    f = lambda k: os.path.join(work, "cpu_" + k)  # noqa: E731
    orc.lib()
    t0 = time.time()
    orc.run_part1(None, None, None, sizes, f("dendro"), f("bingroups"), f("assess"), f("chromgroups"),
                  min_size=5, modularity=0.0, psig=.05, preloaded=(c, bins))
    t1 = time.time()
    orc.run_part2(None, None, None, f("chromgroups"), f("orders"), f("plotorder"), n_scaffolds=n_scaffolds,
                  scan_scaffolds=scan_scaffolds, preloaded=(c, make_bins(lay, orc.Bin)))
    t2 = time.time()
    return dict(value=sample_bins / (t2 - t0), unit="bins/s", cores=1, kind="port",
                sample="%d-bin synthetic map, same generator and settings (minSize 5, modularity 0, nScaffolds %d, "
                       "scanScaffolds %d); part1 %.1f s + part2 %.1f s on 1 of %d host cores; NumPy/SciPy oracle "
                       "without the reference's unused frozen-distribution construction (S2C:364)"
                       % (sample_bins, n_scaffolds, scan_scaffolds, t1 - t0, t2 - t1, os.cpu_count() or 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=16000)
    ap.add_argument("--n-scaffolds", type=int, default=6)
    ap.add_argument("--scan-scaffolds", type=int, default=5)
    ap.add_argument("--part1-only", action="store_true", help="BASELINE configs[1]: clustering + cuts only")
    ap.add_argument("--cpu-sample-bins", type=int, default=1500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-map", action="store_true",
                    help="N > 1: ONE map on all ranks (strong scaling) - Part 1 replicated, Part 2's chromosomes dealt "
                         "to the ranks, orders all-gathered; the default is one independent map per rank (weak)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearse the multi-rank flow with several ranks on one GPU")
    ap.add_argument("--kernel-times", choices=["part1", "all"], default="part1",
                    help="which kernel families get HIP-event timing inside the timed region")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if args.backend == "gloo":                          # rehearsal: more ranks than GPUs share the cards
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from hic_genome_assembler_amd import _lib, dist, synth
    dist.init(args.backend, device=dev)                 # RCCL; no-op for one process
    one_map = args.one_map and world > 1
    shard = (rank, world) if one_map else None
    map_seed = 1 if one_map else 1 + rank
    reduce_dev = dev if args.backend == "nccl" else None
    from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
    from hic_genome_assembler_amd.hostio import Bin

    n = args.bins
    lay = synth.make_layout(n, seed=map_seed)
    contacts = synth.dense_contacts_torch(lay, dev, seed=map_seed, sinkhorn_iters=12)
    torch.cuda.synchronize()
    work = tempfile.mkdtemp(prefix="hicbench_r%d_" % rank)
    sizes = os.path.join(work, "synth.sizes")
    write_sizes(lay, sizes)
    f = lambda k: os.path.join(work, k)  # noqa: E731
    ctx = _lib.Context(local)
    last = {}
    bin_objects = make_bins(lay, Bin)                   # the .bed metadata, parsed once like the matrix

    def step():
        ctx.set_contacts_device(contacts.data_ptr(), n, keepalive=contacts)
        dm = p1.DeviceMatrix(ctx)
        with contextlib.redirect_stdout(io.StringIO()):
            ta = time.perf_counter()
            cuts = p1.runResident(dm, list(bin_objects), sizes, f("dendrogramOrder.txt"), f("binGroups.txt"),
                                  f("assessment.txt"), f("chromosomeGroups.txt"), 5, 0.0, .05)
            last["part1_s"] = time.perf_counter() - ta
            if not args.part1_only:
                p2.runResident(p2.GenomeMatrix(ctx), dm.kept_bins, f("chromosomeGroups.txt"),
                               f("chromosomeOrders.txt"), f("plotOrder.txt"), args.n_scaffolds, args.scan_scaffolds,
                               lay.resolution, shard=shard)
        last["part2_s"] = time.perf_counter() - ta - last["part1_s"]
        last["cuts"] = cuts

    barrier = dist.barrier

    for _ in range(args.warmup):
        step()
    # HIP events around the families that decide the roofline line (nn-chain, row sort, ...).  --kernel-times all
    # also brackets the hundreds of small launches of the scans and of Part 2, which costs about 10 ms per map.
    ctx.timing_enable(0 if os.environ.get("HICMI_BENCH_NO_TIMING") else (1 if args.kernel_times == "all" else 2))
    ctx.timing_reset()
    barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, device=reduce_dev)
    timing = ctx.timing()
    ctx.timing_enable(False)

    if rank == 0:
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        value = (1 if one_map else world) * n / (ms_per_step / 1e3)
        # Part 2 families are summed over the concurrent worker streams (and the queued insertion is timed as
        # one region per chromosome), so their wall-clock share is about 1/workers of the sum
        workers = max(1, min(p2.WORKERS, 8))
        fam = max(timing, key=lambda k: timing[k]["ms"] / (workers if k.startswith("p2_") else 1))
        d = timing[fam]
        avg_ms = d["ms"] / max(d["launches"], 1)
        bytes_per_launch = d["bytes"] / max(d["launches"], 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this very command
        # (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read stream),
        # stored under profiles/: PMC collection cannot run inside the timed process.
        traffic, traffic_src = None, None
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_part1_16k.json")))
        pmc_file = pmc_files[-1] if pmc_files else ""             # the latest committed collection
        pmc_kernel = {"nnchain": "hicmi::k_nn_epoch<false>", "sort_rows": "hicmi::k_sort_rows_rb"}.get(fam)
        if n == 16000 and pmc_kernel and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            f_kb = pmc["FETCH_SIZE"].get(pmc_kernel)
            w_kb = pmc["WRITE_SIZE"].get(pmc_kernel)
            if f_kb and w_kb:
                traffic = (2.0 * f_kb["sum_KB"] + w_kb["sum_KB"]) * 1024.0 / max(f_kb["dispatches"], 1)
                traffic_src = ("profiles/%s (rocprofv3 --pmc passes, 2*FETCH_SIZE + WRITE_SIZE per dispatch)"
                               % os.path.basename(pmc_file))
        out = {
            "metric": "Part1+Part2 wall-clock (s) and bins/s on N x N contact map" if not args.part1_only
                      else "Part1 wall-clock (s) and bins/s on N x N contact map",
            "value": value, "unit": "bins/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if one_map else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("%d-bin synthetic ICE-balanced map, %s on 1xMI355X per rank "
                                    "(BASELINE.json configs[%d]), contacts resident in HBM"
                                    % (n, "-part1 only" if args.part1_only else "full -part1 -part2",
                                       1 if args.part1_only else 2)),
                       "bins": n, "chromosomes_planted": int(lay.chrom_of_bin.max()) + 1,
                       "scaffolds": len(lay.scaffold_names), "cuts_found": len(last.get("cuts", [])),
                       "minSize": 5, "modularity": 0, "psig": 0.05, "nScaffolds": args.n_scaffolds,
                       "scanScaffolds": args.scan_scaffolds, "wall_clock_s": ms_per_step / 1e3,
                       "last_step_part1_s": round(last.get("part1_s", 0.0), 4),
                       "last_step_part2_s": round(last.get("part2_s", 0.0), 4),
                       "part2_workers": p2.WORKERS,
                       "parallelism": ("one map over %d GPUs: Part 1 replicated, Part 2 chromosomes dealt to the ranks, "
                                       "object all-gather of the orders" % world) if one_map
                                      else "1 map per GPU, no collective" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": fam, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": avg_ms, "launches_per_step": d["launches"] / max(args.steps, 1),
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "kernels_ms_per_step": {k: round(v["ms"] / max(args.steps, 1), 3) for k, v in timing.items()
                                    if v["ms"] > 0 or args.kernel_times == "all"},
        }
        if not args.no_cpu_baseline and world == 1:
            with contextlib.redirect_stdout(io.StringIO()):
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_bins, args.n_scaffolds, args.scan_scaffolds, work)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    ctx.close()
    shutil.rmtree(work, ignore_errors=True)
    if world > 1:
        dist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
