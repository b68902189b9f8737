#!/usr/bin/env python3
"""Benchmark of the Hi-C hot path on MI355X: one "step" = -part1 -part2 on a synthetic
ICE-balanced contact map that is already resident in HBM (BASELINE.json metric: Part1+Part2
wall-clock and bins/s).

    python bench.py --gpus 1 --steps 3 --warmup 1            # 16,000-bin map (BASELINE configs[2])
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task prompt) with these extra objects:
  roofline       - the kernel family that took the most device time in the timed region: its ALGORITHMIC bytes per
                   launch (for the nn-chain the kernels count the columns their row scans visit) / its
                   average launch duration (HIP events on the library's own stream, hicmi_timing_*), against the
                   8 TB/s HBM peak; `traffic` = HBM bytes per launch from committed rocprofv3 --pmc passes;
  roofline_all   - the same figure for every Part 1 kernel family (one extra, untimed step with events around every
                   launch), so the HBM-bound kernels are visible next to the latency-bound chain;
  north_star_32k - the same step on a 32,000-bin map (north_star's single-GPU target size), a few steps;
  configs4_64k_f32 - the same step on BASELINE configs[4]'s map (64,000 bins, fp32-valued contacts) on this one GPU;
  mfma           - the matrix-core contraction of Part 2 (k_win_outside_mfma, v_mfma_f64_16x16x4_f64): flops counted by
                   the library / its device time from this run, instruction counters from the latest committed
                   rocprofv3 --pmc collection, against the 78.6 TF FP64-matrix peak;
  e2e            - the drop-in CLI from files to files: 2,000 bins from HiC-Pro text, 16,000 bins from the binary matrix
                   cache (HICMI_MATRIX_CACHE, warm .npy);
  cpu_baseline   - the CPU oracle (a port of the reference's NumPy/SciPy path, oracle/) timed on this box's host
                   cores on BASELINE configs[0] (2,000 bins), with and without the reference's unused
                   frozen-distribution construction (scaffoldToChromosomes.py:364).
N > 1: ONE map over all ranks (strong scaling): Part 2's chromosomes are dealt to the ranks, Part 1 runs on every rank
(its chain does not shard and, since its scans take their decisions on the device, sharding the per-row stages costs
more in per-scan all-gathers than it saves: DESIGN.md section 7); --shard-part1 row-shards those stages all the same
(hic_genome_assembler_amd/dist.py); --weak runs one independent map per rank instead.
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
PMC_KERNELS = {"nnchain": ("hicmi::k_nn_epoch", ), "sort_rows": ("hicmi::k_sort_rows_rb<false>", "hicmi::k_sort_rows_radix"),
               "presort_rows": ("hicmi::k_sort_rows_rb<true>", ), "rank_relabel": ("hicmi::k_rank_relabel", ),
               "rank_rows_tied": ("hicmi::k_rank_rows_tied", ), "row_sums": ("hicmi::k_row_sums", ),
               "build_w": ("hicmi::k_build_w", ), "rank_invert": ("hicmi::k_rank_invert", ), "cut_count": ("hicmi::k_cut", )}


def make_bins(lay, Bin):
    ids = lay.bin_ids
    return [Bin(int(ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], int(lay.start[k]), int(lay.stop[k]), 1.0, 0.)
            for k in range(lay.n_bins)]


def write_sizes(lay, path):
    with open(path, "w") as fh:
        for name, size in zip(lay.scaffold_names, lay.scaffold_sizes_bp):
            fh.write("%s\t%d\n" % (name, size))


def workload_label(n, part1_only, f32=False):
    what = "-part1 only" if part1_only else "full -part1 -part2"
    if n == 16000:
        cfg = "BASELINE.json configs[%d]" % (1 if part1_only else 2)
    elif n == 32000:
        cfg = "the map of BASELINE.json configs[3] on ONE GPU (north_star's single-GPU target size)"
    elif n == 64000:
        cfg = "the map of BASELINE.json configs[4] on ONE GPU, %s contacts" % ("fp32" if f32 else "fp64")
    elif n == 2000:
        cfg = "the map of BASELINE.json configs[0]"
    else:
        cfg = "not a BASELINE.json size"
    return "%d-bin synthetic ICE-balanced map, %s (%s), contacts resident in HBM" % (n, what, cfg)


def cpu_baseline(sample_bins, n_scaffolds, scan_scaffolds, work):
    """the CPU oracle on configs[0] (resident matrix in, files out - the same boundary as the GPU
    step), one core; Part 1 with and without the reference's frozen-distribution construction (S2C:364), whose cost is
    measured on a sample of the argument tuples the run really evaluated and scaled to their number."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import hic_oracle as orc
    from scipy.stats import hypergeom
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(sample_bins, seed=1)
    c = synth.dense_contacts(lay, seed=1, sinkhorn_iters=12)
    bins = make_bins(lay, orc.Bin)
    sizes = os.path.join(work, "cpu.sizes")
    write_sizes(lay, sizes)
    f = lambda k: os.path.join(work, "cpu_" + k)  # noqa: E731
    orc.lib()
    calls = {"n": 0, "args": []}
    plain = orc.hyper_geom

    def counting(x, M, n, N):
        xs = np.atleast_1d(np.asarray(x))
        calls["n"] += xs.size
        if len(calls["args"]) < 4000:
            ns, Ns = np.broadcast_to(np.asarray(n), xs.shape), np.broadcast_to(np.asarray(N), xs.shape)
            for k in range(0, xs.size, max(1, xs.size // 8)):
                calls["args"].append((int(M), int(ns.flat[k]), int(Ns.flat[k])))
        return plain(x, M, n, N)
    orc.hyper_geom = counting
    try:
        t0 = time.time()
        orc.run_part1(None, None, None, sizes, f("dendro"), f("bingroups"), f("assess"), f("chromgroups"),
                      min_size=5, modularity=0.0, psig=.05, preloaded=(c, bins))
        t1 = time.time()
    finally:
        orc.hyper_geom = plain
    orc.run_part2(None, None, None, f("chromgroups"), f("orders"), f("plotorder"), n_scaffolds=n_scaffolds,
                  scan_scaffolds=scan_scaffolds, preloaded=(c, make_bins(lay, orc.Bin)))
    t2 = time.time()
    sample = calls["args"][:3000]
    ta = time.time()
    with np.errstate(all="ignore"):
        for (M, n, N) in sample:
            hypergeom(M, n, N)                                   # S2C:364: built per call, never used
    frozen_each = (time.time() - ta) / max(len(sample), 1)
    frozen_total = frozen_each * calls["n"]
    return dict(value=sample_bins / (t2 - t0), unit="bins/s", cores=1, kind="port",
                value_with_frozen_object=sample_bins / (t2 - t0 + frozen_total),
                part1_s=round(t1 - t0, 2), part2_s=round(t2 - t1, 2), frozen_object_s=round(frozen_total, 2),
                hyper_geom_evaluations=calls["n"],
                sample="BASELINE.json configs[0]: %d-bin synthetic map, same generator and settings (minSize 5, modularity 0, "
                       "nScaffolds %d, scanScaffolds %d), resident matrix in / files out, 1 of %d host cores; NumPy/SciPy "
                       "oracle.  `value` is without the reference's unused frozen-distribution construction (S2C:364); "
                       "`value_with_frozen_object` adds it: %.2f ms per construction (measured on %d of the run's own "
                       "argument tuples) x %d evaluations (the reference's futile window retries, S2C:499-508, excluded)"
                       % (sample_bins, n_scaffolds, scan_scaffolds, os.cpu_count() or 0, frozen_each * 1e3, len(sample),
                          calls["n"]))


def e2e_cli(n_bins, work):
    """Text in, six files out through the drop-in CLI (`run_hicAssembler.py -part1 -part2`, plots off) on a small map:
    what a user of the reference's command line pays besides the resident hot path - HiC-Pro text parsing (once: the matrix
    stays in HBM for Part 2), upload, both parts, files.  16,000 bins are not run here: writing that .matrix file (128 M
    triplets) takes minutes; its parse alone is 3.4 s."""
    from hic_genome_assembler_amd import run_hicAssembler, synth
    lay = synth.make_layout(n_bins, seed=1)
    c = synth.dense_contacts(lay, seed=1, sinkhorn_iters=12)
    paths = synth.write_hicpro(os.path.join(work, "in"), lay, c, "e2e")
    cfg = synth.write_config(os.path.join(work, "config.txt"), paths, os.path.join(work, "out"), os.path.join(work, "plots"),
                             lay.resolution, min_size=5, modularity=0.0, psig=0.05, n_scaffolds=6, scan_scaffolds=5)
    os.environ["HICMI_NO_PLOTS"] = "1"
    times = []
    for _ in range(2):                                     # second run: page cache warm, HIP modules loaded
        t0 = time.time()
        run_hicAssembler.main(["-part1", "-part2", "-c", cfg])
        times.append(time.time() - t0)
    size = os.path.getsize(paths["hicProMatrixFile"])
    return {"bins": n_bins, "seconds_first_run": round(times[0], 3), "seconds_second_run": round(times[1], 3),
            "matrix_text_MB": round(size / 1e6, 1),
            "what": "run_hicAssembler.py -part1 -part2 -config: HiC-Pro text -> six files, plots off, matrix parsed once"}


def e2e_cli_cached(n_bins, work, dev):
    """The same CLI at 16,000 bins with the matrix read from its binary cache (HICMI_MATRIX_CACHE: hostio.py writes a .npy
    copy beside the HiC-Pro text the first time it parses it; S2C:70-98 / OG:30-93 replaced).  Writing the 128 M-line text of a
    16k map takes minutes, so the cache is filled directly here: a one-line placeholder .matrix whose size / mtime the cache
    key names, the synthetic map as the .npy.  Timed: warm .npy (memory-mapped) -> upload -> both parts -> six files."""
    import numpy as np
    from hic_genome_assembler_amd import hostio, run_hicAssembler, synth
    lay = synth.make_layout(n_bins, seed=1)
    c = synth.dense_contacts_torch(lay, dev, seed=1, sinkhorn_iters=12).cpu().numpy()
    paths = synth.write_hicpro(os.path.join(work, "in"), lay, None, "e2e16k")
    npy, key_file = hostio._cache_paths(paths["hicProMatrixFile"], "1")
    np.save(npy, c, allow_pickle=False)
    with open(key_file, "w") as fh:
        json.dump(hostio._cache_key(paths["hicProMatrixFile"], lay.bin_ids), fh)
    del c
    cfg = synth.write_config(os.path.join(work, "config.txt"), paths, os.path.join(work, "out"), os.path.join(work, "plots"),
                             lay.resolution, min_size=5, modularity=0.0, psig=0.05, n_scaffolds=6, scan_scaffolds=5)
    os.environ["HICMI_NO_PLOTS"] = "1"
    os.environ["HICMI_MATRIX_CACHE"] = "1"
    times = []
    try:
        for _ in range(2):
            t0 = time.time()
            run_hicAssembler.main(["-part1", "-part2", "-c", cfg])
            times.append(time.time() - t0)
    finally:
        del os.environ["HICMI_MATRIX_CACHE"]
    return {"bins": n_bins, "seconds_first_run": round(times[0], 3), "seconds_second_run": round(times[1], 3),
            "matrix_cache_MB": round(os.path.getsize(npy) / 1e6, 1),
            "what": "run_hicAssembler.py -part1 -part2 -config with HICMI_MATRIX_CACHE=1: warm .npy (no text parse) -> six files, "
                    "plots off; the text parse of such a map (128 M triplets) is 3.4 s once (DESIGN.md 8b)"}


def pmc_traffic(n, fam, launches_per_step):
    """HBM bytes per launch of a kernel family from the latest committed rocprofv3 --pmc collection for this map size
    (profiles/r*_pmc_part1_<n/1000>k.json, ONE Part 1 pass; 2 x FETCH_SIZE + WRITE_SIZE: MI355X_MICROARCH.md, FETCH_SIZE
    counts half of a wide coalesced read stream on gfx950), summed over every kernel of the family and divided by the
    family's launches per step.  PMC collection cannot run inside the timed process, so the figure goes stale when a
    kernel changes after the collection: the source string names the file and the commit it was collected at."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_part1_%dk.json" % (n // 1000))))
    if not files or fam not in PMC_KERNELS or launches_per_step <= 0:
        return None, None
    with open(files[-1]) as fh:
        pmc = json.load(fh)
    fetch = write = 0.0
    seen = False
    for name, d in pmc.get("FETCH_SIZE", {}).items():
        if name.startswith(PMC_KERNELS[fam]):
            fetch += d["sum_KB"]
            seen = True
    for name, d in pmc.get("WRITE_SIZE", {}).items():
        if name.startswith(PMC_KERNELS[fam]):
            write += d["sum_KB"]
    if not seen:
        return None, None
    commit = pmc.get("_meta", {}).get("commit", "not recorded")
    return ((2.0 * fetch + write) * 1024.0 / launches_per_step,
            "profiles/%s (rocprofv3 --pmc passes over one Part 1 pass, 2*FETCH_SIZE + WRITE_SIZE summed over the family's "
            "kernels / its launches per step; collected at commit %s)" % (os.path.basename(files[-1]), commit))


FP64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md / SURVEY 8d: FP64 vector = FP64 matrix


def family_table(timing, steps, n, workers):
    """Every kernel family with its algorithmic bytes per launch against the HBM peak.  Part 2 families run on
    `workers` concurrent streams (and the lock-step insertion as one region per chromosome group): their `ms_per_step` is
    device time SUMMED over the streams, their share of the wall clock about 1/workers of it."""
    out = {}
    for k, v in timing.items():
        if v["launches"] == 0 or v["ms"] <= 0 or v["bytes"] <= 0 or k == "plot" or k.endswith("_flops"):
            continue
        avg_ms = v["ms"] / v["launches"]
        gbs = v["bytes"] / v["launches"] / (avg_ms * 1e-3) / 1e9
        lps = v["launches"] / steps
        traffic, _src = pmc_traffic(n, k, lps)
        out[k] = {"ms_per_step": round(v["ms"] / steps, 3), "launches_per_step": lps,
                  "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": v["bytes"] / v["launches"],
                  "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic}
        if k.startswith("p2_"):
            out[k]["summed_over_streams"] = workers
    fl = timing.get("p2_window_G_flops")
    if fl and fl["bytes"] > 0 and "p2_window_G" in out and timing["p2_window_G"]["ms"] > 0:
        tf = fl["bytes"] / (timing["p2_window_G"]["ms"] * 1e-3) / 1e12
        out["p2_window_G"].update({"bound": "mfma (the outside table is a GEMM on v_mfma_f64_16x16x4_f64; the family's time also "
                                            "holds the pair tables and the candidate look-ups)",
                                   "gemm_flops_per_step": fl["bytes"] / steps, "achieved_TFLOPs": round(tf, 2),
                                   "frac_fp64_matrix_peak": round(tf / FP64_MATRIX_PEAK_TFLOPS, 4)})
    return out


def mfma_object(timing, steps):
    """north_star: "MFMA-utilisation counters reported against chip peak".  The only matrix-core kernel of the path is Part
    2's k_win_outside_mfma (the reference has no row x row^T contraction; the results-neutral GEMM is
    the window tables' outside term).  Flops: counted by the library (2 m^2 (n - m) per window); time: the window-table
    family of this run (all its kernels, summed over the worker streams); instruction counters: the latest committed
    rocprofv3 --pmc collection (they cannot be read inside the timed process)."""
    out = {"kernel": "hicmi::k_win_outside_mfma", "instruction": "v_mfma_f64_16x16x4_f64", "peak_TFLOPs": FP64_MATRIX_PEAK_TFLOPS}
    fl, fam = timing.get("p2_window_G_flops"), timing.get("p2_window_G")
    if fl and fam and fam["ms"] > 0:
        tf = fl["bytes"] / (fam["ms"] * 1e-3) / 1e12
        out.update({"gemm_flops_per_step": fl["bytes"] / steps, "window_table_family_ms_per_step": round(fam["ms"] / steps, 3),
                    "achieved_TFLOPs_over_the_family": round(tf, 2), "frac_of_peak_over_the_family": round(tf / FP64_MATRIX_PEAK_TFLOPS, 4)})
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma_16k.json")))
    if files:
        with open(files[-1]) as fh:
            pmc = json.load(fh)

        def total(counter):
            return sum(d["sum"] for name, d in pmc.get(counter, {}).items() if name.startswith("hicmi::k_win_outside_mfma"))
        inst, mops, busy, sq = (total(c) for c in ("SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU_MFMA_MOPS_F64",
                                                   "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"))
        ksec = pmc.get("_meta", {}).get("k_win_outside_mfma_ms_per_step")
        out["counters"] = {"source": "profiles/%s (one 16,000-bin step; collected at commit %s)"
                                     % (os.path.basename(files[-1]), pmc.get("_meta", {}).get("commit", "not recorded")),
                           "SQ_INSTS_VALU_MFMA_F64": inst, "GFLOP_per_step": mops * 512.0 / 1e9,
                           "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_BUSY_CYCLES_of_the_kernel": sq,
                           "kernel_ms_per_step_in_that_collection": ksec,
                           "frac_of_peak_of_the_kernel_alone": (round(mops * 512.0 / 1e12 / (ksec * 1e-3) / FP64_MATRIX_PEAK_TFLOPS, 4)
                                                                if ksec else None)}
    return out


class Job:
    """One resident map and the contexts that process it."""

    def __init__(self, args, n, dev, local, seed, shard, f32=False):
        import torch
        from hic_genome_assembler_amd import _lib, synth
        from hic_genome_assembler_amd.hostio import Bin
        self.args, self.n, self.shard = args, n, shard
        self.shard_p1 = shard                              # Part 1's row shard (bench default: None - Part 1 on every rank)
        self.lay = synth.make_layout(n, seed=seed)
        self.contacts = synth.dense_contacts_torch(self.lay, dev, seed=seed, sinkhorn_iters=12)
        if f32:                                            # configs[4]: contacts stored as fp32 (exactly representable values)
            self.contacts = self.contacts.to(torch.float32).to(torch.float64)
        torch.cuda.synchronize()
        self.work = tempfile.mkdtemp(prefix="hicbench_")
        self.sizes = os.path.join(self.work, "synth.sizes")
        write_sizes(self.lay, self.sizes)
        self.ctx = _lib.Context(local)
        self.bins = make_bins(self.lay, Bin)               # the .bed metadata, parsed once like the matrix
        self.last = {}
        self.parts = [0.0, 0.0, 0]                         # host seconds in Part 1 / Part 2 and steps, since the timed region began

    def f(self, k):
        return os.path.join(self.work, k)

    def step(self):
        from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
        a, f = self.args, self.f
        self.ctx.set_contacts_device(self.contacts.data_ptr(), self.n, keepalive=self.contacts)
        dm = p1.DeviceMatrix(self.ctx)
        with contextlib.redirect_stdout(io.StringIO()):
            ta = time.perf_counter()
            # Part 1's four text files are written by a background thread while the device works on (and while Part 2
            # starts from the in-memory groups); finish_files() below is inside the timed step: all six files are on disk
            # when it returns
            cuts = p1.runResident(dm, list(self.bins), self.sizes, f("dendrogramOrder.txt"), f("binGroups.txt"),
                                  f("assessment.txt"), f("chromosomeGroups.txt"), 5, 0.0, .05, shard=self.shard_p1,
                                  overlap_files=True)
            self.last["part1_s"] = time.perf_counter() - ta
            if not a.part1_only:
                p2.runResident(p2.GenomeMatrix(self.ctx), dm.kept_bins, f("chromosomeGroups.txt"),
                               f("chromosomeOrders.txt"), f("plotOrder.txt"), a.n_scaffolds, a.scan_scaffolds,
                               self.lay.resolution, shard=self.shard, chromosomeList=dm.chromosome_groups,
                               on_native_phase=dm.release_files)
            tf = time.perf_counter()
            dm.finish_files()
            if os.environ.get("HICMI_PART2_PROFILE"):
                sys.stderr.write("[bench] waited %.1f ms for Part 1's background files\n" % ((time.perf_counter() - tf) * 1e3))
            if self.shard is not None and not a.part1_only:
                groups = dm.chromosome_groups
                mine = p2.chromosomesOfRank(groups, self.shard[0], self.shard[1])
                self.last["my_chromosomes"] = [int(i) for i in mine]
                self.last["my_load"] = int(sum(len(groups[i]) ** 2 for i in mine))
        self.last["part2_s"] = time.perf_counter() - ta - self.last["part1_s"]
        self.last["cuts"] = cuts
        self.parts[0] += self.last["part1_s"]; self.parts[1] += self.last["part2_s"]; self.parts[2] += 1

    def close(self):
        self.ctx.close()
        shutil.rmtree(self.work, ignore_errors=True)


def timed_run(job, steps, warmup, timing_mode, barrier, reduce_dev):
    import torch
    from hic_genome_assembler_amd import dist
    for _ in range(warmup):
        job.step()
    job.ctx.timing_enable(timing_mode)
    job.ctx.timing_reset()
    job.parts = [0.0, 0.0, 0]
    barrier()
    torch.cuda.synchronize()
    job.ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    job.ctx.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, device=reduce_dev)
    timing = job.ctx.timing()
    stats = job.ctx.nnchain_stats()
    job.ctx.timing_enable(False)
    return elapsed, timing, stats


def roofline_of(timing, stats, steps, n, workers):
    # Part 2 families are summed over the concurrent worker streams (and the queued insertion is timed as
    # one region per chromosome), so their wall-clock share is about 1/workers of the sum
    fam = max(timing, key=lambda k: timing[k]["ms"] / (workers if k.startswith("p2_") else 1))
    d = timing[fam]
    avg_ms = d["ms"] / max(d["launches"], 1)
    bytes_per_launch = d["bytes"] / max(d["launches"], 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(n, fam, d["launches"] / max(steps, 1))
    out = {"bound": "hbm", "kernel": fam, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
           "avg_launch_ms": avg_ms, "launches_per_step": d["launches"] / max(steps, 1),
           "algorithmic_bytes_per_launch": bytes_per_launch}
    if fam == "nnchain" and stats["merges"] > 0:
        nominal = 3.5 * n * n * 8.0 * steps                 # SURVEY 8d's nominal figure for SciPy's loop (~3 scans per merge)
        out.update({"definition": "8 B x (columns visited by the row scans the kernels really ran + 3 x live columns per "
                                  "merge), counted by the kernels (SURVEY 8d); latency-bound: the meaningful rate is merges/s",
                    "merges_per_s": stats["merges"] / (d["ms"] * 1e-3) if d["ms"] > 0 else None,
                    "row_scans_per_merge": stats["scans"] / stats["merges"],
                    "chain_steps_from_neighbour_cache_per_merge": stats["cache_hits"] / stats["merges"],
                    "achieved_nominal_3.5N2": nominal / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else None,
                    "frac_nominal_3.5N2": nominal / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if d["ms"] > 0 else None})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bins", type=int, default=16000)
    ap.add_argument("--n-scaffolds", type=int, default=6)
    ap.add_argument("--scan-scaffolds", type=int, default=5)
    ap.add_argument("--part1-only", action="store_true", help="BASELINE configs[1]: clustering + cuts only")
    ap.add_argument("--f32", action="store_true", help="contacts rounded to fp32 values (BASELINE configs[4] stores fp32)")
    ap.add_argument("--cpu-sample-bins", type=int, default=2000, help="BASELINE configs[0]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-32k", action="store_true", help="skip the north_star_32k object")
    ap.add_argument("--no-64k", action="store_true", help="skip the configs4_64k_f32 object")
    ap.add_argument("--no-table", action="store_true", help="skip the extra untimed step behind roofline_all")
    ap.add_argument("--no-e2e", action="store_true", help="skip the text-to-files run of the CLI at 2,000 bins")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: one independent map per rank (weak scaling) instead of ONE map over all ranks")
    ap.add_argument("--one-map", action="store_true", help="(default for N > 1; kept for earlier command lines)")
    ap.add_argument("--shard-part1", action="store_true",
                    help="N > 1, one map: also row-shard Part 1's row sums / rank rows / scan counts (one all-gather per scan)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearse the multi-rank flow with several ranks on one GPU")
    ap.add_argument("--kernel-times", choices=["part1", "all"], default="part1",
                    help="which kernel families get HIP-event timing inside the timed region")
    args = ap.parse_args()
    if os.environ.get("HICMI_SWITCH_INTERVAL"):
        sys.setswitchinterval(float(os.environ["HICMI_SWITCH_INTERVAL"]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    n_devices = max(1, torch.cuda.device_count())
    if args.backend == "gloo":                          # rehearsal: more ranks than GPUs share the cards
        local %= n_devices
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from hic_genome_assembler_amd import dist
    dist.init(args.backend, device=dev)                 # RCCL; no-op for one process
    one_map = world > 1 and not args.weak
    shard = (rank, world) if one_map else None
    map_seed = 1 if (one_map or world == 1) else 1 + rank
    reduce_dev = dev if args.backend == "nccl" else None
    from hic_genome_assembler_amd import orderGenome as p2

    n = args.bins
    # Part 1's per-row stages are row-sharded over the ranks when asked for, and by themselves where they are worth an
    # all-gather per scan: maps whose every row holds equal similarities (fp32 contacts: k_rank_rows_tied is exposed after
    # the chain) and 48,000 bins or more (the counts of a scan are ~25 ms per map there) - DESIGN.md section 7
    shard_p1 = one_map and (args.shard_part1 or args.f32 or n >= 48000)
    job = Job(args, n, dev, local, map_seed, shard, f32=args.f32)
    if not shard_p1:
        job.shard_p1 = None
    timing_mode = 0 if os.environ.get("HICMI_BENCH_NO_TIMING") else (1 if args.kernel_times == "all" else 2)
    # HIP events around the families that decide the roofline line (nn-chain, row sort, ...).  --kernel-times all
    # also brackets the hundreds of small launches of the scans and of Part 2, which costs about 10 ms per map.
    elapsed, timing, stats = timed_run(job, args.steps, args.warmup, timing_mode, dist.barrier, reduce_dev)

    out = None
    if rank == 0:
        steps = max(args.steps, 1)
        ms_per_step = elapsed / steps * 1e3
        value = (1 if one_map else world) * n / (ms_per_step / 1e3)
        workers = max(1, min(p2.WORKERS, 8))
        lay, last = job.lay, job.last
        out = {
            "metric": "Part1+Part2 wall-clock (s) and bins/s on N x N contact map" if not args.part1_only
                      else "Part1 wall-clock (s) and bins/s on N x N contact map",
            "value": value, "unit": "bins/s", "n_gpus": world if args.backend == "nccl" else min(world, n_devices),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": ("strong" if one_map else "weak") if world > 1 else None,
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload_label(n, args.part1_only, args.f32),
                       "bins": n, "chromosomes_planted": int(lay.chrom_of_bin.max()) + 1,
                       "scaffolds": len(lay.scaffold_names), "cuts_found": len(last.get("cuts", [])),
                       "minSize": 5, "modularity": 0, "psig": 0.05, "nScaffolds": args.n_scaffolds,
                       "scanScaffolds": args.scan_scaffolds, "wall_clock_s": ms_per_step / 1e3,
                       "part1_s_per_step": round(job.parts[0] / max(job.parts[2], 1), 4),
                       "part2_s_per_step": round(job.parts[1] / max(job.parts[2], 1), 4),
                       "part2_workers": p2.WORKERS, "ranks": world,
                       "parallelism": (("ONE map over %d ranks: Part 1's row-independent stages (row sums, rank rows, cut and "
                                        "filter counts) row-sharded with an all-gather of the per-row flags per scan, UPGMA "
                                        "replicated, Part 2's chromosomes dealt to the ranks" % world) if shard_p1 else
                                       ("ONE map over %d ranks: Part 1 on every rank (no collective), Part 2's chromosomes "
                                        "dealt to the ranks, one all-gather of the ordered lists" % world)) if one_map
                                      else "1 independent map per rank, no collective" if world > 1 else "single GPU"},
            "roofline": roofline_of(timing, stats, steps, n, workers),
            "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in timing.items()
                                    if v["ms"] > 0 or args.kernel_times == "all"},
        }
    if world > 1:
        # the record explains its own curve: what every rank spent where, and how Part 2's chromosomes were dealt
        from hic_genome_assembler_amd import orderGenome as _p2
        mine = {"part1_s_per_step": round(job.parts[0] / max(job.parts[2], 1), 4),
                "part2_s_per_step": round(job.parts[1] / max(job.parts[2], 1), 4),
                "chromosomes": job.last.get("my_chromosomes"), "bins_squared_load": job.last.get("my_load")}
        per_rank = dist.gather_results({rank: mine})
        if rank == 0:
            import torch.distributed as tdist
            loads = [per_rank[r]["bins_squared_load"] or 0 for r in sorted(per_rank)]
            out["multi_gpu"] = {"rccl_ranks": tdist.get_world_size(), "backend": tdist.get_backend(),
                                "per_rank": {str(r): per_rank[r] for r in sorted(per_rank)},
                                "part2_load_imbalance_max_over_mean": (max(loads) / (sum(loads) / len(loads))) if sum(loads) else None,
                                "part1": "row-sharded (one all-gather of the owned flags per scan)" if shard_p1 else
                                         "replicated on every rank (the chain does not shard: DESIGN.md section 7)"}
    if world == 1 and not args.no_table:
        # one more step, untimed, with events around EVERY launch: the per-kernel roofline table
        _e, t_all, _s = timed_run(job, 1, 0, 1, dist.barrier, reduce_dev)
        out["roofline_all"] = family_table(t_all, 1, n, max(1, min(p2.WORKERS, 8)))
        if not args.part1_only:
            out["mfma"] = mfma_object(t_all, 1)
    job.close()
    del job
    torch.cuda.empty_cache()
    def side_job(bins, f32, k_steps, seed):
        """The same step on another BASELINE size: its own synthetic map, one warm-up step, k_steps timed steps."""
        j = Job(args, bins, dev, local, seed, None, f32=f32)
        e, t, st = timed_run(j, k_steps, 1, timing_mode, dist.barrier, reduce_dev)
        ms = e / k_steps * 1e3
        obj = {"workload": workload_label(bins, False, f32), "value": bins / (ms / 1e3), "unit": "bins/s",
               "ms_per_step": ms, "steps": k_steps, "warmup": 1, "cuts_found": len(j.last.get("cuts", [])),
               "part1_s_per_step": round(j.parts[0] / max(j.parts[2], 1), 4),
               "part2_s_per_step": round(j.parts[1] / max(j.parts[2], 1), 4),
               "roofline": roofline_of(t, st, k_steps, bins, max(1, min(p2.WORKERS, 8))),
               "kernels_ms_per_step": {k: round(v["ms"] / k_steps, 3) for k, v in t.items() if v["ms"] > 0}}
        j.close()
        del j
        torch.cuda.empty_cache()
        return obj

    if world == 1 and not args.no_32k and n != 32000 and not args.part1_only:
        # north_star's single-GPU target size, same step, a few repetitions (its own synthetic map)
        out["north_star_32k"] = side_job(32000, False, 3, 1)
    if world == 1 and not args.no_64k and n != 64000 and not args.part1_only:
        # BASELINE configs[4]'s map on ONE GPU: 64,000 bins, contacts rounded to fp32 values (every row then holds equal
        # similarities: the exposed part of the row sort is k_rank_rows_tied, reported in kernels_ms_per_step)
        out["configs4_64k_f32"] = side_job(64000, True, 2, 1)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            work = tempfile.mkdtemp(prefix="hicbench_cpu_")
            with contextlib.redirect_stdout(io.StringIO()):
                out["cpu_baseline"] = cpu_baseline(args.cpu_sample_bins, args.n_scaffolds, args.scan_scaffolds, work)
            shutil.rmtree(work, ignore_errors=True)
        else:
            out["cpu_baseline"] = None
        if not args.no_e2e and not args.no_cpu_baseline and world == 1:
            work = tempfile.mkdtemp(prefix="hicbench_e2e_")
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    out["e2e"] = e2e_cli(2000, work)
                    out["e2e_16k_cached"] = e2e_cli_cached(16000, work, dev)
            except SystemExit:
                out["e2e"] = None
            shutil.rmtree(work, ignore_errors=True)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
