#!/usr/bin/env python3
"""Generate tests/golden/<case>/ fixtures by RUNNING THE REFERENCE in this container.

TEST INFRASTRUCTURE.  Run as ``python oracle/gen_golden.py [case ...]`` from the repo
root in the build container (the only place /root/reference exists).  For every case it

1. builds a synthetic HiC-Pro dataset (hic_genome_assembler_amd/synth.py),
2. runs the reference's own ``scaffoldToChromosomes.runPipeline`` and
   ``orderGenome.runPipeline`` (imported through oracle/ref_shim.py) with passive
   recorders wrapped around a few of its functions, and
3. stores inputs-by-seed (+ sha256 of the contact matrix), the recorded intermediates
   (golden.npz) and the reference's output files (verbatim text) under tests/golden/<case>/.

Only data (inputs and expected outputs) is written; no reference source is copied.
"""
import contextlib
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from hic_genome_assembler_amd import synth  # noqa: E402
import ref_shim  # noqa: E402

CASES = {
    # name: dict(layout kwargs, run settings)
    "n160": dict(n=160, seed=3, n_chrom=4, mean_scaffold_bins=6.0, min_size=5, psig=0.05,
                 n_scaffolds=4, scan_scaffolds=3, store_matrices=True),
    "n300_edges": dict(n=300, seed=5, n_chrom=4, mean_scaffold_bins=7.0, min_size=4, psig=0.01,
                       n_scaffolds=5, scan_scaffolds=3, zero_bins=(17, 130, 131), nan_bias=(44, 250),
                       store_matrices=False),
    "n400_default": dict(n=400, seed=7, n_chrom=3, mean_scaffold_bins=9.0, min_size=5, psig=0.05,
                         n_scaffolds=6, scan_scaffolds=5, store_matrices=False),
    "n600": dict(n=600, seed=1, n_chrom=5, mean_scaffold_bins=13.0, min_size=5, psig=0.05,
                 n_scaffolds=4, scan_scaffolds=3, store_matrices=False),
    "n2000": dict(n=2000, seed=2, n_chrom=8, mean_scaffold_bins=13.0, min_size=5, psig=0.05,
                  n_scaffolds=4, scan_scaffolds=3, store_matrices=False),
    # what real HiC-Pro maps look like: 2-decimal values, 45 % exact zeros - ties in the clustering (SciPy's nn_chain
    # tie rule) and in every row of the argsort (NumPy's unstable sort: tie order undefined)
    "n500_sparse": dict(n=500, seed=11, n_chrom=4, mean_scaffold_bins=8.0, min_size=5, psig=0.05,
                        n_scaffolds=4, scan_scaffolds=3, sparse=(0.45, 2), store_matrices=False),
    # the cost loop as Numba compiles it: numpy.trace inside a nopython function is a sequential loop (numba's
    # np_trace), not NumPy's pairwise add.reduce - the shim patches orderGenome.numpyTrace (the name the jitted
    # function calls, OG:8,188) with that loop; the totals (OG:343,448,506) stay NumPy's
    "n160_numba": dict(n=160, seed=3, n_chrom=4, mean_scaffold_bins=6.0, min_size=5, psig=0.05,
                       n_scaffolds=4, scan_scaffolds=3, numba_trace="sequential", store_matrices=False),
}

OUTPUT_FILES = ["dendrogramOrder.txt", "binGroups.txt", "assessment.txt", "chromosomeGroups.txt",
                "chromosomeOrders.txt", "plotOrder.txt"]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_case(name, spec, work):
    lay = synth.make_layout(spec["n"], seed=spec["seed"], n_chrom=spec["n_chrom"],
                            mean_scaffold_bins=spec["mean_scaffold_bins"])
    c = synth.dense_contacts(lay, seed=spec["seed"])
    if spec.get("sparse"):
        c = synth.sparsify(c, *spec["sparse"])
    for b in spec.get("zero_bins", ()):            # bins without any contact: dropped by removeRows
        c[b, :] = 0.0
        c[:, b] = 0.0
    nan_ids = [int(lay.bin_ids[b]) for b in spec.get("nan_bias", ())]
    paths = synth.write_hicpro(os.path.join(work, "hicpro"), lay, c, nan_bias_bins=nan_ids)
    cfg = synth.write_config(os.path.join(work, "config.txt"), paths, os.path.join(work, "out"),
                             os.path.join(work, "plots"), lay.resolution, min_size=spec["min_size"],
                             modularity=0.0, psig=spec["psig"], n_scaffolds=spec["n_scaffolds"],
                             scan_scaffolds=spec["scan_scaffolds"])
    return lay, c, paths, cfg


def run_reference(spec, paths, out_dir, plot_dir):
    s2c, og = ref_shim.load()
    rec = {"hyper": [], "costs": [], "chrom_orders": []}
    numpy_trace = og.numpyTrace
    if spec.get("numba_trace") == "sequential":
        def sequential_trace(a, offset=0):        # numba/np/arraymath.py np_trace: ret = 0; for i in range(n): ret += a[i, k + i]
            rows, cols = a.shape
            n = max(min(rows, cols - offset), 0)
            ret = 0
            for i in range(n):
                ret += a[i, offset + i]
            return ret
        og.numpyTrace = sequential_trace

    # ---- passive recorders (call through, copy results) ----
    orig = {}

    def wrap(mod, fname, fn):
        orig[(mod, fname)] = getattr(mod, fname)
        setattr(mod, fname, fn)

    def convertMatrix(adj, binList, distance=True, similarity=False):
        out = orig[(s2c, "convertMatrix")](adj, binList, distance=distance, similarity=similarity)
        key = "D" if distance else "S"
        rec.setdefault(key + "_list", []).append(np.array(out, dtype=np.float64))
        return out

    def removeRows(matrix, binList, zeroRows=True, biasVals=False):
        m, b = orig[(s2c, "removeRows")](matrix, binList, zeroRows=zeroRows, biasVals=biasVals)
        rec["kept_ids"] = np.array([x.ID for x in b], dtype=np.int64)
        rec["row_sum_seq"] = np.array([float(x.rowSum) for x in b], dtype=np.float64)
        return m, b

    def average(y):
        z = orig[(s2c.scipy.cluster.hierarchy, "average")](y)
        rec["Z"] = np.array(z, dtype=np.float64)
        return z

    def hyper_geom(x, M, n, N):
        p = orig[(s2c, "hyper_geom")](x, M, n, N)
        rec["hyper"].append((int(x), int(M), int(n), int(N), float(p)))
        return p

    def pre_process(argsorted_mat, min_size=5, min_frac=.05, psig=.05):
        rec["argsorted"] = np.array(argsorted_mat, dtype=np.int64)
        rec["hyper_mark_first_pass"] = len(rec["hyper"])
        out = orig[(s2c, "pre_process_all_matrix_breakpoints")](argsorted_mat, min_size=min_size,
                                                                 min_frac=min_frac, psig=psig)
        rec["hyper_mark_filter"] = len(rec["hyper"])
        rec["initial_cuts"] = np.array([int(v) for v in out], dtype=np.int64)
        return out

    def filt(argsorted_mat, original_inds, psig=.05):
        out = orig[(s2c, "filter_noisy_breakpoints")](argsorted_mat, original_inds, psig=psig)
        rec["filtered_cuts"] = np.array([int(v) for v in out], dtype=np.int64)
        return out

    def cost_numba(matrix, total):
        c = orig[(og, "costFunction_numba")](matrix, total)
        rec["costs"].append(float(c))
        return c

    def orderChromosome(chromGroup, matrix, binList, nScaffolds=6, scanScaffolds=5):
        rec.setdefault("cost_marks", []).append(len(rec["costs"]))
        out = orig[(og, "orderChromosome")](chromGroup, matrix, binList, nScaffolds=nScaffolds,
                                            scanScaffolds=scanScaffolds)
        rec["chrom_orders"].append([(s.name, s.orientation) for s in out])
        return out

    wrap(s2c, "convertMatrix", convertMatrix)
    wrap(s2c, "removeRows", removeRows)
    wrap(s2c.scipy.cluster.hierarchy, "average", average)
    wrap(s2c, "hyper_geom", hyper_geom)
    wrap(s2c, "pre_process_all_matrix_breakpoints", pre_process)
    wrap(s2c, "filter_noisy_breakpoints", filt)
    wrap(og, "costFunction_numba", cost_numba)
    wrap(og, "orderChromosome", orderChromosome)

    f = lambda k: os.path.join(out_dir, k)  # noqa: E731
    log = io.StringIO()
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(log):
            s2c.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                            paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"),
                            os.path.join(plot_dir, "a.png"), os.path.join(plot_dir, "b.png"),
                            f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                            True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000)
            t1 = time.time()
            og.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                           f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), plot_dir, "synthetic",
                           os.path.join(plot_dir, "g.png"), "synthetic genome", f("plotOrder.txt"),
                           spec["n_scaffolds"], spec["scan_scaffolds"], 100000)
            t2 = time.time()
    finally:
        for (mod, fname), fn in orig.items():
            setattr(mod, fname, fn)
        og.numpyTrace = numpy_trace
    rec["seconds_part1"] = t1 - t0
    rec["seconds_part2"] = t2 - t1
    return rec


def main(argv):
    names = argv or list(CASES)
    for name in names:
        spec = CASES[name]
        gold = os.path.join(ROOT, "tests", "golden", name)
        work = tempfile.mkdtemp(prefix="hicgold_")
        try:
            print("[%s] building dataset" % name, flush=True)
            lay, c, paths, _cfg = build_case(name, spec, work)
            out_dir, plot_dir = os.path.join(work, "out"), os.path.join(work, "plots")
            print("[%s] running reference" % name, flush=True)
            rec = run_reference(spec, paths, out_dir, plot_dir)
            print("[%s] reference: part1 %.1fs part2 %.1fs" % (name, rec["seconds_part1"], rec["seconds_part2"]),
                  flush=True)
            if os.path.isdir(gold):
                shutil.rmtree(gold)
            os.makedirs(gold)
            for fn in OUTPUT_FILES:
                shutil.copyfile(os.path.join(out_dir, fn), os.path.join(gold, fn))
            hyper = np.array(rec["hyper"], dtype=np.float64).reshape(-1, 5)
            D = rec["D_list"][0]
            S = rec["S_list"][0]
            arrays = dict(
                kept_ids=rec["kept_ids"], row_sum_seq=rec["row_sum_seq"], Z=rec["Z"],
                initial_cuts=rec["initial_cuts"], filtered_cuts=rec["filtered_cuts"],
                hyper_xMnN=hyper[:, :4].astype(np.int64), hyper_p=hyper[:, 4],
                hyper_mark_first_pass=np.int64(rec["hyper_mark_first_pass"]),
                hyper_mark_filter=np.int64(rec["hyper_mark_filter"]),
                costs=np.array(rec["costs"], dtype=np.float64),
                cost_marks=np.array(rec["cost_marks"], dtype=np.int64),
                D_row0=D[0].copy(), D_diag=np.diag(D).copy(), S_row0=S[0].copy(),
                argsorted_head=rec["argsorted"][:, :8].astype(np.int32),
            )
            if spec["store_matrices"]:
                arrays.update(contacts=c, D=D, S=S, argsorted=rec["argsorted"].astype(np.uint16))
            np.savez_compressed(os.path.join(gold, "golden.npz"), **arrays)
            meta = dict(
                case=name, spec={k: (list(v) if isinstance(v, tuple) else v) for k, v in spec.items()},
                sha256=dict(contacts=sha(c), D=sha(D), S=sha(S), argsorted=sha(rec["argsorted"].astype(np.int64)),
                            Z=sha(rec["Z"])),
                chrom_orders=rec["chrom_orders"],
                reference_seconds=dict(part1=rec["seconds_part1"], part2=rec["seconds_part2"]),
                versions=dict(numpy=np.__version__, scipy=__import__("scipy").__version__,
                              python=sys.version.split()[0]),
                note="produced by oracle/gen_golden.py running /root/reference through oracle/ref_shim.py "
                     "(numba.jit = identity, plotting no-op); modularity = 0, hyperGeom = True, hmm = False",
            )
            with open(os.path.join(gold, "case.json"), "w") as fh:
                json.dump(meta, fh, indent=1, sort_keys=True)
            print("[%s] wrote %s" % (name, gold), flush=True)
        finally:
            shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1:])
