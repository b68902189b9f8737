"""The BASELINE.json sizes themselves (configs[1]-[4]) on one MI355X.

* 16,000 bins: Part 1 of the very step bench.py times against the CPU oracle - four files byte for byte
  (configs[1]/[2]; the oracle's SciPy linkage + NumPy argsort + hypergeometric scans take a minute or two); Part 2 of
  configs[2] through fixed-point properties and file equality with the earlier implementations.
* 32,000 bins: the UPGMA tree, leaf order, sampled rank rows and first-scan counts against the CPU oracle
  (north_star's Target size).
* 32,000 bins (configs[3]'s map, north_star's single-GPU target) and 64,000 bins with fp32 contacts
  (configs[4]'s map): the WHOLE resident -part1 -part2, stage by stage, through properties no oracle is needed for -
  leaf order a permutation, heights sorted, rank rows the inverse of the rank matrix and descending in similarity,
  the scans' counts re-derived from fetched rank rows, every ordered chromosome a fixed point of the
  sliding-window search (hicmi_p2_scan_pass) - and the default kernels against the earlier implementations of the
  same stages (file equality, as tests/test_gpu_differential.py does at 8,000 bins).
"""
import contextlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("dendrogramOrder.txt", "binGroups.txt", "assessment.txt", "chromosomeGroups.txt", "chromosomeOrders.txt",
         "plotOrder.txt")


def _bins(lay, cls):
    return [cls(int(lay.bin_ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], int(lay.start[k]), int(lay.stop[k]), 1.0, 0.0)
            for k in range(lay.n_bins)]


def _sizes_file(lay, path):
    with open(path, "w") as fh:
        fh.write("".join("%s\t%d\n" % (nm, sz) for nm, sz in zip(lay.scaffold_names, lay.scaffold_sizes_bp)))
    return str(path)


def test_part1_matches_oracle_at_16000_bins(tmp_path):
    """BASELINE configs[1]: the 16,000-bin map of bench.py's default step (same generator, seed and settings),
    Part 1 on the GPU against the CPU oracle: dendrogram order, bin groups, assessment and chromosome groups byte
    for byte, and the filtered cut indices."""
    import torch
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, scaffoldToChromosomes as p1, synth
    from hic_genome_assembler_amd.hostio import Bin
    n = 16000
    lay = synth.make_layout(n, seed=1)
    dev = torch.device("cuda", 0)
    ct = synth.dense_contacts_torch(lay, dev, seed=1, sinkhorn_iters=12)        # bench.py's map
    torch.cuda.synchronize()
    sizes = _sizes_file(lay, tmp_path / "sizes.txt")
    names = FILES[:4]
    for who in ("oracle", "gpu"):
        (tmp_path / who).mkdir()
    fo = [str(tmp_path / "oracle" / k) for k in names]
    fg = [str(tmp_path / "gpu" / k) for k in names]
    with _lib.Context(0) as ctx, contextlib.redirect_stdout(io.StringIO()):
        ctx.set_contacts_device(ct.data_ptr(), n, keepalive=ct)
        cuts_g = p1.runResident(p1.DeviceMatrix(ctx), _bins(lay, Bin), sizes, *fg, 5, 0.0, .05)
    c = ct.cpu().numpy()
    del ct
    torch.cuda.empty_cache()
    with contextlib.redirect_stdout(io.StringIO()):
        cuts_o = orc.run_part1(None, None, None, sizes, *fo, min_size=5, modularity=0.0, psig=.05,
                               preloaded=(c, _bins(lay, orc.Bin)))
    assert list(cuts_g) == list(cuts_o) and len(cuts_g) >= 10
    for a, b in zip(fg, fo):
        assert open(a).read() == open(b).read(), a


def _check_part2_fixed_points(ctx, p2, gm, kept_bins, ordered, min_checked):
    """Part 2 (a-13..a-15) without an oracle: every ordered chromosome with more than nScaffolds scaffolds is a fixed
    point of scanOrdering (OG:519-544) under its literal score, and the closed-form score agrees with the literal one."""
    checked = 0
    with contextlib.redirect_stdout(io.StringIO()):
        for group in ordered:
            if len(group) <= 6:
                continue
            view, _od = p2.giveNewAdjMat(gm, group, kept_bins)
            total = view.total()
            ids, rev = view.layout.describe(group)
            row = view.layout.node_row(ids, rev)
            exact = float(ctx.p2_score_exact(row[None, :], total)[0])
            fast = float(ctx.p2_score(row[None, :], total)[0])
            assert abs(fast - exact) <= 1e-10 * abs(exact)                      # north_star asks for 1e-5
            view.layout.tables(5)
            floor = exact * (1.0 + 1e-15)         # `total` was rounded in another order during the search
            _i, _r, best2, _cf, improved = ctx.p2_scan_pass(ids, rev, 5, total, floor, None)
            assert not improved and best2 == floor
            gm.chrom = None
            checked += 1
    assert checked >= min_checked
    return checked


def test_part2_at_16000_bins(tmp_path):
    """BASELINE configs[2]'s Part 2 half (OG:608-612 on the 16,000-bin map of bench.py's default step, which the oracle
    cannot order in test time: its brute force alone is 23,040 literal costs of ~10^6 terms per chromosome).  Asserted
    instead: every ordered chromosome is a fixed point of the sliding-window search under its literal score, the closed
    form agrees with the literal score to 1e-10, the files are consistent with each other, and the six files equal those
    of the earlier implementations of every stage (host-decided insertion, per-candidate window kernels, one queue per
    chromosome, cache-less single-workgroup nn-chain, radix sort) run in a second process."""
    import torch
    from hic_genome_assembler_amd import _lib, orderGenome as p2, scaffoldToChromosomes as p1, synth
    from hic_genome_assembler_amd.hostio import Bin
    n, seed = 16000, 1
    lay = synth.make_layout(n, seed=seed)
    ct = synth.dense_contacts_torch(lay, torch.device("cuda", 0), seed=seed, sinkhorn_iters=8)      # = _resident_map(n, 1, False)
    torch.cuda.synchronize()
    out = tmp_path / "default"
    out.mkdir()
    sizes = _sizes_file(lay, out / "sizes.txt")
    f = lambda k: str(out / k)  # noqa: E731
    with _lib.Context(0) as ctx:
        ctx.set_contacts_device(ct.data_ptr(), n, keepalive=ct)
        dm = p1.DeviceMatrix(ctx)
        with contextlib.redirect_stdout(io.StringIO()):
            p1.runResident(dm, _bins(lay, Bin), sizes, *[f(k) for k in FILES[:4]], 5, 0.0, .05)
            gm = p2.GenomeMatrix(ctx)
            ordered = p2.runResident(gm, dm.kept_bins, f(FILES[3]), f(FILES[4]), f(FILES[5]), 6, 5, lay.resolution)
        grouped = [l.split("\t")[0] for l in open(f(FILES[3])).read().splitlines() if not l.startswith("#")]
        plotted = [l.split("\t")[1] for l in open(f(FILES[5])).read().split("\n")[1:]]
        assert sorted(grouped) == sorted(plotted) and len(set(plotted)) == len(plotted)
        assert len(ordered) >= len(set(lay.chrom_of_bin.tolist()))
        _check_part2_fixed_points(ctx, p2, gm, dm.kept_bins, ordered, 8)
    texts = {k: open(f(k)).read() for k in FILES}
    assert texts["chromosomeOrders.txt"].count("\n") > n // 20
    del ct
    torch.cuda.empty_cache()
    script = tmp_path / "variant.py"
    script.write_text(_VARIANT_RUN)
    alt = tmp_path / "earlier"
    alt.mkdir()
    res = subprocess.run([sys.executable, str(script), ROOT, str(n), str(seed), "0", str(alt)],
                         env=dict(os.environ, **_EARLIER), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    for k in FILES:
        assert open(alt / k).read() == texts[k], k


def _oracle_distance_blocked(orc, c, block=2048):
    """orc.to_distance (S2C:147) in row blocks: the same three roundings per element without three N x N temporaries."""
    n = c.shape[0]
    out = np.empty((n, n), np.float64)
    for r0 in range(0, n, block):
        blk = c[r0:r0 + block]
        sig = orc.np_row_sums(blk)
        out[r0:r0 + block] = (1.0 - (blk / sig[:, None])) + 1.0
    return out


def test_upgma_matches_oracle_at_32000_bins():
    """north_star's Target size ("32,000 x 32,000 ... cluster/order outputs identical to the CPU reference"): the map of
    bench.py's north_star_32k object (seed 1), default kernels, against the CPU oracle - the raw merges in merge order,
    the labelled linkage, the leaf order (S2C:194-204) bit for bit; then, for sampled rows of the reordered matrix, the rank
    row against numpy's stable argsort (reversed) of the oracle's similarity row (S2C:149, 157-163, 1132) and the first
    first-pass scan's count x_i (S2C:455-459) re-derived from it.  The oracle's nn-chain takes ~30-40 s of host time here."""
    import time
    import torch
    import hic_oracle as orc
    from hic_genome_assembler_amd import _lib, synth
    n = 32000
    lay = synth.make_layout(n, seed=1)
    ct = synth.dense_contacts_torch(lay, torch.device("cuda", 0), seed=1, sinkhorn_iters=12)        # bench.py's Job(32000)
    torch.cuda.synchronize()
    with _lib.Context(0) as ctx:
        ctx.set_contacts_device(ct.data_ptr(), n, keepalive=ct)
        _np_sum, seq = ctx.row_sums()
        leaves, z = ctx.upgma()
        zraw = ctx.raw_merges()
        c = ct.cpu().numpy()
        t0 = time.time()
        dist = _oracle_distance_blocked(orc, c)
        zraw_o = orc.nn_chain_raw(dist)
        z_o = orc.label_linkage(zraw_o, n)
        leaves_o = orc.leaf_order(z_o, n)
        print("oracle UPGMA at 32,000 bins: %.1f s" % (time.time() - t0))
        assert np.array_equal(zraw, zraw_o)
        assert np.array_equal(z, z_o)
        assert np.array_equal(leaves, leaves_o)
        # ---- rank rows and first-scan counts of sampled rows against the oracle's similarity rows
        ctx.rank_matrix(leaves)
        _sig, x = ctx.cut_scan(0, n, 0.05, want_x=True)
        rng = np.random.default_rng(32)
        order = leaves_o.astype(np.int64)
        for a in [0, 1, 2, n // 2, n - 2, n - 1] + rng.integers(0, n, 42).tolist():
            r = int(order[a])
            assert seq[r] == orc.seq_row_sums(c[r:r + 1])[0]
            sim = seq[r] * (1.0 - (dist[r][order] - 1.0))                          # S2C:149 on the reordered row (S2C:161)
            R_o = np.argsort(sim, kind="stable")[::-1]
            assert np.array_equal(ctx.similarity_row(a), sim)
            assert np.array_equal(ctx.rank_rows(a, 1)[0].astype(np.int64), R_o), a
            if a > 0:
                pr = R_o[:a]
                assert x[a] == np.count_nonzero((pr >= 0) & (pr <= a)), a


# ------------------------------------------------------------------------------------------------ 32k / 64k
def _resident_map(n, seed, f32):
    """The synthetic map on the device (fp64), or - configs[4] - rounded to fp32 and handed over as a host array."""
    import torch
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(n, seed=seed)
    dev = torch.device("cuda", 0)
    c = synth.dense_contacts_torch(lay, dev, seed=seed, sinkhorn_iters=8)
    torch.cuda.synchronize()
    if not f32:
        return lay, c
    host = c.to(torch.float32).cpu().numpy()
    del c
    torch.cuda.empty_cache()
    return lay, host


def _check_stages_and_run(hic, lay, contacts, out_dir):
    """Stage by stage through the C ABI with property checks, then the drop-in's resident pipeline to files."""
    from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
    from hic_genome_assembler_amd.hostio import Bin
    n = lay.n_bins
    rng = np.random.default_rng(n)
    sizes = _sizes_file(lay, os.path.join(out_dir, "sizes.txt"))
    f = lambda k: os.path.join(out_dir, k)  # noqa: E731
    with hic.Context(0) as ctx:
        if isinstance(contacts, np.ndarray):
            ctx.set_contacts(contacts)                                  # fp32 host array: widened on the device
            row_ref = contacts.astype(np.float64).sum(axis=1)
        else:
            ctx.set_contacts_device(contacts.data_ptr(), n, keepalive=contacts)
            row_ref = contacts.sum(dim=1).cpu().numpy()
        np_sum, seq = ctx.row_sums()
        assert np.allclose(np_sum, row_ref, rtol=1e-12) and np.allclose(seq, row_ref, rtol=1e-12)
        # ---- UPGMA (a-3)
        leaves, z = ctx.upgma()
        assert sorted(leaves.tolist()) == list(range(n))
        assert np.all(np.diff(z[:, 2]) >= 0) and z[-1, 3] == n and np.all(z[:, 0] < z[:, 1]) and np.all(z[:, 3] >= 2)
        zraw = ctx.raw_merges()
        assert np.array_equal(np.sort(zraw[:, 2]), z[:, 2])
        chrom = lay.chrom_of_bin[leaves]
        assert np.count_nonzero(np.diff(chrom) != 0) == len(set(chrom.tolist())) - 1      # planted groups contiguous
        # ---- rank matrix (a-5): rows are permutations, descending in similarity, inverse consistent
        ctx.rank_matrix(leaves)
        ar = np.arange(n)
        for r in [0, 1, n // 3, n // 2, n - 2, n - 1] + rng.integers(0, n, 6).tolist():
            R = ctx.rank_rows(r, 1)[0].astype(np.int64)
            inv = ctx.rank_rows(r, 1, inverse=True)[0].astype(np.int64)
            s = ctx.similarity_row(r)
            assert np.array_equal(np.sort(R), ar)
            assert np.all(np.diff(s[R]) <= 0)
            assert np.array_equal(inv[R], ar)
            ties = np.flatnonzero(np.diff(s[R]) == 0)                   # tie rule: equal values by descending column
            assert np.all(R[ties] > R[ties + 1])
        # ---- first-pass counts (a-7) and filter counts (a-9) re-derived from fetched rank rows
        for start in (0, int(n * 0.37)):
            sig, x = ctx.cut_scan(start, n - start, 0.05, want_x=True)
            assert sig[0] == 0 and len(x) == n - start
            for i in [start + 1, start + 2, start + 1000, (start + n) // 2, n - 1]:
                pr = ctx.rank_rows(i, 1)[0].astype(np.int64)[:i - start]
                assert x[i - start] == np.count_nonzero((pr >= start) & (pr <= i)), (start, i)
        start, cut = int(n * 0.2), int(n * 0.26)
        n_rows = min(n - start, n // 5 + 1)
        sig, x = ctx.filter_scan(start, cut, n_rows, n - start, 0.05, want_x=True)
        for k in [0, 1, n_rows // 2, n_rows - 1]:
            pr = ctx.rank_rows(start + k, 1)[0].astype(np.int64)[:cut - start]
            assert x[k] == np.count_nonzero((pr >= start) & (pr <= cut)), k
            assert sig[k] == (1 if hic.hypergeom_sf(int(x[k]), n - start, cut - start, cut - start) < 0.05 else 0)
        # ---- the drop-in's resident pipeline: six files
        bins = _bins(lay, Bin)
        dm = p1.DeviceMatrix(ctx)
        with contextlib.redirect_stdout(io.StringIO()):
            cuts = p1.runResident(dm, list(bins), sizes, f(FILES[0]), f(FILES[1]), f(FILES[2]), f(FILES[3]), 5, 0.0, .05)
            gm = p2.GenomeMatrix(ctx)
            ordered = p2.runResident(gm, dm.kept_bins, f(FILES[3]), f(FILES[4]), f(FILES[5]), 6, 5, lay.resolution)
        assert cuts == sorted(set(cuts)) and 0 < cuts[0] and cuts[-1] < n
        assert [int(l.split("\t")[1]) for l in open(f(FILES[0])).read().split("\n")] == leaves.tolist()
        # every bin of a group file appears exactly once in the plot order; every scaffold once in the orders
        grouped = [l.split("\t")[0] for l in open(f(FILES[3])).read().splitlines() if not l.startswith("#")]
        plotted = [l.split("\t")[1] for l in open(f(FILES[5])).read().split("\n")[1:]]
        assert sorted(grouped) == sorted(plotted) and len(set(plotted)) == len(plotted)
        assert len(ordered) >= len(set(lay.chrom_of_bin.tolist()))
        # ---- Part 2 (a-13..a-15): every chromosome is a fixed point of scanOrdering under its literal score
        _check_part2_fixed_points(ctx, p2, gm, dm.kept_bins, ordered, 8)
    return {k: open(f(k)).read() for k in FILES}


_EARLIER = {"HICMI_P2_HOST_INSERT": "1", "HICMI_P2_WINDOW_DIRECT": "1", "HICMI_PART2_LOCKSTEP": "0", "HICMI_NNCHAIN_WGS": "1",
            "HICMI_NNCHAIN_PLAIN": "1", "HICMI_SORT_RADIX": "1"}

_VARIANT_RUN = r"""
import contextlib, io, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import test_gpu_scale as T
from hic_genome_assembler_amd import _lib, orderGenome as p2, scaffoldToChromosomes as p1
from hic_genome_assembler_amd.hostio import Bin
n, seed, f32, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5]
lay, contacts = T._resident_map(n, seed, f32)
sizes = T._sizes_file(lay, os.path.join(out, "sizes.txt"))
f = lambda k: os.path.join(out, k)
ctx = _lib.Context(0)
if isinstance(contacts, np.ndarray):
    ctx.set_contacts(contacts)
else:
    ctx.set_contacts_device(contacts.data_ptr(), n, keepalive=contacts)
dm = p1.DeviceMatrix(ctx)
with contextlib.redirect_stdout(io.StringIO()):
    p1.runResident(dm, T._bins(lay, Bin), sizes, *[f(k) for k in T.FILES[:4]], 5, 0.0, .05)
    p2.runResident(p2.GenomeMatrix(ctx), dm.kept_bins, f(T.FILES[3]), f(T.FILES[4]), f(T.FILES[5]), 6, 5, lay.resolution)
ctx.close()
"""


@pytest.mark.parametrize("n,f32", [(32000, False), (64000, True)], ids=["32k-fp64", "64k-fp32"])
def test_full_pipeline_at_baseline_size(n, f32, tmp_path):
    """configs[3] (32,000 bins) and configs[4] (64,000 bins, fp32 contacts) on one GPU: the full resident -part1 -part2
    with the property checks of _check_stages_and_run, then the same map through the earlier implementations
    (single-workgroup nn-chain without the neighbour cache, LSD radix row sort, host-decided insertion, per-candidate window
    kernels, one queue per chromosome) in a second process: the six files must be identical."""
    from hic_genome_assembler_amd import _lib as hic
    seed = 3 if n == 32000 else 5
    lay, contacts = _resident_map(n, seed, f32)
    out = tmp_path / "default"
    out.mkdir()
    texts = _check_stages_and_run(hic, lay, contacts, str(out))
    del contacts
    import torch
    torch.cuda.empty_cache()
    assert texts["chromosomeOrders.txt"].count("\n") > n // 20
    script = tmp_path / "variant.py"
    script.write_text(_VARIANT_RUN)
    alt = tmp_path / "earlier"
    alt.mkdir()
    res = subprocess.run([sys.executable, str(script), ROOT, str(n), str(seed), "1" if f32 else "0", str(alt)],
                         env=dict(os.environ, **_EARLIER), capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    for k in FILES:
        assert open(alt / k).read() == texts[k], k
