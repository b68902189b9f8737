"""Pin the CPU oracle (oracle/) against (a) the installed NumPy/SciPy routines the reference
calls and (b) the fixtures produced by running the reference itself (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import scipy.cluster.hierarchy as sch
import scipy.spatial.distance as ssd

import golden_cases as gc
import hic_oracle as orc

FAST_CASES = ["n160", "n300_edges", "n600"]
ALL_CASES = FAST_CASES + ["n400_default", "n2000", "n500_sparse", "n160_numba"]


# ---------------------------------------------------------------- known answers held by the reference
def test_known_answers_from_reference_source():
    # orderGenome.py:193 import-time warm-up value
    m = np.array([[float(i) for i in range(10)] for _ in range(10)])
    perm = np.arange(10, dtype=np.int32)
    c = orc.lib().hio_cost_literal(orc._dp(m), 10, orc._ip(perm), 10, 10.)
    assert c == pytest.approx(35.9, rel=1e-12)
    # scaffoldToChromosomes.py:374 and :429 docstring examples
    assert int(orc.sliding_window_scores([1, 1, 1, 1, 10, 0, 1, 0, 0, 0], 3).max()) == 11
    assert int(orc.sliding_window_scores([1, 1, 1, 1, 1, 0, 1, 0, 0, 0], 3).max()) == 2
    # orderGenome.py:374-379 candidate counts N! * 2^N / 2
    import math
    for k in range(1, 7):
        orders = orc.remove_reverse_duplicates(orc.swap_permutations(list(range(k))))
        orients = orc.plus_minus_perms(k)
        assert len(orders) * len(orients) == max(2, math.factorial(k) * 2 ** k // 2)
    assert ["".join(p) for p in orc.plus_minus_perms(3)] == ["+++", "---", "+--", "-+-", "--+", "++-", "+-+", "-++"]
    # SURVEY 8c probes of scipy.stats.hypergeom.sf through the reference's hyper_geom
    assert float(orc.hyper_geom(8, 600, 50, 50)) == pytest.approx(0.045750386176022624, rel=1e-13)
    assert float(orc.hyper_geom(7, 600, 50, 50)) == pytest.approx(0.11023306317208498, rel=1e-13)


# ---------------------------------------------------------------- third-party restatements
@pytest.mark.parametrize("n", [1, 5, 8, 100, 128, 129, 600, 2000, 4097, 8191, 8192, 8193, 16385, 20000, 32000])
def test_numpy_row_sum_restatement(n):
    rng = np.random.default_rng(n)
    a = rng.random((3, n)) * 1000.0
    got = orc.np_row_sums(a)
    for i in range(3):
        assert got[i] == np.asmatrix(a)[i].sum()           # S2C:147 row.sum()
        assert got[i] == np.sum(np.asmatrix(a), axis=-1)[i, 0]   # S2C:112
    seq = orc.seq_row_sums(a)
    for i in range(3):
        assert seq[i] == sum(np.asarray(a[i]))              # S2C:134


@pytest.mark.parametrize("n", [2, 3, 5, 40, 200, 500])
def test_nn_chain_restatement_matches_scipy(n):
    rng = np.random.default_rng(100 + n)
    d = rng.random((n, n)) + 1.0                            # deliberately asymmetric: only i<j is read
    y = ssd.squareform(d, checks=False)
    z_ref = sch.average(y)
    leaves, z = orc.average_cluster_leaves(d)
    assert np.array_equal(z, z_ref)
    dn = sch.dendrogram(z_ref, no_plot=True, get_leaves=True, count_sort="ascending")
    assert list(leaves) == dn["leaves"]


def test_nn_chain_with_ties_matches_scipy():
    rng = np.random.default_rng(9)
    d = rng.integers(1, 4, size=(60, 60)).astype(np.float64)
    d = np.triu(d, 1) + np.triu(d, 1).T
    z_ref = sch.average(ssd.squareform(d, checks=False))
    leaves, z = orc.average_cluster_leaves(d)
    assert np.array_equal(z, z_ref)
    dn = sch.dendrogram(z_ref, no_plot=True, get_leaves=True, count_sort="ascending")
    assert list(leaves) == dn["leaves"]


def test_cost_literal_matches_numpy_trace_loop():
    rng = np.random.default_rng(4)
    n = 150
    m = rng.random((n, n)); m = m + m.T
    perm = rng.permutation(n).astype(np.int32)
    sub = np.asmatrix(m[np.ix_(perm, perm)])
    total = sum([np.trace(sub, offset=i) for i in range(1, n)])          # OG:343
    cum, cost = 0., 0.
    for i in range(1, n):                                                # OG:323-330
        cum += np.trace(sub, offset=i)
        cost += (cum / total / float(i))
    L = orc.lib()
    t = L.hio_total_upper(orc._dp(m), n, orc._ip(perm), n)
    assert t == total
    assert L.hio_cost_literal(orc._dp(m), n, orc._ip(perm), n, t) == cost


# ---------------------------------------------------------------- golden fixtures (reference outputs)
def _run_oracle(name, tmp_path):
    spec = gc.load_case(name)[0]
    paths = gc.write_case_files(name, str(tmp_path))
    out = str(tmp_path)
    f = lambda k: os.path.join(out, k)  # noqa: E731
    t1, t2 = {}, {}
    orc.run_part1(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                  paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("binGroups.txt"), f("assessment.txt"),
                  f("chromosomeGroups.txt"), min_size=spec["min_size"], modularity=0.0, psig=spec["psig"], trace=t1)
    orc.run_part2(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                  f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), f("plotOrder.txt"),
                  n_scaffolds=spec["n_scaffolds"], scan_scaffolds=spec["scan_scaffolds"], trace=t2)
    return t1, t2, out


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_reproduces_reference_outputs(name, tmp_path):
    """n500_sparse: 2-decimal values and 50 % exact zeros, i.e. ties in SciPy's nn_chain and in every row of NumPy's
    (unstable) argsort - on THIS fixture the build's tie rules give the reference's files, linkage and cuts (tie order
    is implementation-defined in NumPy: parity is pinned for this container's NumPy 2.2.6 only).
    n160_numba: the cost loop summed as Numba compiles numpy.trace (sequential), selected with set_trace_order."""
    spec, meta, gold, lay, c = gc.load_case(name)
    orc.set_trace_order("numba" if spec.get("numba_trace") else "numpy")
    try:
        t1, t2, out = _run_oracle(name, tmp_path)
    finally:
        orc.set_trace_order("numpy")
    # integer / byte outputs: exact
    for fn in gc.OUTPUT_FILES:
        with open(os.path.join(out, fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn
    assert list(t1["initial_cuts"]) == list(gold["initial_cuts"])
    assert list(t1["cuts"]) == list(gold["filtered_cuts"])
    assert np.array_equal(t1["Z"], gold["Z"])
    assert [[list(p) for p in g] for g in t2["chrom_orders"]] == meta["chrom_orders"]
    # fp64 intermediates: bit-exact
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    assert sha(t1["dist_reordered"]) != ""   # (reordered distance is not a recorded stage)
    assert sha(t1["sim"]) == meta["sha256"]["S"]
    if not spec.get("sparse"):                                  # with ties NumPy's unstable order differs inside the runs
        assert sha(t1["R"].astype(np.int64)) == meta["sha256"]["argsorted"]
    else:
        assert np.array_equal(t1["R"][:, :8], gold["argsorted_head"])
        assert np.count_nonzero(c == 0) > 0.4 * c.size and len(np.unique(c)) < c.size // 20
    # every hypergeometric evaluation of the first pass, in call order
    k0, k1 = int(gold["hyper_mark_first_pass"]), int(gold["hyper_mark_filter"])
    ref_x = gold["hyper_xMnN"][k0:k1]
    mine = np.concatenate([np.stack([e["x"], np.full(len(e["x"]), e["M"]), np.arange(1, len(e["x"]) + 1),
                                     np.arange(1, len(e["x"]) + 1)], axis=1) for e in t1["first_pass"]]) \
        if t1["first_pass"] else np.zeros((0, 4), np.int64)
    if spec.get("sparse"):
        # NumPy's unstable argsort orders a run of equal similarities differently from the build's rule (stable
        # ascending, reversed), so a count whose prefix ends INSIDE a run of ties can differ: the arguments (M, n, N)
        # agree everywhere, 43 % of the x do not - "parity unpinned" for tie order.  On this fixture every
        # decision downstream (cut lists, files: asserted above) still coincides, although single counts differ by up
        # to a hundred: that is luck of this map, not a property - on real (sparse) maps cut indices may differ from
        # a reference run, just as two NumPy builds may differ from each other.
        assert np.array_equal(mine[:, 1:], ref_x[:len(mine), 1:])
        differ = mine[:, 0] != ref_x[:len(mine), 0]
        assert 0.2 < np.mean(differ) < 0.7
        return
    assert np.array_equal(mine, ref_x[:len(mine)])
    # what is left are the futile window-shrinking retries of the LAST call (S2C:499-508): the
    # same rows scanned again min_size-1 times; they cannot produce a cut (see hic_oracle docstring)
    rest = ref_x[len(mine):]
    if len(rest):
        last = t1["first_pass"][-1]["x"]
        assert len(rest) == len(last) * (spec["min_size"] - 1)
        assert np.array_equal(rest[:, 0], np.tile(last, spec["min_size"] - 1))
    # Part 2 objective values in the reference's evaluation order: fp64, same summation -> exact
    assert len(t2["costs"]) == len(gold["costs"])
    assert np.array_equal(t2["costs"], gold["costs"])


def test_numba_trace_order_differs_in_the_last_bits_only():
    """The two fixtures of the same 160-bin map: NumPy's pairwise trace vs Numba's sequential loop in the cost
    function - 1,712 objective values, some 40 % of them different in the last bits, none by more than 1e-15
    relative, and (on this map) the same six files."""
    g1, g2 = gc.load_case("n160")[2], gc.load_case("n160_numba")[2]
    a, b = g1["costs"], g2["costs"]
    assert len(a) == len(b) == 1712
    assert 0.2 < np.mean(a != b) < 0.8
    assert np.max(np.abs(a - b) / np.abs(a)) < 2e-15
    for fn in gc.OUTPUT_FILES:
        assert gc.golden_text("n160", fn) == gc.golden_text("n160_numba", fn)
