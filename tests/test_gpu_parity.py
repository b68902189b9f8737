"""GPU parity tests (run with -m gpu on an MI355X): every C-ABI stage of libhicmi against the CPU
oracle on the same inputs, against the reference-generated fixtures in tests/golden/, and - at the
BASELINE.json sizes - through size-independent properties.  Integer/byte outputs and the fp64
Part 1 intermediates must be bit-exact; ordering scores are compared at 1e-10 relative (the
north-star tolerance is 1e-5; the kernel is fp64 with a different summation order than
numpy.trace's pairwise sums)."""
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

CASES = ["n160", "n300_edges", "n400_default", "n600", "n2000", "n500_sparse", "n160_numba"]
# n500_sparse: 2-decimal contacts with 50 % exact zeros (ties in the clustering and in every sorted row);
# n160_numba: the reference's cost loop with numpy.trace summed as Numba compiles it (HICMI_P2_TRACE_ORDER=numba)


@pytest.fixture(scope="module")
def hic():
    from hic_genome_assembler_amd import _lib
    _lib.load()
    return _lib


@pytest.fixture(scope="module")
def orc():
    import hic_oracle
    return hic_oracle


# --------------------------------------------------------------------------------- row sums
@pytest.mark.parametrize("n", [1, 7, 129, 600, 2049, 8200, 16390])
def test_row_sums_bit_exact(hic, orc, n):
    rng = np.random.default_rng(n)
    rows = min(n, 300)
    m = np.zeros((n, n))
    m[:rows] = rng.random((rows, n)) * 1000.0
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        np_sum, seq = ctx.row_sums()
    assert np.array_equal(np_sum[:rows], orc.np_row_sums(m[:rows]))
    assert np.array_equal(seq[:rows], orc.seq_row_sums(m[:rows]))
    assert np.all(np_sum[rows:] == 0) and np.all(seq[rows:] == 0)


def test_compact_drops_rows_and_columns(hic, orc):
    rng = np.random.default_rng(3)
    m = rng.random((50, 50)); m = m + m.T
    keep = np.array([i for i in range(50) if i not in (3, 17, 49)], dtype=np.int32)
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.compact(keep)
        np_sum, seq = ctx.row_sums()
    sub = np.ascontiguousarray(m[np.ix_(keep, keep)])
    assert np.array_equal(np_sum, orc.np_row_sums(sub))
    assert np.array_equal(seq, orc.seq_row_sums(sub))


# --------------------------------------------------------------------------------- UPGMA
def _upgma_both(hic, orc, contacts):
    with hic.Context(0) as ctx:
        ctx.set_contacts(contacts)
        leaves, z = ctx.upgma()
        zraw = ctx.raw_merges()
    dist = orc.to_distance(contacts)
    zraw_o = orc.nn_chain_raw(dist)
    leaves_o, z_o = orc.average_cluster_leaves(dist)
    return leaves, z, zraw, leaves_o, z_o, zraw_o


@pytest.mark.parametrize("n,seed", [(2, 0), (3, 1), (17, 2), (64, 3), (65, 4), (333, 5), (1025, 6), (2500, 7)])
def test_upgma_random_bit_exact(hic, orc, n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T                                  # contacts symmetric; distance is still asymmetric (row sums differ)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)          # merge order, pairs, heights, sizes: bit for bit
    assert np.array_equal(z, z_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("dcap", [1, 2, 7, 64])
@pytest.mark.parametrize("n,seed", [(5, 0), (65, 4), (333, 5), (1025, 6)])
def test_upgma_deferred_columns_any_flush_period(hic, orc, monkeypatch, n, seed, dcap):
    """The nn-chain kernel defers the column half of each update and flushes every DCAP merges
    (k_nnchain.hip); results must not depend on DCAP."""
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", str(dcap))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)
    ties = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    ties = np.triu(ties, 1) + np.triu(ties, 1).T + np.eye(n)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, ties)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


def test_upgma_beyond_one_streaming_pass_bit_exact(hic, orc):
    """17,000 bins: rows longer than the 16,384 a 1024-lane workgroup streams in eight unrolled trips, 17 epochs,
    compactions from a matrix that is not a multiple of anything - still SciPy's linkage, bit for bit."""
    n = 17000
    rng = np.random.default_rng(42)
    c = rng.random((n, n), dtype=np.float32).astype(np.float64) + 0.01
    c = c + c.T
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(z, z_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("wgs", [2, 4, 8, 16])
@pytest.mark.parametrize("n,seed,dcap", [(130, 1, 7), (600, 2, 64), (1025, 6, 1024), (2500, 7, 1024), (4099, 8, 300)])
def test_upgma_column_sliced_chain_bit_exact(hic, orc, monkeypatch, n, seed, dcap, wgs):
    """k_nn_epoch_mwc - the chain as 2, 4, 8 or 16 workgroups of 1,024 lanes that each stream a slice of the columns, the
    neighbour cache replicated in every workgroup, the next scan fused into the update (round 2's default; since round 3
    what rows beyond 32,767 columns run on, and the A/B partner of the one-wave kernel).  HICMI_NNCHAIN_WGS forces it for
    every epoch that is wide enough.  Run three times: the exchange is timing-dependent, the linkage must not be."""
    monkeypatch.setenv("HICMI_NNCHAIN_WGS", str(wgs))
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", str(dcap))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    for _ in range(3):
        leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
        assert np.array_equal(zraw, zraw_o)
        assert np.array_equal(leaves, leaves_o)
    ties = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    ties = np.triu(ties, 1) + np.triu(ties, 1).T + np.eye(n)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, ties)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("slices", [1, 2, 3, 7, 16, 33, 64])
@pytest.mark.parametrize("n,seed,dcap", [(2, 0, 256), (3, 1, 1), (130, 1, 7), (600, 2, 64), (1025, 6, 256), (2500, 7, 256), (4099, 8, 100)])
def test_upgma_one_wave_per_slice_bit_exact(hic, orc, monkeypatch, n, seed, dcap, slices):
    """k_nn_epoch_w1 (round 3, the default up to 32,768 live columns): the chain as up to 64 single-wave workgroups that each
    stream a slice of the columns, reduce in the wave (no workgroup barrier), exchange (min, index, tie) once per merge and
    keep the neighbour cache with time stamps instead of an eager invalidation pass.  HICMI_NNCHAIN_W1_S forces the number
    of slices for every epoch (the plan widens a slice to at most 2,048 columns and drops empty ones), HICMI_NNCHAIN_DCAP the
    flush period.  Against the oracle's raw merges and leaf order on random distances (three runs: the exchange is
    timing-dependent, the linkage must not be) and on tie-heavy integer distances."""
    monkeypatch.setenv("HICMI_NNCHAIN_W1_S", str(slices))
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", str(dcap))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    for _ in range(3 if n <= 1025 else 2):
        leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
        assert np.array_equal(zraw, zraw_o)
        assert np.array_equal(leaves, leaves_o)
    ties = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    ties = np.triu(ties, 1) + np.triu(ties, 1).T + np.eye(n)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, ties)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("xcd", ["off", "5", "8"])
@pytest.mark.parametrize("n,seed,slices", [(130, 1, 2), (1025, 6, 33), (4099, 8, 64), (4099, 9, 0)])
def test_upgma_one_wave_per_slice_spread_out_or_on_another_xcd(hic, orc, monkeypatch, n, seed, slices, xcd):
    """By default the parties of k_nn_epoch_w1 all run on XCD 0 (its LOCAL form: plain stores that stay in that XCD's L2;
    what the other tests of this kernel exercise).  HICMI_NNCHAIN_XCD=off spreads them over the chip again (sc1 stores,
    the protocol of rounds 1-2), a digit names another XCD - 8 one that does not exist: no workgroup of the LOCAL launch
    claims a slice and the launch queued behind it sees that from the count; the linkage is the oracle's either way."""
    monkeypatch.setenv("HICMI_NNCHAIN_XCD", xcd)
    if slices:
        monkeypatch.setenv("HICMI_NNCHAIN_W1_S", str(slices))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    for _ in range(2):
        leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
        assert np.array_equal(zraw, zraw_o)
        assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("epoch", [1, 3])
def test_upgma_one_wave_roll_call_that_is_never_complete(hic, orc, monkeypatch, epoch):
    """The parties of a LOCAL epoch first wait until all of them run (a CU of their XCD may be busy with another process's
    kernels).  Test hook: epoch `epoch` waits for a party that never comes - every party gives up after ~2 ms without having
    touched anything, the spread-out launch queued behind it runs the epoch, and the rest of the chain stays spread out."""
    monkeypatch.setenv("HICMI_NNCHAIN_TEST_ROLLCALL", str(epoch))
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", "300")
    rng = np.random.default_rng(31)
    n = 2100
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        assert ctx.nnchain_stats()["retries"] == 0
        dist = orc.to_distance(c)
        assert np.array_equal(ctx.raw_merges(), orc.nn_chain_raw(dist))
        assert np.array_equal(leaves, orc.average_cluster_leaves(dist)[0])


@pytest.mark.parametrize("cols", [64, 256, 1024, 2048])
def test_upgma_one_wave_per_slice_planned_widths(hic, orc, monkeypatch, cols):
    """The plan itself (HICMI_NNCHAIN_W1_COLS columns per slice: 1, 2, 8 or 16 pairs per lane and streamed row) on a Hi-C-like
    map with planted chromosomes and on its quantised, sparse twin (exact ties in every row: the cache must defer to scans)."""
    from hic_genome_assembler_amd import synth
    monkeypatch.setenv("HICMI_NNCHAIN_W1_COLS", str(cols))
    lay = synth.make_layout(3000, seed=7)
    c = synth.dense_contacts(lay, seed=7, sinkhorn_iters=8)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o) and np.array_equal(z, z_o) and np.array_equal(leaves, leaves_o)
    q = np.round(c, 1)
    q[q <= np.quantile(q, 0.4)] = 0.0
    q = 0.5 * (q + q.T)
    np.fill_diagonal(q, np.maximum(np.diag(q), 1.0))
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, np.ascontiguousarray(q))
    assert np.array_equal(zraw, zraw_o) and np.array_equal(leaves, leaves_o)


def test_upgma_one_wave_per_slice_late_peer_and_divergence(hic, orc, monkeypatch, capfd):
    """The safety nets on the one-wave kernel: an exchange declared late (test hook) re-runs the map - spread over the chip first, on one
    workgroup if that is late too;
    a replica whose merge record differs (test hook: replica 1 flips a bit of the height of merge 123 in its hash) fails the
    call instead of returning a tree."""
    monkeypatch.setenv("HICMI_NNCHAIN_W1_S", "5")
    monkeypatch.setenv("HICMI_NNCHAIN_TEST_LATE", "40")
    rng = np.random.default_rng(12)
    n = 700
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    dist = orc.to_distance(c)
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        ctx.timing_reset()
        leaves, _z = ctx.upgma()
        assert ctx.nnchain_stats()["retries"] == 1
        assert np.array_equal(ctx.raw_merges(), orc.nn_chain_raw(dist))
        assert "answered late" in capfd.readouterr().err
        monkeypatch.delenv("HICMI_NNCHAIN_TEST_LATE")
        monkeypatch.setenv("HICMI_NNCHAIN_TEST_DIVERGE", "123")
        with pytest.raises(hic.HicmiError, match="replicas"):
            ctx.upgma()
        monkeypatch.delenv("HICMI_NNCHAIN_TEST_DIVERGE")
        leaves, _z = ctx.upgma()                       # the context is usable afterwards
        assert np.array_equal(leaves, orc.average_cluster_leaves(dist)[0])


@pytest.mark.parametrize("n,seed,dcap", [(130, 1, 7), (1025, 6, 256), (4099, 8, 300)])
def test_upgma_cluster_sizes_in_global_memory_bit_exact(hic, orc, monkeypatch, n, seed, dcap):
    """k_nn_epoch_mwc<8, ., true>: what rows beyond 32,768 columns run on - the neighbour cache alone in LDS, the cluster
    sizes in a private global array per workgroup.  HICMI_NNCHAIN_GSIZE=1 selects it at every width, here against the
    oracle (random and tie-heavy distances)."""
    monkeypatch.setenv("HICMI_NNCHAIN_WGS", "16" if n == 4099 else "8")
    monkeypatch.setenv("HICMI_NNCHAIN_GSIZE", "1")
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", str(dcap))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    for _ in range(2):
        leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
        assert np.array_equal(zraw, zraw_o)
        assert np.array_equal(leaves, leaves_o)
    ties = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    ties = np.triu(ties, 1) + np.triu(ties, 1).T + np.eye(n)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, ties)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


def test_upgma_wide_rows_match_the_oracle_at_40k(hic, orc):
    """40,000 bins: the first epochs have more than 32,767 live columns and run on the GSIZE variant of k_nn_epoch_mwc
    (cluster sizes in global memory, the cache alone in LDS - what every 64,000-bin map starts on), the later ones on the
    one-wave kernel.  Against the CPU oracle's nn-chain (~1.5 min of host time), at its own width - not only forced
    onto small inputs.  On uniform random distances the cache saves less than on Hi-C maps (2.3 row scans per merge
    against SciPy's 2.9; the 64k bench map runs at 1.28)."""
    import torch
    n = 40000
    g = torch.Generator(device="cuda:0")
    g.manual_seed(9)
    c = torch.rand((n, n), generator=g, device="cuda:0", dtype=torch.float64)
    c = c + c.T + 0.01
    torch.cuda.synchronize()
    with hic.Context(0) as ctx:
        ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
        ctx.timing_reset()
        leaves, _z = ctx.upgma()
        zraw, st = ctx.raw_merges().copy(), ctx.nnchain_stats()
    assert sorted(leaves.tolist()) == list(range(n))
    assert st["scans"] < 2.6 * st["merges"]
    host = c.cpu().numpy()
    del c
    torch.cuda.empty_cache()
    dist = np.empty((n, n), np.float64)
    for r0 in range(0, n, 2048):                            # orc.to_distance (S2C:147) in row blocks
        blk = host[r0:r0 + 2048]
        dist[r0:r0 + 2048] = (1.0 - (blk / orc.np_row_sums(blk)[:, None])) + 1.0
    del host
    zraw_o = orc.nn_chain_raw(dist)
    del dist
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(np.array(leaves), orc.leaf_order(orc.label_linkage(zraw_o, n), n))


def test_upgma_column_sliced_chain_default_width_at_scale(hic, orc):
    """21,000 bins: above the width switch, so the first epochs run on 8 workgroups and the later ones - once
    compaction has shrunk the matrix below it - on one, sharing the saved chain state."""
    n = 21000
    rng = np.random.default_rng(43)
    c = rng.random((n, n), dtype=np.float32).astype(np.float64) + 0.01
    c = c + c.T
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("n,seed,dcap", [(5, 0, 1024), (65, 4, 7), (1025, 6, 64), (2500, 7, 1024)])
def test_upgma_without_neighbour_cache_bit_exact(hic, orc, monkeypatch, n, seed, dcap):
    """HICMI_NNCHAIN_PLAIN=1: k_nn_epoch, the chain that scans a row at every step like SciPy itself (the A/B partner
    of the default k_nn_epoch_nc, and what tests/test_gpu_scale.py compares the default with at 32k / 64k)."""
    monkeypatch.setenv("HICMI_NNCHAIN_PLAIN", "1")
    monkeypatch.setenv("HICMI_NNCHAIN_DCAP", str(dcap))
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)
    ties = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    ties = np.triu(ties, 1) + np.triu(ties, 1).T + np.eye(n)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, ties)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


def test_upgma_neighbour_cache_counters(hic, orc):
    """The neighbour cache answers most chain steps: on a structured map fewer than two row scans per merge are left
    (SciPy's loop does ~2.9), the counters add up, and quantised sparse contacts - exact ties everywhere, the case
    where the cache must defer to a scan - still give SciPy's linkage."""
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(1800, seed=5)
    c = synth.dense_contacts(lay, seed=5, sinkhorn_iters=8)
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        ctx.timing_reset()
        leaves, z = ctx.upgma()
        st = ctx.nnchain_stats()
    leaves_o, z_o = orc.average_cluster_leaves(orc.to_distance(c))
    assert np.array_equal(z, z_o) and np.array_equal(leaves, leaves_o)
    assert st["merges"] == len(c) - 1 and st["retries"] == 0
    assert 0.9 * st["merges"] < st["scans"] < 2.0 * st["merges"]
    assert st["cache_hits"] > st["merges"]
    assert st["scan_columns"] <= st["scans"] * len(c)
    q = np.round(c, 1)
    q[q <= np.quantile(q, 0.4)] = 0.0
    q = 0.5 * (q + q.T)
    np.fill_diagonal(q, np.maximum(np.diag(q), 1.0))
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, np.ascontiguousarray(q))
    assert np.array_equal(zraw, zraw_o) and np.array_equal(leaves, leaves_o)


def test_upgma_late_peer_falls_back_to_one_workgroup(hic, orc, monkeypatch, capfd):
    """A peer workgroup of the column-sliced chain that does not answer in time (test hook: the 40th exchange is
    declared late) must not fail the map: the distances are rebuilt and the chain re-runs (without the one-XCD form first,
    then on one workgroup)."""
    monkeypatch.setenv("HICMI_NNCHAIN_WGS", "4")
    monkeypatch.setenv("HICMI_NNCHAIN_TEST_LATE", "40")
    rng = np.random.default_rng(12)
    n = 700
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        ctx.timing_reset()
        leaves, z = ctx.upgma()
        zraw = ctx.raw_merges()
        assert ctx.nnchain_stats()["retries"] == 1
    dist = orc.to_distance(c)
    assert np.array_equal(zraw, orc.nn_chain_raw(dist))
    assert np.array_equal(leaves, orc.average_cluster_leaves(dist)[0])
    assert "answered late" in capfd.readouterr().err


def test_upgma_replica_divergence_is_an_error(hic, monkeypatch):
    """Every replica of the column-sliced chain records every merge; k_nn_check_replicas compares the records after
    each epoch.  Test hook: replica 1 falsifies the height of merge 123 - the call must fail, not return a tree."""
    monkeypatch.setenv("HICMI_NNCHAIN_WGS", "4")
    monkeypatch.setenv("HICMI_NNCHAIN_TEST_DIVERGE", "123")
    rng = np.random.default_rng(13)
    n = 600
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        with pytest.raises(hic.HicmiError, match="replicas"):
            ctx.upgma()
        monkeypatch.delenv("HICMI_NNCHAIN_TEST_DIVERGE")
        leaves, _z = ctx.upgma()                       # the context is usable afterwards
        assert sorted(leaves.tolist()) == list(range(n))


def test_upgma_widths_agree_at_32k(hic, monkeypatch):
    """BASELINE's 32,000-bin size, where SciPy is too slow to ask: the default (8 column-sliced workgroups while more
    than 20,000 columns are live) must give the very merges of the lone workgroup, which the tests above pin to
    SciPy up to 21,000 bins."""
    n = 32000
    rng = np.random.default_rng(44)
    c = rng.random((n, n), dtype=np.float32)
    c += c.T
    c += np.float32(0.01)
    got = []
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)                       # float32 contacts are widened on the device (configs[4])
        del c
        for width in (None, "1"):
            if width is None:
                monkeypatch.delenv("HICMI_NNCHAIN_WGS", raising=False)
            else:
                monkeypatch.setenv("HICMI_NNCHAIN_WGS", width)
            leaves, _z = ctx.upgma()
            got.append((np.array(leaves), ctx.raw_merges().copy()))
    assert np.array_equal(got[0][1], got[1][1])
    assert np.array_equal(got[0][0], got[1][0])
    assert sorted(got[0][0].tolist()) == list(range(n))


def test_exact_division_by_cluster_size_selftest(hic):
    """k_nnchain's 3-instruction division by (nx+ny) against the '/' operator: 2^29 random operands."""
    with hic.Context(0) as ctx:
        assert ctx.selftest_division(samples=1 << 29, seed=2026) == 0
        assert ctx.selftest_division(samples=1 << 26, seed=7) == 0


def test_upgma_heavy_ties_bit_exact(hic, orc):
    rng = np.random.default_rng(11)
    c = rng.integers(1, 4, size=(200, 200)).astype(np.float64)
    c = np.triu(c, 1) + np.triu(c, 1).T + np.eye(200)
    leaves, z, zraw, leaves_o, z_o, zraw_o = _upgma_both(hic, orc, c)
    assert np.array_equal(zraw, zraw_o)
    assert np.array_equal(leaves, leaves_o)


@pytest.mark.parametrize("name", CASES)
def test_upgma_matches_reference_linkage(hic, name):
    spec, meta, gold, lay, c = gc.load_case(name)
    nan_bins = set(spec.get("nan_bias", ()))
    keep = np.array([i for i in np.flatnonzero(c.sum(axis=1) != 0) if i not in nan_bins])
    sub = np.ascontiguousarray(c[np.ix_(keep, keep)])
    with hic.Context(0) as ctx:
        ctx.set_contacts(sub)
        leaves, z = ctx.upgma()
    assert np.array_equal(z, gold["Z"])          # scipy.cluster.hierarchy.average as run by the reference
    ref_leaves = [int(l.split("\t")[1]) for l in gc.golden_text(name, "dendrogramOrder.txt").split("\n")]
    assert list(leaves) == ref_leaves


# --------------------------------------------------------------------------------- rank matrix
def _oracle_rank(orc, contacts, order):
    bins = [orc.Bin(i, "s", 0, 0, 0.0, 0.0) for i in range(len(contacts))]
    mat, bins = orc.remove_zero_rows(contacts.copy(), bins)
    dist = orc.to_distance(mat)
    dist = dist[:, order][order]
    bins = [bins[i] for i in order]
    sim = orc.to_similarity(dist, bins)
    return sim, orc.rank_order(sim)


@pytest.mark.parametrize("n,seed", [(2, 0), (100, 1), (1000, 2), (3000, 3), (8192, 4), (9001, 5)])
def test_rank_matrix_bit_exact(hic, orc, n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((n, n)) + 0.01
    c = c + c.T
    order = rng.permutation(n)
    rows = np.unique(np.concatenate([np.arange(min(n, 40)), rng.integers(0, n, 40), [n - 1]]))
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        ctx.rank_matrix(order)
        R = ctx.rank_rows()
        inv = ctx.rank_rows(inverse=True)
        srow = ctx.similarity_row(int(rows[0]))
    sim, R_o = _oracle_rank(orc, c, order)
    assert np.array_equal(srow, sim[rows[0]])
    assert np.array_equal(R.astype(np.int64), R_o)
    ar = np.arange(n)
    for r in rows:
        assert np.array_equal(inv[r][R[r]], ar)


def test_rank_matrix_tie_rule(hic, orc):
    """Heavy ties: descending value, equal values by descending column (= numpy stable argsort reversed)."""
    rng = np.random.default_rng(21)
    n = 700
    c = rng.integers(0, 5, size=(n, n)).astype(np.float64)
    c = np.triu(c, 1) + np.triu(c, 1).T + np.eye(n) * 3
    order = rng.permutation(n)
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        ctx.rank_matrix(order)
        R = ctx.rank_rows()
    _sim, R_o = _oracle_rank(orc, c, order)
    assert np.array_equal(R.astype(np.int64), R_o)


# --------------------------------------------------------------------------------- cut scans
@pytest.mark.parametrize("name", ["n160", "n600"])
def test_cut_and_filter_scans_match_oracle(hic, orc, name):
    from scipy.stats import hypergeom
    spec, meta, gold, lay, c = gc.load_case(name)
    n = len(c)
    rng = np.random.default_rng(5)
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        ctx.rank_matrix(leaves)
        R = ctx.rank_rows().astype(np.int64)
        for start in [0, 1, 7, n // 3, n - 2, n - 1]:
            for M in [n - start, max(1, (n - start) // 2), 3]:
                sig, x = ctx.cut_scan(start, M, 0.05, want_x=True)
                x_o = orc.first_pass_counts(R, start)[start:]
                assert np.array_equal(x, x_o)
                L = np.arange(1, n - start)
                with np.errstate(all="ignore"):
                    p = hypergeom.sf(x_o[1:] - 1, M, L, L)
                sig_o = np.concatenate(([0], np.where(p >= 0.05, 0, 1)))
                assert np.array_equal(sig, sig_o)
        for _ in range(12):
            start = int(rng.integers(0, n - 2))
            cut = int(rng.integers(start, n))
            n_rows = int(rng.integers(1, n - start + 1))
            M = n - start
            sig, x = ctx.filter_scan(start, cut, n_rows, M, 0.01, want_x=True)
            sub = R[start:start + n_rows, :cut - start]
            x_o = np.count_nonzero((sub >= start) & (sub <= cut), axis=1)
            assert np.array_equal(x, x_o)
            with np.errstate(all="ignore"):
                p = hypergeom.sf(x_o - 1, M, cut - start, cut - start)
            assert np.array_equal(sig, np.where(p < 0.01, 1, 0))


def test_hypergeom_decisions_match_scipy(hic):
    from scipy.stats import hypergeom
    rng = np.random.default_rng(0)
    for _ in range(4000):
        M = int(rng.integers(1, 64000)); n = int(rng.integers(0, M + 1)); N = int(rng.integers(0, M + 1))
        mean = n * N / M
        sd = max(1.0, (mean * (1 - n / M) * (M - N) / max(M - 1, 1)) ** 0.5)
        x = int(round(mean + rng.normal() * 3 * sd))
        mine = hic.hypergeom_sf(x, M, n, N)
        ref = float(hypergeom.sf(x - 1, M, n, N))
        assert (mine < 0.05) == (ref < 0.05)
        if ref > 1e-300:
            assert abs(mine - ref) <= 1e-10 * ref
    assert np.isnan(hic.hypergeom_sf(3, 10, 11, 5)) and np.isnan(hic.hypergeom_sf(3, 0, 0, 0))


# --------------------------------------------------------------------------------- Part 2 objective
def test_p2_scores_match_literal_cost(hic, orc):
    rng = np.random.default_rng(8)
    n = 260
    m = rng.random((n, n)); m = m + m.T
    sel = rng.permutation(n)[:200].astype(np.int32)
    perms = np.stack([rng.permutation(200) for _ in range(64)]).astype(np.int32)
    perms[1] = perms[0]                                  # identical index lists -> identical scores
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(sel)
        total = ctx.p2_total()
        scores = ctx.p2_score(perms, total)
        short = ctx.p2_score(perms[:, :50] % 50, total)
    L = orc.lib()
    ident = np.arange(200, dtype=np.int32)
    sub = np.ascontiguousarray(m[np.ix_(sel, sel)])
    total_o = L.hio_total_upper(orc._dp(sub), 200, orc._ip(ident), 200)
    assert total == pytest.approx(total_o, rel=1e-12)
    for k in range(64):
        ref = L.hio_cost_literal(orc._dp(sub), 200, orc._ip(perms[k]), 200, total)
        assert scores[k] == pytest.approx(ref, rel=1e-10)
    assert scores[0] == scores[1]
    assert short.shape == (64,)


@pytest.mark.parametrize("n_used", [2, 7, 8, 9, 127, 128, 129, 200, 1000])
def test_p2_literal_scores_bit_exact(hic, orc, n_used):
    """hicmi_p2_total / hicmi_p2_score_exact reproduce the reference's NumPy arithmetic bit for bit
    (numpy.trace pairwise sums per offset, sequential cum/total/i), which is what decides the
    ulp-level `cost > bestCost` comparisons of orderGenome.py:349,359,464,535."""
    rng = np.random.default_rng(n_used)
    n = n_used + 13
    m = rng.random((n, n)) * 7.0; m = m + m.T
    sel = rng.permutation(n)[:n_used].astype(np.int32)
    perms = np.stack([rng.permutation(n_used) for _ in range(9)]).astype(np.int32)
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(sel)
        total = ctx.p2_total()
        exact = ctx.p2_score_exact(perms, total)
        fast = ctx.p2_score(perms, total)
    L = orc.lib()
    sub = np.ascontiguousarray(m[np.ix_(sel, sel)])
    ident = np.arange(n_used, dtype=np.int32)
    assert total == L.hio_total_upper(orc._dp(sub), n_used, orc._ip(ident), n_used)
    for k in range(len(perms)):
        assert exact[k] == L.hio_cost_literal(orc._dp(sub), n_used, orc._ip(perms[k]), n_used, total)
    assert np.allclose(fast, exact, rtol=1e-11, atol=0)


@pytest.mark.parametrize("n_used", [2, 9, 130, 1000, 1100])
def test_p2_literal_scores_in_numba_order_bit_exact(hic, orc, monkeypatch, n_used):
    """HICMI_P2_TRACE_ORDER=numba: the candidates' diagonal sums run left to right (what Numba's np_trace does inside
    costFunction_numba, orderGenome.py:184-191) while the total keeps NumPy's pairwise order - bit for bit against the
    oracle's restatement of that loop, and different from the NumPy order in the last bits."""
    rng = np.random.default_rng(n_used)
    n = n_used + 5
    m = rng.random((n, n)) * 7.0; m = m + m.T
    sel = rng.permutation(n)[:n_used].astype(np.int32)
    perms = np.stack([rng.permutation(n_used) for _ in range(5)]).astype(np.int32)
    L = orc.lib()
    sub = np.ascontiguousarray(m[np.ix_(sel, sel)])
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(sel)
        total = ctx.p2_total()
        numpy_order = ctx.p2_score_exact(perms, total)
        monkeypatch.setenv("HICMI_P2_TRACE_ORDER", "numba")
        assert ctx.p2_total() == total                              # the total is not part of the jitted function
        numba_order = ctx.p2_score_exact(perms, total)
    orc.set_trace_order("numba")
    try:
        want = [L.hio_cost_literal(orc._dp(sub), n_used, orc._ip(perms[k]), n_used, total) for k in range(len(perms))]
    finally:
        orc.set_trace_order("numpy")
    assert numba_order.tolist() == want
    assert np.allclose(numba_order, numpy_order, rtol=1e-13, atol=0)
    if n_used == 130:                                               # (long rows: the last bits of a diagonal vanish in `cum`)
        assert np.any(numba_order != numpy_order)


def test_p2_device_enumeration_matches_explicit_candidates(hic, orc):
    """Insertion and window candidates enumerated by the kernels (k_part2_search.hip) against the
    same candidates spelled out as index lists and scored by hicmi_p2_score / the oracle."""
    from hic_genome_assembler_amd import orderGenome as p2
    rng = np.random.default_rng(17)
    lens = [9, 1, 4, 12, 1, 7, 3, 5, 2]
    n = sum(lens)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int32)
    m = rng.random((n + 5, n + 5)); m = m + m.T
    sel = rng.permutation(n + 5)[:n].astype(np.int32)

    def pos(sid, rev):
        a = np.arange(starts[sid], starts[sid] + lens[sid], dtype=np.int32)
        return a[::-1] if rev else a
    L = orc.lib()
    sub = np.ascontiguousarray(m[np.ix_(sel, sel)])
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(sel)
        ctx.p2_layout(starts, lens)
        # --- arrangement of 7 scaffolds, total in arrangement order, arrangement score
        ids = np.array([3, 0, 5, 8, 1, 6, 2], dtype=np.int32)
        rev = np.array([0, 1, 1, 0, 0, 1, 0], dtype=np.uint8)
        base = np.concatenate([pos(i, r) for i, r in zip(ids, rev)])
        ctx.p2_set_arrangement(ids, rev)
        total = ctx.p2_arrangement_total()
        assert total == L.hio_total_upper(orc._dp(sub), n, orc._ip(base), len(base))
        s_arr = ctx.p2_arrangement_score(total)
        assert s_arr == pytest.approx(L.hio_cost_literal(orc._dp(sub), n, orc._ip(base), len(base), total), rel=1e-12)
        # --- insertions of scaffold 4 and of scaffold 7
        pieces = [pos(i, r) for i, r in zip(ids, rev)]
        for new_id in (4, 7):
            got = ctx.p2_score_insertions(new_id, total)
            rows = [np.concatenate(pieces[:g] + [pos(new_id, r)] + pieces[g:]) for g in range(len(ids) + 1) for r in (0, 1)]
            want = ctx.p2_score(np.stack(rows), total)
            assert np.allclose(got, want, rtol=1e-12, atol=0)     # incremental form vs spelled-out candidates
        # --- windows of k = 3 and k = 5 scaffolds at every position, and the whole arrangement (k = S)
        for k in (3, 5, 7):
            orders, orients = p2._enumeration(k)
            ctx.p2_window_tables(np.asarray(orders, np.int8),
                                 np.asarray([[1 if sg == "-" else 0 for sg in r] for r in orients], np.uint8))
            for first in range(0, len(ids) - k + 1):
                delta = ctx.p2_score_window(first, k)
                head = np.concatenate(pieces[:first]) if first else np.zeros(0, np.int32)
                tail = np.concatenate(pieces[first + k:]) if first + k < len(ids) else np.zeros(0, np.int32)
                rows = []
                for o in orders:
                    for r in orients:
                        mid = [pos(int(ids[first + j]), sg == "-") for j, sg in zip(o, r)]
                        rows.append(np.concatenate([head] + mid + [tail]))
                want = ctx.p2_score(np.stack(rows), total)
                c0 = p2._orient_index(k, ["-" if v else "+" for v in rev[first:first + k]])
                assert np.array_equal(rows[c0], base)
                got = s_arr + (delta - delta[c0]) / total
                assert np.allclose(got, want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("lens, k", [
    ([150, 40, 90, 17, 200, 33, 60, 5, 310, 280, 12, 1, 450], 3),     # 2 / 4 / 8 slot tiles per wave, 1 ... 20 row tiles
    ([300, 250, 100, 64, 16, 700, 15], 3),                             # windows beyond 512 bins: the vector-ALU tables
    ([31, 7, 16, 48, 2, 90, 25, 130, 11, 64, 5, 77, 1200], 5),         # k = 5 with one scaffold far heavier than the rest
])
def test_p2_window_tables_with_wide_windows_and_long_scaffolds(hic, orc, lens, k):
    """The placement tables (k_part2_window.hip) where their work split matters: windows of 100 ... 700 bins, scaffolds of
    1 ... 450 bins (wt_split: several groups of row tiles x several column slices per scaffold, partial tables added by
    k_win_pairs), every kernel instantiation - against the same candidates spelled out and scored by hicmi_p2_score."""
    from hic_genome_assembler_amd import orderGenome as p2
    rng = np.random.default_rng(len(lens) * 100 + k)
    n = sum(lens)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int32)
    m = rng.random((n + 3, n + 3)); m = m + m.T
    sel = rng.permutation(n + 3)[:n].astype(np.int32)

    def pos(sid, rev):
        a = np.arange(starts[sid], starts[sid] + lens[sid], dtype=np.int32)
        return a[::-1] if rev else a
    with hic.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(sel)
        ctx.p2_layout(starts, np.asarray(lens, np.int32))
        ids = rng.permutation(len(lens)).astype(np.int32)
        rev = rng.integers(0, 2, len(lens)).astype(np.uint8)
        pieces = [pos(i, r) for i, r in zip(ids, rev)]
        base = np.concatenate(pieces)
        ctx.p2_set_arrangement(ids, rev)
        total = ctx.p2_arrangement_total()
        s_arr = ctx.p2_arrangement_score(total)
        orders, orients = p2._enumeration(k)
        ctx.p2_window_tables(np.asarray(orders, np.int8),
                             np.asarray([[1 if sg == "-" else 0 for sg in r] for r in orients], np.uint8))
        pick = rng.permutation(len(orders) * len(orients))[:24]          # candidates spelled out per window
        for first in range(0, len(ids) - k + 1):
            delta = ctx.p2_score_window(first, k)
            head = np.concatenate(pieces[:first]) if first else np.zeros(0, np.int32)
            tail = np.concatenate(pieces[first + k:]) if first + k < len(ids) else np.zeros(0, np.int32)
            c0 = p2._orient_index(k, ["-" if v else "+" for v in rev[first:first + k]])
            rows = []
            for c in pick:
                o, r = orders[c // len(orients)], orients[c % len(orients)]
                mid = [pos(int(ids[first + j]), sg == "-") for j, sg in zip(o, r)]
                rows.append(np.concatenate([head] + mid + [tail]))
            want = ctx.p2_score(np.stack(rows), total)
            got = s_arr + (delta[pick] - delta[c0]) / total
            assert np.allclose(got, want, rtol=1e-11, atol=0), (first, np.max(np.abs(got - want) / np.abs(want)))


# --------------------------------------------------------------------------------- end to end
def _run_product(name, tmp_path, record=None):
    from hic_genome_assembler_amd import scaffoldToChromosomes as p1, orderGenome as p2
    spec = gc.load_case(name)[0]
    paths = gc.write_case_files(name, str(tmp_path))
    out = str(tmp_path)
    f = lambda k: os.path.join(out, k)  # noqa: E731
    p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                   paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("a.png"), f("b.png"),
                   f("binGroups.txt"), f("assessment.txt"), f("chromosomeGroups.txt"),
                   True, False, spec["min_size"], 0.0, 20, spec["psig"], 5, .2, 100000)
    if record is not None:
        p2.SCORE_HOOK = lambda fast: record.extend(float(v) for v in fast)
    before = os.environ.get("HICMI_P2_TRACE_ORDER")
    if spec.get("numba_trace"):
        os.environ["HICMI_P2_TRACE_ORDER"] = "numba"
    try:
        p2.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       f("chromosomeGroups.txt"), f("chromosomeOrders.txt"), out, "synthetic", f("g.png"),
                       "synthetic genome", f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000)
    finally:
        p2.SCORE_HOOK = None
        if spec.get("numba_trace"):
            if before is None:
                del os.environ["HICMI_P2_TRACE_ORDER"]
            else:
                os.environ["HICMI_P2_TRACE_ORDER"] = before
    return out


@pytest.mark.parametrize("name", CASES)
def test_pipeline_files_identical_to_reference(hic, name, tmp_path):
    """-part1 -part2 through the drop-in modules: every output file byte-identical to what the
    reference wrote for the same inputs, and every objective value it evaluated reproduced."""
    spec, meta, gold, lay, c = gc.load_case(name)
    scores = []
    out = _run_product(name, tmp_path, record=scores)
    for fn in gc.OUTPUT_FILES:
        with open(os.path.join(out, fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn
    ref = gold["costs"]
    assert len(scores) == len(ref)
    got = np.array(scores)
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), ok)
    assert np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])) < 1e-10


@pytest.mark.parametrize("name", CASES)
def test_pipeline_fused_decision_steps_identical_to_reference(hic, name, tmp_path):
    """The production path: Part 2's inner loops as single C calls (hicmi_p2_decide_window /
    _insertion) and chromosomes ordered concurrently on worker contexts."""
    out = _run_product(name, tmp_path, record=None)
    for fn in gc.OUTPUT_FILES:
        with open(os.path.join(out, fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn


@pytest.mark.parametrize("name", ["n600", "n300_edges"])
def test_resident_path_with_background_file_writer(hic, name, tmp_path):
    """What bench.py times: runResident with the four Part 1 files written by a background thread, Part 2 started from
    the in-memory groups (DeviceMatrix.chromosome_groups) instead of reading chromosomeGroups.txt back, finish_files()
    at the end - the six files must be the reference's."""
    from hic_genome_assembler_amd import orderGenome as p2, scaffoldToChromosomes as p1
    from hic_genome_assembler_amd.hostio import initiateLoci
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    f = lambda k: os.path.join(str(tmp_path), k)  # noqa: E731
    bins = initiateLoci(paths["hicProBedFile"], paths["hicProBiasFile"])
    dm = p1.buildAdjacencyMatrix(paths["hicProMatrixFile"], bins)
    try:
        p1.runResident(dm, bins, paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), f("binGroups.txt"),
                       f("assessment.txt"), f("chromosomeGroups.txt"), spec["min_size"], 0.0, spec["psig"], overlap_files=True)
        assert dm.chromosome_groups is not None
        p2.runResident(p2.GenomeMatrix(dm.ctx), dm.kept_bins, f("chromosomeGroups.txt"), f("chromosomeOrders.txt"),
                       f("plotOrder.txt"), spec["n_scaffolds"], spec["scan_scaffolds"], 100000,
                       chromosomeList=dm.chromosome_groups, on_native_phase=dm.release_files)
        dm.finish_files()
    finally:
        dm.ctx.close()
    assert dm.chromosome_groups == p2.readChromsFromFile(f("chromosomeGroups.txt"))
    for fn in gc.OUTPUT_FILES:
        with open(f(fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn


def test_cli_drop_in(hic, tmp_path):
    from hic_genome_assembler_amd import run_hicAssembler, synth
    name = "n160"
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    cfg = synth.write_config(str(tmp_path / "config.txt"), paths, str(tmp_path / "out"), str(tmp_path / "plots"),
                             lay.resolution, min_size=spec["min_size"], modularity=0.0, psig=spec["psig"],
                             n_scaffolds=spec["n_scaffolds"], scan_scaffolds=spec["scan_scaffolds"])
    run_hicAssembler.main(["-part1", "-part2", "-c", cfg])
    for fn in gc.OUTPUT_FILES:
        with open(str(tmp_path / "out" / fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn


@pytest.mark.parametrize("env", [{"HICMI_P2_HOST_INSERT": "1"}, {"HICMI_P2_INS_MAXC": "1"}, {"HICMI_P2_WINDOW_DIRECT": "1"},
                                 {"HICMI_PART2_LOCKSTEP": "0"}, {"HICMI_P2_WINDOW_VALU": "1"}, {"HICMI_P2_INSB_SPLIT": "1"},
                                 {"HICMI_PART2_WORKERS": "1"}, {"HICMI_PART2_WORKERS": "3"}, {"HICMI_REPARSE_FOR_PART2": "1"},
                                 {"HICMI_NNCHAIN_NO_COMPACT": "1"}, {"HICMI_NNCHAIN_W1": "0"},
                                 {"HICMI_NNCHAIN_W1_MAXS": "3", "HICMI_NNCHAIN_W1_COLS": "64"},
                                 {"HICMI_TIED_FULL": "1", "HICMI_PRESORT_FROM": "100"}],
                         ids=["host-decides-every-step", "device-with-host-steps-on-ties", "per-candidate-window-kernels",
                              "one-queue-per-chromosome", "window-tables-on-the-vector-alu", "base-term-as-its-own-launch",
                              "one-part2-worker", "three-part2-workers", "matrix-parsed-again-for-part2",
                              "nn-chain-without-compaction", "nn-chain-on-the-1024-lane-kernels", "nn-chain-three-narrow-slices",
                              "tied-rows-through-the-full-network"])
def test_insertion_paths_agree(hic, tmp_path, env):
    """orderRemainderScaffolds runs with the per-step decisions on the device (k_part2_insert.hip).  The same
    golden files must come out when the host decides every step, and when the device's short list is capped
    at one candidate so that every tie falls back to a host step in the middle of the queue; likewise with the
    windows scored candidate by candidate instead of from placement tables (k_part2_window.hip) and with one
    insertion queue per chromosome instead of the lock step (the switches are read once per process, hence the
    subprocess).  The same harness covers every other switch that selects product code and has no test of its own:
    the window tables' outside term on the vector ALU instead of the matrix cores, the BASE term as its own launch, the
    number of Part 2 worker threads, the text matrix parsed again for Part 2 like the reference, the nn-chain without
    compaction, on the 1,024-lane kernels of rounds 1-2, and on three 64-column slices of the one-wave kernel."""
    import subprocess
    import sys
    from hic_genome_assembler_amd import synth
    for name in ("n160", "n300_edges"):
        spec, meta, gold, lay, c = gc.load_case(name)
        work = tmp_path / name
        work.mkdir()
        paths = gc.write_case_files(name, str(work))
        cfg = synth.write_config(str(work / "config.txt"), paths, str(work / "out"), str(work / "plots"),
                                 lay.resolution, min_size=spec["min_size"], modularity=0.0, psig=spec["psig"],
                                 n_scaffolds=spec["n_scaffolds"], scan_scaffolds=spec["scan_scaffolds"])
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        res = subprocess.run([sys.executable, os.path.join(root, "run_hicAssembler.py"), "-part1", "-part2", "-c", cfg],
                             env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        for fn in gc.OUTPUT_FILES:
            with open(str(work / "out" / fn)) as fh:
                assert fh.read() == gc.golden_text(name, fn), (name, fn)


def test_part2_full_size_fixed_point_properties(hic, orc, tmp_path):
    """Part 2 at the size of a BASELINE 16k-map chromosome (~1,500 bins, ~110 scaffolds), checked
    through properties the oracle can afford: the final order is a fixed point of the sliding-window
    search (no candidate of a window scores strictly above the final literal score), the literal score
    of the final order is bit-identical between GPU and oracle, and fast vs literal scores agree."""
    from hic_genome_assembler_amd import orderGenome as p2, synth
    from hic_genome_assembler_amd.hostio import Bin
    n = 1500
    lay = synth.make_layout(n, seed=11, n_chrom=1, mean_scaffold_bins=13.0)
    c = synth.dense_contacts(lay, seed=11, sinkhorn_iters=10)
    bins = [Bin(int(lay.bin_ids[k]), lay.scaffold_names[lay.scaffold_of_bin[k]], 0, 0, 1.0, 0.) for k in range(n)]
    group = [[b.ID, b.chrom] for b in bins]
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        gm = p2.GenomeMatrix(ctx)
        ordered = p2.orderChromosome(group, gm, bins, nScaffolds=6, scanScaffolds=5)
        best = p2.orderChromosome.last_cost
        view, _od = p2.giveNewAdjMat(gm, ordered, bins)
        total = view.total()
        ids, rev = view.layout.describe(ordered)
        row = view.layout.node_row(ids, rev)
        exact = ctx.p2_score_exact(row[None, :], total)[0]
        fast = ctx.p2_score(row[None, :], total)[0]
        # one more pass over every window must not find anything
        view.layout.tables(5)
        floor = max(best, exact)          # `total` is re-rounded for the final order: allow for that last bit
        _i, _r, best2, _cf, improved = ctx.p2_scan_pass(ids, rev, 5, total, floor, None)
        assert not improved and best2 == floor
        # all 1920 candidates of two windows, literal scores on the oracle
        orders, orients = p2._enumeration(5)
        L = orc.lib()
        sel_rows = view.layout.node_row(np.arange(len(view.layout.start)), np.zeros(len(view.layout.start), int))
        sub_index = np.array([gm.bin_index(bins)[b] for s in sorted(view.layout.sid, key=view.layout.sid.get)
                              for b in sorted(p2_scaffold_bins(ordered, s))])
        sub = np.ascontiguousarray(c[np.ix_(sub_index, sub_index)])
        assert len(sel_rows) == n
        lit_final = L.hio_cost_literal(orc._dp(sub), n, orc._ip(row), n, total)
        assert exact == lit_final
        assert abs(fast - exact) <= 1e-11 * abs(exact)
        assert abs(exact - best) <= 1e-12 * best              # same arrangement, totals rounded in different orders
        for first in (0, len(ids) // 2):
            fast_w, row_of = p2._window_scores(view, ordered, first, 5)
            lit = np.array([L.hio_cost_literal(orc._dp(sub), n, orc._ip(np.ascontiguousarray(row_of(ci), dtype=np.int32)), n, total)
                            for ci in range(len(fast_w))])
            assert np.max(np.abs(fast_w - lit) / np.abs(lit)) < 1e-11
            assert not np.any(lit > floor)                    # fixed point of scanOrdering


def p2_scaffold_bins(ordered, name):
    for s in ordered:
        if s.name == name:
            return s.binList
    raise KeyError(name)


# --------------------------------------------------------------------------------- BASELINE sizes: properties
@pytest.mark.parametrize("n", [16000, 20000])          # 20000: rows longer than one 16384-element sort tile
def test_full_size_properties(hic, n):
    import torch
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(n, seed=1)
    dev = torch.device("cuda:0")
    c = synth.dense_contacts_torch(lay, dev, seed=1, sinkhorn_iters=8)
    torch.cuda.synchronize()
    with hic.Context(0) as ctx:
        ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
        np_sum, seq = ctx.row_sums()
        t_sum = c.sum(dim=1).cpu().numpy()
        assert np.allclose(np_sum, t_sum, rtol=1e-12) and np.allclose(seq, t_sum, rtol=1e-12)
        leaves, z = ctx.upgma()
        assert sorted(leaves.tolist()) == list(range(n))                       # a permutation
        assert np.all(np.diff(z[:, 2]) >= 0)                                   # heights sorted
        assert z[-1, 3] == n and np.all(z[:, 0] < z[:, 1])
        assert np.all(z[:, 3] >= 2)
        # planted chromosomes come out as contiguous runs of the leaf order
        chrom = lay.chrom_of_bin[leaves]
        assert np.count_nonzero(np.diff(chrom) != 0) == len(set(chrom.tolist())) - 1
        ctx.rank_matrix(leaves)
        for r in [0, 1, n // 2, n - 1]:
            R = ctx.rank_rows(r, 1)[0].astype(np.int64)
            inv = ctx.rank_rows(r, 1, inverse=True)[0].astype(np.int64)
            s = ctx.similarity_row(r)
            assert sorted(R.tolist()) == list(range(n))
            assert np.all(np.diff(s[R]) <= 0)                                  # descending similarity
            assert np.array_equal(inv[R], np.arange(n))
        sig, x = ctx.cut_scan(0, n, 0.05, want_x=True)
        for i in [1, 2, 1000, n - 1]:
            R = ctx.rank_rows(i, 1)[0].astype(np.int64)
            pr = R[:i]
            assert x[i] == np.count_nonzero((pr >= 0) & (pr <= i))


def test_plot_percentiles_and_downsample(hic):
    """the colour limits are numpy.percentile itself (exact order statistics selected on the
    device + NumPy's interpolation), the picture the block means of the transformed, permuted matrix."""
    rng = np.random.default_rng(21)
    for name in ("n300_edges", "n600"):
        spec, meta, gold, lay, c = gc.load_case(name)
        n = len(c)
        with hic.Context(0) as ctx:
            ctx.set_contacts(c)
            np_sum, seq_sum = ctx.row_sums()
            if np.any(np_sum == 0):                                   # the edge-case fixture has empty bins (S2C:100-136)
                keep = np.flatnonzero(np_sum != 0)
                ctx.compact(keep)
                c = np.ascontiguousarray(c[np.ix_(keep, keep)])
                n = len(c)
                np_sum, seq_sum = ctx.row_sums()
            dist = (1.0 - c / np_sum[:, None]) + 1.0
            sim = seq_sum[:, None] * (1.0 - (dist - 1.0))
            perm = rng.permutation(n).astype(np.int32)
            sub = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)[::-1].copy()
            for kind, mat in ((0, c), (1, dist), (2, sim)):
                for order in (None, perm, sub):
                    view = mat if order is None else mat[np.ix_(order, order)]
                    q = [1, 98] if kind else [2, 98]
                    got = ctx.plot_percentiles(kind, order, q + [0, 100, 50, 33.3])
                    want = np.percentile(view, q + [0, 100, 50, 33.3])
                    assert np.array_equal(got, want), (name, kind, got, want)
                    m = len(view)
                    for px in (m, 7, 64):
                        img = ctx.plot_downsample(kind, order, px)
                        edges = (np.arange(px + 1, dtype=np.int64) * m) // px
                        rows = np.add.reduceat(view, edges[:-1], axis=0)
                        blocks = np.add.reduceat(rows, edges[:-1], axis=1)
                        cnt = np.diff(edges)
                        assert np.allclose(img, blocks / (cnt[:, None] * cnt[None, :]), rtol=1e-12, atol=0), (name, kind, px)


def test_cli_writes_plots(hic, tmp_path, monkeypatch):
    """The drop-in run leaves the reference's plot files (avgCluster, outlined, Chr_i, full genome) next to
    the text outputs, which stay identical to the golden ones."""
    from hic_genome_assembler_amd import run_hicAssembler, synth
    monkeypatch.delenv("HICMI_NO_PLOTS", raising=False)
    name = "n300_edges"
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    cfg = synth.write_config(str(tmp_path / "config.txt"), paths, str(tmp_path / "out"), str(tmp_path / "plots"),
                             lay.resolution, min_size=spec["min_size"], modularity=0.0, psig=spec["psig"],
                             n_scaffolds=spec["n_scaffolds"], scan_scaffolds=spec["scan_scaffolds"])
    run_hicAssembler.main(["-part1", "-part2", "-c", cfg])
    for fn in gc.OUTPUT_FILES:
        with open(str(tmp_path / "out" / fn)) as fh:
            assert fh.read() == gc.golden_text(name, fn), fn
    pngs = sorted(p for p in os.listdir(str(tmp_path / "plots")) if p.endswith(".png"))
    assert len(pngs) >= 4 and any(p.startswith("Chr_1") for p in pngs), pngs
    for p in pngs:
        assert os.path.getsize(str(tmp_path / "plots" / p)) > 10000, p


def test_louvain_tail_same_on_gpu_and_oracle(hic, tmp_path, monkeypatch):
    """modularity = .05 (the reference's shipped default): the tail's similarity cells come from the device;
    with the same seed the GPU run and the oracle-backed run must write the same groups (the Louvain code is
    host-side and deterministic, so this pins the device-side inputs of SURVEY 8f N3)."""
    from fake_context import OracleContext
    from hic_genome_assembler_amd import _lib, scaffoldToChromosomes as p1
    name = "n400_default"
    spec, meta, gold, lay, c = gc.load_case(name)
    paths = gc.write_case_files(name, str(tmp_path))
    texts = {}
    for run in ("gpu", "oracle"):
        if run == "oracle":
            monkeypatch.setattr(_lib, "Context", OracleContext)
        out = tmp_path / run
        out.mkdir()
        f = lambda k: str(out / k)  # noqa: E731
        p1.runPipeline(paths["hicProBedFile"], paths["hicProBiasFile"], paths["hicProMatrixFile"],
                       paths["hicProScaffSizeFile"], f("dendrogramOrder.txt"), False, False, f("binGroups.txt"),
                       f("assessment.txt"), f("chromosomeGroups.txt"), True, False, spec["min_size"], 0.05, 4,
                       spec["psig"], 5, 5, lay.resolution)
        texts[run] = [open(f(k)).read() for k in ("binGroups.txt", "assessment.txt", "chromosomeGroups.txt")]
    assert texts["gpu"] == texts["oracle"]
    assert texts["gpu"][0].count("### Chromosome group") >= 2


def test_api_misuse_is_reported_not_executed(hic):
    """Calls in the wrong order or with operands that do not match the resident matrix must come back as
    HicmiError (a negative HICMI_E* code and a message) - never as a kernel launched on bad shapes."""
    spec, meta, gold, lay, c = gc.load_case("n160")
    n = len(c)
    with hic.Context(0) as ctx:
        with pytest.raises(hic.HicmiError):
            ctx.upgma()                                              # no contact matrix yet
        with pytest.raises(hic.HicmiError):
            ctx.plot_downsample(0, None, 4)
        ctx.set_contacts(c)
        with pytest.raises((hic.HicmiError, ValueError)):
            ctx.rank_matrix(list(range(n - 1)))                      # order of the wrong length
        with pytest.raises(hic.HicmiError):
            ctx.cut_scan(0, n, .05)                                  # rank matrix not built
        with pytest.raises(hic.HicmiError):
            ctx.plot_downsample(0, None, n + 1)                      # more pixels than cells
        with pytest.raises(hic.HicmiError):
            ctx.plot_percentiles(3, None, [50])                      # unknown transform
        with pytest.raises(hic.HicmiError):
            ctx.plot_percentiles(0, [n], [50])                       # row outside the matrix
        with pytest.raises(hic.HicmiError):
            ctx.p2_total()                                           # nothing selected
        ctx.p2_select(list(range(20)))
        with pytest.raises(hic.HicmiError):
            ctx.p2_set_arrangement([0], [0])                         # no layout
        ctx.p2_layout([0, 10], [10, 10])
        with pytest.raises(hic.HicmiError):
            ctx.p2_set_arrangement([0, 0], [0, 0])                   # a scaffold twice
        ctx.p2_set_arrangement([0, 1], [0, 1])
        with pytest.raises(hic.HicmiError):
            ctx.p2_score_window(0, 2)                                # window tables not loaded
        with pytest.raises(hic.HicmiError):
            ctx.p2_insert_all([0, 1], [0, 0], [1])                   # scaffold already placed
        # the context is still usable afterwards
        leaves, z = ctx.upgma()
        assert sorted(leaves.tolist()) == list(range(n))


def test_fp32_contacts_are_widened_exactly(hic):
    """BASELINE configs[4] stores the map as fp32: hicmi_set_contacts_host_f32 must leave exactly float64(float32(x))
    on the device (in-place widening, odd and even cell counts), so every downstream result equals the fp64 path's on
    the widened matrix."""
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 64, 301, 1000):
        a32 = (rng.random((n, n)) * 100).astype(np.float32)
        a32 = ((a32 + a32.T) / 2).astype(np.float32)
        wide = a32.astype(np.float64)
        with hic.Context(0) as c32, hic.Context(0) as c64:
            c32.set_contacts(a32)
            c64.set_contacts(wide)
            for got, want in zip(c32.row_sums(), c64.row_sums()):
                assert np.array_equal(got, want), n
            if n >= 7:
                assert np.array_equal(c32.plot_downsample(0, None, n), wide)           # the resident cells themselves
                l32, z32 = c32.upgma()
                l64, z64 = c64.upgma()
                assert np.array_equal(z32, z64) and np.array_equal(l32, l64)


@pytest.mark.parametrize("n,sorter", [(16500, "bitonic"), (16500, "radix"), (8193, "radix"), (64000, "bitonic"), (64000, "radix")])
def test_row_sort_across_tiles_with_ties(hic, monkeypatch, n, sorter):
    """Rows longer than one tile (8192 elements for the LSD radix kernel, 16384 for the bitonic networks) are sorted
    tile by tile and combined - by lower bounds in the other tiles / by merge levels through the scratch buffer; with
    quantised contacts (heavy ties) the order must still be "stable ascending, reversed" of the similarity row.
    64,000 columns: the row length of BASELINE configs[4]."""
    import torch
    if sorter == "radix":                                      # the default is the bitonic network (faster on MI355X)
        monkeypatch.setenv("HICMI_SORT_RADIX", "1")
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    c = torch.randint(0, 40, (n, n), generator=g, device="cuda:0").to(torch.float64)
    c = torch.triu(c) + torch.triu(c, 1).T
    c.fill_diagonal_(3.0)
    torch.cuda.synchronize()
    with hic.Context(0) as ctx:
        ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
        ctx.row_sums()
        order = np.random.default_rng(2).permutation(n).astype(np.int32)
        ctx.rank_matrix(order)
        for r in (0, 3, 8191, 8192, n - 1):
            sim = ctx.similarity_row(r)
            want = np.argsort(sim, kind="stable")[::-1]
            got = ctx.rank_rows(r, 1)[0].astype(np.int64)
            assert np.array_equal(got, want), r
            inv = ctx.rank_rows(r, 1, inverse=True)[0].astype(np.int64)
            assert np.array_equal(inv[want], np.arange(n)), r
            assert len(np.unique(sim)) < n // 10                               # the ties are really there


@pytest.mark.parametrize("n,tie_at,ties_by", [
    (3000, None, "runs"), (3000, 7, "runs"), (3000, 15, "runs"), (3000, 1023, "runs"), (3000, 15, "resort"),
    (2500, "many", "runs"), (2500, "many", "resort"), (2500, "most", "runs"), (2500, "most", "resort"),
    (20000, None, "runs"), (20000, 16383, "runs")])
def test_rows_sorted_beside_the_chain_equal_rows_sorted_after_it(hic, monkeypatch, n, tie_at, ties_by):
    """hicmi_upgma sorts every row in storage numbering on a second stream while the nn-chain runs, and hicmi_rank_matrix
    re-addresses those rows by the leaf order; in a row holding equal similarities the order inside each run of equal
    values depends on the labels and is made afterwards (k_rank_rows_tied; HICMI_PRESORT_TIES=resort: by sorting those
    rows again in full, giving the pre-sort up when most rows need it).  Either way the rank rows must be those of the
    plain path (HICMI_NO_PRESORT=1) - which the other tests pin to the oracle.  tie_at = p: ONE pair of equal values in
    one row, at ascending positions p, p + 1 of the sorted row - inside a lane's 16 elements (7), across two lanes (15),
    across two waves (1023), across two 16384-element tiles (16383)."""
    monkeypatch.setenv("HICMI_PRESORT_FROM", "1000")
    if ties_by == "resort":
        monkeypatch.setenv("HICMI_PRESORT_TIES", "resort")
    rng = np.random.default_rng(17)
    if tie_at == "most":
        c = rng.integers(0, 40, size=(n, n)).astype(np.float64)
        c = np.triu(c) + np.triu(c, 1).T
    else:
        c = rng.random((n, n)) + 0.01
        c = np.triu(c) + np.triu(c, 1).T
    tied = []
    if isinstance(tie_at, int):
        r = 5
        by_value = np.argsort(c[r], kind="stable")
        ja, jb = int(by_value[tie_at]), int(by_value[tie_at + 1])
        c[r, ja] = c[ja, r] = c[r, jb]
        tied = [r]
    elif tie_at == "many":
        tied = list(range(100, 600))
        for r in tied:
            j1, j2 = 700 + (r % 900), 1700 + (r % 700)
            c[r, j2] = c[j2, r] = c[r, j1]
    order_kind = "leaves" if n != 3000 or tie_at in (None, 15) else "random"
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        order = leaves if order_kind == "leaves" else rng.permutation(n).astype(np.int32)
        ctx.rank_matrix(order)
        state, n_tied = ctx.presort_state()
        rows = list(range(n)) if n <= 3000 else [0, 1, 5, 4097, n - 1] + [int(np.where(order == r)[0][0]) for r in tied]
        got = np.stack([ctx.rank_rows(a, 1, inverse=True)[0] for a in rows]) if n > 3000 else ctx.rank_rows(inverse=True)
        monkeypatch.setenv("HICMI_NO_PRESORT", "1")
        ctx.set_contacts(c)
        leaves2, _z = ctx.upgma()
        assert np.array_equal(leaves, leaves2)
        ctx.rank_matrix(order)
        assert ctx.presort_state() == (0, 0)
        want = np.stack([ctx.rank_rows(a, 1, inverse=True)[0] for a in rows]) if n > 3000 else ctx.rank_rows(inverse=True)
        if tied:                                            # the tie is really there, and in the expected row
            a = int(np.where(order == tied[0])[0][0])
            sim = ctx.similarity_row(a)
            assert len(np.unique(sim)) == n - 1
    assert np.array_equal(got, want)
    if tie_at == "most":
        assert (state, n_tied) == ((2, n) if ties_by == "resort" else (1, n))
    else:
        assert state == 1
        # (two close contacts can round to the same similarity: ~n/1000 more rows than the planted ones may be flagged)
        assert len(tied) <= n_tied <= len(tied) + 3 + n // 400


def test_rows_sorted_beside_the_chain_when_no_workgroup_takes_a_row(hic, monkeypatch, capfd):
    """Beside a chain that holds one XCD the pre-sort's workgroups on that XCD leave and the others take the rows from a
    counter.  Where a workgroup runs is the hardware's choice: if ALL of them left (test hook) no row is sorted - the
    counter tells, and hicmi_rank_matrix sorts then and there; the rank rows are those of the plain path."""
    monkeypatch.setenv("HICMI_PRESORT_FROM", "1000")
    monkeypatch.setenv("HICMI_TEST_PRESORT_ALL_LEAVE", "1")
    rng = np.random.default_rng(23)
    n = 2500
    c = rng.random((n, n)) + 0.01
    c = np.triu(c) + np.triu(c, 1).T
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        ctx.rank_matrix(leaves)
        assert ctx.presort_state()[0] == 0
        assert "rows were handed out" in capfd.readouterr().err
        got = ctx.rank_rows(inverse=True)
        monkeypatch.delenv("HICMI_TEST_PRESORT_ALL_LEAVE")
        monkeypatch.setenv("HICMI_NO_PRESORT", "1")
        ctx.set_contacts(c)
        leaves2, _z = ctx.upgma()
        ctx.rank_matrix(leaves2)
        want = ctx.rank_rows(inverse=True)
    assert np.array_equal(leaves, leaves2) and np.array_equal(got, want)


@pytest.mark.parametrize("n", [3000, 20000, 40000])        # 16 / 32 / 64 elements per lane
def test_rank_rows_with_short_runs_of_equal_similarities(hic, monkeypatch, n):
    """fp32-valued contacts give every row a few collisions; such rows are finished by k_rank_rows_short_runs (two rounds of
    16-element block sorts: a run of at most 8 elements lies in an aligned block or in one shifted by 8), rows with a longer
    run by the full network.  Planted here: runs of 2 ... 8 equal values at random sorted positions of many rows (hence at
    every offset against the block, lane and wave boundaries), runs of 9 and 13 in a few rows, and rows that hold both.
    The argsort rows must be numpy's stable argsort, reversed, of the very similarity row the library computes."""
    monkeypatch.setenv("HICMI_PRESORT_FROM", "1000")
    rng = np.random.default_rng(n)
    c = rng.random((n, n), dtype=np.float32).astype(np.float64) + 0.01        # fp32-valued: a few natural collisions as well
    c = np.triu(c) + np.triu(c, 1).T
    rows = rng.choice(n, size=60, replace=False)
    for t, r in enumerate(rows):
        by_value = np.argsort(c[r], kind="stable")
        lengths = [2 + (t % 7)] * 3                                            # short runs: 2 .. 8
        if t % 10 == 3:
            lengths += [9]
        if t % 10 == 7:
            lengths = [13] + lengths
        for L in lengths:
            p0 = int(rng.integers(0, n - L))
            cols = by_value[p0:p0 + L]
            c[r, cols] = c[r, cols[0]]
            c[cols, r] = c[r, cols[0]]
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        ctx.rank_matrix(leaves)
        state, n_tied = ctx.presort_state()
        assert state == 1 and n_tied >= len(rows)
        where = {int(r): k for k, r in enumerate(leaves)}
        check = [where[int(r)] for r in rows] + rng.integers(0, n, 30).tolist() + [0, n - 1]
        for a in check:
            sim = ctx.similarity_row(a)
            assert np.array_equal(ctx.rank_rows(a, 1)[0].astype(np.int64), np.argsort(sim, kind="stable")[::-1]), a


@pytest.mark.parametrize("n", [20000, 40000])              # 32 / 64 elements per lane in k_rank_rows_tied
def test_long_rows_full_of_equal_similarities(hic, monkeypatch, n):
    """Quantised contacts (every row is mostly runs of equal similarities) at row lengths where k_rank_rows_tied keeps 32
    and 64 keys per lane: sampled rows against numpy's stable argsort of the same similarity row, and against the path
    without the pre-sort."""
    import torch
    g = torch.Generator(device="cuda:0")
    g.manual_seed(11)
    c = torch.randint(0, 40, (n, n), generator=g, device="cuda:0").to(torch.float64)
    c = torch.triu(c) + torch.triu(c, 1).T
    c.fill_diagonal_(3.0)
    torch.cuda.synchronize()
    rows = (0, 3, 8191, 16384, n - 1)
    with hic.Context(0) as ctx:
        ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
        leaves, _z = ctx.upgma()
        ctx.rank_matrix(leaves)
        assert ctx.presort_state() == (1, n)
        got = {r: ctx.rank_rows(r, 1, inverse=True)[0].astype(np.int64) for r in rows}
        for r in rows:
            sim = ctx.similarity_row(r)
            want = np.argsort(sim, kind="stable")[::-1]
            assert np.array_equal(got[r][want], np.arange(n)), r
        monkeypatch.setenv("HICMI_NO_PRESORT", "1")
        ctx.set_contacts_device(c.data_ptr(), n, keepalive=c)
        ctx.upgma()
        ctx.rank_matrix(leaves)
        assert ctx.presort_state() == (0, 0)
        for r in rows:
            assert np.array_equal(ctx.rank_rows(r, 1, inverse=True)[0].astype(np.int64), got[r]), r


def _scan_loops_both_ways(hic, c, monkeypatch, capsys, min_size, min_frac, psig, cuts_for_filter=None):
    """pre_process_all_matrix_breakpoints and filter_noisy_breakpoints on one clustered map, with the loops' decisions on
    the device (default) and on the host (HICMI_HOST_SCANS=1): cuts and printed messages of both."""
    from hic_genome_assembler_amd import scaffoldToChromosomes as s2c
    out = {}
    with hic.Context(0) as ctx:
        ctx.set_contacts(c)
        leaves, _z = ctx.upgma()
        ctx.rank_matrix(leaves)
        rm = s2c.RankMatrix(ctx)
        for how in ("device", "host"):
            if how == "host":
                monkeypatch.setenv("HICMI_HOST_SCANS", "1")
            else:
                monkeypatch.delenv("HICMI_HOST_SCANS", raising=False)
            capsys.readouterr()
            cuts = s2c.pre_process_all_matrix_breakpoints(rm, min_size=min_size, min_frac=min_frac, psig=psig)
            kept = s2c.filter_noisy_breakpoints(rm, cuts_for_filter if cuts_for_filter is not None else cuts, psig=psig)
            out[how] = (cuts, kept, capsys.readouterr().out)
    return out


@pytest.mark.parametrize("name,min_size,min_frac", [("n160", 5, .05), ("n600", 5, .05), ("n600", 2, .05), ("n600", 9, .3),
                                                    ("n500_sparse", 5, .05), ("n300_edges", 5, .05), ("n600", 1, .0)])
def test_scan_loops_on_the_device_equal_the_host_loops(hic, monkeypatch, capsys, name, min_size, min_frac):
    """hicmi_first_pass_cuts / hicmi_filter_cuts take the decisions of S2C:413-551 / 553-727 in kernels; the per-scan host
    loops (which the golden pipelines pin to the reference) must give the same cuts and print the same lines."""
    spec, meta, gold, lay, c = gc.load_case(name)
    c = np.asarray(c, np.float64)
    keep = c.sum(axis=1) != 0                                # (removeRows: n300_edges has empty rows)
    c = np.ascontiguousarray(c[keep][:, keep])
    both = _scan_loops_both_ways(hic, c, monkeypatch, capsys, min_size, min_frac, .05)
    assert both["device"][0] == both["host"][0]
    assert both["device"][1] == both["host"][1]
    assert both["device"][2] == both["host"][2]
    if min_size == 5 and min_frac == .05:
        assert len(both["host"][0]) > 0                      # (the case does produce cuts)


@pytest.mark.parametrize("n,seed", [(3000, 3), (8000, 4)])
def test_scan_loops_on_the_device_synthetic_maps(hic, monkeypatch, capsys, n, seed):
    """The same on synthetic maps with planted chromosomes (dozens of cuts, hundreds of scans), and with a candidate
    list denser than anything the first pass produces (every 40th index): many rounds, restarts and merged cuts."""
    from hic_genome_assembler_amd import synth
    lay = synth.make_layout(n, seed=seed)
    c = synth.dense_contacts(lay, seed=seed)
    both = _scan_loops_both_ways(hic, c, monkeypatch, capsys, 5, .05, .05)
    assert both["device"] == both["host"]
    assert len(both["host"][0]) >= 5
    dense = list(range(40, n - 40, 40))
    both = _scan_loops_both_ways(hic, c, monkeypatch, capsys, 5, .05, .05, cuts_for_filter=dense)
    assert both["device"] == both["host"]
    assert 0 < len(both["host"][1]) < len(dense)
    if n == 8000:
        # more candidates than k_filter_decide keeps in LDS (2,048): its lists are walked in global memory
        very_dense = list(range(6, n - 6, 3))
        assert len(very_dense) > 2048
        both = _scan_loops_both_ways(hic, c, monkeypatch, capsys, 5, .05, .05, cuts_for_filter=very_dense)
        assert both["device"] == both["host"]
        assert 0 < len(both["host"][1]) < len(very_dense)
