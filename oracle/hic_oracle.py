"""CPU oracle: a restatement of the reference's Part 1 / Part 2 hot path.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (hic_genome_assembler_amd/) never does and fails loudly
when its HIP library is missing.

Every function cites the reference lines it follows (paths relative to
/root/reference/HIC_ASSEMBLER; S2C = scaffoldToChromosomes.py, OG = orderGenome.py).  fp64
arithmetic with the reference's order of operations; the native third-party pieces
(NumPy pairwise sum, SciPy nn_chain / label / dendrogram) are restated in oracle_c.c and
pinned against the installed NumPy 2.2.6 / SciPy 1.15.3 in tests/test_oracle_cpu.py.  The
hypergeometric tail is scipy.stats.hypergeom.sf itself - the routine the reference calls
(S2C:367), un-vendored and unpinned by the reference (packageInstallCommands.txt:9-17).

Pinning: tests/test_oracle_cpu.py runs this oracle on the datasets of tests/golden/*/ and
compares with what the reference itself produced there (oracle/gen_golden.py).

Documented deviations (do not change results):
 * the unused frozen distribution built at S2C:364 is not constructed;
 * the window-shrinking retry loop S2C:499-508 is elided: its scores are bounded by
   ws < min_size while S2C:488 compares with min_size, so it can never yield a cut;
 * numpy.argsort's undefined tie order (S2C:1132) is fixed to "stable ascending, reversed".
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
from scipy.stats import hypergeom as _hypergeom

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "liboracle.so")
        src = os.path.join(_HERE, "oracle_c.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = ctypes.CDLL(so)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int32)
        L.hio_np_sum.restype = ctypes.c_double
        L.hio_np_sum.argtypes = [dp, ctypes.c_long, ctypes.c_long]
        L.hio_seq_sum.restype = ctypes.c_double
        L.hio_seq_sum.argtypes = [dp, ctypes.c_long, ctypes.c_long]
        L.hio_nn_chain_average.restype = ctypes.c_int
        L.hio_nn_chain_average.argtypes = [dp, ctypes.c_long, ctypes.c_long, dp]
        L.hio_label.restype = ctypes.c_int
        L.hio_label.argtypes = [dp, ctypes.c_long, dp]
        L.hio_leaf_order.restype = ctypes.c_int
        L.hio_leaf_order.argtypes = [dp, ctypes.c_long, ip]
        L.hio_total_upper.restype = ctypes.c_double
        L.hio_total_upper.argtypes = [dp, ctypes.c_long, ip, ctypes.c_long]
        L.hio_cost_literal.restype = ctypes.c_double
        L.hio_cost_literal.argtypes = [dp, ctypes.c_long, ip, ctypes.c_long, ctypes.c_double]
        L.hio_set_trace_order.restype = None
        L.hio_set_trace_order.argtypes = [ctypes.c_int]
        L.hio_cost_literal_batch.restype = None
        L.hio_cost_literal_batch.argtypes = [dp, ctypes.c_long, ip, ctypes.c_long, ctypes.c_long,
                                             ctypes.c_double, dp]
        _LIB = L
    return _LIB


def set_trace_order(order: str):
    """"numpy" (default): numpy.trace's pairwise sums in the cost loop; "numba": the sequential loop Numba compiles for
    numpy.trace inside costFunction_numba (orderGenome.py:184-191).  The totals (OG:343,448,506) are NumPy's either way."""
    lib().hio_set_trace_order(1 if order == "numba" else 0)


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


# --------------------------------------------------------------------------------------------
# summation primitives
def np_row_sums(mat: np.ndarray) -> np.ndarray:
    """``row.sum()`` of every row as NumPy computes it (S2C:147; also S2C:112)."""
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    L = lib()
    return np.array([L.hio_np_sum(_dp(mat[i]), mat.shape[1], 1) for i in range(mat.shape[0])])


def seq_row_sums(mat: np.ndarray) -> np.ndarray:
    """Python ``sum(numpy.asarray(r))`` of every row: left to right (S2C:134)."""
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    L = lib()
    return np.array([L.hio_seq_sum(_dp(mat[i]), mat.shape[1], 1) for i in range(mat.shape[0])])


# --------------------------------------------------------------------------------------------
# Part 1 data model + loaders
class Bin:
    """S2C:24-33."""
    __slots__ = ("ID", "chrom", "start", "stop", "bias", "rowSum")

    def __init__(self, ID, chrom, start, stop, bias, rowSum):
        self.ID, self.chrom, self.start, self.stop, self.bias, self.rowSum = ID, chrom, start, stop, bias, rowSum


def initiate_loci(bed_file, bias_file, bin_ids=None):
    """S2C:35-68 (OG:30-63 when bin_ids is given): line-aligned .bed/.biases read;
    a bias line equal to "nan" drops the bin, an unparsable one becomes 0.0."""
    bins = []
    with open(bed_file) as bed, open(bias_file) as bias:
        while True:
            bl, sl = bed.readline(), bias.readline()
            if not bl:
                break
            cols = bl.strip("\r").strip("\n").split("\t")
            chrom, start, stop, bid = cols[0], int(cols[1]), int(cols[2]), int(cols[3])
            b = sl.strip("\r").strip("\n")
            if bin_ids is not None and bid not in bin_ids:
                continue
            if b != "nan":
                try:
                    v = float(b)
                except Exception:
                    v = 0.0
                bins.append(Bin(bid, chrom, start, stop, v, 0.0))
    return bins


def build_adjacency(matrix_file, bins):
    """S2C:70-98: dense N x N, each triplet mirrored, unknown bin IDs skipped, later lines win."""
    n = len(bins)
    index = {b.ID: i for i, b in enumerate(bins)}
    mat = np.zeros((n, n), dtype=np.float64)
    with open(matrix_file) as fh:
        for line in fh:
            cols = line.strip("\r").strip("\n").split("\t")
            a, b, v = int(cols[0]), int(cols[1]), float(cols[2])
            if a not in index or b not in index:
                continue
            mat[index[a], index[b]] = v
            mat[index[b], index[a]] = v
    return mat


def remove_zero_rows(mat, bins):
    """S2C:100-136 with zeroRows=True, biasVals=False: rows whose NumPy sum is 0 are removed
    with their columns; rowSum becomes the left-to-right Python sum of what is left."""
    sums = np_row_sums(mat)
    drop = [i for i in range(len(bins)) if sums[i] == 0]
    if drop:
        keep = np.array([i for i in range(len(bins)) if sums[i] != 0], dtype=np.int64)
        mat = np.ascontiguousarray(mat[np.ix_(keep, keep)])
        bins = [bins[i] for i in keep]
    rs = seq_row_sums(mat)
    for b, v in zip(bins, rs):
        b.rowSum = float(v)
    return mat, bins


def to_distance(mat):
    """S2C:147: ``(1. - (row/row.sum())) + 1.`` per row - three roundings per element."""
    sig = np_row_sums(mat)
    return (1.0 - (mat / sig[:, None])) + 1.0


def to_similarity(dist, bins):
    """S2C:149: ``rowSum * (1. - (row - 1.))``."""
    rs = np.array([b.rowSum for b in bins], dtype=np.float64)
    return rs[:, None] * (1.0 - (dist - 1.0))


# --------------------------------------------------------------------------------------------
# UPGMA + leaf order
def nn_chain_raw(dist):
    """S2C:194-197: squareform (strict upper triangle, row-major) + SciPy nn_chain('average').
    Returns the merges in merge order: (x, y, height, size)."""
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    n = dist.shape[0]
    zraw = np.zeros((max(n - 1, 0), 4), dtype=np.float64)
    rc = lib().hio_nn_chain_average(_dp(dist), n, dist.shape[1], _dp(zraw))
    assert rc == 0
    return zraw


def label_linkage(zraw, n):
    """SciPy: stable sort by height, union-find relabel (hierarchy.py linkage -> _hierarchy.label)."""
    z = np.zeros_like(zraw)
    rc = lib().hio_label(_dp(np.ascontiguousarray(zraw)), n, _dp(z))
    assert rc == 0
    return z


def leaf_order(z, n):
    """S2C:204: dendrogram(count_sort='ascending') leaves."""
    leaves = np.zeros(n, dtype=np.int32)
    rc = lib().hio_leaf_order(_dp(np.ascontiguousarray(z)), n, _ip(leaves))
    assert rc == 0
    return leaves


def average_cluster_leaves(dist):
    n = dist.shape[0]
    zraw = nn_chain_raw(dist)
    z = label_linkage(zraw, n)
    return leaf_order(z, n), z


# --------------------------------------------------------------------------------------------
# rank matrix + hypergeometric cut scan
def rank_order(sim):
    """S2C:1132 ``argsort(axis=1)[:, ::-1]`` with the tie rule fixed to stable-then-reversed."""
    return np.ascontiguousarray(np.argsort(sim, axis=1, kind="stable")[:, ::-1])


def hyper_geom(x, M, n, N):
    """S2C:352-368 without the unused frozen object: P[X >= x], NaN on invalid arguments."""
    with np.errstate(all="ignore"):
        return _hypergeom.sf(np.asarray(x) - 1, M, n, N)


def sliding_window_scores(sig, half):
    """S2C:370-411 ``break_sigs``: left half-window sum minus right half-window sum, 0 where the
    right half is short (S2C:395-396).  Returns None for the ["NA","NA","NA"] case (S2C:380-381)."""
    sig = np.asarray(sig, dtype=np.int64)
    n = len(sig)
    if half >= n:
        return None
    cs = np.concatenate(([0], np.cumsum(sig)))
    out = np.zeros(n - half, dtype=np.int64)
    i = np.arange(n - half)
    full = i + 2 * half <= n
    ii = i[full]
    out[full] = (cs[ii + half] - cs[ii]) - (cs[ii + 2 * half] - cs[ii + half])
    return out


def first_pass_counts(R, start):
    """S2C:455-459: for rows i > start, x_i = #{v in R[i][:i-start] : start <= v <= i}."""
    n = R.shape[0]
    xs = np.zeros(n, dtype=np.int64)
    for i in range(start + 1, n):
        pr = R[i, : i - start]
        xs[i] = np.count_nonzero((pr >= start) & (pr <= i))
    return xs


def find_matrix_pvalue_breakpoints(R, start, min_size, world_size, psig=.05, trace=None):
    """S2C:413-511 (retry loop S2C:499-508 elided, see module docstring)."""
    n = R.shape[0]
    M = world_size
    loop_count = 0
    xs = first_pass_counts(R, start)
    rows = np.arange(start + 1, n)
    L = rows - start
    while True:
        p = hyper_geom(xs[rows], M, L, L)
        sig = np.concatenate(([0], np.where(p >= psig, 0, 1))).astype(np.int64)   # NaN -> 1 (S2C:466-469)
        if trace is not None:
            trace.append(dict(start=start, M=M, x=xs[rows].copy(), sig=sig.copy()))
        loop_count += 1
        if (sig.sum() / len(sig)) >= .9:
            M = int(M - start)
        else:
            break
        if loop_count >= 5:
            break
    scores = sliding_window_scores(sig, min_size)
    if scores is None:
        return []
    return [int(ii + min_size) for ii in range(len(scores)) if scores[ii] == min_size]


def pre_process_all_matrix_breakpoints(R, min_size=5, min_frac=.05, trace=None):
    """S2C:513-551.  psig is the literal .05 of S2C:535, not the config value."""
    n = R.shape[0]
    stop_ind = int(n - (n * min_frac))
    ind = 0
    cuts = []
    if min_frac == 1:
        return cuts
    while True:
        pre = find_matrix_pvalue_breakpoints(R, ind, min_size, n - ind, psig=.05, trace=trace)
        if len(pre) == 0:
            break
        ind += pre[0]
        cuts.append(ind)
        if (ind >= stop_ind) or ((n - ind) <= min_size):
            break
    return cuts


def filter_noisy_breakpoints(R, original_inds, psig=.05, trace=None):
    """S2C:553-727."""
    if len(original_inds) == 0:
        return []
    n = R.shape[0]
    MD = int(n / 5)
    MAX_ROUNDS = 10 * len(original_inds)
    altered = list(original_inds)
    prev_filtered = {}
    while True:
        start = 0
        filtered = {}
        round_count = 0
        while True:
            if round_count >= MAX_ROUNDS:
                break
            M = n - start
            noise_found = 0
            select_from = 0
            for i, c in enumerate(altered):
                local = c - start
                last = min(n - 1, start + MD)                    # rows farther than MD are forced to 0 (S2C:626-628)
                rows = np.arange(start, last + 1)
                sub = R[start:last + 1, :local]
                x = np.count_nonzero((sub >= start) & (sub <= c), axis=1) if local > 0 else np.zeros(len(rows), np.int64)
                pv = hyper_geom(x, M, local, local)
                pvals = np.zeros(n, dtype=np.int64)
                pvals[rows] = np.where(pv < psig, 1, 0)            # NaN -> 0 here (S2C:633-636)
                if trace is not None:
                    trace.append(dict(start=start, c=c, M=M, x=x.copy(), sig=pvals[rows].copy()))
                sigs = []
                fc_prev = start
                right_most = None
                right_most_ind = None
                for ai_ind, ai in enumerate(altered):
                    ps = pvals[fc_prev:ai] if ai > fc_prev else pvals[0:0]
                    if ai == fc_prev:
                        continue
                    fc_prev = ai
                    if len(ps) == 0:
                        break
                    xx = int(ps.sum())
                    noise_p = float(hyper_geom(xx, M, local, len(ps)))
                    if noise_p < psig:
                        right_most = ai
                        right_most_ind = ai_ind
                        sigs.append(ai)
                if len(sigs) > 0:
                    start = right_most
                    filtered[right_most] = ''
                    noise_found = 1
                    select_from = right_most_ind
                    break
                else:
                    filtered[c] = ''
                    select_from = i
            round_count += 1
            if noise_found == 0:
                break
            altered = altered[select_from:]
        if prev_filtered != filtered:
            altered = sorted(filtered)
            prev_filtered = filtered
        else:
            break
    return sorted(filtered.keys())


# --------------------------------------------------------------------------------------------
# Part 1 writers / assessment
def write_dendrogram_order(labels, leaves, path):
    """S2C:210-220: ``label\\tleaf``, no trailing newline."""
    with open(path, "w") as fh:
        fh.write("\n".join("%s\t%d" % (labels[l], l) for l in leaves))


def write_bin_groupings(cuts, bins, path):
    """S2C:945-964."""
    groups, prev = [], 0
    for c in cuts:
        groups.append(bins[prev:c])
        prev = c
    groups.append(bins[prev:])
    with open(path, "w") as fh:
        for i, g in enumerate(groups):
            fh.write("### Chromosome group " + str(i + 1) + " ###\n")
            for b in g:
                fh.write(str(b.ID) + "\t" + b.chrom + "\t" + str(b.start) + "\t" + str(b.stop) + "\t" + str(b.bias) + "\n")


def read_size_file(path):
    """S2C:968-979."""
    out = {}
    with open(path) as fh:
        for line in fh:
            cols = line.strip("\r").strip("\n").split("\t")
            out[cols[0]] = int(cols[1])
    return out


def read_bin_groupings(path):
    """S2C:981-999."""
    chrom_list, group = [], []
    with open(path) as fh:
        fh.readline()
        for line in fh:
            line = line.strip("\n").strip("\r")
            if line[0] != "#":
                group.append(line)
            else:
                chrom_list.append(group)
                group = []
    chrom_list.append(group)
    return chrom_list


def assess_chromosome_clustering(chrom_list, stats_path, percent_to_assign=51.):
    """S2C:1001-1077."""
    scaffolds = {}
    full = [cc for c in chrom_list for cc in c]
    for node in full:
        bid, scaff = int(node.split("\t")[0]), node.split("\t")[1]
        scaffolds.setdefault(scaff, []).append([bid, scaff])
    for s in scaffolds:
        scaffolds[s] = sorted(scaffolds[s], key=lambda x: x[0])
    final, false_pos, total_assigned = [], 0, 0
    with open(stats_path, "w") as fh:
        for i, c in enumerate(chrom_list):
            fh.write("### Chromosome" + str(i + 1) + " ###\n")
            scaffs = {}
            for line in c:
                bid, scaff = int(line.split("\t")[0]), line.split("\t")[1]
                scaffs.setdefault(scaff, []).append(bid)
            nodes, assigned, fp = [], 0, 0
            fh.write("#Scaffold\tNodesAssigend\tTotalNodes\tAssigned%\n")
            for s, node_list in scaffs.items():
                na, tn = len(node_list), len(scaffolds[s])
                pct = round(((float(na) / float(tn)) * 100.), 2)
                fh.write(str(s) + "\t" + str(na) + "\t" + str(tn) + "\t" + str(pct) + "%\n")
                if pct >= percent_to_assign:
                    nodes += scaffolds[s]
                    assigned += 1
                else:
                    fp += na
            fh.write("Total scaffolds clustered to chromosome " + str(len(scaffs)) + "\n")
            fh.write("Total scaffolds assigned to chromosome " + str(assigned) + "\n")
            if len(nodes) > 0:
                final.append(nodes)
            false_pos += fp
            total_assigned += assigned
            fh.write("####################\n")
        tn = len(full)
        fh.write("Total Nodes " + str(tn) + "\n")
        fh.write("Properly clustered nodes " + str(tn - false_pos) + "\n")
        fh.write("Falsely clustered nodes " + str(false_pos) + "\n")
        fh.write("Total scaffolds assigned to chromosomes " + str(total_assigned) + "\n")
        fh.write("Error rate ~" + str(round((float(false_pos) / float(tn)) * 100., 2)) + "%\n")
    return final


def write_chromosome_groupings(chrom_list, size_dict, path):
    """S2C:1079-1100: groups sorted by total bp, largest first (stable)."""
    sizes = []
    for c in chrom_list:
        scaffs = {cc[1]: '' for cc in c}
        sizes.append(sum(size_dict[s] for s in scaffs))
    order = sorted(range(len(chrom_list)), key=lambda k: sizes[k], reverse=True)
    with open(path, "w") as fh:
        for i, k in enumerate(order):
            fh.write("### Chromosome group " + str(i + 1) + " ###\n")
            for cc in chrom_list[k]:
                fh.write(str(cc[0]) + "\t" + str(cc[1]) + "\n")


def run_part1(bed, bias, matrix, sizes, dendro_file, bin_group_file, assessment_file, chrom_group_file,
              min_size=5, modularity=0.0, psig=.05, trace=None, preloaded=None):
    """S2C:1104-1174 on the hyperGeom=True, hmm=False, modularity=0 path (plots omitted).
    ``preloaded=(matrix, bins)`` skips the text loaders (bench.py's cpu_baseline leg times the same
    resident-matrix path as the GPU)."""
    if modularity not in (0, 0.0, False):
        raise NotImplementedError("oracle covers modularity = 0 only (Louvain tail is unseeded random, S2C:253)")
    if preloaded is not None:
        mat, bins = np.array(preloaded[0], dtype=np.float64), list(preloaded[1])
    else:
        bins = initiate_loci(bed, bias)
        mat = build_adjacency(matrix, bins)
    mat, bins = remove_zero_rows(mat, bins)
    dist = to_distance(mat)
    labels = [b.chrom + "_" + str(b.ID) for b in bins]
    leaves, z = average_cluster_leaves(dist)
    write_dendrogram_order(labels, leaves, dendro_file)
    order = np.asarray(leaves, dtype=np.int64)
    dist = dist[:, order][order]
    bins = [bins[i] for i in order]
    sim = to_similarity(dist, bins)
    R = rank_order(sim)
    initial = pre_process_all_matrix_breakpoints(R, min_size=min_size, min_frac=modularity,
                                                 trace=None if trace is None else trace.setdefault("first_pass", []))
    cuts = filter_noisy_breakpoints(R, initial, psig=psig,
                                    trace=None if trace is None else trace.setdefault("filter", []))
    write_bin_groupings(cuts, bins, bin_group_file)
    size_dict = read_size_file(sizes)
    groups = read_bin_groupings(bin_group_file)
    chr_groups = assess_chromosome_clustering(groups, assessment_file)
    write_chromosome_groupings(chr_groups, size_dict, chrom_group_file)
    if trace is not None:
        trace.update(kept_ids=np.array([b.ID for b in bins]), Z=z, leaves=leaves, dist_reordered=dist, sim=sim, R=R,
                     initial_cuts=initial, cuts=cuts, contacts=mat)
    return cuts


# --------------------------------------------------------------------------------------------
# Part 2
def swap_permutations(elements):
    """OG:381-394: swap-recursion order of all permutations."""
    el = list(elements)
    out = []

    def rec(k):
        if k == len(el):
            out.append(list(el))
            return
        for i in range(k, len(el)):
            el[k], el[i] = el[i], el[k]
            rec(k + 1)
            el[k], el[i] = el[i], el[k]
    rec(0)
    return out


def remove_reverse_duplicates(perms):
    """OG:396-411: keep the first of every (order, reversed order) pair."""
    seen, out = set(), []
    for p in perms:
        t = tuple(p)
        if t[::-1] in seen:
            seen.discard(t[::-1])
        else:
            seen.add(t)
            out.append(p)
    return out


def plus_minus_perms(k):
    """OG:413-430."""
    cands = [["+"] * k]
    for i in range(k):
        cands += swap_permutations(["+"] * i + ["-"] * (k - i))
    seen, out = set(), []
    for p in cands:
        if tuple(p) not in seen:
            seen.add(tuple(p))
            out.append(list(p))
    return out


class Scaffold:
    """OG:239-254."""

    def __init__(self, name, bin_list, orientation):
        self.name, self.binList, self.orientation = name, bin_list, orientation

    def flipOrientation(self):
        self.orientation = "-" if self.orientation == "+" else "+"
        self.binList = self.binList[::-1]


class Part2Oracle:
    """Literal OG:256-586 with the objective evaluated by oracle_c.c (numpy.trace semantics)."""

    def __init__(self, matrix, bins):
        self.matrix = np.ascontiguousarray(matrix, dtype=np.float64)
        self.bin_index = {b.ID: i for i, b in enumerate(bins)}
        self.costs = []           # every evaluation, in the reference's call order

    # --- objective -------------------------------------------------------------------------
    def _global(self, scaff_list):
        return np.array([self.bin_index[n] for s in scaff_list for n in s.binList], dtype=np.int32)

    def total(self, gidx):
        return lib().hio_total_upper(_dp(self.matrix), self.matrix.shape[1], _ip(gidx), len(gidx))

    def cost(self, gidx, total):
        gidx = np.ascontiguousarray(gidx, dtype=np.int32)
        c = lib().hio_cost_literal(_dp(self.matrix), self.matrix.shape[1], _ip(gidx), len(gidx), total)
        self.costs.append(c)
        return c

    # --- OG:256-280 ------------------------------------------------------------------------
    @staticmethod
    def initiate(node_list):
        d = {}
        for bid, scaff in node_list:
            d.setdefault(scaff, Scaffold(scaff, [], "+")).binList.append(bid)
        for s in d.values():
            s.binList = sorted(s.binList)
        lst = sorted(d.values(), key=lambda s: len(s.binList), reverse=True)
        return lst, d

    @staticmethod
    def reorder(order, orients, d):
        """OG:310-321."""
        out = []
        for name, o in zip(order, orients):
            if d[name].orientation != o:
                d[name].flipOrientation()
            out.append(d[name])
        return out

    # --- OG:432-473 ------------------------------------------------------------------------
    def brute_force(self, scaffs, d):
        names = [s.name for s in scaffs]
        orders = remove_reverse_duplicates(swap_permutations(names))
        orients = plus_minus_perms(len(names))
        total = self.total(self._global(scaffs))
        if total == 0:
            return orders[0], orients[0], 0.0
        best, best_order, best_orient = 0., "NA", "NA"
        for o in orders:
            for r in orients:
                lst = self.reorder(o, r, d)
                c = self.cost(self._global(lst), total)
                if c > best:
                    best, best_order, best_orient = c, o, r
        return best_order, best_orient, best

    # --- OG:332-372 ------------------------------------------------------------------------
    def check_all_scores(self, ordered, scaff):
        best, best_i, best_o = 0., 0, "+"
        total = self.total(self._global(ordered + [scaff]))     # adjMat was built with the new scaffold last (OG:484-486)
        for i in range(len(ordered) + 1):
            ordered.insert(i, scaff)
            c = self.cost(self._global(ordered), total)
            if c > best:
                best, best_i, best_o = c, i, ordered[i].orientation
            ordered[i].flipOrientation()
            c = self.cost(self._global(ordered), total)
            if c > best:
                best, best_i, best_o = c, i, ordered[i].orientation
            scaff = ordered.pop(i)
        if scaff.orientation != best_o:
            scaff.flipOrientation()
        ordered.insert(best_i, scaff)
        return ordered, best

    # --- OG:495-549 ------------------------------------------------------------------------
    def scan_ordering(self, ordered, d, best_cost, scan):
        total = self.total(self._global(ordered))
        best_order = [s.name for s in ordered]
        best_orient = [s.orientation for s in ordered]
        while True:
            stop = 0
            for i in range(0, len(ordered) - scan + 1):
                names = [s.name for s in ordered[i:i + scan]]
                orders = remove_reverse_duplicates(swap_permutations(names))
                orients = plus_minus_perms(len(names))
                for o in orders:
                    for r in orients:
                        beg = [s.name for s in ordered[0:i]]
                        win = self.reorder(o, r, d)
                        end = [s.name for s in ordered[i + scan:]]
                        new_order = beg + [s.name for s in win] + end
                        lst = [d[nm] for nm in new_order]
                        new_orient = [s.orientation for s in lst]
                        c = self.cost(self._global(lst), total)
                        if c > best_cost:
                            best_order, best_orient, best_cost, stop = new_order, new_orient, c, 1
                ordered = self.reorder(best_order, best_orient, d)
            if stop == 0:
                break
        return ordered, best_cost

    # --- OG:551-586 ------------------------------------------------------------------------
    def order_chromosome(self, group, n_scaffolds=6, scan_scaffolds=5):
        if n_scaffolds >= 9:
            n_scaffolds = 8
        if scan_scaffolds > n_scaffolds:
            scan_scaffolds = n_scaffolds
        rest, d = self.initiate(group)
        ordered = rest[:n_scaffolds]
        rest = rest[n_scaffolds:]
        o, r, _ = self.brute_force(ordered, d)
        ordered = self.reorder(o, r, d)
        best = None
        # OG:475-493 is a do-while: it always pulls one scaffold, even from an empty list
        while True:
            if rest:
                new = rest.pop(0)
                ordered, best = self.check_all_scores(ordered, new)
            else:
                new = ordered.pop(-1)
                ordered, best = self.check_all_scores(ordered, new)
            if len(rest) == 0:
                break
        if len(ordered) > n_scaffolds:
            ordered, best = self.scan_ordering(ordered, d, best, scan_scaffolds)
        return ordered, best


def read_groupings_to_valid_bins(path):
    """OG:200-214."""
    ids = {}
    with open(path) as fh:
        for line in fh:
            line = line.strip("\r").strip("\n")
            if line[0] != "#":
                ids[int(line.split("\t")[0])] = ''
    return ids


def read_chroms(path):
    """OG:216-237."""
    out, chrom = [], []
    with open(path) as fh:
        fh.readline()
        for line in fh:
            line = line.strip("\r").strip("\n")
            if line[0] != "#":
                chrom.append([int(line.split("\t")[0]), line.split("\t")[1]])
            else:
                out.append(chrom)
                chrom = []
    out.append(chrom)
    return out


def write_scaffold_orderings(orderings, path):
    """OG:630-644."""
    with open(path, "w") as fh:
        for k, group in enumerate(orderings):
            fh.write("### Chromosome grouping " + str(k + 1) + " ###\n")
            for s in group:
                fh.write(s.name + "\t" + s.orientation + "\n")


def write_bin_id_ordering(scaffolds, path):
    """OG:646-660: header then newline-PREFIXED rows, no trailing newline."""
    with open(path, "w") as fh:
        fh.write("#ScaffoldID\tHiCPro-BinID")
        for s in scaffolds:
            for b in s.binList:
                fh.write("\n" + s.name + "\t" + str(b))


def run_part2(bed, bias, matrix, chrom_group_file, chrom_order_file, plot_order_file,
              n_scaffolds=6, scan_scaffolds=5, trace=None, preloaded=None):
    """OG:679-712 (plots omitted).  ``preloaded=(matrix, bins)`` skips the text loaders; bins that
    are not in the group file are never selected, as in the reference's restricted re-load."""
    ids = read_groupings_to_valid_bins(chrom_group_file)
    if preloaded is not None:
        mat, bins = np.ascontiguousarray(preloaded[0], dtype=np.float64), list(preloaded[1])
    else:
        bins = initiate_loci(bed, bias, bin_ids=ids)
        mat = build_adjacency(matrix, bins)
    chroms = read_chroms(chrom_group_file)
    orc = Part2Oracle(mat, bins)
    out, marks, bests = [], [], []
    for group in chroms:
        marks.append(len(orc.costs))
        ordered, best = orc.order_chromosome(group, n_scaffolds, scan_scaffolds)
        out.append(ordered)
        bests.append(best)
    write_scaffold_orderings(out, chrom_order_file)
    write_bin_id_ordering([s for g in out for s in g], plot_order_file)
    if trace is not None:
        trace.update(costs=np.array(orc.costs), cost_marks=np.array(marks), best_costs=bests,
                     chrom_orders=[[(s.name, s.orientation) for s in g] for g in out])
    return out
