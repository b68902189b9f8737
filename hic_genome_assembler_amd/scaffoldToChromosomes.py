"""Part 1 (bins -> chromosome groups) on MI355X: drop-in for the reference module of the same name.

Same ``runPipeline`` signature and intermediate files as /root/reference/HIC_ASSEMBLER/
scaffoldToChromosomes.py (S2C below), and the same stage-function names the reference's notebook
calls, but the N x N matrix lives in HBM behind a libhicmi context (``DeviceMatrix``) and every
heavy step is a hand-written gfx950 kernel reached through the C ABI of include/hicmi.h:

  S2C stage                                   here
  ------------------------------------------  ------------------------------------------------
  removeRows / row sums        S2C:100-136    hicmi_row_sums, hicmi_compact
  convertMatrix(distance)      S2C:138-155    fused into hicmi_upgma (k_build_w)
  squareform+average+dendrogram S2C:187-208   hicmi_upgma (k_nnchain + host label / leaf walk)
  reorderMatrix, similarity    S2C:157-163,149 fused into hicmi_rank_matrix (k_sort_rows)
  numpy.argsort[:, ::-1]       S2C:1132       hicmi_rank_matrix
  find_matrix_pvalue_breakpoints S2C:413-511  hicmi_cut_scan (+ host window logic)
  filter_noisy_breakpoints     S2C:553-727    hicmi_filter_scan (+ host control flow)

Host control flow (loops over cut candidates, file formats, scaffold voting) is restated here in
Python because its decisions are sequential and tiny.  There is no CPU fallback for the kernels.

Not implemented: the HMM boundary finder.  The Louvain tail (``modularity > 0``,
unseeded-random in the reference) is a seeded restatement in modularity.py.  The two Part 1 figures are drawn from the
device-resident matrix by plotContactMaps.py (exact percentiles, figure-resolution block means).
"""
from __future__ import annotations

import collections
import operator
import os
import sys
import time

import numpy as np

from . import _lib
from . import dist as _dist
from . import modularity as louvain
from . import plotContactMaps as plotModule
from .hostio import Bin, initiateLoci, paused_gc, read_contact_matrix, read_contact_matrix_cached  # noqa: F401  (re-exported reference names)


# ------------------------------------------------------------------------------------------------
class DeviceMatrix:
    """The contact map resident on one GPU, plus the host-side bookkeeping the reference keeps in
    its ``numpy.matrix`` / ``binList`` pair."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self.kind = "contacts"         # contacts -> distance -> similarity (labels only; transforms are fused)
        self.order = None              # current row/column order relative to the uploaded matrix
        self.np_sum = None
        self.seq_sum = None

    @property
    def n(self):
        return self.ctx.n

    def __len__(self):
        return self.ctx.n


class RankMatrix:
    """Device-resident result of ``argsort(axis=1)[:, ::-1]`` (S2C:1132) and its inverse."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx

    def __len__(self):
        return self.ctx.n

    def rows(self, row0=0, nrows=None):
        return self.ctx.rank_rows(row0, nrows, inverse=False)


def buildAdjacencyMatrix(matrixFile, binList, binID_dict=False, device=0, ctx=None):
    """S2C:70-98: dense matrix from HiC-Pro triplets, uploaded once to HBM."""
    cache = os.environ.get("HICMI_MATRIX_CACHE")           # "1": beside the text file; or a directory (hostio.py)
    host = read_contact_matrix_cached(matrixFile, binList, cache) if cache else read_contact_matrix(matrixFile, binList)
    ctx = ctx or _lib.Context(device)
    ctx.set_contacts(host)
    print("Rows in adjacency matrix " + str(len(binList)))
    return DeviceMatrix(ctx)


def removeRows(matrix: DeviceMatrix, binList, zeroRows=True, biasVals=False, store_row_sums=True):
    """S2C:100-136: drop rows (and columns) whose NumPy row sum is 0, then store each bin's
    left-to-right row sum.  ``biasVals`` filtering is kept for signature compatibility.
    ``store_row_sums=False``: the caller writes ``Bin.rowSum`` itself, later (``matrix.seq_sum`` holds the values;
    runResident does it beside the chain)."""
    np_sum, seq_sum = matrix.ctx.row_sums()
    first, world = getattr(matrix.ctx, "shard", (0, 1))
    if world > 1:                                          # one map over several GPUs: every rank summed its own rows
        np_sum = _dist.gather_owned(np_sum, 0, first, world)
        seq_sum = _dist.gather_owned(seq_sum, 0, first, world)
        matrix.ctx.set_row_sums(np_sum, seq_sum)
    dropped = np.zeros(len(binList), dtype=bool)
    if zeroRows is True:
        dropped |= np.asarray(np_sum) == 0
    if biasVals is not False:
        bias = np.fromiter((b.bias for b in binList), dtype=np.float64, count=len(binList))
        dropped |= ~dropped & ((bias > biasVals[1]) | (bias < biasVals[0]))
    print("Rows/columns to remove " + str(int(dropped.sum())))
    if dropped.any():
        keep = np.flatnonzero(~dropped).tolist()
        matrix.ctx.compact(keep)
        binList = [binList[i] for i in keep]
        np_sum, seq_sum = matrix.ctx.row_sums()
    matrix.np_sum, matrix.seq_sum = np_sum, seq_sum
    if store_row_sums:
        _store_row_sums(binList, seq_sum)
    return matrix, binList


def _store_row_sums(binList, seq_sum):
    for b, v in zip(binList, np.asarray(seq_sum, dtype=np.float64).tolist()):
        b.rowSum = v


def convertMatrix(adjacencyMatrix: DeviceMatrix, binList, distance=True, similarity=False):
    """S2C:138-155.  The transforms are elementwise and are fused into the kernels that consume
    them (k_build_w for distance, k_sort_rows for similarity), so this only records the stage."""
    if distance is True:
        adjacencyMatrix.kind = "distance"
    elif similarity is True:
        adjacencyMatrix.kind = "similarity"
    return adjacencyMatrix


def averageClusterNodes(adjacencyMatrix: DeviceMatrix, nodeLabels, noPlot=True, meanwhile=None, want_ivl=True):
    """S2C:187-208: UPGMA + count-sorted leaf order.  Returns a dict with the two keys of SciPy's
    dendrogram object the reference uses ('ivl', 'leaves') plus the linkage matrix 'Z'.
    ``meanwhile``: host work that does not need the tree; it runs on this thread while the chain - one native call of
    ~100 ms that releases the interpreter lock - runs on another (``nodeLabels`` may then be a callable returning the
    labels: it is called after ``meanwhile``)."""
    t0 = time.time()
    if meanwhile is None:
        leaves, z = adjacencyMatrix.ctx.upgma(want_linkage=True)
    else:
        import threading
        box = {}

        def chain():
            try:
                box["out"] = adjacencyMatrix.ctx.upgma(want_linkage=True)
            except BaseException as exc:          # re-raised on the calling thread
                box["err"] = exc
        th = threading.Thread(target=chain)
        th.start()
        try:
            meanwhile()
        finally:
            th.join()
        if "err" in box:
            raise box["err"]
        leaves, z = box["out"]
    if callable(nodeLabels):
        nodeLabels = nodeLabels()
    print("Time to cluster " + str(time.time() - t0))
    leaves = np.asarray(leaves).tolist()
    return {"ivl": [nodeLabels[i] for i in leaves] if want_ivl else None, "leaves": leaves, "Z": z}


def dendrogramLeafOrder_toFile(dendrogramObj, outFile, lines=None):
    """S2C:210-220: ``label<TAB>leaf`` lines, no trailing newline.  ``lines``: the line of every leaf, indexed by leaf
    (formatted beside the chain: runResident)."""
    with open(outFile, "w") as fh:
        if lines is not None:
            fh.write("\n".join(map(lines.__getitem__, dendrogramObj["leaves"])))
        else:
            fh.write("\n".join(l + "\t" + str(i) for l, i in zip(dendrogramObj["ivl"], dendrogramObj["leaves"])))


def readDengrogramLeavesFromFile(dendrogramFile):
    """S2C:222-234."""
    out = {"ivl": [], "leaves": []}
    with open(dendrogramFile) as fh:
        for line in fh:
            cols = line.strip("\r").strip("\n").split("\t")
            out["ivl"].append(cols[0])
            out["leaves"].append(int(cols[-1]))
    return out


def reorderMatrix(matrix: DeviceMatrix, binList, newOrder):
    """S2C:157-163: the permutation is applied by the consuming kernel through an index vector."""
    newOrder = np.asarray(newOrder, dtype=np.int64).tolist()
    matrix.order = newOrder if matrix.order is None else [matrix.order[i] for i in newOrder]
    return matrix, [binList[i] for i in newOrder]


def rankOrderMatrix(matrix: DeviceMatrix) -> RankMatrix:
    """S2C:1132 ``numpy.asarray(numpy.argsort(adjMat, axis=1)[:, ::-1])`` on the similarity matrix
    in the current order.  Ties (undefined in the reference) resolve to the larger column first."""
    if matrix.kind != "similarity":
        raise ValueError("rankOrderMatrix expects the similarity stage (call convertMatrix(similarity=True))")
    order = matrix.order if matrix.order is not None else list(range(matrix.n))
    matrix.ctx.rank_matrix(order)
    return RankMatrix(matrix.ctx)


# ------------------------------------------------------------------------------------------------
def hyper_geom(x, M, n, N):
    """S2C:352-368 (= scipy.stats.hypergeom.sf(x-1, M, n, N)), evaluated by libhicmi."""
    return _lib.hypergeom_sf(x, M, n, N)


def _window_scores(input_array, window_size):
    """break_sigs of S2C:390-400 as an int64 array (None for the ["NA","NA","NA"] case, S2C:380-381)."""
    if window_size >= len(input_array):
        return None
    a = np.asarray(input_array, dtype=np.int64)
    h = int(window_size)
    cs = np.concatenate(([0], np.cumsum(a)))
    scores = np.zeros(len(a) - h, dtype=np.int64)
    full = max(0, len(a) - 2 * h + 1)                 # windows whose right half is complete; the rest score 0
    if full:
        scores[:full] = (cs[h:h + full] - cs[:full]) - (cs[2 * h:2 * h + full] - cs[h:h + full])
    return scores


def get_sliding_window_distance_metrics(input_array, window_size, global_index=0):
    """S2C:370-411: left-half minus right-half window sums; 0 where the right half is short."""
    scores = _window_scores(input_array, window_size)
    if scores is None:
        return ["NA", "NA", "NA"]
    best = int(np.flatnonzero(scores == scores.max())[0])
    return [int(v) for v in scores], int(window_size), best


def _device_scan_loops(ctx):
    """The scan loops run with their control flow on the device when the context has the whole rank matrix (one GPU)
    and offers the entry points; HICMI_HOST_SCANS=1 keeps the per-scan host loops (the A/B, and what a row shard uses)."""
    if os.environ.get("HICMI_HOST_SCANS"):
        return False
    _first, world = getattr(ctx, "shard", (0, 1))
    return world == 1 and hasattr(ctx, "first_pass_cuts") and hasattr(ctx, "filter_cuts")


def find_matrix_pvalue_breakpoints(argsorted_mat: RankMatrix, start, min_size, world_size, psig=.05):
    """S2C:413-511 for one ``start``.  Counts and significance flags come from the GPU
    (hicmi_cut_scan); the >= 90 % rule that shrinks M (S2C:473-483) and the window scan run here.
    The window-shrinking retry of S2C:499-508 cannot produce a cut (its scores are < min_size while
    S2C:488 tests == min_size), so it is not repeated."""
    ctx = argsorted_mat.ctx
    first, world = getattr(ctx, "shard", (0, 1))
    M = world_size
    loop_count = 0
    while True:
        sig = ctx.cut_scan(int(start), int(M), float(psig))
        if world > 1:                                      # flags of this rank's rows -> all rows (one all-gather per scan)
            sig = _dist.gather_owned(sig, int(start), first, world)
        loop_count += 1
        if (int(sig.sum()) / len(sig)) >= .9:
            morg = M
            M = int(M - start)
            print("- M value (world_size) changed to dynamic {} --> {}".format(morg, M))
        else:
            break
        if loop_count >= 5:
            break
    scores = _window_scores(sig, min_size)
    if scores is None:
        return [], []
    inds = [int(v) for v in np.flatnonzero(scores == min_size) + min_size]         # S2C:488
    return [min_size] * len(inds), inds


def pre_process_all_matrix_breakpoints(argsorted_mat: RankMatrix, min_size=5, min_frac=.05, psig=.05):
    """S2C:513-551.  As in the reference the scan tests against the literal .05 (S2C:535), not
    the ``psig`` argument."""
    mat_size = len(argsorted_mat)
    stop_ind = int(mat_size - (mat_size * min_frac))
    ind = 0
    cinds = []
    if min_frac == 1:
        return cinds
    if _device_scan_loops(argsorted_mat.ctx) and min_size >= 1:
        # the same loop with its decisions on the device (hicmi_first_pass_cuts): one host round trip per batch of scans
        cinds, m_changes = argsorted_mat.ctx.first_pass_cuts(min_size, stop_ind, .05)
        for morg, m in m_changes:
            print("- M value (world_size) changed to dynamic {} --> {}".format(morg, m))
        print("- Breakpoints found {}".format(len(cinds)))
        return cinds
    while True:
        _vals, pre_cut_inds = find_matrix_pvalue_breakpoints(argsorted_mat, ind, min_size, mat_size - ind, psig=.05)
        if len(pre_cut_inds) == 0:
            break
        ind += pre_cut_inds[0]
        cinds.append(ind)
        if (ind >= stop_ind) or ((mat_size - ind) <= min_size):
            break
    print("- Breakpoints found {}".format(len(cinds)))
    return cinds


def filter_noisy_breakpoints(argsorted_mat: RankMatrix, original_inds, psig=.05):
    """S2C:553-727.  Row tests run on the GPU (hicmi_filter_scan); the merge decisions between
    candidate cuts are scalar hypergeometric tests evaluated by libhicmi's host code."""
    if len(original_inds) == 0:
        return []
    ctx = argsorted_mat.ctx
    first, world = getattr(ctx, "shard", (0, 1))
    n = len(argsorted_mat)
    ascending = all(b > a for a, b in zip(original_inds, original_inds[1:]))
    if _device_scan_loops(ctx) and ascending and 0 <= original_inds[0] and original_inds[-1] < n:
        out, warned = ctx.filter_cuts(original_inds, psig)          # the loops below, on the device (hicmi_filter_cuts)
        for _ in range(warned):
            print("- WARNING - Maximum number of rounds {} exceeded".format(10 * len(original_inds)))
        print("- Original cut indices {}".format(list(original_inds)))
        print("- Filtered cut indices {}".format(out))
        return out
    MD = int(n / 5)
    MAX_ROUNDS = 10 * len(original_inds)
    altered = [int(v) for v in original_inds]
    prev_filtered = {}
    while True:
        start = 0
        filtered = {}
        round_count = 0
        while True:
            if round_count >= MAX_ROUNDS:
                print("- WARNING - Maximum number of rounds {} exceeded".format(MAX_ROUNDS))
                break
            M = n - start
            noise_found = 0
            keep_from = 0
            for i, c in enumerate(altered):
                local = c - start
                n_rows = min(n - start, MD + 1)            # rows beyond start+MD are forced to 0 (S2C:626-628)
                flags = np.zeros(n - start, dtype=np.int64)
                own = ctx.filter_scan(start, c, n_rows, M, float(psig))
                flags[:n_rows] = _dist.gather_owned(own, start, first, world) if world > 1 else own
                sig_cuts = []
                fc_prev = start
                for ai_ind, ai in enumerate(altered):
                    if ai == fc_prev:
                        continue
                    lo, hi = fc_prev, ai
                    fc_prev = ai
                    if hi <= lo:
                        break
                    x = int(flags[lo - start:hi - start].sum())
                    noise_pval = hyper_geom(x, M, local, hi - lo)
                    if noise_pval < psig:
                        sig_cuts.append((ai_ind, ai))
                if sig_cuts:
                    keep_from, start = sig_cuts[-1]
                    filtered[start] = ''
                    noise_found = 1
                    break
                filtered[c] = ''
                keep_from = i
            round_count += 1
            if noise_found == 0:
                break
            altered = altered[keep_from:]
        if prev_filtered != filtered:
            altered = sorted(filtered)
            prev_filtered = filtered
        else:
            break
    out = sorted(filtered.keys())
    print("- Original cut indices {}".format(list(original_inds)))
    print("- Filtered cut indices {}".format(out))
    return out


# ------------------------------------------------------------------------------------------------
def _bin_line(b):
    return f"{b.ID}\t{b.chrom}\t{b.start}\t{b.stop}\t{b.bias}"


def writeBinGroupingsToFile(coords, binList, outFile, bin_lines=None):
    """S2C:945-964.  Returns the groups as lists of the lines written (what
    readBinGroupingsFromFile gives back for this file).  ``bin_lines``: {bin ID: its line} when the caller has formatted
    them already (runResident does, beside the chain)."""
    bounds = [0] + [int(c) for c in coords] + [len(binList)]
    groups, text = [], []
    for g in range(len(bounds) - 1):
        if bin_lines is not None:
            lines = [bin_lines[b.ID] for b in binList[bounds[g]:bounds[g + 1]]]
        else:
            lines = [_bin_line(b) for b in binList[bounds[g]:bounds[g + 1]]]
        text.append("### Chromosome group " + str(g + 1) + " ###\n")
        if lines:
            text.append("\n".join(lines) + "\n")
        groups.append(lines)
    with open(outFile, "w") as fh:
        fh.write("".join(text))
    return groups


def readSizeFileToDict(sizeFile):
    """S2C:968-979."""
    sizes = {}
    with open(sizeFile) as fh:
        for line in fh:
            cols = line.strip("\r").strip("\n").split("\t")
            sizes[cols[0]] = int(cols[1])
    return sizes


def readBinGroupingsFromFile(binGroupingsFile):
    """S2C:981-999: first line skipped unconditionally, every later '#' line starts a new group."""
    groups, cur = [], []
    with open(binGroupingsFile) as fh:
        fh.readline()
        for line in fh:
            line = line.strip("\n").strip("\r")
            if line[0] != "#":
                cur.append(line)
            else:
                groups.append(cur)
                cur = []
    groups.append(cur)
    print(str(len(groups)) + " chromosomes read in from file")
    return groups


def _assess_rows(pairs, scaffDict, percentToAssign):
    """The decisions of assessClusterList on (bin, scaffold) pairs: (rows of the report - scaffold, bins here, bins in
    all, percentage -, the bins of the scaffolds assigned, their names, bins of the others)."""
    members = collections.Counter(map(_second, pairs))       # (a dict: scaffolds in order of first appearance)
    rows, final, names, false_pos = [], [], [], 0
    for s, have in members.items():
        total = len(scaffDict[s])
        pct = round(((float(have) / float(total)) * 100.), 2)
        rows.append((s, have, total, pct))
        if pct >= percentToAssign:
            final += scaffDict[s]
            names.append(s)
        else:
            false_pos += have
    return rows, final, names, false_pos


def _assess_text(rows, assigned, out):
    out.append("#Scaffold\tNodesAssigend\tTotalNodes\tAssigned%\n")
    for s, have, total, pct in rows:
        out.append(str(s) + "\t" + str(have) + "\t" + str(total) + "\t" + str(pct) + "%\n")
    out.append("Total scaffolds clustered to chromosome " + str(len(rows)) + "\n")
    out.append("Total scaffolds assigned to chromosome " + str(assigned) + "\n")


def _assess_group(pairs, scaffDict, out, percentToAssign):
    """assessClusterList on (bin, scaffold) pairs; ``out`` collects the report lines."""
    rows, final, names, false_pos = _assess_rows(pairs, scaffDict, percentToAssign)
    _assess_text(rows, len(names), out)
    return final, false_pos, len(names)


_second = operator.itemgetter(1)


class PairList(list):
    """A group that is known to hold (bin, scaffold) pairs only (_bin_group_pairs): _pairs_of_lines passes it through."""


def _pairs_of_lines(cList):
    """(bin, scaffold) of every bin-grouping line; entries that already are such pairs pass through."""
    if isinstance(cList, PairList):
        return cList
    return [line if isinstance(line, tuple) else tuple(line.split("\t", 2)[:2]) for line in cList]


def _bin_pairs(binList):
    """{bin ID: (bin ID text, scaffold)} - independent of the grouping, so runResident builds it beside the chain."""
    return {b.ID: (str(b.ID), b.chrom) for b in binList}


def _bin_group_pairs(coords, binList, pairs=None):
    """What readBinGroupingsFromFile + the split in assessChromosomeClustering extract from the bin-grouping file:
    per group the (bin ID text, scaffold) pairs - taken from the bins themselves (``pairs``: _bin_pairs of them)."""
    bounds = [0] + [int(c) for c in coords] + [len(binList)]
    if pairs is None:
        return [PairList((str(b.ID), b.chrom) for b in binList[bounds[g]:bounds[g + 1]]) for g in range(len(bounds) - 1)]
    ids = [b.ID for b in binList]
    return [PairList(map(pairs.__getitem__, ids[bounds[g]:bounds[g + 1]])) for g in range(len(bounds) - 1)]


def assessClusterList(cList, scaffDict, outFile, percentToAssign=51.):
    """S2C:1001-1036: a scaffold joins the group holding >= 51 % of its bins and brings ALL its bins."""
    out = []
    res = _assess_group(_pairs_of_lines(cList), scaffDict, out, percentToAssign)
    outFile.write("".join(out))
    return res


def _write_text(path, text):
    with open(path, "w") as fh:
        fh.write(text)


def _scaffold_bins(pairs):
    """{scaffold: [[bin, scaffold], ...] with the bins ascending} over (bin, scaffold) pairs (S2C:1043-1053)."""
    scaffolds = {}
    for bin_id, scaff in pairs:
        scaffolds.setdefault(scaff, []).append(int(bin_id))
    return {s: [[b, s] for b in sorted(ids)] for s, ids in scaffolds.items()}


def assessChromosomeClustering(chromList, statsFile, percentToAssign=51., write=None, scaffolds=None):
    """S2C:1038-1077.  ``chromList``: groups of bin-grouping lines (``ID<TAB>scaffold<TAB>...``).  ``write``: called as
    ``write(fn, *args)`` to put the report on disk (default: at once).  ``scaffolds``: the scaffold -> bins table when
    the caller has built it already (it does not depend on the grouping: every bin is in exactly one group)."""
    groups = [_pairs_of_lines(grp) for grp in chromList]
    if scaffolds is None:
        scaffolds = _scaffold_bins(p for grp in groups for p in grp)
    final, false_pos, assigned, per_group = GroupList(), 0, 0, []
    for grp in groups:
        rows, nodes, names, fp = _assess_rows(grp, scaffolds, percentToAssign)
        per_group.append((rows, len(names)))
        if len(nodes) > 0:
            final.append(nodes)
            final.scaffolds.append(names)
        false_pos += fp
        assigned += len(names)
    total = sum(len(grp) for grp in groups)

    def report():
        """The text of the file (formatted where it is written: on the writer thread when ``write`` defers it)."""
        out = []
        for k, (rows, n_assigned) in enumerate(per_group):
            out.append("### Chromosome" + str(k + 1) + " ###\n")
            _assess_text(rows, n_assigned, out)
            out.append("####################\n")
        out.append("Total Nodes " + str(total) + "\n")
        out.append("Properly clustered nodes " + str(total - false_pos) + "\n")
        out.append("Falsely clustered nodes " + str(false_pos) + "\n")
        out.append("Total scaffolds assigned to chromosomes " + str(assigned) + "\n")
        out.append("Error rate ~" + str(round((float(false_pos) / float(total)) * 100., 2)) + "%\n")
        _write_text(statsFile, "".join(out))
    if write is None:
        report()
    else:
        write(report)
    return final


class GroupList(list):
    """assessChromosomeClustering's result: the groups' [bin, scaffold] entries, and beside them (``scaffolds``) the names of
    the scaffolds of every group - what rankChromosomeGroups would otherwise collect from all the entries again."""

    def __init__(self, *a):
        super().__init__(*a)
        self.scaffolds = []


def rankChromosomeGroups(chromList, scaffSizeDict):
    """The order in which S2C:1079-1100 writes the groups: by total scaffold bp, largest first (stable).  What
    orderGenome.readChromsFromFile gives back for the file written from it."""
    names = getattr(chromList, "scaffolds", None)
    if names is not None and len(names) == len(chromList):
        sizes = [sum(scaffSizeDict[s] for s in grp_names) for grp_names in names]
    else:
        sizes = [sum(scaffSizeDict[s] for s in {e[1]: '' for e in grp}) for grp in chromList]
    ranked = sorted(range(len(chromList)), key=lambda k: sizes[k], reverse=True)
    return [chromList[k] for k in ranked]


def writeChromosomeGroupingsToFile(chromList, scaffSizeDict, outFile, entry_lines=None):
    """S2C:1079-1100: groups ordered by total scaffold bp, largest first (stable).  ``entry_lines``: {bin ID: its
    ``bin<TAB>scaffold`` line with the newline} when formatted already."""
    text = []
    for new_id, grp in enumerate(rankChromosomeGroups(chromList, scaffSizeDict)):
        text.append("### Chromosome group " + str(new_id + 1) + " ###\n")
        if entry_lines is not None:
            text.extend([entry_lines[e[0]] for e in grp])
        else:
            text.extend(str(e[0]) + "\t" + str(e[1]) + "\n" for e in grp)
    with open(outFile, "w") as fh:
        fh.write("".join(text))


# ------------------------------------------------------------------------------------------------
def runPipeline(hicProBedFile, hicProBiasFile, hicProMatrixFile, hicProScaffSizeFile,
                dendrogramOrderFile, avgClusterPlot, avgClusterPlot_outlined,
                binGroupFile, assessmentFile, chromosomeGroupFile,
                hyperGeom, hmm, minSize, modularity, louvainRounds,
                psig, convergenceRounds, lookAhead, resolution, device=0, shard=None, keep_resident=False):
    """S2C:1104-1174, same positional arguments (``device``, ``shard`` and ``keep_resident`` are optional extras;
    ``shard=(rank, world)``: this process is one of ``world`` that work on the same map, see runResident;
    ``keep_resident=True``: the context with the contact matrix in HBM is not closed but returned as
    ``(DeviceMatrix, bins of its rows)`` so that Part 2 of the same run need not parse the text matrix again)."""
    print("########################################")
    print("### Working on Part1 of the pipeline ###")
    t_all = time.time()
    if hyperGeom is not True:
        raise NotImplementedError("only the hyperGeom = True strategy is implemented on MI355X "
                                  "(hmm needs hmmlearn's stochastic EM; SURVEY.md section 2 row 7)")
    binList = initiateLoci(hicProBedFile, hicProBiasFile)
    adjMat = buildAdjacencyMatrix(hicProMatrixFile, binList, device=device)
    try:
        cutIndices = runResident(adjMat, binList, hicProScaffSizeFile, dendrogramOrderFile, binGroupFile,
                                 assessmentFile, chromosomeGroupFile, minSize, modularity, psig,
                                 louvainRounds=louvainRounds, shard=shard)
        # S2C:1124 / S2C:1155-1156: the clustered distance matrix, then - groups outlined - the distance transform of
        # the un-logged similarity matrix, which is the distance matrix again up to a few roundings
        if plotModule.plots_enabled(avgClusterPlot):
            plotModule.plotContactMap(plotModule.DeviceImage(adjMat.ctx, 1, adjMat.order), resolution=resolution,
                                      highlightChroms=False, showPlot=False, savePlot=avgClusterPlot)
        if plotModule.plots_enabled(avgClusterPlot_outlined):
            plotModule.plotContactMap(plotModule.DeviceImage(adjMat.ctx, 1, adjMat.order), resolution=resolution,
                                      highlightChroms=cutIndices, showPlot=False, savePlot=avgClusterPlot_outlined)
    except BaseException:
        adjMat.ctx.close()
        raise
    if not keep_resident:
        adjMat.ctx.close()
    print("Total run-time of Part1 = " + str(time.time() - t_all))
    print("CutIndices = " + str(cutIndices))
    print("- Part 1 (grouping bins to groups) completed successfully")
    return (adjMat, adjMat.kept_bins) if keep_resident else None


class _FileWriter:
    """Part 1's text files written on one background thread while the device keeps working: the strings are built and
    written in submission order; ``finish()`` returns when every file is on disk and re-raises what a writer raised.
    Jobs submitted with ``deferred=True`` wait for ``release()``: formatting 16,000 lines holds the interpreter lock, so
    the caller releases them when it enters a long native call (Part 2's lock-step insertion), not while its own
    Python threads are busy."""

    def __init__(self, enabled):
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max_workers=1) if enabled else None
        self.gate = threading.Event()
        self.jobs = []

    def submit(self, fn, *args, deferred=False):
        if self.pool is None:
            return fn(*args)
        if deferred:
            def held(*a):
                self.gate.wait()
                return fn(*a)
            self.jobs.append(self.pool.submit(held, *args))
        else:
            self.jobs.append(self.pool.submit(fn, *args))

    def release(self):
        self.gate.set()

    def finish(self):
        self.gate.set()
        for j in self.jobs:
            j.result()
        self.jobs = []
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None


def runResident(adjMat: DeviceMatrix, binList, hicProScaffSizeFile, dendrogramOrderFile, binGroupFile,
                assessmentFile, chromosomeGroupFile, minSize, modularity, psig, louvainRounds=20, shard=None,
                overlap_files=False):
    """S2C:1117-1167 on a contact map that is already resident in HBM (what bench.py times): every
    stage after the text loaders, including the small intermediate files the reference round-trips
    through.  Returns the filtered cut indices; ``binList`` is left in .bed order for the caller.

    ``overlap_files=True``: the four files are written by a background thread (the dendrogram order while the rows
    are sorted and scanned, the three group files while the caller goes on); the caller hands
    ``adjMat.chromosome_groups`` - what Part 2 would read back from chromosomeGroupFile - to
    ``orderGenome.runResident(..., chromosomeList=, on_native_phase=adjMat.release_files)`` and calls
    ``adjMat.finish_files()`` before it uses the files."""
    writer = _FileWriter(overlap_files)
    adjMat.finish_files = writer.finish
    adjMat.release_files = writer.release
    marks = [("start", time.perf_counter())] if os.environ.get("HICMI_STEP_PROFILE") else None

    def mark(name):
        if marks is not None:
            marks.append((name, time.perf_counter()))
    with paused_gc():
        t0 = time.time()
        if shard is not None:
            # ONE map over shard[1] ranks: the row sums, the row sort / rank matrix and the per-row counts of
            # the scans are computed for rows == shard[0] (mod shard[1]) only and all-gathered; UPGMA and the host control
            # flow run on every rank (deterministic: all ranks write the same files)
            adjMat.ctx.set_row_shard(shard[0], shard[1])
        adjMat, binList = removeRows(adjMat, binList, zeroRows=True, biasVals=False, store_row_sums=False)
        adjMat.kept_bins = list(binList)                  # rows of the device matrix, in .bed order
        adjMat = convertMatrix(adjMat, binList, distance=True, similarity=False)
        mark("row sums")
        # host work that does not depend on the tree runs while the chain does (a single ~100 ms native call): the labels,
        # the size table and the scaffold -> bins table of the assessment
        prep = {}

        def meanwhile(bl=binList, sums=adjMat.seq_sum):
            _store_row_sums(bl, sums)                     # (Bin.rowSum, S2C:133-135: nothing reads it before the chain ends)
            prep["labels"] = [b.chrom + '_' + str(b.ID) for b in bl]
            prep["sizes"] = readSizeFileToDict(hicProScaffSizeFile)
            prep["scaffolds"] = _scaffold_bins((b.ID, b.chrom) for b in bl)
            prep["pairs"] = _bin_pairs(bl)
            # the text of the three per-bin files, line by line: ~12 ms of formatting at 16k that would otherwise hold the
            # interpreter lock while Part 2's threads want it (the writer thread only joins the lines)
            prep["dend_lines"] = [lab + "\t" + str(i) for i, lab in enumerate(prep["labels"])]
            prep["bin_lines"] = {b.ID: _bin_line(b) for b in bl}
            prep["entry_lines"] = {int(b.ID): str(int(b.ID)) + "\t" + str(b.chrom) + "\n" for b in bl}
        # ('ivl', the labels in leaf order, is only wanted by the dendrogram file, whose lines are ready: prep["dend_lines"])
        dendrogram = averageClusterNodes(adjMat, lambda: prep["labels"], noPlot=True, meanwhile=meanwhile, want_ivl=False)
        mark("UPGMA + leaf order")
        # the reference parses the file back (readDengrogramLeavesFromFile); the leaves are the same integers
        adjMat, binList = reorderMatrix(adjMat, binList, dendrogram['leaves'])
        print("Total run-time to cluster = " + str(time.time() - t0))
        t0 = time.time()
        adjMat = convertMatrix(adjMat, binList, distance=False, similarity=True)
        argsorted_adjMat = rankOrderMatrix(adjMat)
        # the dendrogram file (16k formatted lines: ~5 ms of interpreter time) is handed to the writer thread only now: the
        # two scan loops that follow are native calls that release the interpreter lock, the list work above is not
        writer.submit(dendrogramLeafOrder_toFile, dendrogram, dendrogramOrderFile, prep["dend_lines"])
        mark("reorder + rank matrix")
        initial_cut_inds = pre_process_all_matrix_breakpoints(argsorted_adjMat, min_size=minSize,
                                                              min_frac=modularity, psig=psig)
        mark("first-pass scans")
        cutIndices = filter_noisy_breakpoints(argsorted_adjMat, initial_cut_inds, psig=psig)
        mark("filter scans")
        if modularity is not False and modularity > 0.0:
            # S2C:1148-1152: Louvain on log10(similarity + 1) of the bins after the last cut index (modularity.py:
            # seeded restatement of python-louvain; the cells come from the device in the current order)
            start = sorted(cutIndices)[-1] if len(cutIndices) else 0
            tail_rows = list(adjMat.order[start:])
            if len(tail_rows) > 0:
                sim_tail = adjMat.ctx.plot_downsample(2, tail_rows, len(tail_rows))
                new_order, cutIndices = louvain.modularity_remaining_data(louvain.log_transform(sim_tail), binList,
                                                                          cutIndices, n_rounds=louvainRounds)
                adjMat, binList = reorderMatrix(adjMat, binList, new_order)
        writer.submit(writeBinGroupingsToFile, list(cutIndices), list(binList), binGroupFile, prep["bin_lines"], deferred=True)
        binGroups = _bin_group_pairs(cutIndices, binList, prep["pairs"])
        print("Total run-time to identify chromosome boundaries = " + str(time.time() - t0))
        t0 = time.time()
        fastaSizeDict = prep["sizes"]
        print(str(len(binGroups)) + " chromosomes read in from file")      # == readBinGroupingsFromFile(binGroupFile)
        chrGroups = assessChromosomeClustering(binGroups, assessmentFile,
                                               write=lambda fn, *a: writer.submit(fn, *a, deferred=True),
                                               scaffolds=None if (modularity is not False and modularity > 0.0) else prep["scaffolds"])
        adjMat.chromosome_groups = rankChromosomeGroups(chrGroups, fastaSizeDict)
        writer.submit(writeChromosomeGroupingsToFile, chrGroups, fastaSizeDict, chromosomeGroupFile, prep["entry_lines"],
                      deferred=True)
        print("Total run-time to assign scaffolds to chromosomes = " + str(time.time() - t0))
        mark("groups + assessment")
    if not overlap_files:
        writer.finish()
    if marks is not None:
        sys.stderr.write("[hicmi] part1 host timeline (ms): " + ", ".join(
            "%s %.1f" % (b[0], (b[1] - a[1]) * 1e3) for a, b in zip(marks, marks[1:])) + "\n")
    return cutIndices
