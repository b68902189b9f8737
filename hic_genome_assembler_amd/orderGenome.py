"""Part 2 (order + orient scaffolds inside each chromosome) on MI355X: drop-in for the reference
module of the same name (/root/reference/HIC_ASSEMBLER/orderGenome.py, OG below).

Same ``runPipeline`` signature, same input/output files and the same search - brute force over the
largest scaffolds, greedy insertion of the rest, sliding-window re-permutation to a fixed point -
with the reference's enumeration order and first-strict-maximum tie-breaking.  What changes is where
candidates live and how they are scored:

* a chromosome's contacts are selected ONCE into a device sub-matrix; each scaffold is a contiguous
  range of it (the layout) and an order/orientation is a list of (scaffold, reversed) pairs (the
  arrangement) - the reference instead rebuilds a gathered matrix and an index dictionary for every
  step (giveNewAdjMat, OG:296-308) and a Python index list per candidate;
* the candidates of a step are enumerated by the kernels themselves (k_part2_search.hip):
  2(S+1) insertions per launch, k!/2 * 2^k window candidates per launch with the incremental form
  of the objective;
* the few candidates that can win a step are re-scored in the reference's exact operation order
  (k_p2_diag_sums), because `cost > bestCost` is decided at the last bit (see first_strict_max).
"""
from __future__ import annotations

import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from . import plotContactMaps as plotModule
from .hostio import Bin, initiateLoci, paused_gc, read_contact_matrix, read_contact_matrix_cached  # noqa: F401

SCORE_HOOK = None      # tests: called with the fast scores of every step, in enumeration order
_PROFILE = bool(os.environ.get("HICMI_PART2_PROFILE"))   # per-chromosome wall clock on stderr
WORKERS = int(os.environ.get("HICMI_PART2_WORKERS", "8"))
LOCKSTEP = os.environ.get("HICMI_PART2_LOCKSTEP", "1") != "0"   # all chromosomes' insertion loops in one queue of launches   # chromosomes ordered concurrently (1 = sequential)
START_THREADS = int(os.environ.get("HICMI_PART2_START_THREADS", "0"))   # A/B: the start phase on this many threads (0: the calling thread)
NEAR_TOP = 1e-9        # relative band around a step's best fast score that is re-scored literally


# ------------------------------------------------------------------------------------------------
class GenomeMatrix:
    """Raw contacts of the grouped bins, resident on the GPU (OG:690)."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx
        self.chrom = None              # ChromosomeLayout currently selected on the device
        self._bin_index = None
        self._bin_index_src = None

    def __len__(self):
        return self.ctx.n

    def bin_index(self, binList):
        if self._bin_index is None or self._bin_index_src is not binList:
            self._bin_index = {b.ID: i for i, b in enumerate(binList)}
            self._bin_index_src = binList
        return self._bin_index


class ChromosomeLayout:
    """All scaffolds of one chromosome selected on the device: scaffold s <-> a contiguous range of
    the selection holding its bins in ascending-ID ('+') order."""

    def __init__(self, matrix: GenomeMatrix, scaffolds, binList):
        self.ctx = matrix.ctx
        where = matrix.bin_index(binList)
        self.sid, self.start, self.length, self.names = {}, [], [], []
        sel, pos = [], 0
        for s in scaffolds:
            bins = sorted(s.binList)
            self.sid[s.name] = len(self.start)
            self.names.append(s.name)
            self.start.append(pos)
            self.length.append(len(bins))
            sel.extend(map(where.__getitem__, bins))
            pos += len(bins)
        self.n = pos
        self.ctx.p2_select(sel)
        self.ctx.p2_layout(self.start, self.length)
        self._tables_k = None
        self._pos_cache = {}

    def covers(self, scaffs):
        return all(s.name in self.sid for s in scaffs)

    def describe(self, scaffs):
        """(ids, rev) of an arrangement; rev = 1 for '-'."""
        ids = np.fromiter((self.sid[s.name] for s in scaffs), dtype=np.int32, count=len(scaffs))
        rev = np.fromiter((1 if s.orientation == "-" else 0 for s in scaffs), dtype=np.uint8, count=len(scaffs))
        return ids, rev

    def positions(self, sid, rev):
        """Selection indices of one scaffold laid down forward / reversed (cached)."""
        key = (sid, bool(rev))
        hit = self._pos_cache.get(key)
        if hit is None:
            a = np.arange(self.start[sid], self.start[sid] + self.length[sid], dtype=np.int32)
            hit = self._pos_cache[key] = np.ascontiguousarray(a[::-1]) if rev else a
        return hit

    def node_row(self, ids, rev):
        return np.concatenate([self.positions(int(i), int(r)) for i, r in zip(ids, rev)]) if len(ids) else \
            np.zeros(0, np.int32)

    def tables(self, k):
        if self._tables_k != k:
            orders, orients = _enumeration(k)
            self.ctx.p2_window_tables(np.asarray(orders, dtype=np.int8),
                                      np.asarray([[1 if sg == "-" else 0 for sg in r] for r in orients], dtype=np.uint8))
            self._tables_k = k
        return _enumeration(k)


class SubMatrix:
    """What giveNewAdjMat returns here: the scaffolds of ``scaffList`` in their order and
    orientation AT CREATION (that order fixes the rounding of ``total``, OG:343/448/506), plus a
    cache of literal scores evaluated under that total."""

    def __init__(self, layout: ChromosomeLayout, scaffList):
        self.layout = layout
        self.ctx = layout.ctx
        self.ids, self.rev = layout.describe(scaffList)
        self.n = int(sum(layout.length[i] for i in self.ids))
        self._total = None
        self.exact = {}                # node-row bytes -> literal score under this total

    def __len__(self):
        return self.n

    def total(self) -> float:
        """Sum of everything above the diagonal with the reference's rounding (OG:343, 448, 506)."""
        if self._total is None:
            if self.n < 2:
                self._total = 0.0
            else:
                self.ctx.p2_set_arrangement(self.ids, self.rev)
                self._total = self.ctx.p2_arrangement_total()
        return self._total

    def first_strict_max(self, fast, floor, row_of):
        """The reference's ``if cost > bestCost`` scan over a step's candidates in enumeration order
        (OG:349,359,464,535), starting from ``bestCost = floor``: returns (index, cost) of the winner
        or (-1, floor).

        ``fast`` are the closed-form fp64 scores of all candidates; ``row_of(c)`` gives candidate c's
        bin order as selection indices.  The reference's comparisons are decided at the last bit -
        the same arrangement scored under two differently rounded totals (OG:506 vs OG:343) differs
        by an ulp, and that decides whether a pass "improves" - so every candidate within 1e-9 of
        the step's best is re-scored in the reference's exact operation order
        (hicmi_p2_score_exact) and the decision is taken on those values.  Identical bin orders
        (flipping a one-bin scaffold) share one literal evaluation and tie exactly, as they do in
        the reference."""
        fast = np.asarray(fast, dtype=np.float64)
        if SCORE_HOOK is not None:
            SCORE_HOOK(fast)
        ok = np.isfinite(fast)
        if not ok.any():
            return -1, floor
        top = max(float(fast[ok].max()), float(floor))
        near = np.flatnonzero(ok & (fast >= top - abs(top) * NEAR_TOP))
        if len(near) == 0:
            return -1, floor
        rows = [np.ascontiguousarray(row_of(int(c)), dtype=np.int32) for c in near]
        keys = [r.tobytes() for r in rows]
        todo = {}
        for r, key in zip(rows, keys):
            if key not in self.exact and key not in todo:
                todo[key] = r
        if todo:
            if len(rows[0]) < 2:
                vals = np.zeros(len(todo))                       # range(1, 1) is empty: cost 0.0
            else:
                vals = self.ctx.p2_score_exact(np.stack(list(todo.values())), self.total())
            for key, v in zip(todo, vals):
                self.exact[key] = float(v)
        pick, best = -1, floor
        for c, key in zip(near, keys):
            v = self.exact[key]
            if v > best:
                pick, best = int(c), v
        return pick, best


def buildAdjacencyMatrix(matrixFile, binList, binID_dict=False, device=0, ctx=None):
    """OG:65-93."""
    cache = os.environ.get("HICMI_MATRIX_CACHE")           # "1": beside the text file; or a directory (hostio.py)
    host = read_contact_matrix_cached(matrixFile, binList, cache) if cache else read_contact_matrix(matrixFile, binList)
    ctx = ctx or _lib.Context(device)
    ctx.set_contacts(host)
    print("Rows in adjacency matrix " + str(len(binList)))
    return GenomeMatrix(ctx)


def readGroupingsToValidBins(chromosomeGroupFile):
    """OG:200-214."""
    ids = {}
    with open(chromosomeGroupFile) as fh:
        for line in fh:
            line = line.strip("\r").strip("\n")
            if line[0] != "#":
                ids[int(line.split("\t")[0])] = ''
    return ids


def readChromsFromFile(inFile):
    """OG:216-237."""
    chroms, cur = [], []
    with open(inFile) as fh:
        fh.readline()
        text = fh.read()
    if "\r" in text:                                   # rare: keep the reference's exact stripping (OG:222)
        lines = [ln.strip("\r").strip("\n") for ln in text.splitlines(keepends=True)]
    else:
        lines = text.split("\n")
        if lines and lines[-1] == "":                   # the file's final newline
            lines.pop()
    add = cur.append
    for line in lines:
        if line[0] != "#":
            cols = line.split("\t", 2)
            add([int(cols[0]), cols[1]])
        else:
            chroms.append(cur)
            cur = []
            add = cur.append
    chroms.append(cur)
    print("Chromosomes found " + str(len(chroms)))
    print("Nodes found " + str(sum(len(c) for c in chroms)))
    return chroms


class Scaffold:
    """OG:239-254: name, bins in 5'->3' order of the current orientation, orientation sign."""

    def __init__(self, name, binList, orientation):
        self.name = name
        self.binList = binList
        self.orientation = orientation

    def flipOrientation(self):
        self.orientation = "-" if self.orientation == "+" else "+"
        self.binList = self.binList[::-1]


def initiateBinsAndScaffolds(nodeList):
    """OG:256-280: scaffolds in order of first appearance, bins ascending, then a stable sort by
    bin count, largest first."""
    bins_of = {}
    for bin_id, name in nodeList:
        b = bins_of.get(name)
        if b is None:
            bins_of[name] = [bin_id]
        else:
            b.append(bin_id)
    scaffDict = {name: Scaffold(name, sorted(b), "+") for name, b in bins_of.items()}      # (order of first appearance)
    print("Scaffolds to order for this chromosome " + str(len(scaffDict)))
    for s in scaffDict.values():
        s.nodeCount = len(s.binList)
    scaffList = sorted(scaffDict.values(), key=lambda s: len(s.binList), reverse=True)
    return scaffList, scaffDict


def pullScaffolds(puller, pullee, scaffsToPull):
    """OG:282-294."""
    for _ in range(scaffsToPull):
        if len(pullee) == 0:
            break
        puller.append(pullee.pop(0))
    return puller, pullee


def giveNewAdjMat(matrix: GenomeMatrix, scaffList, binList):
    """OG:296-308.  The device already holds the chromosome's sub-matrix (selected once by
    orderChromosome); this records ``scaffList``'s order/orientation - which fixes how ``total`` is
    rounded - and returns it with {binID: index in that order}.  Called outside orderChromosome it
    selects just these scaffolds."""
    if matrix.chrom is None or not matrix.chrom.covers(scaffList):
        matrix.chrom = ChromosomeLayout(matrix, scaffList, binList)
    return SubMatrix(matrix.chrom, scaffList), _OrderDict(scaffList)


class _OrderDict(dict):
    """{binID: index in the sub-matrix} of giveNewAdjMat (OG:302), filled on first use: the device
    path never needs it, only callers of the reference's function-level API do."""

    def __init__(self, scaffList):
        super().__init__()
        self._nodes = [n for s in scaffList for n in s.binList]
        self._filled = False

    def _fill(self):
        if not self._filled:
            self._filled = True
            self.update((b, i) for i, b in enumerate(self._nodes))

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)


def reorderScaffList(orderList, orientationList, scaffDict):
    """OG:310-321."""
    scaffs, nodes = [], []
    for name, orient in zip(orderList, orientationList):
        s = scaffDict[name]
        if s.orientation != orient:
            s.flipOrientation()
        scaffs.append(s)
        nodes += s.binList
    return scaffs, nodes


def costFunction(matrix, total):
    """OG:323-330 for an explicit (already permuted) host matrix: uploaded and scored on the GPU in
    the reference's operation order.  Kept for callers of the reference's function-level API."""
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64))
    if len(m) < 2:
        return 0.0
    with _lib.Context(0) as ctx:
        ctx.set_contacts(m)
        ident = np.arange(len(m), dtype=np.int32)
        ctx.p2_select(ident)
        return float(ctx.p2_score_exact(ident[None, :], float(total))[0])


costFunction_numba = costFunction      # OG:184-191


def calcPossiblePerms(N):
    """OG:374-379."""
    return math.factorial(N) * (2 ** N) / 2


# ---- enumeration order (defines tie-breaking) --------------------------------------
def permutations(elementList, paths, k=0):
    """OG:381-394: every order of ``elementList``, in the order produced by swapping position k with
    each later position and recursing."""
    if k == len(elementList):
        paths.append(list(elementList))
        return paths
    for i in range(k, len(elementList)):
        elementList[k], elementList[i] = elementList[i], elementList[k]
        permutations(elementList, paths, k + 1)
        elementList[k], elementList[i] = elementList[i], elementList[k]
    return paths


def removeReverseDuplicates(permList):
    """OG:396-411: of each (order, reversed order) pair keep whichever was enumerated first."""
    pending, kept = set(), []
    for p in permList:
        key = tuple(p)
        if key[::-1] in pending:
            pending.discard(key[::-1])
        else:
            pending.add(key)
            kept.append(p)
    return kept


def plusMinusPerms(elementList):
    """OG:413-430: all-plus first, then for i = 0..k-1 the distinct arrangements of i '+' and k-i '-'
    in swap-enumeration order."""
    k = len(elementList)
    seen, out = set(), []
    for cand in [["+"] * k] + [p for i in range(k) for p in permutations(["+"] * i + ["-"] * (k - i), [], 0)]:
        key = tuple(cand)
        if key not in seen:
            seen.add(key)
            out.append(list(cand))
    return out


_ENUM_CACHE = {}


def _enumeration(k):
    """(orders over slots 0..k-1, orientations), cached.  orders[0] is the identity."""
    if k not in _ENUM_CACHE:
        orders = removeReverseDuplicates(permutations(list(range(k)), [], 0))
        orients = plusMinusPerms(list(range(k)))
        _ENUM_CACHE[k] = (orders, orients, {tuple(r): i for i, r in enumerate(orients)})
    return _ENUM_CACHE[k][0], _ENUM_CACHE[k][1]


def _orient_index(k, signs):
    _enumeration(k)
    return _ENUM_CACHE[k][2][tuple(signs)]


def _window_scores(view: SubMatrix, arrangement, first, k, known_fast=None):
    """Fast scores of all k!/2 * 2^k candidates for the window arrangement[first:first+k] (everything
    else fixed), from one hicmi_p2_score_window launch.  Returns (fast, row_of)."""
    layout, ctx = view.layout, view.ctx
    orders, orients = layout.tables(k)
    ids, rev = layout.describe(arrangement)
    total = view.total()
    ctx.p2_set_arrangement(ids, rev)
    delta = ctx.p2_score_window(first, k)
    if k == len(arrangement):
        fast = delta / total                              # nothing outside the window
    else:
        c0 = _orient_index(k, [s.orientation for s in arrangement[first:first + k]])   # orders[0] = identity
        base = ctx.p2_arrangement_score(total) if known_fast is None else known_fast
        fast = base + (delta - delta[c0]) / total
    n_ori = len(orients)
    win = ids[first:first + k]
    ends = []

    def row_of(c):
        if not ends:                                      # built only if a candidate gets short-listed
            ends.append(layout.node_row(ids[:first], rev[:first]))
            ends.append(layout.node_row(ids[first + k:], rev[first + k:]))
        o, r = orders[c // n_ori], orients[c % n_ori]
        mid = [layout.positions(int(win[j]), sg == "-") for j, sg in zip(o, r)]
        return np.concatenate([ends[0]] + mid + [ends[1]])
    return fast, row_of


def _fused(ctx):
    """Use the one-call decision steps of libhicmi (hicmi_p2_decide_*) unless a test wants to see
    every fast score (SCORE_HOOK) or the context is a test double without them."""
    return SCORE_HOOK is None and hasattr(ctx, "p2_decide_window")


# ---- search ---------------------------------------------------------------------------------------
def bruteForceBestScore(sObjList, scaffDict, matrix: SubMatrix, orderDict):
    """OG:432-473: all k!/2 orders x 2^k orientations of the k largest scaffolds in one launch."""
    names = [s.name for s in sObjList]
    k = len(names)
    orders, orients = _enumeration(k)
    total = matrix.total()
    if total == 0:
        print("WARNING/ERROR - Zero contact values found between scaffolds assigned to chromosome group "
              + ",".join(str(e) for e in names))
        return [names[i] for i in orders[0]], list(orients[0]), 0.0
    print("Initial permutations to test " + str(len(orders) * len(orients)) + "...")
    if _fused(matrix.ctx):
        matrix.layout.tables(k)
        matrix.ctx.p2_set_arrangement(*matrix.layout.describe(sObjList))
        best, best_c, _pf = matrix.ctx.p2_decide_window(0, k, total, 0., None)
    else:
        fast, row_of = _window_scores(matrix, sObjList, 0, k)
        best, best_c = matrix.first_strict_max(fast, 0., row_of)      # first strict maximum above 0. (OG:464)
    # the enumeration leaves every scaffold in the last candidate's orientation (OG:459)
    reorderScaffList([names[i] for i in orders[-1]], orients[-1], scaffDict)
    if best < 0:
        raise RuntimeError("no candidate order scored above 0 (the reference fails here too, OG:473 -> OG:576)")
    o, r = orders[best // len(orients)], orients[best % len(orients)]
    return [names[i] for i in o], list(r), best_c


def checkAllScores(adjMat: SubMatrix, orderDict, orderedScaffs, scaffToCheck):
    """OG:332-372: try the scaffold at every gap, both orientations.  The scaffold is flipped once
    per gap and stays flipped, so the orientation tried first alternates with the gap index."""
    layout, ctx = adjMat.layout, adjMat.ctx
    gaps = len(orderedScaffs) + 1
    flip = {"+": "-", "-": "+"}
    new_id = layout.sid[scaffToCheck.name]
    ids, rev = layout.describe(orderedScaffs)
    if _fused(ctx) and len(ids) > 0:
        gap, r, bestCost = ctx.p2_decide_insertion(ids, rev, new_id, scaffToCheck.orientation == "-")
        bestGap, bestOrient = (gap, "-" if r else "+") if gap >= 0 else (0, "+")
        if gaps % 2 == 1:                               # one flip per gap (OG:356)
            scaffToCheck.flipOrientation()
        if scaffToCheck.orientation != bestOrient:
            scaffToCheck.flipOrientation()
        orderedScaffs.insert(bestGap, scaffToCheck)
        return orderedScaffs, bestCost
    total = adjMat.total()
    tags, o = [], scaffToCheck.orientation
    for i in range(gaps):
        tags += [(i, o), (i, flip[o])]
        o = flip[o]
    n_all = adjMat.n
    if n_all < 2:
        fast = np.zeros(2 * gaps)
    elif len(ids) == 0:
        rows = np.stack([layout.positions(new_id, sg == "-") for _i, sg in tags])
        fast = ctx.p2_score(rows, total)
    else:
        ctx.p2_set_arrangement(ids, rev)
        by_gap_rev = ctx.p2_score_insertions(new_id, total)             # [2*gap + (orientation == '-')]
        fast = np.array([by_gap_rev[2 * i + (1 if sg == "-" else 0)] for i, sg in tags])
    pieces = [layout.positions(int(i), int(r)) for i, r in zip(ids, rev)]

    def row_of(c):
        i, sg = tags[c]
        return np.concatenate(pieces[:i] + [layout.positions(new_id, sg == "-")] + pieces[i:])
    pick, bestCost = adjMat.first_strict_max(fast, 0., row_of)
    bestGap, bestOrient = tags[pick] if pick >= 0 else (0, "+")
    if gaps % 2 == 1:                                   # one flip per gap (OG:356)
        scaffToCheck.flipOrientation()
    if scaffToCheck.orientation != bestOrient:
        scaffToCheck.flipOrientation()
    orderedScaffs.insert(bestGap, scaffToCheck)
    return orderedScaffs, bestCost


def _insertion_job(orderedScaffolds, scaffoldList, matrix: GenomeMatrix):
    """(ids, rev, new_ids) for hicmi_p2_insert_all, or None when the whole-loop call does not apply (test
    doubles, score hooks, nothing left to add, a scaffold that was flipped before)."""
    layout = matrix.chrom
    if (layout is not None and _fused(matrix.ctx) and len(scaffoldList) > 0 and len(orderedScaffolds) > 0
            and layout.covers(orderedScaffolds) and layout.covers(scaffoldList)
            and all(s.orientation == "+" for s in scaffoldList)):
        ids, rev = layout.describe(orderedScaffolds)
        return ids, rev, [layout.sid[s.name] for s in scaffoldList]
    return None


def _insertion_result(ids, rev, orderedScaffolds, scaffoldList, matrix: GenomeMatrix):
    """Scaffold objects in the order / orientation hicmi_p2_insert_all returned; empties scaffoldList."""
    layout = matrix.chrom
    by_name = {s.name: s for s in orderedScaffolds + scaffoldList}
    del scaffoldList[:]
    ordered, _nodes = reorderScaffList([layout.names[i] for i in ids], ["-" if r else "+" for r in rev], by_name)
    return ordered


def orderRemainderScaffolds(orderedScaffolds, scaffoldList, orderDict, matrix: GenomeMatrix, binList):
    """OG:475-493 (a do-while: with nothing left to add, the last ordered scaffold is re-inserted)."""
    job = _insertion_job(orderedScaffolds, scaffoldList, matrix)
    if job is not None:
        ids, rev, bestCost = matrix.ctx.p2_insert_all(*job)
        return _insertion_result(ids, rev, orderedScaffolds, scaffoldList, matrix), bestCost
    while True:
        orderedScaffolds, scaffoldList = pullScaffolds(orderedScaffolds, scaffoldList, 1)
        adjMat, orderDict = giveNewAdjMat(matrix, orderedScaffolds, binList)
        newScaff = orderedScaffolds.pop(-1)
        orderedScaffolds, bestCost = checkAllScores(adjMat, orderDict, orderedScaffolds, newScaff)
        if len(scaffoldList) == 0:
            break
    return orderedScaffolds, bestCost


def scanOrdering(orderedScaffolds, scaffoldDict, orderDict, matrix: GenomeMatrix, binList, bestCost, scanScaffolds=5):
    """OG:495-549: slide a window of ``scanScaffolds`` scaffolds along the chromosome; every
    order/orientation of the window is scored on the WHOLE chromosome; repeat until a full pass
    brings no improvement."""
    adjMat, orderDict = giveNewAdjMat(matrix, orderedScaffolds, binList)
    total = adjMat.total()
    bestOrder = [s.name for s in orderedScaffolds]
    bestOrientation = [s.orientation for s in orderedScaffolds]
    roundNumber = 0
    w = scanScaffolds
    orders, orients = _enumeration(w)
    cur_fast = None                                     # fast score of the current arrangement
    if _fused(adjMat.ctx):
        layout = adjMat.layout
        layout.tables(w)
        ids, rev = layout.describe(orderedScaffolds)
        if hasattr(adjMat.ctx, "p2_scan_all"):
            # the whole loop as one native call (the interpreter lock is free for the other chromosomes meanwhile)
            ids, rev, bestCost, cur_fast, roundNumber = adjMat.ctx.p2_scan_all(ids, rev, w, total, bestCost, cur_fast)
            for r in range(roundNumber):
                print("Working on round " + str(r + 1) + " of final step...")
        else:
            while True:
                print("Working on round " + str(roundNumber + 1) + " of final step...")
                ids, rev, bestCost, cur_fast, improved = adjMat.ctx.p2_scan_pass(ids, rev, w, total, bestCost, cur_fast)
                roundNumber += 1
                if not improved:
                    break
        orderedScaffolds, _nodes = reorderScaffList([layout.names[i] for i in ids], ["-" if r else "+" for r in rev],
                                                    scaffoldDict)
        print("Sliding window conversion after " + str(roundNumber) + " rounds")
        print("Best cost at the end of the final step = " + str(bestCost))
        if _PROFILE:
            sys.stderr.write("[hicmi] part2 scan: %d scaffolds, %d rounds\n" % (len(ids), roundNumber))
        return orderedScaffolds, bestCost
    while True:
        improved = False
        print("Working on round " + str(roundNumber + 1) + " of final step...")
        for i in range(0, len(orderedScaffolds) - w + 1):
            if _fused(adjMat.ctx):
                adjMat.layout.tables(w)
                adjMat.ctx.p2_set_arrangement(*adjMat.layout.describe(orderedScaffolds))
                pick, bestCost, cur_fast = adjMat.ctx.p2_decide_window(i, w, total, bestCost, cur_fast)
                fast = None
            else:
                if cur_fast is None:
                    ids, rev = adjMat.layout.describe(orderedScaffolds)
                    adjMat.ctx.p2_set_arrangement(ids, rev)
                    cur_fast = adjMat.ctx.p2_arrangement_score(total)
                fast, row_of = _window_scores(adjMat, orderedScaffolds, i, w, known_fast=cur_fast)
                pick, bestCost = adjMat.first_strict_max(fast, bestCost, row_of)   # strict '>' vs the global best (OG:535)
            if pick >= 0:
                improved = True
                o, r = orders[pick // len(orients)], orients[pick % len(orients)]
                window = orderedScaffolds[i:i + w]
                names = [s.name for s in orderedScaffolds]
                outside = {s.name: s.orientation for s in orderedScaffolds}
                bestOrder = names[:i] + [window[j].name for j in o] + names[i + w:]
                bestOrientation = ([outside[nm] for nm in names[:i]] + list(r) + [outside[nm] for nm in names[i + w:]])
                if fast is not None:
                    cur_fast = float(fast[pick])
            orderedScaffolds, _nodes = reorderScaffList(bestOrder, bestOrientation, scaffoldDict)
        roundNumber += 1
        if not improved:
            break
    print("Sliding window conversion after " + str(roundNumber) + " rounds")
    print("Best cost at the end of the final step = " + str(bestCost))
    return orderedScaffolds, bestCost


def _startChromosome(chromGroup, matrix: GenomeMatrix, binList, nScaffolds=6, scanScaffolds=5):
    """OG:551-576: the selection, the brute-force order of the largest scaffolds; returns the state the
    insertion and scan phases continue from."""
    if nScaffolds >= 9:
        print("Number of initial scaffolds to order by brute force method is set too high... setting it to 8")
        nScaffolds = 8
    if scanScaffolds > nScaffolds:
        scanScaffolds = nScaffolds
    tm = [time.perf_counter()] if _PROFILE else None
    scaffoldList, scaffoldDict = initiateBinsAndScaffolds(chromGroup)
    if tm: tm.append(time.perf_counter())
    matrix.chrom = ChromosomeLayout(matrix, scaffoldList, binList)      # one selection for the whole chromosome
    if tm: tm.append(time.perf_counter())
    orderedScaffolds, scaffoldList = pullScaffolds([], scaffoldList, nScaffolds)
    adjMat, orderDict = giveNewAdjMat(matrix, orderedScaffolds, binList)
    if tm: tm.append(time.perf_counter())
    bfOrder, bfOrient, _bfScore = bruteForceBestScore(orderedScaffolds, scaffoldDict, adjMat, orderDict)
    if tm: tm.append(time.perf_counter())
    orderedScaffolds, _nodes = reorderScaffList(bfOrder, bfOrient, scaffoldDict)
    if tm:
        tm.append(time.perf_counter())
        sys.stderr.write("[hicmi] part2 start of a %d-bin chromosome (ms): scaffolds %.2f, layout + selection %.2f, sub-matrix view %.2f, "
                         "brute force %.2f, reorder %.2f\n" % ((len(chromGroup),) + tuple((b - a) * 1e3 for a, b in zip(tm, tm[1:]))))
    return {"ordered": orderedScaffolds, "rest": scaffoldList, "dict": scaffoldDict, "orderDict": orderDict,
            "nScaffolds": nScaffolds, "scanScaffolds": scanScaffolds}


def _finishChromosome(state, orderedScaffolds, bestCost, matrix: GenomeMatrix, binList):
    """OG:578-586: the sliding-window rounds and the final listing."""
    print("BestCost at the end of first two steps " + str(bestCost))
    if len(orderedScaffolds) > state["nScaffolds"]:
        orderedScaffolds, bestCost = scanOrdering(orderedScaffolds, state["dict"], state["orderDict"], matrix, binList,
                                                  bestCost, scanScaffolds=state["scanScaffolds"])
    print("Final ordering...")
    for s in orderedScaffolds:
        print(s.name, s.orientation)
    orderChromosome.last_cost = bestCost
    return orderedScaffolds


def orderChromosome(chromGroup, matrix: GenomeMatrix, binList, nScaffolds=6, scanScaffolds=5):
    """OG:551-586."""
    state = _startChromosome(chromGroup, matrix, binList, nScaffolds, scanScaffolds)
    orderedScaffolds, bestCost = orderRemainderScaffolds(state["ordered"], state["rest"], state["orderDict"], matrix,
                                                         binList)
    return _finishChromosome(state, orderedScaffolds, bestCost, matrix, binList)


def chromosomesOfRank(chromList, rank, world):
    """The chromosomes one rank orders when a single map is spread over ``world`` processes: largest first, each
    to the rank with the least work so far (work ~ bins squared, the size of the chromosome's sub-matrix; ties go
    to the lowest rank).  Every rank computes the same deal from the same group file - no exchange needed."""
    load = [0] * world
    mine = []
    for i in sorted(range(len(chromList)), key=lambda i: (-len(chromList[i]), i)):
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += len(chromList[i]) ** 2
        if r == rank:
            mine.append(i)
    return sorted(mine)


def orderGenome(matrix: GenomeMatrix, chromList, binList, resolution, nScaffolds=6, scanScaffolds=5, plotChrom=True,
                showPlot=True, savePlotDir=False, plotTitleSuffix=False, shard=None, on_native_phase=None):
    """OG:591-628.  Chromosomes are independent (OG:608-612), so they are ordered concurrently:
    one host thread + one libhicmi context (own HIP stream, own scratch) per chromosome in flight,
    all reading the same device-resident contact matrix.  Results are collected in file order.
    The per-chromosome figures (OG:615-622) are drawn afterwards from the device-resident matrix.

    ``shard=(rank, world)``: one map over several GPUs.  Every rank holds the map, orders only the
    chromosomes ``chromosomesOfRank`` deals to it and draws their figures; one object all-gather of the ordered
    scaffold lists (names, orientations, bin IDs: a few KB) gives every rank the whole genome order."""
    t0 = time.time()
    t0p = time.perf_counter()
    pieces = None                                          # per chromosome: its text in the two output files, when formatted on the way
    orderGenome.file_text = None
    indices = list(range(len(chromList))) if shard is None else chromosomesOfRank(chromList, shard[0], shard[1])
    n_workers = 1 if SCORE_HOOK is not None else max(1, min(WORKERS, len(indices)))
    matrix.bin_index(binList)

    def one(i, m):
        print("#####################\n#####################")
        print("Working on Chr_" + str(i + 1) + "...")
        if not _PROFILE:
            return orderChromosome(chromList[i], m, binList, nScaffolds=nScaffolds, scanScaffolds=scanScaffolds)
        ts = time.perf_counter()
        res = orderChromosome(chromList[i], m, binList, nScaffolds=nScaffolds, scanScaffolds=scanScaffolds)
        te = time.perf_counter()
        sys.stderr.write("[hicmi] part2 chr %d: %d bins, %d scaffolds, start %.1f ms, %.1f ms\n"
                         % (i + 1, len(chromList[i]), len(res), (ts - t0p) * 1e3, (te - ts) * 1e3))
        return res

    if n_workers == 1 or not hasattr(matrix.ctx, "workers"):
        done = {i: one(i, matrix) for i in indices}
    elif LOCKSTEP and hasattr(matrix.ctx, "p2_insert_all_multi"):
        # Three phases over ALL chromosomes, one context each: (1) selection + brute force on worker threads,
        # (2) every chromosome's insertion loop in lock step, decided on the device - one queue of launches
        # serving all of them (hicmi_p2_insert_all_multi), (3) the sliding-window rounds on worker threads.
        lanes = [matrix] + [GenomeMatrix(c) for c in matrix.ctx.workers(len(indices) - 1)]
        for m in lanes[1:]:
            m._bin_index, m._bin_index_src = matrix._bin_index, matrix._bin_index_src
        lanes = dict(zip(indices, lanes))
        todo = sorted(indices, key=lambda i: -len(chromList[i]))                    # largest first
        marks = [time.perf_counter()]
        if _PROFILE:
            sys.stderr.write("[hicmi] part2 set-up (bin index, one context per chromosome): %.1f ms\n" % ((marks[0] - t0p) * 1e3))

        def start(i):
            print("#####################\n#####################")
            print("Working on Chr_" + str(i + 1) + "...")
            return i, _startChromosome(chromList[i], lanes[i], binList, nScaffolds, scanScaffolds)
        with ThreadPoolExecutor(max_workers=n_workers) as pool:
            # the start phase runs on THIS thread, one chromosome after the other: it is half interpreter work and half
            # short native calls, and threads that hand the interpreter lock to each other at every one of those calls
            # took 12-14 ms (16k) / 22-24 ms (32k) where the plain loop takes 7.7 / 13.5 ms
            if START_THREADS > 0:
                with ThreadPoolExecutor(max_workers=START_THREADS) as starters:
                    states = dict(starters.map(start, todo))
            else:
                states = dict(map(start, todo))
            marks.append(time.perf_counter())
            jobs, job_of = [], []
            for i in todo:
                job = _insertion_job(states[i]["ordered"], states[i]["rest"], lanes[i])
                if job is not None:
                    jobs.append((lanes[i].ctx,) + job)
                    job_of.append(i)
            if on_native_phase is not None:
                on_native_phase()                     # a long native call follows: background Python work may take the GIL
            raw = dict(zip(job_of, matrix.ctx.p2_insert_all_multi(jobs)))     # (turned into scaffold lists by the scan threads)
            marks.append(time.perf_counter())

            def finish(i):
                tf = time.perf_counter()
                st = states[i]
                if i in raw:
                    ids, rev, best = raw[i]
                    ordered = _insertion_result(ids, rev, st["ordered"], st["rest"], lanes[i])
                else:                                                           # e.g. nothing left to add (OG:475-493)
                    ordered, best = orderRemainderScaffolds(st["ordered"], st["rest"], st["orderDict"], lanes[i], binList)
                res = _finishChromosome(st, ordered, best, lanes[i], binList)
                # this chromosome's lines of the two output files, formatted here - beside the other chromosomes' native scan
                # calls - instead of for the whole genome at the very end (2.5 ms at 16k, 5 ms at 32k, nothing to hide behind)
                text = (_scaffold_lines(res), _bin_rows(res))
                if _PROFILE:
                    sys.stderr.write("[hicmi] part2 scan of chromosome %d (%d bins): start +%.1f ms, %.1f ms\n"
                                     % (i + 1, len(chromList[i]), (tf - marks[2]) * 1e3, (time.perf_counter() - tf) * 1e3))
                return i, (res, text)
            finished = dict(pool.map(finish, todo))
            done = {i: v[0] for i, v in finished.items()}
            if shard is None:
                pieces = {i: v[1] for i, v in finished.items()}
        marks.append(time.perf_counter())
        if _PROFILE:
            sys.stderr.write("[hicmi] part2 lock step: start %.1f ms, insertion %.1f ms (%d chromosomes), scan %.1f ms\n"
                             % ((marks[1] - marks[0]) * 1e3, (marks[2] - marks[1]) * 1e3, len(jobs),
                                (marks[3] - marks[2]) * 1e3))
    else:
        lanes = [matrix] + [GenomeMatrix(c) for c in matrix.ctx.workers(n_workers - 1)]
        for m in lanes[1:]:
            m._bin_index, m._bin_index_src = matrix._bin_index, matrix._bin_index_src
        free = list(lanes)
        todo = sorted(indices, key=lambda i: -len(chromList[i]))                    # largest first

        def run(i):
            m = free.pop()
            try:
                return i, one(i, m)
            finally:
                free.append(m)
        with ThreadPoolExecutor(max_workers=n_workers) as pool:
            done = dict(pool.map(run, todo))
    if shard is not None:
        from . import dist
        done = dist.gather_results(done)
    fullGenomeOrder = [done[i] for i in range(len(chromList))]
    if pieces is not None and len(pieces) == len(chromList):
        orderGenome.file_text = [pieces[i] for i in range(len(chromList))]
    print("RunTime for total genome = " + str(time.time() - t0))
    if plotChrom is True and plotModule.plots_enabled(savePlotDir):
        # OG:615-622: one figure per chromosome.  The device reductions run here one after the other (a context
        # is not thread-safe), the drawing and PNG encoding on worker threads.
        where = matrix.bin_index(binList)
        todo = []
        for i in indices:
            rows = [where[b] for s in fullGenomeOrder[i] for b in s.binList]
            if len(rows) == 0:
                continue
            img = plotModule.DeviceImage(matrix.ctx, 0, rows)
            img.prefetch([1, 98], plotModule.figure_pixels(len(rows), 24, 24))
            todo.append((i, img))

        def draw(item):
            i, img = item
            chrName = "Chr_" + str(i + 1)
            plotModule.plotContactMap(img, resolution=resolution, tickCount=11, highlightChroms=False, wInches=24,
                                      hInches=24, lP=1, hP=98, reverseColorMap='', showPlot=False,
                                      savePlot=savePlotDir + "/" + chrName + ".png", title=chrName,
                                      titleSuffix=plotTitleSuffix)
        with ThreadPoolExecutor(max_workers=max(1, min(8, len(todo)))) as pool:
            list(pool.map(draw, todo))
        print("RunTime for total genome with plotting and saving .pngs = " + str(time.time() - t0))
    return fullGenomeOrder


def _scaffold_lines(group):
    """A chromosome's lines of the scaffold-order file (OG:638-641)."""
    return "".join([s.name + "\t" + s.orientation + "\n" for s in group])


def _bin_rows(scaffoldList):
    """Newline-PREFIXED ``scaffold<TAB>bin`` rows (OG:654-657) and how many."""
    rows, n_rows = [], 0
    for s in scaffoldList:
        if len(s.binList):
            head = "\n" + s.name + "\t"
            rows.append(head + head.join(map(str, s.binList)))
            n_rows += len(s.binList)
    return "".join(rows), n_rows


def writeScaffoldOrderingsToFile(sOrderings, outFile, lines=None):
    """OG:630-644.  ``lines``: per group, its lines already formatted (_scaffold_lines)."""
    text = []
    for k, group in enumerate(sOrderings):
        text.append("### Chromosome grouping " + str(k + 1) + " ###\n")
        text.append(lines[k] if lines is not None else _scaffold_lines(group))
    with open(outFile, "w") as fh:
        fh.write("".join(text))
    print("Chromosome groups written to file " + str(len(sOrderings)))
    print("Scaffolds written to file " + str(sum(len(g) for g in sOrderings)))


def writeBinIDsOrderingToFile(scaffoldList, outFile, rows=None):
    """OG:646-660: header line, then newline-PREFIXED rows (no trailing newline).  ``rows``: (text, count) pieces that
    together are _bin_rows(scaffoldList)."""
    if rows is None:
        rows = [_bin_rows(scaffoldList)]
    with open(outFile, "w") as fh:
        fh.write("#ScaffoldID\tHiCPro-BinID" + "".join(r[0] for r in rows))
    print("BinIDs written to file " + str(sum(r[1] for r in rows)))


def getChromosomeOutlineCoords(orderedChromosomes):
    """OG:662-674."""
    coords, index = [], 0
    for group in orderedChromosomes:
        index += sum(len(s.binList) for s in group)
        coords.append(index)
    return coords


def runPipeline(hicProBedFile, hicProBiasFile, hicProMatrixFile, chromosomeGroupFile, chromosomeOrderFile,
                savePlotsDirectory, chromosomePlotSuffix, fullGenomePlot, fullGenomePlotTitle, plotOrderFile,
                nScaffolds, scanScaffolds, resolution, device=0, resident=None):
    """OG:679-712, same positional arguments (``device`` and ``resident`` are optional extras).

    ``resident=(DeviceMatrix, bins of its rows)`` from Part 1's ``runPipeline(..., keep_resident=True)``: the contact
    matrix already in HBM is used instead of parsing the HiC-Pro text again.  The reference re-loads the matrix
    restricted to the grouped bins (OG:688-690); here the grouped bins are selected by ID from the full resident
    matrix (raw contacts, .bed order) - the same cells, and bins outside every group are never touched."""
    print("########################################")
    print("### Working on Part2 of the pipeline ###")
    t0 = time.time()
    if resident is not None:
        adjMat, binList = GenomeMatrix(resident[0].ctx), resident[1]
    else:
        binDict = readGroupingsToValidBins(chromosomeGroupFile)
        binList = initiateLoci(hicProBedFile, hicProBiasFile, binID_dict=binDict)
        adjMat = buildAdjacencyMatrix(hicProMatrixFile, binList, device=device)
    try:
        orderedChromosomes = runResident(adjMat, binList, chromosomeGroupFile, chromosomeOrderFile, plotOrderFile,
                                         nScaffolds, scanScaffolds, resolution, savePlotDir=savePlotsDirectory,
                                         plotTitleSuffix=chromosomePlotSuffix)
        if plotModule.plots_enabled(fullGenomePlot):                      # OG:700-707
            where = adjMat.bin_index(binList)
            rows = [where[b] for group in orderedChromosomes for s in group for b in s.binList]
            plotModule.plotContactMap(plotModule.DeviceImage(adjMat.ctx, 0, rows), resolution=resolution, tickCount=11,
                                      highlightChroms=getChromosomeOutlineCoords(orderedChromosomes), wInches=32,
                                      hInches=32, lP=2, hP=98, reverseColorMap='', showPlot=False,
                                      savePlot=fullGenomePlot, title=fullGenomePlotTitle, titleSuffix=False)
    finally:
        adjMat.ctx.close()
    print("Total run-time  for Part2 = " + str(time.time() - t0))
    print("- Part 2 (chromosome ordering) completed successfully")


def runResident(adjMat: GenomeMatrix, binList, chromosomeGroupFile, chromosomeOrderFile, plotOrderFile,
                nScaffolds, scanScaffolds, resolution, savePlotDir=False, plotTitleSuffix=False, shard=None,
                chromosomeList=None, on_native_phase=None):
    """OG:691-709 on contacts that are already resident in HBM (what bench.py times).  ``binList``
    gives the bin of every row of the device matrix; bins that Part 1 did not assign to a group are
    simply never selected, which is what the reference's re-load restricted to grouped bins
    (OG:688-690) amounts to.  ``shard=(rank, world)``: see ``orderGenome``; every rank returns the whole order
    and rank 0 writes the two files.  ``chromosomeList``: the groups as readChromsFromFile would return them for
    chromosomeGroupFile (Part 1's ``DeviceMatrix.chromosome_groups``) when that file is still being written."""
    with paused_gc():
        if chromosomeList is None:
            chromosomeList = readChromsFromFile(chromosomeGroupFile)
        else:
            print("Chromosomes found " + str(len(chromosomeList)))
            print("Nodes found " + str(sum(len(c) for c in chromosomeList)))
        orderedChromosomes = orderGenome(adjMat, chromosomeList, binList, resolution, nScaffolds=nScaffolds,
                                         scanScaffolds=scanScaffolds, plotChrom=True, showPlot=False,
                                         savePlotDir=savePlotDir, plotTitleSuffix=plotTitleSuffix, shard=shard,
                                         on_native_phase=on_native_phase)
        tw = time.perf_counter()
        if shard is None or shard[0] == 0:
            text = orderGenome.file_text                        # formatted per chromosome beside the scans, or None
            writeScaffoldOrderingsToFile(orderedChromosomes, chromosomeOrderFile, None if text is None else [t[0] for t in text])
            writeBinIDsOrderingToFile([s for group in orderedChromosomes for s in group], plotOrderFile,
                                      None if text is None else [t[1] for t in text])
        if _PROFILE:
            sys.stderr.write("[hicmi] part2 files: %.1f ms\n" % ((time.perf_counter() - tw) * 1e3))
    return orderedChromosomes
