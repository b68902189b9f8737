"""Part 2 (order + orient scaffolds inside each chromosome) on MI355X: drop-in for the reference
module of the same name (/root/reference/HIC_ASSEMBLER/orderGenome.py, OG below).

Same ``runPipeline`` signature, same input/output files, same search (brute force over the
largest scaffolds, greedy insertion of the rest, sliding-window re-permutation to a fixed point)
with the same enumeration order and first-strict-maximum tie-breaking.  What changes is where the
objective is evaluated: the reference gathers a permuted copy of the matrix with ``numpy.ix_`` and
runs a Numba loop once per candidate (OG:348,358,463,534); here every candidate of a step is one
row of an index array scored in one launch of k_p2_score (hicmi_p2_score), reading the
chromosome's sub-matrix - resident in HBM - through the permutation.  fp64 on the GPU.
"""
from __future__ import annotations

import math
import time

import numpy as np

from . import _lib
from .hostio import Bin, initiateLoci, read_contact_matrix  # noqa: F401


# ------------------------------------------------------------------------------------------------
class GenomeMatrix:
    """Raw contacts of the grouped bins, resident on the GPU (OG:690)."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx

    def __len__(self):
        return self.ctx.n


class SubMatrix:
    """The device-side selection made by giveNewAdjMat (OG:296-308)."""

    def __init__(self, ctx: _lib.Context, n: int):
        self.ctx = ctx
        self.n = n
        self._total = None

    def __len__(self):
        return self.n

    def total(self) -> float:
        """Sum of everything above the diagonal (OG:343, 448, 506)."""
        if self._total is None:
            self._total = 0.0 if self.n < 2 else self.ctx.p2_total()
        return self._total

    def scores(self, perms) -> np.ndarray:
        """costFunction_numba (OG:184-191) of every row of ``perms`` (positions into this selection),
        closed form in fp64 (k_p2_score)."""
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.shape[1] < 2:
            return np.zeros(perms.shape[0])             # range(1, 1) is empty: cost 0.0
        return self.ctx.p2_score(perms, self.total())

    def first_strict_max(self, perms, floor):
        """The reference's ``if cost > bestCost`` scan over the candidates in enumeration order
        (OG:349,359,464,535), starting from ``bestCost = floor``: returns (index, cost) of the
        winner or (-1, floor).  Those comparisons are decided at the last bit (the same arrangement
        under two differently rounded totals differs by an ulp), so every candidate whose fast
        score is within 1e-9 of the step's best is re-scored in the reference's exact operation
        order (k_p2_diag_sums / hicmi_p2_score_exact) and the decision uses those values."""
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        fast = self.scores(perms)
        ok = np.isfinite(fast)
        if not ok.any():
            return -1, floor
        top = max(float(fast[ok].max()), float(floor))
        near = np.flatnonzero(ok & (fast >= top - abs(top) * 1e-9))
        if len(near) == 0:
            return -1, floor
        if perms.shape[1] < 2:
            exact = np.zeros(len(near))
        else:
            exact = self.ctx.p2_score_exact(perms[near], self.total())
        pick, best = -1, floor
        for c, v in zip(near, exact):
            if v > best:
                pick, best = int(c), float(v)
        return pick, best


def buildAdjacencyMatrix(matrixFile, binList, binID_dict=False, device=0, ctx=None):
    """OG:65-93."""
    host = read_contact_matrix(matrixFile, binList)
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
        for line in fh:
            line = line.strip("\r").strip("\n")
            if line[0] != "#":
                cols = line.split("\t")
                cur.append([int(cols[0]), cols[1]])
            else:
                chroms.append(cur)
                cur = []
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
    scaffDict = {}
    for bin_id, name in nodeList:
        if name not in scaffDict:
            scaffDict[name] = Scaffold(name, [], "+")
        scaffDict[name].binList.append(bin_id)
    print("Scaffolds to order for this chromosome " + str(len(scaffDict)))
    for s in scaffDict.values():
        s.binList = sorted(s.binList)
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
    """OG:296-308: select the bins of ``scaffList`` (in list order, current orientation) on the
    device; returns the selection and {binID: position in the selection}."""
    nodes = [n for s in scaffList for n in s.binList]
    orderDict = {b: i for i, b in enumerate(nodes)}
    where = getattr(matrix, "_bin_index", None)
    if where is None or getattr(matrix, "_bin_index_src", None) is not binList:
        where = {b.ID: i for i, b in enumerate(binList)}
        matrix._bin_index, matrix._bin_index_src = where, binList
    matrix.ctx.p2_select([where[b] for b in nodes])
    return SubMatrix(matrix.ctx, len(nodes)), orderDict


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
    """OG:323-330 for an explicit (already permuted) host matrix: uploaded and scored on the GPU.
    Kept for callers of the reference's function-level API; the pipeline itself batches."""
    m = np.ascontiguousarray(np.asarray(matrix, dtype=np.float64))
    with _lib.Context(0) as ctx:
        ctx.set_contacts(m)
        ctx.p2_select(np.arange(len(m), dtype=np.int32))
        if len(m) < 2:
            return 0.0
        return float(ctx.p2_score(np.arange(len(m), dtype=np.int32)[None, :], float(total))[0])


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
    """(orders, orientations) over positions 0..k-1, cached."""
    if k not in _ENUM_CACHE:
        orders = removeReverseDuplicates(permutations(list(range(k)), [], 0))
        _ENUM_CACHE[k] = (orders, plusMinusPerms(list(range(k))))
    return _ENUM_CACHE[k]


def _positions(scaff, orderDict, orientation):
    """Selection positions of a scaffold's bins when it is laid down with ``orientation``."""
    pos = [orderDict[b] for b in scaff.binList]
    return pos if scaff.orientation == orientation else pos[::-1]


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
    fwd = {s.name: np.asarray(_positions(s, orderDict, "+"), dtype=np.int32) for s in sObjList}
    rev = {nm: p[::-1] for nm, p in fwd.items()}
    rows = []
    for o in orders:
        onames = [names[i] for i in o]
        for r in orients:
            rows.append(np.concatenate([fwd[nm] if sg == "+" else rev[nm] for nm, sg in zip(onames, r)]))
    best, best_c = matrix.first_strict_max(np.stack(rows), 0.)     # first strict maximum above 0. (OG:464)
    # the enumeration leaves every scaffold in the last candidate's orientation (OG:459)
    reorderScaffList([names[i] for i in orders[-1]], orients[-1], scaffDict)
    if best < 0:
        raise RuntimeError("no candidate order scored above 0 (the reference fails here too, OG:473 -> OG:576)")
    o, r = orders[best // len(orients)], orients[best % len(orients)]
    return [names[i] for i in o], list(r), best_c


def checkAllScores(adjMat: SubMatrix, orderDict, orderedScaffs, scaffToCheck):
    """OG:332-372: try the scaffold at every gap, both orientations.  The scaffold is flipped once
    per gap and stays flipped, so the orientation tried first alternates with the gap index."""
    gaps = len(orderedScaffs) + 1
    placed = [np.asarray([orderDict[b] for b in s.binList], dtype=np.int32) for s in orderedScaffs]
    cur = np.asarray([orderDict[b] for b in scaffToCheck.binList], dtype=np.int32)
    flip = {"+": "-", "-": "+"}
    o = scaffToCheck.orientation
    rows, tags = [], []
    for i in range(gaps):
        for _half in range(2):
            rows.append(np.concatenate(placed[:i] + [cur] + placed[i:]))
            tags.append((i, o))
            if _half == 0:
                cur, o = cur[::-1], flip[o]
    pick, bestCost = adjMat.first_strict_max(np.stack(rows), 0.)
    bestGap, bestOrient = tags[pick] if pick >= 0 else (0, "+")
    if gaps % 2 == 1:                                   # one flip per gap (OG:356)
        scaffToCheck.flipOrientation()
    if scaffToCheck.orientation != bestOrient:
        scaffToCheck.flipOrientation()
    orderedScaffs.insert(bestGap, scaffToCheck)
    return orderedScaffs, bestCost


def orderRemainderScaffolds(orderedScaffolds, scaffoldList, orderDict, matrix: GenomeMatrix, binList):
    """OG:475-493 (a do-while: with nothing left to add, the last ordered scaffold is re-inserted)."""
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
    adjMat.total()
    bestOrder = [s.name for s in orderedScaffolds]
    bestOrientation = [s.orientation for s in orderedScaffolds]
    roundNumber = 0
    w = scanScaffolds
    orders, orients = _enumeration(w)
    while True:
        improved = False
        print("Working on round " + str(roundNumber + 1) + " of final step...")
        for i in range(0, len(orderedScaffolds) - w + 1):
            window = orderedScaffolds[i:i + w]
            head = [np.asarray([orderDict[b] for b in s.binList], dtype=np.int32) for s in orderedScaffolds[:i]]
            tail = [np.asarray([orderDict[b] for b in s.binList], dtype=np.int32) for s in orderedScaffolds[i + w:]]
            head = np.concatenate(head) if head else np.zeros(0, np.int32)
            tail = np.concatenate(tail) if tail else np.zeros(0, np.int32)
            fwd = [np.asarray(_positions(s, orderDict, "+"), dtype=np.int32) for s in window]
            rev = [p[::-1] for p in fwd]
            rows = []
            for o in orders:
                for r in orients:
                    rows.append(np.concatenate([head] + [fwd[j] if sg == "+" else rev[j] for j, sg in zip(o, r)] + [tail]))
            pick, bestCost = adjMat.first_strict_max(np.stack(rows), bestCost)   # strict '>' vs the global best (OG:535)
            if pick >= 0:
                improved = True
                o, r = orders[pick // len(orients)], orients[pick % len(orients)]
                names = [s.name for s in orderedScaffolds]
                outside = {s.name: s.orientation for s in orderedScaffolds}
                bestOrder = names[:i] + [window[j].name for j in o] + names[i + w:]
                bestOrientation = ([outside[nm] for nm in names[:i]] + list(r) + [outside[nm] for nm in names[i + w:]])
            orderedScaffolds, _nodes = reorderScaffList(bestOrder, bestOrientation, scaffoldDict)
        roundNumber += 1
        if not improved:
            break
    print("Sliding window conversion after " + str(roundNumber) + " rounds")
    print("Best cost at the end of the final step = " + str(bestCost))
    return orderedScaffolds, bestCost


def orderChromosome(chromGroup, matrix: GenomeMatrix, binList, nScaffolds=6, scanScaffolds=5):
    """OG:551-586."""
    if nScaffolds >= 9:
        print("Number of initial scaffolds to order by brute force method is set too high... setting it to 8")
        nScaffolds = 8
    if scanScaffolds > nScaffolds:
        scanScaffolds = nScaffolds
    scaffoldList, scaffoldDict = initiateBinsAndScaffolds(chromGroup)
    orderedScaffolds, scaffoldList = pullScaffolds([], scaffoldList, nScaffolds)
    adjMat, orderDict = giveNewAdjMat(matrix, orderedScaffolds, binList)
    bfOrder, bfOrient, _bfScore = bruteForceBestScore(orderedScaffolds, scaffoldDict, adjMat, orderDict)
    orderedScaffolds, _nodes = reorderScaffList(bfOrder, bfOrient, scaffoldDict)
    orderedScaffolds, bestCost = orderRemainderScaffolds(orderedScaffolds, scaffoldList, orderDict, matrix, binList)
    print("BestCost at the end of first two steps " + str(bestCost))
    if len(orderedScaffolds) > nScaffolds:
        orderedScaffolds, bestCost = scanOrdering(orderedScaffolds, scaffoldDict, orderDict, matrix, binList,
                                                  bestCost, scanScaffolds=scanScaffolds)
    print("Final ordering...")
    for s in orderedScaffolds:
        print(s.name, s.orientation)
    orderChromosome.last_cost = bestCost
    return orderedScaffolds


def orderGenome(matrix: GenomeMatrix, chromList, binList, resolution, nScaffolds=6, scanScaffolds=5, plotChrom=True,
                showPlot=True, savePlotDir=False, plotTitleSuffix=False):
    """OG:591-628 (chromosomes are independent; plots are not produced)."""
    t0 = time.time()
    fullGenomeOrder = []
    for i, chromGroup in enumerate(chromList):
        print("#####################\n#####################")
        print("Working on Chr_" + str(i + 1) + "...")
        fullGenomeOrder.append(orderChromosome(chromGroup, matrix, binList, nScaffolds=nScaffolds,
                                               scanScaffolds=scanScaffolds))
    print("RunTime for total genome = " + str(time.time() - t0))
    return fullGenomeOrder


def writeScaffoldOrderingsToFile(sOrderings, outFile):
    """OG:630-644."""
    written = 0
    with open(outFile, "w") as fh:
        for k, group in enumerate(sOrderings):
            fh.write("### Chromosome grouping " + str(k + 1) + " ###\n")
            for s in group:
                fh.write(s.name + "\t" + s.orientation + "\n")
                written += 1
    print("Chromosome groups written to file " + str(len(sOrderings)))
    print("Scaffolds written to file " + str(written))


def writeBinIDsOrderingToFile(scaffoldList, outFile):
    """OG:646-660: header line, then newline-PREFIXED rows (no trailing newline)."""
    written = 0
    with open(outFile, "w") as fh:
        fh.write("#ScaffoldID\tHiCPro-BinID")
        for s in scaffoldList:
            for b in s.binList:
                fh.write("\n" + s.name + "\t" + str(b))
                written += 1
    print("BinIDs written to file " + str(written))


def getChromosomeOutlineCoords(orderedChromosomes):
    """OG:662-674."""
    coords, index = [], 0
    for group in orderedChromosomes:
        index += sum(len(s.binList) for s in group)
        coords.append(index)
    return coords


def runPipeline(hicProBedFile, hicProBiasFile, hicProMatrixFile, chromosomeGroupFile, chromosomeOrderFile,
                savePlotsDirectory, chromosomePlotSuffix, fullGenomePlot, fullGenomePlotTitle, plotOrderFile,
                nScaffolds, scanScaffolds, resolution, device=0):
    """OG:679-712, same positional arguments (``device`` is an optional extra)."""
    print("########################################")
    print("### Working on Part2 of the pipeline ###")
    t0 = time.time()
    binDict = readGroupingsToValidBins(chromosomeGroupFile)
    binList = initiateLoci(hicProBedFile, hicProBiasFile, binID_dict=binDict)
    adjMat = buildAdjacencyMatrix(hicProMatrixFile, binList, device=device)
    try:
        runResident(adjMat, binList, chromosomeGroupFile, chromosomeOrderFile, plotOrderFile, nScaffolds,
                    scanScaffolds, resolution)
    finally:
        adjMat.ctx.close()
    print("- plotting is not part of the MI355X hot path: " + str(fullGenomePlot) + " not written")
    print("Total run-time  for Part2 = " + str(time.time() - t0))
    print("- Part 2 (chromosome ordering) completed successfully")


def runResident(adjMat: GenomeMatrix, binList, chromosomeGroupFile, chromosomeOrderFile, plotOrderFile,
                nScaffolds, scanScaffolds, resolution):
    """OG:691-709 on contacts that are already resident in HBM (what bench.py times).  ``binList``
    gives the bin of every row of the device matrix; bins that Part 1 did not assign to a group are
    simply never selected, which is what the reference's re-load restricted to grouped bins
    (OG:688-690) amounts to."""
    chromosomeList = readChromsFromFile(chromosomeGroupFile)
    orderedChromosomes = orderGenome(adjMat, chromosomeList, binList, resolution, nScaffolds=nScaffolds,
                                     scanScaffolds=scanScaffolds, plotChrom=True, showPlot=False)
    writeScaffoldOrderingsToFile(orderedChromosomes, chromosomeOrderFile)
    writeBinIDsOrderingToFile([s for group in orderedChromosomes for s in group], plotOrderFile)
    return orderedChromosomes
