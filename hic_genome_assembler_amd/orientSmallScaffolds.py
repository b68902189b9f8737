"""Part 3 - orientation of scaffolds no longer than one bin from read pairs (orientSmallScaffolds.py;
SURVEY.md section 8f, N4).  Same function names, arguments, files and results as the reference module.

A one-bin scaffold has no orientation signal in the contact map.  The reference looks at the read pairs that
join it to its neighbours in the Part 2 order: pairs landing within ``lengthCutoff`` of the neighbour's facing
end vote, normalised by the restriction sites available on both sides (OSS:179-356).  The only heavy step is
``readValidPairFile`` - a Python loop over a multi-GB HiC-Pro allValidPairs file that keeps the lines naming one
of a few thousand registered scaffold pairs (OSS:159-177).  Here that scan is libhicmi's multi-threaded mmap
parser (``hicmi_scan_valid_pairs``, host code) and the votes are counted on the coordinate arrays with NumPy.
OSS = /root/reference/HIC_ASSEMBLER/orientSmallScaffolds.py.
"""
from __future__ import annotations

import math
import time

import numpy as np

from . import _lib


class RestrictionScaffold:
    """OSS:7-32: a scaffold of the Part 2 order with its size and restriction-site coordinates."""

    def __init__(self, name, orientation, size, resCoords, binCount):
        self.name = name
        self.orientation = orientation
        self.size = size
        self.resCoords = resCoords

    def getBinCount(self, resolution):
        self.binCount = math.ceil(float(self.size) / float(resolution))

    def getResCounts(self, lengthCutoff):
        """Sites within ``lengthCutoff`` of the left end; of the others, those within it of the right end; a
        side without sites counts as one (OSS:17-31)."""
        coords = np.asarray(self.resCoords)
        near_left = coords <= lengthCutoff
        left = int(near_left.sum())
        right = int((~near_left & (coords > (self.size - lengthCutoff))).sum())
        self.resLeft = left if left else 1
        self.resRight = right if right else 1


class PairList:
    """The read pairs kept for one ordered scaffold pair: behaves like the reference's list of
    ``[scaffold1, scaffold2, pos1, pos2]`` rows (len, iteration, indexing) and exposes the two position columns
    as arrays for counting."""

    def __init__(self, key):
        self.key = key
        self.pos = np.zeros((0, 2), dtype=np.int64)

    def __len__(self):
        return len(self.pos)

    def __getitem__(self, i):
        return [self.key[0], self.key[1], int(self.pos[i, 0]), int(self.pos[i, 1])]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def column_of(self, name, other):
        """Positions on scaffold ``name`` (the pair is (name, other) or (other, name))."""
        return self.pos[:, 0] if self.key == (name, other) else self.pos[:, 1]


def readPreliminaryOrientationFromFile(orientationFile):
    """OSS:34-56: the Part 2 order file -> scaffold objects per chromosome and by name (the first line is
    skipped unconditionally, every later '#' line opens a new chromosome)."""
    groups, current, by_name = [], [], {}
    with open(orientationFile) as fh:
        fh.readline()
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] == "#":
                groups.append(current)
                current = []
                continue
            cols = line.split('\t')
            scaffold = RestrictionScaffold(cols[0], cols[1], 0., [], 0)
            by_name[cols[0]] = scaffold
            current.append(scaffold)
    groups.append(current)
    return groups, by_name


def readScaffSizeFile(scaffSizeFile, scaffDict, resolution):
    """OSS:58-73."""
    with open(scaffSizeFile) as fh:
        for line in fh:
            cols = line.strip('\r').strip('\n').split('\t')
            if cols[0] in scaffDict:
                scaffDict[cols[0]].size = float(cols[1])
                scaffDict[cols[0]].getBinCount(resolution)
    return scaffDict


def readRestrictionsFile(restrictionFile, scaffDict):
    """OSS:75-92: column 2 of the digest file, sorted per scaffold."""
    with open(restrictionFile) as fh:
        for line in fh:
            cols = line.strip('\r').strip('\n').split('\t')
            name, coord = cols[0], int(cols[2])
            if name in scaffDict:
                scaffDict[name].resCoords.append(coord)
    for scaffold in scaffDict.values():
        scaffold.resCoords = sorted(scaffold.resCoords)
    return scaffDict


def initiateScaffoldObjects(orientationFile, scaffSizeFile, restrictionFile, resolution):
    """OSS:94-107."""
    groups, by_name = readPreliminaryOrientationFromFile(orientationFile)
    by_name = readScaffSizeFile(scaffSizeFile, by_name, resolution)
    by_name = readRestrictionsFile(restrictionFile, by_name)
    return groups, by_name


def pullTriplets(scaffoldList):
    """OSS:109-137: every one-bin scaffold with the neighbours it has (3 scaffolds, or 2 at a chromosome end;
    a chromosome consisting of that scaffold alone yields nothing)."""
    triplets = []
    last = len(scaffoldList) - 1
    for i, scaffold in enumerate(scaffoldList):
        if scaffold.binCount != 1:
            continue
        before = scaffoldList[i - 1:i] if i > 0 else []
        after = scaffoldList[i + 1:i + 2] if i < last else []
        if before or after:
            triplets.append(before + [scaffold] + after)
    return triplets


def produceReadPairKeys(allChromosomeTriplets):
    """OSS:139-157: both orders of every neighbouring pair of a triplet."""
    keys = {}
    for chromosomeTriplets in allChromosomeTriplets:
        for triplet in chromosomeTriplets:
            for a, b in zip(triplet, triplet[1:]):
                keys[(a.name, b.name)] = PairList((a.name, b.name))
                keys[(b.name, a.name)] = PairList((b.name, a.name))
    return keys


def readValidPairFile(pairFile, pairDict, threads: int = 0):
    """OSS:159-177 by hicmi_scan_valid_pairs: the positions of every line whose (scaffold1, scaffold2) is a
    registered key, in file order."""
    keys = list(pairDict)
    names = sorted({n for key in keys for n in key})
    index = {n: i for i, n in enumerate(names)}
    idx, p1, p2, n_lines = _lib.scan_valid_pairs(pairFile, names, [(index[a], index[b]) for a, b in keys], threads)
    order = np.argsort(idx, kind="stable")                      # group by key, file order kept inside a key
    bounds = np.searchsorted(idx[order], np.arange(len(keys) + 1))
    for k, key in enumerate(keys):
        rows = order[bounds[k]:bounds[k + 1]]
        plist = pairDict[key] if isinstance(pairDict[key], PairList) else PairList(key)
        plist.pos = np.stack([p1[rows], p2[rows]], axis=1) if len(rows) else np.zeros((0, 2), dtype=np.int64)
        pairDict[key] = plist
    for shown in range(10000000, n_lines + 1, 10000000):
        print("Read pairs looked at " + str(shown) + "...")
    return pairDict


def _pairs_between(pairDict, first, second):
    """The reference looks at (first, second) and only if that list is empty at (second, first) (OSS:192-199)."""
    for key in ((first.name, second.name), (second.name, first.name)):
        if len(pairDict[key]) != 0:
            return pairDict[key]
    return None


def _near_end(coords, scaffold, at_start, lengthCutoff):
    """Positions within ``lengthCutoff`` of the scaffold's start (``at_start``) or of its end."""
    return coords <= lengthCutoff if at_start else (scaffold.size - coords) <= lengthCutoff


def orientTrueTriplet(triplet, pairDict, lengthCutoff):
    """OSS:179-240: the middle scaffold is '+' unless the votes towards the left neighbour outweigh those
    towards the right one.  (Both normalisations use the middle scaffold's RIGHT site count, as the reference
    does.)"""
    for s in triplet:
        s.getResCounts(lengthCutoff)
    s0, s1, s2 = triplet
    p, m = 0, 0
    pairs = _pairs_between(pairDict, s1, s2)
    if pairs is not None:
        facing_start = s2.orientation == "+"
        votes = int(_near_end(pairs.column_of(s2.name, s1.name), s2, facing_start, lengthCutoff).sum())
        p = float(votes) / float(s1.resRight + (s2.resLeft if facing_start else s2.resRight))
    pairs = _pairs_between(pairDict, s1, s0)
    if pairs is not None:
        facing_start = s0.orientation == "-"
        votes = int(_near_end(pairs.column_of(s0.name, s1.name), s0, facing_start, lengthCutoff).sum())
        m = float(votes) / float(s1.resRight + (s0.resLeft if facing_start else s0.resRight))
    return s1.name, "+" if p >= m else "-"


def orientLeftEdgeCase(scaffLeft, scaffRight, pairDict, lengthCutoff):
    """OSS:242-288: first scaffold of a chromosome; its two halves compete for the links to the neighbour."""
    scaffLeft.getResCounts(float(scaffLeft.size / 2.))
    scaffRight.getResCounts(lengthCutoff)
    right_plus = scaffRight.orientation == "+"
    window = (0, lengthCutoff) if right_plus else (scaffRight.size - lengthCutoff, scaffRight.size)
    # the lookup order of the reference is (left, right) then (right, left)
    pairs = None
    for key in ((scaffLeft.name, scaffRight.name), (scaffRight.name, scaffLeft.name)):
        if len(pairDict[key]) != 0:
            pairs = pairDict[key]
            break
    p = m = 0
    if pairs is not None:
        left_pos = pairs.column_of(scaffLeft.name, scaffRight.name)
        right_pos = pairs.column_of(scaffRight.name, scaffLeft.name)
        in_window = (window[0] <= right_pos) & (right_pos <= window[1])
        upper_half = left_pos >= float(scaffLeft.size / 2.)
        p, m = int((upper_half & in_window).sum()), int((~upper_half & in_window).sum())
    right_sites = scaffRight.resLeft if right_plus else scaffRight.resRight
    p = float(p) / float(scaffLeft.resRight + right_sites)
    m = float(m) / float(scaffLeft.resLeft + right_sites)
    return scaffLeft.name, "+" if p >= m else "-"


def orientRightEdgeCase(scaffLeft, scaffRight, pairDict, lengthCutoff):
    """OSS:290-336: last scaffold of a chromosome."""
    scaffLeft.getResCounts(lengthCutoff)
    scaffRight.getResCounts(float(scaffRight.size / 2.))
    left_plus = scaffLeft.orientation == "+"
    window = (scaffLeft.size - lengthCutoff, scaffLeft.size) if left_plus else (0, lengthCutoff)
    pairs = None
    for key in ((scaffLeft.name, scaffRight.name), (scaffRight.name, scaffLeft.name)):
        if len(pairDict[key]) != 0:
            pairs = pairDict[key]
            break
    p = m = 0
    if pairs is not None:
        left_pos = pairs.column_of(scaffLeft.name, scaffRight.name)
        right_pos = pairs.column_of(scaffRight.name, scaffLeft.name)
        in_window = (window[0] <= left_pos) & (left_pos <= window[1])
        lower_half = right_pos < float(scaffRight.size / 2.)
        p, m = int((lower_half & in_window).sum()), int((~lower_half & in_window).sum())
    left_sites = scaffLeft.resRight if left_plus else scaffLeft.resLeft
    p = float(p) / float(left_sites + scaffRight.resLeft)
    m = float(m) / float(left_sites + scaffRight.resRight)
    return scaffRight.name, "+" if p >= m else "-"


def orientTriplet(triplet, scaffList, pairDict, lengthCutoff):
    """OSS:338-368."""
    if len(triplet) == 3:
        return orientTrueTriplet(triplet, pairDict, lengthCutoff)
    s0, s1 = triplet
    if s0.name == scaffList[0].name:
        return orientLeftEdgeCase(s0, s1, pairDict, lengthCutoff)
    return orientRightEdgeCase(s0, s1, pairDict, lengthCutoff)


def giveFinalChromOrdering(trips, scaffGroups, scaffDict, validPairs, resolution, lengthCutoff=500000):
    """OSS:370-394: triplets are decided one after the other; a decision is visible to the next triplet."""
    if lengthCutoff < resolution:
        print("lengthCutoff variable is set too low... Setting equal to resolution variable")
        lengthCutoff = resolution
    orders = []
    for chromosomeTriplets, chromosomeScaffs in zip(trips, scaffGroups):
        for trip in chromosomeTriplets:
            name, orientation = orientTriplet(trip, chromosomeScaffs, validPairs, lengthCutoff=lengthCutoff)
            scaffDict[name].orientation = orientation
        orders.append([[s.name, s.orientation] for s in chromosomeScaffs])
    return orders


def writeScaffoldOrderingsToFile(sOrderings, outFile):
    """OSS:396-416."""
    written = 0
    with open(outFile, 'w') as fh:
        for count, group in enumerate(sOrderings, 1):
            fh.write("### Chromosome grouping " + str(count) + " ###" + '\n')
            for s in group:
                if isinstance(s, RestrictionScaffold):
                    fh.write(s.name + '\t' + s.orientation + '\n')
                elif isinstance(s, list):
                    fh.write(s[0] + '\t' + s[1] + '\n')
                else:
                    print("- WARNING invalid output type {}... Expecting list or Scaffold like class...".format(type(s)))
                written += 1
    print("Chromosome groups written to file " + str(len(sOrderings)))
    print("Scaffolds written to file " + str(written))


def runPipeline(chromosomeOrderFile, scaffSizeFile, restrictionSiteFile, validPairFile, finalOrderingFile, lengthCutoff,
                resolution):
    """OSS:420-433."""
    print("########################################")
    print("### Working on Part3 of the pipeline ###")
    startTime = time.time()
    scaffGroups, scaffDict = initiateScaffoldObjects(chromosomeOrderFile, scaffSizeFile, restrictionSiteFile, resolution)
    trips = [pullTriplets(group) for group in scaffGroups]
    validPairs = produceReadPairKeys(trips)
    validPairs = readValidPairFile(validPairFile, validPairs)
    finalChromGroups = giveFinalChromOrdering(trips, scaffGroups, scaffDict, validPairs, resolution=resolution,
                                              lengthCutoff=lengthCutoff)
    writeScaffoldOrderingsToFile(finalChromGroups, finalOrderingFile)
    print("Total run-time  for Part3 = " + str(time.time() - startTime))
    print("- Part 3 (optional orientation of scaffolds smaller than resulution size) completed successfully")
