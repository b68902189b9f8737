"""Part 4 - write the assembled FASTA (writeAssembledFasta.py).  Same function names, arguments and output
bytes as the reference module; WAF = /root/reference/HIC_ASSEMBLER/writeAssembledFasta.py.

Host-side text work only.  The reference reverse-complements with a per-character dictionary look-up and joins
per-character lists (WAF:57-66, 105-108); here sequences stay whole strings, the complement is one
``str.translate`` and a chromosome is written in slices, which is what makes a 3 Gb genome a matter of seconds.
"""
from __future__ import annotations

import gzip
import time

_COMPLEMENT = str.maketrans("ATatGCgcNn", "TAtaCGcgNn")
_KNOWN = frozenset("ATatGCgcNn")


def readFastaIntoMem(fastaFile):
    """WAF:10-33: {entry name: sequence}; gzip when the name contains ".gz"."""
    opener = (lambda p: gzip.open(p, mode='rt')) if ".gz" in fastaFile else (lambda p: open(p, 'r'))
    entries, name, parts = {}, None, None
    with opener(fastaFile) as fh:
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] == '>':
                if name is not None:
                    entries[name] = ''.join(parts)
                name, parts = line[1:], []
                entries[name] = ''
            else:
                parts.append(line)
    if name is not None:
        entries[name] = ''.join(parts)
    return entries


def readChromosomeOrderingFile(chrOrderFile):
    """WAF:35-55: [[scaffold, orientation], ...] per chromosome (first line skipped, '#' opens the next group)."""
    groups, current = [], []
    with open(chrOrderFile) as fh:
        fh.readline()
        for line in fh:
            line = line.strip('\r').strip('\n')
            if line[0] != "#":
                cols = line.split('\t')
                current.append([cols[0], cols[1]])
            else:
                groups.append(current)
                current = []
    groups.append(current)
    return groups


def reverseTranscribeSeq(seq):
    """WAF:57-66: reverse complement over A/C/G/T/N in either case; any other letter is a KeyError, as in the
    reference's dictionary look-up."""
    unknown = set(seq) - _KNOWN
    if unknown:
        raise KeyError(seq[max(seq.rfind(ch) for ch in unknown)])      # the first one the reversed walk meets
    return seq[::-1].translate(_COMPLEMENT)


def writeSeqToFile(fileMan, seq, charsPerLine=50):
    """WAF:68-77."""
    fileMan.write(''.join(seq[i:i + charsPerLine] + '\n' for i in range(0, len(seq), charsPerLine)))
    return fileMan


def writeNewFasta(chrGroups, oldFastaDict, outFile, charsPerLine=50, nGapLength=100):
    """WAF:79-127: one entry ``Chr_i`` per group, scaffolds joined by ``nGapLength`` N, then every entry that is in
    no group, unchanged; prints the reference's assembly statistics."""
    groupedLength, ungroupedLength = 0, 0
    newNsWritten, gaps = 0, 0
    scaffoldsGrouped, ungrouped = 0, 0
    written = set()
    gap = "N" * nGapLength
    with open(outFile, 'w') as out:
        for i, group in enumerate(chrGroups, 1):
            out.write(">Chr_" + str(i) + '\n')
            pieces = []
            for k, (name, orientation) in enumerate(group):
                scaffoldsGrouped += 1
                written.add(name)
                pieces.append(oldFastaDict[name] if orientation == "+" else reverseTranscribeSeq(oldFastaDict[name]))
                if k != len(group) - 1:
                    newNsWritten += nGapLength
                    gaps += 1
                    pieces.append(gap)
            seq = ''.join(pieces)
            groupedLength += len(seq)
            writeSeqToFile(out, seq, charsPerLine=charsPerLine)
        for name in oldFastaDict:
            if name not in written:
                out.write(">" + name + '\n')
                seq = oldFastaDict[name]
                ungroupedLength += len(seq)
                ungrouped += 1
                writeSeqToFile(out, seq, charsPerLine=charsPerLine)
    print("Total scaffolds grouped into chromosomes" + '\t' + str(scaffoldsGrouped))
    print("Total genome length grouped into chromosomes" + '\t' + str(groupedLength - newNsWritten))
    print("Total new gaps introduced" + '\t' + str(gaps))
    print("Total ungrouped scaffolds" + '\t' + str(ungrouped))
    print("Total genome length ungrouped " + '\t' + str(ungroupedLength))


def runPipeline(originalFastaFile, finalOrderingFile, assembledFastaFile):
    """WAF:131-142."""
    print("########################################")
    print("### Working on Part4 of the pipeline ###")
    startTime = time.time()
    fastaDict = readFastaIntoMem(originalFastaFile)
    chrGroups = readChromosomeOrderingFile(finalOrderingFile)
    writeNewFasta(chrGroups, fastaDict, assembledFastaFile, charsPerLine=50, nGapLength=100)
    print("Total run-time  for Part4 = " + str(time.time() - startTime))
    print("- Part 4 (writing of new super-scaffolded genome .fasta) completed successfully")
