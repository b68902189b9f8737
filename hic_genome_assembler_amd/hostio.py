"""HiC-Pro text inputs -> host arrays (reference loaders: scaffoldToChromosomes.py:24-98,
duplicated in orderGenome.py:19-93).

The reference fills a Python list-of-lists cell by cell (~32 bytes per cell and minutes of
interpreter time at N >= 16k).  Here the triplet file goes straight into one dense fp64 array that is
uploaded to HBM once: by libhicmi's multi-threaded mmap parser (csrc/loader.hip, the default), or by
pandas' C tokenizer with ``float_precision='round_trip'`` (an independent second implementation) - both
apply the correctly rounded conversion of Python's ``float()``, which the reference uses at S2C:83.
``read_contact_matrix_cached`` adds a binary cache of the parsed matrix; ``paused_gc`` is the helper the
host stages run under.
"""
from __future__ import annotations

import contextlib
import gc
import hashlib
import json
import os

import numpy as np


@contextlib.contextmanager
def paused_gc():
    """The host stages build tens of thousands of small containers (bins, lines, groups) and no reference
    cycles; a generation-2 pass of the cyclic collector in the middle of a stage costs ~50 ms at 16k bins.
    Reference counting still frees everything; the collector is switched back on afterwards."""
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


class Bin:
    """One genomic bin as read from HiC-Pro (S2C:24-33)."""
    __slots__ = ("ID", "chrom", "start", "stop", "bias", "rowSum")

    def __init__(self, ID, chrom, start, stop, bias, rowSum):
        self.ID = ID
        self.chrom = chrom
        self.start = start
        self.stop = stop
        self.bias = bias
        self.rowSum = rowSum


def initiateLoci(bedFile, biasFile, binID_dict=False):
    """S2C:35-68 / OG:30-63: the .bed and .biases files are read line-aligned; a bias line that is
    exactly "nan" drops the bin, one that does not parse becomes 0.0; with ``binID_dict`` only the
    listed bin IDs are kept."""
    bins = []
    with open(bedFile) as bed, open(biasFile) as bias:
        for bed_line in bed:
            bias_line = bias.readline()
            cols = bed_line.strip("\r").strip("\n").split("\t")
            bid = int(cols[3])
            if binID_dict is not False and bid not in binID_dict:
                continue
            text = bias_line.strip("\r").strip("\n")
            if text == "nan":
                continue
            try:
                value = float(text)
            except Exception:
                value = 0.
            bins.append(Bin(bid, cols[0], int(cols[1]), int(cols[2]), value, 0.))
    print("Genomic loci found" + "\t" + str(len(bins)))
    return bins


def _cache_paths(matrixFile, cache):
    """Where the binary copy of a parsed matrix lives: next to the text file (cache=True / "1") or in
    the directory given."""
    if cache in (True, "1", "true", "yes"):
        base = matrixFile
    else:
        base = os.path.join(str(cache), os.path.basename(matrixFile))
    return base + ".hicmi.npy", base + ".hicmi.json"


def _cache_key(matrixFile, ids):
    st = os.stat(matrixFile)
    return {"source_bytes": int(st.st_size), "source_mtime_ns": int(st.st_mtime_ns), "bins": int(len(ids)),
            "bin_ids_sha1": hashlib.sha1(np.ascontiguousarray(ids, dtype=np.int64).tobytes()).hexdigest(),
            "format": 1}


def read_contact_matrix_cached(matrixFile, binList, cache, engine: str = "native") -> np.ndarray:
    """read_contact_matrix with a binary cache: the dense fp64 array is stored as
    .npy beside a small JSON key (size and mtime of the text file, number and SHA-1 of the bin IDs in
    order); a matching key means the text is not parsed again and the array is memory-mapped read-only.
    A stale or unreadable cache is ignored and rewritten."""
    ids = np.fromiter((b.ID for b in binList), dtype=np.int64, count=len(binList))
    npy, key_file = _cache_paths(matrixFile, cache)
    key = _cache_key(matrixFile, ids)
    try:
        with open(key_file) as fh:
            if json.load(fh) == key:
                mat = np.load(npy, mmap_mode="r", allow_pickle=False)
                if mat.shape == (len(ids), len(ids)) and mat.dtype == np.float64:
                    print("Adjacency matrix read from binary cache " + npy)
                    return mat
    except (OSError, ValueError):
        pass
    mat = read_contact_matrix(matrixFile, binList, engine=engine)
    try:
        os.makedirs(os.path.dirname(os.path.abspath(npy)), exist_ok=True)
        np.save(npy, mat, allow_pickle=False)
        with open(key_file, "w") as fh:
            json.dump(key, fh)
    except OSError as exc:                                  # a read-only input directory must not fail the run
        print("WARNING - could not write the matrix cache: " + str(exc))
    return mat


def read_contact_matrix(matrixFile, binList, chunk_lines: int = 4_000_000, engine: str = "native") -> np.ndarray:
    """S2C:70-98: ``id1<TAB>id2<TAB>value`` triplets into a dense symmetric fp64 array in ``binList``
    order.  Triplets naming an unknown bin are skipped; each one sets [i][j] and [j][i]; when a
    cell is named twice the later line wins, exactly as sequential assignment would.

    ``engine="native"`` (default) is libhicmi's multi-threaded mmap + from_chars parser
    (csrc/loader.hip); ``engine="pandas"`` is an independent second implementation kept for
    cross-checking."""
    if engine == "native":
        from . import _lib
        ids = np.fromiter((b.ID for b in binList), dtype=np.int64, count=len(binList))
        mat, edges = _lib.load_hicpro_matrix(matrixFile, ids)
        print("Edges added to adjacency matrix" + "\t" + str(edges))
        return mat
    import pandas as pd

    n = len(binList)
    ids = np.fromiter((b.ID for b in binList), dtype=np.int64, count=n)
    max_id = int(ids.max()) if n else 0
    lookup = np.full(max_id + 2, -1, dtype=np.int64)
    if n:
        if ids.min() < 0:
            raise ValueError("negative bin ID")
        lookup[ids] = np.arange(n)
    mat = np.zeros((n, n), dtype=np.float64)
    edges = 0
    reader = pd.read_csv(matrixFile, sep="\t", header=None, names=["a", "b", "v"],
                         dtype={"a": np.int64, "b": np.int64, "v": np.float64},
                         float_precision="round_trip", chunksize=chunk_lines, engine="c")
    for chunk in reader:
        a = chunk["a"].to_numpy()
        b = chunk["b"].to_numpy()
        v = chunk["v"].to_numpy()
        ok = (a >= 0) & (a <= max_id) & (b >= 0) & (b <= max_id)
        ia = np.where(ok, lookup[np.where(ok, a, 0)], -1)
        ib = np.where(ok, lookup[np.where(ok, b, 0)], -1)
        ok = (ia >= 0) & (ib >= 0)
        ia, ib, v = ia[ok], ib[ok], v[ok]
        # interleave (i,j),(j,i) per line so that "last assignment wins" follows file order
        rows = np.empty(2 * len(ia), dtype=np.int64)
        cols = np.empty(2 * len(ia), dtype=np.int64)
        rows[0::2], rows[1::2] = ia, ib
        cols[0::2], cols[1::2] = ib, ia
        mat[rows, cols] = np.repeat(v, 2)
        edges += len(ia)
    print("Edges added to adjacency matrix" + "\t" + str(edges))
    return mat
