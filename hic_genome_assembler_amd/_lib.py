"""ctypes binding of libhicmi.so (include/hicmi.h) - the only way the Python host reaches the GPU.

There is deliberately no fallback: if the library has not been built, or no HIP device is
visible, the calls raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C hic_genome_assembler_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HICMI_LIB") or os.path.join(_HERE, "libhicmi.so")     # HICMI_LIB: another build (A/B runs)
_lib = None

c_i64 = ctypes.c_int64
c_dbl = ctypes.c_double
_vp = ctypes.c_void_p

# name: (restype, argtypes) - one row per declaration in include/hicmi.h
SIGNATURES = {
    "hicmi_abi_version": (ctypes.c_int, []),
    "hicmi_last_error": (ctypes.c_char_p, []),
    "hicmi_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "hicmi_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "hicmi_destroy": (ctypes.c_int, [_vp]),
    "hicmi_stream": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "hicmi_synchronize": (ctypes.c_int, [_vp]),
    "hicmi_set_contacts_host": (ctypes.c_int, [_vp, _vp, c_i64]),
    "hicmi_set_contacts_host_f32": (ctypes.c_int, [_vp, _vp, c_i64]),
    "hicmi_set_contacts_device": (ctypes.c_int, [_vp, _vp, c_i64, c_i64]),
    "hicmi_contacts_device": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "hicmi_load_hicpro_matrix": (ctypes.c_int, [ctypes.c_char_p, _vp, c_i64, _vp, ctypes.c_int, ctypes.POINTER(c_i64)]),
    "hicmi_row_sums": (ctypes.c_int, [_vp, _vp, _vp]),
    "hicmi_set_row_shard": (ctypes.c_int, [_vp, c_i64, c_i64]),
    "hicmi_set_row_sums": (ctypes.c_int, [_vp, _vp, _vp]),
    "hicmi_compact": (ctypes.c_int, [_vp, _vp, c_i64]),
    "hicmi_upgma": (ctypes.c_int, [_vp, _vp, _vp]),
    "hicmi_rank_matrix": (ctypes.c_int, [_vp, _vp]),
    "hicmi_presort_state": (ctypes.c_int, [_vp, _vp, _vp]),
    "hicmi_get_rank_rows": (ctypes.c_int, [_vp, c_i64, c_i64, ctypes.c_int, _vp]),
    "hicmi_get_similarity_row": (ctypes.c_int, [_vp, c_i64, _vp]),
    "hicmi_cut_scan": (ctypes.c_int, [_vp, c_i64, c_i64, c_dbl, _vp, _vp]),
    "hicmi_filter_scan": (ctypes.c_int, [_vp, c_i64, c_i64, c_i64, c_i64, c_dbl, _vp, _vp]),
    "hicmi_first_pass_cuts": (ctypes.c_int, [_vp, c_i64, c_i64, c_dbl, _vp, c_i64, _vp, _vp, c_i64, _vp]),
    "hicmi_filter_cuts": (ctypes.c_int, [_vp, _vp, c_i64, c_dbl, _vp, c_i64, _vp, _vp]),
    "hicmi_hypergeom_decide": (ctypes.c_int, [c_i64, c_i64, c_i64, c_i64, c_dbl]),
    "hicmi_hypergeom_sf": (c_dbl, [c_i64, c_i64, c_i64, c_i64]),
    "hicmi_selftest_division": (ctypes.c_int, [_vp, ctypes.c_uint64, c_i64, ctypes.POINTER(ctypes.c_uint64)]),
    "hicmi_label_linkage": (ctypes.c_int, [_vp, c_i64, _vp]),
    "hicmi_leaf_order": (ctypes.c_int, [_vp, c_i64, _vp]),
    "hicmi_get_raw_merges": (ctypes.c_int, [_vp, _vp]),
    "hicmi_nnchain_stats": (ctypes.c_int, [_vp, _vp]),
    "hicmi_p2_select": (ctypes.c_int, [_vp, _vp, c_i64]),
    "hicmi_p2_total": (ctypes.c_int, [_vp, ctypes.POINTER(c_dbl)]),
    "hicmi_p2_score": (ctypes.c_int, [_vp, _vp, c_i64, c_i64, c_dbl, _vp]),
    "hicmi_p2_score_exact": (ctypes.c_int, [_vp, _vp, c_i64, c_i64, c_dbl, _vp]),
    "hicmi_p2_layout": (ctypes.c_int, [_vp, _vp, _vp, c_i64]),
    "hicmi_p2_set_arrangement": (ctypes.c_int, [_vp, _vp, _vp, c_i64]),
    "hicmi_p2_arrangement_total": (ctypes.c_int, [_vp, ctypes.POINTER(c_dbl)]),
    "hicmi_p2_arrangement_score": (ctypes.c_int, [_vp, c_dbl, ctypes.POINTER(c_dbl)]),
    "hicmi_p2_score_insertions": (ctypes.c_int, [_vp, ctypes.c_int32, c_dbl, _vp]),
    "hicmi_p2_window_tables": (ctypes.c_int, [_vp, c_i64, _vp, c_i64, _vp, c_i64]),
    "hicmi_p2_score_window": (ctypes.c_int, [_vp, c_i64, c_i64, _vp]),
    "hicmi_p2_decide_window": (ctypes.c_int, [_vp, c_i64, c_i64, c_dbl, c_dbl, c_dbl, ctypes.POINTER(c_i64),
                                              ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]),
    "hicmi_p2_decide_insertion": (ctypes.c_int, [_vp, _vp, _vp, c_i64, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_int32),
                                                 ctypes.POINTER(c_dbl)]),
    "hicmi_p2_insert_all": (ctypes.c_int, [_vp, _vp, _vp, c_i64, _vp, c_i64, ctypes.POINTER(c_dbl)]),
    "hicmi_scan_valid_pairs": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, _vp, c_i64, _vp, _vp, c_i64, ctypes.c_int,
                                              ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(_vp)]),
    "hicmi_scan_fetch": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "hicmi_plot_percentiles": (ctypes.c_int, [_vp, ctypes.c_int, _vp, c_i64, _vp, c_i64, _vp]),
    "hicmi_plot_downsample": (ctypes.c_int, [_vp, ctypes.c_int, _vp, c_i64, c_i64, _vp]),
    "hicmi_p2_insert_all_multi": (ctypes.c_int, [c_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hicmi_p2_scan_pass": (ctypes.c_int, [_vp, _vp, _vp, c_i64, c_i64, c_dbl, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl),
                                          ctypes.POINTER(ctypes.c_int32)]),
    "hicmi_p2_scan_all": (ctypes.c_int, [_vp, _vp, _vp, c_i64, c_i64, c_dbl, ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl),
                          ctypes.POINTER(c_i64)]),
    "hicmi_timing_reset": (ctypes.c_int, [_vp]),
    "hicmi_timing_enable": (ctypes.c_int, [_vp, ctypes.c_int]),
    "hicmi_timing_get": (ctypes.c_int, [_vp, ctypes.c_char_p, c_i64, _vp, _vp, _vp, c_i64, ctypes.POINTER(c_i64)]),
}


class HicmiError(RuntimeError):
    pass


def load():
    """Load libhicmi.so and attach signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HicmiError("libhicmi.so is not built (%s): run `make -C %s` - this package has no CPU fallback"
                         % (LIB_PATH, os.path.join(_HERE, "csrc")))
    try:
        # PyTorch-ROCm bundles its own libamdhip64; if a second copy (the /opt/rocm one libhicmi links
        # against) initialises first, torch later reports "no ROCm-capable device".  Loading torch
        # first makes both share one HIP runtime.  torch is plumbing only (device buffers, RCCL).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.hicmi_abi_version() != 1:
        raise HicmiError("libhicmi ABI version mismatch")
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HicmiError("libhicmi error %d: %s" % (rc, load().hicmi_last_error().decode("utf-8", "replace")))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def load_hicpro_matrix(path, bin_ids, threads: int = 0):
    """Dense fp64 matrix from a HiC-Pro triplet file, parsed by libhicmi's multi-threaded host loader."""
    ids = np.ascontiguousarray(bin_ids, dtype=np.int64)
    n = len(ids)
    out = np.empty((n, n), dtype=np.float64)
    edges = c_i64()
    _check(load().hicmi_load_hicpro_matrix(os.fsencode(path), _ptr(ids), n, _ptr(out), int(threads), ctypes.byref(edges)))
    return out, edges.value


def scan_valid_pairs(path, names, pairs, threads: int = 0):
    """Lines of a HiC-Pro allValidPairs file that name one of the ordered scaffold pairs ``pairs`` (indices
    into ``names``): returns (pair index, pos1, pos2) arrays in file order and the number of lines read."""
    enc = [n.encode("utf-8") for n in names]
    blob = b"".join(enc)
    off = np.zeros(len(enc) + 1, np.int64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc])
    pa = np.ascontiguousarray([p[0] for p in pairs], dtype=np.int32)
    pb = np.ascontiguousarray([p[1] for p in pairs], dtype=np.int32)
    n_hits, n_lines, handle = c_i64(), c_i64(), _vp()
    _check(load().hicmi_scan_valid_pairs(os.fsencode(path), blob, _ptr(off), len(enc), _ptr(pa), _ptr(pb), len(pa), int(threads),
                                         ctypes.byref(n_hits), ctypes.byref(n_lines), ctypes.byref(handle)))
    idx = np.empty(n_hits.value, np.int32)
    p1 = np.empty(n_hits.value, np.int64)
    p2 = np.empty(n_hits.value, np.int64)
    _check(load().hicmi_scan_fetch(handle, _ptr(idx), _ptr(p1), _ptr(p2)))
    return idx, p1, p2, n_lines.value


def hypergeom_sf(x, M, n, N) -> float:
    """hyper_geom(x, M, n, N) of scaffoldToChromosomes.py:352-368, evaluated by libhicmi's host code."""
    return float(load().hicmi_hypergeom_sf(int(x), int(M), int(n), int(N)))


def hypergeom_decide(x, M, n, N, psig) -> int:
    """1 / 0 / -1: hyper_geom(x, M, n, N) < psig, >= psig, NaN - the early-exit comparison the scan kernels make."""
    return int(load().hicmi_hypergeom_decide(int(x), int(M), int(n), int(N), float(psig)))


class Context:
    """One GPU context (hicmi_ctx): owns the device-resident contact matrix and all stage buffers."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = _vp()
        _check(self._lib.hicmi_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.n = 0
        self._keepalive = None
        self._workers = []
        self._n_window_cand = 0
        self.shard = (0, 1)

    def close(self):
        for w in getattr(self, "_workers", []):
            w.close()
        self._workers = []
        if getattr(self, "_h", None):
            self._lib.hicmi_destroy(self._h)
            self._h = None

    def workers(self, count: int):
        """``count`` extra contexts on the same GPU that share this context's contact matrix (no
        copy): independent work units - Part 2's chromosomes - run on them from host threads."""
        pool = getattr(self, "_workers", None)
        if pool is None:
            pool = self._workers = []
        ptr, n, ld = _vp(), c_i64(), c_i64()
        _check(self._lib.hicmi_contacts_device(self._h, ctypes.byref(ptr), ctypes.byref(n), ctypes.byref(ld)))
        while len(pool) < count:
            pool.append(Context(self.device))
        for w in pool[:count]:
            if getattr(w, "_shared_from", None) != (ptr.value, n.value, ld.value):
                w.set_contacts_device(ptr.value, n.value, ld.value, keepalive=self)
                w._shared_from = (ptr.value, n.value, ld.value)
        return pool[:count]

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- contacts
    def set_contacts(self, mat: np.ndarray):
        """Upload a square contact matrix: fp64, or fp32 (widened on the device, half the transfer)."""
        if isinstance(mat, np.ndarray) and mat.dtype == np.float32:
            mat = np.ascontiguousarray(mat)
            if mat.ndim != 2 or mat.shape[0] != mat.shape[1]:
                raise ValueError("contact matrix must be square")
            _check(self._lib.hicmi_set_contacts_host_f32(self._h, _ptr(mat), mat.shape[0]))
            self.n = mat.shape[0]
            return
        mat = np.ascontiguousarray(mat, dtype=np.float64)
        if mat.ndim != 2 or mat.shape[0] != mat.shape[1]:
            raise ValueError("contact matrix must be square")
        _check(self._lib.hicmi_set_contacts_host(self._h, _ptr(mat), mat.shape[0]))
        self.n = mat.shape[0]

    def set_contacts_device(self, data_ptr: int, n: int, ld: int | None = None, keepalive=None):
        _check(self._lib.hicmi_set_contacts_device(self._h, _vp(data_ptr), n, n if ld is None else ld))
        self.n = n
        self._keepalive = keepalive

    def row_sums(self):
        np_sum = np.empty(self.n, np.float64)
        seq = np.empty(self.n, np.float64)
        _check(self._lib.hicmi_row_sums(self._h, _ptr(np_sum), _ptr(seq)))
        return np_sum, seq

    def set_row_shard(self, first: int, stride: int):
        """One map over several GPUs: this context handles rows first, first + stride, ... of the row-independent
        stages (hicmi_set_row_shard)."""
        _check(self._lib.hicmi_set_row_shard(self._h, int(first), int(stride)))
        self.shard = (int(first), int(stride))

    def set_row_sums(self, np_sum, seq_sum):
        a = np.ascontiguousarray(np_sum, dtype=np.float64)
        b = np.ascontiguousarray(seq_sum, dtype=np.float64)
        if len(a) != self.n or len(b) != self.n:
            raise ValueError("row sums must have n entries")
        _check(self._lib.hicmi_set_row_sums(self._h, _ptr(a), _ptr(b)))

    def compact(self, keep):
        keep = np.ascontiguousarray(keep, dtype=np.int32)
        _check(self._lib.hicmi_compact(self._h, _ptr(keep), len(keep)))
        self.n = len(keep)

    # ---- Part 1
    def upgma(self, want_linkage: bool = True):
        z = np.empty((max(self.n - 1, 0), 4), np.float64) if want_linkage else None
        leaves = np.empty(self.n, np.int32)
        _check(self._lib.hicmi_upgma(self._h, _ptr(z), _ptr(leaves)))
        return leaves, z

    def selftest_division(self, samples=1 << 28, seed=12345):
        bad = ctypes.c_uint64()
        _check(self._lib.hicmi_selftest_division(self._h, seed, samples, ctypes.byref(bad)))
        return bad.value

    def raw_merges(self):
        z = np.empty((max(self.n - 1, 0), 4), np.float64)
        _check(self._lib.hicmi_get_raw_merges(self._h, _ptr(z)))
        return z

    def nnchain_stats(self):
        """Counters of the nn-chain kernels since the last timing_reset (hicmi_nnchain_stats)."""
        out = np.zeros(6, np.float64)
        _check(self._lib.hicmi_nnchain_stats(self._h, _ptr(out)))
        return dict(merges=out[0], scans=out[1], scan_columns=out[2], cache_hits=out[3], retries=int(out[4]))

    def rank_matrix(self, order):
        order = np.ascontiguousarray(order, dtype=np.int32)
        if len(order) != self.n:
            raise ValueError("order must have n entries")
        _check(self._lib.hicmi_rank_matrix(self._h, _ptr(order)))

    def presort_state(self):
        """(state, tied_rows) of the last rank_matrix.  state 0: it sorted every row itself; 1: it re-addressed the
        rows sorted beside the nn-chain and sorted `tied_rows` rows (those holding equal similarities) again;
        2: more than half of the rows hold equal similarities, the pre-sort was discarded (hicmi_presort_state)."""
        st = ctypes.c_int(0)
        tied = c_i64(0)
        _check(self._lib.hicmi_presort_state(self._h, ctypes.byref(st), ctypes.byref(tied)))
        return st.value, int(tied.value)

    def rank_rows(self, row0=0, nrows=None, inverse=False):
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty((nrows, self.n), np.uint16)
        _check(self._lib.hicmi_get_rank_rows(self._h, row0, nrows, 1 if inverse else 0, _ptr(out)))
        return out

    def similarity_row(self, row):
        out = np.empty(self.n, np.float64)
        _check(self._lib.hicmi_get_similarity_row(self._h, row, _ptr(out)))
        return out

    def cut_scan(self, start, M, psig, want_x=False):
        cnt = self.n - start
        sig = np.empty(cnt, np.uint8)
        x = np.empty(cnt, np.int32) if want_x else None
        _check(self._lib.hicmi_cut_scan(self._h, start, M, psig, _ptr(x), _ptr(sig)))
        return (sig, x) if want_x else sig

    def filter_scan(self, start, c, n_rows, M, psig, want_x=False):
        sig = np.empty(n_rows, np.uint8)
        x = np.empty(n_rows, np.int32) if want_x else None
        _check(self._lib.hicmi_filter_scan(self._h, start, c, n_rows, M, psig, _ptr(x), _ptr(sig)))
        return (sig, x) if want_x else sig

    def first_pass_cuts(self, min_size, stop_ind, psig):
        """pre_process_all_matrix_breakpoints' loop on the device (hicmi_first_pass_cuts): (cuts, [(M before, M after), ...])."""
        cuts = np.empty(self.n, np.int32)
        mlog = np.empty((self.n, 2), np.int32)
        nc, nl = c_i64(0), c_i64(0)
        _check(self._lib.hicmi_first_pass_cuts(self._h, int(min_size), int(stop_ind), float(psig), _ptr(cuts), self.n,
                                               ctypes.byref(nc), _ptr(mlog), self.n, ctypes.byref(nl)))
        return [int(v) for v in cuts[:nc.value]], [(int(a), int(b)) for a, b in mlog[:nl.value]]

    def filter_cuts(self, cuts, psig):
        """filter_noisy_breakpoints' loops on the device (hicmi_filter_cuts): (sorted kept cuts, warnings)."""
        cuts = np.ascontiguousarray(cuts, dtype=np.int32)
        out = np.empty(self.n, np.int32)
        m, warned = c_i64(0), c_i64(0)
        _check(self._lib.hicmi_filter_cuts(self._h, _ptr(cuts), len(cuts), float(psig), _ptr(out), self.n,
                                           ctypes.byref(m), ctypes.byref(warned)))
        return [int(v) for v in out[:m.value]], int(warned.value)

    # ---- Part 2
    def p2_select(self, sel):
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        _check(self._lib.hicmi_p2_select(self._h, _ptr(sel), len(sel)))
        self._arr_sig = None

    def p2_total(self) -> float:
        t = c_dbl()
        _check(self._lib.hicmi_p2_total(self._h, ctypes.byref(t)))
        return t.value

    def p2_score(self, perms, total: float):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.ndim != 2:
            raise ValueError("perms must be (n_cand, n_used)")
        out = np.empty(perms.shape[0], np.float64)
        if perms.shape[0]:
            _check(self._lib.hicmi_p2_score(self._h, _ptr(perms), perms.shape[0], perms.shape[1], float(total), _ptr(out)))
        return out

    def p2_score_exact(self, perms, total: float):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        if perms.ndim != 2:
            raise ValueError("perms must be (n_cand, n_used)")
        out = np.empty(perms.shape[0], np.float64)
        if perms.shape[0]:
            _check(self._lib.hicmi_p2_score_exact(self._h, _ptr(perms), perms.shape[0], perms.shape[1], float(total),
                                                  _ptr(out)))
        return out

    # ---- Part 2 search with device-side enumeration
    def p2_layout(self, scaf_start, scaf_len):
        a = np.ascontiguousarray(scaf_start, dtype=np.int32)
        b = np.ascontiguousarray(scaf_len, dtype=np.int32)
        _check(self._lib.hicmi_p2_layout(self._h, _ptr(a), _ptr(b), len(a)))
        self._arr_sig = None

    def p2_set_arrangement(self, ids, rev):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        b = np.ascontiguousarray(rev, dtype=np.uint8)
        sig = a.tobytes() + b.tobytes()
        if sig == getattr(self, "_arr_sig", None):
            return                                   # the device already holds this arrangement
        _check(self._lib.hicmi_p2_set_arrangement(self._h, _ptr(a), _ptr(b), len(a)))
        self._arr_len = len(a)
        self._arr_sig = sig

    def p2_arrangement_total(self) -> float:
        t = c_dbl()
        _check(self._lib.hicmi_p2_arrangement_total(self._h, ctypes.byref(t)))
        return t.value

    def p2_arrangement_score(self, total: float) -> float:
        t = c_dbl()
        _check(self._lib.hicmi_p2_arrangement_score(self._h, float(total), ctypes.byref(t)))
        return t.value

    def p2_score_insertions(self, new_id: int, total: float):
        out = np.empty(2 * (self._arr_len + 1), np.float64)
        _check(self._lib.hicmi_p2_score_insertions(self._h, int(new_id), float(total), _ptr(out)))
        return out

    def p2_window_tables(self, orders, orients):
        a = np.ascontiguousarray(orders, dtype=np.int8)
        b = np.ascontiguousarray(orients, dtype=np.uint8)
        _check(self._lib.hicmi_p2_window_tables(self._h, a.shape[1], _ptr(a), a.shape[0], _ptr(b), b.shape[0]))
        self._n_window_cand = a.shape[0] * b.shape[0]

    def p2_score_window(self, first: int, k: int):
        if not self._n_window_cand:
            raise HicmiError("p2_window_tables has not been called")
        out = np.empty(self._n_window_cand, np.float64)
        _check(self._lib.hicmi_p2_score_window(self._h, int(first), int(k), _ptr(out)))
        return out

    def p2_decide_window(self, first, k, total, floor, cur_fast):
        """One whole window step; returns (pick or -1, literal best, fast score of the resulting arrangement)."""
        pick, best, pf = c_i64(), c_dbl(), c_dbl()
        _check(self._lib.hicmi_p2_decide_window(self._h, int(first), int(k), float(total), float(floor),
                                                float("nan") if cur_fast is None else float(cur_fast),
                                                ctypes.byref(pick), ctypes.byref(best), ctypes.byref(pf)))
        return pick.value, best.value, pf.value

    def p2_decide_insertion(self, ids, rev, new_id, new_rev_now):
        """One whole checkAllScores step; returns (gap or -1, reversed flag, literal best)."""
        a = np.ascontiguousarray(ids, dtype=np.int32)
        b = np.ascontiguousarray(rev, dtype=np.uint8)
        gap, r, best = c_i64(), ctypes.c_int32(), c_dbl()
        _check(self._lib.hicmi_p2_decide_insertion(self._h, _ptr(a), _ptr(b), len(a), int(new_id), int(new_rev_now),
                                                   ctypes.byref(gap), ctypes.byref(r), ctypes.byref(best)))
        self._arr_sig = a.tobytes() + b.tobytes()
        self._arr_len = len(a)
        return gap.value, r.value, best.value

    def p2_insert_all(self, ids, rev, new_ids):
        """orderRemainderScaffolds in one call; returns (ids, rev, bestCost of the last insertion)."""
        s0, k = len(ids), len(new_ids)
        a = np.zeros(s0 + k, np.int32); a[:s0] = ids
        b = np.zeros(s0 + k, np.uint8); b[:s0] = rev
        nw = np.ascontiguousarray(new_ids, dtype=np.int32)
        best = c_dbl()
        _check(self._lib.hicmi_p2_insert_all(self._h, _ptr(a), _ptr(b), s0, _ptr(nw), k, ctypes.byref(best)))
        self._arr_sig = None
        return a, b, best.value

    @staticmethod
    def p2_insert_all_multi(jobs):
        """orderRemainderScaffolds for several chromosomes in lock step (hicmi_p2_insert_all_multi).
        jobs: [(context, ids, rev, new_ids)], one distinct context per chromosome; returns
        [(ids, rev, bestCost of the last insertion)] in the same order."""
        n = len(jobs)
        if n == 0:
            return []
        lib = jobs[0][0]._lib
        keep, a_l, b_l, nw_l = [], [], [], []
        for ctx, ids, rev, new_ids in jobs:
            s0, k = len(ids), len(new_ids)
            a = np.zeros(s0 + k, np.int32); a[:s0] = ids
            b = np.zeros(s0 + k, np.uint8); b[:s0] = rev
            a_l.append(a); b_l.append(b); nw_l.append(np.ascontiguousarray(new_ids, dtype=np.int32))
        handles = (ctypes.c_void_p * n)(*[j[0]._h for j in jobs])
        pa = (ctypes.c_void_p * n)(*[x.ctypes.data for x in a_l])
        pb = (ctypes.c_void_p * n)(*[x.ctypes.data for x in b_l])
        pn = (ctypes.c_void_p * n)(*[x.ctypes.data for x in nw_l])
        s0s = (c_i64 * n)(*[len(j[1]) for j in jobs])
        ks = (c_i64 * n)(*[len(j[3]) for j in jobs])
        best = (c_dbl * n)()
        _check(lib.hicmi_p2_insert_all_multi(n, handles, pa, pb, s0s, pn, ks, best))
        for ctx, _i, _r, _n in jobs:
            ctx._arr_sig = None
        return [(a_l[j], b_l[j], best[j]) for j in range(n)]

    def p2_scan_pass(self, ids, rev, k, total, best, cur_fast):
        """One round of scanOrdering; returns (ids, rev, best, cur_fast, improved)."""
        a = np.ascontiguousarray(ids, dtype=np.int32).copy()
        b = np.ascontiguousarray(rev, dtype=np.uint8).copy()
        bst = c_dbl(float(best))
        cf = c_dbl(float("nan") if cur_fast is None else float(cur_fast))
        imp = ctypes.c_int32()
        _check(self._lib.hicmi_p2_scan_pass(self._h, _ptr(a), _ptr(b), len(a), int(k), float(total), ctypes.byref(bst),
                                            ctypes.byref(cf), ctypes.byref(imp)))
        self._arr_sig = None
        return a, b, bst.value, cf.value, bool(imp.value)

    def p2_scan_all(self, ids, rev, k, total, best, cur_fast):
        """scanOrdering's rounds until one brings no improvement; returns (ids, rev, best, cur_fast, rounds)."""
        a = np.ascontiguousarray(ids, dtype=np.int32).copy()
        b = np.ascontiguousarray(rev, dtype=np.uint8).copy()
        bst = c_dbl(float(best))
        cf = c_dbl(float("nan") if cur_fast is None else float(cur_fast))
        rounds = c_i64()
        _check(self._lib.hicmi_p2_scan_all(self._h, _ptr(a), _ptr(b), len(a), int(k), float(total), ctypes.byref(bst),
                                           ctypes.byref(cf), ctypes.byref(rounds)))
        self._arr_sig = None
        return a, b, bst.value, cf.value, int(rounds.value)

    # ---- plot support
    def plot_percentiles(self, kind, order, q):
        """numpy.percentile (linear) of the cells of the (transformed) matrix restricted to ``order``."""
        qa = np.ascontiguousarray(q, dtype=np.float64)
        out = np.empty(len(qa), np.float64)
        if order is None:
            n_sel, optr, keep = self.n, None, None
        else:
            keep = np.ascontiguousarray(order, dtype=np.int32)
            n_sel, optr = len(keep), _ptr(keep)
        _check(self._lib.hicmi_plot_percentiles(self._h, int(kind), optr, n_sel, _ptr(qa), len(qa), _ptr(out)))
        return out

    def plot_downsample(self, kind, order, px):
        """px x px block means of the (transformed) matrix restricted / permuted by ``order``."""
        if order is None:
            n_sel, optr, keep = self.n, None, None
        else:
            keep = np.ascontiguousarray(order, dtype=np.int32)
            n_sel, optr = len(keep), _ptr(keep)
        out = np.empty((int(px), int(px)), np.float64)
        _check(self._lib.hicmi_plot_downsample(self._h, int(kind), optr, n_sel, int(px), _ptr(out)))
        return out

    # ---- misc
    def synchronize(self):
        _check(self._lib.hicmi_synchronize(self._h))

    def stream(self) -> int:
        s = _vp()
        _check(self._lib.hicmi_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    def timing_enable(self, on=True):
        """True / 1: HIP events around every kernel family; 2: only around the few-launch Part 1 families
        (hicmi.h); False / 0: off."""
        _check(self._lib.hicmi_timing_enable(self._h, int(on)))
        for w in self._workers:
            w.timing_enable(on)

    def timing_reset(self):
        _check(self._lib.hicmi_timing_reset(self._h))
        for w in self._workers:
            w.timing_reset()

    def timing(self):
        """Per kernel family: device ms, launches, algorithmic bytes - summed over this context and
        the worker contexts that share its matrix."""
        out = self._timing_one()
        for w in self._workers:
            for k, v in w._timing_one().items():
                for f in ("ms", "launches", "bytes"):
                    out[k][f] += v[f]
        return out

    def _timing_one(self):
        names = ctypes.create_string_buffer(1024)
        ms = np.zeros(32, np.float64)
        launches = np.zeros(32, np.int64)
        nbytes = np.zeros(32, np.float64)
        cnt = c_i64()
        _check(self._lib.hicmi_timing_get(self._h, names, 1024, _ptr(ms), _ptr(launches), _ptr(nbytes), 32,
                                          ctypes.byref(cnt)))
        out = {}
        for k, nm in enumerate(names.value.decode().split(";")[:cnt.value]):
            out[nm] = dict(ms=float(ms[k]), launches=int(launches[k]), bytes=float(nbytes[k]))
        return out
