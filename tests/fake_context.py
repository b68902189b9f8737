"""A stand-in for hic_genome_assembler_amd._lib.Context that answers every C-ABI call with the CPU
oracle.  TEST INFRASTRUCTURE: lets the CPU suite exercise the product's HOST control flow
(cut-candidate loops, scaffold search, file writers) without a GPU.  It is never importable from
the package and the product has no path that reaches it."""
import numpy as np

import hic_oracle as orc


class OracleContext:
    def __init__(self, device=0):
        self.n = 0
        self.mat = None
        self.shard = (0, 1)
        self._sums = None

    # ---- one map over several ranks: like libhicmi, answer only for the owned rows until the sums are installed
    def set_row_shard(self, first, stride):
        self.shard = (int(first), int(stride))

    def set_row_sums(self, np_sum, seq_sum):
        self._sums = (np.array(np_sum), np.array(seq_sum))

    def _owned(self, first_row, count):
        first, stride = self.shard
        return (np.arange(first_row, first_row + count) % stride) == first

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def set_contacts(self, mat):
        self.mat = np.ascontiguousarray(mat, dtype=np.float64)
        self.n = len(self.mat)
        self._sums = None

    def row_sums(self):
        a, b = orc.np_row_sums(self.mat), orc.seq_row_sums(self.mat)
        if self.shard[1] > 1 and self._sums is None:
            own = self._owned(0, self.n)
            return np.where(own, a, 0.0), np.where(own, b, 0.0)
        return a, b

    def compact(self, keep):
        keep = np.asarray(keep, dtype=np.int64)
        self.set_contacts(self.mat[np.ix_(keep, keep)])
        self._sums = (None, None)                          # libhicmi recomputes all sums on the compacted matrix

    def upgma(self, want_linkage=True):
        leaves, z = orc.average_cluster_leaves(orc.to_distance(self.mat))
        return leaves, z

    # ---- plot support: NumPy restatement of hicmi_plot_percentiles / hicmi_plot_downsample
    def plot_matrix(self, kind, order):
        m = self.mat
        if kind >= 1:
            m = orc.to_distance(self.mat)
        if kind == 2:
            m = orc.seq_row_sums(self.mat)[:, None] * (1.0 - (m - 1.0))
        if order is not None:
            o = np.asarray(order, dtype=np.int64)
            m = m[np.ix_(o, o)]
        return m

    def plot_percentiles(self, kind, order, q):
        return np.percentile(self.plot_matrix(kind, order), q)

    def plot_downsample(self, kind, order, px):
        m = self.plot_matrix(kind, order)
        n = len(m)
        edges = (np.arange(px + 1, dtype=np.int64) * n) // px
        out = np.empty((px, px))
        for r in range(px):
            for c in range(px):
                out[r, c] = m[edges[r]:edges[r + 1], edges[c]:edges[c + 1]].mean()
        return out

    def rank_matrix(self, order):
        order = np.asarray(order, dtype=np.int64)
        dist = orc.to_distance(self.mat)[:, order][order]
        rs = orc.seq_row_sums(self.mat)[order]
        sim = rs[:, None] * (1.0 - (dist - 1.0))
        self.R = orc.rank_order(sim)

    def rank_rows(self, row0=0, nrows=None, inverse=False):
        assert not inverse
        nrows = self.n - row0 if nrows is None else nrows
        return self.R[row0:row0 + nrows].astype(np.uint16)

    def cut_scan(self, start, M, psig, want_x=False):
        x = orc.first_pass_counts(self.R, start)[start:]
        L = np.arange(1, self.n - start)
        p = orc.hyper_geom(x[1:], M, L, L)
        sig = np.concatenate(([0], np.where(p >= psig, 0, 1))).astype(np.uint8)
        if self.shard[1] > 1:
            own = self._owned(start, len(sig))
            sig, x = np.where(own, sig, 0).astype(np.uint8), np.where(own, x, 0)
        return (sig, x.astype(np.int32)) if want_x else sig

    def filter_scan(self, start, c, n_rows, M, psig, want_x=False):
        sub = self.R[start:start + n_rows, :c - start]
        x = np.count_nonzero((sub >= start) & (sub <= c), axis=1)
        p = orc.hyper_geom(x, M, c - start, c - start)
        sig = np.where(p < psig, 1, 0).astype(np.uint8)
        if self.shard[1] > 1:
            own = self._owned(start, len(sig))
            sig, x = np.where(own, sig, 0).astype(np.uint8), np.where(own, x, 0)
        return (sig, x.astype(np.int32)) if want_x else sig

    def p2_select(self, sel):
        sel = np.asarray(sel, dtype=np.int64)
        self.sub = np.ascontiguousarray(self.mat[np.ix_(sel, sel)])

    def p2_total(self):
        n = len(self.sub)
        ident = np.arange(n, dtype=np.int32)
        return orc.lib().hio_total_upper(orc._dp(self.sub), n, orc._ip(ident), n)

    def p2_score(self, perms, total):
        perms = np.ascontiguousarray(perms, dtype=np.int32)
        out = np.zeros(len(perms))
        orc.lib().hio_cost_literal_batch(orc._dp(self.sub), self.sub.shape[1], orc._ip(perms), perms.shape[0],
                                         perms.shape[1], float(total), orc._dp(out))
        return out

    p2_score_exact = p2_score

    # ---- device-side enumeration API, answered by expanding every candidate on the host
    def p2_layout(self, scaf_start, scaf_len):
        self.scaf_start = [int(v) for v in scaf_start]
        self.scaf_len = [int(v) for v in scaf_len]

    def _positions(self, sid, rev):
        a = np.arange(self.scaf_start[sid], self.scaf_start[sid] + self.scaf_len[sid], dtype=np.int32)
        return a[::-1] if rev else a

    def p2_set_arrangement(self, ids, rev):
        self.arr_ids = [int(v) for v in ids]
        self.arr_rev = [int(v) for v in rev]
        self._arr_len = len(self.arr_ids)

    def _row(self, ids, rev):
        return np.concatenate([self._positions(i, r) for i, r in zip(ids, rev)]) if len(ids) else np.zeros(0, np.int32)

    def _literal(self, rows, total):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        return self.p2_score(rows, total)

    def p2_arrangement_total(self):
        row = self._row(self.arr_ids, self.arr_rev)
        return orc.lib().hio_total_upper(orc._dp(self.sub), self.sub.shape[1], orc._ip(row), len(row))

    def p2_arrangement_score(self, total):
        return float(self._literal(self._row(self.arr_ids, self.arr_rev)[None, :], total)[0])

    def p2_score_insertions(self, new_id, total):
        pieces = [self._positions(i, r) for i, r in zip(self.arr_ids, self.arr_rev)]
        rows = []
        for g in range(len(pieces) + 1):
            for r in (0, 1):
                rows.append(np.concatenate(pieces[:g] + [self._positions(int(new_id), r)] + pieces[g:]))
        return self._literal(np.stack(rows), total)

    def p2_window_tables(self, orders, orients):
        self.orders = np.asarray(orders)
        self.orients = np.asarray(orients)
        self._n_window_cand = len(self.orders) * len(self.orients)

    def p2_score_window(self, first, k):
        head = self._row(self.arr_ids[:first], self.arr_rev[:first])
        tail = self._row(self.arr_ids[first + k:], self.arr_rev[first + k:])
        win = self.arr_ids[first:first + k]
        rows = []
        for o in self.orders:
            for r in self.orients:
                mid = [self._positions(win[int(j)], int(rv)) for j, rv in zip(o, r)]
                rows.append(np.concatenate([head] + mid + [tail]))
        # delta may contain any per-window constant; use "score * 1.0" (the caller divides by total again)
        return self._literal(np.stack(rows), 1.0)
