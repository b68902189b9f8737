"""Louvain tail of Part 1 (scaffoldToChromosomes.py:239-349).

The reference re-partitions the bins after the last cut index - the last ``modularity`` fraction of the map,
where groups are too small for the hypergeometric scan - with ``community.best_partition(randomize=True)``
from python-louvain (unpinned in packageInstallCommands.txt; `community` is not installed here, networkx is),
keeping the best of ``louvainRounds`` random starts.  Its random state is never seeded (S2C:253), so the
reference does not reproduce its own output; parity for this stage is statistical: planted groups are
recovered, the graph / weights / modularity value equal networkx's on the reference's own add_edge loop, and the
best-of-rounds score agrees with networkx's independent Louvain within 2 % over ten seeds
(tests/test_modularity_cpu.py).  Parity with python-louvain itself stays unpinned.

This module restates the published algorithm (Blondel et al. 2008, as implemented by python-louvain 0.16:
node order and candidate order shuffled per pass, gain threshold 1e-7, aggregation until the gain stalls)
on a dense weight matrix with a SEEDED generator (``HICMI_LOUVAIN_SEED``, default 0), so a run is
reproducible.  The graph is small (5 % of the bins), the sweep is sequential: host code, NumPy.

Graph of the reference (S2C:285-297): one node per remaining bin, an edge for EVERY ordered pair including
(i, i), weight = the log10(similarity + 1) cell; ``add_edge(a, b)`` followed by ``add_edge(b, a)`` overwrites,
so an unordered pair keeps the weight of the LATER row - the lower triangle.
"""
from __future__ import annotations

import collections
import math
import os
import time

import numpy as np

_MIN = 0.0000001            # python-louvain __MIN


def graph_weights(log_similarity_tail):
    """Symmetric weight matrix of the reference's graph: lower triangle mirrored, diagonal = self loops."""
    a = np.asarray(log_similarity_tail, dtype=np.float64)
    low = np.tril(a)
    return low + np.tril(a, -1).T


class _Status:
    """python-louvain's Status for a dense graph: A symmetric, A[i][i] = weight of the self loop."""

    def __init__(self, A):
        self.A = A
        n = len(A)
        diag = np.diag(A).copy()
        self.total_weight = float((A.sum() + diag.sum()) / 2.0)           # every edge once, loops included
        self.gdegrees = A.sum(axis=1) + diag                              # a self loop counts twice in a degree
        self.loops = diag
        self.node2com = np.arange(n)
        self.degrees = self.gdegrees.copy()
        self.internals = diag.copy()

    def modularity(self):
        links = self.total_weight
        if links <= 0:
            return 0.0
        res = 0.0
        for com in np.unique(self.node2com):
            res += self.internals[com] / links - (self.degrees[com] / (2.0 * links)) ** 2
        return float(res)


def _one_level(st: _Status, rng):
    """One level of python-louvain's __one_level.  The sweep over the nodes is sequential (every move changes the
    state the next node sees); what one node does - the weights to every community, the communities present among
    the OTHER nodes, the best gain over them in shuffled order (first maximum wins, as the reference's strict `>`
    loop) - is array work: O(n) per node instead of a sort plus a Python loop over the communities."""
    A, n = st.A, len(st.A)
    sizes = np.bincount(st.node2com, minlength=n)                          # members per community
    modified, new_mod = True, st.modularity()
    while modified:
        cur_mod, modified = new_mod, False
        for node in rng.permutation(n):
            com_node = st.node2com[node]
            degc_totw = st.gdegrees[node] / (st.total_weight * 2.0)
            # weight from `node` to every community (the graph is complete: every community is a neighbour)
            row = A[node].copy()
            row[node] = 0.0
            sizes[com_node] -= 1
            present = np.flatnonzero(sizes)                                # communities of the other nodes, ascending
            w_to = np.bincount(st.node2com, weights=row, minlength=n)
            w_own = w_to[com_node] if sizes[com_node] > 0 else 0.0
            remove_cost = -w_own + (st.degrees[com_node] - st.gdegrees[node]) * degc_totw
            # __remove
            st.degrees[com_node] -= st.gdegrees[node]
            st.internals[com_node] -= w_own + st.loops[node]
            st.node2com[node] = -1
            best_com, best_increase = com_node, 0.0
            if len(present):
                order = rng.permutation(present)
                incr = remove_cost + w_to[order] - st.degrees[order] * degc_totw
                k = int(np.argmax(incr))                                   # first maximum in the shuffled order
                if incr[k] > best_increase:
                    best_increase, best_com = float(incr[k]), order[k]
            else:
                rng.permutation(present)                                   # (keeps the generator's stream as before)
            # __insert
            w_best = w_to[best_com] if sizes[best_com] > 0 else 0.0
            st.node2com[node] = best_com
            sizes[best_com] += 1
            st.degrees[best_com] += st.gdegrees[node]
            st.internals[best_com] += w_best + st.loops[node]
            if best_com != com_node:
                modified = True
        new_mod = st.modularity()
        if new_mod - cur_mod < _MIN:
            break


def _renumber(node2com):
    """Communities numbered 0.. in order of first appearance."""
    seen, out = {}, np.empty(len(node2com), dtype=np.int64)
    for i, c in enumerate(node2com):
        out[i] = seen.setdefault(int(c), len(seen))
    return out


def _induced(A, part):
    """Community graph: weights between communities summed; a community's internal edges (each once) and its
    members' self loops become its self loop."""
    k = int(part.max()) + 1
    P = np.zeros((len(A), k))
    P[np.arange(len(A)), part] = 1.0
    B = P.T @ A @ P
    loops = P.T @ np.diag(A)
    d = (np.diag(B) + loops) / 2.0
    B[np.arange(k), np.arange(k)] = d
    return B


def best_partition(A, rng):
    """community.best_partition(graph, randomize=True) on the dense weight matrix A: node -> community."""
    A = np.asarray(A, dtype=np.float64)
    n = len(A)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    offdiag_edges = n * (n - 1) // 2 + n
    if offdiag_edges == 0:
        return np.arange(n)
    st = _Status(A.copy())
    _one_level(st, rng)
    mod = st.modularity()
    part = _renumber(st.node2com)
    levels = [part]
    cur = _induced(A, part)
    while True:
        st = _Status(cur)
        _one_level(st, rng)
        new_mod = st.modularity()
        if new_mod - mod < _MIN:
            break
        part = _renumber(st.node2com)
        levels.append(part)
        mod = new_mod
        cur = _induced(cur, part)
    node2com = levels[0].copy()
    for lvl in levels[1:]:
        node2com = lvl[node2com]
    return node2com


def modularity(partition, A):
    """community.modularity(partition, graph) for the dense weight matrix A."""
    A = np.asarray(A, dtype=np.float64)
    part = np.asarray(partition)
    links = (A.sum() + np.trace(A)) / 2.0
    if links == 0:
        raise ValueError("A graph without link has an undefined modularity")
    deg = A.sum(axis=1) + np.diag(A)
    res = 0.0
    for com in np.unique(part):
        members = part == com
        sub = A[np.ix_(members, members)]
        inc = (sub.sum() + np.trace(sub)) / 2.0                    # internal edges once, self loops once
        res += inc / links - (deg[members].sum() / (2.0 * links)) ** 2
    return float(res)


def modularity_rounds(A, louvain_rounds=1, seed=None):
    """S2C:239-262: the best of ``louvain_rounds`` randomised Louvain runs (strict '>' keeps the earliest)."""
    if seed is None:
        seed = int(os.environ.get("HICMI_LOUVAIN_SEED", "0"))
    best_mod_score, best = -2.0, None
    for i in range(0, louvain_rounds):
        part = best_partition(A, np.random.default_rng([seed, i]))
        mod_score = modularity(part, A)
        if mod_score > best_mod_score:
            borg = best_mod_score
            best_mod_score, best = mod_score, part
            print("Previous best modularity score {}, Current best found {}, Louvain round {}".format(borg, mod_score, i + 1))
    return best, best_mod_score


def log_transform(similarity):
    """logTransformMatrix(matrix, logBase=10) (S2C:165-183): log10(v + 1) for non-zero cells, 0 otherwise."""
    s = np.asarray(similarity, dtype=np.float64)
    out = np.zeros_like(s)
    nz = s != 0.0
    out[nz] = np.log(s[nz] + 1.0) / math.log(10)
    return out


def modularity_remaining_data(log_similarity_tail, binList, cutIndices, n_rounds=20, seed=None):
    """S2C:263-349 on the tail sub-matrix (rows/columns from the last cut index on, in the current order).
    Returns (new_order, cutIndices): the permutation of ``binList`` positions to apply - head unchanged, tail
    grouped by community, largest community first - and the extended cut indices."""
    startTime = time.time()
    cutIndices = list(cutIndices)
    if len(cutIndices) == 0:
        print("- Attempting to resolve groupings by modularity alone... This could take a while if matrix size is large "
              "and n_rounds is set high as well...")
        cutIndices = [0]
    cutIndices = sorted(cutIndices)
    startIndex = cutIndices[-1]
    n_total = len(binList)
    A = graph_weights(log_similarity_tail)
    if len(A) != n_total - startIndex:
        raise ValueError("tail matrix does not match binList[startIndex:]")
    print("- Maximizing so-called modularity...")
    print("- Graph created with " + str(len(A)) + " nodes, and " + str(len(A) * (len(A) - 1) // 2 + len(A)) + " edges")
    print("- Performing " + str(n_rounds) + " rounds of the louvain method...")
    node_to_group, _mod_score = modularity_rounds(A, louvain_rounds=n_rounds, seed=seed)
    group_sizes = collections.Counter(node_to_group.tolist())
    group_count = len(group_sizes)
    remaining_groups = [k for k, _v in sorted(group_sizes.items(), key=lambda kv: kv[1], reverse=True)]
    remaining_order = []
    for rg in remaining_groups:
        remaining_order += [startIndex + i for i in range(len(A)) if node_to_group[i] == rg]
        cutIndices.append(cutIndices[-1] + group_sizes[rg])
    new_order = list(range(startIndex)) + remaining_order
    if cutIndices[0] == 0:
        cutIndices.pop(0)
    if cutIndices and cutIndices[-1] == n_total:
        cutIndices.pop(-1)
    total_groups = len(cutIndices) + 1
    print("- Modularity maximization total time = " + str(time.time() - startTime))
    print("- Chromosomes found via HMMs or Hyper geometrics = {}".format(total_groups - group_count))
    print("- Chromosomes found via modularity maximization = " + str(group_count))
    print("- Total chromosomes found {}".format(total_groups))
    return new_order, cutIndices
