/*
 * oracle_c.c - CPU restatement (plain C, fp64, no FMA contraction) of the native pieces the
 * reference reaches through un-vendored third-party code.  TEST INFRASTRUCTURE: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (hic_genome_assembler_amd/) never does.
 *
 * Each function names the reference call site it stands in for (paths relative to
 * /root/reference/HIC_ASSEMBLER) and the third-party routine whose published algorithm it restates:
 *   NumPy 2.2.6  add.reduce pairwise summation         (scaffoldToChromosomes.py:112,147; orderGenome.py:188,343)
 *   SciPy 1.15.3 cluster.hierarchy nn_chain / label     (scaffoldToChromosomes.py:194-197)
 *   SciPy 1.15.3 cluster.hierarchy dendrogram leaf walk (scaffoldToChromosomes.py:204)
 * Pinned by tests/test_oracle_cpu.py against the installed NumPy/SciPy and against the golden
 * fixtures produced by running the reference itself (oracle/gen_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fPIC -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* NumPy float64 add.reduce over a strided 1-D view: pairwise blocks of <=128 with 8 partial   */
/* sums, recursive halving above that, and the reduction machinery hands the inner loop at     */
/* most 8192 elements at a time, accumulating chunk results left to right starting from 0.0.   */
static double pairwise(const double *a, long n, long s)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; i++) r += a[i * s];
        return r;
    }
    if (n <= 128) {
        double r0 = a[0], r1 = a[s], r2 = a[2 * s], r3 = a[3 * s];
        double r4 = a[4 * s], r5 = a[5 * s], r6 = a[6 * s], r7 = a[7 * s];
        long i, lim = n - (n % 8);
        for (i = 8; i < lim; i += 8) {
            r0 += a[(i + 0) * s]; r1 += a[(i + 1) * s]; r2 += a[(i + 2) * s]; r3 += a[(i + 3) * s];
            r4 += a[(i + 4) * s]; r5 += a[(i + 5) * s]; r6 += a[(i + 6) * s]; r7 += a[(i + 7) * s];
        }
        double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; i++) res += a[i * s];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise(a, n2, s) + pairwise(a + n2 * s, n - n2, s);
}

double hio_np_sum(const double *a, long n, long stride)
{
    double acc = 0.0;
    for (long c = 0; c < n; c += 8192) {
        long m = n - c < 8192 ? n - c : 8192;
        acc += pairwise(a + c * stride, m, stride);
    }
    return acc;
}

/* Python builtin sum() over a float64 row: left-to-right (scaffoldToChromosomes.py:134). */
double hio_seq_sum(const double *a, long n, long stride)
{
    double acc = 0.0;
    for (long i = 0; i < n; i++) acc += a[i * stride];
    return acc;
}

/* ------------------------------------------------------------------------------------------ */
/* SciPy squareform(checks=False) + linkage(method='average') nn_chain.                         */
/* D: n x n row-major with leading dimension ld; only entries i<j are read (squareform copies   */
/* the row-major strict upper triangle, scaffoldToChromosomes.py:194).                          */
/* Zraw: (n-1) x 4 output in MERGE order (x, y, height, size) before sorting/labelling.         */
static inline long cidx(long n, long i, long j)
{
    if (i > j) { long t = i; i = j; j = t; }
    return n * i - (i * (i + 1)) / 2 + (j - i - 1);
}

int hio_nn_chain_average(const double *D, long n, long ld, double *Zraw)
{
    if (n < 2) return 0;
    long m = n * (n - 1) / 2;
    double *y = (double *)malloc(sizeof(double) * (size_t)m);
    int *size = (int *)malloc(sizeof(int) * (size_t)n);
    int *chain = (int *)malloc(sizeof(int) * (size_t)n);
    if (!y || !size || !chain) { free(y); free(size); free(chain); return -1; }
    long k = 0;
    for (long i = 0; i < n; i++)
        for (long j = i + 1; j < n; j++) y[k++] = D[i * ld + j];
    for (long i = 0; i < n; i++) size[i] = 1;
    long chain_len = 0;
    for (long step = 0; step < n - 1; step++) {
        long x = 0, yv = 0;
        double cur = 0.0;
        if (chain_len == 0) {
            chain_len = 1;
            for (long i = 0; i < n; i++) if (size[i] > 0) { chain[0] = (int)i; break; }
        }
        for (;;) {
            x = chain[chain_len - 1];
            if (chain_len > 1) { yv = chain[chain_len - 2]; cur = y[cidx(n, x, yv)]; }
            else { cur = INFINITY; }
            for (long i = 0; i < n; i++) {
                if (size[i] == 0 || x == i) continue;
                double d = y[cidx(n, x, i)];
                if (d < cur) { cur = d; yv = i; }          /* strict '<': lowest index wins ties */
            }
            if (chain_len > 1 && yv == chain[chain_len - 2]) break;
            chain[chain_len++] = (int)yv;
        }
        chain_len -= 2;
        if (x > yv) { long t = x; x = yv; yv = t; }
        int nx = size[x], ny = size[yv];
        Zraw[4 * step + 0] = (double)x;
        Zraw[4 * step + 1] = (double)yv;
        Zraw[4 * step + 2] = cur;
        Zraw[4 * step + 3] = (double)(nx + ny);
        size[x] = 0;
        size[yv] = nx + ny;
        for (long i = 0; i < n; i++) {
            if (size[i] == 0 || i == yv) continue;
            double dxi = y[cidx(n, i, x)], dyi = y[cidx(n, i, yv)];
            /* separate mul, mul, add, int add, div - exactly five roundings */
            y[cidx(n, i, yv)] = ((double)nx * dxi + (double)ny * dyi) / (double)(nx + ny);
        }
    }
    free(y); free(size); free(chain);
    return 0;
}

/* Stable sort of merges by height (numpy argsort kind='mergesort') followed by SciPy's
 * union-find relabelling.  Z: (n-1) x 4 output in SciPy's linkage-matrix convention. */
static void merge_sort_idx(const double *h, long *idx, long *tmp, long lo, long hi)
{
    if (hi - lo < 2) return;
    long mid = lo + (hi - lo) / 2;
    merge_sort_idx(h, idx, tmp, lo, mid);
    merge_sort_idx(h, idx, tmp, mid, hi);
    long a = lo, b = mid, o = lo;
    while (a < mid && b < hi) tmp[o++] = (h[idx[b]] < h[idx[a]]) ? idx[b++] : idx[a++];
    while (a < mid) tmp[o++] = idx[a++];
    while (b < hi) tmp[o++] = idx[b++];
    memcpy(idx + lo, tmp + lo, sizeof(long) * (size_t)(hi - lo));
}

static long uf_find(long *parent, long x)
{
    long p = x;
    while (parent[x] != x) x = parent[x];
    while (parent[p] != x) { long nx = parent[p]; parent[p] = x; p = nx; }
    return x;
}

int hio_label(const double *Zraw, long n, double *Z)
{
    long m = n - 1;
    if (m <= 0) return 0;
    long *idx = (long *)malloc(sizeof(long) * (size_t)m);
    long *tmp = (long *)malloc(sizeof(long) * (size_t)m);
    double *h = (double *)malloc(sizeof(double) * (size_t)m);
    long *parent = (long *)malloc(sizeof(long) * (size_t)(2 * n - 1));
    long *sz = (long *)malloc(sizeof(long) * (size_t)(2 * n - 1));
    if (!idx || !tmp || !h || !parent || !sz) { free(idx); free(tmp); free(h); free(parent); free(sz); return -1; }
    for (long i = 0; i < m; i++) { idx[i] = i; h[i] = Zraw[4 * i + 2]; }
    merge_sort_idx(h, idx, tmp, 0, m);
    for (long i = 0; i < 2 * n - 1; i++) { parent[i] = i; sz[i] = i < n ? 1 : 0; }
    long next = n;
    for (long r = 0; r < m; r++) {
        const double *src = Zraw + 4 * idx[r];
        long a = uf_find(parent, (long)src[0]), b = uf_find(parent, (long)src[1]);
        if (a < b) { Z[4 * r] = (double)a; Z[4 * r + 1] = (double)b; }
        else       { Z[4 * r] = (double)b; Z[4 * r + 1] = (double)a; }
        Z[4 * r + 2] = src[2];
        parent[a] = next; parent[b] = next;
        sz[next] = sz[a] + sz[b];
        Z[4 * r + 3] = (double)sz[next];
        next++;
    }
    free(idx); free(tmp); free(h); free(parent); free(sz);
    return 0;
}

/* SciPy dendrogram(count_sort='ascending', get_leaves=True): pre-order walk from the root;
 * the child with the smaller leaf count is visited first, ties keep (Z[i,0], Z[i,1]) order. */
int hio_leaf_order(const double *Z, long n, int32_t *leaves)
{
    if (n == 1) { leaves[0] = 0; return 0; }
    long *stack = (long *)malloc(sizeof(long) * (size_t)(2 * n));
    if (!stack) return -1;
    long sp = 0, out = 0;
    stack[sp++] = 2 * n - 2;
    while (sp > 0) {
        long node = stack[--sp];
        if (node < n) { leaves[out++] = (int32_t)node; continue; }
        const double *row = Z + 4 * (node - n);
        long aa = (long)row[0], ab = (long)row[1];
        long na = aa < n ? 1 : (long)Z[4 * (aa - n) + 3];
        long nb = ab < n ? 1 : (long)Z[4 * (ab - n) + 3];
        if (na > nb) { stack[sp++] = aa; stack[sp++] = ab; }      /* visit ab first */
        else         { stack[sp++] = ab; stack[sp++] = aa; }      /* visit aa first */
    }
    free(stack);
    return out == n ? 0 : -2;
}

/* ------------------------------------------------------------------------------------------ */
/* Part 2 objective, literal form (orderGenome.py:185-191 / 323-330) on the matrix permuted by  */
/* perm (orderGenome.py:348,358,463,534 gather with numpy.ix_): numpy.trace of offset i is a    */
/* pairwise sum over the strided diagonal.                                                     */
static double diag_sum_perm(const double *M, long ld, const int32_t *perm, long n, long off, double *scratch)
{
    long len = n - off;
    for (long a = 0; a < len; a++) scratch[a] = M[(long)perm[a] * ld + perm[a + off]];
    return hio_np_sum(scratch, len, 1);
}

/* numpy.trace as NUMBA compiles it inside costFunction_numba (orderGenome.py:184-191, nopython mode):       */
/* numba's np_trace is `ret = 0; for i in range(n): ret += a[i, k + i]` - sequential, not pairwise.  Selected */
/* with hio_set_trace_order(1) for the cost functions only; the totals stay NumPy's (orderGenome.py:343).    */
static int g_sequential_trace = 0;
void hio_set_trace_order(int sequential) { g_sequential_trace = sequential; }

static double diag_sum_cost(const double *M, long ld, const int32_t *perm, long n, long off, double *scratch)
{
    if (!g_sequential_trace) return diag_sum_perm(M, ld, perm, n, off, scratch);
    double acc = 0.0;
    for (long a = 0; a < n - off; a++) acc += M[(long)perm[a] * ld + perm[a + off]];
    return acc;
}

double hio_total_upper(const double *M, long ld, const int32_t *perm, long n)
{
    /* Python sum() of the per-offset traces, offsets 1..n-1 (orderGenome.py:343,448,506) */
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double total = 0.0;
    for (long off = 1; off < n; off++) total += diag_sum_perm(M, ld, perm, n, off, scratch);
    free(scratch);
    return total;
}

double hio_cost_literal(const double *M, long ld, const int32_t *perm, long n, double total)
{
    double *scratch = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    double cum = 0.0, cost = 0.0;
    for (long off = 1; off < n; off++) {
        cum += diag_sum_cost(M, ld, perm, n, off, scratch);
        cost += (cum / total / (double)off);
    }
    free(scratch);
    return cost;
}

void hio_cost_literal_batch(const double *M, long ld, const int32_t *perms, long n_cand, long n,
                            double total, double *out)
{
    for (long c = 0; c < n_cand; c++) out[c] = hio_cost_literal(M, ld, perms + c * n, n, total);
}

/* ------------------------------------------------------------------------------------------ */
/* Cut scan counts (scaffoldToChromosomes.py:455-459 and 631): number of entries v among the    */
/* first L of a rank-ordered row with lo <= v <= hi.                                           */
long hio_count_prefix(const int64_t *row, long L, long lo, long hi)
{
    long c = 0;
    for (long k = 0; k < L; k++) c += (row[k] >= lo && row[k] <= hi);
    return c;
}
