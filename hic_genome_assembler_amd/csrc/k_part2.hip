// k_part2.hip - gfx950 kernels for Part 2 (orderGenome.py): sub-matrix selection and the
// distance-decay ordering objective evaluated for a batch of candidate orders.
//
// costFunction_numba (orderGenome.py:184-191):
//     cum = 0; cost = 0; for i in 1..n-1: cum += trace(M_perm, offset=i); cost += cum/total/i
// equals   cost = (1/total) * sum_{a<b} M[p(a)][p(b)] * (H[n-1] - H[b-a-1]),
// H[k] = 1 + 1/2 + ... + 1/k.  The candidate's permuted matrix is never built (the reference
// gathers it with numpy.ix_ once per candidate, OG:348,358,463,534): the kernel reads the selected
// sub-matrix through the permutation, which sits in LDS.  fp64 throughout; one workgroup per
// candidate with a fixed reduction order, so identical index lists give bit-identical scores and
// the reference's first-strict-maximum tie-breaking (OG:349,359,464,535) is preserved on the host.
#include <cstdlib>

#include "hicmi_internal.h"

namespace hicmi {

__global__ __launch_bounds__(256) void k_p2_select(const double* __restrict__ C, int64_t ldc,
                                                   const int32_t* __restrict__ sel, int n, double* __restrict__ M2,
                                                   int64_t ld2)
{
    int r = blockIdx.y;
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    M2[(int64_t)r * ld2 + c] = C[(int64_t)sel[r] * ldc + sel[c]];
}

void launch_p2_select(const double* C, int64_t ldc, const int32_t* sel, int n, double* M2, int64_t ld2, hipStream_t s)
{
    hipLaunchKernelGGL(k_p2_select, dim3((n + 255) / 256, n), dim3(256), 0, s, C, ldc, sel, n, M2, ld2);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// -------------------------------------------------------------------------------------------------
// LITERAL objective (bit-exact with the reference's NumPy arithmetic).  The reference decides
// "cost > bestCost" between values that can differ by one ulp (the same arrangement scored under
// two differently rounded totals, OG:506 vs OG:343), so the candidates that can win a step are
// re-scored with exactly the reference's operation order:
//   T_i   = numpy.trace(M_perm, offset=i): float64 add.reduce over the strided diagonal - pairwise
//           leaves of <=128 elements with 8 interleaved partial sums, recursive halving above that,
//           8192-element chunks accumulated left to right (OG:188)
//   total = Python sum of T_1..T_{n-1}, left to right                           (OG:343,448,506)
//   cost  = sum_i ((T_1+..+T_i) / total) / i, left to right                      (OG:185-191)
// The order of additions is fixed by NumPy, but most of them are independent: one 64-lane
// workgroup owns one diagonal, 8 lanes own one leaf (lane k = partial sum k: at most 16 dependent
// adds), the fixed ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) tree is three shuffles, and only the handful
// of leaf results are combined serially along NumPy's recursion tree.
// enumerate_leaves / combine_leaves / MAX_LEAVES: hicmi_internal.h (shared with the row sums of k_part1.hip)

// Where position t of a candidate order sits in the selection: an index list with, optionally, one
// scaffold (selection range [new_start, new_start + L), reversed or not) spliced in at position P - the
// candidates of an insertion step are never materialised.  p == nullptr: the identity list.
struct IndexMap {
    const int32_t* p;
    int P, L, new_start, rev;
    __device__ __forceinline__ int at(int t) const
    {
        if (t < P) return p ? p[t] : t;
        if (t - P < L) return new_start + (rev ? L - 1 - (t - P) : t - P);
        return p ? p[t - L] : t - L;
    }
};
static constexpr int kNoSplice = 0x7fffffff;

// Workgroups go to the 8 XCDs round-robin.  Eight neighbouring diagonals read the same 64-byte lines of a
// row, so each XCD takes whole groups of 8 neighbours (one L2 sees a line once instead of all eight L2s
// pulling it across the fabric): within every 64 workgroups the "XCD" and "member" digits swap.
__device__ __forceinline__ int diag_of_block(int b) { return ((((b >> 6) << 3) + (b & 7)) << 3) + ((b >> 3) & 7) + 1; }
static int diag_grid(int n_used) { return ((n_used - 1) + 63) & ~63; }

// numpy.trace(M_perm, offset=off) for one candidate, by one 64-lane workgroup; the result is returned in lane 0
__device__ __forceinline__ double diag_sum_body(const double* __restrict__ M2, int64_t ld2, const IndexMap& m, int n_used,
                                                int off)
{
    __shared__ int leaf_off[MAX_LEAVES], leaf_len[MAX_LEAVES], leaf_dep[MAX_LEAVES];
    __shared__ double leaf_sum[MAX_LEAVES];
    __shared__ int s_nleaves, st_a[16], st_b[16], st_c[16];
    __shared__ double st_v[16];
    const int lane = threadIdx.x, slot = lane >> 3, k = lane & 7;
    const int len = n_used - off;
    auto elem = [&](int t) -> double { return M2[(int64_t)m.at(t) * ld2 + m.at(t + off)]; };
    double acc = 0.0;                                   // chunk results accumulate left to right from 0.0
    for (int c0 = 0; c0 < len; c0 += 8192) {
        const int clen = len - c0 < 8192 ? len - c0 : 8192;
        __syncthreads();
        if (lane == 0) s_nleaves = enumerate_leaves(c0, clen, leaf_off, leaf_len, leaf_dep, st_a, st_b, st_c);
        __syncthreads();
        const int nl = s_nleaves;
        for (int l0 = 0; l0 < nl; l0 += 8) {
            const int l = l0 + slot;
            double r = 0.0;
            int o = 0, n = 0, lim = 0;
            if (l < nl) { o = leaf_off[l]; n = leaf_len[l]; lim = n - (n % 8); }
            {
                // a leaf has at most 128 elements = 16 per lane: all index loads are issued together, then
                // all matrix loads, and only the adds are serial (two memory round trips instead of 32)
                int64_t idx[16];
                double vals[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const bool ok = l < nl && n >= 8 && q * 8 < lim;
                    const int t = o + q * 8 + k;
                    const int a = ok ? m.at(t) : 0, b = ok ? m.at(t + off) : 0;
                    idx[q] = (int64_t)a * ld2 + b;
                }
#pragma unroll
                for (int q = 0; q < 16; q++) vals[q] = (l < nl && n >= 8 && q * 8 < lim) ? M2[idx[q]] : 0.0;
                r = vals[0];
#pragma unroll
                for (int q = 1; q < 16; q++) if (q * 8 < lim) r += vals[q];
            }
            // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)): every lane of the wave takes part in the shuffles
            double s1 = r + __shfl_down(r, 1, 64);
            double s2 = s1 + __shfl_down(s1, 2, 64);
            double s3 = s2 + __shfl_down(s2, 4, 64);
            if (l < nl && k == 0) {
                double res;
                if (n < 8) { res = 0.0; for (int i = 0; i < n; i++) res += elem(o + i); }
                else { res = s3; for (int i = lim; i < n; i++) res += elem(o + i); }
                leaf_sum[l] = res;
            }
        }
        __syncthreads();
        if (lane == 0) acc += combine_leaves(nl, leaf_sum, leaf_dep, st_v, st_a);
    }
    return acc;
}

// The same diagonal as NUMBA sums it.  Under a real Numba install costFunction_numba (orderGenome.py:184-191) is
// compiled in nopython mode, where numpy.trace is Numba's own np_trace: `ret = 0; for i in range(n): ret += a[i, k + i]`,
// a strictly sequential loop - not NumPy's pairwise add.reduce.  The scores differ in the last bits (<= 1e-15 relative),
// which can decide a `cost > bestCost`.  HICMI_P2_TRACE_ORDER=numba selects this order for the candidates' scores; the
// totals (OG:343,448,506) are computed outside the jitted function and stay NumPy's.  Parity unpinned: Numba is not
// installed here; tests/golden/n160_numba was produced with that loop patched into the shimmed reference.
__device__ __forceinline__ double diag_sum_sequential(const double* __restrict__ M2, int64_t ld2, const IndexMap& m, int n_used,
                                                      int off)
{
    __shared__ double seq_buf[1024];
    const int lane = threadIdx.x, len = n_used - off;
    double acc = 0.0;
    for (int c0 = 0; c0 < len; c0 += 1024) {
        const int cnt = len - c0 < 1024 ? len - c0 : 1024;
        __syncthreads();
        for (int e = lane; e < cnt; e += 64) seq_buf[e] = M2[(int64_t)m.at(c0 + e) * ld2 + m.at(c0 + e + off)];
        __syncthreads();
        if (lane == 0) acc = serial_sum_lds(seq_buf, 0, cnt, acc);
    }
    return acc;
}

static bool numba_trace_order()
{
    const char* v = getenv("HICMI_P2_TRACE_ORDER");
    return v && (v[0] == 'n' || v[0] == 'N') && (v[1] == 'u' || v[1] == 'U') && (v[2] == 'm' || v[2] == 'M') && (v[3] == 'b' || v[3] == 'B');
}

// T[cand][off] for off = 1..n_used-1 (T[cand][0] unused).  perms == nullptr: the identity order.
__global__ __launch_bounds__(64) void k_p2_diag_sums(const double* __restrict__ M2, int64_t ld2,
                                                     const int32_t* __restrict__ perms, int n_used,
                                                     double* __restrict__ T, int sequential)
{
    const int cand = blockIdx.y, off = diag_of_block(blockIdx.x);
    if (off >= n_used) return;
    const IndexMap m = {perms ? perms + (int64_t)cand * n_used : nullptr, kNoSplice, 0, 0, 0};
    const double acc = sequential ? diag_sum_sequential(M2, ld2, m, n_used, off) : diag_sum_body(M2, ld2, m, n_used, off);
    if (threadIdx.x == 0) T[(int64_t)cand * n_used + off] = acc;
}

// The serial parts below are chains of dependent fp64 adds (the reference's order cannot be
// re-associated); everything around them is staged through LDS so the chain never waits on memory.
static constexpr int SERIAL_LDS_MAX = 8192;             // doubles staged in LDS (64 KB); longer inputs stream from L2

__global__ __launch_bounds__(256) void k_p2_total_exact(const double* __restrict__ T, int n_used, double* __restrict__ total)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    double* t = reinterpret_cast<double*>(smem_t);
    const bool staged = n_used <= SERIAL_LDS_MAX;
    if (staged) {
        for (int i = threadIdx.x; i < n_used; i += 256) t[i] = T[i];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double acc = 0.0;                                   // Python sum(): 0 + T_1 + T_2 + ...
    if (staged) acc = serial_sum_lds(t, 1, n_used, 0.0);
    else {
        for (int i = 1; i < n_used; i++) acc += T[i];
    }
    total[0] = acc;
}

// one 256-lane workgroup per candidate: the running sum of T and the final sum of the quotients are serial
// (lane 0), the two divisions per offset are done by all lanes in between.  tg: the candidate's T row;
// wg: its n_used doubles of global work space (used when the row does not fit the LDS staging area)
__device__ __forceinline__ void cost_exact_body(const double* __restrict__ tg, int n_used, double total,
                                                double* __restrict__ wg, double* __restrict__ score_out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    const bool staged = n_used <= SERIAL_LDS_MAX;
    double* w = staged ? reinterpret_cast<double*>(smem_c) : wg;
    if (staged) {
        for (int i = threadIdx.x; i < n_used; i += 256) w[i] = tg[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double cum = 0.0;
        if (staged) serial_prefix_lds(w, 1, n_used, 0.0);
        else {
            for (int i = 1; i < n_used; i++) { cum += tg[i]; w[i] = cum; }
        }
    }
    __syncthreads();
    for (int i = 1 + threadIdx.x; i < n_used; i += 256) w[i] = w[i] / total / (double)i;
    __syncthreads();
    if (threadIdx.x == 0) {
        double cost = 0.0;
        if (staged) cost = serial_sum_lds(w, 1, n_used, 0.0);
        else for (int i = 1; i < n_used; i++) cost += w[i];
        score_out[0] = cost;
    }
}

__global__ __launch_bounds__(256) void k_p2_cost_exact(const double* __restrict__ T, int n_used, double total,
                                                       double* __restrict__ work, double* __restrict__ scores)
{
    const int cand = blockIdx.x;
    cost_exact_body(T + (int64_t)cand * n_used, n_used, total, work + (int64_t)cand * n_used, scores + cand);
}

static std::atomic<int> g_lds_total{0}, g_lds_cost{0}, g_lds_score{0}, g_lds_insb_cost{0};

static size_t serial_lds_bytes(int n) { return n <= SERIAL_LDS_MAX ? (((size_t)n * sizeof(double)) + 15) & ~(size_t)15 : 16; }

void launch_p2_total(const double* M2, int64_t ld2, int n, double* T, double* total, hipStream_t s)
{
    if (n > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3(diag_grid(n), 1), dim3(64), 0, s, M2, ld2, (const int32_t*)nullptr, n, T, 0);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_total_exact), g_lds_total, 65536);
    hipLaunchKernelGGL(k_p2_total_exact, dim3(1), dim3(256), serial_lds_bytes(n), s, T, n, total);
}

void launch_p2_total_perm(const double* M2, int64_t ld2, const int32_t* d_perm, int n, double* T, double* total,
                          hipStream_t s)
{
    if (n > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3(diag_grid(n), 1), dim3(64), 0, s, M2, ld2, d_perm, n, T, 0);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_total_exact), g_lds_total, 65536);
    hipLaunchKernelGGL(k_p2_total_exact, dim3(1), dim3(256), serial_lds_bytes(n), s, T, n, total);
}

// T and work: n_cand x n_used doubles each
void launch_p2_score_exact(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, double total,
                           double* T, double* work, double* scores, hipStream_t s)
{
    if (n_cand <= 0) return;
    if (n_used > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3(diag_grid(n_used), n_cand), dim3(64), 0, s, M2, ld2, perms, n_used, T,
                                       numba_trace_order() ? 1 : 0);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_cost_exact), g_lds_cost, 65536);
    hipLaunchKernelGGL(k_p2_cost_exact, dim3(n_cand), dim3(256), serial_lds_bytes(n_used), s, T, n_used, total, work, scores);
}

// ---- lock-step insertion (k_part2_insert.hip): the same bodies, one layer of workgroups per chromosome ----
// The literal pass of a step, run only for chromosomes whose short list needs it (k_insb_shortlist):
// diagonal sums of "arrangement, then the new scaffold forward" (OG:484-487 -> OG:343: T_total, the step's total)
// and of the short-listed candidates (each workgroup walks the few of its chromosome)
__global__ __launch_bounds__(64) void k_insb_diag_cand(const InsStep* __restrict__ steps, int sequential)
{
    const InsStep& d = steps[blockIdx.y];
    if (!d.active || d.st->fail >= 0) return;
    const int ns = d.st->n_short;
    if (ns == 0) return;
    const int n_used = d.n_arr + d.L, off = diag_of_block(blockIdx.x);
    if (off >= n_used) return;
    {
        const IndexMap m = {d.pos_cur, d.n_arr, 0x3fffffff, d.new_start, 0};
        const double acc = diag_sum_body(d.M2, d.ld2, m, n_used, off);
        if (threadIdx.x == 0) d.T_total[off] = acc;
    }
    for (int q = 0; q < ns; q++) {
        const IndexMap m = {d.pos_cur, d.packed_cur[d.S + d.st->gap[q]], d.L, d.new_start, d.st->rev[q]};
        const double acc = sequential ? diag_sum_sequential(d.M2, d.ld2, m, n_used, off) : diag_sum_body(d.M2, d.ld2, m, n_used, off);
        if (threadIdx.x == 0) d.T_cand[(int64_t)q * n_used + off] = acc;
    }
}

__global__ __launch_bounds__(256) void k_insb_cost(const InsStep* __restrict__ steps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ic[];
    __shared__ double s_total;
    const InsStep& d = steps[blockIdx.y];
    if (!d.active || d.st->fail >= 0) return;
    const int cand = blockIdx.x;
    if (cand >= d.st->n_short) return;
    const int n_used = d.n_arr + d.L;
    {
        // the step's literal total: Python sum(), 0 + T_1 + T_2 + ...  (OG:343) - every candidate's workgroup
        // forms the same value
        double* t = reinterpret_cast<double*>(smem_ic);
        const bool staged = n_used <= SERIAL_LDS_MAX;
        if (staged) {
            for (int i = threadIdx.x; i < n_used; i += 256) t[i] = d.T_total[i];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            double acc = 0.0;
            if (staged) acc = serial_sum_lds(t, 1, n_used, 0.0);
            else for (int i = 1; i < n_used; i++) acc += d.T_total[i];
            s_total = acc;
            if (cand == 0) d.st->total = acc;
        }
        __syncthreads();
    }
    cost_exact_body(d.T_cand + (int64_t)cand * n_used, n_used, s_total, d.work + (int64_t)cand * n_used, d.st->lit + cand);
}

void launch_insb_diag_cand(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s)
{
    if (max_n_used < 2) return;
    hipLaunchKernelGGL(k_insb_diag_cand, dim3(diag_grid(max_n_used), n_chrom), dim3(64), 0, s, steps, numba_trace_order() ? 1 : 0);
}

void launch_insb_cost(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s)
{
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_insb_cost), g_lds_insb_cost, 65536);
    hipLaunchKernelGGL(k_insb_cost, dim3(INS_MAXC, n_chrom), dim3(256), serial_lds_bytes(max_n_used), s, steps);
}

// One workgroup (4 waves) per candidate.  Wave w takes rows a = w, w+4, ... of the candidate's
// order; its lanes sweep b = a+1 .. n-1, so consecutive lanes read M2[p(a)][p(b)] for consecutive
// positions - contiguous wherever the candidate keeps a scaffold's bins together.
__global__ __launch_bounds__(256) void k_p2_score(const double* __restrict__ M2, int64_t ld2,
                                                  const int32_t* __restrict__ perms, int n_used,
                                                  const double* __restrict__ H, double total,
                                                  double* __restrict__ scores)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    __shared__ double s_w[4];
    const int cand = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t* __restrict__ src = perms + (int64_t)cand * n_used;
    for (int i = tid; i < n_used; i += 256) p[i] = src[i];
    __syncthreads();
    const double hn = H[n_used - 1];
    double acc = 0.0;
    for (int a = wave; a < n_used - 1; a += 4) {
        const double* __restrict__ row = M2 + (int64_t)p[a] * ld2;
        for (int b = a + 1 + lane; b < n_used; b += 64) acc += row[p[b]] * (hn - H[b - a - 1]);
    }
    acc = wave_sum(acc);
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    if (tid == 0) scores[cand] = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) / total;
}

void launch_p2_score(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, const double* H,
                     double, double total, double* scores, hipStream_t s)
{
    if (n_cand <= 0) return;
    size_t lds = (((size_t)n_used * sizeof(int32_t)) + 15) & ~(size_t)15;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_score), g_lds_score, lds);
    hipLaunchKernelGGL(k_p2_score, dim3(n_cand), dim3(256), lds, s, M2, ld2, perms, n_used, H, total, scores);
}

}  // namespace hicmi
