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
//           blocks of <=128 with 8 partial sums, recursive halving, 8192-element chunks (OG:188)
//   total = Python sum of T_1..T_{n-1}, left to right                           (OG:343,448,506)
//   cost  = sum_i ((T_1+..+T_i) / total) / i, left to right                      (OG:185-191)
// One lane owns one diagonal (long and short diagonals interleaved for balance); the O(n) cum/cost
// recurrence runs on one lane per candidate.
struct DiagGather {
    const double* __restrict__ M; int64_t ld; const int32_t* __restrict__ p; int off;
    __device__ __forceinline__ double operator()(int t) const { return M[(int64_t)p[t] * ld + p[t + off]]; }
};

__device__ __forceinline__ double pw_leaf_g(const DiagGather& g, int o, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += g(o + i);
        return r;
    }
    double r0 = g(o), r1 = g(o + 1), r2 = g(o + 2), r3 = g(o + 3), r4 = g(o + 4), r5 = g(o + 5), r6 = g(o + 6),
           r7 = g(o + 7);
    int i, lim = n - (n % 8);
    for (i = 8; i < lim; i += 8) {
        r0 += g(o + i); r1 += g(o + i + 1); r2 += g(o + i + 2); r3 += g(o + i + 3);
        r4 += g(o + i + 4); r5 += g(o + i + 5); r6 += g(o + i + 6); r7 += g(o + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += g(o + i);
    return res;
}

__device__ double np_sum_g(const DiagGather& g, int n)
{
    struct Frame { int off, len, n2, stage; double left; };
    double acc = 0.0;
    for (int c = 0; c < n; c += 8192) {
        int m = n - c < 8192 ? n - c : 8192;
        Frame st[10];
        int sp = 1;
        double ret = 0.0;
        st[0] = {c, m, 0, 0, 0.0};
        while (sp > 0) {
            Frame& f = st[sp - 1];
            if (f.len <= 128) { ret = pw_leaf_g(g, f.off, f.len); sp--; continue; }
            if (f.stage == 0) {
                int n2 = f.len / 2;
                n2 -= n2 % 8;
                f.n2 = n2; f.stage = 1;
                st[sp++] = {f.off, n2, 0, 0, 0.0};
            } else if (f.stage == 1) {
                f.left = ret; f.stage = 2;
                st[sp++] = {f.off + f.n2, f.len - f.n2, 0, 0, 0.0};
            } else { ret = f.left + ret; sp--; }
        }
        acc += ret;
    }
    return acc;
}

// T[cand][i] for i = 1..n_used-1 (T[cand][0] unused).  perms == nullptr means the identity order.
__global__ __launch_bounds__(256) void k_p2_diag_sums(const double* __restrict__ M2, int64_t ld2,
                                                      const int32_t* __restrict__ perms, int n_used,
                                                      double* __restrict__ T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    const int cand = blockIdx.y;
    for (int i = threadIdx.x; i < n_used; i += 256) p[i] = perms ? perms[(int64_t)cand * n_used + i] : i;
    __syncthreads();
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n_used - 1) return;
    const int off = (d & 1) ? (n_used - 1 - (d >> 1)) : ((d >> 1) + 1);
    DiagGather g{M2, ld2, p, off};
    T[(int64_t)cand * n_used + off] = np_sum_g(g, n_used - off);
}

__global__ __launch_bounds__(64) void k_p2_total_exact(const double* __restrict__ T, int n_used, double* __restrict__ total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double acc = 0.0;                                   // Python sum(): 0 + T_1 + T_2 + ...
    for (int i = 1; i < n_used; i++) acc += T[i];
    total[0] = acc;
}

__global__ __launch_bounds__(64) void k_p2_cost_exact(const double* __restrict__ T, int n_cand, int n_used, double total,
                                                      double* __restrict__ scores)
{
    int cand = blockIdx.x * 64 + threadIdx.x;
    if (cand >= n_cand) return;
    const double* __restrict__ t = T + (int64_t)cand * n_used;
    double cum = 0.0, cost = 0.0;
    for (int i = 1; i < n_used; i++) {
        cum += t[i];
        cost += (cum / total / (double)i);
    }
    scores[cand] = cost;
}

void launch_p2_total(const double* M2, int64_t ld2, int n, double* T, double* total, hipStream_t s)
{
    size_t lds = (((size_t)n * sizeof(int32_t)) + 15) & ~(size_t)15;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_p2_diag_sums), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (n > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3((n - 1 + 255) / 256, 1), dim3(256), lds, s, M2, ld2,
                                  (const int32_t*)nullptr, n, T);
    hipLaunchKernelGGL(k_p2_total_exact, dim3(1), dim3(64), 0, s, T, n, total);
}

void launch_p2_total_perm(const double* M2, int64_t ld2, const int32_t* d_perm, int n, double* T, double* total,
                          hipStream_t s)
{
    size_t lds = (((size_t)n * sizeof(int32_t)) + 15) & ~(size_t)15;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_p2_diag_sums), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (n > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3((n - 1 + 255) / 256, 1), dim3(256), lds, s, M2, ld2, d_perm, n, T);
    hipLaunchKernelGGL(k_p2_total_exact, dim3(1), dim3(64), 0, s, T, n, total);
}

void launch_p2_score_exact(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, double total,
                           double* T, double* scores, hipStream_t s)
{
    if (n_cand <= 0) return;
    size_t lds = (((size_t)n_used * sizeof(int32_t)) + 15) & ~(size_t)15;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_p2_diag_sums), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (n_used > 1) hipLaunchKernelGGL(k_p2_diag_sums, dim3((n_used - 1 + 255) / 256, n_cand), dim3(256), lds, s, M2, ld2,
                                       perms, n_used, T);
    hipLaunchKernelGGL(k_p2_cost_exact, dim3((n_cand + 63) / 64), dim3(64), 0, s, T, n_cand, n_used, total, scores);
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
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_p2_score), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_p2_score, dim3(n_cand), dim3(256), lds, s, M2, ld2, perms, n_used, H, total, scores);
}

}  // namespace hicmi
