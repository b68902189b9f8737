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

// per-row sums of the strict upper triangle, then one workgroup adds the rows in index order
__global__ __launch_bounds__(256) void k_p2_row_upper(const double* __restrict__ M2, int64_t ld2, int n,
                                                      double* __restrict__ partial)
{
    __shared__ double s_w[4];
    int r = blockIdx.x;
    const double* __restrict__ row = M2 + (int64_t)r * ld2;
    double acc = 0.0;
    for (int c = r + 1 + threadIdx.x; c < n; c += 256) acc += row[c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[r] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void k_p2_sum(const double* __restrict__ partial, int n, double* __restrict__ total)
{
    __shared__ double s_w[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) total[0] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

void launch_p2_total(const double* M2, int64_t ld2, int n, double* partial, double* total, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_p2_row_upper, dim3(n), dim3(256), 0, s, M2, ld2, n, partial);
    hipLaunchKernelGGL(k_p2_sum, dim3(1), dim3(256), 0, s, partial, n, total);
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
