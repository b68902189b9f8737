// k_plot.hip - device side of plot emission (plotContactMaps.py:15-91).
//
// The reference hands the whole N x N matrix to numpy.percentile (colour limits) and to pcolormesh
// (one quad per cell).  A figure has a few thousand pixels per side, so here the matrix never leaves the
// GPU at full size:
//   * the two colour limits are EXACT order statistics of the N^2 cells, found by a most-significant-digit
//     radix select over monotone 64-bit keys (6 histogram passes of 11 bits, all targets at once), and
//     interpolated on the host exactly like numpy.percentile(method="linear");
//   * the picture is the block mean of the (transformed, permuted) matrix at figure resolution.
// kind selects what the reference plots at that point: 0 raw contacts (Part 2), 1 the distance transform
// (S2C:147, first Part 1 plot), 2 the similarity transform (S2C:149, outlined Part 1 plot) - the same
// three-rounding expressions as k_build_w / k_sort_rows.
#include "hicmi_internal.h"

namespace hicmi {

__device__ __forceinline__ double plot_value(const double* __restrict__ C, int64_t ldc, const double* __restrict__ np_sum,
                                             const double* __restrict__ seq_sum, int kind, int i, int j)
{
    const double c = C[(int64_t)i * ldc + j];
    if (kind == 0) return c;
    const double d = (1.0 - c / np_sum[i]) + 1.0;
    if (kind == 1) return d;
    return seq_sum[i] * (1.0 - (d - 1.0));
}

__device__ __forceinline__ unsigned long long key_of(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // ascending keys <=> ascending doubles
}

__host__ __device__ inline double value_of_key(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    union { unsigned long long u; double d; } cv;
    cv.u = b;
    return cv.d;
}

static constexpr int PLOT_T = 4;                       // order statistics selected together (2 percentiles x floor/ceil)
static constexpr int PLOT_BINS = 2048;                 // 11-bit digits

struct SelectState {
    unsigned long long prefix[PLOT_T];                 // key bits fixed so far (right-aligned)
    unsigned long long rank[PLOT_T];                   // rank still to resolve inside the prefix class
};

// histogram of the next digit among the cells whose high bits equal each target's prefix
__global__ __launch_bounds__(256) void k_plot_hist(const double* __restrict__ C, int64_t ldc,
                                                   const double* __restrict__ np_sum, const double* __restrict__ seq_sum,
                                                   int kind, const int32_t* __restrict__ order, int n_sel, int n_targets,
                                                   int shift, int nbits, const SelectState* __restrict__ st,
                                                   unsigned int* __restrict__ hist)
{
    __shared__ unsigned int lh[PLOT_T * PLOT_BINS];
    __shared__ unsigned long long pre[PLOT_T];
    const int tid = threadIdx.x;
    for (int i = tid; i < n_targets * PLOT_BINS; i += 256) lh[i] = 0u;
    if (tid < n_targets) pre[tid] = st->prefix[tid];
    __syncthreads();
    const int hi_shift = shift + nbits;                // bits above the digit; 64 on the first pass
    const unsigned long long mask = (1ull << nbits) - 1ull;
    for (int a = blockIdx.x; a < n_sel; a += gridDim.x) {
        const int i = order ? order[a] : a;
        for (int b = tid; b < n_sel; b += 256) {
            const int j = order ? order[b] : b;
            const unsigned long long k = key_of(plot_value(C, ldc, np_sum, seq_sum, kind, i, j));
            const unsigned long long high = hi_shift >= 64 ? 0ull : (k >> hi_shift);
            const unsigned int digit = (unsigned int)((k >> shift) & mask);
            for (int t = 0; t < n_targets; t++)
                if (high == pre[t]) atomicAdd(&lh[t * PLOT_BINS + digit], 1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < n_targets * PLOT_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// per target: the digit whose cumulative count passes the rank; extends the prefix, reduces the rank
__global__ __launch_bounds__(64) void k_plot_pick(int n_targets, int nbits, SelectState* st, unsigned int* __restrict__ hist)
{
    const int t = threadIdx.x;
    if (t < n_targets) {
        const int bins = 1 << nbits;
        unsigned long long r = st->rank[t], cum = 0ull;
        int digit = bins - 1;
        for (int d = 0; d < bins; d++) {
            const unsigned long long c = hist[t * PLOT_BINS + d];
            if (r < cum + c) { digit = d; break; }
            cum += c;
        }
        st->rank[t] = r - cum;
        st->prefix[t] = (st->prefix[t] << nbits) | (unsigned long long)digit;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_targets * PLOT_BINS; i += 64) hist[i] = 0u;    // ready for the next pass
}

// d_state: SelectState with rank[] filled and prefix[] = 0; d_hist: PLOT_T * PLOT_BINS zeroed counters.
// On return (after the stream drains) prefix[t] is the key of the rank[t]-th smallest cell.
void launch_plot_select(const double* C, int64_t ldc, const double* np_sum, const double* seq_sum, int kind,
                        const int32_t* order, int n_sel, int n_targets, SelectState* d_state, unsigned int* d_hist,
                        hipStream_t s)
{
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    const int widths[6] = {11, 11, 11, 11, 11, 9};
    const int blocks = n_sel < 2048 ? n_sel : 2048;
    for (int p = 0; p < 6; p++) {
        hipLaunchKernelGGL(k_plot_hist, dim3(blocks), dim3(256), 0, s, C, ldc, np_sum, seq_sum, kind, order, n_sel, n_targets,
                           shifts[p], widths[p], d_state, d_hist);
        hipLaunchKernelGGL(k_plot_pick, dim3(1), dim3(64), 0, s, n_targets, widths[p], d_state, d_hist);
    }
}

size_t plot_select_state_bytes() { return sizeof(SelectState); }
size_t plot_select_hist_bytes() { return sizeof(unsigned int) * PLOT_T * PLOT_BINS; }
int plot_select_max_targets() { return PLOT_T; }
void plot_select_fill(void* host_state, const unsigned long long* ranks, int n_targets)
{
    SelectState* st = reinterpret_cast<SelectState*>(host_state);
    for (int t = 0; t < PLOT_T; t++) { st->prefix[t] = 0ull; st->rank[t] = t < n_targets ? ranks[t] : 0ull; }
}
double plot_select_value(const void* host_state, int t)
{
    return value_of_key(reinterpret_cast<const SelectState*>(host_state)->prefix[t]);
}

// out[r][c] = mean of the cells of block (r, c); blocks split [0, n_sel) into px nearly equal parts
__global__ __launch_bounds__(256) void k_plot_downsample(const double* __restrict__ C, int64_t ldc,
                                                         const double* __restrict__ np_sum,
                                                         const double* __restrict__ seq_sum, int kind,
                                                         const int32_t* __restrict__ order, int n_sel, int px,
                                                         double* __restrict__ out)
{
    const int r = blockIdx.x;
    const int a0 = (int)(((int64_t)r * n_sel) / px), a1 = (int)(((int64_t)(r + 1) * n_sel) / px);
    for (int c = threadIdx.x; c < px; c += 256) {
        const int b0 = (int)(((int64_t)c * n_sel) / px), b1 = (int)(((int64_t)(c + 1) * n_sel) / px);
        double acc = 0.0;
        for (int a = a0; a < a1; a++) {
            const int i = order ? order[a] : a;
            for (int b = b0; b < b1; b++) acc += plot_value(C, ldc, np_sum, seq_sum, kind, i, order ? order[b] : b);
        }
        const int cnt = (a1 - a0) * (b1 - b0);
        out[(int64_t)r * px + c] = cnt > 0 ? acc / (double)cnt : 0.0;
    }
}

void launch_plot_downsample(const double* C, int64_t ldc, const double* np_sum, const double* seq_sum, int kind,
                            const int32_t* order, int n_sel, int px, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_plot_downsample, dim3(px), dim3(256), 0, s, C, ldc, np_sum, seq_sum, kind, order, n_sel, px, out);
}

}  // namespace hicmi
