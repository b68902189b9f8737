// k_part1.hip - gfx950 kernels for Part 1 (scaffoldToChromosomes.py): row sums, distance
// transform, UPGMA nearest-neighbour chain, cut-scan counts and hypergeometric flags.
//
// Everything here is fp64 and must reproduce the reference's CPU arithmetic bit for bit, so the
// whole library is compiled with -ffp-contract=off and without fast-math: a*b+c stays two
// roundings, divisions are the correctly rounded v_div_scale/fmas/fixup sequence.
#include "hicmi_internal.h"
#include "hyper.h"

namespace hicmi {

// =================================================================================================
// Row sums (scaffoldToChromosomes.py:112,134,147).
// NumPy's float64 add.reduce: <=128-element blocks with 8 partial sums, recursive halving above
// that, inner loop handed at most 8192 elements at a time, chunk results accumulated left to right
// from 0.0 (restated from NumPy 2.2.6; pinned in tests/test_oracle_cpu.py).
// NumPy's pairwise row sum, one 64-lane workgroup per row: the leaves (<= 128 elements) of a chunk are taken eight at
// a time, lane (slot, k) = (lane / 8, lane % 8) holding partial sum k of leaf `slot` - consecutive lanes read
// consecutive cells - then ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) by shuffles and the tree over the leaf sums by lane 0
// (stacks in LDS).  The earlier one-lane-per-row version walked the tree with a private stack, i.e. in scratch memory.
__global__ __launch_bounds__(64) void k_row_sums_np(const double* __restrict__ C, int64_t ldc, int n,
                                                    double* __restrict__ np_sum, int row_first, int row_stride)
{
    __shared__ int leaf_off[MAX_LEAVES], leaf_len[MAX_LEAVES], leaf_dep[MAX_LEAVES];
    __shared__ double leaf_sum[MAX_LEAVES];
    __shared__ int s_nleaves, st_a[16], st_b[16], st_c[16];
    __shared__ double st_v[16];
    const int row = row_first + blockIdx.x * row_stride, lane = threadIdx.x, slot = lane >> 3, k = lane & 7;   // one of this shard's rows
    const double* __restrict__ a = C + (int64_t)row * ldc;
    double acc = 0.0;                                   // chunk results accumulate left to right from 0.0
    for (int c0 = 0; c0 < n; c0 += 8192) {
        const int clen = n - c0 < 8192 ? n - c0 : 8192;
        __syncthreads();
        if (lane == 0) s_nleaves = enumerate_leaves(c0, clen, leaf_off, leaf_len, leaf_dep, st_a, st_b, st_c);
        __syncthreads();
        const int nl = s_nleaves;
        for (int l0 = 0; l0 < nl; l0 += 8) {
            const int l = l0 + slot;
            int o = 0, m = 0, lim = 0;
            if (l < nl) { o = leaf_off[l]; m = leaf_len[l]; lim = m - (m % 8); }
            double vals[16];
#pragma unroll
            for (int q = 0; q < 16; q++) vals[q] = (l < nl && m >= 8 && q * 8 < lim) ? a[o + q * 8 + k] : 0.0;
            double r = vals[0];
#pragma unroll
            for (int q = 1; q < 16; q++) if (q * 8 < lim) r += vals[q];
            double s1 = r + __shfl_down(r, 1, 64);
            double s2 = s1 + __shfl_down(s1, 2, 64);
            double s3 = s2 + __shfl_down(s2, 4, 64);
            if (l < nl && k == 0) {
                double res;
                if (m < 8) { res = 0.0; for (int i = 0; i < m; i++) res += a[o + i]; }
                else { res = s3; for (int i = lim; i < m; i++) res += a[o + i]; }
                leaf_sum[l] = res;
            }
        }
        __syncthreads();
        if (lane == 0) acc += combine_leaves(nl, leaf_sum, leaf_dep, st_v, st_a);
    }
    if (lane == 0) np_sum[row] = acc;
}

// Python's builtin sum(): strictly left to right, one lane per row (the chain cannot be split); 16-byte loads, eight
// cells fetched ahead of the adds.
__global__ __launch_bounds__(64) void k_row_sums_seq(const double* __restrict__ C, int64_t ldc, int n,
                                                     double* __restrict__ seq_sum, int row_first, int row_stride)
{
    const int row = row_first + (blockIdx.x * 64 + threadIdx.x) * row_stride;
    if (row >= n) return;
    const double* __restrict__ a = C + (int64_t)row * ldc;          // ldc is a multiple of 2 cells or the row is read cell-wise
    double s = 0.0;
    int i = 0;
    if ((((uintptr_t)a) & 15u) == 0) {
        for (; i + 8 <= n; i += 8) {
            const double2 v0 = *reinterpret_cast<const double2*>(a + i), v1 = *reinterpret_cast<const double2*>(a + i + 2);
            const double2 v2 = *reinterpret_cast<const double2*>(a + i + 4), v3 = *reinterpret_cast<const double2*>(a + i + 6);
            s += v0.x; s += v0.y; s += v1.x; s += v1.y; s += v2.x; s += v2.y; s += v3.x; s += v3.y;
        }
    }
    for (; i < n; i++) s += a[i];
    seq_sum[row] = s;
}

// rows row_first, row_first + row_stride, ... (the whole matrix for 0, 1)
void launch_row_sums(const double* C, int64_t ldc, int n, double* np_sum, double* seq_sum, int row_first, int row_stride,
                     hipStream_t s)
{
    const int mine = n > row_first ? (n - row_first + row_stride - 1) / row_stride : 0;
    if (mine <= 0) return;
    hipLaunchKernelGGL(k_row_sums_np, dim3(mine), dim3(64), 0, s, C, ldc, n, np_sum, row_first, row_stride);
    hipLaunchKernelGGL(k_row_sums_seq, dim3((mine + 63) / 64), dim3(64), 0, s, C, ldc, n, seq_sum, row_first, row_stride);
}

// removeRows (S2C:100-136): dst = src[keep][:, keep]
__global__ __launch_bounds__(256) void k_compact(const double* __restrict__ src, int64_t ld_src,
                                                 const int32_t* __restrict__ keep, int n_keep,
                                                 double* __restrict__ dst, int64_t ld_dst)
{
    int r = blockIdx.y;
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_keep) return;
    dst[(int64_t)r * ld_dst + c] = src[(int64_t)keep[r] * ld_src + keep[c]];
}

void launch_compact(const double* src, int64_t ld_src, const int32_t* keep, int n_keep, double* dst, int64_t ld_dst,
                    hipStream_t s)
{
    hipLaunchKernelGGL(k_compact, dim3((n_keep + 255) / 256, n_keep), dim3(256), 0, s, src, ld_src, keep, n_keep, dst,
                       ld_dst);
}

// =================================================================================================
// Distance transform + symmetric working matrix (S2C:147 + squareform's upper-triangle read,
// S2C:194):  for i < j   W[i][j] = W[j][i] = (1. - C[i][j] / rowsum_i) + 1.
// The lower triangle of the reference's (asymmetric) distance matrix never reaches SciPy, so it
// is never computed.  Diagonal and padding columns hold +inf.
__global__ __launch_bounds__(256) void k_build_w(const double* __restrict__ C, int64_t ldc,
                                                 const double* __restrict__ np_sum, int n,
                                                 double* __restrict__ W, int64_t ldw)
{
    __shared__ double tile[64][65];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const double inf = __builtin_inf();
    for (int i = ty; i < 64; i += 4) {
        int gi = bi * 64 + i, gj = bj * 64 + tx;
        double v = inf;
        if (gi < n && gj < n && gi < gj) v = (1.0 - (C[(int64_t)gi * ldc + gj] / np_sum[gi])) + 1.0;
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int gi = bi * 64 + i, gj = bj * 64 + tx;
        if (gi < n && gj < ldw) {
            double v = tile[i][tx];
            if (bi == bj && gi > gj) v = tile[tx][i];          // mirror inside a diagonal tile
            W[(int64_t)gi * ldw + gj] = v;
        }
        if (bi != bj) {                                         // mirrored tile: rows of bj, columns of bi
            int ri = bj * 64 + i, rj = bi * 64 + tx;
            if (ri < n && rj < ldw) W[(int64_t)ri * ldw + rj] = tile[tx][i];
        }
    }
}

void launch_build_w(const double* C, int64_t ldc, const double* np_sum, int n, double* W, int64_t ldw, hipStream_t s)
{
    int tc = (int)((ldw + 63) / 64), tr = (n + 63) / 64;
    hipLaunchKernelGGL(k_build_w, dim3(tc, tr), dim3(256), 0, s, C, ldc, np_sum, n, W, ldw);
}

// =================================================================================================
// Cut-scan counts.  With rank[i][j] = position of column j in row i's descending similarity order,
//   #{ v in R[i][0:L] : lo <= v <= hi }  ==  #{ j in [lo, hi] : rank[i][j] < L }
// so every query of scaffoldToChromosomes.py:455-459 (mode 0: hi = i, L = i - lo) and :631
// (mode 1: hi = c, L = c - lo) is a count over a CONTIGUOUS uint16 segment of one row:
// coalesced 16-byte loads, wave-shuffle + LDS reduction.  HBM-bound: 2 bytes per element read.
__global__ __launch_bounds__(256) void k_cut_count(const uint16_t* __restrict__ rank, int64_t ldr, int row0, int lo,
                                                   int mode, int cparam, int32_t* __restrict__ x_out, int row_step)
{
    __shared__ int s_part[4];
    const int i = row0 + blockIdx.x * row_step;             // row_step > 1: this shard's rows only
    const int hi = mode == 0 ? i : cparam;
    const int thr = hi - lo;
    const uint16_t* __restrict__ r = rank + (int64_t)i * ldr;
    const int tid = threadIdx.x;
    int cnt = 0;
    const int end = hi + 1;                       // half-open [lo, end)
    int body0 = (lo + 7) & ~7;                    // first 16-byte aligned element
    if (body0 > end) body0 = end;
    const int body1 = body0 + ((end - body0) & ~7);
    for (int j = lo + tid; j < body0; j += 256) cnt += (int)r[j] < thr;
    for (int j = body0 + tid * 8; j < body1; j += 256 * 8) {
        uint4 q = *reinterpret_cast<const uint4*>(r + j);
        cnt += (int)(q.x & 0xffffu) < thr; cnt += (int)(q.x >> 16) < thr;
        cnt += (int)(q.y & 0xffffu) < thr; cnt += (int)(q.y >> 16) < thr;
        cnt += (int)(q.z & 0xffffu) < thr; cnt += (int)(q.z >> 16) < thr;
        cnt += (int)(q.w & 0xffffu) < thr; cnt += (int)(q.w >> 16) < thr;
    }
    for (int j = body1 + tid; j < end; j += 256) cnt += (int)r[j] < thr;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) x_out[blockIdx.x * row_step] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// x_out[t] = count of row row0 + t, for t = 0, row_step, 2 row_step, ... < nrows (entries in between are not written)
void launch_cut_count(const uint16_t* rank, int64_t ldr, int row0, int nrows, int lo, int mode, int cparam,
                      int32_t* x_out, int row_step, hipStream_t s)
{
    if (nrows <= 0) return;
    const int blocks = (nrows + row_step - 1) / row_step;
    hipLaunchKernelGGL(k_cut_count, dim3(blocks), dim3(256), 0, s, rank, ldr, row0, lo, mode, cparam, x_out, row_step);
}

// sig flags from counts.  mode 0 (first pass, S2C:455-469): entry t >= 1 tests L = t, NaN -> 1,
// entry 0 is forced to 0.  mode 1 (filter, S2C:631-636): every entry tests L = L_fixed, NaN -> 0.
__global__ __launch_bounds__(256) void k_hyper_flags(const int32_t* __restrict__ x, int nrows, int mode, int L_fixed,
                                                     int64_t M, double psig, uint8_t* __restrict__ sig, int own_first,
                                                     int own_step)
{
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nrows) return;
    if (own_step > 1 && (t < own_first || (t - own_first) % own_step != 0)) { sig[t] = 0; return; }   // another shard's row
    if (mode == 0) {
        if (t == 0) { sig[0] = 0; return; }
        const int dec = hypergeom_decide((int64_t)x[t], M, (int64_t)t, (int64_t)t, psig);
        sig[t] = dec == 0 ? 0 : 1;                              // `pval >= psig` -> 0, NaN -> 1 (S2C:466-469)
    } else {
        const int dec = hypergeom_decide((int64_t)x[t], M, (int64_t)L_fixed, (int64_t)L_fixed, psig);
        sig[t] = dec == 1 ? 1 : 0;                              // `pval < psig` -> 1, NaN -> 0 (S2C:633-636)
    }
}

// own_step > 1: only entries own_first, own_first + own_step, ... are tested, the others are written as 0
void launch_hyper_flags(const int32_t* x, int nrows, int mode, int L_fixed, int64_t M, double psig, uint8_t* sig,
                        int own_first, int own_step, hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_hyper_flags, dim3((nrows + 255) / 256), dim3(256), 0, s, x, nrows, mode, L_fixed, M, psig, sig,
                       own_first, own_step);
}

// fp32 -> fp64 in place.  dst has `cells` doubles; the fp32 image sits in the upper half of the same bytes
// (src = (float*)dst + cells).  Output cell i covers the bytes of input cells 2i - cells and 2i - cells + 1 - none
// for i < cells / 2, and always cells with a smaller index than i (or i itself, read by the same lane first).  So
// ascending order is safe, and a chunk [a, b) may run in parallel when everything it overwrites lies below a:
// 2b - cells - 1 < a, i.e. b <= (a + cells) / 2.  One launch per chunk (stream order between chunks); the chunks
// halve the remaining distance, about log2(cells) launches.
__global__ __launch_bounds__(256) void k_widen_f32(const float* __restrict__ src, double* __restrict__ dst, int64_t i0, int64_t i1)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < i1; i += stride) dst[i] = (double)src[i];
}

void launch_widen_f32(const float* src, double* dst, int64_t cells, hipStream_t s)
{
    int64_t a = 0;
    while (a < cells) {
        int64_t b = a < cells / 2 ? cells / 2 : (a + cells) / 2;
        if (b <= a) b = a + 1;
        if (b > cells) b = cells;
        const int64_t work = b - a;
        int blocks = (int)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096);
        hipLaunchKernelGGL(k_widen_f32, dim3(blocks), dim3(256), 0, s, src, dst, a, b);
        a = b;
    }
}

}  // namespace hicmi
