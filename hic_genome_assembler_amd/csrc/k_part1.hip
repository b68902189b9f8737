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
__device__ __forceinline__ double pw_leaf(const double* __restrict__ a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i, lim = n - (n % 8);
    for (i = 8; i < lim; i += 8) {
        r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}

__device__ double pw_chunk(const double* __restrict__ a, int n)   // n <= 8192
{
    struct Frame { int off, len, n2, stage; double left; };
    Frame st[10];
    int sp = 1;
    double ret = 0.0;
    st[0] = {0, n, 0, 0, 0.0};
    while (sp > 0) {
        Frame& f = st[sp - 1];
        if (f.len <= 128) { ret = pw_leaf(a + f.off, f.len); sp--; continue; }
        if (f.stage == 0) {
            int n2 = f.len / 2;
            n2 -= n2 % 8;
            f.n2 = n2; f.stage = 1;
            st[sp++] = {f.off, n2, 0, 0, 0.0};
        } else if (f.stage == 1) {
            f.left = ret; f.stage = 2;
            st[sp++] = {f.off + f.n2, f.len - f.n2, 0, 0, 0.0};
        } else {
            ret = f.left + ret;
            sp--;
        }
    }
    return ret;
}

// One lane per row; a lane streams its own row, so every fetched line is fully consumed by it.
__global__ __launch_bounds__(64) void k_row_sums(const double* __restrict__ C, int64_t ldc, int n,
                                                 double* __restrict__ np_sum, double* __restrict__ seq_sum)
{
    int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= n) return;
    const double* a = C + (int64_t)row * ldc;
    double acc = 0.0;
    for (int c = 0; c < n; c += 8192) {
        int m = n - c < 8192 ? n - c : 8192;
        acc += pw_chunk(a + c, m);
    }
    np_sum[row] = acc;
    double s = 0.0;                       // builtin sum(): strictly left to right
    for (int i = 0; i < n; i++) s += a[i];
    seq_sum[row] = s;
}

void launch_row_sums(const double* C, int64_t ldc, int n, double* np_sum, double* seq_sum, hipStream_t s)
{
    hipLaunchKernelGGL(k_row_sums, dim3((n + 63) / 64), dim3(64), 0, s, C, ldc, n, np_sum, seq_sum);
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
// UPGMA by nearest-neighbour chain - SciPy's _hierarchy.nn_chain for method='average'
// (scaffoldToChromosomes.py:197; algorithm restated in SURVEY.md A3 and oracle/oracle_c.c).
//
// The algorithm is a chain of ~3(n-1) DEPENDENT O(n) steps, each far too small to amortise a
// grid-wide barrier (a 32k-bin row is 256 KB; an XCD-hierarchical grid barrier costs ~5 us, about
// what one CU needs to stream the row), so it runs as ONE persistent 1024-lane workgroup:
//   scan  : lanes stream row x of W with 16-byte loads, keep (min, lowest index) per lane,
//           wave-shuffle reduce, 16-wave LDS reduce; strict '<' + index order == SciPy's tie rule,
//           and the previous chain element is preferred exactly as SciPy does;
//   merge : Lance-Williams (nx*d_xi + ny*d_yi)/(nx+ny) with five separate fp64 roundings, row y
//           rewritten with coalesced stores and column y scattered so W stays symmetric.
// Liveness (bitmask) and cluster sizes live in LDS; the chain lives in global memory with its top
// 256 entries mirrored in LDS (lane 0 only).
struct ArgMin { double v; int i; };

__device__ __forceinline__ ArgMin argmin_wave(ArgMin a)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(a.v, off, 64);
        int oi = __shfl_xor(a.i, off, 64);
        if (ov < a.v || (ov == a.v && oi < a.i)) { a.v = ov; a.i = oi; }
    }
    return a;
}

// PROFILE = true adds wall-clock stamps (100 MHz) around the phases; lane 0 writes the totals to
// prof[0..4] = {chain bookkeeping, row scan, pick neighbour, merge bookkeeping, Lance-Williams update}.
template <bool PROFILE>
__global__ __launch_bounds__(1024) void k_nnchain(double* __restrict__ W, int64_t ld, int n, int* __restrict__ chain,
                                                  double* __restrict__ zraw, int* __restrict__ status,
                                                  unsigned long long* __restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    uint32_t* alive = reinterpret_cast<uint32_t*>(smem_nn);
    const int nwords = (n + 31) >> 5;
    uint16_t* lsize = reinterpret_cast<uint16_t*>(alive + ((nwords + 3) & ~3));   // cluster sizes (<= 65535 until the last merge)
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int ring[256];                            // top of the chain (chain[i] lives at ring[i & 255])
    __shared__ double s_dprev;
    __shared__ int s_x, s_prev, s_done, s_stop, s_mx, s_my, s_nx, s_ny;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < nwords; w += 1024) {
        int rem = n - w * 32;
        alive[w] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    }
    for (int i = tid; i < n; i += 1024) lsize[i] = 1;
    if (tid == 0) { s_stop = 0; s_done = 0; }
    // lane-0 private chain state
    int len = 0, top = -1, second = -1, first_ptr = 0, ring_lo = 0;
    unsigned long long t_book = 0, t_scan = 0, t_pick = 0, t_merge = 0, t_upd = 0, t0 = 0, t1 = 0;
    __syncthreads();

    for (int step = 0; step < n - 1; step++) {
        if (PROFILE && tid == 0) t0 = wall_clock64();
        if (tid == 0 && len == 0) {
            while (first_ptr < n && !((alive[first_ptr >> 5] >> (first_ptr & 31)) & 1u)) first_ptr++;
            chain[0] = first_ptr; ring[0] = first_ptr; ring_lo = 0; top = first_ptr; second = -1; len = 1;
        }
        int guard = 0;
        double cur = 0.0;
        int ybest = -1;
        while (true) {
            if (tid == 0) { s_x = top; s_prev = (len > 1) ? second : -1; }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_book += t1 - t0; t0 = t1; }
            const int x = s_x, prev = s_prev;
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            ArgMin best = {__builtin_inf(), 0x7fffffff};
#pragma unroll 4
            for (int j = tid * 2; j < n; j += 2048) {
                double2 v = *reinterpret_cast<const double2*>(rowx + j);
                uint32_t bits = alive[j >> 5] >> (j & 31);          // j is even: both bits in one word
                if ((bits & 1u) && j != x && v.x < best.v) { best.v = v.x; best.i = j; }
                if ((bits & 2u) && j + 1 != x && j + 1 < n && v.y < best.v) { best.v = v.y; best.i = j + 1; }
                if ((prev | 1) == (j | 1) && prev >= 0) s_dprev = (prev & 1) ? v.y : v.x;   // d(x, previous chain element)
            }
            best = argmin_wave(best);
            if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_scan += t1 - t0; t0 = t1; }
            if (tid == 0) {
                ArgMin m = {s_v[0], s_i[0]};
                for (int w = 1; w < 16; w++)
                    if (s_v[w] < m.v || (s_v[w] == m.v && s_i[w] < m.i)) { m.v = s_v[w]; m.i = s_i[w]; }
                int y; double c;
                if (prev >= 0) {
                    double dprev = s_dprev;
                    if (m.v < dprev) { y = m.i; c = m.v; } else { y = prev; c = dprev; }
                } else { y = m.i; c = m.v; }
                int done = (prev >= 0 && y == prev);
                if (y < 0 || y >= n || ++guard > n + 2) { s_stop = 1; done = 1; }
                else if (!done) {
                    chain[len] = y; ring[len & 255] = y;
                    if (len - 255 > ring_lo) ring_lo = len - 255;
                    second = top; top = y; len++;
                }
                cur = c; ybest = y;
                s_done = done;
            }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_pick += t1 - t0; t0 = t1; }
            if (s_done) break;
        }
        if (s_stop) break;
        if (tid == 0) {
            int xx = s_x, yy = ybest;
            len -= 2;
            if (xx > yy) { int t = xx; xx = yy; yy = t; }
            int nx = lsize[xx], ny = lsize[yy];
            zraw[4 * step + 0] = (double)xx;
            zraw[4 * step + 1] = (double)yy;
            zraw[4 * step + 2] = cur;
            zraw[4 * step + 3] = (double)(nx + ny);
            lsize[xx] = 0;
            lsize[yy] = (uint16_t)(nx + ny);
            alive[xx >> 5] &= ~(1u << (xx & 31));
            s_mx = xx; s_my = yy; s_nx = nx; s_ny = ny;
            top = len >= 1 ? (len - 1 >= ring_lo ? ring[(len - 1) & 255] : chain[len - 1]) : -1;
            second = len >= 2 ? (len - 2 >= ring_lo ? ring[(len - 2) & 255] : chain[len - 2]) : -1;
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_merge += t1 - t0; t0 = t1; }
        {
            const int mx = s_mx, my = s_my;
            const double fx = (double)s_nx, fy = (double)s_ny, fs = (double)(s_nx + s_ny);
            const double* __restrict__ rx = W + (int64_t)mx * ld;
            double* __restrict__ ry = W + (int64_t)my * ld;
#pragma unroll 2
            for (int j = tid * 2; j < n; j += 2048) {
                double2 a = *reinterpret_cast<const double2*>(rx + j);
                double2 b = *reinterpret_cast<const double2*>(ry + j);
                uint32_t bits = alive[j >> 5] >> (j & 31);
                if ((bits & 1u) && j != my) {
                    double v = (fx * a.x + fy * b.x) / fs;
                    b.x = v;
                    W[(int64_t)j * ld + my] = v;
                }
                if ((bits & 2u) && j + 1 != my && j + 1 < n) {
                    double v = (fx * a.y + fy * b.y) / fs;
                    b.y = v;
                    W[(int64_t)(j + 1) * ld + my] = v;
                }
                *reinterpret_cast<double2*>(ry + j) = b;
            }
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_upd += t1 - t0; }
    }
    if (tid == 0) {
        status[0] = s_stop;
        if (PROFILE) { prof[0] = t_book; prof[1] = t_scan; prof[2] = t_pick; prof[3] = t_merge; prof[4] = t_upd; }
    }
}

void launch_nnchain(double* W, int64_t ldw, int n, int* chain, double* zraw, int* status, unsigned long long* prof,
                    hipStream_t s)
{
    size_t lds = sizeof(uint32_t) * (size_t)((((n + 31) / 32) + 3) & ~3) + sizeof(uint16_t) * (size_t)n;
    lds = (lds + 15) & ~(size_t)15;
    if (prof) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nnchain<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_nnchain<true>, dim3(1), dim3(1024), lds, s, W, ldw, n, chain, zraw, status, prof);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nnchain<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_nnchain<false>, dim3(1), dim3(1024), lds, s, W, ldw, n, chain, zraw, status, prof);
    }
}

// =================================================================================================
// Cut-scan counts.  With rank[i][j] = position of column j in row i's descending similarity order,
//   #{ v in R[i][0:L] : lo <= v <= hi }  ==  #{ j in [lo, hi] : rank[i][j] < L }
// so every query of scaffoldToChromosomes.py:455-459 (mode 0: hi = i, L = i - lo) and :631
// (mode 1: hi = c, L = c - lo) is a count over a CONTIGUOUS uint16 segment of one row:
// coalesced 16-byte loads, wave-shuffle + LDS reduction.  HBM-bound: 2 bytes per element read.
__global__ __launch_bounds__(256) void k_cut_count(const uint16_t* __restrict__ rank, int64_t ldr, int row0, int lo,
                                                   int mode, int cparam, int32_t* __restrict__ x_out)
{
    __shared__ int s_part[4];
    const int i = row0 + blockIdx.x;
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
    if (tid == 0) x_out[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

void launch_cut_count(const uint16_t* rank, int64_t ldr, int row0, int nrows, int lo, int mode, int cparam,
                      int32_t* x_out, hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_cut_count, dim3(nrows), dim3(256), 0, s, rank, ldr, row0, lo, mode, cparam, x_out);
}

// sig flags from counts.  mode 0 (first pass, S2C:455-469): entry t >= 1 tests L = t, NaN -> 1,
// entry 0 is forced to 0.  mode 1 (filter, S2C:631-636): every entry tests L = L_fixed, NaN -> 0.
__global__ __launch_bounds__(256) void k_hyper_flags(const int32_t* __restrict__ x, int nrows, int mode, int L_fixed,
                                                     int64_t M, double psig, uint8_t* __restrict__ sig)
{
    int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nrows) return;
    if (mode == 0) {
        if (t == 0) { sig[0] = 0; return; }
        double p = hypergeom_sf_ge((int64_t)x[t], M, (int64_t)t, (int64_t)t);
        sig[t] = (p >= psig) ? 0 : 1;
    } else {
        double p = hypergeom_sf_ge((int64_t)x[t], M, (int64_t)L_fixed, (int64_t)L_fixed);
        sig[t] = (p < psig) ? 1 : 0;
    }
}

void launch_hyper_flags(const int32_t* x, int nrows, int mode, int L_fixed, int64_t M, double psig, uint8_t* sig,
                        hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_hyper_flags, dim3((nrows + 255) / 256), dim3(256), 0, s, x, nrows, mode, L_fixed, M, psig, sig);
}

}  // namespace hicmi
