// k_part2_search.hip - Part 2 search steps with the candidates ENUMERATED ON THE DEVICE
// (orderGenome.py:332-372 checkAllScores, :432-473 bruteForceBestScore, :495-549 scanOrdering).
//
// The reference builds, for every candidate, a Python list of bin indices and a gathered copy of
// the matrix.  Here a chromosome's sub-matrix is selected once (hicmi_p2_select); its scaffolds are
// contiguous ranges of that selection (the "layout"); the current order/orientation is a short
// list of (scaffold, reversed) pairs (the "arrangement") expanded on the device into
// pos2sel[position] -> selection index.  Candidates are then described by a few integers:
//   insertion : (gap, reversed) of one new scaffold          -> 2(S+1) candidates
//   window    : (order index, orientation index) into the enumeration tables of k scaffolds
//               -> k!/2 * 2^k candidates
// and both are scored INCREMENTALLY (closed form, fp64; w(d) = H[n-1] - H[d-1]):
//   window:    score(c) * total = [pairs outside the window]                      same for every c
//                               + sum_t G[x_t][t]                                  window bin x_t at slot t
//                               + sum_{s<t} M[u_s][u_t] * w(t-s)                   pairs inside the window
//              G[x][t] = sum_{q outside} M[u_x][bin at q] * w(|p0+t-q|): an (m x m) table per window
//              (m = bins in the window): m*m*(n-m) multiply-adds instead of (#candidates * n^2/2).
//   insertion: score * total = BASE - STRADDLE(g) + CROSS(g, r)   (see below)
// Only differences between candidates of one step are used by the host, and the winners are
// re-scored literally (k_p2_diag_sums), so the rounding of these decompositions never reaches an
// output.  Every kernel first copies pos2sel into LDS: the matrix is then reached through ONE level
// of indirection, and consecutive positions inside a scaffold are consecutive addresses.
#include "hicmi_internal.h"

namespace hicmi {

__device__ __forceinline__ double wave_sum_s(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum_256(double v, double* s_w)
{
    v = wave_sum_s(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__device__ __forceinline__ double block_sum_1024(double v, double* s_w)      // s_w: 16 doubles
{
    v = wave_sum_s(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += s_w[i];
    return acc;
}

static size_t perm_lds_bytes(int n) { return (((size_t)n * sizeof(int32_t)) + 15) & ~(size_t)15; }

// pos2sel[q] for the arrangement (arr_id, arr_rev) with prefix positions arr_pos[0..S]; the three
// arrays arrive packed in one buffer: [S ids][S+1 positions][S reversed flags as int32]
__global__ __launch_bounds__(256) void k_arr_materialize(const int32_t* __restrict__ packed, int S,
                                                         const int32_t* __restrict__ scaf_start,
                                                         const int32_t* __restrict__ scaf_len, int n_arr,
                                                         int32_t* __restrict__ pos2sel)
{
    const int32_t* __restrict__ arr_id = packed;
    const int32_t* __restrict__ arr_pos = packed + S;
    const int32_t* __restrict__ arr_rev = packed + 2 * S + 1;
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_arr) return;
    int lo = 0, hi = S;                                   // largest j with arr_pos[j] <= q
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (arr_pos[mid] <= q) lo = mid; else hi = mid;
    }
    int sc = arr_id[lo], off = q - arr_pos[lo], len = scaf_len[sc];
    pos2sel[q] = scaf_start[sc] + (arr_rev[lo] ? len - 1 - off : off);
}

void launch_arr_materialize(const int32_t* packed, int S, const int32_t* scaf_start, const int32_t* scaf_len, int n_arr,
                            int32_t* pos2sel, hipStream_t s)
{
    if (n_arr <= 0) return;
    hipLaunchKernelGGL(k_arr_materialize, dim3((n_arr + 255) / 256), dim3(256), 0, s, packed, S, scaf_start, scaf_len,
                       n_arr, pos2sel);
}

// ---- closed-form score of the arrangement itself --------------------------------------------------
// slab `blk` of `n_blk`: rows blk*4 + wave, stepping by 4*n_blk (256-lane workgroup); p: the arrangement in LDS
__device__ __forceinline__ void base_partial_body(const double* __restrict__ M2, int64_t ld2, const int32_t* p, int n_arr,
                                                  const double* __restrict__ H, int n_tot, int blk, int n_blk,
                                                  double* __restrict__ out)
{
    __shared__ double s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double hn = H[n_tot - 1];
    double acc = 0.0;
    for (int a = blk * 4 + wave; a < n_arr - 1; a += n_blk * 4) {
        const double* __restrict__ row = M2 + (int64_t)p[a] * ld2;
#pragma unroll 4
        for (int b = a + 1 + lane; b < n_arr; b += 64) acc += row[p[b]] * (hn - H[b - a - 1]);
    }
    double sum = block_sum_256(acc, s_w);
    if (threadIdx.x == 0) out[0] = sum;
}

__global__ __launch_bounds__(256) void k_p2_base_partial(const double* __restrict__ M2, int64_t ld2,
                                                         const int32_t* __restrict__ pos2sel, int n_arr,
                                                         const double* __restrict__ H, int n_tot,
                                                         double* __restrict__ partial)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    for (int q = threadIdx.x; q < n_arr; q += 256) p[q] = pos2sel[q];
    __syncthreads();
    base_partial_body(M2, ld2, p, n_arr, H, n_tot, blockIdx.x, gridDim.x, partial + blockIdx.x);
}

static std::atomic<int> g_lds_base{0}, g_lds_straddle{0}, g_lds_cross{0}, g_lds_wdelta{0}, g_lds_wdelta1{0}, g_lds_wdelta_blk{0}, g_lds_wG{0}, g_lds_insb_base{0}, g_lds_insb_fast{0};

// BASE as partial sums over row slabs: out[0..n_blocks)
void launch_p2_base_partial(const double* M2, int64_t ld2, const int32_t* pos2sel, int n_arr, const double* H, int n_tot,
                            int n_blocks, double* out, hipStream_t s)
{
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_base_partial), g_lds_base, perm_lds_bytes(n_arr));
    hipLaunchKernelGGL(k_p2_base_partial, dim3(n_blocks), dim3(256), perm_lds_bytes(n_arr), s, M2, ld2, pos2sel, n_arr, H,
                       n_tot, out);
}

// ---- insertion, incremental form ------------------------------------------------------------------
// Inserting a scaffold of L bins at gap g (position P = arr_pos[g]) shifts every bin at or after P by
// L.  With n' = n_arr + L and w(d) = H[n'-1] - H[d-1]:
//   score * total = BASE - STRADDLE(g) + CROSS(g, r)
//   BASE        = sum_{a<b} M[a][b] w(b-a)                      arrangement pairs at their old distance
//   STRADDLE(g) = sum_{a<P<=b} M[a][b] (w(b-a) - w(b-a+L))      pairs pushed apart by the insertion
//   CROSS(g, r) = new-scaffold x arrangement pairs + pairs inside the new scaffold
// STRADDLE(g+1) - STRADDLE(g) only involves the scaffold between the two gaps, so all gaps together
// cost one pass over the sub-matrix instead of one pass per candidate.
// Both bodies: 1024-lane workgroup, p = the arrangement in LDS, s_w = 16 doubles of LDS.
__device__ __forceinline__ void straddle_body(const double* __restrict__ M2, int64_t ld2, const int32_t* p, int n_arr,
                                              const int32_t* __restrict__ arr_pos, int g, int L,
                                              const double* __restrict__ H, double* s_w, double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P0 = arr_pos[g], P1 = arr_pos[g + 1];
    const int len = P1 - P0, n_out = n_arr - len;
    double acc = 0.0;
    for (int u = P0 + wave; u < P1; u += 16) {             // bins of the scaffold between gap g and g+1
        const double* __restrict__ row = M2 + (int64_t)p[u] * ld2;
        // positions before the scaffold: pairs (a, u) stop straddling; after it: pairs (u, b) start to
#pragma unroll 8
        for (int qq = lane; qq < n_out; qq += 64) {
            const bool before = qq < P0;
            const int pos = before ? qq : qq + len;
            const int d = before ? u - pos : pos - u;
            const double w = H[d + L - 1] - H[d - 1];
            const double v = row[p[pos]] * w;
            acc += before ? -v : v;
        }
    }
    double sum = block_sum_1024(acc, s_w);
    if (threadIdx.x == 0) out[0] = sum;
}

__device__ __forceinline__ void cross_body(const double* __restrict__ M2, int64_t ld2, const int32_t* p, int n_arr,
                                           const int32_t* __restrict__ arr_pos, int g, int r, int new_start, int L,
                                           const double* __restrict__ H, double* s_w, double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = arr_pos[g];
    const double hn = H[n_arr + L - 1];
    double acc = 0.0;
    for (int e = wave; e < L; e += 16) {
        const int xe = new_start + (r ? L - 1 - e : e);
        const double* __restrict__ row = M2 + (int64_t)xe * ld2;
#pragma unroll 4
        for (int q = lane; q < n_arr; q += 64) {
            int d = q < P ? (P + e - q) : (q + L - (P + e));
            acc += row[p[q]] * (hn - H[d - 1]);
        }
        for (int e2 = e + 1 + lane; e2 < L; e2 += 64) {
            int x2 = new_start + (r ? L - 1 - e2 : e2);
            acc += row[x2] * (hn - H[e2 - e - 1]);
        }
    }
    double sum = block_sum_1024(acc, s_w);
    if (threadIdx.x == 0) out[0] = sum;
}

__global__ __launch_bounds__(1024) void k_p2_insert_straddle(const double* __restrict__ M2, int64_t ld2,
                                                            const int32_t* __restrict__ pos2sel, int n_arr,
                                                            const int32_t* __restrict__ arr_pos, int L,
                                                            const double* __restrict__ H, double* __restrict__ D)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    __shared__ double s_w[16];
    for (int q = threadIdx.x; q < n_arr; q += 1024) p[q] = pos2sel[q];
    __syncthreads();
    straddle_body(M2, ld2, p, n_arr, arr_pos, blockIdx.x, L, H, s_w, D + blockIdx.x);
}

__global__ __launch_bounds__(1024) void k_p2_insert_cross(const double* __restrict__ M2, int64_t ld2,
                                                         const int32_t* __restrict__ pos2sel, int n_arr,
                                                         const int32_t* __restrict__ arr_pos, int new_start, int L,
                                                         const double* __restrict__ H, double* __restrict__ cross)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    __shared__ double s_w[16];
    for (int q = threadIdx.x; q < n_arr; q += 1024) p[q] = pos2sel[q];
    __syncthreads();
    cross_body(M2, ld2, p, n_arr, arr_pos, blockIdx.x >> 1, blockIdx.x & 1, new_start, L, H, s_w, cross + blockIdx.x);
}

void launch_p2_insert_delta(const double* M2, int64_t ld2, const int32_t* pos2sel, int n_arr, const int32_t* arr_pos,
                            int S, int new_start, int L, const double* H, int n_base_blocks, double* out, hipStream_t s)
{
    // out: [n_base_blocks partial sums of BASE][S straddle increments][2(S+1) cross terms]
    const size_t lds = perm_lds_bytes(n_arr);
    launch_p2_base_partial(M2, ld2, pos2sel, n_arr, H, n_arr + L, n_base_blocks, out, s);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_insert_straddle), g_lds_straddle, lds);
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_insert_cross), g_lds_cross, lds);
    hipLaunchKernelGGL(k_p2_insert_straddle, dim3(S), dim3(1024), lds, s, M2, ld2, pos2sel, n_arr, arr_pos, L, H,
                       out + n_base_blocks);
    hipLaunchKernelGGL(k_p2_insert_cross, dim3(2 * (S + 1)), dim3(1024), lds, s, M2, ld2, pos2sel, n_arr, arr_pos, new_start,
                       L, H, out + n_base_blocks + S);
}

// ---- lock-step insertion (k_part2_insert.hip): one layer of workgroups per chromosome --------------
__global__ __launch_bounds__(256) void k_insb_base(const InsStep* __restrict__ steps, int n_base_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* p = reinterpret_cast<int32_t*>(smem);
    const InsStep& d = steps[blockIdx.y];
    if (!d.active || d.st->fail >= 0) return;
    for (int q = threadIdx.x; q < d.n_arr; q += 256) p[q] = d.pos_cur[q];
    __syncthreads();
    base_partial_body(d.M2, d.ld2, p, d.n_arr, d.H, d.n_arr + d.L, blockIdx.x, n_base_blocks, d.partial + blockIdx.x);
}

// Lock-step form of STRADDLE and CROSS, one WAVE per matrix row so that short scaffolds do not leave most of a
// large workgroup idle (a late insertion has L = 1..3 bins and the 16-wave bodies above kept 1..3 waves busy):
//   workgroups [0, ceil(n_arr / 4)): wave = arrangement position u.  With its scaffold spanning [P0, P1),
//       s(u) = - sum_{pos < P0} M[u][pos] dw(u - pos) + sum_{pos >= P1} M[u][pos] dw(pos - u),
//       dw(d) = H[d + L - 1] - H[d - 1];   STRADDLE(g+1) - STRADDLE(g) = sum of s(u) over scaffold g
//       (k_insb_shortlist adds the rows of a scaffold in position order)
//   workgroups after those: one per (gap, orientation), CROSS as in cross_body with 4 waves over the new bins.
// partial layout per chromosome: [n_base_blocks BASE slabs][n_arr row values s(u)][2(S+1) CROSS terms]
//   in front of all those (the longest workgroups first): the BASE slabs of k_insb_base - independent of the rest
//       (same inputs, a disjoint part of `partial`), so they ride in the same launch instead of in a launch of their
//       own: one dependent launch fewer in each of the ~136 lock steps of a 16k map
__global__ __launch_bounds__(256) void k_insb_fast(const InsStep* __restrict__ steps, int n_base_blocks, int n_row_blocks,
                                                   int n_cross_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double s_w[4];
    const InsStep& d = steps[blockIdx.y];
    if (!d.active || d.st->fail >= 0) return;
    const int S = d.S, n_arr = d.n_arr, L = d.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t* __restrict__ p = d.pos_cur;
    const int32_t* __restrict__ arr_pos = d.packed_cur + S;
    const double* __restrict__ H = d.H;
    const int n_base_here = (int)gridDim.x - n_row_blocks - n_cross_blocks;      // 0 when BASE has its own launch (A/B)
    const int bx = (int)blockIdx.x - n_base_here;
    if (bx < 0) {
        const int slab = blockIdx.x;
        int32_t* pl = reinterpret_cast<int32_t*>(smem);
        for (int q = threadIdx.x; q < n_arr; q += 256) pl[q] = p[q];
        __syncthreads();
        base_partial_body(d.M2, d.ld2, pl, n_arr, H, n_arr + L, slab, n_base_blocks, d.partial + slab);
        return;
    }
    if (bx < n_row_blocks) {
        const int u = bx * 4 + wave;
        if (u >= n_arr) return;
        int below = 0;                                    // scaffold of position u: the last gap with arr_pos <= u
        for (int j = lane; j <= S; j += 64) below += arr_pos[j] <= u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) below += __shfl_xor(below, off, 64);
        const int g = below - 1;
        const int P0 = arr_pos[g], P1 = arr_pos[g + 1];
        const double* __restrict__ row = d.M2 + (int64_t)p[u] * d.ld2;
        double acc = 0.0;
#pragma unroll 4
        for (int pos = lane; pos < P0; pos += 64) {
            const int dd = u - pos;
            acc -= row[p[pos]] * (H[dd + L - 1] - H[dd - 1]);
        }
#pragma unroll 4
        for (int pos = P1 + lane; pos < n_arr; pos += 64) {
            const int dd = pos - u;
            acc += row[p[pos]] * (H[dd + L - 1] - H[dd - 1]);
        }
        acc = wave_sum_s(acc);
        if (lane == 0) d.partial[n_base_blocks + u] = acc;
        return;
    }
    const int c = bx - n_row_blocks;
    if (c >= 2 * (S + 1)) return;
    const int g = c >> 1, r = c & 1, P = arr_pos[g];
    const double hn = H[n_arr + L - 1];
    double acc = 0.0;
    for (int e = wave; e < L; e += 4) {
        const int xe = d.new_start + (r ? L - 1 - e : e);
        const double* __restrict__ row = d.M2 + (int64_t)xe * d.ld2;
#pragma unroll 4
        for (int q = lane; q < n_arr; q += 64) {
            const int dd = q < P ? (P + e - q) : (q + L - (P + e));
            acc += row[p[q]] * (hn - H[dd - 1]);
        }
        for (int e2 = e + 1 + lane; e2 < L; e2 += 64) {
            const int x2 = d.new_start + (r ? L - 1 - e2 : e2);
            acc += row[x2] * (hn - H[e2 - e - 1]);
        }
    }
    const double sum = block_sum_256(acc, s_w);
    if (threadIdx.x == 0) d.partial[n_base_blocks + n_arr + c] = sum;
}

void launch_insb_fast(const InsStep* steps, int n_chrom, int max_S, int max_n_arr, int n_base_blocks, hipStream_t s)
{
    const size_t lds = perm_lds_bytes(max_n_arr);
    const int n_row_blocks = (max_n_arr + 3) / 4, n_cross_blocks = 2 * (max_S + 1);
    static const bool split = getenv("HICMI_P2_INSB_SPLIT") != nullptr;       // A/B: BASE as its own launch, as before
    if (split) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_insb_base), g_lds_insb_base, lds);
        hipLaunchKernelGGL(k_insb_base, dim3(n_base_blocks, n_chrom), dim3(256), lds, s, steps, n_base_blocks);
    }
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_insb_fast), g_lds_insb_fast, lds);
    hipLaunchKernelGGL(k_insb_fast, dim3(n_row_blocks + n_cross_blocks + (split ? 0 : n_base_blocks), n_chrom), dim3(256), lds, s,
                       steps, n_base_blocks, n_row_blocks, n_cross_blocks);
}

// ---- window: G table ----------------------------------------------------------------------------
// Windows are evaluated in BATCHES (blockIdx.y = window): within one round of scanOrdering most windows
// bring no improvement, and until one does, all of them see the same arrangement.
// block x = window bin at current position p0+x.  Its row of M2, restricted to the positions outside
// [p0, p0+m), is staged through LDS in tiles; the 256 lanes form a 64 (slot t) x 4 (quarter of the
// tile) grid, each lane keeping one accumulator per 64 slots, and the four quarters are added at
// the end.
static constexpr int G_TILE = 2048;
static constexpr int G_TMAX = 8;                        // up to 512 window bins per pass

template <bool H_IN_LDS>
__global__ __launch_bounds__(256) void k_p2_window_G(const double* __restrict__ M2, int64_t ld2,
                                                     const int32_t* __restrict__ pos2sel, int n,
                                                     const WindowBatchEntry* __restrict__ wb,
                                                     const double* H, double* __restrict__ G_all)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
    __shared__ double vals[G_TILE];
    __shared__ double part[4][64];
    const WindowBatchEntry& we = wb[blockIdx.y];
    const int p0 = we.p0, m = we.m;
    const int x = blockIdx.x, tid = threadIdx.x, tl = tid & 63, seg = tid >> 6;
    if (x >= m) return;
    double* __restrict__ G = G_all + we.g_off;
    const double* __restrict__ row = M2 + (int64_t)pos2sel[p0 + x] * ld2;
    const double hn = H[n - 1];
    if (H_IN_LDS) {                                         // the harmonic table is read once per multiply-add
        double* hl = reinterpret_cast<double*>(smem_g);
        for (int i = tid; i < n; i += 256) hl[i] = H[i];
        __syncthreads();
        H = hl;
    }
    const int n_out = n - m;
    for (int tbase = 0; tbase < m; tbase += 64 * G_TMAX) {
        double acc[G_TMAX];
#pragma unroll
        for (int i = 0; i < G_TMAX; i++) acc[i] = 0.0;
        for (int base = 0; base < n_out; base += G_TILE) {
            const int cnt = n_out - base < G_TILE ? n_out - base : G_TILE;
            __syncthreads();
            for (int e = tid; e < cnt; e += 256) {
                int qq = base + e;
                vals[e] = row[pos2sel[qq < p0 ? qq : qq + m]];          // skip the window
            }
            __syncthreads();
            const int e0 = (cnt * seg) >> 2, e1 = (cnt * (seg + 1)) >> 2;
#pragma unroll
            for (int i = 0; i < G_TMAX; i++) {
                const int t = tbase + i * 64 + tl;
                if (t < m) {
                    const int pt = p0 + t;
                    double a = acc[i];
                    for (int e = e0; e < e1; e++) {
                        int qq = base + e;
                        int d = qq < p0 ? pt - qq : qq + m - pt;
                        a += vals[e] * (hn - H[d - 1]);
                    }
                    acc[i] = a;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < G_TMAX; i++) {
            __syncthreads();
            part[seg][tl] = acc[i];
            __syncthreads();
            const int t = tbase + i * 64 + tl;
            if (seg == 0 && t < m) G[(int64_t)x * m + t] = (part[0][tl] + part[1][tl]) + (part[2][tl] + part[3][tl]);
        }
    }
}

// ---- window: per-candidate delta -----------------------------------------------------------------
// block = (candidate, window).  Slot j of the candidate holds window scaffold jj = orders[o][j] laid
// down reversed iff orients[r][j]; a bin's window-local index x is its offset inside the CURRENT
// window layout (that is how G is indexed).
static constexpr int WD_PER_WAVE = 4;                    // candidates a wave scores one after the other
// WD_WAVES = 4: 16 candidates per 256-lane workgroup (single-wave workgroups made the launch rate the
// bottleneck: 61,000 of them per batch); WD_WAVES = 1 keeps windows of more than 4096 bins within the LDS.
// BLOCK_IN_LDS (windows of at most 128 bins): the window's m x m block of the matrix, in the window's current
// layout, is staged in LDS once per workgroup - all 1920 candidates of a window read the same 34 KB block in a
// different order, which otherwise comes out of L2 once per candidate (2.3 GB per batch of 32 windows).
template <int WD_WAVES, bool BLOCK_IN_LDS>
__global__ __launch_bounds__(WD_WAVES * 64) void k_p2_window_delta(const double* __restrict__ M2, int64_t ld2, int n, int k,
                                                         const WindowBatchEntry* __restrict__ wb,
                                                         const int8_t* __restrict__ orders,
                                                         const uint8_t* __restrict__ orients, int n_ori, int n_cand,
                                                         const double* __restrict__ H, const double* __restrict__ G_all,
                                                         double* __restrict__ delta_all,
                                                         const int32_t* __restrict__ pos2sel, int max_m)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the window's scaffolds (once per workgroup) and each wave's candidate order / orientation row live in LDS:
    // private arrays indexed at run time would be placed in scratch memory
    __shared__ int s_start[8], s_len[8], s_off[8], s_rev[8];
    __shared__ int s_ord[WD_WAVES][8], s_ori[WD_WAVES][8], s_slot[WD_WAVES][9];
    const WindowBatchEntry& we = wb[blockIdx.y];
    const int m = we.m;
    const double* __restrict__ G = G_all + we.g_off;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* useq = reinterpret_cast<int32_t*>(smem) + (int64_t)wave * 2 * m;     // selection index at slot t
    int32_t* xseq = useq + m;                                                    // window-local index at slot t
    double* wt = reinterpret_cast<double*>(smem + (((size_t)max_m * 2 * WD_WAVES * sizeof(int32_t) + 15) & ~(size_t)15));
    double* wblk = wt + max_m;                               // wt[d] = H[n-1] - H[d-1], the weight of a pair d slots apart
    if (tid < k) {
        s_start[tid] = we.w.start[tid]; s_len[tid] = we.w.len[tid]; s_off[tid] = we.w.off[tid]; s_rev[tid] = we.w.rev[tid];
    }
    if (BLOCK_IN_LDS) {
        const int32_t* __restrict__ wsel = pos2sel + we.p0;                      // selection index of window bin x
        for (int e = tid; e < m * m; e += WD_WAVES * 64) {
            const int x = e / m, y = e - x * m;
            wblk[e] = M2[(int64_t)wsel[x] * ld2 + wsel[y]];
        }
    }
    const double hn = H[n - 1];
    for (int d = tid; d < m; d += WD_WAVES * 64) wt[d] = d ? hn - H[d - 1] : 0.0;    // wt[0] = 0 mutes the lower half
    const int mm = m * m, ds = 64 / m, dt = 64 - ds * m;
    for (int it = 0; it < WD_PER_WAVE; it++) {
        const int c = (blockIdx.x * WD_WAVES + wave) * WD_PER_WAVE + it;
        const bool live = c < n_cand;
        if (live && lane < k) {
            s_ord[wave][lane] = orders[(int64_t)(c / n_ori) * k + lane];
            s_ori[wave][lane] = orients[(int64_t)(c % n_ori) * k + lane];
        }
        __syncthreads();
        if (live && lane == 0) {
            int acc_len = 0;
            s_slot[wave][0] = 0;
            for (int j = 0; j < k; j++) { acc_len += s_len[s_ord[wave][j]]; s_slot[wave][j + 1] = acc_len; }
        }
        __syncthreads();
        if (live) {
            for (int t = lane; t < m; t += 64) {
                int j = 0;
                while (j + 1 < k && s_slot[wave][j + 1] <= t) j++;
                const int jj = s_ord[wave][j], len = s_len[jj];
                const int ep = t - s_slot[wave][j];
                const int e = s_ori[wave][j] ? len - 1 - ep : ep;      // offset inside the scaffold, selection order
                useq[t] = s_start[jj] + e;
                xseq[t] = s_off[jj] + (s_rev[jj] ? len - 1 - e : e);
            }
        }
        __syncthreads();
        if (live) {
            double acc = 0.0;
            for (int t = lane; t < m; t += 64) acc += G[(int64_t)xseq[t] * m + t];
            // pairs inside the window: the m x m grid is walked flat (s = q / m, t = q % m kept up to date by
            // carries).  No branch in the body - the lower half is multiplied by wt[0] = 0 - so that the eight
            // unrolled trips issue their index, matrix and weight loads together instead of one dependent chain
            // per trip.
            int ps = lane / m, pt = lane - ps * m;
#pragma unroll 8
            for (int q = lane; q < mm; q += 64) {
                const double v = BLOCK_IN_LDS ? wblk[xseq[ps] * m + xseq[pt]] : M2[(int64_t)useq[ps] * ld2 + useq[pt]];
                acc += v * wt[pt > ps ? pt - ps : 0];
                const int npt = pt + dt;
                const bool carry = npt >= m;
                ps += ds + (carry ? 1 : 0);
                pt = carry ? npt - m : npt;
            }
            acc = wave_sum_s(acc);
            if (lane == 0) delta_all[(int64_t)blockIdx.y * n_cand + c] = acc;
        }
        __syncthreads();
    }
}

// one launch pair for n_win windows described by the device array wb; max_m = largest window (bins)
void launch_p2_window_batch(const double* M2, int64_t ld2, const int32_t* pos2sel, int n, int k,
                            const WindowBatchEntry* wb, int n_win, int max_m, const int8_t* orders, const uint8_t* orients,
                            int n_ord, int n_ori, const double* H, double* G_all, double* delta_all, hipStream_t s)
{
    if (n_win <= 0 || max_m <= 0) return;
    const size_t h_lds = (((size_t)n * sizeof(double)) + 15) & ~(size_t)15;
    if (h_lds <= 96 * 1024) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_window_G<true>), g_lds_wG, h_lds);
        hipLaunchKernelGGL(k_p2_window_G<true>, dim3(max_m, n_win), dim3(256), h_lds, s, M2, ld2, pos2sel, n, wb, H, G_all);
    } else {
        hipLaunchKernelGGL(k_p2_window_G<false>, dim3(max_m, n_win), dim3(256), 0, s, M2, ld2, pos2sel, n, wb, H, G_all);
    }
    const int n_cand = n_ord * n_ori;
    const size_t wt_bytes = (size_t)max_m * sizeof(double);
    const size_t seq4 = ((((size_t)max_m * 2 * 4 * sizeof(int32_t)) + 15) & ~(size_t)15) + wt_bytes;
    if (max_m <= 128) {
        const size_t lds = seq4 + (size_t)max_m * max_m * sizeof(double);
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_window_delta<4, true>), g_lds_wdelta_blk, lds);
        hipLaunchKernelGGL((k_p2_window_delta<4, true>), dim3((n_cand + 4 * WD_PER_WAVE - 1) / (4 * WD_PER_WAVE), n_win), dim3(256),
                           lds, s, M2, ld2, n, k, wb, orders, orients, n_ori, n_cand, H, G_all, delta_all, pos2sel, max_m);
    } else if (max_m <= 4096) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_window_delta<4, false>), g_lds_wdelta, seq4);
        hipLaunchKernelGGL((k_p2_window_delta<4, false>), dim3((n_cand + 4 * WD_PER_WAVE - 1) / (4 * WD_PER_WAVE), n_win), dim3(256),
                           seq4, s, M2, ld2, n, k, wb, orders, orients, n_ori, n_cand, H, G_all, delta_all, pos2sel, max_m);
    } else {
        const size_t lds = ((((size_t)max_m * 2 * sizeof(int32_t)) + 15) & ~(size_t)15) + wt_bytes;
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_p2_window_delta<1, false>), g_lds_wdelta1, lds);
        hipLaunchKernelGGL((k_p2_window_delta<1, false>), dim3((n_cand + WD_PER_WAVE - 1) / WD_PER_WAVE, n_win), dim3(64), lds, s, M2,
                           ld2, n, k, wb, orders, orients, n_ori, n_cand, H, G_all, delta_all, pos2sel, max_m);
    }
}

}  // namespace hicmi
