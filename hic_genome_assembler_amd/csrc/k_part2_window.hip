// k_part2_window.hip - window scoring by PLACEMENT TABLES (bruteForceBestScore orderGenome.py:432-473,
// scanOrdering :495-549): all k!/2 * 2^k orders x orientations of a window of k scaffolds.
//
// With w(d) = H[n-1] - H[d-1], the part of  score * total  that differs between candidates is
//     sum over window scaffolds j            of  Q[j][r_j][B_j]        pairs (bin of j, bin outside the window)
//   + sum over window scaffold pairs a -> b  of  P[a][b][r_a][r_b][G_ab] pairs (bin of a, bin of b), a placed first
//   + a constant (pairs inside one scaffold: a reversal keeps their distances)
// where r = orientation, B_j = the SET of window scaffolds placed before j (it fixes j's offset) and
// G_ab = the set placed between a and b (it fixes their gap).  A set is a k-bit mask, so the tables have
//   Q: k * 2 * 2^(k-1)   and   P: k(k-1) * 4 * 2^(k-2)   entries
// (k = 5: 160 + 640; k = 6: 384 + 1920) against 1,920 / 23,040 candidates that each used to walk all m^2/2
// pairs of the window: the matrix is read once per table, and a candidate is k + k(k-1)/2 look-ups.
// For the brute-force step over the six largest scaffolds this is ~180x fewer multiply-adds, and it no
// longer grows with (number of candidates) x (window bins)^2.
//
// Only differences between candidates of a step reach a decision, and the winners are re-scored literally
// (k_p2_diag_sums), so the summation order of these tables never reaches an output.
#include <cstdlib>

#include "hicmi_internal.h"

namespace hicmi {

__device__ __forceinline__ double wave_sum_w(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// per-window table layout (doubles): [Q: WT_SLICES x (k*2 << k)][P: k*k*4 << k][C: 8].  The Q entries of window
// scaffold j are accumulated as nrg x nq partial tables by separate workgroups - groups of 16-row tiles of the scaffold
// times slices of the outside positions, as many as bring a workgroup near WT_UNIT matrix instructions per wave
// (wt_split: a 300-bin scaffold in a 400-bin window is ~40x the work of a 12-bin one in a 90-bin window, and one
// workgroup per (window, scaffold, slice) left the launch waiting for the few heavy ones) - and k_win_pairs' diagonal
// workgroups add the partial tables, in order, into slice 0 before the candidates look entries up.
static constexpr int WT_SLICES = 64;
static constexpr int WT_UNIT = 160;
__host__ __device__ inline int64_t wt_q_size(int k) { return ((int64_t)k * 2) << k; }
__host__ __device__ inline int64_t wt_p_size(int k) { return ((int64_t)k * k * 4) << k; }
int64_t window_table_doubles(int k) { return WT_SLICES * wt_q_size(k) + wt_p_size(k) + 8; }

// partial tables of a scaffold of Lj bins in a window of m bins with n_out positions outside: nrg groups of row tiles
// (group g takes the tiles g, g + nrg, ...) x nq column slices; the same on the host (grid size) and in every kernel
__host__ __device__ inline void wt_split(int Lj, int m, int n_out, int& nrg, int& nq)
{
    const int nr = Lj > 0 ? (Lj + 15) >> 4 : 1, tiles_w = ((((m + 15) >> 4) + 3) >> 2);
    const int64_t work = (int64_t)nr * tiles_w * ((n_out + 3) >> 2);     // matrix instructions per wave
    int64_t units = (work + WT_UNIT - 1) / WT_UNIT;
    units = units < 1 ? 1 : (units > WT_SLICES ? WT_SLICES : units);
    nrg = nr < (int)units ? nr : (int)units;
    nq = (int)units / nrg;
    const int by_cols = n_out >= 64 ? n_out / 32 : 1;
    nq = nq > by_cols ? by_cols : nq;
    nq = nq < 1 ? 1 : nq;
}

static constexpr int WT_ACC = 8;                        // table entries a wave accumulates per pass over the data

// insert a zero bit at position b (b below the current width)
__device__ __forceinline__ int spread_bit(int v, int b) { return ((v >> b) << (b + 1)) | (v & ((1 << b) - 1)); }

// ---- Q: window scaffold j against everything outside the window -------------------------------------------------
// workgroup = (scaffold j, window); wave w takes entries w*8 .. w*8+7, then +32, ...; lanes sweep the outside
// positions; every matrix element read feeds 8 table entries.
template <bool H_IN_LDS>
__global__ __launch_bounds__(256) void k_win_outside(const double* __restrict__ M2, int64_t ld2,
                                                     const int32_t* __restrict__ pos2sel, int n, int k,
                                                     const WindowBatchEntry* __restrict__ wb, const double* H,
                                                     double* __restrict__ tables, int64_t table_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    __shared__ int s_len[8], s_glen[256];
    const WindowBatchEntry& we = wb[blockIdx.y];
    const int j = blockIdx.x, p0 = we.p0, m = we.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nrg, nq;
    wt_split(we.w.len[j], m, n - m, nrg, nq);
    if ((int)blockIdx.z >= nrg * nq) return;                // (the whole workgroup, before any barrier)
    if (tid < k) s_len[tid] = we.w.len[tid];
    const double hn = H[n - 1];
    if (H_IN_LDS) {
        double* hl = reinterpret_cast<double*>(smem_w);
        for (int i = tid; i < n; i += 256) hl[i] = H[i];
        H = hl;
    }
    __syncthreads();
    if (tid < (1 << k)) {
        int g = 0;
        for (int b = 0; b < k; b++) if ((tid >> b) & 1) g += s_len[b];
        s_glen[tid] = g;
    }
    __syncthreads();
    const int Lj = s_len[j], startj = we.w.start[j], n_out = n - m;
    const int rt = blockIdx.z % nrg, qs = blockIdx.z / nrg;
    const int q_lo = (int)(((int64_t)qs * n_out) / nq), q_hi = (int)(((int64_t)(qs + 1) * n_out) / nq);
    double* __restrict__ Q = tables + (int64_t)blockIdx.y * table_stride + blockIdx.z * wt_q_size(k) + (((int64_t)j * 2) << k);
    const int n_entries = 2 << (k - 1);                     // (orientation, set of the other k-1 scaffolds)
    for (int e0 = wave * WT_ACC; e0 < n_entries; e0 += 4 * WT_ACC) {
        double acc[WT_ACC];
        int off[WT_ACC], rev[WT_ACC];
#pragma unroll
        for (int u = 0; u < WT_ACC; u++) {
            const int e = e0 + u < n_entries ? e0 + u : n_entries - 1;
            rev[u] = e >> (k - 1);
            off[u] = p0 + s_glen[spread_bit(e & ((1 << (k - 1)) - 1), j)];
            acc[u] = 0.0;
        }
        for (int i = rt * 16; i < Lj; i = (i & 15) == 15 ? i + 1 + (nrg - 1) * 16 : i + 1) {   // the row tiles rt, rt + nrg, ...
            const double* __restrict__ row = M2 + (int64_t)(startj + i) * ld2;
#pragma unroll 2
            for (int qq = q_lo + lane; qq < q_hi; qq += 64) {
                const int pos = qq < p0 ? qq : qq + m;      // positions outside the window keep their place
                const double v = row[pos2sel[pos]];
#pragma unroll
                for (int u = 0; u < WT_ACC; u++) {
                    const int slot = off[u] + (rev[u] ? Lj - 1 - i : i);
                    const int d = slot > pos ? slot - pos : pos - slot;
                    acc[u] += v * (hn - H[d - 1]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WT_ACC; u++) {
            const double sum = wave_sum_w(acc[u]);
            const int e = e0 + u;
            if (lane == 0 && e < n_entries)
                Q[((int64_t)(e >> (k - 1)) << k) | spread_bit(e & ((1 << (k - 1)) - 1), j)] = sum;
        }
    }
}

// ---- Q on the matrix cores --------------------------------------------------------------------------------------
// The same table as k_win_outside, formed through the dense product behind it (the results-neutral
// contraction of this path; the north_star's "row x row^T similarity" does not exist in the reference):
//     G[i][s] = sum over outside positions q of  M[row i of scaffold j][column at q] * w(|p0 + s - pos(q)|),   s = 0 .. m-1
// i.e. G = A (L_j x n_out) . T (n_out x m) with the Toeplitz decay T[q][s] = H[n-1] - H[|p0 + s - pos(q)| - 1] generated on
// the fly from H, on v_mfma_f64_16x16x4_f64 (A: one f64 per lane, row = lane & 15, k = lane >> 4; B: k = lane >> 4,
// column = lane & 15; C/D: four f64 per lane, row = (lane >> 4) + 4 r, column = lane & 15).  A table entry is then a
// diagonal of G:  Q[j][rev][B] = sum_i G[i][glen(B) + (rev ? L_j - 1 - i : i)].
// Workgroup = (scaffold j, window, slice of the outside positions); 16-row tiles of A are staged through LDS 64 columns at
// a time (coalesced along q), wave w accumulates the slot tiles w, w + 4, ...; fp64 throughout.  Every entry is a
// different summation order than the vector-ALU kernel's: neither reaches an output (winners are re-scored literally).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
static constexpr int WM_QC = 128;                         // outside positions staged per trip (256 threads: 16 * WM_QC / 256 rows each)
static constexpr int WM_MAX_TILES = 8;                    // slot tiles per wave: windows up to 4 * 8 * 16 = 512 bins

// TILES: slot tiles per wave (2, 4 or 8 by the widest window of the batch - the accumulators are most of the kernel's
// registers, and with 2 a CU holds five workgroups instead of three)
template <bool H_IN_LDS, int TILES>
__global__ __launch_bounds__(256) void k_win_outside_mfma(const double* __restrict__ M2, int64_t ld2,
                                                          const int32_t* __restrict__ pos2sel, int n, int k,
                                                          const WindowBatchEntry* __restrict__ wb, const double* H,
                                                          double* __restrict__ tables, int64_t table_stride, int m_pad_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    double* As = reinterpret_cast<double*>(smem_w);                          // [16][WM_QC + 1]
    double* Gt = As;                                                          // [16][m_pad_max], after the last fragment is read
    const int as_doubles = 16 * (WM_QC + 1) > 16 * m_pad_max ? 16 * (WM_QC + 1) : 16 * m_pad_max;
    __shared__ int s_len[8], s_glen[256], s_pos[WM_QC], s_col[WM_QC];
    const WindowBatchEntry& we = wb[blockIdx.y];
    const int j = blockIdx.x, p0 = we.p0, m = we.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nrg, nq;
    wt_split(we.w.len[j], m, n - m, nrg, nq);
    if ((int)blockIdx.z >= nrg * nq) return;                // (the whole workgroup, before any barrier)
    if (tid < k) s_len[tid] = we.w.len[tid];
    const double hn = H[n - 1];
    if (H_IN_LDS) {
        double* hl = As + as_doubles;
        for (int i = tid; i < n; i += 256) hl[i] = H[i];
        H = hl;
    }
    __syncthreads();
    if (tid < (1 << k)) {
        int g = 0;
        for (int b = 0; b < k; b++) if ((tid >> b) & 1) g += s_len[b];
        s_glen[tid] = g;
    }
    __syncthreads();
    const int Lj = s_len[j], startj = we.w.start[j], n_out = n - m;
    const int rt = blockIdx.z % nrg, qs = blockIdx.z / nrg;
    const int q_lo = (int)(((int64_t)qs * n_out) / nq), q_hi = (int)(((int64_t)(qs + 1) * n_out) / nq);
    double* __restrict__ Q = tables + (int64_t)blockIdx.y * table_stride + blockIdx.z * wt_q_size(k) + (((int64_t)j * 2) << k);
    const int n_entries = 2 << (k - 1);                     // (orientation, set of the other k-1 scaffolds): <= 256
    const int m_pad = (m + 15) & ~15, n_tiles = m_pad >> 4;
    // this thread's table entry (threads >= n_entries idle in the reduction)
    const int e_rev = tid < n_entries ? tid >> (k - 1) : 0;
    const int e_off = tid < n_entries ? s_glen[spread_bit(tid & ((1 << (k - 1)) - 1), j)] : 0;
    double qacc = 0.0;
    for (int r0 = rt * 16; r0 < Lj; r0 += nrg * 16) {
        mfma_d4 acc[TILES];
#pragma unroll
        for (int t = 0; t < TILES; t++) acc[t] = mfma_d4{0.0, 0.0, 0.0, 0.0};
        for (int qc = q_lo; qc < q_hi; qc += WM_QC) {
            const int cols = q_hi - qc < WM_QC ? q_hi - qc : WM_QC;
            if (q_hi - q_lo > WM_QC || r0 == rt * 16) {     // a slice of one trip keeps its positions over the row tiles
                __syncthreads();                            // the previous trip's fragments have been read
                if (tid < WM_QC) {
                    const int qq = qc + tid;
                    const int pos = tid < cols ? (qq < p0 ? qq : qq + m) : -1;     // positions outside the window keep their place
                    s_pos[tid] = pos;
                    s_col[tid] = pos >= 0 ? pos2sel[pos] : 0;
                }
            }
            __syncthreads();
            {                                               // 16 rows x WM_QC columns, coalesced along q, all loads in flight
                constexpr int RPT = 16 * WM_QC / 256;       // rows per thread
                const int cq = tid % WM_QC, rb = (tid / WM_QC) * RPT;
                const int c = s_col[cq];
                const bool okc = cq < cols;
                double v[RPT];
#pragma unroll
                for (int r = 0; r < RPT; r++)
                    v[r] = okc && r0 + rb + r < Lj ? M2[(int64_t)(startj + r0 + rb + r) * ld2 + c] : 0.0;
#pragma unroll
                for (int r = 0; r < RPT; r++) As[(rb + r) * (WM_QC + 1) + cq] = v[r];
            }
            __syncthreads();
            const int kk_end = (cols + 3) >> 2;
#pragma unroll 2
            for (int kk = 0; kk < kk_end; kk++) {
                const int kq = kk * 4 + (lane >> 4);
                const double a = As[(lane & 15) * (WM_QC + 1) + kq];
                int pos = s_pos[kq];
                pos = pos < 0 ? p0 : pos;                   // a padded column carries a == 0: any finite weight does
#pragma unroll
                for (int t = 0; t < TILES; t++) {
                    const int st = wave + 4 * t;
                    if (st < n_tiles) {                     // uniform per wave
                        int d = p0 + st * 16 + (lane & 15) - pos;
                        d = d < 0 ? -d : d;
                        d = d < 1 ? 1 : (d > n - 1 ? n - 1 : d);    // padded rows / columns carry a == 0
                        const double b = hn - H[d - 1];
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TILES; t++) {
            const int st = wave + 4 * t;
            if (st < n_tiles) {
#pragma unroll
                for (int r = 0; r < 4; r++) Gt[((lane >> 4) + 4 * r) * m_pad_max + st * 16 + (lane & 15)] = acc[t][r];
            }
        }
        __syncthreads();
        if (tid < n_entries) {
            const int rows = Lj - r0 < 16 ? Lj - r0 : 16;
            for (int r = 0; r < rows; r++) {
                const int i = r0 + r;
                qacc += Gt[r * m_pad_max + e_off + (e_rev ? Lj - 1 - i : i)];
            }
        }
        __syncthreads();                                    // Gt shares As
    }
    if (tid < n_entries) Q[((int64_t)e_rev << k) | spread_bit(tid & ((1 << (k - 1)) - 1), j)] = qacc;
}

// ---- P: window scaffold a placed before b; C: pairs inside one scaffold ------------------------------------------
// workgroup = (a * k + b, window).  Every element of the La x Lb block feeds 8 entries per pass.
template <bool H_IN_LDS>
__global__ __launch_bounds__(256) void k_win_pairs(const double* __restrict__ M2, int64_t ld2, int n, int k,
                                                   const WindowBatchEntry* __restrict__ wb, const double* H,
                                                   double* __restrict__ tables, int64_t table_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    __shared__ int s_len[8], s_glen[256];
    __shared__ double s_part[4];
    const WindowBatchEntry& we = wb[blockIdx.y];
    const int a = blockIdx.x / k, b = blockIdx.x - a * k, m = we.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < k) s_len[tid] = we.w.len[tid];
    const double hn = H[n - 1];
    if (H_IN_LDS) {                                         // only distances below m are needed here
        double* hl = reinterpret_cast<double*>(smem_w);
        for (int i = tid; i < m; i += 256) hl[i] = H[i];
        H = hl;
    }
    __syncthreads();
    if (tid < (1 << k)) {
        int g = 0;
        for (int c = 0; c < k; c++) if ((tid >> c) & 1) g += s_len[c];
        s_glen[tid] = g;
    }
    __syncthreads();
    double* __restrict__ base = tables + (int64_t)blockIdx.y * table_stride;
    const int La = s_len[a], starta = we.w.start[a];
    if (a == b) {
        double acc = 0.0;
        const int pairs = La * La;
        for (int flat = tid; flat < pairs; flat += 256) {
            const int e = flat / La, f = flat - e * La;
            if (f > e) acc += M2[(int64_t)(starta + e) * ld2 + starta + f] * (hn - H[f - e - 1]);
        }
        acc = wave_sum_w(acc);
        if (lane == 0) s_part[wave] = acc;
        __syncthreads();
        if (tid == 0) base[WT_SLICES * wt_q_size(k) + wt_p_size(k) + a] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        // the partial outside tables of scaffold a (k_win_outside*, earlier on this stream) are added into slice 0
        int nrg, nq;
        wt_split(La, m, n - m, nrg, nq);
        const int nz = nrg * nq;
        const int64_t q_size = wt_q_size(k);
        for (int e = tid; e < (2 << k) && nz > 1; e += 256) {
            if ((e >> a) & 1) continue;                     // (scaffold a is never in its own "before" set)
            double* __restrict__ q0 = base + (((int64_t)a * 2) << k) + e;
            double sum = q0[0];
            for (int z = 1; z < nz; z++) sum += q0[z * q_size];
            q0[0] = sum;
        }
        return;
    }
    const int Lb = s_len[b], startb = we.w.start[b];
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    double* __restrict__ P = base + WT_SLICES * wt_q_size(k) + ((((int64_t)a * k + b) * 4) << k);
    const int n_sets = k >= 2 ? 1 << (k - 2) : 1, n_entries = 4 * n_sets, pairs = La * Lb;
    for (int e0 = wave * WT_ACC; e0 < n_entries; e0 += 4 * WT_ACC) {
        double acc[WT_ACC];
        int gap[WT_ACC], ra[WT_ACC], rb[WT_ACC];
#pragma unroll
        for (int u = 0; u < WT_ACC; u++) {
            const int e = e0 + u < n_entries ? e0 + u : n_entries - 1;
            const int orient = e / n_sets, set = e - orient * n_sets;
            ra[u] = orient >> 1; rb[u] = orient & 1;
            gap[u] = s_glen[spread_bit(spread_bit(set, lo), hi)];
            acc[u] = 0.0;
        }
#pragma unroll 2
        for (int flat = lane; flat < pairs; flat += 64) {
            const int ea = flat / Lb, eb = flat - ea * Lb;
            const double v = M2[(int64_t)(starta + ea) * ld2 + startb + eb];
#pragma unroll
            for (int u = 0; u < WT_ACC; u++) {
                const int i = ra[u] ? La - 1 - ea : ea, jj = rb[u] ? Lb - 1 - eb : eb;
                const int d = La - i + gap[u] + jj;         // slot i of a ... gap ... slot jj of b
                acc[u] += v * (hn - H[d - 1]);
            }
        }
#pragma unroll
        for (int u = 0; u < WT_ACC; u++) {
            const double sum = wave_sum_w(acc[u]);
            const int e = e0 + u;
            if (lane == 0 && e < n_entries) {
                const int orient = e / n_sets, set = e - orient * n_sets;
                P[((int64_t)orient << k) | spread_bit(spread_bit(set, lo), hi)] = sum;
            }
        }
    }
}

// ---- candidates: k + k(k-1)/2 look-ups each ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_candidates(int k, const int8_t* __restrict__ orders,
                                                        const uint8_t* __restrict__ orients, int n_ori, int n_cand,
                                                        const double* __restrict__ tables, int64_t table_stride,
                                                        double* __restrict__ delta_all)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cand) return;
    const double* __restrict__ base = tables + (int64_t)blockIdx.y * table_stride;
    const double* __restrict__ Q = base;
    const int64_t q_size = wt_q_size(k);
    const double* __restrict__ P = base + WT_SLICES * q_size;
    const double* __restrict__ C = P + wt_p_size(k);
    const int8_t* __restrict__ ord = orders + (int64_t)(c / n_ori) * k;
    const uint8_t* __restrict__ ori = orients + (int64_t)(c % n_ori) * k;
    double sum = 0.0;
    for (int j = 0; j < k; j++) sum += C[j];
    int before = 0;
    for (int s = 0; s < k; s++) {
        const int a = ord[s], ra = ori[s] ? 1 : 0;
        const int64_t qi = ((int64_t)(a * 2 + ra) << k) | before;
        sum += Q[qi];                                       // (slice 0 holds the sum of the partial tables)
        int between = 0;
        for (int t = s + 1; t < k; t++) {
            const int b = ord[t], rb = ori[t] ? 1 : 0;
            sum += P[((((int64_t)a * k + b) * 4 + ra * 2 + rb) << k) | between];
            between |= 1 << b;
        }
        before |= 1 << a;
    }
    delta_all[(int64_t)blockIdx.y * n_cand + c] = sum;
}

static std::atomic<int> g_lds_out{0}, g_lds_pairs{0};

// tables: n_win * window_table_doubles(k) doubles of scratch; delta_all: n_win x (n_ord * n_ori)
void launch_p2_window_tables(const double* M2, int64_t ld2, const int32_t* pos2sel, int n, int k,
                             const WindowBatchEntry* wb, const WindowBatchEntry* h_wb, int n_win, int max_m, const int8_t* orders, const uint8_t* orients,
                             int n_ord, int n_ori, const double* H, double* tables, double* delta_all, hipStream_t s)
{
    if (n_win <= 0 || k < 1) return;
    const int64_t stride = window_table_doubles(k);
    const size_t h_all = (((size_t)n * sizeof(double)) + 15) & ~(size_t)15;
    const size_t h_win = (((size_t)max_m * sizeof(double)) + 15) & ~(size_t)15;
    // the outside table through the matrix cores (v_mfma_f64_16x16x4_f64) for windows of up to 512 bins and at most 256
    // table entries per scaffold (k <= 8); HICMI_P2_WINDOW_VALU=1 keeps the vector-ALU kernel (A/B)
    const int m_pad_max = (max_m + 15) & ~15;
    // grid z = the most partial tables any (window, scaffold) of the batch asks for
    int n_slices = 1;
    for (int w = 0; w < n_win; w++)
        for (int j = 0; j < k; j++) {
            int nrg, nq;
            wt_split(h_wb[w].w.len[j], h_wb[w].m, n - h_wb[w].m, nrg, nq);
            n_slices = nrg * nq > n_slices ? nrg * nq : n_slices;
        }
    const size_t lds_mfma = (size_t)16 * (size_t)(WM_QC + 1 > m_pad_max ? WM_QC + 1 : m_pad_max) * sizeof(double);
    static const bool valu_only = getenv("HICMI_P2_WINDOW_VALU") != nullptr;
    if (!valu_only && m_pad_max <= 16 * 4 * WM_MAX_TILES && k <= 8) {
        const bool h_lds = lds_mfma + h_all <= 150 * 1024;
        const size_t lds = lds_mfma + (h_lds ? h_all : 0);
        const int tiles = m_pad_max <= 128 ? 2 : (m_pad_max <= 256 ? 4 : 8);
#define HICMI_WIN_MFMA(HL, T)                                                                                                   \
        do {                                                                                                                    \
            static std::atomic<int> lds_set{0};                                                                                 \
            ensure_dynamic_lds(reinterpret_cast<const void*>(k_win_outside_mfma<HL, T>), lds_set, lds);                         \
            hipLaunchKernelGGL((k_win_outside_mfma<HL, T>), dim3(k, n_win, n_slices), dim3(256), lds, s, M2, ld2, pos2sel, n, k, \
                               wb, H, tables, stride, m_pad_max);                                                               \
        } while (0)
        if (h_lds) { if (tiles == 2) HICMI_WIN_MFMA(true, 2); else if (tiles == 4) HICMI_WIN_MFMA(true, 4); else HICMI_WIN_MFMA(true, 8); }
        else       { if (tiles == 2) HICMI_WIN_MFMA(false, 2); else if (tiles == 4) HICMI_WIN_MFMA(false, 4); else HICMI_WIN_MFMA(false, 8); }
#undef HICMI_WIN_MFMA
    }
    else if (h_all <= 96 * 1024) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_win_outside<true>), g_lds_out, h_all);
        hipLaunchKernelGGL(k_win_outside<true>, dim3(k, n_win, n_slices), dim3(256), h_all, s, M2, ld2, pos2sel, n, k, wb, H, tables, stride);
    } else {
        hipLaunchKernelGGL(k_win_outside<false>, dim3(k, n_win, n_slices), dim3(256), 0, s, M2, ld2, pos2sel, n, k, wb, H, tables, stride);
    }
    if (h_win <= 96 * 1024) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(k_win_pairs<true>), g_lds_pairs, h_win);
        hipLaunchKernelGGL(k_win_pairs<true>, dim3(k * k, n_win), dim3(256), h_win, s, M2, ld2, n, k, wb, H, tables, stride);
    } else {
        hipLaunchKernelGGL(k_win_pairs<false>, dim3(k * k, n_win), dim3(256), 0, s, M2, ld2, n, k, wb, H, tables, stride);
    }
    const int n_cand = n_ord * n_ori;
    hipLaunchKernelGGL(k_win_candidates, dim3((n_cand + 255) / 256, n_win), dim3(256), 0, s, k, orders, orients, n_ori, n_cand,
                       tables, stride, delta_all);
}

}  // namespace hicmi
