// k_sort.hip - per-row descending rank order of the reordered similarity matrix
// (scaffoldToChromosomes.py:1131-1132: convertMatrix(similarity) then numpy.argsort(axis=1)[:, ::-1]).
//
// The similarity matrix is never materialised: a workgroup computes the keys of its row straight
// from the contact matrix through the leaf order,
//     d = (1. - c / rowsum_np) + 1.      (S2C:147, as stored by Part 1)
//     s = rowsum_seq * (1. - (d - 1.))   (S2C:149)
// maps fp64 -> order-preserving uint64, and sorts (key, column) pairs ascending with a bitonic
// network: 8192-element tiles in LDS (80 KB), strides >= 8192 through a per-workgroup scratch
// buffer that stays in L2/Infinity Cache.  Ascending (key, column) order reversed is "descending
// similarity, ties by descending column" = numpy's stable argsort reversed - the tie rule this
// build fixes (NumPy's default unstable sort leaves tie order undefined).
#include "hicmi_internal.h"

namespace hicmi {

static constexpr int SORT_TILE = 8192;
static constexpr int SORT_THREADS = 1024;

int sort_padded_size(int n)
{
    int p = 2;
    while (p < n) p <<= 1;
    return p;
}

int sort_workgroups(int n)
{
    int g = 512;                                   // 2 workgroups per CU (80 KB LDS each)
    return n < g ? n : g;
}

size_t sort_scratch_bytes(int n)
{
    size_t P = (size_t)sort_padded_size(n);
    if (P < 32) P = 32;
    return (size_t)sort_workgroups(n) * P * (sizeof(uint64_t) + sizeof(uint16_t));
}

__device__ __forceinline__ uint64_t key_of(double s)
{
    s = s + 0.0;                                   // -0.0 -> +0.0 (numpy compares them equal)
    uint64_t u = (uint64_t)__double_as_longlong(s);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ double similarity(double c, double sig, double rs)
{
    double d = (1.0 - (c / sig)) + 1.0;
    return rs * (1.0 - (d - 1.0));
}

__device__ __forceinline__ void cmpx(uint64_t* __restrict__ k, uint16_t* __restrict__ ix, int i, int l, bool asc)
{
    uint64_t ka = k[i], kb = k[l];
    uint16_t ia = ix[i], ib = ix[l];
    bool gt = ka > kb || (ka == kb && ia > ib);
    if (gt == asc) { k[i] = kb; k[l] = ka; ix[i] = ib; ix[l] = ia; }
}

// all (k, j) stages with k in [k_lo, k_hi] and j < min(k, tile) on one LDS-resident tile
__device__ __forceinline__ void lds_stages(uint64_t* lk, uint16_t* li, int tile, int base, int k_lo, int k_hi, int tid)
{
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        int j0 = (k >> 1) < tile ? (k >> 1) : (tile >> 1);
        for (int j = j0; j >= 1; j >>= 1) {
            for (int p = tid; p < (tile >> 1); p += SORT_THREADS) {
                int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                cmpx(lk, li, i, i + j, ((base + i) & k) == 0);
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(SORT_THREADS) void k_sort_rows(
    const double* __restrict__ C, int64_t ldc, const int32_t* __restrict__ order, const double* __restrict__ np_sum,
    const double* __restrict__ seq_sum, int n, int P, uint64_t* __restrict__ skeys, uint16_t* __restrict__ sidx,
    uint16_t* __restrict__ R, int64_t ldr, int row_first, int row_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = P < SORT_TILE ? P : SORT_TILE;
    uint64_t* lk = reinterpret_cast<uint64_t*>(smem);
    uint16_t* li = reinterpret_cast<uint16_t*>(smem + (size_t)tile * sizeof(uint64_t));
    const int tid = threadIdx.x;
    const int ntiles = P / tile;
    uint64_t* gk = skeys + (size_t)blockIdx.x * (size_t)P;
    uint16_t* gi = sidx + (size_t)blockIdx.x * (size_t)P;

    for (int row = row_first + blockIdx.x * row_stride; row < n; row += gridDim.x * row_stride) {
        const int pa = order[row];
        const double sig = np_sum[pa], rs = seq_sum[pa];
        const double* __restrict__ crow = C + (int64_t)pa * ldc;
        uint16_t* __restrict__ out = R + (int64_t)row * ldr;

        // ---- phase 1: build keys and fully sort every tile (directions follow the global index)
        for (int t = 0; t < ntiles; t++) {
            const int base = t * tile;
            for (int e = tid; e < tile; e += SORT_THREADS) {
                int b = base + e;
                lk[e] = b < n ? key_of(similarity(crow[order[b]], sig, rs)) : ~0ull;
                li[e] = (uint16_t)b;
            }
            __syncthreads();
            lds_stages(lk, li, tile, base, 2, tile, tid);
            if (ntiles == 1) {
                for (int e = tid; e < n; e += SORT_THREADS) out[n - 1 - e] = li[e];
            } else {
                for (int e = tid; e < tile; e += SORT_THREADS) { gk[base + e] = lk[e]; gi[base + e] = li[e]; }
            }
            __syncthreads();
        }
        // ---- merge levels above the tile size
        for (int k = tile << 1; k <= P && ntiles > 1; k <<= 1) {
            for (int j = k >> 1; j >= tile; j >>= 1) {
                for (int p = tid; p < (P >> 1); p += SORT_THREADS) {
                    int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    cmpx(gk, gi, i, i + j, (i & k) == 0);
                }
                __syncthreads();
            }
            for (int t = 0; t < ntiles; t++) {
                const int base = t * tile;
                for (int e = tid; e < tile; e += SORT_THREADS) { lk[e] = gk[base + e]; li[e] = gi[base + e]; }
                __syncthreads();
                // remaining strides tile/2 .. 1 of level k
                for (int j = tile >> 1; j >= 1; j >>= 1) {
                    for (int p = tid; p < (tile >> 1); p += SORT_THREADS) {
                        int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                        cmpx(lk, li, i, i + j, ((base + i) & k) == 0);
                    }
                    __syncthreads();
                }
                if (k == P) {
                    for (int e = tid; e < tile; e += SORT_THREADS) {
                        int q = base + e;
                        if (q < n) out[n - 1 - q] = li[e];
                    }
                } else {
                    for (int e = tid; e < tile; e += SORT_THREADS) { gk[base + e] = lk[e]; gi[base + e] = li[e]; }
                }
                __syncthreads();
            }
        }
    }
}

// ---- register-blocked network --------------------------------------------------------------------
// The LDS version above moves every element through LDS at every one of the 105 stages of a 16384-element
// sort.  Here a lane keeps RB_E = 16 CONSECUTIVE elements (key, column) in registers:
//   strides 1..8      : compare-exchange inside the lane's registers, no memory at all
//   strides 16..512   : the partner is 1..32 lanes away in the same wave -> wave shuffles, no barrier
//   strides >= 1024   : the partner lane is in another wave -> the two lanes swap their blocks through
//                       LDS (80 KB, two halves of 8 elements); 10 of the 105 stages
// Rows longer than 16384 are sorted tile by tile and merged through the per-workgroup scratch as before.
// Measured at 16k / 32k bins: 16.3 -> 12.3 ms / 75 -> 55 ms.  (32 elements x 512 lanes and 16 x 512 were
// slower: 14.2 / 15.6 ms at 16k.)
static constexpr int RB_E = 16;                        // elements per lane
static constexpr int RB_T = 1024;                      // lanes per workgroup
static constexpr int RB_TILE = RB_E * RB_T;            // 16384

#define RB_LT(ka, ia, kb, ib) ((ka) < (kb) || ((ka) == (kb) && (ia) < (ib)))
#define RB_CMPX(a, b, asc)                                                                    \
    {                                                                                         \
        const bool gt_ = RB_LT(K[b], I[b], K[a], I[a]);                                       \
        if (gt_ == (asc)) { const uint64_t tk_ = K[a]; K[a] = K[b]; K[b] = tk_; const uint32_t ti_ = I[a]; I[a] = I[b]; I[b] = ti_; } \
    }

// stages j = min(j_max, RB_E/2) .. 1 of a level whose direction is the same for the whole lane
__device__ __forceinline__ void rb_inreg_tail(uint64_t (&K)[RB_E], uint32_t (&I)[RB_E], bool asc, int j_max)
{
#pragma unroll
    for (int j = RB_E / 2; j >= 1; j >>= 1) {
        if (j <= j_max) {
#pragma unroll
            for (int q = 0; q < RB_E; q++)
                if ((q & j) == 0) RB_CMPX(q, q | j, asc);
        }
    }
}

// partner lane = lane ^ M inside the wave; the lane with the clear bit keeps the minima when ascending.  The partner's
// (key, column) come through DPP / permlane swaps (xor_lane, hicmi_internal.h), not through the LDS crossbar.
template <int M>
__device__ __forceinline__ void rb_xor_stage(uint64_t (&K)[RB_E], uint32_t (&I)[RB_E], int lane, bool keep_min)
{
#pragma unroll
    for (int q = 0; q < RB_E; q++) {
        const uint32_t ol = xor_lane<M>((uint32_t)K[q], lane), oh = xor_lane<M>((uint32_t)(K[q] >> 32), lane);
        const uint64_t ok = ((uint64_t)oh << 32) | ol;
        const uint32_t oi = xor_lane<M>(I[q], lane);
        const bool other_lt = RB_LT(ok, oi, K[q], I[q]);
        if (other_lt == keep_min) { K[q] = ok; I[q] = oi; }
        // four exchanges in flight at a time: with all sixteen hoisted the kernel needed 204 bytes of scratch per lane - spills
        // inside the stages, 1.7 GB of write traffic for a 0.5 GB result in the round-1 counters
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void rb_shuffle_stage(uint64_t (&K)[RB_E], uint32_t (&I)[RB_E], int lane, int m, bool keep_min)
{
    switch (m) {
    case 1: rb_xor_stage<1>(K, I, lane, keep_min); break;
    case 2: rb_xor_stage<2>(K, I, lane, keep_min); break;
    case 4: rb_xor_stage<4>(K, I, lane, keep_min); break;
    case 8: rb_xor_stage<8>(K, I, lane, keep_min); break;
    case 16: rb_xor_stage<16>(K, I, lane, keep_min); break;
    default: rb_xor_stage<32>(K, I, lane, keep_min); break;
    }
}

// partner lane = tid ^ m in another wave: both write their block, both read the other's
__device__ __forceinline__ void rb_lds_stage(uint64_t (&K)[RB_E], uint32_t (&I)[RB_E], int tid, int m, bool keep_min,
                                             uint64_t* xk, uint16_t* xi)
{
    constexpr int H = RB_E / 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int q = 0; q < H; q++) { xk[q * RB_T + tid] = K[h * H + q]; xi[q * RB_T + tid] = (uint16_t)I[h * H + q]; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < H; q++) {
            const uint64_t ok = xk[q * RB_T + (tid ^ m)];
            const uint32_t oi = xi[q * RB_T + (tid ^ m)];
            const bool other_lt = RB_LT(ok, oi, K[h * H + q], I[h * H + q]);
            if (other_lt == keep_min) { K[h * H + q] = ok; I[h * H + q] = oi; }
        }
        __syncthreads();
    }
}

// stages j = j_start .. 1 of level k on the tile held by the workgroup (j_start < tile, k >= 2 * RB_E)
__device__ __forceinline__ void rb_tile_stages(uint64_t (&K)[RB_E], uint32_t (&I)[RB_E], int tid, int base, int k, int j_start,
                                               uint64_t* xk, uint16_t* xi)
{
    const bool asc = ((base + RB_E * tid) & k) == 0;
    for (int j = j_start; j >= RB_E; j >>= 1) {
        const int m = j / RB_E;
        const bool keep_min = ((tid & m) == 0) == asc;
        if (m < 64) rb_shuffle_stage(K, I, tid, m, keep_min);
        else rb_lds_stage(K, I, tid, m, keep_min, xk, xi);
    }
    rb_inreg_tail(K, I, asc, j_start);
}

// The lane's 16 ascending elements e0 .. e0 + 15 go to out[n - 1 - e]: 32 contiguous bytes in reverse order.  Written as two
// 16-byte stores when the row length keeps them aligned (2-byte stores to 16 different addresses per lane cost 3.4x the
// bytes in write traffic: 1.73 GB for a 0.5 GB matrix at 16k); element by element otherwise and at the row's end.
__device__ __forceinline__ void rb_store_reversed(uint16_t* __restrict__ out, int n, int e0, const uint32_t (&I)[RB_E])
{
    if (e0 + RB_E <= n && ((n - e0) & 7) == 0) {
        uint4 lo, hi;                                       // lo: positions n-16-e0 .. n-9-e0 = elements 15 .. 8
        lo.x = I[15] | (I[14] << 16); lo.y = I[13] | (I[12] << 16); lo.z = I[11] | (I[10] << 16); lo.w = I[9] | (I[8] << 16);
        hi.x = I[7] | (I[6] << 16); hi.y = I[5] | (I[4] << 16); hi.z = I[3] | (I[2] << 16); hi.w = I[1] | (I[0] << 16);
        uint4* dst = reinterpret_cast<uint4*>(out + (n - RB_E - e0));
        dst[0] = lo; dst[1] = hi;
    } else {
#pragma unroll
        for (int q = 0; q < RB_E; q++) { const int e = e0 + q; if (e < n) out[n - 1 - e] = (uint16_t)I[q]; }
    }
}

// Equal keys on two ADJACENT elements of the sorted row (real cells only): the order of such a pair is decided by the
// column labels, everything else in the row by the values alone.  Bit q of the lane's word says "ascending element
// e0 + q has the key of element e0 + q - 1"; the lane compares its 16 elements with each other and its first with the
// previous lane's last (through xk); `carry` (LDS) holds the last key of the previous tile of a long row.  The words go
// to bits[e0 / 16] (k_rank_rows_tied reads them); returns whether the row has any such pair.
__device__ __forceinline__ bool rb_row_tie_bits(const uint64_t (&K)[RB_E], int tid, int e0, int n, bool first_tile, uint64_t* xk,
                                                uint16_t* __restrict__ bits)
{
    __shared__ uint64_t s_carry;                            // (xk is rewritten by the next tile's exchange stages)
    uint32_t w = 0;
#pragma unroll
    for (int q = 1; q < RB_E; q++) w |= (uint32_t)((K[q] == K[q - 1]) && (e0 + q < n)) << q;
    xk[tid] = K[RB_E - 1];
    const uint64_t carried = s_carry;                       // written after the barriers of the previous call
    __syncthreads();
    const uint64_t before = tid > 0 ? xk[tid - 1] : carried;
    if ((tid > 0 || !first_tile) && before == K[0] && e0 < n) w |= 1u;
    if (tid == RB_T - 1) s_carry = K[RB_E - 1];
    if (e0 < n) bits[e0 / RB_E] = (uint16_t)w;
    return __syncthreads_or(w != 0) != 0;
}

// TIES: the kernel also flags the rows that hold equal keys (tie_flag[row] = 1, *tie_count = how many so far), writes
// the "same key as the element before" bits of every row (tie_bits, ld_bits 16-bit words per row) and gives up once
// more than tie_limit rows are flagged (the pre-sort of api.hip: start_presort uses tie_limit = n, i.e. never).
// row_list != nullptr: sort rows row_list[0 .. n_list) instead of row_first, row_first + row_stride, ...
template <bool TIES>
__global__ __launch_bounds__(RB_T) void k_sort_rows_rb(
    const double* __restrict__ C, int64_t ldc, const int32_t* __restrict__ order, const double* __restrict__ np_sum,
    const double* __restrict__ seq_sum, int n, int P, uint64_t* __restrict__ skeys, uint16_t* __restrict__ sidx,
    uint16_t* __restrict__ R, int64_t ldr, const int32_t* __restrict__ inv, int row_first, int row_stride,
    const int32_t* __restrict__ row_list, int n_list, uint8_t* __restrict__ tie_flag, unsigned* __restrict__ tie_count,
    unsigned tie_limit, uint16_t* __restrict__ tie_bits, int64_t ld_bits, int avoid_xcc, unsigned* __restrict__ row_counter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_next;
    // avoid_xcc >= 0 (the pre-sort beside the nn-chain): workgroups that land on that XCD leave at once - the chain's
    // single-wave workgroups want ALL of its CUs (k_nn_epoch_w1, LOCAL) - and the rows are dealt out by a counter
    // instead of by block index, so that the others take the whole matrix
    if (avoid_xcc == -2 || (avoid_xcc >= 0 && (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xfu) == avoid_xcc)) return;   // (-2: test hook)
    uint64_t* xk = reinterpret_cast<uint64_t*>(smem);                                   // (RB_E / 2) x RB_T keys
    uint16_t* xi = reinterpret_cast<uint16_t*>(smem + (size_t)(RB_E / 2) * RB_T * sizeof(uint64_t));
    const int tile = P < RB_TILE ? P : RB_TILE;            // P >= RB_E (launcher)
    const int ntiles = P / tile;
    const int tid0 = threadIdx.x;
    const bool live = RB_E * tid0 < tile;                   // short rows use the first tile / RB_E lanes
    uint64_t* gk = skeys + (size_t)blockIdx.x * (size_t)P;
    uint16_t* gi = sidx + (size_t)blockIdx.x * (size_t)P;
    uint64_t K[RB_E];
    uint32_t I[RB_E];

    const int n_rows = row_list ? n_list : (n - row_first + row_stride - 1) / row_stride;          // this shard's rows
    auto next_row = [&](int prev) -> int {
        if (!row_counter) return prev < 0 ? (int)blockIdx.x : prev + (int)gridDim.x;
        if (threadIdx.x == 0) s_next = (int)atomicAdd(row_counter, 1u);
        __syncthreads();
        const int v = s_next;
        __syncthreads();
        return v;
    };
    for (int it = next_row(-1); it < n_rows; it = next_row(it)) {
        const int row = row_list ? row_list[it] : row_first + it * row_stride;
        // the lane index is re-read through an opaque statement per row: otherwise lane-dependent addresses are hoisted out
        // of the row loop, spilled, and re-loaded / re-stored in every row (1.2 GB of scratch writes per 16k map)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (TIES) {
            if (__syncthreads_or(tid == 0 && *(volatile unsigned*)tie_count > tie_limit)) break;
        }
        bool row_tie = false;
        const int pa = order[row];
        const double sig = np_sum[pa], rs = seq_sum[pa];
        const double* __restrict__ crow = C + (int64_t)pa * ldc;
        uint16_t* __restrict__ out = R + (int64_t)row * ldr;
        for (int t = 0; t < ntiles; t++) {
            const int base = t * tile;
#pragma unroll
            for (int q = 0; q < RB_E; q++) {
                // the row is read in STORAGE order (coalesced) and each cell carries its position in the leaf order:
                // the same (key, position) pairs as gathering crow[order[b]], without pulling a 64-byte line per cell
                const int b = base + RB_E * tid + q;
                const bool real = live && b < n;
                K[q] = real ? key_of(similarity(crow[b], sig, rs)) : ~0ull;
                I[q] = real ? (uint32_t)inv[b] : (uint32_t)(b & 0xffff);
            }
            // levels 2 .. RB_E / 2: the direction alternates inside the lane (all indices are compile-time)
#pragma unroll
            for (int k = 2; k < RB_E; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j >= 1; j >>= 1) {
#pragma unroll
                    for (int q = 0; q < RB_E; q++)
                        if ((q & j) == 0) RB_CMPX(q, q | j, (q & k) == 0);
                }
            }
            rb_inreg_tail(K, I, ((base + RB_E * tid) & RB_E) == 0, RB_E / 2);                   // level RB_E: per lane
            for (int k = 2 * RB_E; k <= tile; k <<= 1) rb_tile_stages(K, I, tid, base, k, k >> 1, xk, xi);
            if (ntiles == 1) {
                int tid2 = tid;                             // (output addresses formed here, not before the sort and spilled)
                asm volatile("" : "+v"(tid2));
                if (live) rb_store_reversed(out, n, RB_E * tid2, I);
                if (TIES) row_tie = rb_row_tie_bits(K, tid, RB_E * tid, live ? n : 0, true, xk, tie_bits + (int64_t)row * ld_bits);
            } else {
#pragma unroll
                for (int q = 0; q < RB_E; q++) { gk[base + RB_E * tid + q] = K[q]; gi[base + RB_E * tid + q] = (uint16_t)I[q]; }
            }
        }
        // ---- merge levels above the tile size: strides >= tile in the scratch, the rest per tile in registers
        for (int k = tile << 1; k <= P && ntiles > 1; k <<= 1) {
            __syncthreads();
            for (int j = k >> 1; j >= tile; j >>= 1) {
                for (int p = tid; p < (P >> 1); p += RB_T) {
                    int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    cmpx(gk, gi, i, i + j, (i & k) == 0);
                }
                __syncthreads();
            }
            for (int t = 0; t < ntiles; t++) {
                const int base = t * tile;
#pragma unroll
                for (int q = 0; q < RB_E; q++) { K[q] = gk[base + RB_E * tid + q]; I[q] = gi[base + RB_E * tid + q]; }
                rb_tile_stages(K, I, tid, base, k, tile >> 1, xk, xi);
                if (k == P) {
                    int tid2 = tid;
                    asm volatile("" : "+v"(tid2));
                    rb_store_reversed(out, n, base + RB_E * tid2, I);
                    if (TIES) row_tie |= rb_row_tie_bits(K, tid, base + RB_E * tid, n, t == 0, xk, tie_bits + (int64_t)row * ld_bits);
                } else {
#pragma unroll
                    for (int q = 0; q < RB_E; q++) { gk[base + RB_E * tid + q] = K[q]; gi[base + RB_E * tid + q] = (uint16_t)I[q]; }
                }
            }
        }
        __syncthreads();
        if (TIES && row_tie && tid == 0) { tie_flag[row] = 1; atomicAdd(tie_count, 1u); }
    }
}

// ---- LSD radix sort ---------------------------------------------------------------------------------------------
// The bitonic networks above do O(n log^2 n) compare-exchanges on 64-bit keys: 105 stages for a 16384-element row, and
// rows beyond one tile go through global scratch at every merge level.  This kernel sorts (key, column) ascending by
// LSD radix, 4 bits per pass: the 16 bits of the column first (the tie rule: equal keys by ascending column), then the
// key digits that actually vary in the row (OR / AND of the row's keys: sign and high exponent bits never do).
// A lane keeps 16 consecutive elements in registers; a pass is
//   count the lane's digits (packed 8-bit counters, which also give each element its rank among the lane's equals),
//   per-lane counts -> LDS, digit-major [16][1024]; an exclusive scan over that array (each lane 16 consecutive entries,
//   wave scan, 16 wave totals) turns them into the lane's first output slot per digit,
//   scatter: the three components (key high, key low, column) go through LDS to their slots and are read back 16
//   consecutive per lane - stable, so the passes compose.
// ~18 passes x ~400 instructions instead of 105 stages x ~300.  Output: the RANK row directly - rank[row][col] =
// position of col in descending similarity (ties: larger column first) - scattered into an LDS image of the row and
// written with 16-byte stores; the argsort row itself (R) is never materialised (hicmi_get_rank_rows inverts a rank
// row on demand).  Rows longer than one 16384-element tile: every tile is sorted, parked in the workgroup's scratch,
// and an element's final position is its position in its own tile plus its lower bound in every other tile (one
// binary search per lane and tile, then a forward walk: the lane's 16 elements are consecutive, so are their bounds).
static constexpr int RX_T = 1024, RX_E = 8, RX_TILE = RX_T * RX_E;      // 8 elements per lane: 16 would not fit 128 registers
static constexpr size_t RX_CNT = 32768, RX_STAGE = (size_t)RX_TILE * 4, RX_STIDX = (size_t)RX_TILE * 2;
static constexpr size_t RX_LDS = RX_CNT + RX_STAGE + RX_STIDX;            // 80 KB: two workgroups per CU

__device__ __forceinline__ bool rx_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) { return ka < kb || (ka == kb && ia < ib); }

__global__ __launch_bounds__(RX_T) void k_sort_rows_radix(
    const double* __restrict__ C, int64_t ldc, const int32_t* __restrict__ order, const double* __restrict__ np_sum,
    const double* __restrict__ seq_sum, int n, uint64_t* __restrict__ skeys, uint16_t* __restrict__ sidx,
    uint16_t* __restrict__ rank, int64_t ldr, const int32_t* __restrict__ inv, int row_first, int row_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* cnt = reinterpret_cast<uint16_t*>(smem);                       // [16][RX_T]
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem + RX_CNT);            // [RX_TILE]
    uint16_t* stidx = reinterpret_cast<uint16_t*>(smem + RX_CNT + RX_STAGE); // [RX_TILE]
    uint16_t* image = reinterpret_cast<uint16_t*>(smem);                     // [n] in the last phase (aliases everything)
    __shared__ uint32_t s_wsum[16];
    __shared__ unsigned long long s_or[16], s_and[16];
    const int tid0 = threadIdx.x, lane = tid0 & 63, wave = tid0 >> 6;
    const int ntiles = (n + RX_TILE - 1) / RX_TILE;
    uint64_t* gk = skeys + (size_t)blockIdx.x * (size_t)ntiles * RX_TILE;
    uint16_t* gi = sidx + (size_t)blockIdx.x * (size_t)ntiles * RX_TILE;
    uint32_t KH[RX_E], KL[RX_E], I[RX_E];                    // key (high, low word) and column of the lane's 16 elements

    for (int row = row_first + blockIdx.x * row_stride; row < n; row += gridDim.x * row_stride) {
        const int pa = order[row];
        const double sig = np_sum[pa], rs = seq_sum[pa];
        const double* __restrict__ crow = C + (int64_t)pa * ldc;
        for (int t = 0; t < ntiles; t++) {
            // the lane index is re-read through an opaque statement in every tile and every pass: otherwise the compiler
            // hoists dozens of lane-dependent addresses out of the loops and spills them (392 bytes of scratch measured)
            int tid = tid0;
            asm volatile("" : "+v"(tid));
            const int base = t * RX_TILE;
            unsigned long long o_acc = 0ull, a_acc = ~0ull;
#pragma unroll
            for (int q = 0; q < RX_E; q++) {
                // storage order (coalesced); each cell carries its position in the leaf order (= its column label)
                const int b = base + RX_E * tid + q;
                const bool real = b < n;
                const uint64_t k = real ? key_of(similarity(crow[b], sig, rs)) : ~0ull;    // padding sorts last in every pass
                KH[q] = (uint32_t)(k >> 32); KL[q] = (uint32_t)k;
                I[q] = real ? (uint32_t)inv[b] : 0xffffu;
                if (real) { o_acc |= k; a_acc &= k; }
            }
            // key bits that differ somewhere in the tile
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { o_acc |= __shfl_xor(o_acc, off, 64); a_acc &= __shfl_xor(a_acc, off, 64); }
            __syncthreads();                                    // the previous tile / row is done with the LDS
            if (lane == 0) { s_or[wave] = o_acc; s_and[wave] = a_acc; }
            __syncthreads();
            unsigned long long varying;
            {
                unsigned long long oo = 0ull, aa = ~0ull;
#pragma unroll
                for (int w = 0; w < 16; w++) { oo |= s_or[w]; aa &= s_and[w]; }
                varying = oo ^ aa;
            }
            for (int pass = 0; pass < 20; pass++) {
                // the word this pass's digit comes from: the column (4 passes), the key's low word (8), its high word (8)
                const int word = pass < 4 ? 0 : (pass < 12 ? 1 : 2);
                const int shift = word == 0 ? 4 * pass : (word == 1 ? 4 * (pass - 4) : 4 * (pass - 12));
                if (word == 0) { if (pass > 0 && ((n - 1) >> shift) == 0) continue; }       // columns are below n
                else if (((varying >> (4 * (pass - 4))) & 15ull) == 0ull) continue;           // this digit is the same everywhere
                asm volatile("" : "+v"(tid));
                // ---- 1. the lane's digit counts; rk = rank of each element among the lane's equal digits
                unsigned long long c0 = 0ull, c1 = 0ull, rk = 0ull;
#pragma unroll
                for (int q = 0; q < RX_E; q++) {
                    const uint32_t src = word == 0 ? I[q] : (word == 1 ? KL[q] : KH[q]);
                    const int d = (int)((src >> shift) & 15u);
                    const int sh = (d & 7) * 8;
                    const unsigned long long cur = d < 8 ? c0 : c1;
                    rk |= ((cur >> sh) & 0xffull) << (4 * q);          // < 16: four bits are enough
                    if (d < 8) c0 += 1ull << sh; else c1 += 1ull << sh;
                }
#pragma unroll
                for (int d = 0; d < 16; d++) cnt[d * RX_T + tid] = (uint16_t)(((d < 8 ? c0 : c1) >> ((d & 7) * 8)) & 0xffull);
                __syncthreads();
                // ---- 2. exclusive scan over cnt[] (16 digits x 1024 lanes) in linear order: this lane owns entries 16 tid .. 16 tid + 15
                {
                    const uint4 v0 = *reinterpret_cast<const uint4*>(cnt + 16 * tid), v1 = *reinterpret_cast<const uint4*>(cnt + 16 * tid + 8);
                    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    uint32_t sum = 0;
#pragma unroll
                    for (int u = 0; u < 8; u++) sum += (w[u] & 0xffffu) + (w[u] >> 16);
                    uint32_t incl = sum;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
                    if (lane == 63) s_wsum[wave] = incl;
                    __syncthreads();
                    uint32_t run = incl - sum;
#pragma unroll
                    for (int ww = 0; ww < 16; ww++) if (ww < wave) run += s_wsum[ww];
                    uint32_t o[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const uint32_t e0 = run; run += w[u] & 0xffffu; const uint32_t e1 = run; run += w[u] >> 16; o[u] = e0 | (e1 << 16); }
                    uint4 o0, o1;
                    o0.x = o[0]; o0.y = o[1]; o0.z = o[2]; o0.w = o[3]; o1.x = o[4]; o1.y = o[5]; o1.z = o[6]; o1.w = o[7];
                    *reinterpret_cast<uint4*>(cnt + 16 * tid) = o0; *reinterpret_cast<uint4*>(cnt + 16 * tid + 8) = o1;
                }
                __syncthreads();
                // ---- 3. destinations, then the three components through LDS (stable scatter, 16 consecutive back)
                uint32_t dst2[RX_E / 2];                        // two 14-bit destinations per register
#pragma unroll
                for (int q = 0; q < RX_E; q++) {
                    const uint32_t src = word == 0 ? I[q] : (word == 1 ? KL[q] : KH[q]);
                    const int d = (int)((src >> shift) & 15u);
                    const uint32_t dq = (uint32_t)cnt[d * RX_T + tid] + (uint32_t)((rk >> (4 * q)) & 15ull);
                    if (q & 1) dst2[q >> 1] |= dq << 16; else dst2[q >> 1] = dq;
                }
#define RX_DST(q) ((dst2[(q) >> 1] >> (((q) & 1) * 16)) & 0xffffu)
#pragma unroll
                for (int q = 0; q < RX_E; q++) stage[RX_DST(q)] = KH[q];
                __syncthreads();
#pragma unroll
                for (int u = 0; u < RX_E / 4; u++) {
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + RX_E * tid + 4 * u);
                    KH[4 * u] = v.x; KH[4 * u + 1] = v.y; KH[4 * u + 2] = v.z; KH[4 * u + 3] = v.w;
                }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < RX_E; q++) { stage[RX_DST(q)] = KL[q]; stidx[RX_DST(q)] = (uint16_t)I[q]; }
#undef RX_DST
                __syncthreads();
#pragma unroll
                for (int u = 0; u < RX_E / 4; u++) {
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + RX_E * tid + 4 * u);
                    KL[4 * u] = v.x; KL[4 * u + 1] = v.y; KL[4 * u + 2] = v.z; KL[4 * u + 3] = v.w;
                }
                {
                    const uint4 v0 = *reinterpret_cast<const uint4*>(stidx + RX_E * tid);          // RX_E = 8 columns
                    const uint32_t w[4] = {v0.x, v0.y, v0.z, v0.w};
#pragma unroll
                    for (int u = 0; u < 4; u++) { I[2 * u] = w[u] & 0xffffu; I[2 * u + 1] = w[u] >> 16; }
                }
                __syncthreads();
            }
            if (ntiles > 1) {
#pragma unroll
                for (int q = 0; q < RX_E; q++) { gk[base + RX_E * tid + q] = ((uint64_t)KH[q] << 32) | KL[q]; gi[base + RX_E * tid + q] = (uint16_t)I[q]; }
            }
        }
        __syncthreads();                                        // the image aliases the pass buffers
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (ntiles == 1) {
#pragma unroll
            for (int q = 0; q < RX_E; q++) { const int pos = RX_E * tid + q; if (pos < n) image[I[q]] = (uint16_t)(n - 1 - pos); }
        } else {
            __threadfence_block();
            for (int t = 0; t < ntiles; t++) {
                asm volatile("" : "+v"(tid));
                const int base = t * RX_TILE, len = n - base < RX_TILE ? n - base : RX_TILE;
                if (RX_E * tid >= len) continue;
                uint32_t lb[RX_E];                               // sum over the other tiles of the element's lower bound there
#pragma unroll
                for (int q = 0; q < RX_E; q++) lb[q] = 0;
                for (int o = 0; o < ntiles; o++) {
                    if (o == t) continue;
                    const uint64_t* __restrict__ ok = gk + o * RX_TILE;
                    const uint16_t* __restrict__ oi = gi + o * RX_TILE;
                    const int olen = n - o * RX_TILE < RX_TILE ? n - o * RX_TILE : RX_TILE;
                    const uint64_t k0 = gk[base + RX_E * tid];
                    const uint32_t i0 = gi[base + RX_E * tid];
                    int lo = 0, hi2 = olen;                      // lower bound of the lane's first element in tile o
                    while (lo < hi2) {
                        const int mid = (lo + hi2) >> 1;
                        if (rx_less(ok[mid], oi[mid], k0, i0)) lo = mid + 1; else hi2 = mid;
                    }
                    // the other elements: walk forward (consecutive elements have neighbouring bounds).  (key, column)
                    // pairs are all different, so "strictly less" counts exactly the elements that come before ours
#pragma unroll
                    for (int q = 0; q < RX_E; q++) {
                        if (q > 0 && RX_E * tid + q < len) {
                            const uint64_t kq = gk[base + RX_E * tid + q];
                            const uint32_t iq = gi[base + RX_E * tid + q];
                            while (lo < olen && rx_less(ok[lo], oi[lo], kq, iq)) lo++;
                        }
                        lb[q] += (uint32_t)lo;
                    }
                }
#pragma unroll
                for (int q = 0; q < RX_E; q++)
                    if (RX_E * tid + q < len) image[gi[base + RX_E * tid + q]] = (uint16_t)(n - 1 - (RX_E * tid + q) - (int)lb[q]);
            }
        }
        __syncthreads();
        uint16_t* __restrict__ out = rank + (int64_t)row * ldr;
        for (int e = tid * 8; e < n; e += RX_T * 8) *reinterpret_cast<uint4*>(out + e) = *reinterpret_cast<const uint4*>(image + e);
        __syncthreads();
    }
}

static constexpr int RX_MAXWG = 512;
size_t sort_radix_scratch_bytes(int n)
{
    const size_t ntiles = ((size_t)n + RX_TILE - 1) / RX_TILE;
    return ntiles > 1 ? (size_t)RX_MAXWG * ntiles * RX_TILE * (sizeof(uint64_t) + sizeof(uint16_t)) : 16;
}

// rank[row][col] = position of col in row's descending order, for this shard's rows (no argsort rows, no inversion pass)
void launch_rank_rows_radix(const double* C, int64_t ldc, const int32_t* order, const int32_t* inv, const double* np_sum,
                            const double* seq_sum, int n, void* scratch, uint16_t* rank, int64_t ldr, int row_first,
                            int row_stride, hipStream_t s)
{
    const int ntiles = (n + RX_TILE - 1) / RX_TILE;
    const int mine = n > row_first ? (n - row_first + row_stride - 1) / row_stride : 0;
    if (mine <= 0) return;
    // LDS: the pass buffers (80 KB), or the row's rank image if that is larger (2 bytes per column)
    const size_t image = (((size_t)n * 2) + 15) & ~(size_t)15;
    const size_t lds = image > RX_LDS ? image : RX_LDS;
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int wgs = mine < 256 * per_cu ? mine : 256 * per_cu;
    uint64_t* skeys = reinterpret_cast<uint64_t*>(scratch);
    uint16_t* sidx = reinterpret_cast<uint16_t*>(skeys + (size_t)RX_MAXWG * (size_t)ntiles * RX_TILE);
    static std::atomic<int> have{0};
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_sort_rows_radix), have, lds);
    hipLaunchKernelGGL(k_sort_rows_radix, dim3(wgs), dim3(RX_T), lds, s, C, ldc, order, np_sum, seq_sum, n, skeys, sidx, rank, ldr,
                       inv, row_first, row_stride);
}

void launch_sort_rows(const double* C, int64_t ldc, const int32_t* order, const int32_t* inv, const double* np_sum,
                      const double* seq_sum, int n, void* scratch, uint16_t* R, int64_t ldr, int row_first, int row_stride,
                      hipStream_t s, const SortExtras& x)
{
    unsigned* const tie_rows = x.tie_count;
    const int max_workgroups = x.max_workgroups;
    static const bool lds_network = getenv("HICMI_SORT_LDS") != nullptr;      // A/B switch: the LDS-resident network
    if (!lds_network || tie_rows || x.row_list) {
        int P = sort_padded_size(n);
        if (P < RB_E) P = RB_E;
        int wgs = sort_workgroups(n);
        if (max_workgroups > 0 && wgs > max_workgroups) wgs = max_workgroups;
        if (x.avoid_xcc != -1 && x.row_counter) wgs = (wgs * 8 + 6) / 7;      // an eighth of the grid leaves at once
        if (wgs > sort_workgroups(n)) wgs = sort_workgroups(n);               // (the scratch is sized for that many)
        if (x.row_list && wgs > x.n_list) wgs = x.n_list;
        if (wgs < 1) return;
        const int avoid = x.row_counter ? x.avoid_xcc : -1;
        uint64_t* skeys = reinterpret_cast<uint64_t*>(scratch);
        uint16_t* sidx = reinterpret_cast<uint16_t*>(skeys + (size_t)wgs * (size_t)P);
        const size_t lds = (size_t)(RB_E / 2) * RB_T * (sizeof(uint64_t) + sizeof(uint16_t));
        if (tie_rows) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_rows_rb<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_sort_rows_rb<true>, dim3(wgs), dim3(RB_T), lds, s, C, ldc, order, np_sum, seq_sum, n, P, skeys,
                               sidx, R, ldr, inv, row_first, row_stride, x.row_list, x.n_list, x.tie_flag, x.tie_count, x.tie_limit,
                               x.tie_bits, x.ld_bits, avoid, x.row_counter);
            return;
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_rows_rb<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sort_rows_rb<false>, dim3(wgs), dim3(RB_T), lds, s, C, ldc, order, np_sum, seq_sum, n, P, skeys,
                           sidx, R, ldr, inv, row_first, row_stride, x.row_list, x.n_list, (uint8_t*)nullptr, (unsigned*)nullptr, 0u,
                           (uint16_t*)nullptr, (int64_t)0, avoid, x.row_counter);
        return;
    }
    const int P = sort_padded_size(n);
    const int tile = P < SORT_TILE ? P : SORT_TILE;
    const int wgs = sort_workgroups(n);
    uint64_t* skeys = reinterpret_cast<uint64_t*>(scratch);
    uint16_t* sidx = reinterpret_cast<uint16_t*>(skeys + (size_t)wgs * (size_t)P);
    size_t lds = (size_t)tile * (sizeof(uint64_t) + sizeof(uint16_t));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_sort_rows, dim3(wgs), dim3(SORT_THREADS), lds, s, C, ldc, order, np_sum, seq_sum, n, P, skeys,
                       sidx, R, ldr, row_first, row_stride);
}

// rank[row][R[row][k]] = k : scatter inside LDS (2 bytes per bin), coalesced in and out.
__global__ __launch_bounds__(1024) void k_rank_invert(const uint16_t* __restrict__ R, uint16_t* __restrict__ rank,
                                                      int64_t ldr, int n, int row_first, int row_stride,
                                                      const int32_t* __restrict__ row_list, int n_list)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* inv = reinterpret_cast<uint16_t*>(smem);
    const int n_rows = row_list ? n_list : (n - row_first + row_stride - 1) / row_stride;
    for (int it = blockIdx.x; it < n_rows; it += gridDim.x) {
        const int row = row_list ? row_list[it] : row_first + it * row_stride;
        const uint16_t* __restrict__ r = R + (int64_t)row * ldr;
        for (int k = threadIdx.x; k < n; k += 1024) inv[r[k]] = (uint16_t)k;
        __syncthreads();
        uint16_t* __restrict__ o = rank + (int64_t)row * ldr;
        for (int b = threadIdx.x; b < n; b += 1024) o[b] = inv[b];
        __syncthreads();
    }
}

void launch_rank_invert(const uint16_t* R, uint16_t* rank, int64_t ldr, int n, int row_first, int row_stride, hipStream_t s,
                        const int32_t* row_list, int n_list)
{
    if (row_list && n_list < 1) return;
    size_t lds = ((size_t)n * sizeof(uint16_t) + 15) & ~(size_t)15;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_invert), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = n < 1024 ? n : 1024;
    if (row_list && grid > n_list) grid = n_list;
    hipLaunchKernelGGL(k_rank_invert, dim3(grid), dim3(1024), lds, s, R, rank, ldr, n, row_first, row_stride, row_list, n_list);
}

// rank[a][b] = rank_storage[order[a]][order[b]]: the rank rows computed in STORAGE labels (rows and columns numbered as the
// matrix is stored) re-addressed by the leaf order.  Valid for rows without equal keys - there the position of a column
// depends on the values only.  One workgroup per row: the source row into LDS (coalesced), gathered through the order.
__global__ __launch_bounds__(1024) void k_rank_relabel(const uint16_t* __restrict__ rank_storage, uint16_t* __restrict__ rank,
                                                       int64_t ldr, int n, const int32_t* __restrict__ order, int row_first,
                                                       int row_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* src = reinterpret_cast<uint16_t*>(smem);
    const int n8 = (n + 7) & ~7;                            // ldr is a multiple of 64: whole 16-byte groups stay inside the row
    for (int a = row_first + blockIdx.x * row_stride; a < n; a += gridDim.x * row_stride) {
        const uint4* __restrict__ in = reinterpret_cast<const uint4*>(rank_storage + (int64_t)order[a] * ldr);
        for (int g = threadIdx.x; g < n8 / 8; g += 1024) reinterpret_cast<uint4*>(src)[g] = in[g];
        __syncthreads();
        uint4* __restrict__ o = reinterpret_cast<uint4*>(rank + (int64_t)a * ldr);
        for (int g = threadIdx.x; g < n8 / 8; g += 1024) {
            uint32_t w[4];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                const int b0 = 8 * g + 2 * h, b1 = b0 + 1;
                const uint32_t lo = b0 < n ? src[order[b0]] : 0u, hi = b1 < n ? src[order[b1]] : 0u;
                w[h] = lo | (hi << 16);
            }
            o[g] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();
    }
}

void launch_rank_relabel(const uint16_t* rank_storage, uint16_t* rank, int64_t ldr, int n, const int32_t* order, int row_first,
                         int row_stride, hipStream_t s)
{
    size_t lds = ((size_t)((n + 7) & ~7) * sizeof(uint16_t) + 15) & ~(size_t)15;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_relabel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int grid = n < 2048 ? n : 2048;
    hipLaunchKernelGGL(k_rank_relabel, dim3(grid), dim3(1024), lds, s, rank_storage, rank, ldr, n, order, row_first, row_stride);
}

__global__ __launch_bounds__(256) void k_similarity_row(const double* __restrict__ C, int64_t ldc,
                                                        const int32_t* __restrict__ order,
                                                        const double* __restrict__ np_sum,
                                                        const double* __restrict__ seq_sum, int n, int row,
                                                        double* __restrict__ out)
{
    int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    int pa = order[row];
    out[b] = similarity(C[(int64_t)pa * ldc + order[b]], np_sum[pa], seq_sum[pa]);
}

void launch_similarity_row(const double* C, int64_t ldc, const int32_t* order, const double* np_sum,
                           const double* seq_sum, int n, int row, double* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_similarity_row, dim3((n + 255) / 256), dim3(256), 0, s, C, ldc, order, np_sum, seq_sum, n, row,
                       out);
}

}  // namespace hicmi
