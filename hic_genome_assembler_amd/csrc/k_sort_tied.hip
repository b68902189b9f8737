// k_sort_tied.hip - rank rows of the rows that hold EQUAL similarities, from the sort made beside the nn-chain.
//
// Replaces, for those rows, the second half of numpy.argsort(similarity, axis=1)[:, ::-1] (scaffoldToChromosomes.py:1132
// after reorderMatrix, S2C:157-163): the values were already sorted - in storage numbering, before the leaf order was
// known (api.hip: start_presort) - and what is left is the order INSIDE each run of equal values, which numpy's stable
// sort decides by the column's position in the leaf order (reversed: larger leaf position first).
//
// Input per row: the storage-label argsort row R_s (descending similarity) and one bit per sorted element, "same key as
// the element before" (k_sort_rows_rb<true>).  In ascending order e = n - 1 - position:
//     run(e)  = first ascending index of e's run of equal keys      (prefix maximum over the bits)
//     key32   = run(e) << 16 | leaf position of the column at e     (ascending = the final ascending order)
// and sorting key32 ascending gives the row: rank[leaf position] = n - 1 - (index in the sorted sequence).
//
// The sort is the register-blocked bitonic network of k_sort.hip on 32-bit keys WITHOUT a payload (the leaf position is
// the key's low half): a compare-exchange is v_min_u32 / v_max_u32 instead of a 64-bit compare and six selects, and a lane
// can hold 64 elements, so even a 65,536-column row is sorted in the registers of one workgroup - no scratch round trips.
// HBM-bound in the end (2 B in, 2 B out per cell plus the gathers of the leaf positions from L2).
#include "hicmi_internal.h"

namespace hicmi {

static constexpr int CK_T = 1024;                          // lanes per workgroup
static constexpr int CK_CHUNK = 16;                        // keys per lane exchanged through LDS at a time (64 KB)

template <int E>
__device__ __forceinline__ void ck_inreg_tail(uint32_t (&K)[E], bool asc, int j_max)
{
#pragma unroll
    for (int j = E / 2; j >= 1; j >>= 1) {
        if (j <= j_max) {
#pragma unroll
            for (int q = 0; q < E; q++)
                if ((q & j) == 0) {
                    const uint32_t lo = K[q] < K[q | j] ? K[q] : K[q | j];
                    const uint32_t hi = K[q] < K[q | j] ? K[q | j] : K[q];
                    K[q] = asc ? lo : hi;
                    K[q | j] = asc ? hi : lo;
                }
        }
    }
}

template <int E, int M>
__device__ __forceinline__ void ck_xor_stage(uint32_t (&K)[E], int lane, bool keep_min)
{
#pragma unroll
    for (int q = 0; q < E; q++) {
        const uint32_t o = xor_lane<M>(K[q], lane);
        const uint32_t lo = o < K[q] ? o : K[q], hi = o < K[q] ? K[q] : o;
        K[q] = keep_min ? lo : hi;
    }
}

// partner lane = lane ^ m inside the wave (m = 1 .. 32)
template <int E>
__device__ __forceinline__ void ck_shuffle_stage(uint32_t (&K)[E], int lane, int m, bool keep_min)
{
    switch (m) {
    case 1: ck_xor_stage<E, 1>(K, lane, keep_min); break;
    case 2: ck_xor_stage<E, 2>(K, lane, keep_min); break;
    case 4: ck_xor_stage<E, 4>(K, lane, keep_min); break;
    case 8: ck_xor_stage<E, 8>(K, lane, keep_min); break;
    case 16: ck_xor_stage<E, 16>(K, lane, keep_min); break;
    default: ck_xor_stage<E, 32>(K, lane, keep_min); break;
    }
}

template <int E>
__device__ __forceinline__ void ck_lds_stage(uint32_t (&K)[E], int tid, int m, bool keep_min, uint32_t* xk)
{
#pragma unroll
    for (int c = 0; c < E / CK_CHUNK; c++) {
#pragma unroll
        for (int q = 0; q < CK_CHUNK; q++) xk[q * CK_T + tid] = K[c * CK_CHUNK + q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CK_CHUNK; q++) {
            const uint32_t o = xk[q * CK_T + (tid ^ m)];
            const uint32_t mine = K[c * CK_CHUNK + q];
            const uint32_t lo = o < mine ? o : mine, hi = o < mine ? mine : o;
            K[c * CK_CHUNK + q] = keep_min ? lo : hi;
        }
        __syncthreads();
    }
}

// One workgroup per listed row; E * 1024 >= padded row length P.  Dynamic LDS: max(64 KB exchange, 2 B x n rank image).
template <int E>
__global__ __launch_bounds__(CK_T) void k_rank_rows_tied(
    const uint16_t* __restrict__ R_storage, const uint16_t* __restrict__ tie_bits, int64_t ld_bits,
    const int32_t* __restrict__ order, const int32_t* __restrict__ inv, int n, int P, const int32_t* __restrict__ row_list,
    int n_list, uint16_t* __restrict__ rank, int64_t ldr, const int32_t* __restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* xk = reinterpret_cast<uint32_t*>(smem);
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);     // after the sort: the rank row
    __shared__ uint32_t s_wmax[CK_T / 64];
    const int tid0 = threadIdx.x;
    uint32_t K[E];

    for (int it = blockIdx.x; it < n_list; it += gridDim.x) {
        if (done && done[it]) continue;                     // (finished by k_rank_rows_short_runs)
        int tid = tid0;
        asm volatile("" : "+v"(tid));                       // (see k_sort_rows_rb: keeps lane addresses out of the row loop)
        const int a = row_list[it];                         // row in leaf numbering
        const int srow = order[a];                          // ... is this storage row
        const uint16_t* __restrict__ rs = R_storage + (int64_t)srow * ldr;
        const uint16_t* __restrict__ bits = tie_bits + (int64_t)srow * ld_bits;
        const int e0 = E * tid;
        const bool live = e0 < P;

        // ---- run starts: prefix maximum of (bit ? 0 : e) over ascending e
        uint32_t run_local = 0;                             // maximum over the lane's own elements
#pragma unroll
        for (int g = 0; g < E / 16; g++) {
            const int eg = e0 + 16 * g;
            const uint32_t w = (live && eg < n) ? bits[eg / 16] : 0u;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int e = eg + q;
                const bool same = (w >> q) & 1u;
                if (!same && e < n) run_local = (uint32_t)e;          // e ascends: the last start wins
                K[16 * g + q] = same ? 0xffffffffu : (uint32_t)e;     // for now: own start or "inherits"
            }
        }
        // exclusive prefix maximum across lanes: wave scan, then the 16 wave maxima
        uint32_t incl = run_local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((tid & 63) >= d && o > incl) incl = o;
        }
        if ((tid & 63) == 63) s_wmax[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = (uint32_t)__shfl_up((int)incl, 1, 64);
        if ((tid & 63) == 0) before = 0;
        for (int w = 0; w < (tid >> 6); w++) before = s_wmax[w] > before ? s_wmax[w] : before;
        // ---- composite keys
        uint32_t cur = before;
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int e = e0 + q;
            if (K[q] != 0xffffffffu) cur = K[q];
            uint32_t key = 0xffffffffu;                     // pads sort to the end
            if (live && e < n) key = (cur << 16) | (uint32_t)inv[rs[n - 1 - e]];
            K[q] = key;
        }
        // ---- bitonic network, ascending over all P elements
#pragma unroll
        for (int k = 2; k < E; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j >= 1; j >>= 1) {
#pragma unroll
                for (int q = 0; q < E; q++)
                    if ((q & j) == 0) {
                        const uint32_t lo = K[q] < K[q | j] ? K[q] : K[q | j];
                        const uint32_t hi = K[q] < K[q | j] ? K[q | j] : K[q];
                        K[q] = (q & k) == 0 ? lo : hi;
                        K[q | j] = (q & k) == 0 ? hi : lo;
                    }
            }
        }
        ck_inreg_tail<E>(K, (e0 & E) == 0, E / 2);
        for (int k = 2 * E; k <= P; k <<= 1) {
            const bool asc = (e0 & k) == 0;
            for (int j = k >> 1; j >= E; j >>= 1) {
                const int m = j / E;
                const bool keep_min = ((tid & m) == 0) == asc;
                if (m < 64) ck_shuffle_stage<E>(K, tid, m, keep_min);
                else ck_lds_stage<E>(K, tid, m, keep_min, xk);
            }
            ck_inreg_tail<E>(K, asc, E / 2);
        }
        // ---- rank image: the element at ascending index e is column (key & 0xffff) and takes position n - 1 - e
        __syncthreads();                                    // (xk of the last exchange is free)
        if (live) {
#pragma unroll
            for (int q = 0; q < E; q++) {
                const int e = e0 + q;
                if (e < n) img[K[q] & 0xffffu] = (uint16_t)(n - 1 - e);
            }
        }
        __syncthreads();
        uint4* __restrict__ o = reinterpret_cast<uint4*>(rank + (int64_t)a * ldr);
        const int groups = (n + 7) / 8;                     // ldr is a multiple of 64: the last group stays inside the row
        for (int g = tid; g < groups; g += CK_T) o[g] = reinterpret_cast<const uint4*>(img)[g];
        __syncthreads();
    }
}

// ---- rows whose runs of equal keys are all SHORT --------------------------------------------------------------------
// fp32-valued contacts give every row a few collisions - pairs, now and then a triple - and the full network above then
// does all log^2 stages on a sequence that is sorted already except inside those runs (64,000 bins: 73 ms per map).  A run
// of at most 8 elements lies inside an aligned block of 16 or inside a block shifted by 8, and sorting a block only
// permutes the elements inside its runs (the run number is the key's high half, and the runs are in order): two rounds
// of 16-element bitonic sorts in registers - 20 compare-exchange layers and two 8-key hand-overs between neighbouring
// lanes instead of 136 layers, about 50 of them through LDS.  Rows with a longer run (sparse maps: the run of zeros) are
// left to k_rank_rows_tied, which skips the rows marked done here.
static constexpr int CKS_MAXRUN = 8;

template <int E>
__device__ __forceinline__ void cks_sort_blocks16(uint32_t (&A)[E])
{
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j >= 1; j >>= 1) {
#pragma unroll
            for (int q = 0; q < E; q++)
                if ((q & j) == 0) {
                    const uint32_t lo = A[q] < A[q | j] ? A[q] : A[q | j];
                    const uint32_t hi = A[q] < A[q | j] ? A[q | j] : A[q];
                    const bool asc = k == 16 || (q & k) == 0;         // the last merge: every block ascending
                    A[q] = asc ? lo : hi;
                    A[q | j] = asc ? hi : lo;
                }
        }
    }
}

template <int E>
__global__ __launch_bounds__(CK_T) void k_rank_rows_short_runs(
    const uint16_t* __restrict__ R_storage, const uint16_t* __restrict__ tie_bits, int64_t ld_bits,
    const int32_t* __restrict__ order, const int32_t* __restrict__ inv, int n, int P, const int32_t* __restrict__ row_list,
    int n_list, uint16_t* __restrict__ rank, int64_t ldr, int32_t* __restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* halo = reinterpret_cast<uint32_t*>(smem);    // 8 keys per lane (32 KB), before the rank image uses the space
    uint16_t* img = reinterpret_cast<uint16_t*>(smem);
    __shared__ uint32_t s_wmax[CK_T / 64];
    __shared__ int s_maxrun;
    const int tid0 = threadIdx.x;
    uint32_t K[E];

    for (int it = blockIdx.x; it < n_list; it += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int a = row_list[it];
        const int srow = order[a];
        const uint16_t* __restrict__ rs = R_storage + (int64_t)srow * ldr;
        const uint16_t* __restrict__ bits = tie_bits + (int64_t)srow * ld_bits;
        const int e0 = E * tid;
        const bool live = e0 < P;
        if (tid == 0) s_maxrun = 0;
        // ---- run starts: prefix maximum of (bit ? 0 : e) over ascending e (as in k_rank_rows_tied)
        uint32_t run_local = 0;
#pragma unroll
        for (int g = 0; g < E / 16; g++) {
            const int eg = e0 + 16 * g;
            const uint32_t w = (live && eg < n) ? bits[eg / 16] : 0u;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int e = eg + q;
                const bool same = (w >> q) & 1u;
                if (!same && e < n) run_local = (uint32_t)e;
                K[16 * g + q] = same ? 0xffffffffu : (uint32_t)e;
            }
        }
        uint32_t incl = run_local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
            if ((tid & 63) >= d && o > incl) incl = o;
        }
        if ((tid & 63) == 63) s_wmax[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = (uint32_t)__shfl_up((int)incl, 1, 64);
        if ((tid & 63) == 0) before = 0;
        for (int w = 0; w < (tid >> 6); w++) before = s_wmax[w] > before ? s_wmax[w] : before;
        // ---- composite keys, and the longest run of the row
        uint32_t cur = before;
        int longest = 0;
#pragma unroll
        for (int q = 0; q < E; q++) {
            const int e = e0 + q;
            if (K[q] != 0xffffffffu) cur = K[q];
            uint32_t key = 0xffffffffu;
            if (live && e < n) {
                key = (cur << 16) | (uint32_t)inv[rs[n - 1 - e]];
                const int len = e - (int)cur + 1;
                longest = len > longest ? len : longest;
            }
            K[q] = key;
        }
        if (longest > CKS_MAXRUN) atomicMax(&s_maxrun, longest);
        __syncthreads();
        if (s_maxrun > CKS_MAXRUN) { __syncthreads(); continue; }      // a long run: the full network's row (uniform)
        // ---- round 1: aligned blocks of 16
        cks_sort_blocks16<E>(K);
        // ---- round 2: blocks shifted by 8 - the lane's elements 8 .. E-1 and the first 8 of the next lane
        uint32_t H[8];
#pragma unroll
        for (int q = 0; q < 8; q++) halo[q * CK_T + tid] = K[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; q++) H[q] = tid + 1 < CK_T ? halo[q * CK_T + tid + 1] : 0xffffffffu;
        __syncthreads();
        {
            uint32_t L[E];
#pragma unroll
            for (int q = 0; q < E - 8; q++) L[q] = K[q + 8];
#pragma unroll
            for (int q = 0; q < 8; q++) L[E - 8 + q] = H[q];
            cks_sort_blocks16<E>(L);
#pragma unroll
            for (int q = 0; q < E - 8; q++) K[q + 8] = L[q];
#pragma unroll
            for (int q = 0; q < 8; q++) H[q] = L[E - 8 + q];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) halo[q * CK_T + tid] = H[q];       // the next lane's first 8, sorted with my last 8
        __syncthreads();
        if (tid > 0) {
#pragma unroll
            for (int q = 0; q < 8; q++) K[q] = halo[q * CK_T + tid - 1];
        }
        __syncthreads();
        // ---- rank image and the row (as in k_rank_rows_tied)
        if (live) {
#pragma unroll
            for (int q = 0; q < E; q++) {
                const int e = e0 + q;
                if (e < n) img[K[q] & 0xffffu] = (uint16_t)(n - 1 - e);
            }
        }
        __syncthreads();
        uint4* __restrict__ o = reinterpret_cast<uint4*>(rank + (int64_t)a * ldr);
        const int groups = (n + 7) / 8;
        for (int g = tid; g < groups; g += CK_T) o[g] = reinterpret_cast<const uint4*>(img)[g];
        if (tid == 0) done[it] = 1;
        __syncthreads();
    }
}

template <int E>
static void launch_tied(const uint16_t* R_storage, const uint16_t* tie_bits, int64_t ld_bits, const int32_t* order,
                        const int32_t* inv, int n, int P, const int32_t* row_list, int n_list, uint16_t* rank, int64_t ldr,
                        int32_t* done, hipStream_t s)
{
    size_t lds = (size_t)CK_CHUNK * CK_T * sizeof(uint32_t);
    const size_t image = ((size_t)((n + 7) & ~7) * sizeof(uint16_t) + 15) & ~(size_t)15;
    if (image > lds) lds = image;
    static size_t have = 0;
    if (lds > have) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_rows_tied<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_rows_short_runs<E>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        have = lds;
    }
    const int grid = n_list < 1024 ? n_list : 1024;
    // rows whose runs are short are finished by the two-round block sort; the full network takes what is left
    static const bool full_only = getenv("HICMI_TIED_FULL") != nullptr;        // A/B: every row through the full network
    if (done && !full_only) {
        hipMemsetAsync(done, 0, sizeof(int32_t) * (size_t)n_list, s);
        hipLaunchKernelGGL(k_rank_rows_short_runs<E>, dim3(grid), dim3(CK_T), lds, s, R_storage, tie_bits, ld_bits, order, inv, n, P,
                           row_list, n_list, rank, ldr, done);
    }
    hipLaunchKernelGGL(k_rank_rows_tied<E>, dim3(grid), dim3(CK_T), lds, s, R_storage, tie_bits, ld_bits, order, inv, n, P,
                       row_list, n_list, rank, ldr, (done && !full_only) ? done : nullptr);
}

void launch_rank_rows_tied(const uint16_t* R_storage, const uint16_t* tie_bits, int64_t ld_bits, const int32_t* order,
                           const int32_t* inv, int n, const int32_t* row_list, int n_list, uint16_t* rank, int64_t ldr,
                           hipStream_t s, int32_t* done)
{
    if (n_list < 1) return;
    int P = sort_padded_size(n);
    if (P < 16) P = 16;
    if (P <= 16 * CK_T) launch_tied<16>(R_storage, tie_bits, ld_bits, order, inv, n, P, row_list, n_list, rank, ldr, done, s);
    else if (P <= 32 * CK_T) launch_tied<32>(R_storage, tie_bits, ld_bits, order, inv, n, P, row_list, n_list, rank, ldr, done, s);
    else launch_tied<64>(R_storage, tie_bits, ld_bits, order, inv, n, P, row_list, n_list, rank, ldr, done, s);
}

}  // namespace hicmi
