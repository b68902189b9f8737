// hicmi_internal.h - shared declarations between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <functional>

namespace hicmi {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a launch needs more than any earlier one
// (the call costs as much as a launch, and the search paths launch tens of thousands of kernels)
inline void ensure_dynamic_lds(const void* func, std::atomic<int>& have, size_t bytes)
{
    if ((int)bytes > have.load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        have.store((int)bytes, std::memory_order_relaxed);
    }
}

// ---- strictly left-to-right fp64 chains over LDS (the reference's Python sum() / running sums cannot be
// re-associated).  One lane runs them; the next 8 operands are fetched while the current 8 are added, so
// the chain waits on the adder only.
static constexpr int SERIAL_BATCH = 16;                 // operands in flight: 16 dependent fp64 adds cover an LDS round trip

__device__ __forceinline__ double serial_sum_lds(const double* t, int from, int to, double acc)
{
    constexpr int B = SERIAL_BATCH;
    int i = from;
    if (i + B <= to) {
        double a[B];
#pragma unroll
        for (int q = 0; q < B; q++) a[q] = t[i + q];
        while (i + 2 * B <= to) {
            double b[B];
#pragma unroll
            for (int q = 0; q < B; q++) b[q] = t[i + B + q];
#pragma unroll
            for (int q = 0; q < B; q++) acc += a[q];
#pragma unroll
            for (int q = 0; q < B; q++) a[q] = b[q];
            i += B;
        }
#pragma unroll
        for (int q = 0; q < B; q++) acc += a[q];
        i += B;
    }
    for (; i < to; i++) acc += t[i];
    return acc;
}

// t[i] <- acc + t[from] + ... + t[i] for i in [from, to), left to right
__device__ __forceinline__ double serial_prefix_lds(double* t, int from, int to, double acc)
{
    constexpr int B = SERIAL_BATCH;
    int i = from;
    if (i + B <= to) {
        double a[B];
#pragma unroll
        for (int q = 0; q < B; q++) a[q] = t[i + q];
        while (i + 2 * B <= to) {
            double b[B];
#pragma unroll
            for (int q = 0; q < B; q++) b[q] = t[i + B + q];
#pragma unroll
            for (int q = 0; q < B; q++) { acc += a[q]; t[i + q] = acc; }
#pragma unroll
            for (int q = 0; q < B; q++) a[q] = b[q];
            i += B;
        }
#pragma unroll
        for (int q = 0; q < B; q++) { acc += a[q]; t[i + q] = acc; }
        i += B;
    }
    for (; i < to; i++) { acc += t[i]; t[i] = acc; }
    return acc;
}

// ---- NumPy's pairwise summation tree (add.reduce of one contiguous or strided run of <= 8192 elements) -----------
static constexpr int MAX_LEAVES = 64;                  // an 8192-element chunk splits into <= 64 leaves

// Leaves of NumPy's pairwise recursion over [off, off+len), in order, with their depth in the
// recursion tree.  The explicit stack lives in LDS (st_off/st_len/st_dep: 16 entries each): private
// arrays indexed at run time would be spilled to scratch memory, a global-memory round trip per access.
__device__ __forceinline__ int enumerate_leaves(int off, int len, int* leaf_off, int* leaf_len, int* leaf_dep,
                                                int* st_off, int* st_len, int* st_dep)
{
    int sp = 1, n = 0;
    st_off[0] = off; st_len[0] = len; st_dep[0] = 0;
    while (sp > 0) {
        sp--;
        const int o = st_off[sp], l = st_len[sp], d = st_dep[sp];
        if (l <= 128) { leaf_off[n] = o; leaf_len[n] = l; leaf_dep[n] = d; n++; continue; }
        int n2 = l / 2;
        n2 -= n2 % 8;
        st_off[sp] = o + n2; st_len[sp] = l - n2; st_dep[sp] = d + 1; sp++;      // right half (processed second)
        st_off[sp] = o; st_len[sp] = n2; st_dep[sp] = d + 1; sp++;                // left half
    }
    return n;
}

// left + right at every split of the same tree: leaves arrive in order; whenever the two newest partial
// results sit at the same depth they are the two halves of one node (left first) and are replaced by
// their sum one level up.  val/dep: LDS stacks of 16 entries.
__device__ __forceinline__ double combine_leaves(int n_leaves, const double* leaf_sum, const int* leaf_dep, double* val,
                                                 int* dep)
{
    int sp = 0;
    for (int l = 0; l < n_leaves; l++) {
        double v = leaf_sum[l];
        int d = leaf_dep[l];
        while (sp > 0 && dep[sp - 1] == d) { v = val[sp - 1] + v; d--; sp--; }
        val[sp] = v; dep[sp] = d; sp++;
    }
    return val[0];
}

// ---- launchers (defined next to their kernels) --------------------------------------------------
// k_part1.hip
void launch_row_sums(const double* C, int64_t ldc, int n, double* np_sum, double* seq_sum, int row_first, int row_stride,
                     hipStream_t s);
void launch_compact(const double* src, int64_t ld_src, const int32_t* keep, int n_keep, double* dst, int64_t ld_dst,
                    hipStream_t s);
void launch_widen_f32(const float* src, double* dst, int64_t cells, hipStream_t s);   // src = (float*)dst + cells
void launch_build_w(const double* C, int64_t ldc, const double* np_sum, int n, double* W, int64_t ldw, hipStream_t s);
// k_nnchain.hip
size_t nnchain_workspace_bytes(int n);
int launch_nnchain(double* W, double* W2, int64_t ldw, int n, int* chain, double* zraw, void* workspace, bool profile,
                   int dcap, bool compact, int fallback, hipStream_t s, const std::function<void()>& after_first_rowmin = nullptr);
                                              // returns the number of epoch launches; fallback: 0, 1 (spread out), 2 (one workgroup);
                                              // after_first_rowmin: called once, when the first epoch's cache pass is queued
void launch_selftest_division(unsigned long long seed, int blocks, int iters, unsigned long long* d_mismatches, hipStream_t s);
const int* nnchain_state_ptr(void* workspace);                    // 16 ints: [0] merges done ... [5] stop code (0 = none,
                                                                  // 1 = guard / NaN, 2 = a peer workgroup answered late, 3 = replicas disagree)
const unsigned long long* nnchain_prof_ptr(void* workspace);      // 8 counters, contiguous after the state: [0..4] phase totals
                                                                  // (100 MHz ticks), [5] columns visited by scans, [6] scans, [7] cache hits
void launch_cut_count(const uint16_t* rank, int64_t ldr, int row0, int nrows, int lo, int mode, int cparam,
                      int32_t* x_out, int row_step, hipStream_t s);
void launch_hyper_flags(const int32_t* x, int nrows, int mode, int L_fixed, int64_t M, double psig, uint8_t* sig,
                        int own_first, int own_step, hipStream_t s);

// k_part1_scan.hip: the cut-scan loops with their control flow on the device.  One record in device memory carries the
// arguments of the next scan and the loop state; the host reads it back once per batch of scans.
struct ScanState {
    int done;              // the loop has ended: later launches return at once
    int mode;              // 0 first pass (S2C:413-551), 1 filter (S2C:553-727)
    int start, cut, n_rows, recount;
    long long M;
    int min_size, stop_ind, loop_count, n_cuts, n_log, scans;          // first pass
    int MD, n_alt, alt_off, f_i, f_keep_from, f_noise, f_round, f_max_rounds, f_warned;   // filter
    int pad;
    unsigned long long bytes;                                           // rank bytes the scans' queries cover
};
void launch_first_pass_pairs(const uint16_t* rank, int64_t ldr, int n, ScanState* st, int32_t* x, uint8_t* sig, double psig,
                             int32_t* cuts, int32_t* mlog, int log_cap, int pairs, hipStream_t s);
void launch_filter_pairs(const uint16_t* rank, int64_t ldr, int n, int max_rows, ScanState* st, int32_t* x, uint8_t* sig,
                         double psig, int32_t* alt, uint8_t* filt, uint8_t* prev, int32_t* seg, int32_t* seg_x, int pairs,
                         hipStream_t s);

// The value lane ^ M holds, M a power of two below 64, without the LDS crossbar (ds_bpermute, which __shfl_xor compiles
// to, issues at a fraction of the VALU rate and was what the bitonic networks' in-wave stages waited for): DPP quad
// permutes and row mirrors inside a row of 16 lanes, gfx950's v_permlane16_swap / v_permlane32_swap across rows.
#if defined(__HIPCC__)
template <int M>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, int lane)
{
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);       // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        const int t = __builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, true);                            // row_half_mirror: ^ 7
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x1B, 0xf, 0xf, true);                               // quad_perm [3,2,1,0]: ^ 3
    } else if constexpr (M == 8) {
        const int t = __builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, true);                            // row_mirror: ^ 15
        return (uint32_t)__builtin_amdgcn_mov_dpp(t, 0x141, 0xf, 0xf, true);                              // ^ 7
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);     // odd rows of one copy <-> even rows of the other
        return (lane & 16) ? r[0] : r[1];
    } else {
        static_assert(M == 32, "xor_lane: M must be 1, 2, 4, 8, 16 or 32");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // upper half of one copy <-> lower half of the other
        return (lane & 32) ? r[0] : r[1];
    }
}
#endif

// k_sort.hip
int  sort_padded_size(int n);                 // power of two >= n
int  sort_workgroups(int n);                  // persistent workgroups the sort kernel wants
size_t sort_scratch_bytes(int n);             // device scratch (keys + indices) for all workgroups
struct SortExtras {                        // optional modes of launch_sort_rows (register-blocked bitonic kernel only)
    const int32_t* row_list = nullptr; int n_list = 0;        // sort exactly these rows (instead of the row shard)
    uint8_t* tie_flag = nullptr; unsigned* tie_count = nullptr; unsigned tie_limit = 0;   // flag rows holding equal keys
    uint16_t* tie_bits = nullptr; int64_t ld_bits = 0;        // ... and which sorted elements repeat the key before them
    int max_workgroups = 0;                                    // cap on the grid (0: the default)
    int avoid_xcc = -1; unsigned* row_counter = nullptr;       // leave this XCD to another kernel; rows dealt out by a (zeroed) counter
};
// s_getreg operand of HW_REG_XCC_ID (id 20), bits 3:0: which of the 8 XCDs a wave runs on
static constexpr int GETREG_XCC_ID = 20 | (0 << 6) | (3 << 11);
// The XCD the nn-chain's one-wave kernel claims for itself (k_nnchain.hip), or -1 when it runs spread over all of them
int nnchain_local_xcc(int n);
void launch_sort_rows(const double* C, int64_t ldc, const int32_t* order, const int32_t* inv, const double* np_sum,
                      const double* seq_sum, int n, void* scratch, uint16_t* R, int64_t ldr, int row_first, int row_stride,
                      hipStream_t s, const SortExtras& x = SortExtras());
// Rank rows of rows that hold equal keys, from the storage-label sort + its tie bits (k_sort_tied.hip).
void launch_rank_rows_tied(const uint16_t* R_storage, const uint16_t* tie_bits, int64_t ld_bits, const int32_t* order,
                           const int32_t* inv, int n, const int32_t* row_list, int n_list, uint16_t* rank, int64_t ldr,
                           hipStream_t s, int32_t* done = nullptr);     // done: n_list ints of scratch (rows finished by the short-run kernel)
// rank[a][b] = rank_storage[order[a]][order[b]] (rows without equal keys; see k_sort.hip)
void launch_rank_relabel(const uint16_t* rank_storage, uint16_t* rank, int64_t ldr, int n, const int32_t* order, int row_first,
                         int row_stride, hipStream_t s);
size_t sort_radix_scratch_bytes(int n);
void launch_rank_rows_radix(const double* C, int64_t ldc, const int32_t* order, const int32_t* inv, const double* np_sum,
                            const double* seq_sum, int n, void* scratch, uint16_t* rank, int64_t ldr, int row_first,
                            int row_stride, hipStream_t s);   // LSD radix: writes the rank rows directly
void launch_rank_invert(const uint16_t* R, uint16_t* rank, int64_t ldr, int n, int row_first, int row_stride, hipStream_t s,
                        const int32_t* row_list = nullptr, int n_list = 0);
void launch_similarity_row(const double* C, int64_t ldc, const int32_t* order, const double* np_sum,
                           const double* seq_sum, int n, int row, double* out, hipStream_t s);

// k_part2.hip
void launch_p2_select(const double* C, int64_t ldc, const int32_t* sel, int n, double* M2, int64_t ld2, hipStream_t s);
void launch_p2_total(const double* M2, int64_t ld2, int n, double* T, double* total, hipStream_t s);
void launch_p2_score_exact(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, double total,
                           double* T, double* work, double* scores, hipStream_t s);
void launch_p2_score(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, const double* H,
                     double inv_total_unused, double total, double* scores, hipStream_t s);

void launch_p2_total_perm(const double* M2, int64_t ld2, const int32_t* d_perm, int n, double* T, double* total,
                          hipStream_t s);

// k_part2_search.hip
struct WindowDesc {            // the k <= 8 scaffolds of a window, passed to the kernel by value
    int32_t start[8];          // selection range start of window scaffold j
    int32_t len[8];
    int32_t off[8];            // offset of scaffold j inside the window in the CURRENT arrangement
    uint8_t rev[8];            // current orientation of scaffold j
};
void launch_arr_materialize(const int32_t* packed, int S, const int32_t* scaf_start, const int32_t* scaf_len, int n_arr,
                            int32_t* pos2sel, hipStream_t s);
void launch_p2_base_partial(const double* M2, int64_t ld2, const int32_t* pos2sel, int n_arr, const double* H, int n_tot,
                            int n_blocks, double* out, hipStream_t s);
void launch_p2_insert_delta(const double* M2, int64_t ld2, const int32_t* pos2sel, int n_arr, const int32_t* arr_pos,
                            int S, int new_start, int L, const double* H, int n_base_blocks, double* out, hipStream_t s);
struct WindowBatchEntry {      // one window of a batch (device array)
    WindowDesc w;
    int32_t p0, m;             // first position and number of bins of the window
    int64_t g_off;             // offset of its m x m table in the batch's G buffer
};
void launch_p2_window_batch(const double* M2, int64_t ld2, const int32_t* pos2sel, int n, int k,
                            const WindowBatchEntry* wb, int n_win, int max_m, const int8_t* orders, const uint8_t* orients,
                            int n_ord, int n_ori, const double* H, double* G_all, double* delta_all, hipStream_t s);


// k_part2_window.hip: the same scores from placement tables (one pass over the matrix per table, a few look-ups
// per candidate); tables = n_win * window_table_doubles(k) doubles of scratch
int64_t window_table_doubles(int k);
void launch_p2_window_tables(const double* M2, int64_t ld2, const int32_t* pos2sel, int n, int k,
                             const WindowBatchEntry* wb, const WindowBatchEntry* h_wb, int n_win, int max_m, const int8_t* orders, const uint8_t* orients,
                             int n_ord, int n_ori, const double* H, double* tables, double* delta_all, hipStream_t s);

// Lock-step insertion (k_part2_insert.hip): orderRemainderScaffolds for several chromosomes at once, every
// decision taken on the device.  One InsStep per (step, chromosome), built by the host in advance.
static constexpr int INS_MAXC = 8;       // candidates re-scored literally per step; more -> the host decides that step
struct InsState {
    int32_t fail;                        // -1, or the first step the device could not decide
    int32_t n_short;                     // short-listed candidates of the current step
    int32_t direct, pad;                 // 1: the step has ONE candidate near the top - slot 0 is taken without a literal score
    int32_t gap[INS_MAXC], rev[INS_MAXC], idx[INS_MAXC];   // idx = position in the reference's enumeration
    double total;                        // literal total of the step (OG:343), formed when a literal pass runs
    double lit[INS_MAXC];                // literal scores of the short list
};
struct InsLog { int32_t gap, rev; double best; int32_t n_short, pad; };
struct InsStep {
    const double* M2; const double* H; int64_t ld2;
    const int32_t* pos_cur; int32_t* pos_nxt;            // arrangement as bin order (ping-pong)
    const int32_t* packed_cur; int32_t* packed_nxt;       // [S ids][S+1 prefix positions][S reversed flags]
    double *T_total, *T_cand, *work, *partial;
    InsState* st; InsLog* log;                            // log: this step's entry
    int32_t n_arr, S, L, new_start, new_id, active, step, last;   // last: the job's final step (its score is returned)
};
// k_part2.hip / k_part2_search.hip / k_part2_insert.hip: one launch serves all chromosomes (blockIdx.y)
void launch_insb_reset(const InsStep* steps, int n_chrom, hipStream_t s);
void launch_insb_fast(const InsStep* steps, int n_chrom, int max_S, int max_n_arr, int n_base_blocks, hipStream_t s);
void launch_insb_shortlist(const InsStep* steps, int n_chrom, int max_S, int max_n_arr, int n_base_blocks, double near_top,
                           int max_c, hipStream_t s);
void launch_insb_diag_cand(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s);
void launch_insb_cost(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s);
void launch_insb_apply(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s);

// k_plot.hip
void launch_plot_select(const double* C, int64_t ldc, const double* np_sum, const double* seq_sum, int kind,
                        const int32_t* order, int n_sel, int n_targets, struct SelectState* d_state, unsigned int* d_hist,
                        hipStream_t s);
size_t plot_select_state_bytes();
size_t plot_select_hist_bytes();
int plot_select_max_targets();
void plot_select_fill(void* host_state, const unsigned long long* ranks, int n_targets);
double plot_select_value(const void* host_state, int t);
void launch_plot_downsample(const double* C, int64_t ldc, const double* np_sum, const double* seq_sum, int kind,
                            const int32_t* order, int n_sel, int px, double* out, hipStream_t s);

}  // namespace hicmi
