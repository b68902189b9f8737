// hicmi_internal.h - shared declarations between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hicmi {

// ---- launchers (defined next to their kernels) --------------------------------------------------
// k_part1.hip
void launch_row_sums(const double* C, int64_t ldc, int n, double* np_sum, double* seq_sum, hipStream_t s);
void launch_compact(const double* src, int64_t ld_src, const int32_t* keep, int n_keep, double* dst, int64_t ld_dst,
                    hipStream_t s);
void launch_build_w(const double* C, int64_t ldc, const double* np_sum, int n, double* W, int64_t ldw, hipStream_t s);
void launch_nnchain(double* W, int64_t ldw, int n, int* size, int* chain, double* zraw, int* status, hipStream_t s);
void launch_cut_count(const uint16_t* rank, int64_t ldr, int row0, int nrows, int lo, int mode, int cparam,
                      int32_t* x_out, hipStream_t s);
void launch_hyper_flags(const int32_t* x, int nrows, int mode, int L_fixed, int64_t M, double psig, uint8_t* sig,
                        hipStream_t s);

// k_sort.hip
int  sort_padded_size(int n);                 // power of two >= n
int  sort_workgroups(int n);                  // persistent workgroups the sort kernel wants
size_t sort_scratch_bytes(int n);             // device scratch (keys + indices) for all workgroups
void launch_sort_rows(const double* C, int64_t ldc, const int32_t* order, const double* np_sum, const double* seq_sum,
                      int n, void* scratch, uint16_t* R, int64_t ldr, hipStream_t s);
void launch_rank_invert(const uint16_t* R, uint16_t* rank, int64_t ldr, int n, hipStream_t s);
void launch_similarity_row(const double* C, int64_t ldc, const int32_t* order, const double* np_sum,
                           const double* seq_sum, int n, int row, double* out, hipStream_t s);

// k_part2.hip
void launch_p2_select(const double* C, int64_t ldc, const int32_t* sel, int n, double* M2, int64_t ld2, hipStream_t s);
void launch_p2_total(const double* M2, int64_t ld2, int n, double* T, double* total, hipStream_t s);
void launch_p2_score_exact(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, double total,
                           double* T, double* scores, hipStream_t s);
void launch_p2_score(const double* M2, int64_t ld2, const int32_t* perms, int n_cand, int n_used, const double* H,
                     double inv_total_unused, double total, double* scores, hipStream_t s);

}  // namespace hicmi
