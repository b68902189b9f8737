// k_part2_insert.hip - orderRemainderScaffolds (orderGenome.py:475-493 -> checkAllScores :332-372) with
// the decision of every step taken ON THE DEVICE, so that a whole chromosome's insertion phase is one
// queue of kernels and one host synchronisation.
//
// Why it can be queued ahead: the scaffold added at step t, the number of scaffolds S0 + t and the number
// of bins of the arrangement are known before any score is - only the gap and the orientation chosen
// are not, and those stay in device memory:
//   k_p2_diag_sums (tail form) + k_ins_shortlist : literal total of "arrangement, then the new scaffold"
//                            (OG:343); fast score of the 2(S+1) candidates in the reference's enumeration
//                            order from the BASE / STRADDLE / CROSS terms (k_part2_search.hip); the
//                            candidates within 1e-9 of the best become the short list
//   k_ins_expand           : their bin orders
//   k_p2_diag_sums + k_p2_cost_exact : literal scores (NumPy's summation order) of the short list
//   k_ins_apply            : first strict maximum above 0. in enumeration order (OG:349,359) - or gap 0,
//                            '+' when nothing scored above 0. (OG:341,367) - written to the log and applied
//                            to the arrangement (ping-pong buffers).
// A step whose short list is longer than INS_MAXC (ties: e.g. a scaffold without contacts scores the
// same everywhere) sets InsState::fail; every later kernel then returns at once and the host decides
// that step through hicmi_p2_decide_insertion before queueing the rest.
#include "hicmi_internal.h"

namespace hicmi {

__global__ void k_ins_reset(InsState* st)
{
    st->fail = -1;
    st->n_short = 0;
}

void launch_ins_reset(InsState* st, hipStream_t s) { hipLaunchKernelGGL(k_ins_reset, dim3(1), dim3(1), 0, s, st); }

static constexpr int SL_THREADS = 256;
static constexpr int SL_STAGE_MAX = 8192;              // doubles staged in LDS (64 KB)
static constexpr int SL_LIST = 64;

// partial: [n_base_blocks BASE slabs][S STRADDLE increments][2(S+1) CROSS terms], as written by
// launch_p2_insert_delta.  The arithmetic (and its order) is that of hicmi_p2_score_insertions +
// hicmi_p2_decide_insertion on the host, so both paths short-list the same candidates.
__global__ __launch_bounds__(SL_THREADS) void k_ins_shortlist(const double* __restrict__ T, int n_used,
                                                              const double* __restrict__ partial, int n_base_blocks,
                                                              int S, int step, double near_top, int max_c, InsState* st)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sl[];
    double* buf = reinterpret_cast<double*>(smem_sl);    // T staged for the serial total, then the STRADDLE prefix
    __shared__ double s_total, s_base, s_wmax[SL_THREADS / 64];
    __shared__ int s_any[SL_THREADS / 64], s_cnt, s_list[SL_LIST];
    const int tid = threadIdx.x;
    if (st->fail >= 0) return;
    const bool staged = n_used <= SL_STAGE_MAX;
    const int n_buf = (staged ? n_used : 0) > S + 1 ? (staged ? n_used : 0) : S + 1;
    double* part = buf + n_buf;                          // BASE slabs and STRADDLE increments, staged
    if (staged) for (int i = tid; i < n_used; i += SL_THREADS) buf[i] = T[i];
    for (int i = tid; i < n_base_blocks + S; i += SL_THREADS) part[i] = partial[i];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (tid == 0) {
        double acc = 0.0;                                // Python sum(): 0 + T_1 + T_2 + ...   (OG:343)
        if (staged) acc = serial_sum_lds(buf, 1, n_used, 0.0);
        else for (int i = 1; i < n_used; i++) acc += T[i];
        s_total = acc;
        s_base = serial_sum_lds(part, 0, n_base_blocks, 0.0);
        buf[0] = 0.0;                                    // prefix[g] = increments 0..g-1, left to right
        for (int g = 0; g < S; g++) buf[g + 1] = part[n_base_blocks + g];
        serial_prefix_lds(buf, 1, S + 1, 0.0);
    }
    __syncthreads();
    const double total = s_total, base = s_base;
    const double* __restrict__ cross = partial + n_base_blocks + S;
    const int n_cand = 2 * (S + 1);
    // candidate i: gap i/2; the scaffold arrives '+', is tested as it is and then flipped, and stays flipped
    // for the next gap (OG:344-365), so gap g tests orientation g&1 first
    auto fast_of = [&](int i) -> double {
        const int g = i >> 1, first = g & 1;
        const int r = (i & 1) ? (first ^ 1) : first;
        return (base - buf[g] + cross[2 * g + r]) / total;
    };
    double mx = -__builtin_inf();
    int any = 0;
    for (int i = tid; i < n_cand; i += SL_THREADS) {
        const double v = fast_of(i);
        if (isfinite(v)) { any = 1; if (v > mx) mx = v; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(any, off, 64);
        if (o > mx) mx = o;
        any |= oa;
    }
    if ((tid & 63) == 0) { s_wmax[tid >> 6] = mx; s_any[tid >> 6] = any; }
    __syncthreads();
    double top = 0.0;                                    // the floor of checkAllScores is 0. (OG:341)
    int have = 0;
    for (int w = 0; w < SL_THREADS / 64; w++) { if (s_wmax[w] > top) top = s_wmax[w]; have |= s_any[w]; }
    if (!have) {
        if (tid == 0) { st->n_short = 0; st->total = total; }
        return;
    }
    const double thr = top - fabs(top) * near_top;
    for (int i = tid; i < n_cand; i += SL_THREADS) {
        const double v = fast_of(i);
        if (isfinite(v) && v >= thr) {
            const int slot = atomicAdd(&s_cnt, 1);
            if (slot < SL_LIST) s_list[slot] = i;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int cnt = s_cnt;
        if (cnt > max_c) { st->fail = step; st->n_short = 0; return; }
        for (int a = 1; a < cnt; a++) {                  // enumeration order
            const int v = s_list[a];
            int b = a - 1;
            while (b >= 0 && s_list[b] > v) { s_list[b + 1] = s_list[b]; b--; }
            s_list[b + 1] = v;
        }
        for (int q = 0; q < cnt; q++) {
            const int i = s_list[q], g = i >> 1, first = g & 1;
            st->idx[q] = i; st->gap[q] = g; st->rev[q] = (i & 1) ? (first ^ 1) : first;
        }
        st->n_short = cnt;
        st->total = total;
    }
}

static std::atomic<int> g_lds_shortlist{0};

void launch_ins_shortlist(const double* T, int n_used, const double* partial, int n_base_blocks, int S, int step,
                          double near_top, int max_c, InsState* st, hipStream_t s)
{
    const int need = ((n_used <= SL_STAGE_MAX ? n_used : 0) > S + 1 ? (n_used <= SL_STAGE_MAX ? n_used : 0) : S + 1)
                     + n_base_blocks + S;
    const size_t lds = (((size_t)need * sizeof(double)) + 15) & ~(size_t)15;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_ins_shortlist), g_lds_shortlist, lds);
    hipLaunchKernelGGL(k_ins_shortlist, dim3(1), dim3(SL_THREADS), lds, s, T, n_used, partial, n_base_blocks, S, step,
                       near_top, max_c < 1 ? 1 : (max_c > INS_MAXC ? INS_MAXC : max_c), st);
}

// bin order of short-listed candidate q: the arrangement with the new scaffold (selection range
// [new_start, new_start + L)) laid down at its gap, reversed or not
__global__ __launch_bounds__(256) void k_ins_expand(const int32_t* __restrict__ pos2sel, int n_arr,
                                                    const int32_t* __restrict__ arr_pos, int new_start, int L,
                                                    const InsState* __restrict__ st, int32_t* __restrict__ perms)
{
    const int q = blockIdx.y;
    if (st->fail >= 0 || q >= st->n_short) return;
    const int pos = blockIdx.x * 256 + threadIdx.x, n_new = n_arr + L;
    if (pos >= n_new) return;
    const int P = arr_pos[st->gap[q]], r = st->rev[q];
    int v;
    if (pos < P) v = pos2sel[pos];
    else if (pos < P + L) v = new_start + (r ? L - 1 - (pos - P) : pos - P);
    else v = pos2sel[pos - L];
    perms[(int64_t)q * n_new + pos] = v;
}

void launch_ins_expand(const int32_t* pos2sel, int n_arr, const int32_t* arr_pos, int new_start, int L, const InsState* st,
                       int32_t* perms, hipStream_t s)
{
    hipLaunchKernelGGL(k_ins_expand, dim3((n_arr + L + 255) / 256, INS_MAXC), dim3(256), 0, s, pos2sel, n_arr, arr_pos,
                       new_start, L, st, perms);
}

// decision + application.  packed_*: [S ids][S+1 prefix positions][S reversed flags] (S+1 scaffolds on output)
__global__ __launch_bounds__(256) void k_ins_apply(const int32_t* __restrict__ pos2sel_in, int n_arr,
                                                   const int32_t* __restrict__ packed_in, int S, int new_id, int new_start,
                                                   int L, const InsState* __restrict__ st,
                                                   int32_t* __restrict__ packed_out, int32_t* __restrict__ pos2sel_out,
                                                   InsLog* __restrict__ log_entry)
{
    if (st->fail >= 0) return;
    double best = 0.0;
    int pick = -1;
    const int ns = st->n_short;
    for (int q = 0; q < ns; q++) if (st->lit[q] > best) { best = st->lit[q]; pick = q; }   // first strict maximum
    const int gap = pick >= 0 ? st->gap[pick] : 0, rev = pick >= 0 ? st->rev[pick] : 0;
    const int32_t* __restrict__ id_in = packed_in;
    const int32_t* __restrict__ pos_in = packed_in + S;
    const int32_t* __restrict__ rev_in = packed_in + 2 * S + 1;
    const int P = pos_in[gap], n_new = n_arr + L;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos < n_new) {
        int v;
        if (pos < P) v = pos2sel_in[pos];
        else if (pos < P + L) v = new_start + (rev ? L - 1 - (pos - P) : pos - P);
        else v = pos2sel_in[pos - L];
        pos2sel_out[pos] = v;
    }
    if (blockIdx.x == 0) {
        const int S1 = S + 1;
        int32_t* __restrict__ id_out = packed_out;
        int32_t* __restrict__ pos_out = packed_out + S1;
        int32_t* __restrict__ rev_out = packed_out + 2 * S1 + 1;
        for (int j = threadIdx.x; j < S1; j += 256) {
            id_out[j] = j < gap ? id_in[j] : (j == gap ? new_id : id_in[j - 1]);
            rev_out[j] = j < gap ? rev_in[j] : (j == gap ? rev : rev_in[j - 1]);
        }
        for (int j = threadIdx.x; j <= S1; j += 256) pos_out[j] = j <= gap ? pos_in[j] : pos_in[j - 1] + L;
        if (threadIdx.x == 0) { log_entry->gap = gap; log_entry->rev = rev; log_entry->best = pick >= 0 ? best : 0.0; }
    }
}

void launch_ins_apply(const int32_t* pos2sel_in, int n_arr, const int32_t* packed_in, int S, int new_id, int new_start,
                      int L, const InsState* st, int32_t* packed_out, int32_t* pos2sel_out, InsLog* log_entry, hipStream_t s)
{
    hipLaunchKernelGGL(k_ins_apply, dim3((n_arr + L + 255) / 256), dim3(256), 0, s, pos2sel_in, n_arr, packed_in, S, new_id,
                       new_start, L, st, packed_out, pos2sel_out, log_entry);
}

}  // namespace hicmi
