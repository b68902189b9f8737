// k_part2_insert.hip - orderRemainderScaffolds (orderGenome.py:475-493 -> checkAllScores :332-372) with
// the decision of every step taken ON THE DEVICE and the chromosomes of a genome advanced in LOCK STEP:
// the insertion phase of all chromosomes is one queue of kernels and one host synchronisation.
//
// Why it can be queued ahead: the scaffold added at step t, the number of scaffolds S0 + t and the number
// of bins of the arrangement are known before any score is - only the gap and the orientation chosen
// are not, and those stay in device memory.  The host builds one InsStep record per (step, chromosome)
// (hicmi_internal.h) and every kernel takes the records of one step; blockIdx.y = chromosome:
//   k_insb_base/_fast (k_part2_search.hip)  BASE / STRADDLE / CROSS terms of the 2(S+1) fast scores
//   k_insb_shortlist      fast scores in the reference's enumeration order; the candidates within 1e-9 of the
//                         best become the short list.  ONE candidate (the usual case) is taken as it is: the
//                         literal pass could not pick another, and only a job's final score is ever read
//   k_insb_diag_cand + k_insb_cost (k_part2.hip)  the literal pass, for the chromosomes that still need it:
//                         diagonal sums of "arrangement, then the new scaffold" (OG:343), their serial Python
//                         sum (the step's total) and the literal scores (NumPy's summation order) of the short
//                         list; the candidates' bin orders are never materialised (IndexMap)
//   k_insb_apply          first strict maximum above 0. in enumeration order (OG:349,359) - or gap 0, '+' when
//                         nothing scored above 0. (OG:341,367) - written to the log and applied to the
//                         arrangement (ping-pong buffers).
// A step whose short list is longer than the cap (ties: e.g. a scaffold without contacts scores the same
// at both ends) sets that chromosome's InsState::fail; its later kernels return at once and the host
// decides the step through hicmi_p2_decide_insertion before queueing the rest.
#include "hicmi_internal.h"

namespace hicmi {

__global__ void k_insb_reset(const InsStep* __restrict__ steps)
{
    InsState* st = steps[blockIdx.x].st;
    if (threadIdx.x == 0 && steps[blockIdx.x].active) { st->fail = -1; st->n_short = 0; st->direct = 0; }
}

void launch_insb_reset(const InsStep* steps, int n_chrom, hipStream_t s)
{
    hipLaunchKernelGGL(k_insb_reset, dim3(n_chrom), dim3(64), 0, s, steps);
}

static constexpr int SL_THREADS = 256;
static constexpr int SL_LIST = 64;
static constexpr int SL_STAGE_MAX = 8192;              // STRADDLE row values staged in LDS (64 KB); longer: read from L2

// The fast scores are those of hicmi_p2_score_insertions without the division by the step's literal total - a
// common positive factor, irrelevant to a ranking with a relative band - so the total (a full pass over the
// sub-matrix plus a serial sum) is only formed for the steps that go on to the literal pass.
__global__ __launch_bounds__(SL_THREADS) void k_insb_shortlist(const InsStep* __restrict__ steps, int n_base_blocks,
                                                               double near_top, int max_c)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sl[];
    double* buf = reinterpret_cast<double*>(smem_sl);    // the STRADDLE prefix
    __shared__ double s_base, s_wmax[SL_THREADS / 64];
    __shared__ int s_any[SL_THREADS / 64], s_cnt, s_list[SL_LIST];
    const InsStep& d = steps[blockIdx.x];
    InsState* st = d.st;
    const int tid = threadIdx.x;
    if (!d.active || st->fail >= 0) return;
    const int S = d.S, n_arr = d.n_arr;
    const double* __restrict__ partial = d.partial;
    const int32_t* __restrict__ arr_pos = d.packed_cur + S;
    double* part = buf + S + 1;                          // BASE slabs, then the STRADDLE row values s(u), staged
    const bool staged = n_arr <= SL_STAGE_MAX;
    const int n_stage = n_base_blocks + (staged ? n_arr : 0);
    for (int i = tid; i < n_stage; i += SL_THREADS) part[i] = partial[i];
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // STRADDLE(g+1) - STRADDLE(g): the rows of scaffold g, added in position order (one lane per scaffold)
    for (int g = tid; g < S; g += SL_THREADS) {
        const int P0 = arr_pos[g], P1 = arr_pos[g + 1];
        double acc = 0.0;
        if (staged) for (int u = P0; u < P1; u++) acc += part[n_base_blocks + u];
        else for (int u = P0; u < P1; u++) acc += partial[n_base_blocks + u];
        buf[g + 1] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        s_base = serial_sum_lds(part, 0, n_base_blocks, 0.0);
        buf[0] = 0.0;                                    // prefix[g] = increments 0..g-1, left to right
        serial_prefix_lds(buf, 1, S + 1, 0.0);
    }
    __syncthreads();
    const double base = s_base;
    const double* __restrict__ cross = partial + n_base_blocks + n_arr;
    const int n_cand = 2 * (S + 1);
    // candidate i: gap i/2; the scaffold arrives '+', is tested as it is and then flipped, and stays flipped
    // for the next gap (OG:344-365), so gap g tests orientation g&1 first
    auto fast_of = [&](int i) -> double {
        const int g = i >> 1, first = g & 1;
        const int r = (i & 1) ? (first ^ 1) : first;
        return base - buf[g] + cross[2 * g + r];
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
        if (tid == 0) { st->n_short = 0; st->direct = 0; }
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
        int cnt = s_cnt;
        st->direct = 0;
        if (cnt > max_c && !(d.L == 1 && cnt <= 2 * max_c && cnt <= SL_LIST)) { st->fail = d.step; st->n_short = 0; return; }
        for (int a = 1; a < cnt; a++) {                  // enumeration order
            const int v = s_list[a];
            int b = a - 1;
            while (b >= 0 && s_list[b] > v) { s_list[b + 1] = s_list[b]; b--; }
            s_list[b + 1] = v;
        }
        if (d.L == 1) {
            // a one-bin scaffold reads the same in both orientations: the flipped twin of a listed candidate is the
            // same bin order, scores the same literal value and comes second, so it can never be a strict maximum
            int w = 0;
            for (int q = 0; q < cnt; q++) {
                const int i = s_list[q];
                if ((i & 1) && w > 0 && s_list[w - 1] == i - 1) continue;
                s_list[w++] = i;
            }
            cnt = w;
            if (cnt > max_c) { st->fail = d.step; st->n_short = 0; return; }
        }
        for (int q = 0; q < cnt; q++) {
            const int i = s_list[q], g = i >> 1, first = g & 1;
            st->idx[q] = i; st->gap[q] = g; st->rev[q] = (i & 1) ? (first ^ 1) : first;
        }
        // One candidate within near_top of a positive best score: no other can overtake it in the literal
        // arithmetic (the two agree to ~1e-13) and its literal score is positive like its fast one, so the
        // literal pass is skipped - except at the job's final step, whose score is handed back (OG:493).
        if (cnt == 1 && !d.last && top > 0.0) { st->direct = 1; st->n_short = 0; }
        else st->n_short = cnt;
    }
}

static std::atomic<int> g_lds_shortlist{0};

void launch_insb_shortlist(const InsStep* steps, int n_chrom, int max_S, int max_n_arr, int n_base_blocks, double near_top,
                           int max_c, hipStream_t s)
{
    const int need = (max_S + 1) + n_base_blocks + (max_n_arr < SL_STAGE_MAX ? max_n_arr : SL_STAGE_MAX);
    const size_t lds = (((size_t)need * sizeof(double)) + 15) & ~(size_t)15;
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_insb_shortlist), g_lds_shortlist, lds);
    hipLaunchKernelGGL(k_insb_shortlist, dim3(n_chrom), dim3(SL_THREADS), lds, s, steps, n_base_blocks, near_top,
                       max_c < 1 ? 1 : (max_c > INS_MAXC ? INS_MAXC : max_c));
}

// decision + application.  packed_*: [S ids][S+1 prefix positions][S reversed flags] (S+1 scaffolds on output)
__global__ __launch_bounds__(256) void k_insb_apply(const InsStep* __restrict__ steps)
{
    const InsStep& d = steps[blockIdx.y];
    const InsState* __restrict__ st = d.st;
    if (!d.active || st->fail >= 0) return;
    const int n_arr = d.n_arr, S = d.S, L = d.L, n_new = n_arr + L;
    if ((int)blockIdx.x * 256 >= n_new && blockIdx.x != 0) return;
    double best = 0.0;
    int pick = -1;
    const int ns = st->n_short, direct = st->direct;
    for (int q = 0; q < ns; q++) if (st->lit[q] > best) { best = st->lit[q]; pick = q; }   // first strict maximum
    if (direct) { pick = 0; best = __builtin_nan(""); }   // not a final step: the value is never read
    const int gap = pick >= 0 ? st->gap[pick] : 0, rev = pick >= 0 ? st->rev[pick] : 0;
    const int32_t* __restrict__ id_in = d.packed_cur;
    const int32_t* __restrict__ pos_in = d.packed_cur + S;
    const int32_t* __restrict__ rev_in = d.packed_cur + 2 * S + 1;
    const int P = pos_in[gap];
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos < n_new) {
        int v;
        if (pos < P) v = d.pos_cur[pos];
        else if (pos < P + L) v = d.new_start + (rev ? L - 1 - (pos - P) : pos - P);
        else v = d.pos_cur[pos - L];
        d.pos_nxt[pos] = v;
    }
    if (blockIdx.x == 0) {
        const int S1 = S + 1;
        int32_t* __restrict__ id_out = d.packed_nxt;
        int32_t* __restrict__ pos_out = d.packed_nxt + S1;
        int32_t* __restrict__ rev_out = d.packed_nxt + 2 * S1 + 1;
        for (int j = threadIdx.x; j < S1; j += 256) {
            id_out[j] = j < gap ? id_in[j] : (j == gap ? d.new_id : id_in[j - 1]);
            rev_out[j] = j < gap ? rev_in[j] : (j == gap ? rev : rev_in[j - 1]);
        }
        for (int j = threadIdx.x; j <= S1; j += 256) pos_out[j] = j <= gap ? pos_in[j] : pos_in[j - 1] + L;
        if (threadIdx.x == 0) { d.log->gap = gap; d.log->rev = rev; d.log->best = pick >= 0 ? best : 0.0; d.log->n_short = direct ? -1 : ns; }
    }
}

void launch_insb_apply(const InsStep* steps, int n_chrom, int max_n_used, hipStream_t s)
{
    hipLaunchKernelGGL(k_insb_apply, dim3((max_n_used + 255) / 256, n_chrom), dim3(256), 0, s, steps);
}

}  // namespace hicmi
