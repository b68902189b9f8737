// k_part1_scan.hip - the two cut-scan loops of Part 1 with their control flow on the device.
//
//   pre_process_all_matrix_breakpoints   scaffoldToChromosomes.py:513-551 (+ find_matrix_pvalue_breakpoints, S2C:413-511,
//                                        get_sliding_window_distance_metrics, S2C:370-411)
//   filter_noisy_breakpoints             scaffoldToChromosomes.py:553-727
//
// Both are chains of scans in which every scan's arguments come out of the previous scan's flags: ~230 scans per 16k map,
// each a few tens of microseconds of device work - and, driven from the host, as much again of launch + download +
// decision latency.  Here a scan is two launches that take their arguments from a small state record in device memory:
//   k_cut_rows     one workgroup per row: the count over the row's rank segment (the k_cut_count query) and, by its first
//                  lane, the hypergeometric decision -> one flag per row
//   k_*_decide     one workgroup: reads the flags, takes the reference's decision, writes the next scan's arguments
// so the host enqueues a batch of such pairs back to back and only looks at the state record once per batch; launches
// behind the end of the loop find `done` set and return.  The decisions are the reference's, statement by statement
// (cited below); everything is integer work except the scalar p-values of the filter, which use the same hyper.h routine
// as the host-side hicmi_hypergeom_sf.
#include "hicmi_internal.h"
#include "hyper.h"

namespace hicmi {

// ---- the scan itself ------------------------------------------------------------------------------------------------
// mode 0 (first pass, S2C:455-469): entry t = row start + t counts #{ j in [start, start + t] : rank < t } and tests
//   hyper_geom(x, M, t, t) >= psig -> 0, otherwise (NaN included) 1; entry 0 is 0.
// mode 1 (filter, S2C:626-636): entry t < n_rows counts #{ j in [start, cut] : rank < cut - start } and tests
//   hyper_geom(x, M, L, L) < psig -> 1 with L = cut - start, otherwise (NaN included) 0.
__global__ __launch_bounds__(256) void k_cut_rows(const uint16_t* __restrict__ rank, int64_t ldr, int n,
                                                  const ScanState* __restrict__ st, int32_t* __restrict__ x,
                                                  uint8_t* __restrict__ sig, double psig)
{
    __shared__ int s_part[4];
    if (st->done) return;
    const int mode = st->mode, lo = st->start, t = blockIdx.x;
    const int rows = mode == 0 ? n - lo : st->n_rows;
    if (t >= rows) return;
    const int tid = threadIdx.x;
    if (mode == 0 && t == 0) { if (tid == 0) sig[0] = 0; return; }
    const int i = lo + t;
    const int hi = mode == 0 ? i : st->cut;
    const int thr = hi - lo;
    int total;
    if (st->recount) {
        const uint16_t* __restrict__ r = rank + (int64_t)i * ldr;
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
        if (tid != 0) return;
        total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        x[t] = total;
    } else {
        if (tid != 0) return;
        total = x[t];                                 // the same start: only M has changed (S2C:473-483)
    }
    const int64_t M = st->M;
    if (mode == 0) {
        const int dec = hypergeom_decide((int64_t)total, M, (int64_t)t, (int64_t)t, psig);
        sig[t] = dec == 0 ? 0 : 1;
    } else {
        const int dec = hypergeom_decide((int64_t)total, M, (int64_t)thr, (int64_t)thr, psig);
        sig[t] = dec == 1 ? 1 : 0;
    }
}

// ---- first pass: what happens between two scans (S2C:430-511 and 513-551) --------------------------------------------
__global__ __launch_bounds__(1024) void k_first_pass_decide(int n, ScanState* __restrict__ st, const uint8_t* __restrict__ sig,
                                                            int32_t* __restrict__ cuts, int32_t* __restrict__ mlog, int log_cap)
{
    __shared__ int s_sum[16];
    __shared__ int s_first;
    __shared__ int s_go;
    if (st->done) return;
    const int tid = threadIdx.x;
    const int start = st->start, cnt = n - start, h = st->min_size;
    // (int(sig.sum()) / len(sig)) >= .9  (S2C:473)
    int part = 0;
    for (int e = tid; e < cnt; e += 1024) part += sig[e];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((tid & 63) == 0) s_sum[tid >> 6] = part;
    if (tid == 0) s_first = 0x7fffffff;
    __syncthreads();
    if (tid == 0) {
        int total = 0;
        for (int w = 0; w < 16; w++) total += s_sum[w];
        int go = 1;                                      // 1: on to the window scan, 0: scan again with the new M
        st->scans++;
        if (st->recount) { const unsigned long long m = (unsigned long long)(cnt - 1); st->bytes += m * (m + 3); }   // sum_{L=1..m} 2 (L+1)
        st->loop_count++;
        if ((double)total / (double)cnt >= .9) {
            const int64_t morg = st->M;
            st->M = morg - start;                        // M = int(M - start)  (S2C:475-477)
            if (st->n_log < log_cap) { mlog[2 * st->n_log] = (int32_t)morg; mlog[2 * st->n_log + 1] = (int32_t)st->M; }
            st->n_log++;
            if (st->loop_count < 5) { go = 0; st->recount = 0; }      // S2C:482-483: at most 5 scans per start
        }
        s_go = go;
    }
    __syncthreads();
    if (!s_go) return;
    // window scores (S2C:390-400): left half minus right half; == min_size means h flags set followed by h clear (S2C:488)
    const int full = cnt - 2 * h + 1;                    // windows whose right half is complete; the others score 0
    if (h < cnt) {                                       // (window_size >= len -> ["NA","NA","NA"], S2C:380-381)
        for (int p = tid; p < full; p += 1024) {
            bool ok = true;
            for (int k = 0; k < h && ok; k++) ok = sig[p + k] == 1 && sig[p + h + k] == 0;
            if (ok) { atomicMin(&s_first, p); break; }  // later windows of this lane are further right
        }
    }
    __syncthreads();
    if (tid != 0) return;
    if (s_first == 0x7fffffff) { st->done = 1; return; }                 // no cut: S2C:538-539
    const int ind = start + s_first + h;                                 // ind += pre_cut_inds[0]  (S2C:540)
    cuts[st->n_cuts++] = ind;
    if (ind >= st->stop_ind || (n - ind) <= h) { st->done = 1; return; } // S2C:544-545
    st->start = ind; st->M = n - ind; st->loop_count = 0; st->recount = 1;
}

// ---- filter: what happens between two scans (S2C:553-727) ------------------------------------------------------------
// Lists live in global memory behind the state record: `alt` (the candidate cuts still in play, `alt_off` = what
// altered[keep_from:] has dropped), `filt` / `prev` (the dictionaries `filtered` / `prev_filtered` as one byte per index).
__device__ __forceinline__ void filter_begin_round(ScanState* st, int n)
{
    st->M = n - st->start;                               // M = n - start  (S2C:598)
    st->f_i = 0;
    st->f_keep_from = 0;
}

// arguments of the scan for candidate alt[alt_off + f_i] (S2C:604-631)
__device__ __forceinline__ void filter_set_scan(ScanState* st, const int32_t* alt, int n)
{
    st->cut = alt[st->alt_off + st->f_i];
    const int rest = n - st->start;
    st->n_rows = rest < st->MD + 1 ? rest : st->MD + 1;
    st->recount = 1;
}

static constexpr int FD_CAP = 2048;                     // candidates kept in LDS; longer lists are walked in global memory

__global__ __launch_bounds__(1024) void k_filter_decide(int n, ScanState* __restrict__ st, const uint8_t* __restrict__ sig,
                                                        int32_t* __restrict__ alt, uint8_t* __restrict__ filt,
                                                        uint8_t* __restrict__ prev, int32_t* __restrict__ seg_g,
                                                        int32_t* __restrict__ seg_x_g, double psig)
{
    __shared__ int s_alt[FD_CAP], s_seg[3 * FD_CAP], s_seg_x[FD_CAP];
    __shared__ int s_nseg, s_last, s_phase, s_diff, s_wave[16];
    if (st->done) return;
    const int tid = threadIdx.x;
    const int start = st->start, c = st->cut, n_rows = st->n_rows, n_alt = st->n_alt - st->alt_off;
    const bool in_lds = n_alt <= FD_CAP;
    int* a = in_lds ? s_alt : alt + st->alt_off;
    int* seg = in_lds ? s_seg : seg_g;                   // [3k] = lo, [3k+1] = hi, [3k+2] = index of hi in the list
    int* seg_x = in_lds ? s_seg_x : seg_x_g;
    if (in_lds)
        for (int k = tid; k < n_alt; k += 1024) s_alt[k] = alt[st->alt_off + k];
    __syncthreads();
    // segments between consecutive candidates, from `start` (S2C:640-652)
    if (tid == 0) {
        int ns = 0, fc_prev = start;
        for (int k = 0; k < n_alt; k++) {
            const int ai = a[k];
            if (ai == fc_prev) continue;
            const int lo = fc_prev, hi = ai;
            fc_prev = ai;
            if (hi <= lo) break;
            seg[3 * ns] = lo; seg[3 * ns + 1] = hi; seg[3 * ns + 2] = k;
            seg_x[ns] = 0;
            ns++;
        }
        s_nseg = ns; s_last = -1;
        st->scans++;
        st->bytes += 2ull * (unsigned long long)n_rows * (unsigned long long)(c - start + 1);
    }
    __syncthreads();
    const int ns = s_nseg;
    // x = flags[lo - start : hi - start].sum(), rows beyond start + n_rows being 0 (S2C:626-628, 660): every set flag
    // finds its segment (they are consecutive and ascending: first segment whose hi lies beyond the row) and adds itself
    for (int e = tid; e < n_rows; e += 1024) {
        if (!sig[e]) continue;
        const int row = start + e;
        int lo_k = 0, hi_k = ns;                          // first k with seg[3k+1] > row
        while (lo_k < hi_k) {
            const int mid = (lo_k + hi_k) >> 1;
            if (seg[3 * mid + 1] > row) hi_k = mid; else lo_k = mid + 1;
        }
        if (lo_k < ns && seg[3 * lo_k] <= row) atomicAdd(&seg_x[lo_k], 1);
    }
    __syncthreads();
    // noise_pval = hyper_geom(x, M, local, hi - lo) < psig  (S2C:664-671); the last significant segment wins (S2C:674-676)
    const int64_t M = st->M;
    const int local = c - start;
    for (int k = tid; k < ns; k += 1024) {
        const double p = hypergeom_sf_ge((int64_t)seg_x[k], M, (int64_t)local, (int64_t)(seg[3 * k + 1] - seg[3 * k]));
        if (p < psig) atomicMax(&s_last, k);
    }
    __syncthreads();
    if (tid == 0) {
        // phase 0: next candidate of this round; 1: the round is over, a new one starts; 2: the pass is over
        int phase = 0;
        if (s_last >= 0) {                                // S2C:674-680
            st->f_keep_from = seg[3 * s_last + 2];
            st->start = seg[3 * s_last + 1];
            filt[st->start] = 1;
            st->f_noise = 1;
            phase = 1;
        } else {
            filt[c] = 1;                                  // S2C:682-683
            st->f_keep_from = st->f_i;
            st->f_i++;
            if (st->f_i >= n_alt) phase = 1;
        }
        if (phase == 1) {                                 // S2C:685-690
            st->f_round++;
            if (st->f_noise == 0) phase = 2;
            else {
                st->alt_off += st->f_keep_from;           // altered = altered[keep_from:]
                st->f_noise = 0;
                if (st->f_round >= st->f_max_rounds) { st->f_warned++; phase = 2; }     // S2C:592-595
                else if (st->alt_off >= st->n_alt) phase = 2;                            // (an empty list scans nothing)
                else filter_begin_round(st, n);
            }
        }
        s_phase = phase;
        s_diff = 0;
    }
    __syncthreads();
    if (s_phase != 2) {
        if (tid == 0) filter_set_scan(st, alt, n);
        return;
    }
    // prev_filtered != filtered ?  altered = sorted(filtered), prev_filtered = filtered, again : done   (S2C:692-697)
    __threadfence_block();
    const int chunk = (n + 1023) / 1024;                  // each lane owns `chunk` consecutive indices
    const int e0 = tid * chunk, e1 = e0 + chunk < n ? e0 + chunk : n;
    int mine = 0, diff = 0;
    for (int e = e0; e < e1; e++) { const int f = filt[e]; mine += f; diff |= f != prev[e]; }
    if (diff) s_diff = 1;
    int incl = mine;                                      // inclusive prefix of the counts: wave scan, then the wave totals
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if ((tid & 63) >= d) incl += o; }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    if (!s_diff) { if (tid == 0) st->done = 1; return; }  // `filtered` (== prev_filtered) is the answer
    int base = incl - mine, total = 0;
    for (int w = 0; w < 16; w++) { if (w < (tid >> 6)) base += s_wave[w]; total += s_wave[w]; }
    for (int e = e0; e < e1; e++) {
        const int f = filt[e];
        if (f) alt[base++] = e;
        prev[e] = (uint8_t)f; filt[e] = 0;
    }
    if (tid == 0) {
        st->n_alt = total; st->alt_off = 0;
        st->start = 0; st->f_round = 0; st->f_noise = 0;
        if (total == 0) st->done = 1;                     // (cannot happen: every round files at least one index)
    }
    __syncthreads();
    if (tid == 0 && total) { __threadfence_block(); filter_begin_round(st, n); filter_set_scan(st, alt, n); }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
void launch_first_pass_pairs(const uint16_t* rank, int64_t ldr, int n, ScanState* st, int32_t* x, uint8_t* sig, double psig,
                             int32_t* cuts, int32_t* mlog, int log_cap, int pairs, hipStream_t s)
{
    for (int k = 0; k < pairs; k++) {
        hipLaunchKernelGGL(k_cut_rows, dim3(n), dim3(256), 0, s, rank, ldr, n, st, x, sig, psig);
        hipLaunchKernelGGL(k_first_pass_decide, dim3(1), dim3(1024), 0, s, n, st, sig, cuts, mlog, log_cap);
    }
}

void launch_filter_pairs(const uint16_t* rank, int64_t ldr, int n, int max_rows, ScanState* st, int32_t* x, uint8_t* sig,
                         double psig, int32_t* alt, uint8_t* filt, uint8_t* prev, int32_t* seg, int32_t* seg_x, int pairs,
                         hipStream_t s)
{
    for (int k = 0; k < pairs; k++) {
        hipLaunchKernelGGL(k_cut_rows, dim3(max_rows), dim3(256), 0, s, rank, ldr, n, st, x, sig, psig);
        hipLaunchKernelGGL(k_filter_decide, dim3(1), dim3(1024), 0, s, n, st, sig, alt, filt, prev, seg, seg_x, psig);
    }
}

}  // namespace hicmi
