// k_nnchain.hip - UPGMA by nearest-neighbour chain: SciPy's _hierarchy.nn_chain for
// method='average' (scaffoldToChromosomes.py:197; algorithm restated in SURVEY.md A3 and
// oracle/oracle_c.c), bit for bit.
//
// The algorithm is a chain of ~3(n-1) DEPENDENT O(n) steps - row scans for the nearest neighbour and
// Lance-Williams updates - each far too small to amortise a grid-wide barrier (a 16k-bin row is
// 128 KB; an XCD-hierarchical grid barrier costs ~5 us, about what one CU needs to stream the row).
// So the chain runs as ONE persistent 1024-lane workgroup:
//   scan   : lanes stream row x of W with 16-byte loads, keep (min, lowest index) per lane,
//            wave-shuffle + LDS arg-min; strict '<' + index order == SciPy's tie rule, and the
//            previous chain element is preferred exactly as SciPy does;
//   update : (nx*d_xi + ny*d_yi)/(nx+ny) with five separate fp64 roundings; row y is rewritten with
//            coalesced stores.
// What a single CU cannot do cheaply is the other half of keeping W symmetric: scattering column y
// (n 8-byte stores to n different lines per merge).  That is DEFERRED: a merged cluster becomes
// "dirty" (time-stamped in LDS); for a pair (a, b) the row of the cluster that merged LAST is
// authoritative, so scans and updates read a dirty partner's value from the partner's own fresh row
// (one extra gathered load per dirty cluster, all in flight together).  After DCAP merges the
// workgroup saves its state and exits; k_nn_flush - a full-chip kernel - writes all dirty columns at
// once, and the next epoch resumes.  Launches are queued back to back without host synchronisation.
//
// Liveness / dirty bitmasks and cluster sizes live in LDS; the chain lives in global memory with its
// top 256 entries mirrored in LDS.
#include "hicmi_internal.h"

namespace hicmi {

static constexpr int NN_THREADS = 1024;
static constexpr int NN_DMAX = 1024;                     // at most one dirty entry per lane
static constexpr int NN_MAXWG = 16;                      // workgroups of the column-sliced chain (k_nn_epoch_mw / _mwc)
static constexpr int NN_HEAD = 1536;                     // state 64 B, counters 64 B, hand-off slot 128 B, mailboxes 1024 B, profile detail 256 B
static constexpr int NN_W1_MAXS = 64;                    // column slices (single-wave workgroups) of k_nn_epoch_w1
static constexpr int NN_W1_MAIL = 2 * NN_W1_MAXS * 2 * NN_W1_MAXS * 16;   // its mailboxes: 2 parities x 64 readers x (2 slots x 64 writers) x 16 bytes

struct ArgMin { double v; int i; };

// num / fs, correctly rounded, for fs an integer-valued double below 2^17 and rcp = RN(1 / fs):
// q0 = RN(num*rcp) is within 2 ulp of the quotient, r = num - fs*q0 is exact in an FMA, and
// q0 + r*rcp equals num/fs to within 2^-105 relative - far closer than a quotient with a 17-bit
// divisor can come to a rounding boundary (>= 2^-71 relative) - so the final rounding is the correct
// one.  Three instructions instead of the ~30 of a general fp64 division; hicmi_selftest_division
// compares it with '/' on random operands.
__device__ __forceinline__ double div_by_small_int(double num, double fs, double rcp)
{
    double q = num * rcp;
    double r = fma(-fs, q, num);
    return fma(r, rcp, q);
}

// Lexicographic (value, index) minimum over the 64 lanes of a wave, result in every lane.  Inside a
// row of 16 lanes the exchange is four DPP moves (quad swaps, half-row mirror, row mirror: register to
// register, no LDS round trip like ds_bpermute); the four row results are read with v_readlane and
// combined as scalars.  The minimum is exact whatever the order of the comparisons.
template <int CTRL>
__device__ __forceinline__ ArgMin argmin_dpp_step(ArgMin a)
{
    const int lo = __double2loint(a.v), hi = __double2hiint(a.v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(a.i, a.i, CTRL, 0xf, 0xf, false);
    const double ov = __hiloint2double(ohi, olo);
    if (ov < a.v || (ov == a.v && oi < a.i)) { a.v = ov; a.i = oi; }
    return a;
}

__device__ __forceinline__ ArgMin argmin_row16(ArgMin a)
{
    a = argmin_dpp_step<0xB1>(a);                          // quad_perm [1,0,3,2]
    a = argmin_dpp_step<0x4E>(a);                          // quad_perm [2,3,0,1]
    a = argmin_dpp_step<0x141>(a);                         // row_half_mirror
    a = argmin_dpp_step<0x140>(a);                         // row_mirror
    return a;
}

__device__ __forceinline__ ArgMin argmin_readlane(ArgMin a, int src)
{
    ArgMin r;
    r.v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a.v), src), __builtin_amdgcn_readlane(__double2loint(a.v), src));
    r.i = __builtin_amdgcn_readlane(a.i, src);
    return r;
}

__device__ __forceinline__ ArgMin argmin_wave(ArgMin a)
{
    a = argmin_row16(a);
    ArgMin m = argmin_readlane(a, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const ArgMin o = argmin_readlane(a, r);
        if (o.v < m.v || (o.v == m.v && o.i < m.i)) m = o;
    }
    return m;
}

// ---- (value, index) minimum that also knows whether the minimum is attained more than once -----------------------
// The neighbour cache (k_nn_epoch_nc) may answer "who is the nearest neighbour of x" without the distance only when
// no second column ties with it (SciPy prefers the previous chain element on an exact tie).
struct ArgMinT { double v; int i; int t; };

__device__ __forceinline__ ArgMinT argmint_join(ArgMinT a, double ov, int oi, int ot)
{
    if (ov < a.v || (ov == a.v && oi < a.i)) { a.t = ot | (ov == a.v ? 1 : 0); a.v = ov; a.i = oi; }
    else if (ov == a.v && oi != a.i) a.t = 1;
    return a;
}

template <int CTRL>
__device__ __forceinline__ ArgMinT argmint_dpp_step(ArgMinT a)
{
    const int lo = __double2loint(a.v), hi = __double2hiint(a.v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(a.i, a.i, CTRL, 0xf, 0xf, false);
    const int ot = __builtin_amdgcn_update_dpp(a.t, a.t, CTRL, 0xf, 0xf, false);
    return argmint_join(a, __hiloint2double(ohi, olo), oi, ot);
}

__device__ __forceinline__ ArgMinT argmint_row16(ArgMinT a)
{
    a = argmint_dpp_step<0xB1>(a);
    a = argmint_dpp_step<0x4E>(a);
    a = argmint_dpp_step<0x141>(a);
    a = argmint_dpp_step<0x140>(a);
    return a;
}

__device__ __forceinline__ ArgMinT argmint_wave(ArgMinT a)
{
    a = argmint_row16(a);
    ArgMinT m;
    m.v = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a.v), 0), __builtin_amdgcn_readlane(__double2loint(a.v), 0));
    m.i = __builtin_amdgcn_readlane(a.i, 0);
    m.t = __builtin_amdgcn_readlane(a.t, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ov = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a.v), r), __builtin_amdgcn_readlane(__double2loint(a.v), r));
        m = argmint_join(m, ov, __builtin_amdgcn_readlane(a.i, r), __builtin_amdgcn_readlane(a.t, r));
    }
    return m;
}

// The same result as argmint_wave for non-NaN values, in about a third of the instructions: first the minimum VALUE
// (v_min_f64 over DPP moves - no compare-and-select on three registers per step), then the lowest INDEX among the lanes
// that hold it (v_min_u32), then "is it attained more than once" as one ballot.  NaN never wins a v_min (IEEE minNum),
// exactly as it never wins the `<` of argmint_join.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double fmin_dpp_step(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return fmin(v, __hiloint2double(ohi, olo));
}
template <int CTRL>
__device__ __forceinline__ unsigned int umin_dpp_step(unsigned int v)
{
    const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ double readlane_f64(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// rows: how many rows of 16 lanes take part (4 = the whole wave; 1 or 2: the first 16 / 32 lanes, each ROW reduced on its own
// when per_row is set - used for the two results of the fused pass that sit in rows 0 and 1)
__device__ __forceinline__ ArgMinT argmint_wave_fast(ArgMinT a)
{
    double m = a.v;
    m = fmin_dpp_step<0xB1>(m); m = fmin_dpp_step<0x4E>(m); m = fmin_dpp_step<0x141>(m); m = fmin_dpp_step<0x140>(m);
    const double m0 = readlane_f64(m, 0), m1 = readlane_f64(m, 16), m2 = readlane_f64(m, 32), m3 = readlane_f64(m, 48);
    const double mv = fmin(fmin(m0, m1), fmin(m2, m3));
    const bool at_min = a.v == mv;
    unsigned int ix = at_min ? (unsigned int)a.i : 0xffffffffu;
    ix = umin_dpp_step<0xB1>(ix); ix = umin_dpp_step<0x4E>(ix); ix = umin_dpp_step<0x141>(ix); ix = umin_dpp_step<0x140>(ix);
    const unsigned int i0 = (unsigned int)__builtin_amdgcn_readlane((int)ix, 0), i1 = (unsigned int)__builtin_amdgcn_readlane((int)ix, 16);
    const unsigned int i2 = (unsigned int)__builtin_amdgcn_readlane((int)ix, 32), i3 = (unsigned int)__builtin_amdgcn_readlane((int)ix, 48);
    unsigned int mi = i0 < i1 ? i0 : i1;
    const unsigned int mi2 = i2 < i3 ? i2 : i3;
    mi = mi < mi2 ? mi : mi2;
    ArgMinT r;
    r.v = mv;
    r.i = mi == 0xffffffffu ? 0x7fffffff : (int)mi;       // nothing compared (all NaN / no candidates)
    r.t = __ballot(at_min && ((unsigned int)a.i != mi || a.t != 0)) != 0ull ? 1 : 0;
    return r;
}
// the same per row of 16 lanes (result in every lane of its row): the cross-wave step of the reductions
__device__ __forceinline__ ArgMinT argmint_row16_fast(ArgMinT a)
{
    double m = a.v;
    m = fmin_dpp_step<0xB1>(m); m = fmin_dpp_step<0x4E>(m); m = fmin_dpp_step<0x141>(m); m = fmin_dpp_step<0x140>(m);
    const bool at_min = a.v == m;
    unsigned int ix = at_min ? (unsigned int)a.i : 0xffffffffu;
    ix = umin_dpp_step<0xB1>(ix); ix = umin_dpp_step<0x4E>(ix); ix = umin_dpp_step<0x141>(ix); ix = umin_dpp_step<0x140>(ix);
    // tie: some lane of the row holds the minimum at another index, or its own flag is set
    unsigned int tf = (at_min && ((unsigned int)a.i != ix || a.t != 0)) ? 1u : 0u;
    tf |= (unsigned int)__builtin_amdgcn_update_dpp((int)tf, (int)tf, 0xB1, 0xf, 0xf, false);
    tf |= (unsigned int)__builtin_amdgcn_update_dpp((int)tf, (int)tf, 0x4E, 0xf, 0xf, false);
    tf |= (unsigned int)__builtin_amdgcn_update_dpp((int)tf, (int)tf, 0x141, 0xf, 0xf, false);
    tf |= (unsigned int)__builtin_amdgcn_update_dpp((int)tf, (int)tf, 0x140, 0xf, 0xf, false);
    ArgMinT r;
    r.v = m; r.i = ix == 0xffffffffu ? 0x7fffffff : (int)ix; r.t = (int)tf;
    return r;
}

// workspace layout (all 16-byte aligned)
struct NNWorkspace {
    int* state;                 // [0] step [1] len [2] top [3] second [4] first_ptr [5] stop code [6] n_dirty [7] n after compaction
                                // [8] profile on [9] first step of the epoch that ran last [10] test: exchange that is declared late
                                // [11] test: step whose record replica 1 falsifies
    unsigned long long* prof;   // [0..4] phase totals (100 MHz ticks), [5] columns visited by row scans, [6] row scans,
                                // [7] chain steps answered by the neighbour cache
    void* mail;                 // k_nn_epoch_mw: 2 x NN_MAXWG 16-byte mailbox slots; k_nn_epoch_mwc: two slots per workgroup
    uint32_t* alive;            // nwords
    uint16_t* size;             // n
    int* gtime;                 // n: dirty time stamp of a slot in the finished epoch, -1 = clean
    int* dslot;                 // NN_DMAX
    int* dtime;                 // NN_DMAX
    int* orig;                  // n: original bin of each current slot (changes at every compaction)
    int* newidx;                // n: scratch of the compaction (old slot -> new slot, -1 = dead)
    int* oldidx;                // n: scratch of the compaction (new slot -> old slot)
    double* nnval;              // n: neighbour cache - distance to the nearest live cluster of each slot
    uint32_t* nnc;              // n: neighbour cache - its slot (low 16 bits, 0xffff = unknown) | tie flag << 16
    double* rec;                // NN_MAXWG x NN_DMAX x 4: the merges of the last epoch as every replica of k_nn_epoch_mw saw them
    int* size_rep;              // NN_W1_MAXS x n: cluster sizes, one private copy per replica of k_nn_epoch_w1 and of
                                // k_nn_epoch_mwc<.., .., true> (rows beyond 32,768 columns: the sizes do not fit the LDS beside the cache)
    void* mailw;                // k_nn_epoch_w1: NN_W1_MAIL bytes of mailboxes, then NN_W1_MAXS 8-byte merge-record hashes, then 16 bytes
                                // per lane and replica where the stores of masked-out elements land
};
static constexpr int NN_STOP_GUARD = 1, NN_STOP_LATE = 2, NN_STOP_DIVERGED = 3;
static constexpr uint32_t NN_NOIDX = 0xffffu;

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

size_t nnchain_workspace_bytes(int n)
{
    size_t nwords = (size_t)(n + 31) / 32;
    return NN_HEAD + align16(nwords * 4) + align16((size_t)n * 2) + 4 * align16((size_t)n * 4) + 2 * align16(NN_DMAX * 4) +
           align16((size_t)n * 8) + align16((size_t)n * 4) + (size_t)NN_MAXWG * NN_DMAX * 4 * 8 +
           (size_t)NN_W1_MAXS * align16((size_t)n * 4) + NN_W1_MAIL + NN_W1_MAXS * 8 + NN_W1_MAXS * 64 * 16;
}

static NNWorkspace carve(void* ws, int n)
{
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    size_t nwords = (size_t)(n + 31) / 32;
    NNWorkspace w;
    w.state = reinterpret_cast<int*>(p);
    w.prof = reinterpret_cast<unsigned long long*>(p + 64);
    w.mail = p + 256;                                        // 2 parities x NN_MAXWG workgroups x 2 slots x 16 bytes
    p += NN_HEAD;
    w.alive = reinterpret_cast<uint32_t*>(p); p += align16(nwords * 4);
    w.size = reinterpret_cast<uint16_t*>(p); p += align16((size_t)n * 2);
    w.gtime = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.dslot = reinterpret_cast<int*>(p); p += align16(NN_DMAX * 4);
    w.dtime = reinterpret_cast<int*>(p); p += align16(NN_DMAX * 4);
    w.orig = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.newidx = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.oldidx = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.nnval = reinterpret_cast<double*>(p); p += align16((size_t)n * 8);
    w.nnc = reinterpret_cast<uint32_t*>(p); p += align16((size_t)n * 4);
    w.rec = reinterpret_cast<double*>(p); p += (size_t)NN_MAXWG * NN_DMAX * 4 * 8;
    w.size_rep = reinterpret_cast<int*>(p); p += (size_t)NN_W1_MAXS * align16((size_t)n * 4);
    w.mailw = p;
    return w;
}

__global__ __launch_bounds__(256) void k_nn_init(NNWorkspace w, int n)
{
    const int nwords = (n + 31) >> 5;
    const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int i = gid; i < nwords; i += stride) {
        int rem = n - i * 32;
        w.alive[i] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    }
    for (int i = gid; i < n; i += stride) { w.size[i] = 1; w.gtime[i] = -1; w.orig[i] = i; }
    if (gid < 16) w.state[gid] = 0;
    if (gid < 8) w.prof[gid] = 0ull;
    if (gid < 32) reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(w.state) + 1280)[gid] = 0ull;   // profile detail
}

// PROFILE adds wall-clock stamps (100 MHz) around the phases, accumulated in w.prof[0..4] =
// {chain bookkeeping, row scan, pick neighbour, merge bookkeeping, Lance-Williams update}.
template <bool PROFILE>
__global__ __launch_bounds__(NN_THREADS) void k_nn_epoch(double* __restrict__ W, int64_t ld, int n,
                                                         int* __restrict__ chain, double* __restrict__ zraw,
                                                         NNWorkspace w, int dcap, int total_steps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    const int nwords = (n + 31) >> 5, nw4 = (nwords + 3) & ~3;
    uint32_t* alive = reinterpret_cast<uint32_t*>(smem_nn);
    uint32_t* smask = alive + nw4;                          // alive AND not dirty: what the streaming passes visit
    uint16_t* lsize = reinterpret_cast<uint16_t*>(smask + nw4);
    __shared__ int dslot[NN_DMAX], dtime[NN_DMAX];
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int ring[256];
    __shared__ double s_dprev;
    __shared__ int s_x, s_prev, s_done, s_stop, s_mx, s_my, s_nx, s_ny, s_tx, s_ty, s_ey;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int step = w.state[0];
    if (step >= total_steps || w.state[5]) {               // finished (or stopped) in an earlier epoch
        if (tid == 0) w.state[6] = 0;
        return;
    }
    for (int i = tid; i < nwords; i += NN_THREADS) { alive[i] = w.alive[i]; smask[i] = w.alive[i]; }
    for (int i = tid; i < n; i += NN_THREADS) { lsize[i] = w.size[i]; w.gtime[i] = -1; }
    // lane-0 private chain state
    int len = w.state[1], top = w.state[2], second = w.state[3], first_ptr = w.state[4], ring_lo = len;
    unsigned long long t_book = 0, t_scan = 0, t_pick = 0, t_merge = 0, t_upd = 0, t0 = 0, t1 = 0;
    unsigned long long c_cols = 0, c_scans = 0;
    if (tid == 0) { s_stop = 0; s_done = 0; w.state[9] = step; }
    __syncthreads();
    int D = 0;                                              // dirty entries (uniform across lanes)
    uint32_t xbit = 0u;                                     // lane 0: the scan row's own bit in smask

    for (; step < total_steps && D < dcap; step++) {
        if (PROFILE && tid == 0) t0 = wall_clock64();
        if (tid == 0 && len == 0) {
            while (first_ptr < n && !((alive[first_ptr >> 5] >> (first_ptr & 31)) & 1u)) first_ptr++;
            chain[0] = first_ptr; ring[0] = first_ptr; ring_lo = 0; top = first_ptr; second = -1; len = 1;
        }
        int guard = 0;
        double cur = 0.0;
        int ybest = -1;
        while (true) {
            if (tid == 0) {
                s_x = top; s_prev = (len > 1) ? second : -1; s_tx = -1;
                xbit = smask[top >> 5] & (1u << (top & 31));         // the row's own column is skipped by masking it
                smask[top >> 5] &= ~xbit;
                c_scans++; c_cols += (unsigned long long)(total_steps + 1 - step);
            }
            __syncthreads();
            const int x = s_x, prev = s_prev;
            if (tid < D && dslot[tid] == x) s_tx = dtime[tid];       // is the row itself dirty, and since when
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_book += t1 - t0; t0 = t1; }
            const int tx = s_tx;
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            // d(x, previous chain element) when that element is clean (a dirty one is handled below)
            if (tid == 64 && prev >= 0 && ((smask[prev >> 5] >> (prev & 31)) & 1u)) s_dprev = rowx[prev];
            // dirty partners: the value comes from whichever row was rewritten last
            ArgMin cand = {__builtin_inf(), 0x7fffffff};
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != x && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    const double v = dtime[tid] > tx ? W[(int64_t)d * ld + x] : rowx[d];
                    cand.v = v; cand.i = d;
                    if (d == prev) s_dprev = v;
                }
            }
            ArgMin best = {__builtin_inf(), 0x7fffffff};
#pragma unroll 8
            for (int j = tid * 2; j < n; j += 2 * NN_THREADS) {     // ascending j per lane: strict '<' keeps the lowest index
                double2 v = *reinterpret_cast<const double2*>(rowx + j);
                uint32_t bits = smask[j >> 5] >> (j & 31);          // j even: both bits in one word; bits past n are 0
                if ((bits & 1u) && v.x < best.v) { best.v = v.x; best.i = j; }
                if ((bits & 2u) && v.y < best.v) { best.v = v.y; best.i = j + 1; }
            }
            if (cand.v < best.v || (cand.v == best.v && cand.i < best.i)) best = cand;
            best = argmin_wave(best);
            if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_scan += t1 - t0; t0 = t1; }
            if (wave == 0) {
                ArgMin m = {lane < 16 ? s_v[lane] : __builtin_inf(), lane < 16 ? s_i[lane] : 0x7fffffff};
                m = argmin_row16(m);                                // the 16 wave results sit in row 0
                if (lane == 0) {
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
                    smask[x >> 5] |= xbit;                          // un-mask the row's own column
                    s_done = done;
                }
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
            zraw[4 * (int64_t)step + 0] = (double)xx;
            zraw[4 * (int64_t)step + 1] = (double)yy;
            zraw[4 * (int64_t)step + 2] = cur;
            zraw[4 * (int64_t)step + 3] = (double)(nx + ny);
            lsize[xx] = 0;
            lsize[yy] = (uint16_t)(nx + ny);
            alive[xx >> 5] &= ~(1u << (xx & 31));
            smask[xx >> 5] &= ~(1u << (xx & 31));
            s_mx = xx; s_my = yy; s_nx = nx; s_ny = ny; s_tx = -1; s_ty = -1; s_ey = -1;
            top = len >= 1 ? (len - 1 >= ring_lo ? ring[(len - 1) & 255] : chain[len - 1]) : -1;
            second = len >= 2 ? (len - 2 >= ring_lo ? ring[(len - 2) & 255] : chain[len - 2]) : -1;
        }
        __syncthreads();
        const int mx = s_mx, my = s_my;
        if (tid < D) {
            if (dslot[tid] == mx) s_tx = dtime[tid];
            if (dslot[tid] == my) { s_ty = dtime[tid]; s_ey = tid; }
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_merge += t1 - t0; t0 = t1; }
        {
            const int tmx = s_tx, tmy = s_ty;
            const double fx = (double)s_nx, fy = (double)s_ny, fs = (double)(s_nx + s_ny);
            const double rcp = 1.0 / fs;
            const double* __restrict__ rx = W + (int64_t)mx * ld;
            double* __restrict__ ry = W + (int64_t)my * ld;
            // dirty partners first (their loads overlap the streaming pass); results are stored after
            // the streaming pass has rewritten row y
            double dv = 0.0; int dd = -1;
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != my && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    const double dxi = dtime[tid] > tmx ? W[(int64_t)d * ld + mx] : rx[d];
                    const double dyi = dtime[tid] > tmy ? W[(int64_t)d * ld + my] : ry[d];
                    dv = div_by_small_int(fx * dxi + fy * dyi, fs, rcp);
                    dd = d;
                }
            }
#pragma unroll 4
            for (int j = tid * 2; j < n; j += 2 * NN_THREADS) {
                double2 a = *reinterpret_cast<const double2*>(rx + j);
                double2 b = *reinterpret_cast<const double2*>(ry + j);
                uint32_t bits = smask[j >> 5] >> (j & 31);
                if ((bits & 1u) && j != my) b.x = div_by_small_int(fx * a.x + fy * b.x, fs, rcp);
                if ((bits & 2u) && j + 1 != my) b.y = div_by_small_int(fx * a.y + fy * b.y, fs, rcp);
                *reinterpret_cast<double2*>(ry + j) = b;
            }
            __syncthreads();
            if (dd >= 0) ry[dd] = dv;
            if (tid == 0) {                                 // cluster y is dirty from now on
                if (s_ey >= 0) dslot[s_ey] = -1;            // its older entry is superseded
                dslot[D] = my; dtime[D] = step;
                smask[my >> 5] &= ~(1u << (my & 31));
            }
            D++;
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_upd += t1 - t0; }
    }
    // ---- save state for the flush kernel and the next epoch
    __syncthreads();
    for (int i = tid; i < nwords; i += NN_THREADS) w.alive[i] = alive[i];
    for (int i = tid; i < n; i += NN_THREADS) w.size[i] = lsize[i];
    if (tid < D) {
        w.dslot[tid] = dslot[tid]; w.dtime[tid] = dtime[tid];
        if (dslot[tid] >= 0) w.gtime[dslot[tid]] = dtime[tid];
    }
    if (tid == 0) {
        w.state[0] = step; w.state[1] = len; w.state[2] = top; w.state[3] = second; w.state[4] = first_ptr;
        w.state[5] = s_stop; w.state[6] = D;
        w.prof[5] += c_cols; w.prof[6] += c_scans;
        if (PROFILE) { w.prof[0] += t_book; w.prof[1] += t_scan; w.prof[2] += t_pick; w.prof[3] += t_merge; w.prof[4] += t_upd; }
    }
}

// ---- the chain with a neighbour cache (k_nn_epoch_nc) ----------------------------------------------------------
// SciPy's nn_chain re-scans a whole row at EVERY chain step: ~2.9 scans per merge on the Hi-C maps.  Most of those
// scans re-derive what is already known: the nearest neighbour of a row changes only when (a) that neighbour
// merges, or (b) a freshly merged cluster lands closer - and (b) is seen by the Lance-Williams update, which
// computes d(i, new) for every live i anyway.  So every slot keeps (nnval, nnidx, tie): distance and LOWEST slot of
// its nearest live cluster, and whether a second column attains the same distance.
//   update of merge (x, y) -> y : for every live j  v = d(j, y')
//        nnidx[j] in {x, y}                       -> unknown (the row is scanned when the chain next visits it)
//        v <  nnval[j]                            -> unknown              (fp rounding can do this: reducibility is not exact)
//        v == nnval[j]                            -> (v, min(nnidx[j], y'), tie)
//      and the new row's own minimum is reduced in the same pass (4 of 5 merged clusters are visited again).
//   chain step at x with previous element p: SciPy takes  argmin_{i != x} d(x, i)  with ties to the LOWEST index, but
//      prefers p when d(x, p) equals the minimum.  With the cache: nnidx[x] == p -> reciprocal pair; no tie flag ->
//      nnidx[x] (it is strictly closer than p); tie flag -> scan the row (the value decides).  The merge height
//      d(x, p) is read by one lane during the update, off the critical path.
// Result: ~1.26 scans per merge instead of ~2.9, every one of them a row whose neighbour really is unknown; the
// chain steps in between are a few LDS reads by lane 0.  Tie-heavy inputs fall back to scans and stay exact.
// Everything else (deferred column writes, epochs, compaction, DPP reductions) is k_nn_epoch's.
static constexpr int NN_NC_MAX = 24576;                  // largest live width whose cache fits the LDS next to the sizes

__global__ __launch_bounds__(256) void k_nn_rowmin(const double* __restrict__ W, int64_t ld, int n, NNWorkspace w)
{
    __shared__ double s_v[4];
    __shared__ int s_i[4], s_t[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    if (!((w.alive[r >> 5] >> (r & 31)) & 1u)) { if (tid == 0) w.nnc[r] = NN_NOIDX; return; }
    const double* __restrict__ row = W + (int64_t)r * ld;
    ArgMinT best = {__builtin_inf(), 0x7fffffff, 0};
    for (int j = tid * 2; j < n; j += 512) {                 // ascending j per lane; ld is a multiple of 16 and padding is +inf
        const double2 v = *reinterpret_cast<const double2*>(row + j);
        const uint32_t bits = w.alive[j >> 5] >> (j & 31);
        if ((bits & 1u) && j != r && v.x <= best.v) { if (v.x < best.v) { best.v = v.x; best.i = j; best.t = 0; } else best.t = 1; }
        if ((bits & 2u) && j + 1 != r && j + 1 < n && v.y <= best.v) { if (v.y < best.v) { best.v = v.y; best.i = j + 1; best.t = 0; } else best.t = 1; }
    }
    best = argmint_wave(best);
    if ((tid & 63) == 0) { s_v[tid >> 6] = best.v; s_i[tid >> 6] = best.i; s_t[tid >> 6] = best.t; }
    __syncthreads();
    if (tid == 0) {
        ArgMinT m = {s_v[0], s_i[0], s_t[0]};
        for (int q = 1; q < 4; q++) m = argmint_join(m, s_v[q], s_i[q], s_t[q]);
        w.nnval[r] = m.v;
        w.nnc[r] = (m.i >= 0 && m.i < n) ? ((uint32_t)m.i | ((uint32_t)(m.t ? 1 : 0) << 16)) : NN_NOIDX;
    }
}

template <bool PROFILE>
__global__ __launch_bounds__(NN_THREADS) void k_nn_epoch_nc(double* __restrict__ W, int64_t ld, int n,
                                                            int* __restrict__ chain, double* __restrict__ zraw,
                                                            NNWorkspace w, int dcap, int total_steps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    const int nwords = (n + 31) >> 5, nw4 = (nwords + 3) & ~3, n2 = (n + 7) & ~7;
    uint32_t* alive = reinterpret_cast<uint32_t*>(smem_nn);
    uint32_t* smask = alive + nw4;                          // alive AND not dirty: what the streaming passes visit
    uint32_t* tieb = smask + nw4;                           // the cached minimum of the slot is attained more than once
    uint16_t* lsize = reinterpret_cast<uint16_t*>(tieb + nw4);
    uint16_t* nnidx = lsize + n2;                           // cached nearest slot, NN_NOIDX = unknown
    __shared__ int dslot[NN_DMAX], dtime[NN_DMAX];
    __shared__ double s_v[16];
    __shared__ int s_i[16], s_t[16];
    __shared__ int ring[256];
    __shared__ double s_dprev;
    __shared__ int s_x, s_prev, s_act, s_stop, s_mx, s_my, s_nx, s_ny, s_tx, s_ty, s_ey;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The cached distances are streamed by every update and must not be stored to inside that loop: a store through
    // the same pointer makes the compiler order every later load of the unrolled loop behind it - one memory round trip
    // per iteration instead of one per pass (measured: update 39 -> 107 ms at 16k).  So a row whose new distance is
    // strictly SMALLER than its cached minimum (fp rounding; rare) simply loses its cache entry; only lane 0 stores
    // distances (scan results, the merged row's minimum), between the passes.
    const double* nnval_ld = w.nnval;
    int step = w.state[0];
    if (step >= total_steps || w.state[5]) {               // finished (or stopped) in an earlier epoch
        if (tid == 0) w.state[6] = 0;
        return;
    }
    for (int i = tid; i < nwords; i += NN_THREADS) { alive[i] = w.alive[i]; smask[i] = w.alive[i]; tieb[i] = 0u; }
    __syncthreads();
    for (int i = tid; i < n; i += NN_THREADS) {
        lsize[i] = w.size[i]; w.gtime[i] = -1;
        const uint32_t c = w.nnc[i];
        nnidx[i] = (uint16_t)(c & 0xffffu);
        if (c >> 16) atomicOr(&tieb[i >> 5], 1u << (i & 31));
    }
    // lane-0 private chain state
    int len = w.state[1], top = w.state[2], second = w.state[3], first_ptr = w.state[4], ring_lo = len;
    unsigned long long t_book = 0, t_scan = 0, t_pick = 0, t_merge = 0, t_upd = 0, t0 = 0, t1 = 0;
    unsigned long long c_cols = 0, c_scans = 0, c_hits = 0;
    if (tid == 0) { s_stop = 0; w.state[9] = step; }
    __syncthreads();
    int D = 0;                                              // dirty entries (uniform across lanes)
    uint32_t xbit = 0u;                                     // lane 0: the scan row's own bit in smask
    int guard = 0, stop_code = 0;                           // lane 0
    bool stopped = false;                                   // uniform

    for (; step < total_steps && D < dcap; step++) {
        if (PROFILE && tid == 0) t0 = wall_clock64();
        while (true) {
            // ---- walk the chain on cached neighbours (lane 0) until a row has to be scanned or a pair is reciprocal
            if (tid == 0) {
                if (len == 0) {
                    while (first_ptr < n && !((alive[first_ptr >> 5] >> (first_ptr & 31)) & 1u)) first_ptr++;
                    chain[0] = first_ptr; ring[0] = first_ptr; ring_lo = 0; top = first_ptr; second = -1; len = 1;
                }
                int act = 0;                                 // 1: scan row `top`, 2: merge (top, second), 3: stop
                while (act == 0) {
                    const int x = top, prev = (len > 1) ? second : -1;
                    const uint32_t idx = nnidx[x];
                    if (idx == NN_NOIDX) act = 1;
                    else if ((int)idx == prev) act = 2;
                    else if (prev >= 0 && ((tieb[x >> 5] >> (x & 31)) & 1u)) act = 1;      // an exact tie: the value decides
                    else if (++guard > 4 * n + 8) { stop_code = NN_STOP_GUARD; act = 3; }
                    else {
                        chain[len] = (int)idx; ring[len & 255] = (int)idx;
                        if (len - 255 > ring_lo) ring_lo = len - 255;
                        second = top; top = (int)idx; len++;
                        c_hits++;
                    }
                }
                s_act = act; s_x = top; s_prev = (len > 1) ? second : -1; s_tx = -1;
                if (act == 1) {
                    xbit = smask[top >> 5] & (1u << (top & 31));     // the row's own column is skipped by masking it
                    smask[top >> 5] &= ~xbit;
                    c_scans++; c_cols += (unsigned long long)(total_steps + 1 - step);
                }
            }
            __syncthreads();
            if (s_act != 1) break;
            // ---- scan row x: lexicographic (value, index) minimum over the live columns, and whether it is unique
            const int x = s_x, prev = s_prev;
            if (tid < D && dslot[tid] == x) s_tx = dtime[tid];       // is the row itself dirty, and since when
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_book += t1 - t0; t0 = t1; }
            const int tx = s_tx;
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            // gathered values are loaded here and used after the streaming loop (see the update below)
            const bool want_dp = tid == 64 && prev >= 0 && ((smask[prev >> 5] >> (prev & 31)) & 1u);
            double dpv = 0.0, cv = 0.0; int cd = -1;
            if (want_dp) dpv = rowx[prev];
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != x && ((alive[d >> 5] >> (d & 31)) & 1u)) { cv = dtime[tid] > tx ? W[(int64_t)d * ld + x] : rowx[d]; cd = d; }
            }
            ArgMinT best = {__builtin_inf(), 0x7fffffff, 0};
#pragma unroll 8
            for (int j = tid * 2; j < n; j += 2 * NN_THREADS) {     // ascending j per lane: the first of equal values stays
                double2 v = *reinterpret_cast<const double2*>(rowx + j);
                uint32_t bits = smask[j >> 5] >> (j & 31);          // j even: both bits in one word; bits past n are 0
                if ((bits & 1u) && v.x <= best.v) { if (v.x < best.v) { best.v = v.x; best.i = j; best.t = 0; } else best.t = 1; }
                if ((bits & 2u) && v.y <= best.v) { if (v.y < best.v) { best.v = v.y; best.i = j + 1; best.t = 0; } else best.t = 1; }
            }
            if (want_dp) s_dprev = dpv;
            if (cd >= 0) { best = argmint_join(best, cv, cd, 0); if (cd == prev) s_dprev = cv; }
            best = argmint_wave_fast(best);
            if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; s_t[wave] = best.t; }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_scan += t1 - t0; t0 = t1; }
            if (wave == 0) {
                ArgMinT m = {lane < 16 ? s_v[lane] : __builtin_inf(), lane < 16 ? s_i[lane] : 0x7fffffff, lane < 16 ? s_t[lane] : 0};
                m = argmint_row16_fast(m);                               // the 16 wave results sit in row 0
                if (lane == 0) {
                    smask[x >> 5] |= xbit;                          // un-mask the row's own column
                    if (m.i < 0 || m.i >= n) { s_stop = NN_STOP_GUARD; stop_code = NN_STOP_GUARD; }   // NaN distances: nothing compares
                    else {
                        nnidx[x] = (uint16_t)m.i;
                        if (m.t) tieb[x >> 5] |= (1u << (x & 31)); else tieb[x >> 5] &= ~(1u << (x & 31));
                        w.nnval[x] = m.v;
                        // SciPy: the previous chain element wins unless something is STRICTLY closer
                        int y = m.i;
                        if (prev >= 0 && !(m.v < s_dprev)) y = prev;
                        if (y != prev) {
                            if (++guard > 4 * n + 8) { s_stop = NN_STOP_GUARD; stop_code = NN_STOP_GUARD; }
                            chain[len] = y; ring[len & 255] = y;
                            if (len - 255 > ring_lo) ring_lo = len - 255;
                            second = top; top = y; len++;
                        } else {
                            // reciprocal by the tie rule although the cache names another column: let the walk see it
                            nnidx[x] = (uint16_t)prev;
                        }
                    }
                }
            }
            __syncthreads();
            if (PROFILE && tid == 0) { t1 = wall_clock64(); t_pick += t1 - t0; t0 = t1; }
            if (s_stop) { stopped = true; break; }
        }
        if (stopped || s_act == 3) break;
        // ---- merge (top, second)
        if (tid == 0) {
            int xx = top, yy = second;
            len -= 2;
            if (xx > yy) { int t = xx; xx = yy; yy = t; }
            int nx = lsize[xx], ny = lsize[yy];
            zraw[4 * (int64_t)step + 0] = (double)xx;
            zraw[4 * (int64_t)step + 1] = (double)yy;
            zraw[4 * (int64_t)step + 3] = (double)(nx + ny);
            lsize[xx] = 0;
            lsize[yy] = (uint16_t)(nx + ny);
            alive[xx >> 5] &= ~(1u << (xx & 31));
            smask[xx >> 5] &= ~(1u << (xx & 31));
            s_mx = xx; s_my = yy; s_nx = nx; s_ny = ny; s_tx = -1; s_ty = -1; s_ey = -1;
            top = len >= 1 ? (len - 1 >= ring_lo ? ring[(len - 1) & 255] : chain[len - 1]) : -1;
            second = len >= 2 ? (len - 2 >= ring_lo ? ring[(len - 2) & 255] : chain[len - 2]) : -1;
        }
        __syncthreads();
        const int mx = s_mx, my = s_my;
        if (tid < D) {
            if (dslot[tid] == mx) s_tx = dtime[tid];
            if (dslot[tid] == my) { s_ty = dtime[tid]; s_ey = tid; }
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_merge += t1 - t0; t0 = t1; }
        {
            const int tmx = s_tx, tmy = s_ty;
            const double fx = (double)s_nx, fy = (double)s_ny, fs = (double)(s_nx + s_ny);
            const double rcp = 1.0 / fs;
            const double* __restrict__ rx = W + (int64_t)mx * ld;
            double* __restrict__ ry = W + (int64_t)my * ld;
            // Every gathered value is LOADED here and USED after the streaming pass: a use in front of the loop (a store of
            // the height, a compare of a dirty partner's new distance) makes its wave wait for the gather before it even
            // issues its streaming loads - one more memory round trip on the critical path of every merge.
            // the merge height d(x, y): the row of whichever cluster merged last is the authoritative one (column mx of
            // row my is not rewritten below: mx is dead)
            double height = 0.0;
            if (tid == NN_THREADS - 1) height = tmx > tmy ? rx[my] : ry[mx];
            ArgMinT rbest = {__builtin_inf(), 0x7fffffff, 0};   // minimum of the new row: the merged cluster's own cache entry
            double dxi = 0.0, dyi = 0.0, nvd = 0.0; int dd = -1;
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != my && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    dxi = dtime[tid] > tmx ? W[(int64_t)d * ld + mx] : rx[d];
                    dyi = dtime[tid] > tmy ? W[(int64_t)d * ld + my] : ry[d];
                    nvd = nnval_ld[d];
                    dd = d;
                }
            }
#pragma unroll 4
            for (int j = tid * 2; j < n; j += 2 * NN_THREADS) {
                double2 a = *reinterpret_cast<const double2*>(rx + j);
                double2 b = *reinterpret_cast<const double2*>(ry + j);
                const double2 nv = *reinterpret_cast<const double2*>(nnval_ld + j);
                const uint32_t ip = *reinterpret_cast<const uint32_t*>(nnidx + j);
                uint32_t bits = smask[j >> 5] >> (j & 31);
                if ((bits & 1u) && j != my) {
                    b.x = div_by_small_int(fx * a.x + fy * b.x, fs, rcp);
                    if (b.x <= rbest.v) { if (b.x < rbest.v) { rbest.v = b.x; rbest.i = j; rbest.t = 0; } else rbest.t = 1; }
                    const uint32_t id = ip & 0xffffu;
                    if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[j] = (uint16_t)NN_NOIDX;
                    else if (id != NN_NOIDX && b.x <= nv.x) {
                        if (b.x < nv.x) nnidx[j] = (uint16_t)NN_NOIDX;
                        else { if ((uint32_t)my < id) nnidx[j] = (uint16_t)my; atomicOr(&tieb[j >> 5], 1u << (j & 31)); }
                    }
                }
                if ((bits & 2u) && j + 1 != my) {
                    b.y = div_by_small_int(fx * a.y + fy * b.y, fs, rcp);
                    if (b.y <= rbest.v) { if (b.y < rbest.v) { rbest.v = b.y; rbest.i = j + 1; rbest.t = 0; } else rbest.t = 1; }
                    const uint32_t id = ip >> 16;
                    if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[j + 1] = (uint16_t)NN_NOIDX;
                    else if (id != NN_NOIDX && b.y <= nv.y) {
                        if (b.y < nv.y) nnidx[j + 1] = (uint16_t)NN_NOIDX;
                        else { if ((uint32_t)my < id) nnidx[j + 1] = (uint16_t)my; atomicOr(&tieb[(j + 1) >> 5], 1u << ((j + 1) & 31)); }
                    }
                }
                *reinterpret_cast<double2*>(ry + j) = b;
            }
            double dv = 0.0;
            if (dd >= 0) {                                     // the dirty partner: its new distance, its cache entry
                dv = div_by_small_int(fx * dxi + fy * dyi, fs, rcp);
                const uint32_t id = nnidx[dd];
                if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[dd] = (uint16_t)NN_NOIDX;
                else if (id != NN_NOIDX && dv <= nvd) {
                    if (dv < nvd) nnidx[dd] = (uint16_t)NN_NOIDX;
                    else { if ((uint32_t)my < id) nnidx[dd] = (uint16_t)my; atomicOr(&tieb[dd >> 5], 1u << (dd & 31)); }
                }
                rbest = argmint_join(rbest, dv, dd, 0);
            }
            if (tid == NN_THREADS - 1) zraw[4 * (int64_t)step + 2] = height;
            rbest = argmint_wave_fast(rbest);
            if (lane == 0) { s_v[wave] = rbest.v; s_i[wave] = rbest.i; s_t[wave] = rbest.t; }
            __syncthreads();
            if (dd >= 0) ry[dd] = dv;
            if (wave == 0) {
                ArgMinT m = {lane < 16 ? s_v[lane] : __builtin_inf(), lane < 16 ? s_i[lane] : 0x7fffffff, lane < 16 ? s_t[lane] : 0};
                m = argmint_row16_fast(m);
                if (lane == 0) {                             // cluster y is dirty from now on; its neighbour is known
                    if (s_ey >= 0) dslot[s_ey] = -1;         // its older entry is superseded
                    dslot[D] = my; dtime[D] = step;
                    smask[my >> 5] &= ~(1u << (my & 31));
                    if (m.i >= 0 && m.i < n) {
                        nnidx[my] = (uint16_t)m.i; w.nnval[my] = m.v;
                        if (m.t) tieb[my >> 5] |= (1u << (my & 31)); else tieb[my >> 5] &= ~(1u << (my & 31));
                    } else nnidx[my] = (uint16_t)NN_NOIDX;
                }
            }
            D++;
        }
        __syncthreads();
        if (PROFILE && tid == 0) { t1 = wall_clock64(); t_upd += t1 - t0; }
    }
    // ---- save state for the flush kernel and the next epoch
    __syncthreads();
    for (int i = tid; i < nwords; i += NN_THREADS) w.alive[i] = alive[i];
    for (int i = tid; i < n; i += NN_THREADS) {
        w.size[i] = lsize[i];
        w.nnc[i] = (uint32_t)nnidx[i] | (((tieb[i >> 5] >> (i & 31)) & 1u) << 16);
    }
    if (tid < D) {
        w.dslot[tid] = dslot[tid]; w.dtime[tid] = dtime[tid];
        if (dslot[tid] >= 0) w.gtime[dslot[tid]] = dtime[tid];
    }
    if (tid == 0) {
        w.state[0] = step; w.state[1] = len; w.state[2] = top; w.state[3] = second; w.state[4] = first_ptr;
        w.state[5] = stop_code; w.state[6] = D;
        w.prof[5] += c_cols; w.prof[6] += c_scans; w.prof[7] += c_hits;
        if (PROFILE) { w.prof[0] += t_book; w.prof[1] += t_scan; w.prof[2] += t_pick; w.prof[3] += t_merge; w.prof[4] += t_upd; }
    }
}

// ---- the same chain on several workgroups: protocol and helpers ------------------------------------------
// (k_nn_epoch_mw, the cache-less sliced kernel of round 1 this text was written for, is gone: k_nn_epoch_mwc and
//  k_nn_epoch_w1 below keep its protocol.)
// A lone CU streams a row at ~77 GB/s: at 16k bins a 128 KB scan is two thirds transfer, one third latency.
// Here NWG workgroups (one CU each, any XCD) run the SAME chain as replicated state machines: each keeps the
// full LDS state (liveness, sizes, dirty list, chain) and takes every decision itself, but streams only ITS
// SLICE of the columns of a row - in scans and in Lance-Williams updates.  The one thing a decision needs from
// the others is their slice's (min, index): ONE exchange per scan through 16-byte mailbox slots
// {value, index, sequence number} written and polled with sc1 (write-through / L1-bypassing) accesses, no
// fences (MI355X_MICROARCH.md, "Valid forms": sc1 stores, every storing wave drained behind a workgroup
// barrier, one lane signals, the polling wave loads after its poll, the others after a barrier).  Slots are
// double-buffered by the parity of the sequence number: a workgroup can be at most one exchange ahead.
// Visibility of matrix bytes: a workgroup plainly loads only columns of its own slice, which only it ever
// writes (sc1 stores); every element that may lie in another slice - dirty partners' W[d][x], d(x, prev) - is
// read with an sc1 load, and was written before the writer's previous exchange.  The single element that is
// needed BEFORE an exchange has happened - W[y'][z] for the cluster y' merged a moment ago and the row z the
// next scan visits - is handed from the owner of column z (which computes it in its update) to the owner of
// column y' (the only reader: it lists y' among its dirty candidates) through a tagged 16-byte slot of its own.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// (s_nop: a vector-memory store of more than 64 bits must not be followed at once by a vector instruction that overwrites
//  its data registers; the compiler pads its own stores, but it does not look inside inline assembly - seen as wrong
//  bytes in the stored row when the next instruction was an address computation into the same registers)
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// plain forms for workgroups that share an XCD (k_nn_epoch_w1, LOCAL): the line stays in that XCD's L2, where the peers'
// sc1 loads find it - a hop of 0.28 us instead of 0.44, a 64-party exchange round of 0.70 us instead of 1.45
__device__ __forceinline__ void st16_l2(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st8_l2(double* p, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x4 ld16_sc1(const void* p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ double ld8_sc1(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st8_sc1(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Streaming loads of the workgroup's own slice are sc1 too: plain (and nontemporal) re-loads of a line this
// workgroup had itself rewritten with sc1 stores returned pre-update values now and then (observed on gfx950;
// with every load sc1 the chain is bit-exact and repeatable).  Issued in pairs from inline assembly - the
// compiler's wait-count pass does not see them - and drained by one s_waitcnt that also "produces" the registers,
// so no use can be scheduled ahead of it.
#define NN_LD16_SC1(reg, ptr) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(reg) : "v"(ptr) : "memory")
#define NN_DRAIN2(a, b) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b)::"memory")
#define NN_DRAIN4(a, b, c, d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
__device__ __forceinline__ double2 mw_pair(u32x4 r)
{
    double2 v;
    v.x = __hiloint2double((int)r.y, (int)r.x); v.y = __hiloint2double((int)r.w, (int)r.z);
    return v;
}

// A 16-byte sc1 store may land as two 8-byte halves, so each half carries the sequence number itself:
//   {value bits 31..0, seq} {value bits 63..32, index | (seq & 0x7fff) << 17}     (index < 2^17; 0x1ffff = none)
__device__ __forceinline__ double mw_value(u32x4 p) { return __hiloint2double((int)p.z, (int)p.x); }

// Every replica of k_nn_epoch_mwc decides every merge itself from values that crossed workgroups through sc1 accesses
// (measured behaviour, not an architectural guarantee: MI355X_MICROARCH.md).  A stale read would show up as replicas
// that disagree - so their records of the epoch are compared, and a difference stops the chain with an error
// instead of returning a wrong tree.
__global__ __launch_bounds__(256) void k_nn_check_replicas(NNWorkspace w, int nwg)
{
    if (w.state[5]) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int count = w.state[0] - w.state[9];
    if (k >= count || k >= NN_DMAX) return;
    const unsigned long long* r0 = reinterpret_cast<const unsigned long long*>(w.rec + (int64_t)k * 4);
    bool same = true;
    for (int g = 1; g < nwg; g++) {
        const unsigned long long* rg = reinterpret_cast<const unsigned long long*>(w.rec + ((int64_t)g * NN_DMAX + k) * 4);
        same = same && rg[0] == r0[0] && rg[1] == r0[1] && rg[2] == r0[2] && rg[3] == r0[3];
    }
    if (!same) atomicCAS(&w.state[5], 0, NN_STOP_DIVERGED);
}

// ---- column slices AND the neighbour cache: k_nn_epoch_mwc ---------------------------------------------------------
// A lone CU ingests a row at ~55-77 GB/s whatever the loop looks like (the limit is the CU's outstanding misses), so
// with the scans the cache leaves (~1.3 per merge) and the update, a merge on one workgroup still streams ~3.7 rows.
// Here NWG workgroups are replicas of ONE state machine, as in k_nn_epoch_mw: each streams its column slice only.
// On top of that:
//  * the neighbour cache (nnidx, tie flags) is replicated in every workgroup's LDS and kept identical: scan results
//    and the merged row's minimum arrive through the exchanges; "neighbour merged -> unknown" is applied by every
//    replica to all rows; the one change only a slice's owner can see - a merged cluster landing AT OR BELOW a row's
//    cached minimum - is signalled by a flag in the owner's message and answered by every replica dropping the cache
//    entries of that whole slice (rare: exact ties or an fp-rounding fluke; costs scans, never correctness).
//    Cached distances (nnval) are private to the owner of the column: nobody else reads them.
//  * the scan that follows almost every merge - the new chain top `a`, whose neighbour was one of the merged pair -
//    is FUSED into the update pass: the pass streams rows x, y and a together; d(a, y') is the element of the new row
//    that the owner of column a computes anyway, so the hand-off slot of k_nn_epoch_mw is not needed, and one
//    exchange carries both the merged row's minimum and row a's.
// Per merge: ~1.3 passes over a slice and ~1.3 exchanges (k_nn_epoch_mw: 3.9 passes, 2.9 exchanges).
#define NN_LD16(reg, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(reg) : "v"(ptr) : "memory")
#define NN_DRAIN8(a, b, c, d, e, f, g, h) \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"memory")

#define NN_LD8_SC1(reg, ptr) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(reg) : "v"(ptr) : "memory")
#define NN_LD8(reg, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(reg) : "v"(ptr) : "memory")
#define NN_DRAIN6(a, b, c, d, e, f) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)::"memory")
__device__ __forceinline__ double nn_f64(unsigned long long bits) { return __longlong_as_double((long long)bits); }

//   {value bits 31..0, seq} {value bits 63..32, index (17 bits) | tie << 17 | event << 18 | (seq & 0x1fff) << 19}
__device__ __forceinline__ u32x4 mwc_pack(double v, int idx, int tie, int ev, unsigned int seq)
{
    u32x4 p;
    p.x = (unsigned int)__double2loint(v); p.y = seq;
    p.z = (unsigned int)__double2hiint(v);
    p.w = ((unsigned int)((idx >= 0 && idx < 0x1ffff) ? idx : 0x1ffff)) | ((unsigned int)(tie ? 1 : 0) << 17) |
          ((unsigned int)(ev ? 1 : 0) << 18) | ((seq & 0x1fffu) << 19);
    return p;
}
__device__ __forceinline__ bool mwc_ready(u32x4 p, unsigned int seq) { return p.y == seq && (p.w >> 19) == (seq & 0x1fffu); }
__device__ __forceinline__ int mwc_index(u32x4 p) { const int i = (int)(p.w & 0x1ffffu); return i == 0x1ffff ? 0x7fffffff : i; }
__device__ __forceinline__ int mwc_tie(u32x4 p) { return (int)((p.w >> 17) & 1u); }
__device__ __forceinline__ int mwc_event(u32x4 p) { return (int)((p.w >> 18) & 1u); }

static constexpr int NN_MWC_MAX = 32768;                 // sizes + cache of every column in the LDS of every replica
static constexpr int NN_MWC_GMAX = 65535;                // GSIZE: the cache alone in LDS (2 bytes per column; 0xffff = unknown),
                                                         // the sizes in a private global array per replica - as far as the
                                                         // 160 KB go (launch_nnchain checks: 64,000 columns fit)

// GSIZE: cluster sizes in global memory (w.size_rep, one copy per workgroup, read and written by lane 0 with L1-bypassing
// accesses: two loads per merge on the critical path, ~0.7 us where a merge costs ~13) instead of in LDS.
template <int NWG, bool PROF, bool GSIZE = false>
__global__ __launch_bounds__(NN_THREADS) void k_nn_epoch_mwc(double* __restrict__ W, int64_t ld, int n,
                                                             int* __restrict__ chain_all, double* __restrict__ zraw,
                                                             NNWorkspace w, int dcap, int total_steps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    const int nwords = (n + 31) >> 5, nw4 = (nwords + 3) & ~3, n2 = (n + 7) & ~7;
    uint32_t* alive = reinterpret_cast<uint32_t*>(smem_nn);
    uint32_t* smask = alive + nw4;
    uint32_t* tieb = smask + nw4;
    uint16_t* lsize = reinterpret_cast<uint16_t*>(tieb + nw4);
    uint16_t* nnidx = GSIZE ? lsize : lsize + n2;
    __shared__ int dslot[NN_DMAX], dtime[NN_DMAX];
    __shared__ double s_v[32];
    __shared__ int s_i[32], s_t[32];
    __shared__ int ring[256];
    __shared__ double s_dprev, s_my_v, s_a_v;
    __shared__ int s_x, s_prev, s_act, s_stop, s_mx, s_my, s_nx, s_ny, s_tx, s_ty, s_ey, s_a, s_ta, s_ev, s_evmask, s_my_i, s_my_t,
        s_a_i, s_a_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    int* __restrict__ gsize = w.size_rep + (int64_t)wg * (((int64_t)n + 3) & ~(int64_t)3);   // (GSIZE only)
    int* __restrict__ chain = chain_all + (int64_t)wg * (n + 2);           // every workgroup keeps its own copy
    u32x4* mail = reinterpret_cast<u32x4*>(w.mail);
    int step = w.state[0];
    if (step >= total_steps || w.state[5]) {
        if (tid == 0 && wg == 0) w.state[6] = 0;
        return;
    }
    // column slice of this workgroup: multiples of 64 so that mask words and 16-byte loads never straddle
    const int slice = (((n + NWG - 1) / NWG) + 63) & ~63;
    const int c0 = wg * slice < n ? wg * slice : n;
    const int c1 = c0 + slice < n ? c0 + slice : n;
    for (int i = tid; i < nwords; i += NN_THREADS) { alive[i] = w.alive[i]; smask[i] = w.alive[i]; tieb[i] = 0u; }
    __syncthreads();
    for (int i = tid; i < n; i += NN_THREADS) {
        if (GSIZE) __hip_atomic_store(gsize + i, (int)w.size[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else lsize[i] = w.size[i];
        const uint32_t c = w.nnc[i];
        nnidx[i] = (uint16_t)(c & 0xffffu);
        if (c >> 16) atomicOr(&tieb[i >> 5], 1u << (i & 31));
    }
    if (GSIZE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int len = w.state[1], top = w.state[2], second = w.state[3], first_ptr = w.state[4];
    if (tid == 0) { s_stop = 0; s_ev = 0; }
    const int* __restrict__ chain0 = chain_all;             // chain prefix of the earlier epochs: workgroup 0's copy
    // the top of the saved chain goes into the LDS ring right away: pops must not wait for global memory
    int ring_lo = len > 256 ? len - 256 : 0;
    if (tid < 256 && ring_lo + tid < len) ring[(ring_lo + tid) & 255] = chain0[ring_lo + tid];
    __syncthreads();
    int D = 0;
    uint32_t xbit = 0u;
    unsigned int xseq = 0u;                                  // exchanges so far (uniform)
    __shared__ unsigned long long s_tp[16];                  // profile: [0..4] phases, [8..14] detail of the fused pass (lane 0 only)
    unsigned long long t0 = 0;
    if (tid < 16) s_tp[tid] = 0;
    const bool prof = PROF && wg == 0 && tid == 0;       // the time stamps cost ~26 registers: a separate instantiation
    int lowmark = len;                                       // lane 0: chain entries below this are still the earlier epochs'
    const int step0 = step;
    const int inject_late = w.state[10], inject_wrong = w.state[11];      // test hooks (0 = off)
    __shared__ unsigned long long s_cnt[3];                  // lane 0: columns visited by scans, scans, cache hits
    if (tid == 0) { s_cnt[0] = 0; s_cnt[1] = 0; s_cnt[2] = 0; }
    if (tid == 0 && wg == 0) w.state[9] = step0;
    int guard = 0, stop_code = 0;                            // lane 0
    bool stopped = false;                                    // uniform

    for (; step < total_steps && D < dcap; step++) {
        if (prof) t0 = wall_clock64();
        while (true) {
            // ---- walk the chain on cached neighbours (lane 0 of every replica, identical state -> identical walk)
            if (tid == 0) {
                if (len == 0) {
                    while (first_ptr < n && !((alive[first_ptr >> 5] >> (first_ptr & 31)) & 1u)) first_ptr++;
                    chain[0] = first_ptr; ring[0] = first_ptr; ring_lo = 0; top = first_ptr; second = -1; len = 1; lowmark = 0;
                }
                int act = stop_code ? 3 : 0;                 // 1: scan row `top`, 2: merge (top, second), 3: stop
                while (act == 0) {
                    const int x = top, prev = (len > 1) ? second : -1;
                    const uint32_t idx = nnidx[x];
                    if (idx == NN_NOIDX) act = 1;
                    else if ((int)idx == prev) act = 2;
                    else if (prev >= 0 && ((tieb[x >> 5] >> (x & 31)) & 1u)) act = 1;
                    else if (++guard > 4 * n + 8) { stop_code = NN_STOP_GUARD; act = 3; }
                    else {
                        chain[len] = (int)idx; ring[len & 255] = (int)idx;
                        if (len - 255 > ring_lo) ring_lo = len - 255;
                        second = top; top = (int)idx; len++;
                        s_cnt[2]++;
                    }
                }
                s_act = act; s_x = top; s_prev = (len > 1) ? second : -1; s_tx = -1;
                if (act == 1) {
                    xbit = smask[top >> 5] & (1u << (top & 31));
                    atomicAnd(&smask[top >> 5], ~xbit);
                    s_cnt[1]++; s_cnt[0] += (unsigned long long)(total_steps + 1 - step);
                }
            }
            __syncthreads();
            if (s_act != 1) break;
            // ---- a scan on its own: this slice of row x, then one exchange
            const int x = s_x, prev = s_prev;
            if (tid < D && dslot[tid] == x) s_tx = dtime[tid];
            __syncthreads();
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[0] += t1 - t0; t0 = t1; }
            const int tx = s_tx;
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            // every load is ISSUED before anything waits: the gathered ones (d(x, prev), the dirty partners' values from
            // whichever row is the authoritative one) and the streamed slice are in flight together
            unsigned long long r_dp = 0, r_cv = 0;
            const bool want_dp = tid == 64 && prev >= 0 && ((smask[prev >> 5] >> (prev & 31)) & 1u);
            if (want_dp) NN_LD8_SC1(r_dp, rowx + prev);
            int cd = -1; bool cmine = false;
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != x && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    cmine = d >= c0 && d < c1;
                    if (cmine || d == prev) {
                        cd = d;
                        const double* src = dtime[tid] > tx ? W + (int64_t)d * ld + x : rowx + d;
                        NN_LD8_SC1(r_cv, src);
                    }
                }
            }
            ArgMinT best = {__builtin_inf(), 0x7fffffff, 0};
            ArgMinT cand = {__builtin_inf(), 0x7fffffff, 0};
            for (int j0 = c0 + tid * 2; j0 < c1; j0 += 4 * NN_THREADS) {    // two 16-byte loads in flight per lane
                const int j1 = j0 + 2 * NN_THREADS;
                const bool two = j1 < c1;
                u32x4 r0, r1;
                NN_LD16_SC1(r0, rowx + j0);
                NN_LD16_SC1(r1, rowx + (two ? j1 : j0));
                NN_DRAIN2(r0, r1);
                {
                    const double2 v = mw_pair(r0);
                    const uint32_t bits = smask[j0 >> 5] >> (j0 & 31);
                    if ((bits & 1u) && v.x <= best.v) { if (v.x < best.v) { best.v = v.x; best.i = j0; best.t = 0; } else best.t = 1; }
                    if ((bits & 2u) && v.y <= best.v) { if (v.y < best.v) { best.v = v.y; best.i = j0 + 1; best.t = 0; } else best.t = 1; }
                }
                if (two) {
                    const double2 v = mw_pair(r1);
                    const uint32_t bits = smask[j1 >> 5] >> (j1 & 31);
                    if ((bits & 1u) && v.x <= best.v) { if (v.x < best.v) { best.v = v.x; best.i = j1; best.t = 0; } else best.t = 1; }
                    if ((bits & 2u) && v.y <= best.v) { if (v.y < best.v) { best.v = v.y; best.i = j1 + 1; best.t = 0; } else best.t = 1; }
                }
            }
            NN_DRAIN2(r_dp, r_cv);
            if (want_dp) s_dprev = nn_f64(r_dp);
            if (cd >= 0) {
                const double v = nn_f64(r_cv);
                if (cmine) { cand.v = v; cand.i = cd; }
                if (cd == prev) s_dprev = v;
            }
            if (cand.i != 0x7fffffff) best = argmint_join(best, cand.v, cand.i, 0);
            best = argmint_wave_fast(best);
            if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; s_t[wave] = best.t; }
            __syncthreads();
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[1] += t1 - t0; t0 = t1; }
            xseq++;
            if (wave == 0) {
                ArgMinT m = {lane < 16 ? s_v[lane] : __builtin_inf(), lane < 16 ? s_i[lane] : 0x7fffffff, lane < 16 ? s_t[lane] : 0};
                m = argmint_row16_fast(m);                               // this slice's result, in every lane of row 0
                u32x4* slots = mail + (xseq & 1u) * (NN_MAXWG * 2);
                if (lane == 0) st16_sc1(slots + wg * 2, mwc_pack(m.v, m.i, m.t, 0, xseq));
                ArgMinT o = {__builtin_inf(), 0x7fffffff, 0};
                int late = 0;
                if (lane < NWG) {
                    if (lane == wg) o = m;
                    else {
                        u32x4 r = ld16_sc1(slots + lane * 2);
                        int budget = 1000000;
                        while (!mwc_ready(r, xseq) && --budget > 0) { __builtin_amdgcn_s_sleep(1); r = ld16_sc1(slots + lane * 2); }
                        if (!mwc_ready(r, xseq)) late = 1;
                        o.v = mw_value(r); o.i = mwc_index(r); o.t = mwc_tie(r);
                    }
                }
                if (inject_late > 0 && (int)xseq == inject_late) late = 1;      // test hook: a peer that never answers
                late = __any(late);
                m = argmint_row16_fast(o);
                if (lane == 0) {
                    atomicOr(&smask[x >> 5], xbit);
                    if (late) { s_stop = NN_STOP_LATE; stop_code = NN_STOP_LATE; }
                    else if (m.i < 0 || m.i >= n) { s_stop = NN_STOP_GUARD; stop_code = NN_STOP_GUARD; }
                    else {
                        nnidx[x] = (uint16_t)m.i;
                        if (m.t) atomicOr(&tieb[x >> 5], 1u << (x & 31)); else atomicAnd(&tieb[x >> 5], ~(1u << (x & 31)));
                        if (x >= c0 && x < c1) w.nnval[x] = m.v;        // the owner of column x keeps the distance
                        int y = m.i;
                        if (prev >= 0 && !(m.v < s_dprev)) y = prev;    // SciPy: the previous element wins unless STRICTLY closer
                        if (y != prev) {
                            if (++guard > 4 * n + 8) { s_stop = NN_STOP_GUARD; stop_code = NN_STOP_GUARD; }
                            chain[len] = y; ring[len & 255] = y;
                            if (len - 255 > ring_lo) ring_lo = len - 255;
                            second = top; top = y; len++;
                        } else nnidx[x] = (uint16_t)prev;               // reciprocal by the tie rule: let the walk see it
                    }
                }
            }
            __syncthreads();
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[2] += t1 - t0; t0 = t1; }
            if (s_stop) { stopped = true; break; }
        }
        if (stopped || s_act == 3) break;
        // ---- merge (top, second); the row the chain returns to - `a` - is scanned in the same pass if it needs it
        if (tid == 0) {
            int xx = top, yy = second;
            len -= 2;
            if (xx > yy) { int t = xx; xx = yy; yy = t; }
            int nx, ny;
            if (GSIZE) {
                nx = __hip_atomic_load(gsize + xx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ny = __hip_atomic_load(gsize + yy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gsize + xx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gsize + yy, nx + ny, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                nx = lsize[xx]; ny = lsize[yy];
                lsize[xx] = 0;
                lsize[yy] = (uint16_t)(nx + ny);
            }
            atomicAnd(&alive[xx >> 5], ~(1u << (xx & 31)));
            atomicAnd(&smask[xx >> 5], ~(1u << (xx & 31)));
            s_mx = xx; s_my = yy; s_nx = nx; s_ny = ny; s_tx = -1; s_ty = -1; s_ey = -1; s_ta = -1;
            // entries no push of this epoch has overwritten live in workgroup 0's copy (saved by the last epoch)
            if (len < lowmark) lowmark = len;
            const int i1 = len - 1, i2 = len - 2;
            top = len >= 1 ? (i1 >= ring_lo ? ring[i1 & 255] : (i1 < lowmark ? chain0[i1] : chain[i1])) : -1;
            second = len >= 2 ? (i2 >= ring_lo ? ring[i2 & 255] : (i2 < lowmark ? chain0[i2] : chain[i2])) : -1;
            int a = -1;
            if (top >= 0) {
                const uint32_t ia = nnidx[top];
                if (ia == NN_NOIDX || (int)ia == xx || (int)ia == yy) a = top;
            }
            s_a = a; s_prev = (a >= 0 && len > 1) ? second : -1;
            if (a >= 0) { s_cnt[1]++; s_cnt[0] += (unsigned long long)(total_steps - step); }
        }
        __syncthreads();
        const int mx = s_mx, my = s_my, a = s_a, aprev = s_prev;
        // lane 0, once the exchange has delivered the merged row's minimum and row a's: cluster y is dirty from now on and its
        // neighbour is known; the fused scan of row a is decided exactly like a scan on its own
        auto after_exchange = [&]() {
            if (s_ey >= 0) dslot[s_ey] = -1;
            dslot[D] = my; dtime[D] = step;
            atomicAnd(&smask[my >> 5], ~(1u << (my & 31)));
            if (s_my_i >= 0 && s_my_i < n) {
                nnidx[my] = (uint16_t)s_my_i;
                if (s_my_t) atomicOr(&tieb[my >> 5], 1u << (my & 31)); else atomicAnd(&tieb[my >> 5], ~(1u << (my & 31)));
                if (my >= c0 && my < c1) w.nnval[my] = s_my_v;
            } else nnidx[my] = (uint16_t)NN_NOIDX;
            if (a >= 0) {
                if (s_a_i < 0 || s_a_i >= n) stop_code = NN_STOP_GUARD;
                else {
                    nnidx[a] = (uint16_t)s_a_i;
                    if (s_a_t) atomicOr(&tieb[a >> 5], 1u << (a & 31)); else atomicAnd(&tieb[a >> 5], ~(1u << (a & 31)));
                    if (a >= c0 && a < c1) w.nnval[a] = s_a_v;
                    int y = s_a_i;
                    if (aprev >= 0 && !(s_a_v < s_dprev)) y = aprev;
                    if (y != aprev) {
                        if (++guard > 4 * n + 8) stop_code = NN_STOP_GUARD;
                        chain[len] = y; ring[len & 255] = y;
                        if (len - 255 > ring_lo) ring_lo = len - 255;
                        second = top; top = y; len++;
                    } else nnidx[a] = (uint16_t)aprev;
                }
            }
        };
        if (tid < D) {
            if (dslot[tid] == mx) s_tx = dtime[tid];
            if (dslot[tid] == my) { s_ty = dtime[tid]; s_ey = tid; }
            if (dslot[tid] == a) s_ta = dtime[tid];
        }
        __syncthreads();
        if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[3] += t1 - t0; t0 = t1; }
        {
            const int tmx = s_tx, tmy = s_ty, ta = s_ta;
            const double fx = (double)s_nx, fy = (double)s_ny, fs = (double)(s_nx + s_ny);
            const double rcp = 1.0 / fs;
            const double* __restrict__ rx = W + (int64_t)mx * ld;
            double* __restrict__ ry = W + (int64_t)my * ld;
            const double* __restrict__ ra = W + (int64_t)(a >= 0 ? a : mx) * ld;
            // ---- issue every gathered load (nothing waits yet)
            unsigned long long r_h = 0, r_dp = 0, r_dxi = 0, r_dyi = 0, r_nvd = 0, r_av = 0;
            // the merge height d(x, y): the row of whichever cluster merged last is the authoritative one
            if (tid == NN_THREADS - 1) { const double* src = tmx > tmy ? rx + my : ry + mx; NN_LD8_SC1(r_h, src); }
            const bool want_dp = tid == 64 && aprev >= 0 && ((smask[aprev >> 5] >> (aprev & 31)) & 1u);
            if (want_dp) NN_LD8_SC1(r_dp, ra + aprev);
            int dd = -1, ad = -1;                                // dirty partner this lane updates / offers to row a's scan
            bool amine = false;
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != my && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    const bool mine = d >= c0 && d < c1;
                    if (mine) {                                      // columns of this slice only: rx[d], ry[d] are its own
                        dd = d;
                        const double* sx = dtime[tid] > tmx ? W + (int64_t)d * ld + mx : rx + d;
                        const double* sy = dtime[tid] > tmy ? W + (int64_t)d * ld + my : ry + d;
                        NN_LD8_SC1(r_dxi, sx);
                        NN_LD8_SC1(r_dyi, sy);
                        NN_LD8(r_nvd, w.nnval + d);
                    }
                    if (a >= 0 && d != a && (mine || d == aprev)) {
                        ad = d; amine = mine;
                        const double* sa = dtime[tid] > ta ? W + (int64_t)d * ld + a : ra + d;
                        NN_LD8_SC1(r_av, sa);
                    }
                }
            }
            ArgMinT rbest = {__builtin_inf(), 0x7fffffff, 0};   // this slice of the new row: the merged cluster's cache entry
            ArgMinT abest = {__builtin_inf(), 0x7fffffff, 0};   // this slice of row a: the streamed columns (ascending per lane)
            ArgMinT acand = {__builtin_inf(), 0x7fffffff, 0};   // ... and its gathered candidates (dirty partners, the new cluster)
            int ev = 0;                                         // a cached minimum of this slice was reached or undercut
            bool first = true;
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[8] += t1 - t0; t0 = t1; }
            // one 16-byte pair per lane and trip (x, y, a and the cached distances: four loads in flight); two pairs at once
            // would not fit the 128 registers a 1024-lane workgroup leaves a lane
            for (int j0 = c0 + tid * 2; j0 < c1 || first; j0 += 2 * NN_THREADS) {
                const bool any = j0 < c1;
                const int j = any ? j0 : 0;                      // lanes without a pair re-read column 0 (always in bounds)
                u32x4 qa, qb, qc, qn;
                NN_LD16_SC1(qa, rx + j);
                NN_LD16_SC1(qb, ry + j);
                NN_LD16_SC1(qc, ra + j);
                NN_LD16(qn, w.nnval + j);
                if (first) {
                    // while the loads are in flight: every replica drops "my neighbour is x or y" for the rows outside its
                    // slice (LDS only; the own slice is handled with the streamed / gathered values below)
                    // eight entries per 16-byte LDS read; a word can only matter if one of its halves equals x or y, which the
                    // zero-halfword test on (word XOR pattern) finds without unpacking (it never misses; a rare false hit only
                    // costs the exact compare)
                    const uint32_t px = (uint32_t)mx * 0x00010001u, py = (uint32_t)my * 0x00010001u;
                    for (int i = tid * 8; i < n; i += 8 * NN_THREADS) {
                        if (i >= c0 && i < c1) continue;                 // slices are multiples of 64 columns: never straddled
                        const uint4 v = *reinterpret_cast<const uint4*>(nnidx + i);
                        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t tx2 = wv[u] ^ px, ty2 = wv[u] ^ py;
                            if ((((tx2 - 0x00010001u) & ~tx2) | ((ty2 - 0x00010001u) & ~ty2)) & 0x80008000u) {
                                const uint32_t i0 = wv[u] & 0xffffu, i1 = wv[u] >> 16;
                                if (i0 == (uint32_t)mx || i0 == (uint32_t)my) nnidx[i + 2 * u] = (uint16_t)NN_NOIDX;
                                if (i1 == (uint32_t)mx || i1 == (uint32_t)my) nnidx[i + 2 * u + 1] = (uint16_t)NN_NOIDX;
                            }
                        }
                    }
                    if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[9] += t1 - t0; t0 = t1; }
                }
                NN_DRAIN4(qa, qb, qc, qn);
                if (prof && first) { const unsigned long long t1 = wall_clock64(); s_tp[10] += t1 - t0; t0 = t1; }
                first = false;
                if (!any) continue;
                const double2 xa = mw_pair(qa);
                double2 b = mw_pair(qb);
                const double2 va = mw_pair(qc);
                const double2 nv = mw_pair(qn);
                const uint32_t bits = smask[j >> 5] >> (j & 31);
                const uint32_t ip = *reinterpret_cast<const uint32_t*>(nnidx + j);
                const bool w0 = (bits & 1u) && j != my, w1 = (bits & 2u) && j + 1 != my;
                if (w0) {
                    b.x = div_by_small_int(fx * xa.x + fy * b.x, fs, rcp);
                    if (b.x <= rbest.v) { if (b.x < rbest.v) { rbest.v = b.x; rbest.i = j; rbest.t = 0; } else rbest.t = 1; }
                    const uint32_t id = ip & 0xffffu;
                    if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[j] = (uint16_t)NN_NOIDX;
                    else if (id != NN_NOIDX && b.x <= nv.x) ev = 1;
                    if (a >= 0) {
                        if (j == a) acand = argmint_join(acand, b.x, my, 0);          // d(a, y'), computed a moment ago
                        else if (va.x <= abest.v) { if (va.x < abest.v) { abest.v = va.x; abest.i = j; abest.t = 0; } else abest.t = 1; }
                    }
                }
                if (w1) {
                    b.y = div_by_small_int(fx * xa.y + fy * b.y, fs, rcp);
                    if (b.y <= rbest.v) { if (b.y < rbest.v) { rbest.v = b.y; rbest.i = j + 1; rbest.t = 0; } else rbest.t = 1; }
                    const uint32_t id = ip >> 16;
                    if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[j + 1] = (uint16_t)NN_NOIDX;
                    else if (id != NN_NOIDX && b.y <= nv.y) ev = 1;
                    if (a >= 0) {
                        if (j + 1 == a) acand = argmint_join(acand, b.y, my, 0);
                        else if (va.y <= abest.v) { if (va.y < abest.v) { abest.v = va.y; abest.i = j + 1; abest.t = 0; } else abest.t = 1; }
                    }
                }
                // a pair store must not carry the OLD value of a dirty column (its own 8-byte store, below, is not ordered
                // against this one): elements that were not recomputed are left alone
                if (w0 && w1) {
                    u32x4 pk;
                    pk.x = (unsigned int)__double2loint(b.x); pk.y = (unsigned int)__double2hiint(b.x);
                    pk.z = (unsigned int)__double2loint(b.y); pk.w = (unsigned int)__double2hiint(b.y);
                    st16_sc1(ry + j, pk);
                }
                else if (w0) st8_sc1(ry + j, b.x);
                else if (w1) st8_sc1(ry + j + 1, b.y);
            }
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[11] += t1 - t0; t0 = t1; }
            // ---- the gathered values: dirty partners of the update, candidates of row a's scan
            NN_DRAIN6(r_h, r_dp, r_dxi, r_dyi, r_nvd, r_av);
            if (want_dp) s_dprev = nn_f64(r_dp);
            if (dd >= 0) {
                const double dv = div_by_small_int(fx * nn_f64(r_dxi) + fy * nn_f64(r_dyi), fs, rcp);
                st8_sc1(ry + dd, dv);
                rbest = argmint_join(rbest, dv, dd, 0);
                const uint32_t id = nnidx[dd];
                if (id == (uint32_t)mx || id == (uint32_t)my) nnidx[dd] = (uint16_t)NN_NOIDX;
                else if (id != NN_NOIDX && dv <= nn_f64(r_nvd)) ev = 1;
                if (dd == a) acand = argmint_join(acand, dv, my, 0);        // d(a, y') for the scan of row a
            }
            if (ad >= 0) {
                const double v = nn_f64(r_av);
                if (amine) acand = argmint_join(acand, v, ad, 0);
                if (ad == aprev) s_dprev = v;
            }
            if (ev) s_ev = 1;
            if (acand.i != 0x7fffffff) abest = argmint_join(abest, acand.v, acand.i, acand.t);
            rbest = argmint_wave_fast(rbest);
            abest = argmint_wave_fast(abest);
            if (lane == 0) { s_v[wave] = rbest.v; s_i[wave] = rbest.i; s_t[wave] = rbest.t; s_v[16 + wave] = abest.v; s_i[16 + wave] = abest.i; s_t[16 + wave] = abest.t; }
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[12] += t1 - t0; t0 = t1; }
            if (tid == NN_THREADS - 1) {
                const double height = nn_f64(r_h);
                if (wg == 0) {
                    zraw[4 * (int64_t)step + 0] = (double)mx; zraw[4 * (int64_t)step + 1] = (double)my;
                    zraw[4 * (int64_t)step + 2] = height; zraw[4 * (int64_t)step + 3] = fs;
                }
                double* r = w.rec + ((int64_t)wg * NN_DMAX + (step - step0)) * 4;   // this replica's record of the merge
                r[0] = (double)mx; r[1] = (double)my; r[3] = fs;
                r[2] = (inject_wrong > 0 && wg == 1 && step == inject_wrong) ? height + 1.0 : height;
            }
            // the stores above are inline assembly the compiler's wait-count pass does not see: every storing wave drains
            // them before the barrier behind which the exchange signals
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[13] += t1 - t0; t0 = t1; }
            __syncthreads();
            if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[14] += t1 - t0; t0 = t1; }
            // ---- one exchange: the merged row's minimum (+ event flag) and row a's minimum
            xseq++;
            if (wave == 0) {
                ArgMinT m = {lane < 32 ? s_v[lane] : __builtin_inf(), lane < 32 ? s_i[lane] : 0x7fffffff, lane < 32 ? s_t[lane] : 0};
                m = argmint_row16_fast(m);                               // lanes 0-15: new row, lanes 16-31: row a (this slice)
                u32x4* slots = mail + (xseq & 1u) * (NN_MAXWG * 2);
                if (lane == 0) st16_sc1(slots + wg * 2, mwc_pack(m.v, m.i, m.t, s_ev, xseq));
                if (lane == 16) st16_sc1(slots + wg * 2 + 1, mwc_pack(m.v, m.i, m.t, 0, xseq));
                ArgMinT o = {__builtin_inf(), 0x7fffffff, 0};
                int late = 0, evbit = 0;
                const int peer = lane & 15, which = lane >> 4;      // which: 0 = new row, 1 = row a
                if (lane < 32 && peer < NWG) {
                    if (peer == wg) { o = m; evbit = which == 0 ? s_ev : 0; }
                    else {
                        u32x4 r = ld16_sc1(slots + peer * 2 + which);
                        int budget = 1000000;
                        while (!mwc_ready(r, xseq) && --budget > 0) { __builtin_amdgcn_s_sleep(1); r = ld16_sc1(slots + peer * 2 + which); }
                        if (!mwc_ready(r, xseq)) late = 1;
                        o.v = mw_value(r); o.i = mwc_index(r); o.t = mwc_tie(r);
                        evbit = which == 0 ? mwc_event(r) : 0;
                    }
                }
                if (inject_late > 0 && (int)xseq == inject_late) late = 1;
                late = __any(late);
                const unsigned long long evm = __ballot(evbit != 0);
                m = argmint_row16_fast(o);
                const double av = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(m.v), 16), __builtin_amdgcn_readlane(__double2loint(m.v), 16));
                const int ai = __builtin_amdgcn_readlane(m.i, 16), at = __builtin_amdgcn_readlane(m.t, 16);
                if (lane == 0) {
                    if (late) { s_stop = NN_STOP_LATE; stop_code = NN_STOP_LATE; }
                    s_evmask = (int)(evm & 0xffffull);
                    s_my_v = m.v; s_my_i = m.i; s_my_t = m.t;
                    s_a_v = av; s_a_i = ai; s_a_t = at;
                    s_ev = 0;
                    // the common case - nobody is late, no slice lost its cache - is finished right here by the lane that holds the
                    // results: no barrier between the exchange and the bookkeeping, none between the bookkeeping and the next walk
                    if (!late && (evm & 0xffffull) == 0ull) after_exchange();
                }
            }
            __syncthreads();
            if (s_stop) { stopped = true; break; }
            if (s_evmask) {                                        // rare: whole slices lose their cache entries first
                const int evmask = s_evmask;
                for (int g = 0; g < NWG; g++) {
                    if (!((evmask >> g) & 1)) continue;
                    const int g0 = g * slice < n ? g * slice : n, g1 = g0 + slice < n ? g0 + slice : n;
                    for (int i = g0 + tid; i < g1; i += NN_THREADS) nnidx[i] = (uint16_t)NN_NOIDX;
                }
                __syncthreads();
                if (tid == 0) after_exchange();
            }
            D++;
        }
        // no barrier here: lane 0 goes straight into the next walk, everybody else to the barrier behind it
        if (prof) { const unsigned long long t1 = wall_clock64(); s_tp[2] += t1 - t0; }
    }
    if (GSIZE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wg != 0) return;
    for (int i = tid; i < nwords; i += NN_THREADS) w.alive[i] = alive[i];
    for (int i = tid; i < n; i += NN_THREADS) {
        w.size[i] = GSIZE ? (uint16_t)__hip_atomic_load(gsize + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : lsize[i];
        w.gtime[i] = -1;
        w.nnc[i] = (uint32_t)nnidx[i] | (((tieb[i >> 5] >> (i & 31)) & 1u) << 16);
    }
    __syncthreads();
    if (tid < D) { w.dslot[tid] = dslot[tid]; w.dtime[tid] = dtime[tid]; }
    __syncthreads();
    if (tid < D && dslot[tid] >= 0) w.gtime[dslot[tid]] = dtime[tid];
    if (tid == 0) {
        w.state[0] = step; w.state[1] = len; w.state[2] = top; w.state[3] = second; w.state[4] = first_ptr;
        w.state[5] = stop_code; w.state[6] = D;
        w.prof[5] += s_cnt[0]; w.prof[6] += s_cnt[1]; w.prof[7] += s_cnt[2];
        if (prof) {
            for (int q = 0; q < 4; q++) w.prof[q] += s_tp[q];
            // detail of the fused pass (profile builds only): issue gathers / LDS pass / loads arrive / compute + stores /
            // gathered values + reductions / stores acknowledged / barrier - and the "update" phase is their sum
            unsigned long long* p2 = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(w.state) + 1280);
            unsigned long long upd = 0;
            for (int q = 0; q < 7; q++) { p2[q] += s_tp[8 + q]; upd += s_tp[8 + q]; }
            w.prof[4] += upd;
        }
    }
}

// ---- one WAVE per column slice (k_nn_epoch_w1, round 3) -----------------------------------------------------------
// k_nn_epoch_mwc spends most of a merge on things a 1024-lane workgroup does to itself: at 16,000 columns on sixteen
// slices a lane owns ONE column, but every merge pays three-plus workgroup barriers, cross-wave reductions through LDS,
// one lane's serial bookkeeping while fifteen waves wait, and an LDS pass of all 1024 lanes over the whole neighbour
// cache.  Here a slice is ONE wave (a 64-lane workgroup, up to 64 of them, one per CU):
//  * no workgroup barrier anywhere: reductions are DPP-only (argmint_wave_fast), the exchange is polled by one lane
//    per peer and reduced in the wave, every lane carries the (uniform) chain state itself - in scalar registers;
//  * every load of a pass - up to TRIPS 16-byte pairs per streamed row and lane, the gathered values, the two cluster
//    sizes - is in flight before anything waits; the cached distances of the own columns live in LDS;
//  * ONE 32-bit word per slot holds everything the serial part of a merge asks about it:
//        idx (15 bits, 0x7fff = unknown) | tie << 15 | stamp << 16 | mtime << 24
//    the cached nearest slot and its tie flag, the epoch-local time the entry was written at, and the time the slot
//    itself last merged (0 = not in this epoch, 255 = dead).  The cache is invalidated LAZILY: an entry is valid iff
//    mtime[idx] != 255 and mtime[idx] <= stamp - two dependent LDS reads per chain step, no bitmaps, no eager pass over
//    all n entries per merge.  k_nn_settle turns the entries back into the stamp-free form between epochs;
//  * NO deferred column writes: with up to 64 workgroups at work the column half of a merge is a handful of scattered
//    8-byte stores per lane (the owner of column j writes W[y'][j] AND W[j][y']), so the matrix stays symmetric and
//    current - no dirty list, no gathered partner values, no "which row is authoritative" logic, no flush kernel;
//  * a single wave issues one instruction every ~4-5 cycles whatever it is, so the serial part is written for instruction
//    count: select-based arg-min updates instead of branches, cluster sizes in a private global array per replica (two
//    loads per merge, issued with the row loads), the replicas' merge records compared as one running hash per replica
//    (k_nn_check_hashes) instead of 4 stores per merge and replica.
// Visibility between the workgroups is the protocol of k_nn_epoch_mw / _mwc unchanged (sc1 stores and loads of every
// matrix byte, stores drained before the mailbox store, bounded spins, double-buffered 16-byte slots that carry their
// sequence number in both halves).
static constexpr int NN_W1_DCAP = 252;                   // merges per epoch: 8-bit times
// columns per slice the plan aims at (the 64 parties stay until 8,192 columns are left).  Swept with the parties on one XCD,
// 64 | 96 | 128 | 160 | 192 | 256 | 320: 16k 78.3 | 78.3 | 78.0 | 80.1 | 80.2 | 79.8 | 96.3 ms, 32k 199.0 | 197.9 | 192.6 | 194.7 | 195.4 |
// 194.3 | 216.7 ms, 2k 8.2 | 8.1 | 8.1 | 8.8 | - | 9.0 | - ms (round 3's first plan was 256: spread out, an exchange cost twice as much)
static constexpr int NN_W1_COLS = 128;
static constexpr int NN_W1_MAX = 32767;                  // 15-bit slot numbers; one 4-byte word per column in the LDS of every replica
static constexpr uint32_t W1_NOIDX = 0x7fffu;
// Loads of this kernel are COMPILER-VISIBLE (its wait-count pass tracks them): the streamed pairs are raw buffer loads with
// the sc1 bit (buffer_load_dwordx4 ... offen sc1; a row is one buffer resource, so lanes beyond the slice read zeros
// instead of needing a branch), the gathered values and sizes relaxed agent-scope atomic loads (global_load ... sc1).
// The earlier kernels issue their loads from inline assembly and wait in a later statement; with up to ~60 loads in
// flight per lane the register allocator then copies a destination register before its load has landed (seen: a fault at
// 16 pairs per lane).  All loads of a pass are issued before NN_FENCE; NN_KEEP makes every use come after it.
#define NN_FENCE() asm volatile("" ::: "memory")
#define NN_KEEP(reg) asm volatile("" : "+v"(reg))
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w1_row(const double* row, int cols)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, cols * 8, 0x00020000);
}
__device__ __forceinline__ u32x4 w1_ld16(__amdgpu_buffer_rsrc_t r, int byte_offset) { return __builtin_amdgcn_raw_buffer_load_b128(r, byte_offset, 0, 16); }
__device__ __forceinline__ unsigned long long w1_ld8(const double* p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int w1_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// both mailbox slots of a peer: issue and wait in ONE statement
__device__ __forceinline__ void w1_poll(u32x4& a, u32x4& b, const u32x4* pa, const u32x4* pb)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
}
__device__ __forceinline__ void st4_sc1(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double bcast_f64(unsigned long long bits, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane((int)(bits >> 32), src), __builtin_amdgcn_readlane((int)(bits & 0xffffffffull), src));
}
// streaming arg-min update without a branch: ascending j per lane, so the first of equal values stays and sets the tie flag
__device__ __forceinline__ void w1_upd(ArgMinT& b, bool live, double v, int j)
{
    const bool lt = live && v < b.v, eq = live && v == b.v;
    b.t = lt ? 0 : (eq ? 1 : b.t);
    b.i = lt ? j : b.i;
    b.v = lt ? v : b.v;
}

// Wave minimum of an fp64 value, in every lane: four DPP exchanges inside the rows of 16 lanes, then gfx950's
// v_permlane16_swap / v_permlane32_swap across the rows (min of both halves of a swap needs no select).  v_min_f64 comes from
// inline assembly: fmin() under IEEE rules costs a canonicalising v_max_f64 per operand - on ONE wave, where every
// instruction is ~4 cycles of the merge's critical path, a reduction of ~55 instructions becomes ~20.
__device__ __forceinline__ double w1_vmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int CTRL>
__device__ __forceinline__ double w1_min_dpp(double v)
{
    const int olo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);      // (every lane has a source: no copy of v first)
    const int ohi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return w1_vmin(v, __hiloint2double(ohi, olo));
}
__device__ __forceinline__ double w1_wave_min(double v)
{
    v = w1_min_dpp<0xB1>(v); v = w1_min_dpp<0x4E>(v); v = w1_min_dpp<0x141>(v); v = w1_min_dpp<0x140>(v);
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = w1_vmin(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = w1_vmin(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
    }
    return v;
}
// Wave arg-min when the candidates' indices do not decrease with the lane (a lane owns a contiguous run of columns;
// a peer owns a contiguous slice): the lowest index at the minimum is the lowest LANE at the minimum - one ballot instead
// of a second reduction.  `odd` >= 0: candidates may also carry that one index out of order (row a's scan is offered
// d(a, y') by whoever owns column a; y' need not lie in its slice) - it wins an exact tie iff it is the lower index.
// Result uniform; i = 0x7fffffff when nothing compares (no candidates / NaN).
__device__ __forceinline__ ArgMinT argmint_wave_mono(ArgMinT a, int odd = -1)
{
    const double m = w1_wave_min(a.v);
    const unsigned long long at = __ballot(a.v == m && a.i != 0x7fffffff);
    ArgMinT r;
    r.v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(m)), __builtin_amdgcn_readfirstlane(__double2loint(m)));
    r.i = 0x7fffffff; r.t = 0;
    if (at) {
        const unsigned long long at_odd = odd >= 0 ? (at & __ballot(a.i == odd)) : 0ull;
        const unsigned long long at_ord = at & ~at_odd;
        int l = at_ord ? (int)__ffsll((long long)at_ord) - 1 : (int)__ffsll((long long)at_odd) - 1;
        r.i = __builtin_amdgcn_readlane(a.i, l);
        if (at_odd && odd < r.i) { r.i = odd; l = (int)__ffsll((long long)at_odd) - 1; }
        r.t = ((at & (at - 1ull)) != 0ull || __builtin_amdgcn_readlane(a.t, l) != 0) ? 1 : 0;
    }
    return r;
}
//   {value bits 31..0, seq} {value bits 63..32, index (15 bits, 0x7fff = none) | tie << 15 | event << 16 | (seq & 0x7fff) << 17}
__device__ __forceinline__ u32x4 w1_mail(double v, int idx, int tie, int ev, unsigned int seq)
{
    u32x4 p;
    p.x = (unsigned int)__double2loint(v); p.y = seq;
    p.z = (unsigned int)__double2hiint(v);
    p.w = ((unsigned int)((idx >= 0 && idx < 0x7fff) ? idx : 0x7fff)) | ((unsigned int)(tie ? 1 : 0) << 15) | ((unsigned int)(ev ? 1 : 0) << 16) |
          ((seq & 0x7fffu) << 17);
    return p;
}
__device__ __forceinline__ bool w1_mail_ready(u32x4 p, unsigned int seq) { return p.y == seq && (p.w >> 17) == (seq & 0x7fffu); }

// LOCAL: every party runs on the XCD `xcc_target` - the grid is 16 x S_arg single-wave workgroups, those that land
// elsewhere leave at once and the first S_arg on the target claim the slices through a counter (w.state[15], zeroed by the
// host) - so the stores a peer reads (mailboxes, the merged row and column) are PLAIN: they stay in the XCD's L2, which
// the peers' sc1 loads read.  Which workgroups share an XCD is never assumed (MI355X_MICROARCH.md: placement is
// undefined): a party knows its XCD from HW_REG_XCC_ID.  Should fewer than S_arg workgroups reach the target, or not all
// be resident at once, the exchange's bounded spin ends the epoch as "a peer answered late" and the host re-runs the
// chain without LOCAL.
template <int TRIPS, bool PROF, bool LOCAL>
__global__ __launch_bounds__(64) void k_nn_epoch_w1(double* __restrict__ W, int64_t ld, int n, int* __restrict__ chain_all,
                                                    double* __restrict__ zraw, NNWorkspace w, int dcap, int total_steps, int slice,
                                                    int S_arg, int xcc_target, int rollcall_need)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    const int lane = (int)threadIdx.x;
    int S_ = (int)gridDim.x, wg_ = (int)blockIdx.x;
    if (LOCAL) {
        if ((int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xfu) != xcc_target) return;
        if (w1_uni(__hip_atomic_load(&w.state[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;   // given up earlier in this chain
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&w.state[15], 1);
        slot = w1_uni(slot);
        if (slot >= S_arg) return;
        // roll call, before anything is touched: a claimed slot is a RUNNING party, so the counter reaching S_arg says all
        // of them are resident.  If it does not within ~2 ms (the XCD is busy with somebody else's kernels) the epoch is
        // given up - state[12] - and the spread-out launch queued right behind this one runs it instead (xcc_target -2).
        int claimed = 0, budget = 4000;
        do {
            claimed = w1_uni(__hip_atomic_load(&w.state[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (claimed >= rollcall_need) break;
            __builtin_amdgcn_s_sleep(8);
        } while (--budget > 0);
        if (claimed < rollcall_need) { if (lane == 0) atomicExch(&w.state[12], 1); return; }
        if (w1_uni(__hip_atomic_load(&w.state[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) return;
        S_ = S_arg; wg_ = slot;
    } else if (xcc_target == -2) {
        // the launch behind a LOCAL one: it has nothing to do when that one ran - S_arg slots claimed and nobody gave up.
        // (Fewer claims than slots: not enough of its workgroups were dealt to the target XCD - placement is not ours to
        // choose; then nobody may even have noticed, so this launch decides by the count, and the chain stays spread out.)
        const int gave_up = w1_uni(__hip_atomic_load(&w.state[12], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int claims = w1_uni(__hip_atomic_load(&w.state[15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (gave_up == 0 && claims >= S_arg) return;
        if (gave_up == 0 && lane == 0 && blockIdx.x == 0) atomicExch(&w.state[12], 1);
    }
    const int S = S_, wg = wg_;
    const int n4 = (n + 3) & ~3;
    uint32_t* meta = reinterpret_cast<uint32_t*>(smem_nn);   // idx | tie << 15 | stamp << 16 | mtime << 24, every slot
    double* nnv = reinterpret_cast<double*>(meta + n4);      // cached distances of the OWN columns (owner-private)
    __shared__ int ring[256];

    int step = w1_uni(w.state[0]);
    if (step >= total_steps || w.state[5]) {
        if (lane == 0 && wg == 0) w.state[6] = 0;
        return;
    }
    // slice: a multiple of 64 columns (16-byte loads never straddle), at most 128 * TRIPS.  A lane owns the 2 * TRIPS
    // consecutive columns from jl0 on (pair t: jl0 + 2t, jl0 + 2t + 1): indices ascend with the lane (argmint_wave_mono)
    const int c0 = wg * slice < n ? wg * slice : n;
    const int c1 = c0 + slice < n ? c0 + slice : n;
    const int jl0 = c0 + 2 * TRIPS * lane;
    double* dump = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(w.mailw) + NN_W1_MAIL + NN_W1_MAXS * 8) + ((int64_t)wg * 64 + lane) * 2;
    int* __restrict__ gsize = w.size_rep + (int64_t)wg * n4;
    int* __restrict__ chain = chain_all + (int64_t)wg * (n + 2);
    const int* __restrict__ chain0 = chain_all;              // chain prefix of the earlier epochs: workgroup 0's copy
    u32x4* mailw = reinterpret_cast<u32x4*>(w.mailw);

    for (int i = lane; i < n4; i += 64) {
        uint32_t m = 0xff000000u | W1_NOIDX;                 // (padding behind n: dead)
        if (i < n) {
            const uint32_t c = w.nnc[i];
            const uint32_t idx = (c & 0xffffu) >= W1_NOIDX ? W1_NOIDX : (c & 0xffffu);
            m = idx | (((c >> 16) & 1u) << 15) | (((w.alive[i >> 5] >> (i & 31)) & 1u) ? 0u : 0xff000000u);
            st4_sc1(gsize + i, (int)w.size[i]);
        }
        meta[i] = m;
    }
    for (int i = c0 + lane; i < c1; i += 64) nnv[i - c0] = w.nnval[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int len = w1_uni(w.state[1]), top = w1_uni(w.state[2]), second = w1_uni(w.state[3]), first_ptr = w1_uni(w.state[4]);
    int ring_lo = len > 256 ? len - 256 : 0;
    for (int k = lane; k < 256; k += 64) if (ring_lo + k < len) ring[(ring_lo + k) & 255] = chain0[ring_lo + k];
    const int step0 = step;
    const int inject_late = w1_uni(w.state[10]), inject_wrong = w1_uni(w.state[11]);      // test hooks (0 = off)
    if (lane == 0 && wg == 0) w.state[9] = step0;
    __syncthreads();

    int tl = 0;                                              // the local clock: merges since the time stamps were last renormalised
    int done_e = 0;                                          // merges of this epoch so far
    int lowmark = len;                                       // chain entries below this are still the earlier epochs'
    unsigned int xseq = (unsigned int)w1_uni(w.state[14]);   // exchanges so far in this map: the mailboxes are cleared once per map
    const unsigned int xseq0 = xseq;
    int guard = 0, stop_code = 0, why = 0;                   // why: which check stopped the chain (diagnostics, state[13])
    unsigned long long hash = 0x243F6A8885A308D3ull;         // of this replica's merge records
    unsigned long long c_cols = 0, c_scans = 0, c_hits = 0;
    unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;   // PROF: walk, scan, exchange+after, merge book, issue, arrive, compute, reduce, drain, -
    const bool prof = PROF && wg == 0;
    unsigned long long npolls = 0;                           // PROF: polls of the slowest-answering peer, summed over the exchanges (per lane)
    unsigned long long kinds[6] = {0, 0, 0, 0, 0, 0};        // PROF: stand-alone scans by cause
    int last_R = -1, last_A = -1, last_my = -1, pushed_cached = 0;
#define W1_STAMP(k) do { if (prof) { const unsigned long long t1_ = wall_clock64(); tp[k] += t1_ - t0; t0 = t1_; } } while (0)

    auto umeta = [&](int i) -> uint32_t { return (uint32_t)w1_uni((int)meta[i]); };       // uniform i
    auto entry_valid = [&](uint32_t e) -> bool {             // uniform
        const uint32_t idx = e & W1_NOIDX;
        if (idx == W1_NOIDX || (int)idx >= n) return false;
        const uint32_t mt = umeta((int)idx) >> 24;           // 255: dead; otherwise valid iff written after the slot's last merge
        return mt != 255u && mt <= ((e >> 16) & 0xffu);
    };
    auto push = [&](int v) {
        if (lane == 0) { st4_sc1(chain + len, v); ring[len & 255] = v; }
        if (len - 255 > ring_lo) ring_lo = len - 255;
        second = top; top = v; len++;
    };
    auto chain_at = [&](int i) -> int {                      // an entry below the LDS ring (rare: chains deeper than 256)
        return w1_uni(__hip_atomic_load((i < lowmark ? chain0 : chain) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    // one exchange: slot 0 = (m0, event flag), slot 1 = m1 (when `two`); results uniform.  Returns 1 when a peer was late.
    auto exchange = [&](ArgMinT m0, int ev0, ArgMinT m1, bool two, int odd1, ArgMinT& r0, ArgMinT& r1, unsigned long long& evm) -> int {
        xseq++;
        // Every reader has an INBOX of its own per parity - [slot 0 of every writer][slot 1 of every writer] - and every writer
        // pushes its slots into all of them (lane p: reader p's inbox).  Polling then touches lines nobody else reads: with
        // 64 readers spinning on the same 16 lines an exchange round took 1.4 us, with private inboxes 1.0 us (0.7 at 16).
        u32x4* box = mailw + (size_t)(xseq & 1u) * (NN_W1_MAXS * 2 * NN_W1_MAXS);
        if (lane < S && lane != wg) {
            u32x4* theirs = box + (size_t)lane * (2 * NN_W1_MAXS) + wg;
            if (LOCAL) {
                st16_l2(theirs, w1_mail(m0.v, m0.i, m0.t, ev0, xseq));
                if (two) st16_l2(theirs + NN_W1_MAXS, w1_mail(m1.v, m1.i, m1.t, 0, xseq));
            } else {
                st16_sc1(theirs, w1_mail(m0.v, m0.i, m0.t, ev0, xseq));
                if (two) st16_sc1(theirs + NN_W1_MAXS, w1_mail(m1.v, m1.i, m1.t, 0, xseq));
            }
        }
        const u32x4* slots = box + (size_t)wg * (2 * NN_W1_MAXS);
        ArgMinT o0 = {__builtin_inf(), 0x7fffffff, 0}, o1 = {__builtin_inf(), 0x7fffffff, 0};
        int late = 0, evbit = 0;
        if (lane < S) {
            if (lane == wg) { o0 = m0; o1 = m1; evbit = ev0; }
            else {
                u32x4 pa, pb;
                const u32x4* sa = slots + lane;
                const u32x4* sb = sa + (two ? NN_W1_MAXS : 0);
                w1_poll(pa, pb, sa, sb);
                int budget = 1000000;
                while (!(w1_mail_ready(pa, xseq) && w1_mail_ready(pb, xseq)) && --budget > 0) {
                    if (budget < 999990) __builtin_amdgcn_s_sleep(1);
                    w1_poll(pa, pb, sa, sb);
                }
                if (PROF) npolls += (unsigned long long)(1000001 - budget);
                if (!(w1_mail_ready(pa, xseq) && w1_mail_ready(pb, xseq))) late = 1;
                const int ia = (int)(pa.w & 0x7fffu), ib = (int)(pb.w & 0x7fffu);
                o0.v = mw_value(pa); o0.i = ia == 0x7fff ? 0x7fffffff : ia; o0.t = (int)((pa.w >> 15) & 1u); evbit = (int)((pa.w >> 16) & 1u);
                o1.v = mw_value(pb); o1.i = ib == 0x7fff ? 0x7fffffff : ib; o1.t = (int)((pb.w >> 15) & 1u);
            }
        }
        if (inject_late > 0 && (int)xseq == inject_late) late = 1;       // test hook: a peer that never answers
        (void)xseq0;
        W1_STAMP(9);                                                     // (PROF: post + polling, until every peer has answered)
        evm = __ballot(evbit != 0);
        r0 = argmint_wave_mono(o0);                                      // a peer's columns lie below the next peer's
        if (two) r1 = argmint_wave_mono(o1, odd1);                       // (slot 1 may carry y', an index outside its sender's slice)
        return __any(late) ? 1 : 0;
    };

    for (; step < total_steps && done_e < dcap; step++) {
        if (tl >= NN_W1_DCAP) {
            // The 8-bit times are about to run out: settle the cache in place (what k_nn_settle does between epochs) - stale
            // entries become "unknown", the others restart at stamp 0, every slot's merge time at 0.  ~8 us per 252 merges;
            // an epoch boundary (state out, three launches, state in on 64 workgroups) costs ~100 us, so an epoch now runs
            // until the next compaction is due instead of 252 merges.
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const uint32_t e = meta[i];
                const uint32_t idx = e & W1_NOIDX;
                bool ok = false;
                if (idx != W1_NOIDX && (int)idx < n) { const uint32_t mt = meta[idx] >> 24; ok = mt != 255u && mt <= ((e >> 16) & 0xffu); }
                if (!ok) meta[i] = e | W1_NOIDX;
            }
            __syncthreads();
            for (int i = lane; i < n; i += 64) {
                const uint32_t e = meta[i];
                meta[i] = (e & 0xffffu) | ((e >> 24) == 255u ? 0xff000000u : 0u);
            }
            __syncthreads();
            tl = 0;
        }
        if (prof) t0 = wall_clock64();
        int act = 0;
        while (true) {
            // ---- walk the chain on cached neighbours (every lane, identical state -> identical walk)
            __syncthreads();
            if (len == 0) {
                int base = first_ptr & ~63, found = -1;
                while (base < n && found < 0) {
                    const int i = base + lane;
                    const unsigned long long b = __ballot(i < n && i >= first_ptr && (meta[i] >> 24) != 255u);
                    if (b) found = base + (int)__ffsll((long long)b) - 1;
                    base += 64;
                }
                if (found < 0) { first_ptr = n; stop_code = NN_STOP_GUARD; why = 1; }
                else {
                    first_ptr = found;
                    if (lane == 0) { st4_sc1(chain, first_ptr); ring[0] = first_ptr; }
                    ring_lo = 0; top = first_ptr; second = -1; len = 1; lowmark = 0;
                }
            }
            act = stop_code ? 3 : 0;                         // 1: scan row `top`, 2: merge (top, second), 3: stop
            while (act == 0) {
                const int prev = (len > 1) ? second : -1;
                const uint32_t e = umeta(top);
                if (!entry_valid(e)) act = 1;
                else if ((int)(e & W1_NOIDX) == prev) act = 2;
                else if (prev >= 0 && ((e >> 15) & 1u)) act = 1;             // an exact tie: the value decides
                else if (++guard > 4 * n + 8) { stop_code = NN_STOP_GUARD; act = 3; why = 2; }
                else { push((int)(e & W1_NOIDX)); c_hits++; pushed_cached = 1; }
            }
            if (act != 1) break;
            // ---- a scan on its own: this slice of row x, then one exchange
            const int x = top, prev = (len > 1) ? second : -1;      // (uniform: the chain state lives in scalar registers)
            c_scans++; c_cols += (unsigned long long)(total_steps + 1 - step);
            if (PROF) {
                const uint32_t ex0 = umeta(x);
                const int kind = len == 1 ? 0 : ((ex0 & W1_NOIDX) != W1_NOIDX && entry_valid(ex0) ? 1 : (x == last_R ? 2 : (x == last_A ? 3 : (pushed_cached ? 4 : 5))));
                kinds[kind]++;
            }
            const uint32_t ex = umeta(x);
            W1_STAMP(0);
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            const __amdgpu_buffer_rsrc_t bx = w1_row(rowx, (c1 + 1) & ~1);
            unsigned long long r_dp = 0;
            if (lane == 63 && prev >= 0) r_dp = w1_ld8(rowx + prev);
            u32x4 q[TRIPS];
#pragma unroll
            for (int t = 0; t < TRIPS; t++) q[t] = w1_ld16(bx, (jl0 + 2 * t) * 8);
            NN_FENCE();
            NN_KEEP(r_dp);
#pragma unroll
            for (int t = 0; t < TRIPS; t++) NN_KEEP(q[t]);
            ArgMinT best = {__builtin_inf(), 0x7fffffff, 0};
#pragma unroll
            for (int t = 0; t < TRIPS; t++) {
                const int j = jl0 + 2 * t;
                const int jj = j < c1 ? j : c0;              // lanes behind the slice read a valid word and ignore it
                const double2 v = mw_pair(q[t]);
                const uint2 m2 = *reinterpret_cast<const uint2*>(meta + jj);
                w1_upd(best, j < c1 && (m2.x >> 24) != 255u && j != x, v.x, j);
                w1_upd(best, j < c1 && (m2.y >> 24) != 255u && j + 1 != x, v.y, j + 1);
            }
            const double dprev = prev >= 0 ? bcast_f64(r_dp, 63) : __builtin_inf();
            best = argmint_wave_mono(best);
            W1_STAMP(1);
            ArgMinT m, m_unused = {__builtin_inf(), 0x7fffffff, 0};
            unsigned long long evm_unused;
            const int late = exchange(best, 0, m_unused, false, -1, m, m_unused, evm_unused);
            if (late) stop_code = NN_STOP_LATE;
            else if (m.i < 0 || m.i >= n) { stop_code = NN_STOP_GUARD; why = 3; }
            else {
                int y = m.i;
                if (prev >= 0 && !(m.v < dprev)) y = prev;               // SciPy: the previous element wins unless STRICTLY closer
                if (lane == 0) {
                    // (reciprocal by the tie rule: the entry names prev, so that the walk sees the pair)
                    meta[x] = (uint32_t)(y == prev && prev >= 0 ? prev : m.i) | ((uint32_t)(m.t ? 1 : 0) << 15) | ((uint32_t)tl << 16) | (ex & 0xff000000u);
                    if (x >= c0 && x < c1) nnv[x - c0] = m.v;            // the owner of column x keeps the distance
                }
                if (y != prev) {
                    if (++guard > 4 * n + 8) { stop_code = NN_STOP_GUARD; why = 4; }
                    push(y);
                }
            }
            W1_STAMP(2);
        }
        if (act == 3) break;
        // ---- merge (top, second); the row the chain returns to - `a` - is scanned in the same pass if it needs it
        int mx = top, my = second;                              // (uniform)
        if (mx > my) { const int t = mx; mx = my; my = t; }
        // the loads that only need (x, y) go out BEFORE the chain bookkeeping - sizes, the height, both streamed rows - so
        // that they land while ~0.8 us of serial code runs; row a's loads follow once a is known and land during the
        // Lance-Williams pass over x and y
        const double* __restrict__ rx = W + (int64_t)mx * ld;
        double* __restrict__ ry = W + (int64_t)my * ld;
        const __amdgpu_buffer_rsrc_t bx = w1_row(rx, (c1 + 1) & ~1), by = w1_row(ry, (c1 + 1) & ~1);
        int r_nx = __hip_atomic_load(gsize + mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int r_ny = __hip_atomic_load(gsize + my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long r_h = 0, r_dp = 0;
        if (lane == 62) r_h = w1_ld8(rx + my);                   // the merge height d(x, y)
        u32x4 qa[TRIPS], qb[TRIPS], qc[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; t++) {
            const int off = (jl0 + 2 * t) * 8;
            qa[t] = w1_ld16(bx, off);
            qb[t] = w1_ld16(by, off);
        }
        NN_FENCE();
        len -= 2;
        if (lane == 0) meta[mx] = 0xff000000u | W1_NOIDX;       // x is dead
        if (len < lowmark) lowmark = len;
        {
            const int i1 = len - 1, i2 = len - 2;
            top = len >= 1 ? (i1 >= ring_lo ? w1_uni(ring[i1 & 255]) : chain_at(i1)) : -1;
            second = len >= 2 ? (i2 >= ring_lo ? w1_uni(ring[i2 & 255]) : chain_at(i2)) : -1;
        }
        __syncthreads();
        int a = -1;
        uint32_t ea = 0u;
        if (top >= 0) {
            ea = umeta(top);
            if (!entry_valid(ea) || (int)(ea & W1_NOIDX) == my) a = top;
        }
        const int aprev = (a >= 0 && len > 1) ? second : -1;
        if (a >= 0) { c_scans++; c_cols += (unsigned long long)(total_steps - step); }
        W1_STAMP(3);
        ArgMinT R = {__builtin_inf(), 0x7fffffff, 0}, A = {__builtin_inf(), 0x7fffffff, 0};
        unsigned long long evm = 0ull;
        double dprev = __builtin_inf(), fs = 0.0;
        {
            const double* __restrict__ ra = W + (int64_t)(a >= 0 ? a : mx) * ld;
            const __amdgpu_buffer_rsrc_t ba = w1_row(ra, (c1 + 1) & ~1);
            if (lane == 63 && aprev >= 0) r_dp = w1_ld8(ra + aprev);
#pragma unroll
            for (int t = 0; t < TRIPS; t++) {
                qc[t] = u32x4{0u, 0u, 0u, 0u};
                if (a >= 0) qc[t] = w1_ld16(ba, (jl0 + 2 * t) * 8);
            }
            NN_FENCE();
            W1_STAMP(4);
            NN_KEEP(r_nx); NN_KEEP(r_ny); NN_KEEP(r_h);
#pragma unroll
            for (int t = 0; t < TRIPS; t++) { NN_KEEP(qa[t]); NN_KEEP(qb[t]); }
            W1_STAMP(5);
            const int nx = __builtin_amdgcn_readfirstlane(r_nx), ny = __builtin_amdgcn_readfirstlane(r_ny);
            if (lane == 0) { st4_sc1(gsize + mx, 0); st4_sc1(gsize + my, nx + ny); }
            const double fx = (double)nx, fy = (double)ny;
            fs = (double)(nx + ny);
            const double rcp = 1.0 / fs;
            ArgMinT rbest = {__builtin_inf(), 0x7fffffff, 0};   // this slice of the new row: the merged cluster's cache entry
            ArgMinT abest = {__builtin_inf(), 0x7fffffff, 0};   // this slice of row a: the streamed columns (ascending per lane)
            bool ev = false;                                    // a cached minimum of this slice was reached or undercut
            double ya = 0.0; bool has_ya = false;               // d(a, y'): the element of the new row at column a (its owner only)
#pragma unroll
            for (int t = 0; t < TRIPS; t++) {
                const int j = jl0 + 2 * t;
                const bool in = j < c1;
                const int jj = in ? j : c0;
                const double2 xa = mw_pair(qa[t]);
                const double2 bo = mw_pair(qb[t]);
                double2 b;
                const double2 nv = *reinterpret_cast<const double2*>(nnv + (jj - c0));
                const uint2 m2 = *reinterpret_cast<const uint2*>(meta + jj);
                const bool w0 = in && (m2.x >> 24) != 255u && j != my, w1 = in && (m2.y >> 24) != 255u && j + 1 != my;
                b.x = div_by_small_int(fx * xa.x + fy * bo.x, fs, rcp);
                b.y = div_by_small_int(fx * xa.y + fy * bo.y, fs, rcp);
                w1_upd(rbest, w0, b.x, j);
                w1_upd(rbest, w1, b.y, j + 1);
                ev = ev || (w0 && (m2.x & W1_NOIDX) != W1_NOIDX && b.x <= nv.x) || (w1 && (m2.y & W1_NOIDX) != W1_NOIDX && b.y <= nv.y);
                if (a >= 0) {                                   // (uniform)
                    if (w0 && j == a) { ya = b.x; has_ya = true; }
                    if (w1 && j + 1 == a) { ya = b.y; has_ya = true; }
                }
                // row y': the pair goes back as one 16-byte store - elements that were not recomputed (the diagonal, dead
                // columns) carry the value just loaded; column y': the recomputed values at W[j][y'], one scattered 8-byte
                // store per element (read later by the owner of column y' only; masked-out ones land in a dump slot)
                {
                    const double sx = w0 ? b.x : bo.x, sy = w1 ? b.y : bo.y;
                    u32x4 pk;
                    pk.x = (unsigned int)__double2loint(sx); pk.y = (unsigned int)__double2hiint(sx);
                    pk.z = (unsigned int)__double2loint(sy); pk.w = (unsigned int)__double2hiint(sy);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, by, j * 8, 0, LOCAL ? 0 : 16);      // (sc1 unless LOCAL; lanes behind the slice are out of range: dropped)
                    if (LOCAL) {
                        st8_l2(w0 ? W + (int64_t)j * ld + my : dump, b.x);
                        st8_l2(w1 ? W + (int64_t)(j + 1) * ld + my : dump + 1, b.y);
                    } else {
                        st8_sc1(w0 ? W + (int64_t)j * ld + my : dump, b.x);
                        st8_sc1(w1 ? W + (int64_t)(j + 1) * ld + my : dump + 1, b.y);
                    }
                }
            }
            W1_STAMP(6);
            // row a: its loads went out after the bookkeeping and have had the pass above to land
            NN_KEEP(r_dp);
#pragma unroll
            for (int t = 0; t < TRIPS; t++) NN_KEEP(qc[t]);
            if (a >= 0) {                                       // (uniform)
#pragma unroll
                for (int t = 0; t < TRIPS; t++) {
                    const int j = jl0 + 2 * t;
                    const bool in = j < c1;
                    const double2 va = mw_pair(qc[t]);
                    const uint2 m2 = *reinterpret_cast<const uint2*>(meta + (in ? j : c0));
                    w1_upd(abest, in && (m2.x >> 24) != 255u && j != my && j != a, va.x, j);
                    w1_upd(abest, in && (m2.y >> 24) != 255u && j + 1 != my && j + 1 != a, va.y, j + 1);
                }
            }
            if (aprev >= 0) dprev = bcast_f64(r_dp, 63);
            rbest = argmint_wave_mono(rbest);
            if (a >= 0) {
                // the merged cluster itself is a candidate of row a's scan, offered by the owner of column a: its index - y' -
                // need not lie in that workgroup's slice, so the reduction over the peers is the full lexicographic one
                // (nobody else can work d(a, y') out: both copies of d(a, y) are being overwritten in this very pass)
                abest = argmint_wave_mono(abest);
                const unsigned long long yb = __ballot(has_ya);
                if (yb) abest = argmint_join(abest, bcast_f64((unsigned long long)__double_as_longlong(ya), (int)__ffsll((long long)yb) - 1), my, 0);
            }
            const int evu = __any(ev) ? 1 : 0;
            const unsigned long long hb = (unsigned long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(r_h >> 32), 62) << 32) |
                                                                (unsigned)__builtin_amdgcn_readlane((int)(r_h & 0xffffffffull), 62));
            if (wg == 0 && lane == 0) {
                zraw[4 * (int64_t)step + 0] = (double)mx; zraw[4 * (int64_t)step + 1] = (double)my;
                zraw[4 * (int64_t)step + 2] = __longlong_as_double((long long)hb); zraw[4 * (int64_t)step + 3] = fs;
            }
            // this replica's record of the merge, folded into a running checksum (scalar: rotate, xor)
            hash = ((hash << 7) | (hash >> 57)) ^ hb ^ (((unsigned long long)(unsigned)mx << 40) | ((unsigned long long)(unsigned)my << 16) | (unsigned long long)(unsigned)(nx + ny));
            if (inject_wrong > 0 && wg == 1 && step == inject_wrong) hash ^= 1ull;    // test hook: a replica that read a stale height
            W1_STAMP(7);
            // every store of the pass is acknowledged before the mailbox store signals for them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            W1_STAMP(8);
            const int late = exchange(rbest, evu, abest, a >= 0, my, R, A, evm);
            if (late) stop_code = NN_STOP_LATE;
        }
        if (stop_code) break;
        if (evm) {                                              // rare: whole slices lose their cache entries first
            for (int g = 0; g < S; g++) {
                if (!((evm >> g) & 1ull)) continue;
                const int g0 = g * slice < n ? g * slice : n, g1 = g0 + slice < n ? g0 + slice : n;
                for (int i = g0 + lane; i < g1; i += 64) meta[i] |= W1_NOIDX;
            }
            __syncthreads();
        }
        // ---- the merged cluster's neighbour is known; the fused scan of row a is decided like a scan on its own
        tl++; done_e++;
        if (PROF) { last_R = R.i; last_A = a >= 0 ? A.i : -1; last_my = my; pushed_cached = 0; }
        if (lane == 0) {
            const bool known = R.i >= 0 && R.i < n;
            meta[my] = (known ? (uint32_t)R.i | ((uint32_t)(R.t ? 1 : 0) << 15) : W1_NOIDX) | ((uint32_t)tl << 16) | ((uint32_t)tl << 24);
            if (known && my >= c0 && my < c1) nnv[my - c0] = R.v;
        }
        if (a >= 0) {
            if (A.i < 0 || A.i >= n) { stop_code = NN_STOP_GUARD; why = 5; }
            else {
                int y = A.i;
                if (aprev >= 0 && !(A.v < dprev)) y = aprev;
                if (lane == 0) {
                    meta[a] = (uint32_t)(y == aprev && aprev >= 0 ? aprev : A.i) | ((uint32_t)(A.t ? 1 : 0) << 15) | ((uint32_t)tl << 16) | (ea & 0xff000000u);
                    if (a >= c0 && a < c1) nnv[a - c0] = A.v;
                }
                if (y != aprev) {
                    if (++guard > 4 * n + 8) { stop_code = NN_STOP_GUARD; why = 6; }
                    push(y);
                }
            }
        }
        W1_STAMP(2);
        if (stop_code) { step++; break; }
    }
    // ---- save the state for the flush / settle kernels and the next epoch
    __syncthreads();
    unsigned long long npolls_max = 0;
    if (PROF) { for (int l = 0; l < 64; l++) { const unsigned long long v = (unsigned long long)__builtin_amdgcn_readlane((int)npolls, l); if (v > npolls_max) npolls_max = v; } }
    for (int i = c0 + lane; i < c1; i += 64) w.nnval[i] = nnv[i - c0];      // every owner: its cached distances
    if (lane == 0) reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(w.mailw) + NN_W1_MAIL)[wg] = hash;
    if (wg != 0) return;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const uint32_t m = i < n ? meta[i] : 0xff000000u;
        const unsigned long long al = __ballot((m >> 24) != 255u);
        if (lane == 0) { w.alive[i0 >> 5] = (uint32_t)al; if (i0 + 32 < ((n + 31) & ~31)) w.alive[(i0 >> 5) + 1] = (uint32_t)(al >> 32); }
        if (i < n) {
            w.size[i] = (uint16_t)__hip_atomic_load(gsize + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w.gtime[i] = -1;                                    // (no deferred columns)
            w.newidx[i] = (int)m;                               // with time stamps: k_nn_settle strips them (newidx: free between compactions)
        }
    }
    if (lane == 0) {
        w.state[0] = step; w.state[1] = len; w.state[2] = top; w.state[3] = second; w.state[4] = first_ptr;
        w.state[5] = stop_code; w.state[6] = 0; if (why) w.state[13] = why;
        w.state[14] = (int)xseq;
        w.prof[5] += c_cols; w.prof[6] += c_scans; w.prof[7] += c_hits;
        if (PROF) {
            w.prof[0] += tp[0]; w.prof[1] += tp[1]; w.prof[2] += tp[2]; w.prof[3] += tp[3];
            unsigned long long* p2 = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(w.state) + 1280);
            // detail of the fused pass: issue loads / - / loads arrive / compute + stores / gathered values + reductions /
            // stores acknowledged / -
            p2[0] += tp[4]; p2[2] += tp[5]; p2[3] += tp[6]; p2[4] += tp[7]; p2[5] += tp[8];
            w.prof[4] += tp[4] + tp[5] + tp[6] + tp[7] + tp[8];
            w.prof[2] += tp[9];                                   // "pick" = post + polling + reductions + what follows the exchange
            p2[8] += tp[9]; p2[9] += (unsigned long long)(xseq - xseq0); p2[10] += npolls_max;
            for (int q = 0; q < 6; q++) p2[12 + q] += kinds[q];
        }
    }
#undef W1_STAMP
}

// Between two epochs of k_nn_epoch_w1: cache entries that went stale during the epoch (their slot died, or merged after
// the entry was written) become "unknown", the others lose their time stamp - the form k_nn_remap, k_nn_epoch_nc and the
// next epoch expect (idx | tie << 16, 0xffff = unknown).
__global__ __launch_bounds__(256) void k_nn_settle(NNWorkspace w, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t e = (uint32_t)w.newidx[i];                  // the epoch's words: idx | tie << 15 | stamp << 16 | mtime << 24
    const uint32_t idx = e & W1_NOIDX;
    uint32_t out = NN_NOIDX;
    if (idx != W1_NOIDX && (int)idx < n) {
        const uint32_t mt = (uint32_t)w.newidx[idx] >> 24;
        if (mt != 255u && mt <= ((e >> 16) & 0xffu)) out = idx | (((e >> 15) & 1u) << 16);
    }
    w.nnc[i] = out;
}

// Every replica of k_nn_epoch_w1 hashes the merges it decided; a difference (a stale read between workgroups) stops the
// chain instead of returning a wrong tree.
__global__ __launch_bounds__(64) void k_nn_check_hashes(NNWorkspace w, int S)
{
    const unsigned long long* h = reinterpret_cast<const unsigned long long*>(reinterpret_cast<const unsigned char*>(w.mailw) + NN_W1_MAIL);
    const int lane = threadIdx.x;
    const bool bad = lane < S && h[lane] != h[0];
    if (__any(bad) && lane == 0 && w.state[5] == 0) w.state[5] = NN_STOP_DIVERGED;
}

// Full-chip flush of the deferred column writes: for every dirty cluster d (time td) and every live
// row i that did not merge after td, W[i][d] = W[d][i].  Reads are coalesced along row d.
__global__ __launch_bounds__(256) void k_nn_flush(double* __restrict__ W, int64_t ld, int n, NNWorkspace w)
{
    const int e = blockIdx.y;
    if (e >= w.state[6]) return;
    const int d = w.dslot[e];
    if (d < 0 || !((w.alive[d >> 5] >> (d & 31)) & 1u)) return;
    const int td = w.dtime[e];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || i == d) return;
    if (!((w.alive[i >> 5] >> (i & 31)) & 1u)) return;
    if (w.gtime[i] > td) return;                           // row i is the authoritative one for this pair
    W[(int64_t)i * ld + d] = W[(int64_t)d * ld + i];
}

// ---- compaction ------------------------------------------------------------------------------------
// Dead clusters still occupy columns of every row that is scanned.  When fewer than 3/4 of the
// current slots are alive, the live rows/columns are copied (order preserved, so "lowest index wins"
// ties are unaffected) into the other buffer by the whole chip, and the chain continues on the
// smaller matrix.  Raw merges are recorded with current slot numbers and translated to original bins
// at the end of every compaction interval.
__global__ __launch_bounds__(1024) void k_nn_remap(NNWorkspace w, int n_cur, int* __restrict__ chain)
{
    __shared__ int wsum[1024];
    __shared__ int s_total;
    const int tid = threadIdx.x, nwords = (n_cur + 31) >> 5;
    if (w.state[5]) return;
    // exclusive prefix of the per-word popcounts (nwords <= 2048: two words per lane)
    const int w0 = 2 * tid, w1 = 2 * tid + 1;
    const uint32_t a0 = w0 < nwords ? w.alive[w0] : 0u, a1 = w1 < nwords ? w.alive[w1] : 0u;
    const int c0 = __popc(a0), c1 = __popc(a1);
    wsum[tid] = c0 + c1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = tid >= off ? wsum[tid - off] : 0;
        __syncthreads();
        wsum[tid] += v;
        __syncthreads();
    }
    const int base0 = wsum[tid] - (c0 + c1), base1 = base0 + c0;
    if (tid == 1023) s_total = wsum[1023];
    for (int b = 0; b < 32; b++) {
        const int i0 = w0 * 32 + b, i1 = w1 * 32 + b;
        if (i0 < n_cur) {
            if ((a0 >> b) & 1u) { int nw = base0 + __popc(a0 & ((1u << b) - 1u)); w.newidx[i0] = nw; w.oldidx[nw] = i0; }
            else w.newidx[i0] = -1;
        }
        if (i1 < n_cur) {
            if ((a1 >> b) & 1u) { int nw = base1 + __popc(a1 & ((1u << b) - 1u)); w.newidx[i1] = nw; w.oldidx[nw] = i1; }
            else w.newidx[i1] = -1;
        }
    }
    __threadfence_block();
    __syncthreads();
    const int n_new = s_total;
    // sizes / original bins move down in place: chunk by chunk, reads of a chunk finish before its writes,
    // and a write never lands beyond the chunk being processed (new index <= old index)
    for (int c0i = 0; c0i < n_cur; c0i += 1024) {
        const int i = c0i + tid;
        int nw = -1; uint16_t sz = 0; int og = 0; uint32_t cc = NN_NOIDX; double cv = 0.0;
        if (i < n_cur) { nw = w.newidx[i]; sz = w.size[i]; og = w.orig[i]; cc = w.nnc[i]; cv = w.nnval[i]; }
        __syncthreads();
        if (nw >= 0) {
            w.size[nw] = sz; w.orig[nw] = og;
            // neighbour cache: the cached slot moves with its cluster (a live row never caches a dead slot while the
            // cache is in use; epochs that do not maintain it leave stale entries, which are rebuilt before use)
            const uint32_t id = cc & 0xffffu;
            const int nid = (id != NN_NOIDX && (int)id < n_cur) ? w.newidx[id] : -1;
            w.nnc[nw] = nid >= 0 ? ((uint32_t)nid | (cc & 0x10000u)) : NN_NOIDX;
            w.nnval[nw] = cv;
        }
        __syncthreads();
    }
    const int len = w.state[1];
    for (int i = tid; i < len; i += 1024) chain[i] = w.newidx[chain[i]];
    const int nw_words = (n_new + 31) >> 5;
    for (int i = tid; i < nw_words; i += 1024) {
        int rem = n_new - i * 32;
        w.alive[i] = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    }
    for (int i = tid; i < n_new; i += 1024) w.gtime[i] = -1;
    __syncthreads();
    if (tid == 0) {
        if (w.state[2] >= 0) w.state[2] = w.newidx[w.state[2]];
        if (w.state[3] >= 0) w.state[3] = w.newidx[w.state[3]];
        w.state[4] = 0;                                  // every slot is alive again: the lowest live index is 0
        w.state[7] = n_new;
    }
}

__global__ __launch_bounds__(256) void k_nn_compact_copy(const double* __restrict__ src, double* __restrict__ dst,
                                                         int64_t ld, NNWorkspace w, int n_new)
{
    if (w.state[5]) return;
    const int r = blockIdx.x;
    const double* __restrict__ srow = src + (int64_t)w.oldidx[r] * ld;
    double* __restrict__ drow = dst + (int64_t)r * ld;
    for (int c = threadIdx.x; c < n_new; c += 256) drow[c] = srow[w.oldidx[c]];
    if (threadIdx.x == 0 && n_new < ld) drow[n_new] = __builtin_inf();
}

__global__ __launch_bounds__(256) void k_nn_translate(double* __restrict__ zraw, int s0, int s1, NNWorkspace w)
{
    const int s = s0 + blockIdx.x * 256 + threadIdx.x;
    if (s >= s1 || s >= w.state[0]) return;
    zraw[4 * (int64_t)s + 0] = (double)w.orig[(int)zraw[4 * (int64_t)s + 0]];
    zraw[4 * (int64_t)s + 1] = (double)w.orig[(int)zraw[4 * (int64_t)s + 1]];
}

template <int NWG>
static void launch_mwc_n(bool profile, size_t lds, hipStream_t s, double* cur, int64_t ldw, int n_cur, int* chain, double* zraw,
                         NNWorkspace w, int dcap, int total_steps)
{
    if (profile) hipLaunchKernelGGL((k_nn_epoch_mwc<NWG, true>), dim3(NWG), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
    else hipLaunchKernelGGL((k_nn_epoch_mwc<NWG, false>), dim3(NWG), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
}

static void launch_mwc(int wgs, bool profile, size_t lds, hipStream_t s, double* cur, int64_t ldw, int n_cur, int* chain,
                       double* zraw, NNWorkspace w, int dcap, int total_steps, bool gsize = false)
{
    if (gsize) {                                           // (8 or 16 slices: what rows beyond 32,768 columns use)
        if (wgs == 16) {
            if (profile) hipLaunchKernelGGL((k_nn_epoch_mwc<16, true, true>), dim3(16), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else hipLaunchKernelGGL((k_nn_epoch_mwc<16, false, true>), dim3(16), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        } else {
            if (profile) hipLaunchKernelGGL((k_nn_epoch_mwc<8, true, true>), dim3(8), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else hipLaunchKernelGGL((k_nn_epoch_mwc<8, false, true>), dim3(8), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        }
        return;
    }
    if (wgs == 16) launch_mwc_n<16>(profile, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
    else if (wgs == 2) launch_mwc_n<2>(profile, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
    else if (wgs == 4) launch_mwc_n<4>(profile, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
    else launch_mwc_n<8>(profile, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
}

static void mwc_set_lds(int bytes)
{
    const void* fns[] = {reinterpret_cast<const void*>(k_nn_epoch_mwc<2, false>), reinterpret_cast<const void*>(k_nn_epoch_mwc<2, true>),
                         reinterpret_cast<const void*>(k_nn_epoch_mwc<4, false>), reinterpret_cast<const void*>(k_nn_epoch_mwc<4, true>),
                         reinterpret_cast<const void*>(k_nn_epoch_mwc<8, false>), reinterpret_cast<const void*>(k_nn_epoch_mwc<8, true>),
                         reinterpret_cast<const void*>(k_nn_epoch_mwc<16, false>), reinterpret_cast<const void*>(k_nn_epoch_mwc<16, true>)};
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Largest row the GSIZE variant can take: bitmaps (12 bytes per 32 columns) + cache (2 bytes per column) + the kernel's
// static LDS within the 160 KB of a CU.  Also raises the kernels' dynamic-LDS limit to what is left.
static int mwc_gsize_max_columns()
{
    static int cached = -1;
    if (cached >= 0) return cached;
    size_t stat = 0;
    const void* fns[] = {reinterpret_cast<const void*>(k_nn_epoch_mwc<8, false, true>), reinterpret_cast<const void*>(k_nn_epoch_mwc<8, true, true>),
                         reinterpret_cast<const void*>(k_nn_epoch_mwc<16, false, true>), reinterpret_cast<const void*>(k_nn_epoch_mwc<16, true, true>)};
    for (const void* f : fns) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, f) != hipSuccess) { cached = 0; return 0; }
        if (a.sharedSizeBytes > stat) stat = a.sharedSizeBytes;
    }
    const size_t room = 160 * 1024 > stat + 256 ? 160 * 1024 - stat - 256 : 0;
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)room);
    int n = NN_MWC_GMAX;
    while (n > 0) {
        const size_t nw4 = ((((size_t)n + 31) / 32) + 3) & ~(size_t)3;
        if (align16(nw4 * 12 + (((size_t)n + 7) & ~(size_t)7) * 2) <= room) break;
        n -= 64;
    }
    cached = n > 0 ? n : 0;
    return cached;
}

// ---- k_nn_epoch_w1: slices, trips and LDS of one epoch ----------------------------------------------------------
template <int TRIPS>
static void launch_w1_t(bool profile, int S, int slice, size_t lds, hipStream_t s, double* cur, int64_t ldw, int n_cur, int* chain,
                        double* zraw, NNWorkspace w, int dcap, int total_steps, int xcc, bool fail_rollcall)
{
    if (xcc >= 0) {                                              // LOCAL (see the kernel), and behind it the launch that takes over if it gives up
        const int need = fail_rollcall ? 0x40000000 : S;          // (test hook: a party that never comes)
        hipMemsetAsync(w.state + 15, 0, sizeof(int), s);
        if (profile) hipLaunchKernelGGL((k_nn_epoch_w1<TRIPS, true, true>), dim3(16 * S), dim3(64), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, slice, S, xcc, need);
        else hipLaunchKernelGGL((k_nn_epoch_w1<TRIPS, false, true>), dim3(16 * S), dim3(64), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, slice, S, xcc, need);
        hipLaunchKernelGGL((k_nn_epoch_w1<TRIPS, false, false>), dim3(S), dim3(64), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, slice, S, -2, S);
        return;
    }
    if (profile) hipLaunchKernelGGL((k_nn_epoch_w1<TRIPS, true, false>), dim3(S), dim3(64), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, slice, S, -1, S);
    else hipLaunchKernelGGL((k_nn_epoch_w1<TRIPS, false, false>), dim3(S), dim3(64), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, slice, S, -1, S);
}

static void launch_w1(bool profile, int S, int slice, size_t lds, hipStream_t s, double* cur, int64_t ldw, int n_cur, int* chain,
                      double* zraw, NNWorkspace w, int dcap, int total_steps, int xcc, bool fail_rollcall)
{
    const int trips = (slice + 127) / 128;
    if (trips <= 1) launch_w1_t<1>(profile, S, slice, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, xcc, fail_rollcall);
    else if (trips <= 2) launch_w1_t<2>(profile, S, slice, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, xcc, fail_rollcall);
    else if (trips <= 4) launch_w1_t<4>(profile, S, slice, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, xcc, fail_rollcall);
    else if (trips <= 8) launch_w1_t<8>(profile, S, slice, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, xcc, fail_rollcall);
    else launch_w1_t<16>(profile, S, slice, lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, xcc, fail_rollcall);
}

// Dynamic LDS the kernels may ask for: what the CU has (160 KB) minus their static part.
static size_t w1_lds_room()
{
    static size_t room = 0;
    if (room) return room;
#define W1_FNS(T) reinterpret_cast<const void*>(k_nn_epoch_w1<T, false, false>), reinterpret_cast<const void*>(k_nn_epoch_w1<T, true, false>), \
                  reinterpret_cast<const void*>(k_nn_epoch_w1<T, false, true>), reinterpret_cast<const void*>(k_nn_epoch_w1<T, true, true>)
    const void* fns[] = {W1_FNS(1), W1_FNS(2), W1_FNS(4), W1_FNS(8), W1_FNS(16)};
#undef W1_FNS
    size_t stat = 0;
    for (const void* f : fns) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, f) != hipSuccess) return 0;
        if (a.sharedSizeBytes > stat) stat = a.sharedSizeBytes;
    }
    const size_t r = 160 * 1024 > stat + 256 ? 160 * 1024 - stat - 256 : 0;
    for (const void* f : fns) hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)r);
    room = r;
    return room;
}

// Slices of an epoch with n live columns: about `cols` columns per single-wave workgroup, at most max_s of them; the
// slice width is a multiple of 64 and at most 16 trips x 128 columns.  False: the kernel cannot take this width.
static bool w1_plan(int n, int cols, int max_s, int force_s, int* S_out, int* slice_out, size_t* lds_out)
{
    if (n > NN_W1_MAX || n < 2) return false;
    if (max_s > NN_W1_MAXS) max_s = NN_W1_MAXS;
    if (max_s < 1) max_s = 1;
    int S = force_s > 0 ? (force_s < NN_W1_MAXS ? force_s : NN_W1_MAXS) : (n + cols - 1) / cols;
    if (S > max_s && force_s <= 0) S = max_s;
    if (S < 1) S = 1;
    int slice = (((n + S - 1) / S) + 63) & ~63;
    while (slice > 16 * 128 && S < NN_W1_MAXS) { S++; slice = (((n + S - 1) / S) + 63) & ~63; }
    if (slice > 16 * 128) return false;
    S = (n + slice - 1) / slice;                                  // (rounding the width up can leave the last slices empty)
    const int n4 = (n + 3) & ~3;
    const size_t lds = align16((size_t)n4 * 4 + (size_t)slice * 8);
    if (lds > w1_lds_room()) return false;
    *S_out = S; *slice_out = slice; *lds_out = lds;
    return true;
}

// The XCD the one-wave kernel claims (its LOCAL form), or -1: HICMI_NNCHAIN_XCD=off | 0..7 (default 0).  The pre-sort that runs
// beside the chain leaves that XCD alone (api.hip: start_presort), so its 32 CUs are free for up to 64 parties.
// Which XCC ids this device's workgroups report at all: the lowest is the default target.  (A whole MI355X answers 0 ... 7; a
// partition of one may answer with a single id, and not necessarily 0 - a fixed target would then never be claimed.)
__global__ void k_probe_xcc(int* __restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(GETREG_XCC_ID) & 0xfu);
}

static int w1_probed_xcc()
{
    static std::atomic<int> cached[64];
    static std::atomic<bool> init{false};
    if (!init.exchange(true)) for (auto& c : cached) c.store(-2);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cached[dev].load();
    if (v != -2) return v;
    v = 0;
    int* d = nullptr;
    if (hipMalloc((void**)&d, 256 * sizeof(int)) == hipSuccess) {
        int h[256];
        hipLaunchKernelGGL(k_probe_xcc, dim3(256), dim3(64), 0, 0, d);
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            v = 15;
            for (int i = 0; i < 256; i++) if (h[i] >= 0 && h[i] < v) v = h[i];
            if (v > 7) v = 0;
        }
        (void)hipFree(d);
    }
    cached[dev].store(v);
    return v;
}

static int w1_xcc_env()
{
    const char* t = getenv("HICMI_NNCHAIN_XCD");          // off | 0 .. 7 (8: an XCD that does not exist - nobody claims a slice; tests)
    if (!t) return w1_probed_xcc();
    return (t[0] < '0' || t[0] > '8') ? -1 : t[0] - '0';
}

// ... for the pre-sort's benefit: the XCD it should leave alone, i.e. the chain's XCD if the chain's FIRST epoch at n columns
// already runs there.  (A wider map starts spread out - 8 parties per XCD, which the pre-sort's 192 workgroups leave room
// for on every XCD - and by the time it has shrunk to one XCD's capacity the pre-sort is over.)
int nnchain_local_xcc(int n)
{
    const int xcc = w1_xcc_env();
    if (xcc < 0 || n > NN_W1_MAX) return -1;
    const char* w1_text = getenv("HICMI_NNCHAIN_W1");
    if ((w1_text && atoi(w1_text) == 0) || getenv("HICMI_NNCHAIN_WGS") || getenv("HICMI_NNCHAIN_PLAIN") || getenv("HICMI_NNCHAIN_GSIZE")) return -1;
    if (getenv("HICMI_NNCHAIN_XCD_WIDE")) return xcc;
    int S = 0, slice = 0; size_t lds = 0;
    if (!w1_plan(n, NN_W1_COLS, NN_W1_MAXS, 0, &S, &slice, &lds)) return -1;
    return S <= 32 * (int)((160 * 1024) / (lds + 2048)) ? xcc : -1;
}

// W and W2: two n x ldw buffers (W holds the distances on entry; both are scratch afterwards).
// Returns the number of epoch launches.  force_single: never the column-sliced kernel (the retry after a late peer).
int launch_nnchain(double* W, double* W2, int64_t ldw, int n, int* chain, double* zraw, void* workspace, bool profile,
                   int dcap, bool compact, int fallback, hipStream_t s, const std::function<void()>& after_first_rowmin)
{
    bool told = false;
    auto tell = [&] { if (!told && after_first_rowmin) after_first_rowmin(); told = true; };
    // fallback (the retries after "a peer answered late"): 1 = the one-wave kernel spread over the chip, never on one XCD;
    // 2 = one workgroup, which waits for nobody
    const bool force_single = fallback >= 2;
    int epochs = 0;
    NNWorkspace w = carve(workspace, n);
    // Column-sliced chain on several workgroups (k_nn_epoch_mw): its fixed cost per scan (one exchange) is paid back
    // by the shorter streams from about 20k live columns on.  HICMI_NNCHAIN_WGS = 1, 2, 4 or 8 forces a width for
    // every epoch (tests, A/B).  Narrower epochs run on one workgroup with the neighbour cache (k_nn_epoch_nc);
    // HICMI_NNCHAIN_PLAIN=1 selects the cache-less k_nn_epoch instead (A/B, and the reference point of the tests).
    const char* wgs_text = force_single ? "1" : getenv("HICMI_NNCHAIN_WGS");
    const int wgs_env = wgs_text ? atoi(wgs_text) : 0;
    const int wgs = wgs_text ? (wgs_env >= 16 ? 16 : (wgs_env >= 8 ? 8 : (wgs_env >= 4 ? 4 : (wgs_env >= 2 ? 2 : 1)))) : 8;
    // 16 slices instead of 8 while the rows are long: a merge costs ~7.4 / 8.5 / 11.4 us at 2,000 / 4,000 / 8,000 columns per
    // slice, and the sixteen-way exchange only a little more than the eight-way one.  Measured (nn-chain per map, threshold
    // off / 24,000 / 16,000 / 12,000 live columns): 32k 272 / 264 / 256 / 259 ms, 64k 730 / 682 / 678 / - ms, 16k 118.3 / - / - / 116.8
    const int w16_from = wgs_text ? 0x7fffffff : 14000;
    // live columns from which an epoch runs sliced: with the cache and the fused scan a merge costs ~1.3 exchanges instead
    // of ~2.9, so eight slices pay from ~6,000 columns on (16k map: nn-chain 200 -> 153 ms; 4,000 and 8,000 measure the same)
    const int mw_from = wgs_text ? 64 * wgs : 6000;
    const bool plain = getenv("HICMI_NNCHAIN_PLAIN") != nullptr;
    // Rebuild the whole cache (k_nn_rowmin, a full-chip pass over the flushed matrix: 0.13 ms at 8,000 live columns) before
    // every epoch that has at most this many live columns: rows whose neighbour merged are then known again without a scan
    // of their own.  Worth 1.5 ms per 16k map (scans per merge 1.28 -> 1.23: most such rows are walked within the epoch that
    // invalidated them); above ~12,000 columns the pass costs more than the scans it saves.  0 = never.
    const int refresh_below = 8000;
    // The matrix is compacted - live rows and columns copied into the other buffer - when HALF of its columns have merged
    // away.  (Rounds 1-2 and the first one-wave kernel compacted at three quarters: with deferred column writes a merge's
    // cost grew with the width.  The one-wave kernel's does not until the slice drops under the next power-of-two number of
    // pairs per lane, an epoch boundary costs ~100 us plus the copy, and a 32k map is down to one XCD's capacity after ONE
    // compaction instead of two.  Measured on one box, 3/4 | 2/3 | 3/5 | 1/2 | 2/5 | 1/3 | 1/4 of the columns left:
    // 16k 81.6 | 80.4 | 80.1 | 80.0 | 80.3 | 80.5 | 80.8 ms, 32k 203.5 | 197.6 | 200.2 | 194.0 | 198.3 | 202.4 | 206.0 ms;
    // 64k 602.7 -> 586.6 ms.)  HICMI_NNCHAIN_COMPACT_AT=<num>/<den> sets another fraction (A/B).
    int cnum = 1, cden = 2;
    if (const char* t = getenv("HICMI_NNCHAIN_COMPACT_AT")) {
        int a = 0, b = 0;
        if (sscanf(t, "%d/%d", &a, &b) == 2 && a >= 1 && b > a && b <= 64) { cnum = a; cden = b; }
    }
    const bool force_gsize = getenv("HICMI_NNCHAIN_GSIZE") != nullptr;   // the GSIZE variant of k_nn_epoch_mwc at every width (tests)
    // One wave per column slice (k_nn_epoch_w1): the default at every width up to 32,768 live columns.  HICMI_NNCHAIN_W1=0
    // selects the 1024-lane kernels of rounds 1-2 instead (A/B; so does every switch that names one of them);
    // HICMI_NNCHAIN_W1_S forces the number of slices (tests), HICMI_NNCHAIN_W1_COLS / _MAXS set the columns per slice the
    // plan aims at and the largest number of slices.
    const char* w1_text = getenv("HICMI_NNCHAIN_W1");
    const char* w1_s_text = getenv("HICMI_NNCHAIN_W1_S");
    const char* w1_cols_text = getenv("HICMI_NNCHAIN_W1_COLS");
    const char* w1_maxs_text = getenv("HICMI_NNCHAIN_W1_MAXS");
    const bool w1_on = !(w1_text && atoi(w1_text) == 0) && !wgs_text && !force_single && !plain && !force_gsize;
    const int w1_force_s = w1_s_text ? atoi(w1_s_text) : 0;
    const int w1_cols = w1_cols_text ? (atoi(w1_cols_text) > 64 ? atoi(w1_cols_text) : 64) : NN_W1_COLS;
    const int w1_max_s = w1_maxs_text ? atoi(w1_maxs_text) : NN_W1_MAXS;
    const bool dcap_forced = getenv("HICMI_NNCHAIN_DCAP") != nullptr;
    const bool w1_xcd_wide = getenv("HICMI_NNCHAIN_XCD_WIDE") != nullptr;
    const int gsize_max = (n > NN_MWC_MAX || force_gsize) ? mwc_gsize_max_columns() : 0;
    if (dcap < 1) dcap = 1;
    if (dcap > NN_DMAX) dcap = NN_DMAX;
    hipLaunchKernelGGL(k_nn_init, dim3(64), dim3(256), 0, s, w, n);
    if (w1_on) hipMemsetAsync(w.mailw, 0, NN_W1_MAIL, s);          // (its exchange numbers run on over the epochs of a map)
    if (profile) { static const int one = 1; hipMemcpyAsync(w.state + 8, &one, sizeof(int), hipMemcpyHostToDevice, s); }
    {
        static int hooks[2];                                     // test hooks: see NNWorkspace::state[10], [11]
        const char* a = getenv("HICMI_NNCHAIN_TEST_LATE"); const char* b = getenv("HICMI_NNCHAIN_TEST_DIVERGE");
        hooks[0] = (a && fallback == 0) ? atoi(a) : 0; hooks[1] = b ? atoi(b) : 0;      // (the retries run without the late-peer hook)
        if (hooks[0] || hooks[1]) hipMemcpyAsync(w.state + 10, hooks, sizeof(hooks), hipMemcpyHostToDevice, s);
    }
    const int total_steps = n - 1;
    int n_cur = n, done = 0, interval_start = 0;
    double *cur = W, *other = W2;
    bool cache_valid = false;                                    // w.nnval / w.nnc describe the current matrix
    {
        const int nwords = (n + 31) / 32, nw4 = (nwords + 3) & ~3;
        size_t lds_max = align16((size_t)nw4 * 8 + (size_t)n * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        const int nc = n < NN_NC_MAX ? n : NN_NC_MAX, ncw4 = (((nc + 31) / 32) + 3) & ~3;
        size_t lds_nc = align16((size_t)ncw4 * 12 + (size_t)((nc + 7) & ~7) * 4);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch_nc<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nc);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch_nc<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_nc);
        const int mc = n < NN_MWC_MAX ? n : NN_MWC_MAX, mcw4 = (((mc + 31) / 32) + 3) & ~3;
        size_t lds_mc = align16((size_t)mcw4 * 12 + (size_t)((mc + 7) & ~7) * 4);
        mwc_set_lds((int)lds_mc);
    }
    while (done < total_steps) {
        const int nwords = (n_cur + 31) / 32, nw4 = (nwords + 3) & ~3;
        const size_t lds = align16((size_t)nw4 * 8 + (size_t)n_cur * 2);
        const bool sliced = wgs > 1 && n_cur >= mw_from;
        const int wgs_e = n_cur >= w16_from ? 16 : wgs;            // this epoch's width (k_nn_epoch_mwc only)
        // rows beyond 32,768 columns (or HICMI_NNCHAIN_GSIZE=1: all, for the tests): cluster sizes in global memory, the
        // cache alone in LDS - 64,000 columns fit; 8 or 16 slices
        const bool gsize = wgs_e >= 8 && (n_cur > NN_MWC_MAX || force_gsize) && n_cur <= gsize_max;
        int w1_S = 0, w1_slice = 0; size_t w1_lds = 0;
        bool flush_needed = true;
        int dcap_e = dcap;                                          // merges of this epoch
        if (w1_on && w1_plan(n_cur, w1_cols, w1_max_s, w1_force_s, &w1_S, &w1_slice, &w1_lds)) {
            if (!cache_valid || (refresh_below > 0 && n_cur <= refresh_below)) hipLaunchKernelGGL(k_nn_rowmin, dim3(n_cur), dim3(256), 0, s, cur, ldw, n_cur, w);
            cache_valid = true;
            tell();
            // the one-wave kernel renormalises its time stamps itself: an epoch runs until the next compaction is due
            // (half of the columns have merged away), at least 256 merges; HICMI_NNCHAIN_DCAP still forces a length
            if (!dcap_forced) {
                const int until = compact ? (n - done) - (int)(((int64_t)n_cur * cnum) / cden) : 4096;
                dcap_e = until > 256 ? until : 256;
            }
            // all parties on one XCD while they fit there together: 32 CUs x the workgroups a CU's LDS holds
            int xcc = (fallback >= 1 ? -1 : w1_xcc_env());
            if (xcc >= 0) {
                const int per_cu = (int)((160 * 1024) / (w1_lds + 2048));
                if (w1_S > 32 * per_cu) {
                    // too many parties for one XCD: spread out as before - or (HICMI_NNCHAIN_XCD_WIDE=1, A/B) fewer, wider slices
                    int S2 = 0, slice2 = 0; size_t lds2 = 0;
                    if (w1_xcd_wide && w1_force_s <= 0 && w1_plan(n_cur, w1_cols, 32 * per_cu, 0, &S2, &slice2, &lds2) &&
                        S2 <= 32 * (int)((160 * 1024) / (lds2 + 2048))) { w1_S = S2; w1_slice = slice2; w1_lds = lds2; }
                    else xcc = -1;
                }
            }
            // test hook: HICMI_NNCHAIN_TEST_ROLLCALL=k makes epoch k's roll call (1-based) wait for a party that never comes
            const char* rc_text = getenv("HICMI_NNCHAIN_TEST_ROLLCALL");
            launch_w1(profile, w1_S, w1_slice, w1_lds, s, cur, ldw, n_cur, chain, zraw, w, dcap_e, total_steps, xcc,
                      rc_text && atoi(rc_text) == epochs + 1);
            hipLaunchKernelGGL(k_nn_settle, dim3((n_cur + 255) / 256), dim3(256), 0, s, w, n_cur);
            hipLaunchKernelGGL(k_nn_check_hashes, dim3(1), dim3(64), 0, s, w, w1_S);
            flush_needed = false;                                   // it keeps the matrix symmetric itself
        }
        else if (sliced && !plain && (n_cur <= NN_MWC_MAX || gsize)) {
            // column slices + neighbour cache + the next scan fused into the update (k_nn_epoch_mwc)
            if (!cache_valid || (refresh_below > 0 && n_cur <= refresh_below)) hipLaunchKernelGGL(k_nn_rowmin, dim3(n_cur), dim3(256), 0, s, cur, ldw, n_cur, w);
            cache_valid = true;
            tell();
            const size_t lds_c = align16((size_t)nw4 * 12 + (size_t)((n_cur + 7) & ~7) * (gsize ? 2 : 4));
            hipMemsetAsync(reinterpret_cast<unsigned char*>(w.state) + 128, 0, 1152, s);      // mailboxes
            launch_mwc(wgs_e, profile, lds_c, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps, gsize);
            hipLaunchKernelGGL(k_nn_check_replicas, dim3((NN_DMAX + 255) / 256), dim3(256), 0, s, w, wgs_e);
        }
        else if (!plain && n_cur <= NN_NC_MAX) {
            if (!cache_valid || (refresh_below > 0 && n_cur <= refresh_below)) hipLaunchKernelGGL(k_nn_rowmin, dim3(n_cur), dim3(256), 0, s, cur, ldw, n_cur, w);
            cache_valid = true;
            tell();
            const size_t lds_nc = align16((size_t)nw4 * 12 + (size_t)((n_cur + 7) & ~7) * 4);
            if (profile) hipLaunchKernelGGL(k_nn_epoch_nc<true>, dim3(1), dim3(NN_THREADS), lds_nc, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else hipLaunchKernelGGL(k_nn_epoch_nc<false>, dim3(1), dim3(NN_THREADS), lds_nc, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        }
        else {
            tell();
            if (profile) hipLaunchKernelGGL(k_nn_epoch<true>, dim3(1), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else hipLaunchKernelGGL(k_nn_epoch<false>, dim3(1), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            cache_valid = false;
        }
        epochs++;
        const int did = total_steps - done < dcap_e ? total_steps - done : dcap_e;
        done += did;
        if (done >= total_steps) break;
        if (flush_needed) hipLaunchKernelGGL(k_nn_flush, dim3((n_cur + 255) / 256, dcap), dim3(256), 0, s, cur, ldw, n_cur, w);
        const int live = n - done;
        if (compact && other && live >= 2 && (int64_t)live * cden <= (int64_t)n_cur * cnum) {
            hipLaunchKernelGGL(k_nn_translate, dim3((done - interval_start + 255) / 256), dim3(256), 0, s, zraw, interval_start,
                               done, w);
            interval_start = done;
            hipLaunchKernelGGL(k_nn_remap, dim3(1), dim3(1024), 0, s, w, n_cur, chain);
            hipLaunchKernelGGL(k_nn_compact_copy, dim3(live), dim3(256), 0, s, cur, other, ldw, w, live);
            double* t = cur; cur = other; other = t;
            n_cur = live;
        }
    }
    hipLaunchKernelGGL(k_nn_translate, dim3((total_steps - interval_start + 255) / 256), dim3(256), 0, s, zraw, interval_start,
                       total_steps, w);
    tell();
    return epochs;
}

// ---- self test of div_by_small_int against the hardware-correct '/' ------------------------------------
__global__ __launch_bounds__(256) void k_selftest_division(unsigned long long seed, int iters, unsigned long long* mismatches)
{
    unsigned long long st = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 256 + threadIdx.x + 1));
    unsigned long long bad = 0;
    for (int it = 0; it < iters; it++) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;                       // xorshift64
        const unsigned long long r1 = st;
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const unsigned long long r2 = st;
        // numerator: random mantissa, exponent spread over 2^-8 .. 2^24; divisor: integer in [2, 131071]
        const int e = (int)(r2 % 33u) - 8;
        const double a = ldexp(1.0 + (double)(r1 >> 12) * (1.0 / 4503599627370496.0), e);
        const double fs = (double)(2 + (int)((r2 >> 8) % 131070u));
        const double rcp = 1.0 / fs;
        if (div_by_small_int(a, fs, rcp) != a / fs) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

void launch_selftest_division(unsigned long long seed, int blocks, int iters, unsigned long long* d_mismatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_selftest_division, dim3(blocks), dim3(256), 0, s, seed, iters, d_mismatches);
}

// status word and phase profile live at the start of the workspace
const int* nnchain_state_ptr(void* workspace) { return reinterpret_cast<const int*>(workspace); }
const unsigned long long* nnchain_prof_ptr(void* workspace)
{
    return reinterpret_cast<const unsigned long long*>(reinterpret_cast<unsigned char*>(workspace) + 64);
}

}  // namespace hicmi
