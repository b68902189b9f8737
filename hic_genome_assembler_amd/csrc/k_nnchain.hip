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
static constexpr int NN_MAXWG = 8;                       // workgroups of the column-sliced chain (k_nn_epoch_mw)

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

// workspace layout (all 16-byte aligned)
struct NNWorkspace {
    int* state;                 // [0] step [1] len [2] top [3] second [4] first_ptr [5] stop [6] n_dirty
    unsigned long long* prof;   // 6 phase totals
    void* mail;                 // k_nn_epoch_mw: 2 x NN_MAXWG 16-byte mailbox slots
    uint32_t* alive;            // nwords
    uint16_t* size;             // n
    int* gtime;                 // n: dirty time stamp of a slot in the finished epoch, -1 = clean
    int* dslot;                 // NN_DMAX
    int* dtime;                 // NN_DMAX
    int* orig;                  // n: original bin of each current slot (changes at every compaction)
    int* newidx;                // n: scratch of the compaction (old slot -> new slot, -1 = dead)
    int* oldidx;                // n: scratch of the compaction (new slot -> old slot)
};

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

size_t nnchain_workspace_bytes(int n)
{
    size_t nwords = (size_t)(n + 31) / 32;
    return 512 + align16(nwords * 4) + align16((size_t)n * 2) + 4 * align16((size_t)n * 4) + 2 * align16(NN_DMAX * 4);
}

static NNWorkspace carve(void* ws, int n)
{
    unsigned char* p = reinterpret_cast<unsigned char*>(ws);
    size_t nwords = (size_t)(n + 31) / 32;
    NNWorkspace w;
    w.state = reinterpret_cast<int*>(p);
    w.prof = reinterpret_cast<unsigned long long*>(p + 64);
    w.mail = p + 256;
    p += 512;
    w.alive = reinterpret_cast<uint32_t*>(p); p += align16(nwords * 4);
    w.size = reinterpret_cast<uint16_t*>(p); p += align16((size_t)n * 2);
    w.gtime = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.dslot = reinterpret_cast<int*>(p); p += align16(NN_DMAX * 4);
    w.dtime = reinterpret_cast<int*>(p); p += align16(NN_DMAX * 4);
    w.orig = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.newidx = reinterpret_cast<int*>(p); p += align16((size_t)n * 4);
    w.oldidx = reinterpret_cast<int*>(p);
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
    if (tid == 0) { s_stop = 0; s_done = 0; }
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
        if (PROFILE) { w.prof[0] += t_book; w.prof[1] += t_scan; w.prof[2] += t_pick; w.prof[3] += t_merge; w.prof[4] += t_upd; }
    }
}

// ---- the same chain on several workgroups (k_nn_epoch_mw) -----------------------------------------------
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
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
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
__device__ __forceinline__ u32x4 mw_pack(double v, int idx, unsigned int seq)
{
    u32x4 p;
    p.x = (unsigned int)__double2loint(v); p.y = seq;
    p.z = (unsigned int)__double2hiint(v);
    p.w = ((unsigned int)(idx < 0x1ffff ? idx : 0x1ffff)) | ((seq & 0x7fffu) << 17);
    return p;
}
__device__ __forceinline__ bool mw_ready(u32x4 p, unsigned int seq) { return p.y == seq && (p.w >> 17) == (seq & 0x7fffu); }
__device__ __forceinline__ double mw_value(u32x4 p) { return __hiloint2double((int)p.z, (int)p.x); }
__device__ __forceinline__ int mw_index(u32x4 p) { const int i = (int)(p.w & 0x1ffffu); return i == 0x1ffff ? 0x7fffffff : i; }

template <int NWG>
__global__ __launch_bounds__(NN_THREADS) void k_nn_epoch_mw(double* __restrict__ W, int64_t ld, int n,
                                                            int* __restrict__ chain_all, double* __restrict__ zraw,
                                                            NNWorkspace w, int dcap, int total_steps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_nn[];
    const int nwords = (n + 31) >> 5, nw4 = (nwords + 3) & ~3;
    uint32_t* alive = reinterpret_cast<uint32_t*>(smem_nn);
    uint32_t* smask = alive + nw4;
    uint16_t* lsize = reinterpret_cast<uint16_t*>(smask + nw4);
    __shared__ int dslot[NN_DMAX], dtime[NN_DMAX];
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int ring[256];
    __shared__ double s_dprev, s_fresh;
    __shared__ int s_x, s_prev, s_done, s_stop, s_mx, s_my, s_nx, s_ny, s_tx, s_ty, s_ey, s_nextx, s_fresh_x, s_fresh_y, s_fresh_tag,
        s_fresh_local, s_late;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    int* __restrict__ chain = chain_all + (int64_t)wg * (n + 2);           // every workgroup keeps its own copy
    u32x4* mail = reinterpret_cast<u32x4*>(w.mail);
    u32x4* fresh_slot = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(w.state) + 128);
    int step = w.state[0];
    if (step >= total_steps || w.state[5]) {
        if (tid == 0 && wg == 0) w.state[6] = 0;
        return;
    }
    // column slice of this workgroup: multiples of 64 so that mask words and 16-byte loads never straddle
    const int slice = (((n + NWG - 1) / NWG) + 63) & ~63;
    const int c0 = wg * slice < n ? wg * slice : n;
    const int c1 = c0 + slice < n ? c0 + slice : n;
    for (int i = tid; i < nwords; i += NN_THREADS) { alive[i] = w.alive[i]; smask[i] = w.alive[i]; }
    for (int i = tid; i < n; i += NN_THREADS) lsize[i] = w.size[i];
    int len = w.state[1], top = w.state[2], second = w.state[3], first_ptr = w.state[4], ring_lo = len;
    if (tid == 0) { s_stop = 0; s_done = 0; s_fresh_x = -1; s_fresh_y = -1; s_fresh_tag = 0; s_fresh_local = 0; s_late = 0; }
    // the chain prefix of the earlier epochs (saved by workgroup 0) is read from workgroup 0's copy
    const int* __restrict__ chain0 = chain_all;
    __syncthreads();
    int D = 0;
    uint32_t xbit = 0u;
    unsigned int xseq = 0u;                                  // exchanges so far (uniform)
    unsigned long long tp[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
    const bool prof = w.state[8] != 0 && wg == 0 && tid == 0;
    int lowmark = len;                                       // lane 0: chain entries below this are still the earlier epochs'

    for (; step < total_steps && D < dcap; step++) {
        if (prof) t0 = wall_clock64();
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
                xbit = smask[top >> 5] & (1u << (top & 31));
                smask[top >> 5] &= ~xbit;
            }
            __syncthreads();
            const int x = s_x, prev = s_prev;
            if (tid < D && dslot[tid] == x) s_tx = dtime[tid];
            __syncthreads();
            if (prof) { t1 = wall_clock64(); tp[0] += t1 - t0; t0 = t1; }
            const int tx = s_tx;
            const double* __restrict__ rowx = W + (int64_t)x * ld;
            const bool use_fresh = (s_fresh_x == x);
            const int fresh_y = s_fresh_y;
            if (tid == 64 && prev >= 0 && ((smask[prev >> 5] >> (prev & 31)) & 1u)) s_dprev = ld8_sc1(rowx + prev);
            ArgMin cand = {__builtin_inf(), 0x7fffffff};
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != x && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    const bool mine = d >= c0 && d < c1;
                    if (mine || d == prev) {
                        double v;
                        if (use_fresh && d == fresh_y) {                     // merged a moment ago: no exchange since
                            if (s_fresh_local) v = s_fresh;
                            else {
                                const unsigned int tag = (unsigned int)s_fresh_tag;
                                u32x4 r = ld16_sc1(fresh_slot);
                                int budget = 1000000;
                                while (!mw_ready(r, tag) && --budget > 0) { __builtin_amdgcn_s_sleep(1); r = ld16_sc1(fresh_slot); }
                                if (!mw_ready(r, tag)) s_late = 1;
                                v = mw_value(r);
                            }
                        }
                        else v = dtime[tid] > tx ? ld8_sc1(W + (int64_t)d * ld + x) : ld8_sc1(rowx + d);
                        if (mine) { cand.v = v; cand.i = d; }
                        if (d == prev) s_dprev = v;
                    }
                }
            }
            ArgMin best = {__builtin_inf(), 0x7fffffff};
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
                    if ((bits & 1u) && v.x < best.v) { best.v = v.x; best.i = j0; }
                    if ((bits & 2u) && v.y < best.v) { best.v = v.y; best.i = j0 + 1; }
                }
                if (two) {
                    const double2 v = mw_pair(r1);
                    const uint32_t bits = smask[j1 >> 5] >> (j1 & 31);
                    if ((bits & 1u) && v.x < best.v) { best.v = v.x; best.i = j1; }
                    if ((bits & 2u) && v.y < best.v) { best.v = v.y; best.i = j1 + 1; }
                }
            }
            if (cand.v < best.v || (cand.v == best.v && cand.i < best.i)) best = cand;
            best = argmin_wave(best);
            if (lane == 0) { s_v[wave] = best.v; s_i[wave] = best.i; }
            __syncthreads();
            if (prof) { t1 = wall_clock64(); tp[1] += t1 - t0; t0 = t1; }
            xseq++;
            if (wave == 0) {
                ArgMin m = {lane < 16 ? s_v[lane] : __builtin_inf(), lane < 16 ? s_i[lane] : 0x7fffffff};
                m = argmin_row16(m);                                // this slice's result, in every lane of row 0
                u32x4* slots = mail + (xseq & 1u) * NN_MAXWG;
                if (lane == 0) st16_sc1(slots + wg, mw_pack(m.v, m.i, xseq));
                ArgMin o = {__builtin_inf(), 0x7fffffff};
                int late = 0;
                if (lane < NWG) {
                    if (lane == wg) o = m;
                    else {
                        u32x4 r = ld16_sc1(slots + lane);
                        int budget = 1000000;
                        while (!mw_ready(r, xseq) && --budget > 0) { __builtin_amdgcn_s_sleep(1); r = ld16_sc1(slots + lane); }
                        if (!mw_ready(r, xseq)) late = 1;
                        o.v = mw_value(r); o.i = mw_index(r);
                    }
                }
                late = __any(late);
                m = argmin_row16(o);
                if (lane == 0) {
                    int y; double c;
                    if (prev >= 0) {
                        double dprev = s_dprev;
                        if (m.v < dprev) { y = m.i; c = m.v; } else { y = prev; c = dprev; }
                    } else { y = m.i; c = m.v; }
                    int done = (prev >= 0 && y == prev);
                    if (late || s_late || y < 0 || y >= n || ++guard > n + 2) { s_stop = 1; done = 1; }
                    else if (!done) {
                        chain[len] = y; ring[len & 255] = y;
                        if (len - 255 > ring_lo) ring_lo = len - 255;
                        second = top; top = y; len++;
                    }
                    cur = c; ybest = y;
                    smask[x >> 5] |= xbit;
                    s_done = done;
                    s_fresh_x = -1;                                 // an exchange has happened: memory is current
                }
            }
            __syncthreads();
            if (prof) { t1 = wall_clock64(); tp[2] += t1 - t0; t0 = t1; }
            if (s_done) break;
        }
        if (s_stop) break;
        if (tid == 0) {
            int xx = s_x, yy = ybest;
            len -= 2;
            if (xx > yy) { int t = xx; xx = yy; yy = t; }
            int nx = lsize[xx], ny = lsize[yy];
            if (wg == 0) {
                zraw[4 * (int64_t)step + 0] = (double)xx;
                zraw[4 * (int64_t)step + 1] = (double)yy;
                zraw[4 * (int64_t)step + 2] = cur;
                zraw[4 * (int64_t)step + 3] = (double)(nx + ny);
            }
            lsize[xx] = 0;
            lsize[yy] = (uint16_t)(nx + ny);
            alive[xx >> 5] &= ~(1u << (xx & 31));
            smask[xx >> 5] &= ~(1u << (xx & 31));
            s_mx = xx; s_my = yy; s_nx = nx; s_ny = ny; s_tx = -1; s_ty = -1; s_ey = -1;
            // entries no push of this epoch has overwritten live in workgroup 0's copy (saved by the last epoch)
            if (len < lowmark) lowmark = len;
            const int i1 = len - 1, i2 = len - 2;
            top = len >= 1 ? (i1 >= ring_lo ? ring[i1 & 255] : (i1 < lowmark ? chain0[i1] : chain[i1])) : -1;
            second = len >= 2 ? (i2 >= ring_lo ? ring[i2 & 255] : (i2 < lowmark ? chain0[i2] : chain[i2])) : -1;
            // the row the next scan visits (its d(., y') is needed before any exchange)
            int nextx = top;
            if (len == 0) {
                int fp = first_ptr;
                while (fp < n && !((alive[fp >> 5] >> (fp & 31)) & 1u)) fp++;
                nextx = fp < n ? fp : -1;
            }
            s_nextx = nextx;
        }
        __syncthreads();
        const int mx = s_mx, my = s_my;
        if (tid < D) {
            if (dslot[tid] == mx) s_tx = dtime[tid];
            if (dslot[tid] == my) { s_ty = dtime[tid]; s_ey = tid; }
        }
        __syncthreads();
        if (prof) { t1 = wall_clock64(); tp[3] += t1 - t0; t0 = t1; }
        {
            const int tmx = s_tx, tmy = s_ty, nextx = s_nextx;
            const double fx = (double)s_nx, fy = (double)s_ny, fs = (double)(s_nx + s_ny);
            const double rcp = 1.0 / fs;
            const double* __restrict__ rx = W + (int64_t)mx * ld;
            double* __restrict__ ry = W + (int64_t)my * ld;
            double dv = 0.0; int dd = -1;
            double fresh_out = __builtin_nan("");              // the owner of column nextx hands W[y'][nextx] on
            bool have_fresh = false;
            if (tid < D) {
                const int d = dslot[tid];
                if (d >= 0 && d != my && ((alive[d >> 5] >> (d & 31)) & 1u)) {
                    if (d >= c0 && d < c1) {                          // columns of this slice only: rx[d], ry[d] are its own
                        const double dxi = dtime[tid] > tmx ? ld8_sc1(W + (int64_t)d * ld + mx) : ld8_sc1(rx + d);
                        const double dyi = dtime[tid] > tmy ? ld8_sc1(W + (int64_t)d * ld + my) : ld8_sc1(ry + d);
                        dv = div_by_small_int(fx * dxi + fy * dyi, fs, rcp);
                        dd = d;
                        if (d == nextx) { fresh_out = dv; have_fresh = true; }
                    }
                }
            }
            for (int j0 = c0 + tid * 2; j0 < c1; j0 += 4 * NN_THREADS) {
                const int j1 = j0 + 2 * NN_THREADS;
                const bool two = j1 < c1;
                u32x4 ra0, rb0, ra1, rb1;
                NN_LD16_SC1(ra0, rx + j0);
                NN_LD16_SC1(rb0, ry + j0);
                NN_LD16_SC1(ra1, rx + (two ? j1 : j0));
                NN_LD16_SC1(rb1, ry + (two ? j1 : j0));
                NN_DRAIN4(ra0, rb0, ra1, rb1);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    if (h == 1 && !two) break;
                    const int j = h ? j1 : j0;
                    const double2 a = mw_pair(h ? ra1 : ra0);
                    double2 b = mw_pair(h ? rb1 : rb0);
                    const uint32_t bits = smask[j >> 5] >> (j & 31);
                    if ((bits & 1u) && j != my) b.x = div_by_small_int(fx * a.x + fy * b.x, fs, rcp);
                    if ((bits & 2u) && j + 1 != my) b.y = div_by_small_int(fx * a.y + fy * b.y, fs, rcp);
                    if (j == nextx && (bits & 1u)) { fresh_out = b.x; have_fresh = true; }
                    if (j + 1 == nextx && (bits & 2u)) { fresh_out = b.y; have_fresh = true; }
                    u32x4 pk;
                    pk.x = (unsigned int)__double2loint(b.x); pk.y = (unsigned int)__double2hiint(b.x);
                    pk.z = (unsigned int)__double2loint(b.y); pk.w = (unsigned int)__double2hiint(b.y);
                    st16_sc1(ry + j, pk);
                }
            }
            if (have_fresh && nextx != my) {
                st16_sc1(fresh_slot, mw_pack(fresh_out, 0, (unsigned int)(step + 1)));
                s_fresh = fresh_out;
            }
            // the pair stores above are inline assembly the compiler's wait-count pass does not see: drain them before
            // the barrier, or a dirty column's 8-byte store below could be overtaken by the pair store that still
            // carries its old value
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (dd >= 0) st8_sc1(ry + dd, dv);
            if (tid == 0) {
                if (s_ey >= 0) dslot[s_ey] = -1;
                dslot[D] = my; dtime[D] = step;
                smask[my >> 5] &= ~(1u << (my & 31));
                s_fresh_x = (nextx >= 0 && nextx != my) ? nextx : -1;
                s_fresh_y = my;
                s_fresh_tag = step + 1;
                s_fresh_local = (nextx >= c0 && nextx < c1) ? 1 : 0;
            }
            D++;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains before the next exchange signals
        __syncthreads();
        if (prof) { t1 = wall_clock64(); tp[4] += t1 - t0; }
    }
    __syncthreads();
    if (wg != 0) return;
    // chain entries this epoch wrote into a private copy: workgroup 0's copy is the one the next epoch reads
    for (int i = tid; i < nwords; i += NN_THREADS) w.alive[i] = alive[i];
    for (int i = tid; i < n; i += NN_THREADS) { w.size[i] = lsize[i]; w.gtime[i] = -1; }
    __syncthreads();
    if (tid < D) {
        w.dslot[tid] = dslot[tid]; w.dtime[tid] = dtime[tid];
    }
    __syncthreads();
    if (tid < D && dslot[tid] >= 0) w.gtime[dslot[tid]] = dtime[tid];
    if (tid == 0) {
        w.state[0] = step; w.state[1] = len; w.state[2] = top; w.state[3] = second; w.state[4] = first_ptr;
        w.state[5] = s_stop; w.state[6] = D;
        if (prof) for (int q = 0; q < 5; q++) w.prof[q] += tp[q];
    }
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
        int nw = -1; uint16_t sz = 0; int og = 0;
        if (i < n_cur) { nw = w.newidx[i]; sz = w.size[i]; og = w.orig[i]; }
        __syncthreads();
        if (nw >= 0) { w.size[nw] = sz; w.orig[nw] = og; }
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

// W and W2: two n x ldw buffers (W holds the distances on entry; both are scratch afterwards).
// Returns the number of k_nn_epoch launches.
int launch_nnchain(double* W, double* W2, int64_t ldw, int n, int* chain, double* zraw, void* workspace, bool profile,
                   int dcap, bool compact, hipStream_t s)
{
    int epochs = 0;
    NNWorkspace w = carve(workspace, n);
    // Column-sliced chain on several workgroups (k_nn_epoch_mw): its fixed cost per scan (one exchange) is paid back
    // by the shorter streams from about 16k live columns on (12.5 us per merge at 16k either way; at 32k 18.8 -> 12.8).
    // HICMI_NNCHAIN_WGS = 1, 2, 4 or 8 forces a width for every epoch (tests, A/B).
    const char* wgs_text = getenv("HICMI_NNCHAIN_WGS");
    const int wgs_env = wgs_text ? atoi(wgs_text) : 0;
    const int wgs = wgs_text ? (wgs_env >= 8 ? 8 : (wgs_env >= 4 ? 4 : (wgs_env >= 2 ? 2 : 1))) : 8;
    const int mw_from = wgs_text ? 64 * wgs : 20000;          // live columns from which an epoch runs sliced
    if (dcap < 1) dcap = 1;
    if (dcap > NN_DMAX) dcap = NN_DMAX;
    hipLaunchKernelGGL(k_nn_init, dim3(64), dim3(256), 0, s, w, n);
    if (profile) { static const int one = 1; hipMemcpyAsync(w.state + 8, &one, sizeof(int), hipMemcpyHostToDevice, s); }
    const int total_steps = n - 1;
    int n_cur = n, done = 0, interval_start = 0;
    double *cur = W, *other = W2;
    {
        const int nwords = (n + 31) / 32, nw4 = (nwords + 3) & ~3;
        size_t lds_max = align16((size_t)nw4 * 8 + (size_t)n * 2);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch_mw<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch_mw<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_nn_epoch_mw<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    }
    while (done < total_steps) {
        const int nwords = (n_cur + 31) / 32, nw4 = (nwords + 3) & ~3;
        const size_t lds = align16((size_t)nw4 * 8 + (size_t)n_cur * 2);
        if (profile && !(wgs_text && wgs > 1)) hipLaunchKernelGGL(k_nn_epoch<true>, dim3(1), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        else if (wgs > 1 && n_cur >= mw_from) {
            hipMemsetAsync(reinterpret_cast<unsigned char*>(w.state) + 128, 0, 384, s);      // hand-off slot + mailboxes
            if (wgs == 2) hipLaunchKernelGGL(k_nn_epoch_mw<2>, dim3(2), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else if (wgs == 4) hipLaunchKernelGGL(k_nn_epoch_mw<4>, dim3(4), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
            else hipLaunchKernelGGL(k_nn_epoch_mw<8>, dim3(8), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        }
        else hipLaunchKernelGGL(k_nn_epoch<false>, dim3(1), dim3(NN_THREADS), lds, s, cur, ldw, n_cur, chain, zraw, w, dcap, total_steps);
        epochs++;
        const int did = total_steps - done < dcap ? total_steps - done : dcap;
        done += did;
        if (done >= total_steps) break;
        hipLaunchKernelGGL(k_nn_flush, dim3((n_cur + 255) / 256, dcap), dim3(256), 0, s, cur, ldw, n_cur, w);
        const int live = n - done;
        if (compact && other && live >= 2 && (int64_t)live * 4 <= (int64_t)n_cur * 3) {
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
