// all-to-all exchange among S single-wave workgroups: time per round, by mailbox layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void poll2(u32x4& a, u32x4& b, const u32x4* pa, const u32x4* pb)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
}
// MODE 0: shared slots [parity][2][S] (every reader polls the same 2*S slots); MODE 1: inboxes [parity][reader][2][S] (every writer
// pushes its two slots into every reader's inbox); MODE 2: like 0 but ONE slot per peer
// MODE 3: inboxes for up to 128 parties (lane p polls writers p and p + 64: four loads per poll)
__global__ __launch_bounds__(64) void k_allx128(u32x4* mail, int S, int iters, unsigned long long* out)
{
    const int wg = blockIdx.x, lane = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (unsigned it = 1; it <= (unsigned)iters; it++) {
        u32x4 v = {it, it, it, it};
        u32x4* base = mail + (size_t)(it & 1u) * 128 * 256;           // [reader][2 slots][128 writers]
        for (int r = lane; r < S; r += 64) { st16(base + (size_t)r * 256 + wg, v); st16(base + (size_t)r * 256 + 128 + wg, v); }
        const u32x4* mine = base + (size_t)wg * 256;
        for (int p = lane; p < S; p += 64) {
            if (p == wg) continue;
            u32x4 a, b; int budget = 1000000;
            do { poll2(a, b, mine + p, mine + 128 + p); } while (!(a.x == it && a.w == it && b.x == it && b.w == it) && --budget > 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = wall_clock64();
    if (lane == 0) out[wg] = t1 - t0;
}
template <int MODE>
__global__ __launch_bounds__(64) void k_allx(u32x4* mail, int S, int iters, unsigned long long* out, unsigned long long* polls)
{
    const int wg = blockIdx.x, lane = threadIdx.x;
    unsigned long long np = 0;
    unsigned long long t0 = wall_clock64();
    for (unsigned it = 1; it <= (unsigned)iters; it++) {
        u32x4 v = {it, it, it, it};
        const u32x4 *sa, *sb;
        if (MODE == 1) {
            u32x4* base = mail + (size_t)(it & 1u) * 64 * 128;
            if (lane < S) { st16(base + (size_t)lane * 128 + wg, v); st16(base + (size_t)lane * 128 + 64 + wg, v); }
            sa = base + (size_t)wg * 128 + lane; sb = sa + 64;
        } else {
            u32x4* base = mail + (size_t)(it & 1u) * 128;
            if (lane == 0) st16(base + wg, v);
            if (lane == 1 && MODE == 0) st16(base + 64 + wg, v);
            sa = base + lane; sb = MODE == 0 ? sa + 64 : sa;
        }
        if (lane < S && lane != wg) {
            u32x4 a, b; int budget = 1000000;
            do { poll2(a, b, sa, sb); np++; } while (!(a.x == it && a.w == it && b.x == it && b.w == it) && --budget > 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = wall_clock64();
    if (lane == 0) out[wg] = t1 - t0;
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) { unsigned long long x = ((unsigned long long)__builtin_amdgcn_readlane((int)(np >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)np, l); if (x > m) m = x; }
    if (lane == 0) polls[wg] = m;
}
template <int MODE> static void run(const char* name, int S, u32x4* mail, unsigned long long* out, unsigned long long* polls)
{
    const int iters = 20000;
    (void)hipMemset(mail, 0, 2 * 64 * 128 * 16);
    hipLaunchKernelGGL(k_allx<MODE>, dim3(S), dim3(64), 0, 0, mail, S, iters, out, polls);
    (void)hipDeviceSynchronize();
    unsigned long long h[64], p[64];
    (void)hipMemcpy(h, out, 8 * 64, hipMemcpyDeviceToHost); (void)hipMemcpy(p, polls, 8 * 64, hipMemcpyDeviceToHost);
    printf("%-34s S=%2d: %.3f us per exchange round, %.2f polls (slowest peer, wg 0)\n", name, S, h[0] / 100.0 / iters, (double)p[0] / iters);
}
int main()
{
    u32x4* mail; unsigned long long *out, *polls;
    (void)hipMalloc(&mail, 2 * 128 * 256 * 16); (void)hipMalloc(&out, 8 * 128); (void)hipMalloc(&polls, 8 * 128);
    for (int S : {2, 8, 16, 32, 64}) {
        run<0>("shared slots, two per peer", S, mail, out, polls);
        run<2>("shared slots, one per peer", S, mail, out, polls);
        run<1>("inboxes (writers push), two slots", S, mail, out, polls);
    }
    for (int S : {64, 96, 128}) {
        const int iters = 20000;
        (void)hipMemset(mail, 0, 2 * 128 * 256 * 16);
        hipLaunchKernelGGL(k_allx128, dim3(S), dim3(64), 0, 0, mail, S, iters, out);
        (void)hipDeviceSynchronize();
        unsigned long long h[128];
        (void)hipMemcpy(h, out, 8 * 128, hipMemcpyDeviceToHost);
        printf("inboxes, two peers per polling lane      S=%3d: %.3f us per exchange round\n", S, h[0] / 100.0 / iters);
    }
    return 0;
}
