// ping-pong between two single-wave workgroups: round-trip time by store flavour and placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
template <int MODE>   // 0: sc1 store + sc1 load; 1: plain store + sc1 load; 2: plain store + sc0 sc1 load ; 3: nt store + sc1 load
__global__ __launch_bounds__(64) void k_pp(u32x4* slots, int peer_a, int peer_b, int iters, unsigned long long* out, unsigned* xcc)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = xcc_id();
    if (b != peer_a && b != peer_b) return;
    const int me = b == peer_a ? 0 : 1;
    u32x4* mine = slots + me * 8, *theirs = slots + (1 - me) * 8;    // 128 B apart
    unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; it++) {
        if (me == 0) {
            u32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
            if (threadIdx.x == 0) {
                if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine), "v"(v) : "memory");
                else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(mine), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine), "v"(v) : "memory");
            }
        }
        // wait for the peer's it (me==1 waits for A's store first, then answers)
        u32x4 r; int budget = 10000000;
        do {
            if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(theirs) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(theirs) : "memory");
        } while (__builtin_amdgcn_readfirstlane((int)r.x) != it && --budget > 0);
        if (budget <= 0) { if (threadIdx.x == 0) out[2 + me] = 0xdeadull; return; }
        if (me == 1) {
            u32x4 v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it};
            if (threadIdx.x == 0) {
                if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine), "v"(v) : "memory");
                else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(mine), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(mine), "v"(v) : "memory");
            }
        }
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[me] = t1 - t0; out[2 + me] = 1; }
}
template <int MODE> static void run(const char* name, int pa, int pb, u32x4* slots, unsigned long long* out, unsigned* xcc)
{
    const int iters = 20000;
    hipMemset(slots, 0, 4096); hipMemset(out, 0, 64);
    hipLaunchKernelGGL(k_pp<MODE>, dim3(64), dim3(64), 0, 0, slots, pa, pb, iters, out, xcc);
    hipDeviceSynchronize();
    unsigned long long h[4]; unsigned hx[64];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost);
    printf("%-28s blocks %2d(xcc %u) <-> %2d(xcc %u): %s  round trip %.3f us (one hop %.3f us)\n", name, pa, hx[pa], pb, hx[pb],
           (h[2] == 1 && h[3] == 1) ? "ok" : "TIMEOUT", h[0] / 100.0 / iters, h[0] / 200.0 / iters);
}
int main()
{
    u32x4* slots; unsigned long long* out; unsigned* xcc;
    hipMalloc(&slots, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, 256);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("sc1 store, sc1 load", 0, 8, slots, out, xcc);
        run<0>("sc1 store, sc1 load", 0, 1, slots, out, xcc);
        run<1>("plain store, sc1 load", 0, 8, slots, out, xcc);
        run<1>("plain store, sc1 load", 0, 1, slots, out, xcc);
        run<2>("plain store, sc0 sc1 load", 0, 8, slots, out, xcc);
        run<3>("nt store, sc1 load", 0, 8, slots, out, xcc);
    }
    return 0;
}
