// all-to-all exchange among S single-wave workgroups placed on ONE XCD (the blocks b with b % 8 == 0 of a grid of 8 S:
// round-robin placement, checked through HW_REG_XCC_ID) against the same exchange spread over all eight; stores with and
// without sc1 (a plain store stays in the XCD's L2: only valid when every party reports the same XCC id)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool SC1> __device__ __forceinline__ void st16(void* p, u32x4 v)
{
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void poll2(u32x4& a, u32x4& b, const u32x4* pa, const u32x4* pb)
{
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb) : "memory");
}
template <bool SC1>
__global__ __launch_bounds__(64) void k_allx(u32x4* mail, int S, int stride, int iters, unsigned long long* out, int* xcc)
{
    if (blockIdx.x % stride) return;
    const int wg = blockIdx.x / stride, lane = threadIdx.x;
    if (wg >= S) return;
    if (lane == 0) xcc[wg] = (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf);   // HW_REG_XCC_ID, bits 3:0
    unsigned long long t0 = wall_clock64();
    for (unsigned it = 1; it <= (unsigned)iters; it++) {
        u32x4 v = {it, it, it, it};
        u32x4* base = mail + (size_t)(it & 1u) * 64 * 128;
        if (lane < S) { st16<SC1>(base + (size_t)lane * 128 + wg, v); st16<SC1>(base + (size_t)lane * 128 + 64 + wg, v); }
        const u32x4* sa = base + (size_t)wg * 128 + lane; const u32x4* sb = sa + 64;
        if (lane < S && lane != wg) {
            u32x4 a, b; int budget = 2000000;
            do { poll2(a, b, sa, sb); } while (!(a.x == it && a.w == it && b.x == it && b.w == it) && --budget > 0);
            if (budget <= 0) { if (lane == (wg + 1) % S) out[64 + wg] = it; break; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long t1 = wall_clock64();
    if (lane == 0) out[wg] = t1 - t0;
}
template <bool SC1> static void run(const char* name, int S, int stride, u32x4* mail, unsigned long long* out, int* xcc)
{
    const int iters = 20000;
    (void)hipMemset(mail, 0, 2 * 64 * 128 * 16); (void)hipMemset(out, 0, 8 * 128);
    hipLaunchKernelGGL(k_allx<SC1>, dim3(S * stride), dim3(64), 0, 0, mail, S, stride, iters, out, xcc);
    (void)hipDeviceSynchronize();
    unsigned long long h[128]; int x[64];
    (void)hipMemcpy(h, out, 8 * 128, hipMemcpyDeviceToHost); (void)hipMemcpy(x, xcc, 4 * 64, hipMemcpyDeviceToHost);
    int same = 1, stuck = 0; for (int i = 1; i < S; i++) same &= x[i] == x[0];
    for (int i = 0; i < S; i++) stuck |= h[64 + i] != 0;
    printf("%-22s S=%2d stride %d: %.3f us per exchange round (%s%s)\n", name, S, stride, h[0] / 100.0 / iters,
           same ? "one XCD" : "several XCDs", stuck ? ", TIMEOUT" : "");
}
int main()
{
    u32x4* mail; unsigned long long* out; int* xcc;
    (void)hipMalloc(&mail, 2 * 64 * 128 * 16); (void)hipMalloc(&out, 8 * 128); (void)hipMalloc(&xcc, 4 * 64);
    for (int S : {8, 16, 32, 64}) {
        run<true>("sc1 stores", S, 1, mail, out, xcc);
        run<true>("sc1 stores", S, 8, mail, out, xcc);
        run<false>("plain stores", S, 8, mail, out, xcc);
    }
    return 0;
}
