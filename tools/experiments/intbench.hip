// Microbenchmark: integer-multiply issue rates on gfx950 (grounds the VALU ceiling of the NTT butterfly).
// hipcc --offload-arch=gfx950 -O3 tools/experiments/intbench.hip -o /tmp/intbench && /tmp/intbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(u64 *out, u32 a, u32 b)
{
    u64 x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    u32 y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { // v_mad_u64_u32, 4 independent chains
            x0 = (u64)(u32)x0 * a + x0; x1 = (u64)(u32)x1 * a + x1; x2 = (u64)(u32)x2 * a + x2; x3 = (u64)(u32)x3 * a + x3;
        } else if (MODE == 1) { // v_mul_lo_u32
            y0 = y0 * a + b; y1 = y1 * a + b; y2 = y2 * a + b; y3 = y3 * a + b;
        } else if (MODE == 2) { // v_mul_hi_u32
            y0 = __umulhi(y0, a) ^ b; y1 = __umulhi(y1, a) ^ b; y2 = __umulhi(y2, a) ^ b; y3 = __umulhi(y3, a) ^ b;
        } else if (MODE == 3) { // 32-bit add (full-rate reference)
            y0 = (y0 + a) ^ b; y1 = (y1 + a) ^ b; y2 = (y2 + a) ^ b; y3 = (y3 + a) ^ b;
        } else if (MODE == 4) { // 24-bit mad
            y0 = __umul24(y0, a) + b; y1 = __umul24(y1, a) + b; y2 = __umul24(y2, a) + b; y3 = __umul24(y3, a) + b;
        } else if (MODE == 5) { // 64-bit add (v_add_co + v_addc)
            x0 += (x0 >> 3) ^ a; x1 += (x1 >> 3) ^ a; x2 += (x2 >> 3) ^ a; x3 += (x3 >> 3) ^ a;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3;
}
template <int MODE>
void run(const char *name, int ops_per_iter)
{
    u64 *d;
    hipMalloc(&d, 256 * 8 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8; // 8 blocks/CU = 32 waves/CU
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 678u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * ITERS * ops_per_iter;
    printf("%-28s %8.3f ms  %8.2f Tops/s  -> %.2f lane-ops/clk/CU @2.4GHz\n", name, ms, ops / ms / 1e9,
           ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
}
int main()
{
    run<0>("v_mad_u64_u32", 4);
    run<1>("v_mul_lo_u32 (+add)", 4);
    run<2>("v_mul_hi_u32 (+xor)", 4);
    run<3>("v_add_u32 + xor", 8);
    run<4>("v_mad_u32_u24", 4);
    run<5>("64-bit add + shift + xor", 4);
    return 0;
}
