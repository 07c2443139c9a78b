// What does one level of pointer indirection cost a small kernel on a dependent chain?  The batched kernels of the plan read their operands
// as kernel argument -> item table (device memory) -> [source-list table ->] operand; a by-value item in the kernel arguments removes one
// dependent memory round trip per level.  Chain of 1 000 launches in a captured graph, 64 workgroups x 256 threads, each launch reads 16 bytes
// per thread of what the previous one wrote and writes 16 bytes: per-launch time with 0, 1 and 2 table levels in front of the operand.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/item_indirection_bench.hip -o /tmp/iib && /tmp/iib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
struct View { u64 *p; long stride; };
struct Item { View src, dst; int first, count; const u64 *add, *mul; };
struct Src { View v; const u64 *plain; };

template <int LEVELS>
__global__ __launch_bounds__(256) void step(Item by_value, const Item *__restrict__ items, const Src *__restrict__ srcs, int which)
{
    Item it = by_value;
    if (LEVELS >= 1) it = items[which + (blockIdx.y >> 1)];
    const u64 *in = it.src.p;
    if (LEVELS >= 2) in = srcs[it.first].v.p;
    const size_t k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(in + k);
    v.x = v.x * 3 + 1, v.y = v.y * 5 + 7;
    *reinterpret_cast<ulonglong2 *>(it.dst.p + k) = v;
}

int main()
{
    const int n = 1000, wgs = 64;
    u64 *a, *b;
    CK(hipMalloc(&a, wgs * 256 * 16));
    CK(hipMalloc(&b, wgs * 256 * 16));
    CK(hipMemset(a, 1, wgs * 256 * 16));
    Item *d_items, h_items[2];
    Src *d_srcs, h_srcs[2];
    h_items[0] = Item{ { a, 0 }, { b, 0 }, 0, 1, nullptr, nullptr };
    h_items[1] = Item{ { b, 0 }, { a, 0 }, 1, 1, nullptr, nullptr };
    h_srcs[0] = Src{ { a, 0 }, nullptr }, h_srcs[1] = Src{ { b, 0 }, nullptr };
    CK(hipMalloc(&d_items, sizeof h_items));
    CK(hipMalloc(&d_srcs, sizeof h_srcs));
    CK(hipMemcpy(d_items, h_items, sizeof h_items, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_srcs, h_srcs, sizeof h_srcs, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int levels = 0; levels <= 2; levels++) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; i++) {
            const int w = i & 1;
            if (levels == 0) hipLaunchKernelGGL(step<0>, dim3(wgs), dim3(256), 0, s, h_items[w], d_items, d_srcs, w);
            if (levels == 1) hipLaunchKernelGGL(step<1>, dim3(wgs), dim3(256), 0, s, h_items[w], d_items, d_srcs, w);
            if (levels == 2) hipLaunchKernelGGL(step<2>, dim3(wgs), dim3(256), 0, s, h_items[w], d_items, d_srcs, w);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("table levels in front of the operand: %d   %.2f us per dependent launch\n", levels, best * 1e3f / n);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
