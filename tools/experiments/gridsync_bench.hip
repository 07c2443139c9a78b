// Cost of a grid-wide barrier on gfx950 (cooperative launch, all workgroups resident) against the ~2.8 us a dependent kernel
// launch costs: decides whether a persistent "phase interpreter" kernel could beat launch chains for single-ciphertext steps.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/gridsync_bench.hip -o /tmp/gridsync && timeout 60 /tmp/gridsync
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
namespace cg = cooperative_groups;

__global__ void cg_sync_kernel(int iters, unsigned long long *out)
{
    cg::grid_group g = cg::this_grid();
    unsigned long long acc = 0;
    for (int i = 0; i < iters; i++) {
        acc += i;
        g.sync();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = acc;
}

// hand-rolled barrier: one device-scope atomic per workgroup, monotonic counter
__global__ void atomic_sync_kernel(int iters, unsigned int *counter, unsigned long long *out)
{
    unsigned long long acc = 0;
    unsigned int target = 0;
    for (int i = 0; i < iters; i++) {
        acc += i;
        target += gridDim.x;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            while (__atomic_load_n(counter, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = acc;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));               \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned long long *out;
    unsigned int *counter;
    CK(hipMalloc(&out, 8));
    CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        const int blocks = prop.multiProcessorCount * per_cu, threads = 256;
        int iters = 2000;
        void *args[] = { &iters, &out };
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel((void *)cg_sync_kernel, dim3(blocks), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("cooperative_groups grid.sync, %4d workgroups x %d threads: %.2f us per barrier\n", blocks, threads, ms * 1e3 / iters);
        void *args2[] = { &iters, &counter, &out };
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(counter, 0, 4));
            CK(hipEventRecord(e0, 0));
            CK(hipLaunchCooperativeKernel((void *)atomic_sync_kernel, dim3(blocks), dim3(threads), args2, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("atomic counter barrier,       %4d workgroups x %d threads: %.2f us per barrier\n", blocks, threads, ms * 1e3 / iters);
    }
    return 0;
}
