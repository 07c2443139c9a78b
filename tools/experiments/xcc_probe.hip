// Which XCD does workgroup i of a 1-D grid land on?  (gfx950: 8 XCDs x 32 CUs; s_getreg_b32 HW_REG_XCC_ID)
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void probe(unsigned *xcc, unsigned *cu, int spin)
{
    if (threadIdx.x == 0) {
        unsigned x, h;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
        xcc[blockIdx.x] = x & 0xf;
        cu[blockIdx.x] = h;
    }
    // keep the workgroup resident for a while so that later ones cannot reuse its slot
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(10);
}

int main()
{
    const int n = 8192;
    unsigned *dx, *dc;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dc, n * 4);
    for (int spin : { 0, 200 }) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, dx, dc, spin);
        std::vector<unsigned> x(n), c(n);
        hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
        int ok = 0;
        for (int i = 0; i < n; i++) ok += (x[i] == (unsigned)(i % 8));
        printf("spin %d: xcc == id %% 8 for %d of %d workgroups; first 24:", spin, ok, n);
        for (int i = 0; i < 24; i++) printf(" %u", x[i]);
        printf("\n");
    }
    return 0;
}
