// Self-check of dacapo_amd/csrc/lane_xchg.hpp on the device: for every distance D, lane_swap<D> must exchange x1 of lane L
// with x0 of lane L ^ D (bit D of L clear) and leave everything else in place.
//   hipcc --offload-arch=gfx950 -O3 -I dacapo_amd/csrc tools/experiments/laneswap_test.hip -o /tmp/laneswap_test && /tmp/laneswap_test
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "lane_xchg.hpp"

template <int D>
__global__ void k(uint64_t *out)
{
    const uint64_t l = threadIdx.x;
    uint64_t x0 = (l << 8) | 0xA000000000000000ull, x1 = (l << 8) | 0xB000000000000001ull;
    dacapo::lane_swap<D>(x0, x1);
    out[2 * l] = x0, out[2 * l + 1] = x1;
}

template <int D>
int check()
{
    uint64_t *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k<D>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(d);
    int bad = 0;
    for (uint64_t l = 0; l < 64; l++) {
        const uint64_t own0 = (l << 8) | 0xA000000000000000ull, own1 = (l << 8) | 0xB000000000000001ull;
        const uint64_t p = l ^ D, p0 = (p << 8) | 0xA000000000000000ull, p1 = (p << 8) | 0xB000000000000001ull;
        const uint64_t want0 = (l & D) ? p1 : own0, want1 = (l & D) ? own1 : p0;
        if (h[2 * l] != want0 || h[2 * l + 1] != want1) bad++;
    }
    printf("lane_swap<%2d>: %s\n", D, bad ? "WRONG" : "ok");
    return bad;
}

int main()
{
    int bad = check<1>() + check<2>() + check<4>() + check<8>() + check<16>() + check<32>();
    return bad ? 1 : 0;
}
