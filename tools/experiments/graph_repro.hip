// Repro/exploration: multi-stream capture patterns on ROCm 7.2 (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k(int *p) { p[threadIdx.x] += 1; }
int main(int argc, char **argv)
{
    int nlanes = argc > 1 ? atoi(argv[1]) : 3, nops = argc > 2 ? atoi(argv[2]) : 200, mode = argc > 3 ? atoi(argv[3]) : 0;
    std::vector<hipStream_t> s(nlanes);
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 4096));
    std::vector<hipEvent_t> ev(nops + 3 * nlanes + 8);
    for (auto &evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    std::vector<int> last_ev(nlanes, -1);
    srand(1);
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed));
    if ((mode & 1) == 0) { // upfront fork
        CK(hipEventRecord(ev[nops], s[0]));
        for (int l = 1; l < nlanes; l++) CK(hipStreamWaitEvent(s[l], ev[nops], 0));
    }
    std::vector<bool> used(nlanes, (mode & 1) == 0);
    used[0] = true;
    for (int i = 0; i < nops; i++) {
        int lane = rand() % nlanes;
        if (!used[lane]) { used[lane] = true; CK(hipEventRecord(ev[nops + 1 + lane], s[0])); CK(hipStreamWaitEvent(s[lane], ev[nops + 1 + lane], 0)); }
        // wait on up to 2 other lanes' last events
        for (int w = 0; w < ((mode & 2) ? 0 : 2); w++) {
            int o = rand() % nlanes;
            if (o != lane && last_ev[o] >= 0) CK(hipStreamWaitEvent(s[lane], ev[last_ev[o]], 0));
        }
        int nk = 1 + rand() % 3;
        for (int j = 0; j < nk; j++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[lane], d);
        CK(hipEventRecord(ev[i], s[lane]));
        last_ev[lane] = i;
    }
    for (int l = 1; l < nlanes; l++) if (used[l]) { CK(hipEventRecord(ev[nops + 1 + nlanes + l], s[l])); CK(hipStreamWaitEvent(s[0], ev[nops + 1 + nlanes + l], 0)); }
    printf("ending capture\n"); fflush(stdout);
    hipGraph_t g; CK(hipStreamEndCapture(s[0], &g));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    printf("captured %zu nodes\n", nn); fflush(stdout);
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("instantiated\n"); fflush(stdout);
    CK(hipGraphLaunch(ge, s[0])); CK(hipStreamSynchronize(s[0]));
    printf("launched ok\n");
    return 0;
}
