// Exploration: explicit hipGraph construction from per-op single-stream captures (development aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
__global__ void k(int *p, int v) { atomicAdd(&p[threadIdx.x], v); }
int main(int argc, char **argv)
{
    int nops = argc > 1 ? atoi(argv[1]) : 2000, width = argc > 2 ? atoi(argv[2]) : 8;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipGraph_t main_g; CK(hipGraphCreate(&main_g, 0));
    std::vector<hipGraphNode_t> tail(nops); // last node of each op
    srand(1);
    long total_nodes = 0, expect = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nops; i++) {
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        int nk = 1 + rand() % 4;
        for (int j = 0; j < nk; j++) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d, i + j); expect += i + j; }
        hipGraph_t g; CK(hipStreamEndCapture(s, &g));
        size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
        // dependencies of the op: up to 3 earlier ops within a window
        std::vector<hipGraphNode_t> deps;
        for (int w = 0; w < 3 && i > 0; w++) { int o = i - 1 - rand() % (i < width ? i : width); bool dup = false; for (auto x : deps) dup |= (x == tail[o]); if (!dup) deps.push_back(tail[o]); }
        hipGraphNode_t prev = nullptr;
        for (size_t n = 0; n < nn; n++) {
            hipKernelNodeParams p; CK(hipGraphKernelNodeGetParams(nodes[n], &p));
            hipGraphNode_t nw;
            if (n == 0) CK(hipGraphAddKernelNode(&nw, main_g, deps.data(), deps.size(), &p));
            else CK(hipGraphAddKernelNode(&nw, main_g, &prev, 1, &p));
            prev = nw; total_nodes++;
        }
        tail[i] = prev;
        CK(hipGraphDestroy(g));
    }
    auto t1 = std::chrono::steady_clock::now();
    hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, main_g, nullptr, nullptr, 0));
    auto t2 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    auto t3 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    auto t4 = std::chrono::steady_clock::now();
    int h[64]; CK(hipMemcpy(h, d, 256, hipMemcpyDeviceToHost));
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    printf("nodes %ld build %.1f ms instantiate %.1f ms launch1 %.2f ms launch2 %.2f ms  check %s (%d vs %ld)\n", total_nodes, ms(t0, t1), ms(t1, t2),
           ms(t2, t3), ms(t3, t4), h[0] == 2 * expect ? "OK" : "BAD", h[0], 2 * expect);
    return 0;
}
