// Experiment (round 4, DESIGN.md section 4 "Scheduling notes"): can a HIP graph on this runtime run INDEPENDENT chains of small kernels
// concurrently, and what does a dependency that crosses chains cost?  W chains of L dependent kernels (each ~5 us of ALU work on 64
// workgroups: a single-ciphertext phase kernel's shape), built (a) explicitly -- hipGraphAddKernelNode with the chain edges only, (b) the same
// plus a cross edge between the chains every X kernels (a "wave" join like the plan's), (c) captured from W streams with event fork / join.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/graph_overlap.hip -o /tmp/graph_overlap && /tmp/graph_overlap [L=400]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e_)); exit(1);} } while (0)
__global__ void spin(unsigned *p, int iters)
{
    unsigned v = p[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    p[blockIdx.x * 256 + threadIdx.x] = v;
}
static double run(hipGraphExec_t ge, hipStream_t s, int reps = 5)
{
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    double best = 1e9;
    for (int r = 0; r < reps; r++) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    return best;
}
int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 400, iters = argc > 2 ? atoi(argv[2]) : 2500;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *buf[8];
    for (auto &b : buf) { CK(hipMalloc(&b, 64 * 256 * 4)); CK(hipMemset(b, 1, 64 * 256 * 4)); }
    // one kernel's duration
    { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, buf[0], iters);
      CK(hipEventRecord(e0, s)); for (int i = 0; i < 20; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, buf[0], iters); CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("kernel: %.2f us each back to back on a stream (64 workgroups x 256 threads)\n", ms * 50); }
    for (int W : { 1, 2, 4, 8 }) {
        for (int X : { 0, 1, 4 }) { // cross edge every X kernels (0: none)
            if (W == 1 && X) continue;
            hipGraph_t g; CK(hipGraphCreate(&g, 0));
            std::vector<hipGraphNode_t> prev(W, nullptr);
            for (int i = 0; i < L; i++) {
                std::vector<hipGraphNode_t> now(W);
                for (int w = 0; w < W; w++) {
                    void *args[2] = { &buf[w], (void *)&iters };
                    hipKernelNodeParams p{}; p.func = (void *)spin; p.gridDim = dim3(64); p.blockDim = dim3(256); p.kernelParams = args;
                    std::vector<hipGraphNode_t> deps;
                    if (prev[w]) deps.push_back(prev[w]);
                    if (X && i && i % X == 0 && prev[(w + 1) % W] && W > 1) deps.push_back(prev[(w + 1) % W]); // the neighbour chain's previous kernel
                    CK(hipGraphAddKernelNode(&now[w], g, deps.data(), deps.size(), &p));
                }
                prev = now;
            }
            hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            const double ms = run(ge, s);
            printf("explicit graph: %d chain(s) x %d kernels, cross edge every %d: %.3f ms = %.2f us per chain step (%d kernels in flight per step)\n", W, L, X, ms,
                   ms * 1e3 / L, W);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    // captured from two streams with a fork / join around every step (what the plan records today)
    {
        hipStream_t a; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        std::vector<hipEvent_t> ev(2 * L);
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < L; i++) {
            CK(hipEventRecord(ev[2 * i], s)); CK(hipStreamWaitEvent(a, ev[2 * i], 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, buf[0], iters);
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, a, buf[1], iters);
            CK(hipEventRecord(ev[2 * i + 1], a)); CK(hipStreamWaitEvent(s, ev[2 * i + 1], 0));
        }
        hipGraph_t g; CK(hipStreamEndCapture(s, &g));
        hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double ms = run(ge, s);
        printf("captured from 2 streams, fork / join around every step: %.3f ms = %.2f us per step\n", ms, ms * 1e3 / L);
    }
    return 0;
}
