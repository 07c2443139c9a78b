// Cost of an XCD-LOCAL cluster barrier on gfx950 in the LATENCY regime (few clusters resident, nothing else on the chip), beside what a
// dependent kernel launch costs in a captured graph.  Answers round 5's review item 2(a): tools/experiments/gridsync_bench.hip measured a
// DEVICE-wide barrier (one counter polled by 256-512 workgroups across 8 XCDs, ~14 us) and profiles/r02_experiments.txt item 11 a cluster
// exchange in the THROUGHPUT regime (4 096 limbs); neither is the regime of a 1-3 prime rescale / key switch, where a launch holds 2-8 limbs.
//
// Placement: workgroup i of a 1-D grid runs on XCD i mod 8 (tools/experiments/xcc_probe.hip), so cluster c of XCD x owns grid indices
// 8 (C c + t) + x, t < C.  Every workgroup re-checks HW_REG_XCC_ID.  The counter is a relaxed agent-scope atomic (executed in the XCD's L2,
// no sc1, no buffer_wbl2 / buffer_inv); the payload is written with plain stores (write-through to L2), s_waitcnt vmcnt(0) before the
// arrive, and read back with sc1 loads (miss the CU's L1, hit the XCD's L2) -- nothing is invalidated or written back.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/cluster_barrier_bench.hip -o /tmp/cluster_barrier && timeout 120 /tmp/cluster_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                                                                              \
    do {                                                                                                                                   \
        hipError_t e_ = (x);                                                                                                               \
        if (e_ != hipSuccess) {                                                                                                            \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                                                                          \
            exit(1);                                                                                                                       \
        }                                                                                                                                  \
    } while (0)

typedef unsigned long long u64;

// POLL: 0 = fetch_add(0) (an RMW at L2), 1 = sc1 load
template <int POLL>
__device__ __forceinline__ bool cluster_wait(int *cnt, int target, int sleep)
{
    int spins = 0;
    for (;;) {
        int v = POLL == 0 ? __hip_atomic_fetch_add(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= target) return true;
        if (sleep) __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) return false; // bounded: a lost partner must not hang the box
    }
}

// `iters` barriers back to back; PAYLOAD words of u64 per thread are written before and a partner's are read after each barrier.
// SPREAD: the members of a cluster are CONSECUTIVE grid indices, i.e. one per XCD (what an XCD-oblivious grid would do) -- same code, the
// counter and the payload then cross XCDs.  WT: the payload is written with agent-scope (sc1, write-through) stores instead of plain ones.
template <int POLL, int PAYLOAD, bool SPREAD = false, bool WT = false>
__global__ __launch_bounds__(1024) void barrier_loop(int C, int iters, int sleep, int *__restrict__ cnt, u64 *__restrict__ buf, int *__restrict__ err,
                                                      u64 *__restrict__ sink)
{
    const unsigned id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const int t = SPREAD ? (int)(id % (unsigned)C) : (int)(k % (unsigned)C), cl = SPREAD ? (int)(id / (unsigned)C) : (int)(k / (unsigned)C) * 8 + (int)xcd; // member, cluster
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if ((xcc & 0xfu) != xcd) atomicOr(err, 1);
        if (id < 8) err[1 + id] = (int)(xcc & 0xfu); // where the first eight workgroups of this launch ran
    }
    int *my = cnt + cl * 32; // one 128-byte line per cluster
    const size_t per_wg = (size_t)blockDim.x * (PAYLOAD ? PAYLOAD : 1);
    u64 *base = buf + (size_t)cl * C * per_wg * 2; // two halves, alternating per iteration (no WAR on the half being read)
    u64 acc = 0;
    int target = 0;
    __shared__ int dead; // a timed-out barrier ends the loop: one bounded spin per launch, not one per iteration
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    for (int i = 0; i < iters && !dead; i++) {
        if (PAYLOAD) {
            u64 *w = base + (size_t)(i & 1) * C * per_wg + (size_t)t * per_wg;
#pragma unroll
            for (int j = 0; j < PAYLOAD; j++) {
                if (WT)
                    __hip_atomic_store(w + j * blockDim.x + threadIdx.x, ((u64)i << 32) | (u64)(t * 1024 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    w[j * blockDim.x + threadIdx.x] = ((u64)i << 32) | (u64)(t * 1024 + j);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        target += C;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(my, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!cluster_wait<POLL>(my, target, sleep)) atomicOr(err, 2), dead = 1;
        }
        __syncthreads();
        if (PAYLOAD) {
            // the transposed read of a phase exchange: thread reads word j of partner (t + 1 + j) % C
#pragma unroll
            for (int j = 0; j < PAYLOAD; j++) {
                const int p = (t + 1 + j) % C;
                const u64 *r = base + (size_t)(i & 1) * C * per_wg + (size_t)p * per_wg;
                u64 v = __hip_atomic_load(r + j * blockDim.x + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (((u64)i << 32) | (u64)(p * 1024 + j))) atomicOr(err, 4); // stale or missing data
                acc += v;
            }
        }
    }
    if (acc == 0x12345) *sink = acc;
}

// DEVICE-wide barrier over G workgroups (all resident).  FLAT: one counter, every workgroup adds and polls it.  Hierarchical: a workgroup
// arrives at its XCD's counter (XCC_ID read at run time; G / 8 arrivals expected per XCD); the arrival that completes an XCD adds to the global
// counter, polls it, and then releases its XCD by bumping that XCD's release word, which the others poll: 8 cross-XCD pollers instead of G.
// Payload as above, always written with sc1 stores and read with sc1 loads from a workgroup half the grid away.
template <bool FLAT, int PAYLOAD>
__global__ __launch_bounds__(1024) void device_barrier_loop(int iters, int *__restrict__ cnt, u64 *__restrict__ buf, int *__restrict__ err, u64 *__restrict__ sink)
{
    const int G = gridDim.x, id = blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    int *glob = cnt, *arrive = cnt + 32 * (1 + xcc), *release = cnt + 32 * (9 + xcc);
    const size_t per_wg = (size_t)blockDim.x * (PAYLOAD ? PAYLOAD : 1);
    u64 acc = 0;
    __shared__ int dead;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    for (int i = 0; i < iters && !dead; i++) {
        if (PAYLOAD) {
            u64 *w = buf + (size_t)(i & 1) * G * per_wg + (size_t)id * per_wg;
#pragma unroll
            for (int j = 0; j < PAYLOAD; j++)
                __hip_atomic_store(w + j * blockDim.x + threadIdx.x, ((u64)i << 32) | (u64)(id * 16 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FLAT) {
                __hip_atomic_fetch_add(glob, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!cluster_wait<1>(glob, (i + 1) * G, 0)) atomicOr(err, 2), dead = 1;
            } else {
                const int per_xcd = G / 8;
                const int a = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == (i + 1) * per_xcd - 1) { // this XCD is complete
                    __hip_atomic_fetch_add(glob, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!cluster_wait<1>(glob, (i + 1) * 8, 0)) atomicOr(err, 2), dead = 1;
                    __hip_atomic_fetch_add(release, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (!cluster_wait<1>(release, i + 1, 0))
                    atomicOr(err, 2), dead = 1;
            }
        }
        __syncthreads();
        if (PAYLOAD) {
            const int p = (id + G / 2 + 1) % G;
            const u64 *r = buf + (size_t)(i & 1) * G * per_wg + (size_t)p * per_wg;
#pragma unroll
            for (int j = 0; j < PAYLOAD; j++) {
                u64 v = __hip_atomic_load(r + j * blockDim.x + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (((u64)i << 32) | (u64)(p * 16 + j))) atomicOr(err, 4);
                acc += v;
            }
        }
    }
    if (acc == 0x12345) *sink = acc;
}

template <bool FLAT, int PAYLOAD>
static float run_device(int G, int threads, int iters, int *cnt, u64 *buf, int *err, u64 *sink, hipEvent_t e0, hipEvent_t e1, int *errs)
{
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(cnt, 0, 4 * 32 * 32));
        CK(hipMemset(err, 0, 64));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((device_barrier_loop<FLAT, PAYLOAD>), dim3(G), dim3(threads), 0, 0, iters, cnt, buf, err, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int h;
        CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        *errs |= h;
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f / iters;
}

// the launch chain the barrier would replace: each kernel writes its payload, the next one reads a partner's
template <int PAYLOAD>
__global__ __launch_bounds__(1024) void chain_step(int C, int i, u64 *__restrict__ buf, int *__restrict__ err, u64 *__restrict__ sink)
{
    const unsigned id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const int t = (int)(k % (unsigned)C), cl = (int)(k / (unsigned)C) * 8 + (int)xcd;
    const size_t per_wg = (size_t)blockDim.x * (PAYLOAD ? PAYLOAD : 1);
    u64 *base = buf + (size_t)cl * C * per_wg * 2;
    u64 acc = 0;
    if (PAYLOAD && i > 0) {
#pragma unroll
        for (int j = 0; j < PAYLOAD; j++) {
            const int p = (t + 1 + j) % C;
            const u64 *r = base + (size_t)((i - 1) & 1) * C * per_wg + (size_t)p * per_wg;
            u64 v = r[j * blockDim.x + threadIdx.x];
            if (v != (((u64)(i - 1) << 32) | (u64)(p * 1024 + j))) atomicOr(err, 4);
            acc += v;
        }
    }
    if (PAYLOAD) {
        u64 *w = base + (size_t)(i & 1) * C * per_wg + (size_t)t * per_wg;
#pragma unroll
        for (int j = 0; j < PAYLOAD; j++) w[j * blockDim.x + threadIdx.x] = ((u64)i << 32) | (u64)(t * 1024 + j);
    }
    if (acc == 0x12345) *sink = acc;
}

template <int POLL, int PAYLOAD, bool SPREAD = false, bool WT = false>
static float run_barrier(int C, int clusters_per_xcd, int threads, int iters, int sleep, int *cnt, u64 *buf, int *err, u64 *sink, hipEvent_t e0, hipEvent_t e1,
                         int *errs)
{
    const int grid = 8 * C * clusters_per_xcd;
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(cnt, 0, 4 * 32 * 8 * clusters_per_xcd));
        CK(hipMemset(err, 0, 64));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((barrier_loop<POLL, PAYLOAD, SPREAD, WT>), dim3(grid), dim3(threads), 0, 0, C, iters, sleep, cnt, buf, err, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int h[9];
        CK(hipMemcpy(h, err, 36, hipMemcpyDeviceToHost));
        if (!SPREAD && (h[0] & 1) && !(*errs & 1)) {
            printf("  (placement: C = %d, %d per XCD, %d threads, rep %d: the first eight workgroups ran on XCDs", C, clusters_per_xcd, threads, rep);
            for (int i = 0; i < 8; i++) printf(" %d", h[1 + i]);
            printf(")\n");
        }
        *errs |= h[0];
        if (rep && ms < best) best = ms;
    }
    return best * 1e3f / iters;
}

template <int PAYLOAD>
static float run_chain(int C, int clusters_per_xcd, int threads, int iters, u64 *buf, int *err, u64 *sink, hipEvent_t e0, hipEvent_t e1, int *errs)
{
    const int grid = 8 * C * clusters_per_xcd;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((chain_step<PAYLOAD>), dim3(grid), dim3(threads), 0, s, C, i, buf, err, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(err, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int h;
        CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        *errs |= h;
        if (rep && ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(s));
    return best * 1e3f / iters;
}

int main()
{
    int *cnt, *err;
    u64 *buf, *sink;
    const int max_cl = 8 * 4;
    CK(hipMalloc(&cnt, 4 * 32 * 64));
    CK(hipMalloc(&err, 64));
    CK(hipMalloc(&sink, 8));
    CK(hipMalloc(&buf, (size_t)max_cl * 32 * 1024 * 8 * 2 * sizeof(u64)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    int errs = 0;
    printf("XCD-local cluster barrier, us per barrier (best of 3 after a warm-up launch; %d barriers per launch)\n", iters);
    printf("%-9s %-8s %-8s | %-12s %-12s %-12s %-12s | %-14s %-14s | %-14s %-14s\n", "cluster", "per_xcd", "threads", "rmw_poll", "rmw+sleep", "ld_poll", "ld+sleep",
           "ld+16B/thr", "ld+64B/thr", "chain 16B/thr", "chain 64B/thr");
    for (int threads : { 256, 1024 })
        for (int per_xcd : { 1, 2, 4 })
            for (int C : { 2, 4, 8, 16, 32 }) {
                if (C * per_xcd > 32 * (threads == 256 ? 4 : 1)) continue; // keep every workgroup resident (32 CUs per XCD; 1024-thread workgroups: one per CU here)
                if (threads == 1024 && C * per_xcd > 32) continue;
                float a = run_barrier<0, 0>(C, per_xcd, threads, iters, 0, cnt, buf, err, sink, e0, e1, &errs);
                float b = run_barrier<0, 0>(C, per_xcd, threads, iters, 1, cnt, buf, err, sink, e0, e1, &errs);
                float c = run_barrier<1, 0>(C, per_xcd, threads, iters, 0, cnt, buf, err, sink, e0, e1, &errs);
                float d = run_barrier<1, 0>(C, per_xcd, threads, iters, 1, cnt, buf, err, sink, e0, e1, &errs);
                float p2 = run_barrier<1, 2>(C, per_xcd, threads, iters, 0, cnt, buf, err, sink, e0, e1, &errs);
                float p8 = run_barrier<1, 8>(C, per_xcd, threads, iters, 0, cnt, buf, err, sink, e0, e1, &errs);
                float l2 = run_chain<2>(C, per_xcd, threads, 500, buf, err, sink, e0, e1, &errs);
                float l8 = run_chain<8>(C, per_xcd, threads, 500, buf, err, sink, e0, e1, &errs);
                printf("%-9d %-8d %-8d | %-12.2f %-12.2f %-12.2f %-12.2f | %-14.2f %-14.2f | %-14.2f %-14.2f\n", C, per_xcd, threads, a, b, c, d, p2, p8, l2, l8);
                fflush(stdout);
            }
    printf("\nclusters whose members sit on DIFFERENT XCDs (consecutive grid indices), 256 threads, 1 cluster of each kind per XCD-octet: us per barrier\n");
    printf("%-9s | %-12s %-12s | %-22s %-22s\n", "cluster", "rmw_poll", "ld_poll", "plain st + sc1 ld 16B", "sc1 st + sc1 ld 16B");
    int errs_spread_plain = 0, errs_spread_wt = 0;
    for (int C : { 8, 16, 32 }) {
        int e_ = 0;
        float a = run_barrier<0, 0, true>(C, 1, 256, iters, 0, cnt, buf, err, sink, e0, e1, &e_);
        float c = run_barrier<1, 0, true>(C, 1, 256, iters, 0, cnt, buf, err, sink, e0, e1, &e_);
        float p = run_barrier<1, 2, true, false>(C, 1, 256, iters, 0, cnt, buf, err, sink, e0, e1, &errs_spread_plain);
        float w = run_barrier<1, 2, true, true>(C, 1, 256, iters, 0, cnt, buf, err, sink, e0, e1, &errs_spread_wt);
        printf("%-9d | %-12.2f %-12.2f | %-22.2f %-22.2f   (flags: barrier %d, plain stores %d, sc1 stores %d)\n", C, a, c, p, w, e_ & ~1, errs_spread_plain & ~1, errs_spread_wt & ~1);
        fflush(stdout);
    }
    printf("\nDEVICE-wide barrier, all workgroups resident: us per barrier (payload written with sc1 stores, read with sc1 loads from half the grid away)\n");
    printf("%-9s %-8s | %-12s %-12s | %-16s %-16s\n", "wgs", "threads", "flat", "hierarchical", "hier + 16B/thr", "hier + 64B/thr");
    int errs_dev = 0;
    for (int threads : { 256, 1024 })
        for (int G : { 64, 128, 256, 512 }) {
            if (threads == 1024 && G > 256) continue;
            float f = run_device<true, 0>(G, threads, 500, cnt, buf, err, sink, e0, e1, &errs_dev);
            float h = run_device<false, 0>(G, threads, iters, cnt, buf, err, sink, e0, e1, &errs_dev);
            float h2 = run_device<false, 2>(G, threads, iters, cnt, buf, err, sink, e0, e1, &errs_dev);
            float h8 = run_device<false, 8>(G, threads, iters, cnt, buf, err, sink, e0, e1, &errs_dev);
            printf("%-9d %-8d | %-12.2f %-12.2f | %-16.2f %-16.2f   (flags %d)\n", G, threads, f, h, h2, h8, errs_dev);
            fflush(stdout);
        }
    errs |= (errs_dev & ~1) | (errs_spread_wt & ~1);
    printf("error flags (1 = a workgroup off its XCD, 2 = a barrier timed out, 4 = stale / missing payload): %d\n", errs);
    return (errs & ~1) ? 2 : 0; // (flag 1 alone: placement differs from id % 8 but no barrier or payload failed)
}
