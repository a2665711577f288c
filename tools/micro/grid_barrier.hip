// Micro-benchmark: latency of a device-wide barrier between co-resident workgroups on MI355X
// (monotonic counter(s), release/acquire at agent scope).  hipcc --offload-arch=gfx950 -O3 -o gb grid_barrier.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NC, bool FENCE>
__global__ __launch_bounds__(512) void bar_kernel(int* cnt, int iters, float* sink, const float* src) {
    const int wg = blockIdx.x, n = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // some dependent work: a store, then the barrier, then a load of another WG's store
        if (threadIdx.x == 0) { if (FENCE) sink[wg] = (float)it; else __hip_atomic_store(sink + wg, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt + (wg % NC) * 1024, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // targets: counter c receives arrivals of WGs with wg % NC == c
            bool done = false;
            int spins = 0;
            while (!done && spins < (1 << 22)) {
                done = true;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int members = (n - c + NC - 1) / NC;
                    if (__hip_atomic_load(cnt + c * 1024, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (it + 1) * members) done = false;
                }
                if (!done) __builtin_amdgcn_s_sleep(1);
                ++spins;
            }
        }
        __syncthreads();
        if (FENCE) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); acc += sink[(wg + 1) % n]; }
        else acc += __hip_atomic_load(sink + (wg + 1) % n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0 && acc == -1.f) sink[0] = acc;
}

__global__ void empty_kernel(float* sink) { if (sink == nullptr) sink[0] = 0; }

int main() {
    int* cnt; float* sink;
    hipMalloc(&cnt, 64 * 1024 * sizeof(int));
    hipMalloc(&sink, 4096 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int nwg : {64, 256}) {
        for (int variant = 0; variant < 4; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(cnt, 0, 64 * 1024 * sizeof(int));
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL((bar_kernel<1, true>), dim3(nwg), dim3(512), 0, 0, cnt, iters, sink, sink);
                if (variant == 1) hipLaunchKernelGGL((bar_kernel<8, true>), dim3(nwg), dim3(512), 0, 0, cnt, iters, sink, sink);
                if (variant == 2) hipLaunchKernelGGL((bar_kernel<1, false>), dim3(nwg), dim3(512), 0, 0, cnt, iters, sink, sink);
                if (variant == 3) hipLaunchKernelGGL((bar_kernel<8, false>), dim3(nwg), dim3(512), 0, 0, cnt, iters, sink, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("nwg=%d variant=%s: %.2f us per barrier\n", nwg, variant == 0 ? "1 counter, fences" : (variant == 1 ? "8 counters, fences" : (variant == 2 ? "1 counter, sc1 data no fences" : "8 counters, sc1 data no fences")), best * 1000.f / iters);
        }
    }
    // dependent empty-kernel chain in a graph for comparison
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, sink);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph of empty 256x512 kernels: %.2f us per kernel\n", ms * 1000.f / 2000);
    return 0;
}
