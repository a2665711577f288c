// Experiment helper (not part of the product): occupies `n_wg` compute units for a fixed time with workgroups that hold a
// whole CU's LDS and do nothing, so that the behaviour of another stream's kernels on the REMAINING CUs can be timed in
// isolation (tools/token_hog_probe.py).   hipcc -O2 --offload-arch=gfx950 -shared -fPIC cu_hog.hip -o libcuhog.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(512) void cu_hog_kernel(long long ticks) {
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();   // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
    if (ticks < 0) lds[threadIdx.x] = 0;
}

extern "C" int cu_hog_launch(int n_wg, double seconds, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)cu_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(cu_hog_kernel, dim3(n_wg), dim3(512), 150 * 1024, (hipStream_t)stream, (long long)(seconds * 1e8));
    return (int)hipGetLastError();
}

// The same with memory traffic: every workgroup streams float4 reads over `bytes` of `buf` until the time is up (loaded-
// latency experiments: how much slower does a latency-bound chain on the other CUs get when the memory system is busy?)
__global__ __launch_bounds__(512) void mem_hog_kernel(const float4* __restrict__ buf, long n4, long long ticks, float* sink) {
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long i = ((long)blockIdx.x * 512 + threadIdx.x) % n4;
    const long step = (long)gridDim.x * 512;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 v = buf[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            i += step;
            if (i >= n4) i -= n4;
        }
    }
    if (acc.x == 12345.678f) sink[0] = acc.y + acc.z + acc.w + lds[0];
}

extern "C" int mem_hog_launch(int n_wg, double seconds, const void* buf, long bytes, void* sink, int lds_kb, void* stream) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)mem_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(mem_hog_kernel, dim3(n_wg), dim3(512), (size_t)lds_kb * 1024, (hipStream_t)stream, (const float4*)buf, bytes / 16,
                       (long long)(seconds * 1e8), (float*)sink);
    return (int)hipGetLastError();
}
