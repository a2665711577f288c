// Experiment helper (not part of the product): background kernels of ONE kind of load each, to find out what a convolution
// launch loses to when the token loops of other batches run beside it (tools/conv_contention_probe.py).  Every hog
// workgroup has the footprint of the product's decode kernels -- 256 threads, <= 48 VGPRs, a few KB of LDS -- so that it is
// dispatched NEXT TO a convolution workgroup (8 waves x ~210 VGPRs, ~100 KB of LDS per CU), runs for a fixed wall time and
// reports how many 16-byte loads it made.
//   mode 0  idle      s_sleep only: occupies wave slots / registers, issues nothing
//   mode 1  stream    16-byte loads walking a multi-GB buffer once, default cache policy (allocates in L2 / Infinity Cache)
//   mode 2  stream-nt the same with the non-temporal policy (the attention's key / value reads)
//   mode 3  l2        every workgroup re-reads its own 256 KB window: L2 hits, no HBM traffic, no eviction of other data
//   mode 4  valu      dependent FMAs: vector issue slots, no memory
//   mode 5  lds       ds_read_b128 in a loop: LDS bandwidth, no global memory
//   hipcc -O2 --offload-arch=gfx950 -shared -fPIC hogs.hip -o libhogs.so
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void hog_kernel(const f32x4* __restrict__ buf, long n4, long long ticks, unsigned long long* count, float* sink) {
    __shared__ f32x4 lds[512];
    const long long t0 = wall_clock64();   // 100 MHz
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    unsigned long long n = 0;
    if (MODE == 5) { lds[threadIdx.x] = acc; lds[threadIdx.x + 256] = acc; __syncthreads(); }
    const long step = (long)gridDim.x * 256;
    long i = ((long)blockIdx.x * 256 + threadIdx.x) % n4;
    const long win = 256 * 1024 / 16;                                  // mode 3: a 256 KB window per workgroup
    const long w0 = ((long)blockIdx.x * win) % (n4 - win);
    long j = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (MODE == 0) {
            __builtin_amdgcn_s_sleep(127);
        } else if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 v = MODE == 2 ? __builtin_nontemporal_load(buf + i) : buf[i];
                acc += v;
                i += step;
                if (i >= n4) i -= n4;
            }
            n += 4;
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc += buf[w0 + j];
                j += 256;
                if (j >= win) j -= win;
            }
            n += 4;
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = acc * 1.0001f + 0.5f;
            n += 64;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += lds[(threadIdx.x + 16 * u + (int)n) & 511];
            n += 4;
        }
    }
    if (threadIdx.x == 0) atomicAdd(count, n);   // per-thread operations of this workgroup (all its threads do the same number)
    if (acc[0] == 12345.678f) sink[0] = acc[1] + acc[2] + acc[3];
}

extern "C" int hog_launch(int mode, int n_wg, double seconds, const void* buf, long bytes, void* count, void* sink, void* stream) {
    const long long ticks = (long long)(seconds * 1e8);
    const f32x4* b = (const f32x4*)buf;
    const long n4 = bytes / 16;
    unsigned long long* c = (unsigned long long*)count;
    float* s = (float*)sink;
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
    case 0: hipLaunchKernelGGL(hog_kernel<0>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    case 1: hipLaunchKernelGGL(hog_kernel<1>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    case 2: hipLaunchKernelGGL(hog_kernel<2>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    case 3: hipLaunchKernelGGL(hog_kernel<3>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    case 4: hipLaunchKernelGGL(hog_kernel<4>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    case 5: hipLaunchKernelGGL(hog_kernel<5>, dim3(n_wg), dim3(256), 0, st, b, n4, ticks, c, s); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
