// Experiment helper (not part of the product): at what RATE does the chip execute v_mfma_f32_32x32x16_bf16 when nothing but the
// matrix pipe is busy -- and how much of that rate do the other parts of a convolution step (operand fetches from LDS) cost?
// The convolution kernels of this library deliver ~1.0 matrix-busy GHz per SIMD on random data whatever their idle time
// (profiles/r06_pmc_conv_pt.txt): is that the power manager's rate for matrix work as such, or for matrix work + the rest of the kernel?
//   mode 0: operands in registers, random bf16 data        mode 1: operands in registers, all zero
//   mode 2: operands re-read from LDS every tap at the convolution's ratio (12 ds_read_b128 per 24 matrix instructions), random data
//   mode 3: as 2, plus a second wave per SIMD that streams 16-byte global loads and converts fp32 -> split bf16 (the staging role's work)
// One workgroup per CU, one matrix wave per SIMD (as in the convolution); 24 independent-accumulator matrix instructions per tap.
// Prints per mode: wall time, shader cycles (s_memtime), clock, matrix-pipe busy share and busy x clock.
//   hipcc -O2 --offload-arch=gfx950 mfma_rate_probe.hip -o mfma_rate_probe && ./mfma_rate_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const u32x4* __restrict__ src, const f32x4* __restrict__ stream, unsigned long long* cycles, float* sink, int iters, int stream_items) {
    __shared__ u32x4 buf[12 * 64 * 4];
    __shared__ volatile int done;
    if (threadIdx.x == 0) done = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (MODE >= 2)
        for (int i = tid; i < 12 * 64 * 4; i += 512) buf[i] = src[i];
    __syncthreads();
    if (wave >= 4) {
        if (MODE == 3) {   // the staging role: stream fp32 rows, split to hi / lo bf16, keep a checksum
            float a = 0.f;
            const f32x4* p = stream + ((size_t)blockIdx.x * 256 + (tid - 256));
            for (int it = 0; !done; ++it) {   // until the matrix waves have finished
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = p[((size_t)(it * 8 + i) * 65536) % (size_t)stream_items];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const __bf16 h = (__bf16)v[i][j];
                        const __bf16 l = (__bf16)(v[i][j] - (float)h);
                        a += (float)h + (float)l;
                    }
            }
            if (a == 12345.f) sink[tid] = a;
        }
        return;
    }
    bf16x8 fa[4][2], fb[2][2];
    const u32x4* base = (MODE == 1) ? nullptr : src + (wave * 12) * 64 + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) fa[m][h] = __builtin_bit_cast(bf16x8, MODE == 1 ? u32x4{0, 0, 0, 0} : base[(m * 2 + h) * 64]);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) fb[p][h] = __builtin_bit_cast(bf16x8, MODE == 1 ? u32x4{0, 0, 0, 0} : base[(8 + p * 2 + h) * 64]);
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {   // the tap's operands from LDS (conflict-free 16-byte slots), as the convolution's MFMA waves fetch them
            const u32x4* lb = buf + wave * 12 * 64 + lane + ((it & 1) ? 0 : 0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h) fa[m][h] = __builtin_bit_cast(bf16x8, lb[(m * 2 + h) * 64]);
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int h = 0; h < 2; ++h) fb[p][h] = __builtin_bit_cast(bf16x8, lb[(8 + p * 2 + h) * 64]);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][1], fb[p][0], acc[m][p], 0, 0, 0);
                acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][0], fb[p][1], acc[m][p], 0, 0, 0);
                acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][0], fb[p][0], acc[m][p], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    if (tid == 0) done = 1;
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][p][r];
    if (s == 12345.f) sink[tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<unsigned> h(12 * 64 * 4 * 4);
    srand(7);
    for (auto& v : h) {   // two random bf16 in [-2, 2) per word: sign, exponent 125..127, 7 random mantissa bits
        unsigned w = 0;
        for (int k = 0; k < 2; ++k) w |= (((unsigned)(rand() & 1) << 15) | ((125u + rand() % 3) << 7) | (unsigned)(rand() & 127)) << (16 * k);
        v = w;
    }
    u32x4* src; f32x4* stream; unsigned long long* cyc; float* sink;
    const int stream_items = 1 << 24;   // 256 MB of fp32 rows
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&stream, (size_t)stream_items * 16)); CK(hipMemset(stream, 0x3c, (size_t)stream_items * 16));
    CK(hipMalloc(&cyc, n_cu * 4 * 8)); CK(hipMalloc(&sink, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d CUs, %d iterations of 24 v_mfma_f32_32x32x16_bf16 per matrix wave (one per SIMD); s_memtime ticks assumed to be 100 MHz unless they match the clock\n", n_cu, iters);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(n_cu), dim3(512), 0, 0, src, stream, cyc, sink, iters, stream_items); break;
                case 1: hipLaunchKernelGGL(probe<1>, dim3(n_cu), dim3(512), 0, 0, src, stream, cyc, sink, iters, stream_items); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(n_cu), dim3(512), 0, 0, src, stream, cyc, sink, iters, stream_items); break;
                default: hipLaunchKernelGGL(probe<3>, dim3(n_cu), dim3(512), 0, 0, src, stream, cyc, sink, iters, stream_items); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> c(n_cu * 4);
            CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : c) avg += (double)v; avg /= c.size();
            const int it = iters;
            const double mfma_cycles = 24.0 * 32.0 * it;             // matrix-pipe cycles of one SIMD (32 per instruction)
            const double busy_ghz = mfma_cycles / (ms * 1e6);         // busy cycles per nanosecond
            if (rep == 2)
                printf("mode %d: %8.3f ms  s_memtime ticks %.3e (%.3f GHz if they are shader cycles)  matrix busy-GHz %.3f = %.1f %% of 2.4  (%.0f TFLOP/s dense bf16 chip-wide)\n", mode, ms, avg,
                       avg / (ms * 1e6), busy_ghz, 100.0 * busy_ghz / 2.4, 2.0 * 32 * 32 * 16 * 24.0 * it * n_cu * 4 / (ms * 1e9));
        }
    }
    return 0;
}
