// Experiment helper (not part of the product): what does a B-fragment fetch cost the LDS pipe?
//   (a) ds_read_b128, 64 lanes, consecutive 16-byte slots            (the conv kernel's fragment read)
//   (b) the same instruction with only lanes 31 and 63 active         (exec-masked edge fix)
//   (c) 64 lanes, two distinct addresses (lanes 0-31 one slot, 32-63 another)  (broadcast edge fix)
//   (d) no LDS: the fragment of the next tap from the current one by v_mov_b32_dpp wave_shl:1 (4 per fragment)
// 8 waves per workgroup, one workgroup per CU; cycles per instruction from s_memtime around an unrolled loop.
// Also prints what wave_shl:1 does to the lane ids (lane i must receive lane i + 1).
//   hipcc -O2 --offload-arch=gfx950 lds_dpp_probe.hip -o lds_dpp_probe && ./lds_dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* cycles, u32x4* sink, int iters) {
    __shared__ u32x4 buf[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 512) buf[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0}, cur = buf[tid];
    int addr = (MODE == 2) ? (tid >> 5) * 37 : tid;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0 || MODE == 2) {
                const u32x4 v = buf[(addr + 64 * u + it) & 4095];
                acc += v;
            } else if (MODE == 1) {
                if ((lane & 31) == 31) {
                    const u32x4 v = buf[(addr + 64 * u + it) & 4095];
                    acc += v;
                }
            } else {
                u32x4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)cur[c], 0x130, 0xf, 0xf, false);
                cur = r + u32x4{1, 1, 1, 1};
                acc += cur;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void semantics(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xf, 0xf, false);
}

int main() {
    int* d_sem; hipMalloc(&d_sem, 64 * sizeof(int));
    semantics<<<1, 64>>>(d_sem);
    int sem[64]; hipMemcpy(sem, d_sem, sizeof(sem), hipMemcpyDeviceToHost);
    printf("wave_shl:1 of the lane id: lane 0 <- %d, lane 30 <- %d, lane 31 <- %d, lane 32 <- %d, lane 62 <- %d, lane 63 <- %d\n", sem[0], sem[30], sem[31], sem[32], sem[62], sem[63]);
    const int wgs = 256, iters = 2000;
    unsigned long long* d_cyc; u32x4* d_sink;
    hipMalloc(&d_cyc, wgs * sizeof(unsigned long long)); hipMalloc(&d_sink, (size_t)wgs * 512 * sizeof(u32x4));
    const char* names[4] = {"ds_read_b128, 64 lanes, consecutive slots", "ds_read_b128, lanes 31 and 63 only (exec mask)", "ds_read_b128, 64 lanes, 2 distinct addresses",
                            "4 x v_mov_b32_dpp wave_shl:1 (+4 v_add)"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) probe<0><<<wgs, 512>>>(d_cyc, d_sink, iters);
            if (mode == 1) probe<1><<<wgs, 512>>>(d_cyc, d_sink, iters);
            if (mode == 2) probe<2><<<wgs, 512>>>(d_cyc, d_sink, iters);
            if (mode == 3) probe<3><<<wgs, 512>>>(d_cyc, d_sink, iters);
            hipDeviceSynchronize();
        }
        std::vector<unsigned long long> c(wgs);
        hipMemcpy(c.data(), d_cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double s = 0; for (auto v : c) s += (double)v;
        // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part: report ticks per 16-instruction group per wave and per CU
        printf("%-52s %8.3f ticks per fragment per wave (8 waves per CU issuing)\n", names[mode], s / wgs / iters / 16);
    }
    return 0;
}
