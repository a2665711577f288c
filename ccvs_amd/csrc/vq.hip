// Vector quantiser: fused distance + argmin, and the code -> feature gather.
//
// ccvs_vq_argmin   <- VectorQuantizer.forward, modules/quantize.py:40-50
// ccvs_embed_gather <- VectorQuantizer.embed_code + transposes, quantize.py:76-83,
//                      quantized_video_model.py:832-833
// ccvs_l2_normalize_channels <- the encoder's `normalize_out`, skip_autoencoder.py:348-349
#include "common.h"

// One workgroup = 32 rows (latent positions).  Their C-vector tile [C][32] sits in LDS
// (read straight from NCHW: 32 consecutive positions per channel are contiguous); each of
// the 4 waves walks its share of the codebook in blocks of 32 codes, streaming the
// transposed codebook [C][n_e] (128-B coalesced per half-wave, L2-resident) into the A
// operand of v_mfma_f32_32x32x2_f32.  D[code][row]: a lane owns one row and 16 codes, so
// the running (min, index) lives in registers; ties keep the lowest index like
// torch.argmin.  d = (|z|^2 + |e|^2) - 2 z.e in the reference's association order.
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ cbt,
                                                        const float* __restrict__ e_sq, int64_t* __restrict__ idx, long rows, int C,
                                                        int HW, int n_e) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* zt = smem;             // [C][32]
    float* zz = smem + C * 32;    // [32]
    float* red_d = zz + 32;       // [4][32]
    int* red_i = (int*)(red_d + 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r0 = (long)blockIdx.x * 32;

    for (int e = tid; e < C * 32; e += 256) {
        const int c = e >> 5, j = e & 31;
        const long r = r0 + j;
        float v = 0.f;
        if (r < rows) {
            const long n = r / HW, p = r - n * HW;
            v = z[(n * C + c) * HW + p];
        }
        zt[e] = v;
    }
    __syncthreads();
    {   // |z|^2 per row: 8 threads per row
        const int j = tid >> 3, s = tid & 7;
        float a = 0.f;
        for (int c = s; c < C; c += 8) { const float v = zt[c * 32 + j]; a += v * v; }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        if (s == 0) zz[j] = a;
    }
    __syncthreads();

    const int j = lane & 31, khalf = lane >> 5;
    const float zzj = zz[j];
    float best_d = INFINITY;
    int best_i = 0x7fffffff;
    for (int m = wave; m < n_e / 32; m += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* ap = cbt + (long)khalf * n_e + m * 32 + j;
        const float* bp = zt + khalf * 32 + j;
#pragma unroll 8
        for (int k0 = 0; k0 < C; k0 += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(long)k0 * n_e], bp[k0 * 32], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            const float d = (zzj + e_sq[code]) - 2.f * acc[r];
            if (d < best_d) { best_d = d; best_i = code; }
        }
    }
    {   // the two lane halves hold the same row
        const float od = __shfl_xor(best_d, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (od < best_d || (od == best_d && oi < best_i)) { best_d = od; best_i = oi; }
    }
    if (lane < 32) { red_d[wave * 32 + j] = best_d; red_i[wave * 32 + j] = best_i; }
    __syncthreads();
    if (tid < 32 && r0 + tid < rows) {
        float bd = red_d[tid];
        int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float od = red_d[w * 32 + tid];
            const int oi = red_i[w * 32 + tid];
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        idx[r0 + tid] = (int64_t)bi;
    }
}

// e_dim = 1 (the state quantiser, VectorQuantizer(state_num, 1), state_model.py:55): one scalar per row, so the
// reference's  sum(z^2) + sum(e^2) - 2 * (z @ e^T)  is  (z*z + e*e) - 2*(z*e)  with every operation rounded once
// (clang's fp contraction off so no fma changes a rounding).  One thread per row walks the codes in index order.
__global__ __launch_bounds__(256) void vq_argmin_scalar_kernel(const float* __restrict__ z, const float* __restrict__ cbt,
                                                               const float* __restrict__ e_sq, int64_t* __restrict__ idx, long rows,
                                                               int n_e) {
#pragma clang fp contract(off)
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float v = z[r];
    const float zz = v * v;
    float best_d = INFINITY;
    int best_i = 0;
    for (int j = 0; j < n_e; ++j) {
        const float d = (zz + e_sq[j]) - 2.f * (v * cbt[j]);
        if (d < best_d) { best_d = d; best_i = j; }
    }
    idx[r] = (int64_t)best_i;
}

extern "C" int ccvs_vq_argmin(const float* z, const float* codebook_t, const float* e_sq, int64_t* idx, int32_t N, int32_t C, int32_t HW,
                              int32_t n_e, void* stream) {
    CCVS_REQUIRE(z && codebook_t && e_sq && idx, "ccvs_vq_argmin: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && HW > 0 && n_e > 0, "ccvs_vq_argmin: empty tensor");
    const long rows = (long)N * HW;
    if (C == 1) {
        hipLaunchKernelGGL(vq_argmin_scalar_kernel, dim3((unsigned)cdiv64(rows, 256)), dim3(256), 0, (hipStream_t)stream, z, codebook_t, e_sq,
                           idx, rows, n_e);
        CCVS_CHECK_LAUNCH("ccvs_vq_argmin");
        return CCVS_OK;
    }
    CCVS_REQUIRE(n_e % 32 == 0, "ccvs_vq_argmin: n_e=%d must be a multiple of 32", n_e);
    CCVS_REQUIRE(C % 2 == 0 && C <= 1024, "ccvs_vq_argmin: C=%d must be 1 or even and <= 1024", C);
    const size_t smem = (size_t)(C * 32 + 32 + 128 + 128) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)vq_argmin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((unsigned)cdiv64(rows, 32)), dim3(256), smem, (hipStream_t)stream, z, codebook_t, e_sq, idx,
                       rows, C, HW, n_e);
    CCVS_CHECK_LAUNCH("ccvs_vq_argmin");
    return CCVS_OK;
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ code, const float* __restrict__ cb,
                                                           float* __restrict__ z, long total, int C, int HW, int n_e) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C);
        const long n = t / C;
        long k = code[n * HW + p];
        k = k < 0 ? 0 : (k >= n_e ? n_e - 1 : k);
        z[i] = cb[k * C + c];
    }
}

extern "C" int ccvs_embed_gather(const int64_t* code, const float* codebook, float* z, int32_t N, int32_t C, int32_t HW, int32_t n_e,
                                 void* stream) {
    CCVS_REQUIRE(code && codebook && z, "ccvs_embed_gather: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && HW > 0 && n_e > 0, "ccvs_embed_gather: empty tensor");
    const long total = (long)N * C * HW;
    const int blocks = (int)(cdiv64(total, 256) < 1048576 ? cdiv64(total, 256) : 1048576);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, code, codebook, z, total, C, HW, n_e);
    CCVS_CHECK_LAUNCH("ccvs_embed_gather");
    return CCVS_OK;
}

// One lane = one latent position (consecutive lanes = consecutive positions: every channel plane is read coalesced); the channel
// loop runs twice -- sum of squares in channel order, then the division -- the second pass finds the planes in L2 (the tensor is the
// encoder's output: z_size x 8 x 8 ... 32 x 32 per frame).
__global__ __launch_bounds__(256) void l2_normalize_channels_kernel(float* __restrict__ x, long rows, int C, long HW) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        const long n = r / HW, pix = r - n * HW;
        float* xp = x + n * C * HW + pix;
        float a = 0.f;
        for (int c = 0; c < C; ++c) { const float v = xp[(long)c * HW]; a += v * v; }
        const float nrm = sqrtf(a);
        for (int c = 0; c < C; ++c) xp[(long)c * HW] = xp[(long)c * HW] / nrm;
    }
}

extern "C" int ccvs_l2_normalize_channels(float* x, int32_t N, int32_t C, int64_t HW, void* stream) {
    CCVS_REQUIRE(x, "ccvs_l2_normalize_channels: null pointer");
    CCVS_REQUIRE(N > 0 && C > 0 && HW > 0, "ccvs_l2_normalize_channels: empty tensor");
    const long rows = (long)N * HW;
    const int blocks = (int)(cdiv64(rows, 256) < 65536 ? cdiv64(rows, 256) : 65536);
    hipLaunchKernelGGL(l2_normalize_channels_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, C, (long)HW);
    CCVS_CHECK_LAUNCH("ccvs_l2_normalize_channels");
    return CCVS_OK;
}
