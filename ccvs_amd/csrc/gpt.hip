// Transformer kernels: embedding, LayerNorm, nn.Linear GEMM on the fp32 matrix cores,
// KV-cached causal attention and the top-k / softmax / pick of the sampling loop.
// Reference: models/skip_vid_generator/models/mingpt.py:33-117,186-305 and
// models/skip_vid_generator/models/transformer_model.py:256-260,395-409.
#include "common.h"

// ---------------------------------------------------------------------------------------
// x[(b,t)][:] = tok_emb[idx[b*idx_sB + t]] + pos_table[pos_off[b] + pos0 + t]
// (mingpt.py:234-236,242-244; the factored s_emb/t_emb (+delta_length) or flat pos_emb rows
// are pre-summed by the host into pos_table once per call).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gpt_embed_kernel(const int64_t* __restrict__ idx, long idx_sB, const int32_t* __restrict__ pos_off,
                                                        int pos0, int Tq, const float* __restrict__ tok, const float* __restrict__ pos,
                                                        float* __restrict__ x, long total, int C, int vocab) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const long b = r / Tq;
        const int tq = (int)(r - b * Tq);
        long t = idx[b * idx_sB + tq];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        const long prow = (pos_off ? pos_off[b] : 0) + pos0 + tq;
        x[i] = tok[t * C + c] + pos[prow * C + c];
    }
}

extern "C" int ccvs_gpt_embed(const int64_t* idx, int64_t idx_sB, const int32_t* pos_off, int32_t pos0, int32_t Tq, const float* tok_emb,
                              const float* pos_table, float* x, int32_t B, int32_t C, int32_t vocab, void* stream) {
    CCVS_REQUIRE(idx && tok_emb && pos_table && x, "ccvs_gpt_embed: null pointer");
    CCVS_REQUIRE(B > 0 && Tq > 0 && C > 0 && vocab > 0 && pos0 >= 0, "ccvs_gpt_embed: empty tensor");
    const long total = (long)B * Tq * C;
    hipLaunchKernelGGL(gpt_embed_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, idx, (long)idx_sB, pos_off,
                       pos0, Tq, tok_emb, pos_table, x, total, C, vocab);
    CCVS_CHECK_LAUNCH("ccvs_gpt_embed");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5), one wave per row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, int rows, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mean = wave_sum(s) / C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / C + 1e-5f);
    float* yr = y + (long)row * C;
    for (int c = lane; c < C; c += 64) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
}

extern "C" int ccvs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t C, void* stream) {
    CCVS_REQUIRE(x && gamma && beta && y, "ccvs_layernorm: null pointer");
    CCVS_REQUIRE(rows > 0 && C > 0, "ccvs_layernorm: empty tensor");
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, C);
    CCVS_CHECK_LAUNCH("ccvs_layernorm");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// y[M,N] = epilogue(x[M,K] @ W[N,K]^T + bias) on v_mfma_f32_16x16x4_f32.
// Workgroup = 16 rows of x (staged in LDS, K chunks of GEMM_KC) x (16*WN) columns; its 4
// waves are WN column groups x (4/WN) K slices.  Each lane streams W straight from HBM as
// float4 along K (row n = lane&15, k = k0 + 4*(lane>>4) + t): the 4 floats feed 4
// consecutive MFMAs whose k-slot (lane>>4) then means "k0 + 4*slot + t" for both
// operands.  Weights are read once per 16-row block, never staged (decode is a weight
// stream: M = batch = 16 rows exactly fills the 16x16x4 tile).  K slices are summed
// through LDS in a fixed order: results are bitwise reproducible run to run.
// ---------------------------------------------------------------------------------------
#define GEMM_KC 512
#define GEMM_LDX (GEMM_KC + 4)

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

template <int WN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      float* __restrict__ y, long ldy, int M, int N, int K, int epi) {
    constexpr int KS = 4 / WN;
    __shared__ __attribute__((aligned(16))) float xs[16 * GEMM_LDX];
    __shared__ __attribute__((aligned(16))) float red[KS > 1 ? (KS - 1) * WN * 64 * 4 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wk = wave / WN;
    const int m0 = blockIdx.y * 16;
    const int ncol0 = (blockIdx.x * WN + wn) * 16;
    const int li = lane & 15, g = lane >> 4;
    const int nrow = min(ncol0 + li, N - 1);  // clamp: tail columns are computed on a valid row and dropped
    const float* wp = w + (long)nrow * K + 4 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    for (int kc = 0; kc < K; kc += GEMM_KC) {
        const int klen = min(GEMM_KC, K - kc);
        __syncthreads();
        for (int e = tid * 4; e < 16 * klen; e += 256 * 4) {
            const int r = e / klen, c = e - r * klen;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + r < M) v = *reinterpret_cast<const float4*>(x + (long)(m0 + r) * ldx + kc + c);
            *reinterpret_cast<float4*>(xs + r * GEMM_LDX + c) = v;
        }
        __syncthreads();
        const int kper = klen / KS;  // klen is a multiple of 16*KS (checked on the host)
        const int kb = wk * kper;
        const float* xp = xs + li * GEMM_LDX + kb + 4 * g;
        const float* wq = wp + kc + kb;
#pragma unroll 4
        for (int k0 = 0; k0 < kper; k0 += 16) {
            const float4 wv = *reinterpret_cast<const float4*>(wq + k0);
            const float4 xv = *reinterpret_cast<const float4*>(xp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, wv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, wv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, wv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, wv.w, acc, 0, 0, 0);
        }
    }
    if (KS > 1) {
        __syncthreads();
        if (wk > 0) *reinterpret_cast<f32x4*>(red + (((wk - 1) * WN + wn) * 64 + lane) * 4) = acc;
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(red + (((s - 1) * WN + wn) * 64 + lane) * 4);
            acc += o;
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane&15]
    const int col = ncol0 + li;
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * g + r;
            if (row < M) {
                float v = acc[r] + bv;
                if (epi == 1) v = gelu_erf(v);
                if (epi == 2) v += res[(long)row * ldy + col];
                y[(long)row * ldy + col] = v;
            }
        }
    }
}

extern "C" int ccvs_gemm_nt(const float* x, int64_t ldx, const float* w, const float* bias, const float* res, float* y, int64_t ldy,
                            int32_t M, int32_t N, int32_t K, int32_t epilogue, void* stream) {
    CCVS_REQUIRE(x && w && y, "ccvs_gemm_nt: null pointer");
    CCVS_REQUIRE(M > 0 && N > 0 && K > 0, "ccvs_gemm_nt: empty tensor");
    CCVS_REQUIRE(K % 16 == 0 && ldx % 4 == 0, "ccvs_gemm_nt: K=%d must be a multiple of 16 (ldx %% 4 == 0)", K);
    CCVS_REQUIRE(epilogue >= 0 && epilogue <= 2 && (epilogue != 2 || res), "ccvs_gemm_nt: bad epilogue");
    hipStream_t st = (hipStream_t)stream;
    const int mblocks = cdiv(M, 16);
    // few rows (decode): 16 columns per workgroup with a 4-way K split keeps every CU streaming;
    // many rows (prefill): 64 columns per workgroup.
    const bool split_ok = (K % 64 == 0) && (K <= GEMM_KC || K % GEMM_KC == 0);
    if (mblocks <= 4 && split_ok) {
        hipLaunchKernelGGL((gemm_nt_kernel<1>), dim3(cdiv(N, 16), mblocks), dim3(256), 0, st, x, (long)ldx, w, bias, res, y, (long)ldy, M, N,
                           K, epilogue);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<4>), dim3(cdiv(N, 64), mblocks), dim3(256), 0, st, x, (long)ldx, w, bias, res, y, (long)ldy, M, N,
                           K, epilogue);
    }
    CCVS_CHECK_LAUNCH("ccvs_gemm_nt");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// KV cache append and causal attention over the cache (head dim 64).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_append_kernel(const float* __restrict__ k, const float* __restrict__ v, long sB, long ld,
                                                        float* __restrict__ kc, float* __restrict__ vc, long total, int H, int Tq,
                                                        int pos0, int Tmax, int D) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d = (int)(i % D);
        long t = i / D;
        const int h = (int)(t % H);
        t /= H;
        const int tq = (int)(t % Tq);
        const long b = t / Tq;
        const long src = b * sB + tq * ld + h * D + d;
        const long dst = ((b * H + h) * Tmax + pos0 + tq) * D + d;
        kc[dst] = k[src];
        vc[dst] = v[src];
    }
}

extern "C" int ccvs_kv_append(const float* k, const float* v, int64_t sB, int64_t ld, float* kcache, float* vcache, int32_t B, int32_t H,
                              int32_t Tq, int32_t pos0, int32_t Tmax, int32_t D, void* stream) {
    CCVS_REQUIRE(k && v && kcache && vcache, "ccvs_kv_append: null pointer");
    CCVS_REQUIRE(B > 0 && H > 0 && Tq > 0 && D > 0 && pos0 >= 0 && pos0 + Tq <= Tmax, "ccvs_kv_append: positions %d..%d exceed cache %d",
                 pos0, pos0 + Tq, Tmax);
    const long total = (long)B * Tq * H * D;
    hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, k, v, (long)sB, (long)ld, kcache,
                       vcache, total, H, Tq, pos0, Tmax, D);
    CCVS_CHECK_LAUNCH("ccvs_kv_append");
    return CCVS_OK;
}

// One workgroup per (batch, head, query).  Scores: one key per thread (row of D floats,
// q broadcast from LDS); softmax over the L = pos0+t+1 visible keys; PV: lane = head dim
// (coalesced V rows), 4 waves take keys round-robin and are summed through LDS.
template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, long q_sB, long ldq, const float* __restrict__ kc,
                                                        const float* __restrict__ vc, float* __restrict__ out, int H, int Tq, int pos0,
                                                        int Tmax, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* qs = smem;            // [64]
    float* red = smem + 64;      // [8]
    float* pv = smem + 80;       // [4][64]
    float* ps = smem + 80 + 256; // [L]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x % Tq;
    const int bh = blockIdx.x / Tq;
    const int b = bh / H, h = bh - b * H;
    const int L = pos0 + t + 1;
    const float* kbase = kc + (long)bh * Tmax * D;
    const float* vbase = vc + (long)bh * Tmax * D;
    if (tid < D) qs[tid] = q[(long)b * q_sB + (long)t * ldq + h * D + tid];
    __syncthreads();

    float lmax = -INFINITY;
    for (int j = tid; j < L; j += 256) {
        const float4* kr = reinterpret_cast<const float4*>(kbase + (long)j * D);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < D / 4; ++i) {
            const float4 kv = kr[i];
            const float4 qv = *reinterpret_cast<const float4*>(qs + 4 * i);
            s += kv.x * qv.x + kv.y * qv.y + kv.z * qv.z + kv.w * qv.w;
        }
        s *= scale;  // 1/sqrt(D)
        ps[j] = s;
        lmax = fmaxf(lmax, s);
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int j = tid; j < L; j += 256) {
        const float e = expf(ps[j] - gmax);
        ps[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.f / (((red[4] + red[5]) + red[6]) + red[7]);
    float acc = 0.f;
    if (lane < D)
        for (int j = wave; j < L; j += 4) acc += ps[j] * vbase[(long)j * D + lane];
    pv[wave * 64 + lane] = acc;
    __syncthreads();
    if (tid < D) {
        const float o = ((pv[tid] + pv[64 + tid]) + pv[128 + tid]) + pv[192 + tid];
        out[((long)b * Tq + t) * (H * D) + h * D + tid] = o * inv;
    }
}

extern "C" int ccvs_attention(const float* q, int64_t q_sB, int64_t ldq, const float* kcache, const float* vcache, float* out, int32_t B,
                              int32_t H, int32_t Tq, int32_t pos0, int32_t Tmax, int32_t D, void* stream) {
    CCVS_REQUIRE(q && kcache && vcache && out, "ccvs_attention: null pointer");
    CCVS_REQUIRE(D == 64 || D == 32 || D == 16, "ccvs_attention: head dim %d unsupported (16, 32, 64)", D);
    CCVS_REQUIRE(B > 0 && H > 0 && Tq > 0 && pos0 >= 0 && pos0 + Tq <= Tmax, "ccvs_attention: bad positions");
    const size_t smem = (size_t)(80 + 256 + pos0 + Tq) * sizeof(float);
    CCVS_REQUIRE(smem <= 64 * 1024, "ccvs_attention: sequence too long for the LDS score buffer");
    const dim3 grid((unsigned)((long)B * H * Tq));
    const float scale = 1.0f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    if (D == 64) hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), smem, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, Tmax, scale);
    else if (D == 32) hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), smem, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, Tmax, scale);
    else hipLaunchKernelGGL((attention_kernel<16>), grid, dim3(256), smem, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, Tmax, scale);
    CCVS_CHECK_LAUNCH("ccvs_attention");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// get_icode: temperature, top-k mask (ties with the k-th value kept), softmax, then
// argmax(p) (greedy) or argmax(p / Exp(1) noise) (== torch.multinomial(p, 1)).
// One workgroup per row; the k-th largest value is found by a 32-step radix descent on
// order-preserving integer keys.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int block_sum_int(int v, int* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sample_topk_kernel(const float* __restrict__ logits, long ld, const float* __restrict__ noise,
                                                          int64_t* __restrict__ out, long out_stride, int V, int top_k, float temperature) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [V]
    int* redi = (int*)(smem + V);     // [4]
    float* redf = smem + V + 4;       // [4]
    int* redj = (int*)(smem + V + 8); // [4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* lr = logits + (long)b * ld;
    float lmax = -INFINITY;
    for (int j = tid; j < V; j += 256) {
        const float v = lr[j] / temperature;
        xs[j] = v;
        lmax = fmaxf(lmax, v);
    }
    lmax = wave_max(lmax);
    __syncthreads();
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));

    unsigned thr = 0u;  // key of the k-th largest value; 0 keeps everything
    if (top_k > 0 && top_k < V) {
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = thr | (1u << bit);
            int cnt = 0;
            for (int j = tid; j < V; j += 256) cnt += (fkey(xs[j]) >= cand) ? 1 : 0;
            if (block_sum_int(cnt, redi, tid) >= top_k) thr = cand;
        }
    }
    float lsum = 0.f;
    for (int j = tid; j < V; j += 256) {
        const float e = (fkey(xs[j]) >= thr) ? expf(xs[j] - gmax) : 0.f;
        xs[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) redf[wave] = lsum;
    __syncthreads();
    const float tot = ((redf[0] + redf[1]) + redf[2]) + redf[3];
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int j = tid; j < V; j += 256) {
        float p = xs[j] / tot;
        if (noise) p = p / noise[(long)b * V + j];
        if (p > best) { best = p; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { redf[wave] = best; redj[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (redf[w] > best || (redf[w] == best && redj[w] < bi)) { best = redf[w]; bi = redj[w]; }
        out[(long)b * out_stride] = bi;
    }
}

extern "C" int ccvs_sample_topk(const float* logits, int64_t ld, const float* noise, int64_t* out, int64_t out_stride, int32_t B, int32_t V,
                                int32_t top_k, float temperature, void* stream) {
    CCVS_REQUIRE(logits && out, "ccvs_sample_topk: null pointer");
    CCVS_REQUIRE(B > 0 && V > 0 && temperature > 0.f, "ccvs_sample_topk: bad arguments");
    const size_t smem = (size_t)(V + 16) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sample_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    CCVS_REQUIRE(smem <= 160 * 1024, "ccvs_sample_topk: vocabulary %d too large", V);
    hipLaunchKernelGGL(sample_topk_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, logits, (long)ld, noise, out, (long)out_stride, V,
                       top_k, temperature);
    CCVS_CHECK_LAUNCH("ccvs_sample_topk");
    return CCVS_OK;
}
