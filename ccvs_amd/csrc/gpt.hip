// Transformer kernels: embedding, LayerNorm, nn.Linear GEMM on the fp32 matrix cores,
// KV-cached causal attention and the top-k / softmax / pick of the sampling loop.
// Reference: models/skip_vid_generator/models/mingpt.py:33-117,186-305 and
// models/skip_vid_generator/models/transformer_model.py:256-260,395-409.
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------
// x[(b,t)][:] = tok_emb[idx[b*idx_sB + t]] + pos_table[pos_off[b] + pos0 + t]
// (mingpt.py:234-236,242-244; the factored s_emb/t_emb (+delta_length) or flat pos_emb rows
// are pre-summed by the host into pos_table once per call).
// ---------------------------------------------------------------------------------------
// -DCCVS_TOKEN_PRIO=n (an experiment, tools/r05/prio_ab.sh): the kernels of a decode step raise their waves' issue priority --
// they are few, short and mostly waiting on memory, beside convolution waves that issue MFMAs back to back.
#ifdef CCVS_TOKEN_PRIO
#define TOKEN_PRIO() __builtin_amdgcn_s_setprio(CCVS_TOKEN_PRIO)
#else
#define TOKEN_PRIO() ((void)0)
#endif
__global__ __launch_bounds__(256) void gpt_embed_kernel(const int64_t* __restrict__ idx, long idx_sB, const int32_t* __restrict__ pos_off,
                                                        int pos0, const int32_t* __restrict__ pos_dev, int grp_rows, int Tq,
                                                        const float* __restrict__ tok, const float* __restrict__ pos,
                                                        float* __restrict__ x, long total, int C, int vocab) {
    TOKEN_PRIO();
    // device-resident position: lets a captured hipGraph replay at advancing positions; with row groups (grp_rows > 0)
    // batch row b reads the word of its group, pos_dev[b / grp_rows]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        const long b = r / Tq;
        const int tq = (int)(r - b * Tq);
        long t = idx[b * idx_sB + tq];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        const long prow = (pos_off ? pos_off[b] : 0) + pos0 + (pos_dev ? pos_dev[grp_rows > 0 ? b / grp_rows : 0] : 0) + tq;
        x[i] = tok[t * C + c] + pos[prow * C + c];
    }
}

extern "C" int ccvs_gpt_embed(const int64_t* idx, int64_t idx_sB, const int32_t* pos_off, int32_t pos0, const int32_t* pos_dev, int32_t Tq,
                              const float* tok_emb, const float* pos_table, float* x, int32_t B, int32_t C, int32_t vocab, void* stream) {
    CCVS_REQUIRE(idx && tok_emb && pos_table && x, "ccvs_gpt_embed: null pointer");
    CCVS_REQUIRE(B > 0 && Tq > 0 && C > 0 && vocab > 0 && (pos0 >= 0 || pos_dev), "ccvs_gpt_embed: empty tensor");
    const long total = (long)B * Tq * C;
    hipLaunchKernelGGL(gpt_embed_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, idx, (long)idx_sB, pos_off,
                       pos0, pos_dev, 0, Tq, tok_emb, pos_table, x, total, C, vocab);
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
// nn.Linear as a weight stream on v_mfma_f32_16x16x4_f32:  y[M,N] = epi(x[M,K] @ W[N,K]^T + b).
//
// The decode regime (M = batch = 16 rows) is a pure HBM stream of W, split over many small
// dependent launches: what matters is bytes in flight per CU and a short critical path.
//   * workgroup = 16 rows x 16 columns, 8 waves = 8 K slices; each lane issues ALL its float4
//     loads of a 128-deep K batch before the first MFMA (W: row n = lane&15, k = k0+4*(lane>>4)+t;
//     x: same k map, row = lane&15, served by L1/L2), so a wave keeps 16 KiB in flight;
//   * no LDS staging, no barrier in the main loop; the 8 K slices are summed through LDS in a
//     fixed order (bitwise reproducible, no float atomics);
//   * LayerNorm is folded in algebraically: with W' = W*gamma, s[n] = sum_k W'[n][k] and
//     b' = b + W beta (packed once by the host),
//         LN(x) @ W^T + b = rstd * (x @ W'^T - mean * s) + b'
//     so the GEMM runs on the raw activations and the row statistics (sum x, sum x^2, which the
//     lanes accumulate from the very x values they feed to the MFMA) are applied in the
//     epilogue: the separate LayerNorm launch and its round trip through HBM disappear;
//   * the QKV projection can scatter its K and V column blocks straight into the KV cache at a
//     host- or device-resident position (no separate append launch).
// Larger M (prefill) runs the same kernel over ceil(M/16) row blocks, W then coming from L2.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// row statistics of the folded LayerNorm: one definition, floating-point contraction off, so that the plain and the
// row-blocked kernel (and any future variant) accumulate bit-identically
__device__ __forceinline__ void ln_accum(const float4& v, float& sx, float& sxx) {
#pragma clang fp contract(off)
    sx += (v.x + v.y) + (v.z + v.w);
    sxx += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
}

__device__ __forceinline__ void ln_accum(const f32x4& v, float& sx, float& sxx) {
#pragma clang fp contract(off)
    sx += (v[0] + v[1]) + (v[2] + v[3]);
    sxx += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
}

// 16 bytes at descriptor `r`, per-lane byte offset `voff` + wave-uniform byte offset `soff` (buffer_load_dwordx4 ... offen)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>   // AUX = 2: non-temporal (streamed once: do not displace what other kernels keep in L2 / the Infinity Cache)
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    return __builtin_bit_cast(f32x4, v);
}

// ---------------------------------------------------------------------------------------
// Agent-scope ("sc1") accesses for data that one workgroup hands to another INSIDE a launch (the persistent decode step below; the
// split-K slabs since round 2).  Per-XCD L2s are not coherent and a CU's L1 is never refreshed by another CU's stores
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"): the producer stores sc1 (write-through)
// and drains (s_waitcnt vmcnt(0)) before its workgroup signals; EVERY consumer load of those bytes is an sc1 load (it bypasses
// L1) issued after the consumer's poll has matched and its workgroup has passed a barrier.  COH = false: the plain access (a
// kernel boundary orders it), so that one body serves the launch chain and the persistent kernel with identical arithmetic.
// ---------------------------------------------------------------------------------------
template <bool COH>
__device__ __forceinline__ float ld_act(const float* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool COH>
__device__ __forceinline__ void st_act(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// threadIdx.x, opaque to the optimiser when OPAQUE: inside the persistent step's phase loop every lane-derived constant of every
// phase body (LDS offsets, row / column maps, masks -- dozens of registers) is loop-invariant and gets hoisted in front of the loop,
// where it stays live across all the other phases (with a 64-register cap: 19 spills, all of them such values); behind an asm
// statement the optimiser cannot move, each body recomputes its handful of shifts and masks where it runs.
template <bool OPAQUE>
__device__ __forceinline__ int thread_id() {
    int t = threadIdx.x;
    if constexpr (OPAQUE) asm volatile("" : "+v"(t));
    return t;
}
// a pointer read from the persistent step's phase table (constant memory) is a GENERIC pointer to the compiler: the cast through
// address_space(1) makes its accesses global_* instead of flat_* (COH = false: the kernel's own argument, already global)
template <bool COH, typename T>
__device__ __forceinline__ T* gp_(T* p) {
    if constexpr (COH) return (T*)(__attribute__((address_space(1))) T*)p;
    else return p;
}
// 16 bytes, sc1 (global_load_dwordx4 ... sc1 through a one-off buffer descriptor: aux bit 4)
__device__ __forceinline__ f32x4 ld_act4_sc1(const float* p) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 16, 0x00020000);
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 16);
    return __builtin_bit_cast(f32x4, v);
}

struct Gemm16 {
    const float* x; long ldx;
    const float* w; const float* bias; const float* res; float* y; long ldy;
    int M, N, K, epi, ks;
    const float* ln_s; float ln_eps;
    float* kcache; float* vcache; int C, H, D, Tq, Tmax, pos0; const int32_t* pos_dev;
    int seq;           // 1: whole-sequence call (CCVS_GEMM_SEQ): the row-blocked form whatever M is (see launch_gemm16)
    int grp_rows;      // > 0: batch row b takes its device-resident position from pos_dev[b / grp_rows] (row groups of a decode step)
    int kz;            // K slices across workgroups (gridDim.z); > 1 only with a workspace
    float* ws_slabs;   // [tile][kz][64 lanes][4] partial accumulators
    int* ws_count;     // [tile] arrival counters, zero between launches
};

#define GEMM_U 4  // one-block form: K steps (of 16) whose loads are issued together (4 float4 of W + 4 of x per lane: the 48-VGPR budget)
#define GEMM_U_RB 8  // the same for the prefill form (512 threads, its own CU)
#define GEMM_WS_TILES 1024  // 16 x 16 output tiles a workspace covers (64 column tiles x 16 row blocks: every split-K launch fits)
#define GEMM_DECODE_MAX_M 256  // up to this many rows the weight-stream kernel below runs; beyond, the prefill form

// LayerNorm statistics -> (mean, rstd) and the epilogue of one output value: single definitions with floating-point
// contraction off, so that every instantiation of the kernel below rounds identically (a row's result must not depend on
// how many other rows share its launch: the grouped decode step is bit-identical to one step per batch).
__device__ __forceinline__ void ln_finish(float a, float b, int K, float eps, float& mean, float& rstd) {
#pragma clang fp contract(off)
    mean = a / K;
    const float var = fmaxf(b / K - mean * mean, 0.f);
    rstd = rsqrtf(var + eps);
}
__device__ __forceinline__ float gemm_epilogue(float v, bool ln, float rstd, float mean, float sn, float bv, int epi, float rv) {
#pragma clang fp contract(off)
    if (ln) v = rstd * (v - mean * sn);
    v += bv;
    if (epi == 1) v = gelu_erf(v);
    if (epi == 2) v += rv;
    return v;
}

// The weight-stream kernel (M <= GEMM_DECODE_MAX_M rows: decode steps, small prefills).
//
// FOOTPRINT FIRST.  The decode step runs beside the frame decoder of another batch, whose convolution workgroups (8 waves x
// 210-225 VGPRs, ~100 KB of LDS) hold every CU: two such waves per SIMD leave 48 VGPRs per lane and ~56 KB of LDS.  A
// workgroup that fits into that remainder is dispatched at once next to the convolution; one that does not waits for a
// convolution workgroup to retire (~100 us).  Measured with a chain of dependent 16 MB weight-stream launches beside the
// real decoder (tools/chain_probe.py, profiles/r03_chain_probe.txt): 512 threads x 200 VGPRs 41 us per launch, 512 x 100
// (the round-2 kernel) 16 us, 256 x 56 14 us, 256 x <= 48 VGPRs with <= 56 KB LDS 10 us (5.8 us alone).  Hence: 256 threads,
// at most 48 VGPRs, a few KB of LDS -- bytes in flight per workgroup are what the registers allow (4 float4 of W and of x
// per lane), and the launch gets its bandwidth from the NUMBER of workgroups.
//   * workgroup = 16 rows x 16 columns (up to 16 rows in the launch; 2 x 2 such blocks beyond: see the template), 4 waves = 4 K
//     slices; each lane issues the float4 loads of a 64-deep K batch of W
//     (row n = lane&15, k = k0+4*(lane>>4)+t) and of x (same k map, row = lane&15, served by L2) before the MFMAs of the batch;
//   * no LDS staging, no barrier in the main loop; the K slices are summed through LDS in a fixed order (wave 0 first;
//     bitwise reproducible, no float atomics); deep K (>= 2048) also splits over `kz` workgroups (last-arriver reduction);
//   * more rows than 16 (the stacked rows of several batches in one decode step, ccvs_gpt_decode.groups) = more row blocks
//     in gridDim.y: they run at the same time on other CUs and share the weight tile through L2 / the Infinity Cache.  A
//     row's arithmetic depends on N and K only (K slicing, MFMA order, slab order, epilogue), never on M;
//   * LayerNorm folded in algebraically and the K / V column blocks of the QKV projection scattered straight into the KV
//     cache, as described above.
// The operands of the address arithmetic come first and as plain kernel arguments: with -mllvm
// -amdgpu-kernarg-preload-count they are delivered in SGPRs when the wave starts (gfx950 kernarg preload), so the weight
// and activation loads of this latency-bound kernel go out without first waiting for a scalar load of the argument block.
#define GEMM_WAVES 4
// RB x CB blocks of 16 x 16 outputs per workgroup (round 5).  With one block per workgroup the rows' activations are re-read by
// every column tile and the weights by every row block: a 64-row step moved 8 x its unique bytes through the CUs' vector-memory
// pipelines (33.5 MB per 1024 x 1024 layer for 4 MB of weights), and that pipeline -- ~12 bytes per clock and CU, L2 hits
// included -- is what the token loops and the frame decoder beside them compete for: with those re-reads switched off
// (every lane reading row 0 of x and of W -- wrong results; an experiment switch of the round, removed again: reading it cost the one-block form 0.5 us in front of its first load) the bench line went from 213 to 237 frames/s (profiles/r05_gemm_tile_ab.txt).  A 2 x 2 block
// tile halves both re-reads at 8 + 8 + 16 registers for the two operands and the four accumulators (one 16-deep K batch in
// flight instead of four: U) -- 32 VGPRs + 16 AGPRs, the same 48; the blocks' arithmetic is the one-block kernel's, bit for bit
// -- the same K slices per wave, the same MFMA order per block, the same slice order in the reduction -- so a row does not
// depend on the tile it falls into.  Same box, 20 batches: 210.5 -> 227.8 frames/s, the token step beside the decoder 7.88 ->
// 7.22 ms, the convolutions beside it 87 -> 95 TFLOP/s; alone the step costs the same (3.17 against 3.25 ms).  Larger tiles
// (4 x 2, 2 x 4: 112 registers, a quarter of the workgroups) lose: 206-208 frames/s, 4.07 ms alone.
// LDS of one tile: the K slices' partial blocks, their LayerNorm sums, the rows' (mean, rstd)
#define GEMM16_RED_WORDS(NB) (GEMM_WAVES * (NB) * 64 * 4)
#define GEMM16_STAT_WORDS(RB) (GEMM_WAVES * (RB) * 16 * 2)
#define GEMM16_FIN_WORDS(RB) ((RB) * 16 * 2)
// One tile (block (bx, by, bz) of the launch's grid) of the weight-stream GEMM: the body of gemm16_kernel, and of the GEMM phases of
// the persistent decode step (gpt_step_kernel: COH = true -- the activations x / res / y / the new cache rows are handed over
// between workgroups of ONE launch, so they are read and written with sc1 accesses; the arithmetic is the same instruction for
// instruction, so a row's bits do not depend on which of the two forms computed it).
// PT: `Gemm16` (the kernel's own argument block) or `__attribute__((address_space(4))) Gemm16` (an entry of the persistent step's phase
// table in constant memory: every `p.field` is then a scalar load at its point of use -- the epilogue's operands are not held
// in registers across the K loop -- and the pointers read from it are generic, hence GP() = the cast through the global address space).
template <int WNT, int RB, int CB, int U, bool COH, typename PT>   // WNT = 2: weights with the non-temporal policy
__device__ __forceinline__ void gemm16_tile(const float* __restrict__ x_, const float* __restrict__ w_, long ldx_, int K_, int N_, int M_, int ks_,
                                            int kz_, const PT& p, int bx_, int by_, int bz_, float* __restrict__ red, float* __restrict__ stat,
                                            float* __restrict__ fin) {
#define GP(ptr) gp_<COH>(ptr)
    constexpr int NB = RB * CB;
    constexpr int XAUX = COH ? 16 : 0;   // sc1 on the activation loads
    const int tid = thread_id<COH>(), lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = by_ * 16 * RB, ncol0 = bx_ * 16 * CB;
    const int kper = K_ / (ks_ * kz_);
    const int kbase = bz_ * (K_ / kz_);
    const bool active = wave < ks_;
    // Buffer loads: a wave-uniform 128-bit descriptor per tensor in SGPRs + ONE 32-bit byte offset per lane and block (the
    // K position goes into the scalar offset / the instruction's immediate) instead of 64-bit pointer pairs and their
    // adds -- the 48-VGPR budget.  The launcher checks that both tensors stay below 2^31 bytes.
    const unsigned koff = kbase + (active ? wave : 0) * kper + 4 * g;
    unsigned wofs[CB], xofs[RB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {   // tails: computed on a valid row, dropped at the store
        const int nrow = min(ncol0 + 16 * cb + li, N_ - 1);
        wofs[cb] = ((unsigned)nrow * (unsigned)K_ + koff) * 4u;
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int mrow = min(m0 + 16 * rb + li, M_ - 1);
        xofs[rb] = ((unsigned)mrow * (unsigned)ldx_ + koff) * 4u;
    }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_), 0, N_ * K_ * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_), 0, (int)(M_ * ldx_ * 4), 0x00020000);
    f32x4 acc[RB][CB];
    float sx[RB], sxx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        sx[rb] = 0.f; sxx[rb] = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = {0.f, 0.f, 0.f, 0.f};
    }
    if (active) {
        // full batches: U unconditional 16-byte loads per block row of W and of x in flight per lane, then the MFMAs (one
        // accumulator chain per block: the dependent MFMAs of a chain cost a few cycles each)
        int k0 = 0;
        for (; k0 + 16 * U <= kper; k0 += 16 * U) {
            f32x4 wv[CB][U], xv[RB][U];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int u = 0; u < U; ++u) wv[cb][u] = buf_load4<WNT>(wr, wofs[cb], k0 * 4 + 64 * u);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int u = 0; u < U; ++u) xv[rb][u] = buf_load4<XAUX>(xr, xofs[rb], k0 * 4 + 64 * u);
            __builtin_amdgcn_sched_barrier(0);   // every load of the batch goes out before its first MFMA (hipcc otherwise sinks half of them between the MFMAs)
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[rb][u][t], wv[cb][u][t], acc[rb][cb], 0, 0, 0);
                    if (p.ln_s) ln_accum(xv[rb][u], sx[rb], sxx[rb]);
                }
            }
        }
        for (; k0 < kper; k0 += 16) {  // remainder (small K only)
            f32x4 w1[CB], x1[RB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) w1[cb] = buf_load4<WNT>(wr, wofs[cb], k0 * 4);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) x1[rb] = buf_load4<XAUX>(xr, xofs[rb], k0 * 4);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[rb][t], w1[cb][t], acc[rb][cb], 0, 0, 0);
                if (p.ln_s) ln_accum(x1[rb], sx[rb], sxx[rb]);
            }
        }
    }
    if (p.ln_s) {  // lanes li, li+16, li+32, li+48 hold the four k-groups of row li
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            float a = sx[rb], b = sxx[rb];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (g == 0) { stat[((wave * RB + rb) * 16 + li) * 2] = a; stat[((wave * RB + rb) * 16 + li) * 2 + 1] = b; }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) *reinterpret_cast<f32x4*>(red + ((wave * NB + rb * CB + cb) * 64 + lane) * 4) = acc[rb][cb];
    __syncthreads();
    if (p.ln_s && tid < 16 * RB) {
        const int rb = tid >> 4, r = tid & 15;
        float a = 0.f, b = 0.f;
        for (int w = 0; w < GEMM_WAVES; ++w) { a += stat[((w * RB + rb) * 16 + r) * 2]; b += stat[((w * RB + rb) * 16 + r) * 2 + 1]; }
        float mean, rstd;
        ln_finish(a, b, p.K, p.ln_eps, mean, rstd);
        fin[tid * 2] = mean;
        fin[tid * 2 + 1] = rstd;
    }
    __syncthreads();
    // wave w finishes the blocks w, w + 4, ... (one block per workgroup: wave 0, as before)
#pragma unroll 1
    for (int blk = wave; blk < NB; blk += GEMM_WAVES) {
        const int rb = blk / CB, cb = blk - rb * CB;
        f32x4 a4;
        if constexpr (NB == 1) a4 = acc[0][0];   // (wave 0: its own slice is still in registers)
        else a4 = *reinterpret_cast<const f32x4*>(red + (blk * 64 + lane) * 4);
#pragma unroll
        for (int s = 1; s < GEMM_WAVES; ++s) a4 += *reinterpret_cast<const f32x4*>(red + ((s * NB + blk) * 64 + lane) * 4);
        const int mb0 = m0 + 16 * rb, nb0 = ncol0 + 16 * cb;
        if (mb0 >= p.M || nb0 >= p.N) continue;   // a block wholly outside the matrix (ragged row / column counts)

        bool finisher = true;
        if (p.kz > 1) {
            // K is also split over gridDim.z workgroups (few output columns: keeps all 256 CUs streaming W).
            // Each publishes its 16x16 partial, the LAST arriver sums the slabs in slice order (bitwise
            // reproducible) and runs the epilogue.  The slabs are written and read with agent-scope (sc1)
            // accesses, which go through to memory and bypass the per-XCD L2s: no release / acquire fence is
            // needed -- on this chip a fence is an L2 write-back + invalidate costing several us, and all the
            // hand-off needs is slab stores acknowledged (vmcnt 0) before the ticket, and slab loads after it.
            const int tile = (mb0 >> 4) * ((p.N + 15) >> 4) + (nb0 >> 4);   // the 16 x 16 block's own slabs and counter
            float* slabs = GP(p.ws_slabs) + (long)tile * p.kz * 256;
            float* mine = slabs + (bz_ * 64 + lane) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) __hip_atomic_store(mine + c, a4[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            int ticket = 0;
            if (lane == 0) ticket = __hip_atomic_fetch_add(GP(p.ws_count) + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            finisher = ticket == p.kz - 1;
            if (finisher) {
#pragma unroll
                for (int c = 0; c < 4; ++c) a4[c] = __hip_atomic_load(slabs + lane * 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int z = 1; z < p.kz; ++z) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) a4[c] += __hip_atomic_load(slabs + (z * 64 + lane) * 4 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0) __hip_atomic_store(GP(p.ws_count) + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // D[row = 4*g + r][col = li]
        const int col = nb0 + li;
        if (finisher && col < p.N) {
            const float bv = p.bias ? GP(p.bias)[col] : 0.f;
            const float sn = p.ln_s ? GP(p.ln_s)[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mb0 + 4 * g + r;
                if (row >= p.M) continue;
                const float rv = p.epi == 2 ? ld_act<COH>(GP(p.res) + (long)row * p.ldy + col) : 0.f;
                const float v = gemm_epilogue(a4[r], p.ln_s != nullptr, fin[(rb * 16 + 4 * g + r) * 2 + 1], fin[(rb * 16 + 4 * g + r) * 2], sn, bv, p.epi, rv);
                if (p.kcache && col >= p.C) {
                    const int cc = col - p.C;
                    float* cache = cc >= p.C ? GP(p.vcache) : GP(p.kcache);
                    const int c2 = cc >= p.C ? cc - p.C : cc;
                    const int h = c2 / p.D, d = c2 - h * p.D;
                    const int b = row / p.Tq, t = row - b * p.Tq;
                    const int pos = p.pos0 + (p.pos_dev ? GP(p.pos_dev)[p.grp_rows > 0 ? b / p.grp_rows : 0] : 0);
                    if (pos + t < p.Tmax) st_act<COH>(cache + (((long)b * p.H + h) * p.Tmax + pos + t) * p.D + d, v);
                } else {
                    st_act<COH>(GP(p.y) + (long)row * p.ldy + col, v);
                }
            }
        }
    }
#undef GP
}

// -DCCVS_GEMM16_NUM_VGPR=48 (an experiment, tools/r06/run3.sh): hipcc then allocates the one-block form 46 registers and no AGPRs
// instead of 50 + 4 (56 allocated: it does not fit beside two 232-register convolution waves), the 2 x 2 form 48 + 0 instead of 32 + 16.
#ifdef CCVS_GEMM16_NUM_VGPR
#define GEMM16_ATTR __attribute__((amdgpu_num_vgpr(CCVS_GEMM16_NUM_VGPR)))
#else
#define GEMM16_ATTR
#endif
template <int WNT, int RB, int CB, int U>
__global__ __launch_bounds__(64 * GEMM_WAVES) GEMM16_ATTR void gemm16_kernel(const float* __restrict__ x_, const float* __restrict__ w_, long ldx_, int K_, int N_,
                                                                int M_, int ks_, int kz_, Gemm16 p) {
    TOKEN_PRIO();
    __shared__ __attribute__((aligned(16))) float red[GEMM16_RED_WORDS(RB * CB)];
    __shared__ float stat[GEMM16_STAT_WORDS(RB)];
    __shared__ float fin[GEMM16_FIN_WORDS(RB)];
    gemm16_tile<WNT, RB, CB, U, false>(x_, w_, ldx_, K_, N_, M_, ks_, kz_, p, blockIdx.x, blockIdx.y, blockIdx.z, red, stat, fin);
}

// Prefill form (M > GEMM_DECODE_MAX_M rows): one workgroup computes RB row blocks of 16 against the SAME 16 output columns,
// so a weight tile is fetched once per 16*RB rows.  K sliced over the 8 waves only (kz = 1: with hundreds of row blocks
// the chip is full without split-K), argument block read the ordinary way.
template <int RB>
__global__ __launch_bounds__(512) void gemm16_rb_kernel(Gemm16 p) {
    __shared__ __attribute__((aligned(16))) float red[RB * 8 * 64 * 4];
    __shared__ float stat[RB * 8 * 16 * 2];
    __shared__ float fin[RB * 16 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16 * RB, ncol0 = blockIdx.x * 16;
    const int nrow = min(ncol0 + li, p.N - 1);
    const int kper = p.K / p.ks;
    const bool active = wave < p.ks;
    const float* wp = p.w + (long)nrow * p.K + (active ? wave : 0) * kper + 4 * g;
    const float* xp[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xp[rb] = p.x + (long)min(m0 + 16 * rb + li, p.M - 1) * p.ldx + (active ? wave : 0) * kper + 4 * g;
    f32x4 acc0[RB], acc1[RB];
    float sx[RB], sxx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { acc0[rb] = {0.f, 0.f, 0.f, 0.f}; acc1[rb] = {0.f, 0.f, 0.f, 0.f}; sx[rb] = 0.f; sxx[rb] = 0.f; }
    if (active) {
        int k0 = 0;
        for (; k0 + 16 * GEMM_U_RB <= kper; k0 += 16 * GEMM_U_RB) {
            float4 wv[GEMM_U_RB];
#pragma unroll
            for (int u = 0; u < GEMM_U_RB; ++u) wv[u] = *reinterpret_cast<const float4*>(wp + k0 + 16 * u);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                float4 xv[GEMM_U_RB];
#pragma unroll
                for (int u = 0; u < GEMM_U_RB; ++u) xv[u] = *reinterpret_cast<const float4*>(xp[rb] + k0 + 16 * u);
#pragma unroll
                for (int u = 0; u < GEMM_U_RB; ++u) {
                    acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].x, wv[u].x, acc0[rb], 0, 0, 0);
                    acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].y, wv[u].y, acc1[rb], 0, 0, 0);
                    acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].z, wv[u].z, acc0[rb], 0, 0, 0);
                    acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u].w, wv[u].w, acc1[rb], 0, 0, 0);
                    if (p.ln_s) ln_accum(xv[u], sx[rb], sxx[rb]);
                }
            }
        }
        for (; k0 < kper; k0 += 16) {
            const float4 w1 = *reinterpret_cast<const float4*>(wp + k0);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float4 x1 = *reinterpret_cast<const float4*>(xp[rb] + k0);
                acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, w1.x, acc0[rb], 0, 0, 0);
                acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, w1.y, acc1[rb], 0, 0, 0);
                acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, w1.z, acc0[rb], 0, 0, 0);
                acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, w1.w, acc1[rb], 0, 0, 0);
                if (p.ln_s) ln_accum(x1, sx[rb], sxx[rb]);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const f32x4 acc = acc0[rb] + acc1[rb];
        if (p.ln_s) {
            float a = sx[rb], b = sxx[rb];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (g == 0) { stat[((rb * 8 + wave) * 16 + li) * 2] = a; stat[((rb * 8 + wave) * 16 + li) * 2 + 1] = b; }
        }
        *reinterpret_cast<f32x4*>(red + ((rb * 8 + wave) * 64 + lane) * 4) = acc;
    }
    __syncthreads();
    if (p.ln_s && tid < 16 * RB) {
        const int rb = tid >> 4, r = tid & 15;
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 8; ++w) { a += stat[((rb * 8 + w) * 16 + r) * 2]; b += stat[((rb * 8 + w) * 16 + r) * 2 + 1]; }
        float mean, rstd;
        ln_finish(a, b, p.K, p.ln_eps, mean, rstd);
        fin[tid * 2] = mean;
        fin[tid * 2 + 1] = rstd;
    }
    __syncthreads();
    if (wave >= RB) return;  // wave rb finishes row block rb
    const int rb = wave;
    f32x4 acc = *reinterpret_cast<const f32x4*>(red + ((rb * 8) * 64 + lane) * 4);
#pragma unroll
    for (int s2 = 1; s2 < 8; ++s2) acc += *reinterpret_cast<const f32x4*>(red + ((rb * 8 + s2) * 64 + lane) * 4);
    const int col = ncol0 + li;
    if (col >= p.N) return;
    const float bv = p.bias ? p.bias[col] : 0.f;
    const float sn = p.ln_s ? p.ln_s[col] : 0.f;
    int pos0 = p.pos0;
    if (p.kcache && p.pos_dev) pos0 += *p.pos_dev;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * rb + 4 * g + r;
        if (row >= p.M) continue;
        const float v = gemm_epilogue(acc[r], p.ln_s != nullptr, fin[(rb * 16 + 4 * g + r) * 2 + 1], fin[(rb * 16 + 4 * g + r) * 2], sn, bv, p.epi,
                                      p.epi == 2 ? p.res[(long)row * p.ldy + col] : 0.f);
        if (p.kcache && col >= p.C) {
            const int cc = col - p.C;
            float* cache = cc >= p.C ? p.vcache : p.kcache;
            const int c2 = cc >= p.C ? cc - p.C : cc;
            const int h = c2 / p.D, d = c2 - h * p.D;
            const int b = row / p.Tq, t = row - b * p.Tq;
            if (pos0 + t < p.Tmax) cache[(((long)b * p.H + h) * p.Tmax + pos0 + t) * p.D + d] = v;
        } else {
            p.y[(long)row * p.ldy + col] = v;
        }
    }
}

// Whole-sequence form (CCVS_GEMM_SEQ: prefill, teacher-forced forward, re-prefill of a slid window): a dense LDS-tiled GEMM
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, 157 TFLOP/s peak).  With hundreds to thousands of rows
// the weights are no longer a stream to be read once but operands to be re-used: the row-blocked kernel above fetches every
// operand from global memory per 64 rows x 16 columns (52-82 TFLOP/s).
//   * workgroup = 4 waves = a 128 x 128 output tile, wave w the 64 x 64 quadrant (w >> 1, w & 1): 2 x 2 MFMA blocks, 64
//     accumulator registers; K in stages of 16;
//   * both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, the next stage is in flight
//     while this one is multiplied), double-buffered, one barrier per stage.  LDS image of an operand stage:
//     [k quad q = 0..3][row 0..127][4 floats], quads 129 units apart -- a wave-instruction of the DMA fills 64 consecutive
//     rows of one quad (the per-lane SOURCE address does the transposition), and the fragment read of a 32-row block is 32
//     consecutive 16-byte units per lane half: conflict-free ds_read_b128;
//   * lane half h reads k = 8 s + 4 h .. + 3 of sub-step s for both operands, and MFMA j of the sub-step contracts the pair
//     (8 s + j, 8 s + 4 + j): the contraction order over k is free as long as both operands agree;
//   * the folded LayerNorm's row statistics are summed from the x stages in LDS (thread t: row t >> 1, k quads 2 (t & 1), +1);
//   * epilogue as everywhere (LayerNorm fold, bias, GELU, residual, K / V scatter into the cache); a 32 x 32 block leaves as
//     128-byte row segments.
// A row's arithmetic depends on (N, K) only -- never on M or on the tile the row falls into -- so a batch prefilled alone and
// the same batch inside a stacked token group agree bit for bit (tests/test_pipeline_gpu.py).
// Workgroup order: XCD j (linear id % 8) walks the row tiles j, j + 8, ...: the workgroups that share a row tile of x meet
// in one L2, and every XCD streams the weights once per round of eight row tiles.
#define GS_BM 128
#define GS_QS (GS_BM + 1)   // 16-byte units between the k quads of an operand stage
typedef __attribute__((address_space(3))) void gs_lds_void;
__global__ __launch_bounds__(256, 4) void gemm_seq_kernel(Gemm16 p, int row_tiles, int col_tiles) {
    // two separate LDS variables, one per stage buffer, and the stage loop unrolled by two: hipcc orders a ds_read behind an
    // in-flight LDS-DMA unless the two provably touch different LDS objects -- with one array indexed by (stage & 1) every
    // fragment read waited (vmcnt 0) for the DMA of the NEXT stage issued just before it, and nothing overlapped
    __shared__ __attribute__((aligned(16))) f32x4 sm0[2 * 4 * GS_QS];   // [x | w][quad][row]
    __shared__ __attribute__((aligned(16))) f32x4 sm1[2 * 4 * GS_QS];
    __shared__ float fin[GS_BM * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int rt, ct;
    {
        const int w = blockIdx.x;
        if (row_tiles % 8 == 0) {
            const int i = w >> 3;
            rt = (w & 7) + 8 * (i / col_tiles);
            ct = i % col_tiles;
        } else {
            rt = w / col_tiles;
            ct = w - rt * col_tiles;
        }
    }
    const int m0 = rt * GS_BM, n0 = ct * GS_BM;
    // DMA: 16 wave-instructions per stage, 4 per wave: wave q fills k quad q of both operands, two halves of 64 rows each.
    // buffer_load ... lds: a 128-bit descriptor per operand in SGPRs, a per-lane byte offset that never changes and the
    // stage's K offset as the scalar offset -- no address VGPR is rewritten inside the loop (with global_load_lds hipcc builds
    // a 64-bit address in a VGPR pair per DMA and waits, vmcnt 0, for the previous DMA before it overwrites the pair).
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long)p.M * p.ldx * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long)p.N * p.K * 4), 0x00020000);
    unsigned xoff[2], woff[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        xoff[hf] = (unsigned)((long)min(m0 + 64 * hf + lane, p.M - 1) * p.ldx + 4 * wave) * 4u;
        woff[hf] = (unsigned)((long)min(n0 + 64 * hf + lane, p.N - 1) * p.K + 4 * wave) * 4u;
    }
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bi][bj][r] = 0.f;
    float sx = 0.f, sxx = 0.f;
    const int nst = p.K / 16;
#define GS_DMA(k0, SM)                                                                                             \
    _Pragma("unroll") for (int hf = 0; hf < 2; ++hf) {                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (gs_lds_void*)((SM) + wave * GS_QS + 64 * hf), 16, xoff[hf], (k0) * 4, 0, 0);               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (gs_lds_void*)((SM) + 4 * GS_QS + wave * GS_QS + 64 * hf), 16, woff[hf], (k0) * 4, 0, 0);  \
    }
#define GS_STAGE(SM, NEXT_DMA)                                                                                     \
    {                                                                                                              \
        const f32x4* xa = (SM);                                                                                    \
        const f32x4* wb = (SM) + 4 * GS_QS;                                                                        \
        f32x4 a[2][2], b[2][2], t0, t1;                                                                            \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                         \
            const int q = 2 * s2 + h;                                                                              \
            _Pragma("unroll") for (int bi = 0; bi < 2; ++bi) a[s2][bi] = xa[q * GS_QS + 64 * wr + 32 * bi + l31];  \
            _Pragma("unroll") for (int bj = 0; bj < 2; ++bj) b[s2][bj] = wb[q * GS_QS + 64 * wc + 32 * bj + l31];  \
        }                                                                                                          \
        if (p.ln_s) {                                                                                              \
            const int r = tid >> 1, q0 = 2 * (tid & 1);                                                            \
            t0 = xa[q0 * GS_QS + r];                                                                               \
            t1 = xa[(q0 + 1) * GS_QS + r];                                                                         \
        }                                                                                                          \
        /* every LDS read of this stage is issued BEFORE the next stage's DMA: hipcc orders a ds_read behind any */ \
        /* LDS-DMA still in flight (s_waitcnt vmcnt(0)), which would serialise the stages                        */ \
        NEXT_DMA                                                                                                   \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
                _Pragma("unroll") for (int bi = 0; bi < 2; ++bi)                                                   \
                    _Pragma("unroll") for (int bj = 0; bj < 2; ++bj)                                               \
                        acc[bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2][bi][j], b[s2][bj][j], acc[bi][bj], 0, 0, 0); \
        if (p.ln_s) {                                                                                              \
            ln_accum(t0, sx, sxx);                                                                                 \
            ln_accum(t1, sx, sxx);                                                                                 \
        }                                                                                                          \
    }
    GS_DMA(0, sm0)
    for (int st = 0; st < nst; st += 2) {
        __syncthreads();                                   // stage st has landed (the barrier drains vmcnt) and the other buffer is free
        GS_STAGE(sm0, if (st + 1 < nst) { GS_DMA(16 * (st + 1), sm1) })
        if (st + 1 >= nst) break;
        __syncthreads();
        GS_STAGE(sm1, if (st + 2 < nst) { GS_DMA(16 * (st + 2), sm0) })
    }
#undef GS_DMA
#undef GS_STAGE
    if (p.ln_s) {
        sx += __shfl_xor(sx, 1, 64);
        sxx += __shfl_xor(sxx, 1, 64);
        if (!(tid & 1)) {
            float mean, rstd;
            ln_finish(sx, sxx, p.K, p.ln_eps, mean, rstd);
            fin[(tid >> 1) * 2] = mean;
            fin[(tid >> 1) * 2 + 1] = rstd;
        }
        __syncthreads();
    }
    int pos0 = p.pos0;
    if (p.kcache && p.pos_dev) pos0 += *p.pos_dev;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int col = n0 + 64 * wc + 32 * bj + l31;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
        const float sn = p.ln_s ? p.ln_s[col] : 0.f;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = 64 * wr + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int row = m0 + lr;
                if (row >= p.M) continue;
                const float v = gemm_epilogue(acc[bi][bj][r], p.ln_s != nullptr, p.ln_s ? fin[lr * 2 + 1] : 0.f, p.ln_s ? fin[lr * 2] : 0.f, sn, bv,
                                              p.epi, p.epi == 2 ? p.res[(long)row * p.ldy + col] : 0.f);
                if (p.kcache && col >= p.C) {
                    const int cc = col - p.C;
                    float* cache = cc >= p.C ? p.vcache : p.kcache;
                    const int c2 = cc >= p.C ? cc - p.C : cc;
                    const int hh = c2 / p.D, d = c2 - hh * p.D;
                    const int b = row / p.Tq, t = row - b * p.Tq;
                    if (pos0 + t < p.Tmax) cache[(((long)b * p.H + hh) * p.Tmax + pos0 + t) * p.D + d] = v;
                } else {
                    p.y[(long)row * p.ldy + col] = v;
                }
            }
        }
    }
}

// K slices across workgroups (split-K): spreads GEMMs with few output columns over the chip.  Pays only for deep K:
// the release/acquire hand-off costs ~3-4 us (measured).
static int getenv_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// K slices across workgroups.  A function of N and K only -- never of M -- so that a row's K partition, hence its rounding,
// does not depend on how many rows share the launch.
// Cache policy of the decode step's once-read streams (CCVS_DECODE_NT: bit 0 = keys / values of the attention, bit 1 = weights)
static int decode_nt() {
    static const int v = getenv_int("CCVS_DECODE_NT", 1);
    return v;
}

static int gemm_kz(const Gemm16& g) {
    static int kz_max = -1;
    if (kz_max < 0) { const char* e = getenv("CCVS_GEMM_KZ_MAX"); kz_max = e ? atoi(e) : 4; }
    static const int kz_min_k = getenv_int("CCVS_GEMM_KZ_MINK", 2048);   // the shallowest K that is split over workgroups
    const int tiles = cdiv(g.N, 16);
    int kz = 1;
    if (g.ws_slabs && !g.ln_s)
        while (g.K >= kz_min_k && kz < kz_max && tiles * kz * 2 <= 256 && g.K % (16 * GEMM_WAVES * kz * 2) == 0) kz *= 2;
    return kz;
}

static int launch_gemm16(Gemm16& g, hipStream_t st, const char* name) {
    if (!(g.x && g.w && g.y)) { ccvs_set_error("%s: null pointer", name); return CCVS_ERR_ARG; }
    if (!(g.M > 0 && g.N > 0 && g.K > 0)) { ccvs_set_error("%s: empty tensor", name); return CCVS_ERR_ARG; }
    if (g.K % 16 != 0 || g.ldx % 4 != 0) { ccvs_set_error("%s: K=%d must be a multiple of 16 (ldx %% 4 == 0)", name, g.K); return CCVS_ERR_ARG; }
    if (g.epi < 0 || g.epi > 2 || (g.epi == 2 && !g.res)) { ccvs_set_error("%s: bad epilogue", name); return CCVS_ERR_ARG; }
    // Which kernel runs is a property of the CALL KIND, never of how many rows share the launch: the two forms partition K
    // differently (4 waves x kz slices and one accumulator chain vs 8 waves and two chains), so their low bits differ, and a
    // row must come out the same whether its batch is prefilled alone or stacked with the other batches of a token group
    // (B t0 <= 256 alone, G B t0 > 256 stacked).  Whole-sequence calls (prefill, teacher-forced forward, re-prefill of a slid
    // window: CCVS_GEMM_SEQ in the epilogue word, Tq > 1 for the QKV form) always take the row-blocked form; single-position
    // calls (decode steps) take the weight-stream form up to GEMM_DECODE_MAX_M rows and the row-blocked one beyond.
    const bool decode_form = !g.seq && g.M <= GEMM_DECODE_MAX_M;
    if (decode_form && ((long)g.N * g.K * 4 >= (1L << 31) || (long)g.M * g.ldx * 4 >= (1L << 31))) {
        ccvs_set_error("%s: operand beyond 2^31 bytes (32-bit buffer offsets)", name);
        return CCVS_ERR_ARG;
    }
    if (!decode_form && g.grp_rows > 0) { ccvs_set_error("%s: row groups need M <= %d", name, GEMM_DECODE_MAX_M); return CCVS_ERR_ARG; }
    g.kz = decode_form ? gemm_kz(g) : 1;
    if (g.kz > 1 && cdiv(g.N, 16) * cdiv(g.M, 16) > GEMM_WS_TILES) g.kz = 1;   // cannot happen for M <= 256 (kz > 1 needs <= 64 column tiles)
    g.ks = decode_form ? GEMM_WAVES : 8;
    while (g.ks > 1 && g.K % (16 * g.ks * g.kz) != 0) g.ks >>= 1;
    static const int seq_dense = getenv_int("CCVS_GEMM_SEQ_DENSE", 1);   // 0: the row-blocked weight-stream form of rounds 2-3
    if (!decode_form && g.seq && seq_dense && (long)g.M * g.ldx * 4 < (1L << 31) && (long)g.N * g.K * 4 < (1L << 31)) {   // (buffer descriptors: 32-bit byte offsets)
        const int rt = cdiv(g.M, GS_BM), ct = cdiv(g.N, GS_BM);
        hipLaunchKernelGGL(gemm_seq_kernel, dim3(rt * ct), dim3(256), 0, st, g, rt, ct);
    } else if (!decode_form)
        hipLaunchKernelGGL((gemm16_rb_kernel<4>), dim3(cdiv(g.N, 16), cdiv(g.M, 64), 1), dim3(512), 0, st, g);
    else {
        // 2 x 2 blocks of 16 x 16 per workgroup once there are two row blocks (stacked batches) -- the blocks' bits do not depend on the tile
        static const int tile2 = getenv_int("CCVS_GEMM_TILE2", 1);   // 0: one block per workgroup (rounds 3-4)
        const bool t2 = tile2 && g.M > 32 && g.N >= 32;   // from three row blocks on: with two, N / 32 workgroups are too few alone (2.03 against 1.53 ms per step)
#define GEMM16_LAUNCH(WNTv, RBv, CBv, Uv)                                                                                                         \
    hipLaunchKernelGGL((gemm16_kernel<WNTv, RBv, CBv, Uv>), dim3(cdiv(g.N, 16 * CBv), cdiv(g.M, 16 * RBv), g.kz), dim3(64 * GEMM_WAVES), 0, st, g.x, \
                       g.w, g.ldx, g.K, g.N, g.M, g.ks, g.kz, g)
        if (decode_nt() & 2) { if (t2) GEMM16_LAUNCH(2, 2, 2, 1); else GEMM16_LAUNCH(2, 1, 1, GEMM_U); }
        else { if (t2) GEMM16_LAUNCH(0, 2, 2, 1); else GEMM16_LAUNCH(0, 1, 1, GEMM_U); }
#undef GEMM16_LAUNCH
    }
    CCVS_CHECK_LAUNCH(name);
    return CCVS_OK;
}

// Workspace layout: [GEMM_WS_TILES x kz<=4 x 256 floats of split-K slabs][GEMM_WS_TILES arrival counters][StepBar: the grid barrier of
// the persistent decode step, below] -- zeroed once by the caller, left consistent by every launch.
#define GEMM_WS_SLAB_BYTES ((size_t)GEMM_WS_TILES * 4 * 256 * sizeof(float))
#define GEMM_WS_BAR_OFFSET (GEMM_WS_SLAB_BYTES + (size_t)GEMM_WS_TILES * sizeof(int))   // 4 MB + 4 KB: 128-byte aligned
#define GEMM_WS_BAR_BYTES 2048
extern "C" int64_t ccvs_gemm_workspace_bytes(void) { return (int64_t)(GEMM_WS_BAR_OFFSET + GEMM_WS_BAR_BYTES); }

extern "C" int ccvs_gemm_nt(const float* x, int64_t ldx, const float* w, const float* bias, const float* res, float* y, int64_t ldy,
                            int32_t M, int32_t N, int32_t K, int32_t epilogue, void* workspace, void* stream) {
    Gemm16 g = {};
    g.x = x; g.ldx = ldx; g.w = w; g.bias = bias; g.res = res; g.y = y; g.ldy = ldy; g.M = M; g.N = N; g.K = K;
    g.epi = epilogue & 0xff; g.seq = (epilogue & CCVS_GEMM_SEQ) ? 1 : 0;
    if (workspace) {
        g.ws_slabs = (float*)workspace;
        g.ws_count = (int*)((char*)workspace + GEMM_WS_SLAB_BYTES);
    }
    return launch_gemm16(g, (hipStream_t)stream, "ccvs_gemm_nt");
}

extern "C" int ccvs_gemm_ln(const float* x, int64_t ldx, const float* w_gamma, const float* bias_beta, const float* w_rowsum, float eps,
                            float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t epilogue, void* stream) {
    CCVS_REQUIRE(w_rowsum, "ccvs_gemm_ln: null pointer");
    CCVS_REQUIRE((epilogue & 0xff) == 0 || (epilogue & 0xff) == 1, "ccvs_gemm_ln: bad epilogue");
    Gemm16 g = {};
    g.x = x; g.ldx = ldx; g.w = w_gamma; g.bias = bias_beta; g.y = y; g.ldy = ldy; g.M = M; g.N = N; g.K = K;
    g.epi = epilogue & 0xff; g.seq = (epilogue & CCVS_GEMM_SEQ) ? 1 : 0;
    g.ln_s = w_rowsum; g.ln_eps = eps;
    return launch_gemm16(g, (hipStream_t)stream, "ccvs_gemm_ln");
}

extern "C" int ccvs_gemm_ln_qkv(const float* x, int64_t ldx, const float* w_gamma, const float* bias_beta, const float* w_rowsum, float eps,
                                float* q, float* kcache, float* vcache, int32_t B, int32_t Tq, int32_t C, int32_t H, int32_t pos0,
                                const int32_t* pos_dev, int32_t Tmax, void* stream) {
    CCVS_REQUIRE(w_rowsum && kcache && vcache, "ccvs_gemm_ln_qkv: null pointer");
    CCVS_REQUIRE(B > 0 && Tq > 0 && H > 0 && C % H == 0 && pos0 >= 0 && pos0 + Tq <= Tmax, "ccvs_gemm_ln_qkv: bad shape / positions");
    Gemm16 g = {};
    g.x = x; g.ldx = ldx; g.w = w_gamma; g.bias = bias_beta; g.y = q; g.ldy = C; g.M = B * Tq; g.N = 3 * C; g.K = C; g.epi = 0;
    g.ln_s = w_rowsum; g.ln_eps = eps;
    g.kcache = kcache; g.vcache = vcache; g.C = C; g.H = H; g.D = C / H; g.Tq = Tq; g.Tmax = Tmax; g.pos0 = pos0; g.pos_dev = pos_dev;
    g.seq = Tq > 1 ? 1 : 0;
    return launch_gemm16(g, (hipStream_t)stream, "ccvs_gemm_ln_qkv");
}

// ---------------------------------------------------------------------------------------
// KV cache append and causal attention over the cache (head dim 16 / 32 / 64).
// Positions may come from a device-resident int32 (`pos_dev`, added to pos0) so that a decode
// step captured in a hipGraph replays at advancing positions without re-capture.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_append_kernel(const float* __restrict__ k, const float* __restrict__ v, long sB, long ld,
                                                        float* __restrict__ kc, float* __restrict__ vc, long total, int H, int Tq,
                                                        int pos0, const int32_t* __restrict__ pos_dev, int Tmax, int D) {
    if (pos_dev) pos0 += *pos_dev;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d = (int)(i % D);
        long t = i / D;
        const int h = (int)(t % H);
        t /= H;
        const int tq = (int)(t % Tq);
        const long b = t / Tq;
        if (pos0 + tq >= Tmax) continue;
        const long src = b * sB + tq * ld + h * D + d;
        const long dst = ((b * H + h) * Tmax + pos0 + tq) * D + d;
        kc[dst] = k[src];
        vc[dst] = v[src];
    }
}

extern "C" int ccvs_kv_append(const float* k, const float* v, int64_t sB, int64_t ld, float* kcache, float* vcache, int32_t B, int32_t H,
                              int32_t Tq, int32_t pos0, const int32_t* pos_dev, int32_t Tmax, int32_t D, void* stream) {
    CCVS_REQUIRE(k && v && kcache && vcache, "ccvs_kv_append: null pointer");
    CCVS_REQUIRE(B > 0 && H > 0 && Tq > 0 && D > 0 && pos0 >= 0 && pos0 + Tq <= Tmax, "ccvs_kv_append: positions %d..%d exceed cache %d",
                 pos0, pos0 + Tq, Tmax);
    const long total = (long)B * Tq * H * D;
    hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, k, v, (long)sB, (long)ld, kcache,
                       vcache, total, H, Tq, pos0, pos_dev, Tmax, D);
    CCVS_CHECK_LAUNCH("ccvs_kv_append");
    return CCVS_OK;
}

// Prefill form (Tq > 1): flash-style attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), exact fp32 products.
//
// A workgroup = 4 waves = 4 blocks of 32 queries of one (batch, head); the K / V rows of the cache are walked in tiles of
// 32 keys staged ONCE per workgroup in LDS (K transposed to [d][key], V as [key][d]) and shared by its 128 queries.
// Per wave and key tile:
//   S^T = K . Q^T   (A = K tile from LDS, B = the wave's Q block held in registers, pre-scaled by 1/sqrt(D)): the 32x32
//         accumulator layout gives every lane ONE query (column lane%32) and 16 of the 32 keys, so the causal mask, the
//         running max / sum of the online softmax and the exponentials are per-lane register work plus one xor-32 shuffle;
//   O^T += V^T . P^T   the probabilities are used as the B operand straight from the S accumulator registers: MFMA step
//         j contracts over the key that accumulator register j holds in each lane half (the contraction order over keys
//         is free), and the A operand reads the matching V row from LDS -- P never moves;
//   O is rescaled by exp(m_old - m_new) when the running max moves.
// Query t (position pos0 + t) sees keys 0 .. pos0 + t; tiles beyond a wave's last query are skipped by that wave.  Two
// tile buffers: tile i+1 is fetched into registers before tile i is consumed and stored after it, one barrier per tile.
// The result goes through LDS so that it is written with lanes along the head dimension (coalesced rows of `out`).
// Replaces the softmax(QK^T / sqrt(d) masked) V of mingpt.py:67-77 for whole sequences (prefill, teacher-forced forward,
// the re-prefill of a slid token window).
template <int D>
__global__ __launch_bounds__(256) void attention_prefill_kernel(const float* __restrict__ q, long q_sB, long ldq, const float* __restrict__ kc,
                                                                const float* __restrict__ vc, float* __restrict__ out, int H, int Tq, int pos0,
                                                                const int32_t* __restrict__ pos_dev, int Tmax, float scale) {
    constexpr int MT = (D + 31) / 32;   // 32-row tiles of the head dimension in O^T
    constexpr int DP = MT * 32;         // head dimension padded to the tile (zero columns of V)
    constexpr int DK = D / 2;           // MFMA steps of S (K = 2 per step)
    constexpr int KS = 33;              // row stride of the transposed K tile
    constexpr int F4 = D / 4;           // float4 per cache row
    constexpr int NLD = (32 * F4 + 255) / 256;  // float4 per thread and tile
    constexpr int TILE_WORDS = D * KS + 32 * DP;
    constexpr int OUT_WORDS = 128 * (D + 1);
    constexpr int SMEM_WORDS = 2 * TILE_WORDS > OUT_WORDS ? 2 * TILE_WORDS : OUT_WORDS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    // grid = (batch x head, query block): blocks are dispatched x-fastest, so ALL (batch, head) pairs of the last query block
    // -- the one with the most visible keys under the causal mask -- start first and the short blocks fill the tail
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    if (pos_dev) pos0 += *pos_dev;
    const int q0 = (gridDim.y - 1 - blockIdx.y) * 128;     // first query of the workgroup
    const int qw = q0 + wave * 32;                         // first query of the wave
    const int qidx = qw + qi;                              // this lane's query
    const int q_last_wg = min(q0 + 127, Tq - 1), q_last_w = min(qw + 31, Tq - 1);
    const int ntiles = (min(pos0 + q_last_wg + 1, Tmax) + 31) >> 5;
    const float* kbase = kc + (long)bh * Tmax * D;
    const float* vbase = vc + (long)bh * Tmax * D;

    // Q block of the wave: lane holds Q[qidx][2i + half], i = 0 .. DK-1 (B operand of step i)
    float qreg[DK];
    {
        const float* qp = q + (long)b * q_sB + (long)min(qidx, Tq - 1) * ldq + h * D + half;
#pragma unroll
        // scores are kept in the log2 domain (scale * log2(e) folded into Q): the softmax then needs v_exp_f32 only
        for (int i = 0; i < DK; ++i) qreg[i] = (qidx < Tq) ? qp[2 * i] * (scale * 1.44269504088896340736f) : 0.f;
    }
    if (MT * 32 > D) {  // zero the padding columns of both V buffers once (D = 16)
        for (int e = tid; e < 2 * 32 * (DP - D); e += 256) {
            const int bsel = e / (32 * (DP - D)), r = e - bsel * 32 * (DP - D);
            smem[bsel * TILE_WORDS + D * KS + (r / (DP - D)) * DP + D + r % (DP - D)] = 0.f;
        }
    }

    float4 kreg[NLD], vreg[NLD];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = min(tid + 256 * r, 32 * F4 - 1);
            const int key = e / F4, d4 = e - key * F4;
            // cache rows >= pos0 + Tq have never been written: their scores are masked (select) but a 0 x NaN in the PV
            // product would still poison the sum, so such rows are staged as zeros (and never read: the address is clamped)
            const bool written = kt * 32 + key < pos0 + Tq;
            const long row = min(kt * 32 + key, pos0 + Tq - 1);
            const float4 kv = *reinterpret_cast<const float4*>(kbase + row * D + 4 * d4);
            const float4 vv = *reinterpret_cast<const float4*>(vbase + row * D + 4 * d4);
            kreg[r] = written ? kv : make_float4(0.f, 0.f, 0.f, 0.f);
            vreg[r] = written ? vv : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {
        float* kt_s = smem + buf * TILE_WORDS;
        float* v_s = kt_s + D * KS;
#pragma unroll
        for (int r = 0; r < NLD; ++r) {
            const int e = tid + 256 * r;
            if (e < 32 * F4) {
                const int key = e / F4, d4 = e - key * F4;
                kt_s[(4 * d4 + 0) * KS + key] = kreg[r].x;
                kt_s[(4 * d4 + 1) * KS + key] = kreg[r].y;
                kt_s[(4 * d4 + 2) * KS + key] = kreg[r].z;
                kt_s[(4 * d4 + 3) * KS + key] = kreg[r].w;
                *reinterpret_cast<float4*>(v_s + key * DP + 4 * d4) = vreg[r];
            }
        }
    };

    f32x16 acc_o[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[mt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    fetch(0);
    __syncthreads();   // padding columns zeroed
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) fetch(kt + 1);
        const int key0 = kt * 32;
        if (key0 <= pos0 + q_last_w && qw < Tq) {   // wave-uniform: this tile holds keys visible to the wave
            const float* kt_s = smem + buf * TILE_WORDS;
            const float* v_s = kt_s + D * KS;
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int i = 0; i < DK; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kt_s[(2 * i + half) * KS + qi], qreg[i], s, 0, 0, 0);
            // s[r] = S[query qidx][key key0 + 8*(r/4) + 4*half + r%4]
            const int lim = pos0 + qidx - key0;   // keys with index <= lim are visible
            float m_t = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = 8 * (r >> 2) + 4 * half + (r & 3);
                s[r] = (kk <= lim && qidx < Tq) ? s[r] : -INFINITY;
                m_t = fmaxf(m_t, s[r]);
            }
            m_t = fmaxf(m_t, __shfl_xor(m_t, 32, 64));
            const float m_new = fmaxf(m_run, m_t);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;   // nothing visible yet: every p below is 2^-inf = 0
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
            float l_t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
                l_t += s[r];
            }
            l_t += __shfl_xor(l_t, 32, 64);
            l_run = l_run * alpha + l_t;
            m_run = m_new;
            const bool rescale = __any(alpha != 1.f);   // wave-uniform: the running max of no lane moved -> O keeps its scale
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc_o[mt][r] *= alpha;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int kk = 8 * (j >> 2) + 4 * half + (j & 3);
                    acc_o[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v_s[kk * DP + mt * 32 + qi], s[j], acc_o[mt], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < ntiles) stash(buf ^ 1);   // the other buffer: its last readers passed the barrier of the previous tile
        __syncthreads();
    }
    // O[d][query] / l  ->  LDS [query][d]  ->  out rows (lanes along d)
    float* o_s = smem;
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = mt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
            if (d < D) o_s[(wave * 32 + qi) * (D + 1) + d] = acc_o[mt][r] * inv;
        }
    __syncthreads();
    for (int e = tid; e < 128 * D; e += 256) {
        const int ql = e / D, d = e - ql * D;
        if (q0 + ql < Tq) out[((long)b * Tq + q0 + ql) * (H * D) + h * D + d] = o_s[ql * (D + 1) + d];
    }
}

// 16 bytes, optionally with the non-temporal policy (a stream read once: keys / values of a decode step)
template <bool NT>
__device__ __forceinline__ f32x4 ld_f4(const float* p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    return NT ? __builtin_nontemporal_load(q) : *q;
}

// Decode form (Tq = 1): one workgroup per (batch, head) streams that head's K and V rows once.
// A key row of D floats is read by D/4 consecutive lanes as float4 (a wave-instruction covers
// 64/(D/4) whole rows: 1 KiB, fully coalesced); the partial dots are summed across those lanes
// with xor shuffles.  PV uses the same lane map (lane owns 4 head dims of one key slot); the
// key slots of a wave and the 4 waves are reduced through LDS in a fixed order.
// Footprint (see gemm16_kernel): 256 threads, <= 48 VGPRs, LDS = scores + 4 KB -- the workgroup is dispatched next to a
// convolution workgroup of the frame decoder instead of waiting for one to retire; a stream of this shape keeps 3.5 TB/s
// beside the decoder (5.5 alone; tools/chain_probe.py).  Four key rows per lane are requested together (16 KB in flight
// per workgroup); the first V batch is requested before the softmax reduction starts.
// (bh = the (batch row, head) pair; smem = 16 + 1024 + Tmax floats.  COH: the persistent decode step -- q, the cache row of the
//  CURRENT position (written by the QKV phase of the same launch) and the output are handed between workgroups of one launch:
//  sc1 accesses; the older cache rows were written by earlier launches and stay on the plain / non-temporal stream.)
template <int D, bool NT, bool COH>
__device__ __forceinline__ void attention_decode_item(const float* __restrict__ q, long q_sB, const float* __restrict__ kc,
                                                      const float* __restrict__ vc, float* __restrict__ out, int H, int pos0,
                                                      const int32_t* __restrict__ pos_dev, int grp_rows, int Tmax, float scale, int bh,
                                                      float* smem) {
    constexpr int LPK = D / 4;     // lanes per key row
    constexpr int KPI = 64 / LPK;  // key rows per wave-instruction
    constexpr int NW = 4;
    constexpr int AU = 4;          // key rows per lane whose loads are issued together
    constexpr int BATCH = NW * KPI * AU;
    float* red = smem;                    // [16]
    float* pv = smem + 16;                // [NW][64][4]
    float* ps = smem + 16 + NW * 256;     // [Tmax]
    const int tid = thread_id<COH>(), lane = tid & 63, wave = tid >> 6;
    const int b = bh / H, h = bh - b * H;
    if (pos_dev) {   // row groups of a decode step: one cache length per group
        // (COH: the length words are constant for the whole launch until the pick's last row moves them on, after every other reader:
        //  a SCALAR load through the constant address space -- as a vector load of a uniform address the value, and with it the
        //  loop bounds and row addresses, is per-lane data to the compiler)
        if constexpr (COH) pos0 += ((const __attribute__((address_space(4))) int32_t*)pos_dev)[grp_rows > 0 ? b / grp_rows : 0];
        else pos0 += pos_dev[grp_rows > 0 ? b / grp_rows : 0];
    }
    const int L = min(pos0 + 1, Tmax);
    const int kk = lane / LPK, d4 = lane - kk * LPK;
    const float* kbase = kc + (long)bh * Tmax * D + 4 * d4;
    const float* vbase = vc + (long)bh * Tmax * D + 4 * d4;
    const int jw = wave * KPI + kk;  // this lane's row within a batch, + u * NW * KPI
    const int nbatch = (L + BATCH - 1) / BATCH;

    // unconditional loads from clamped rows (predicated loads would serialise)
    // COH: rows >= L - 1 (the position this step appends, and the clamped tail of the last batch) take the row fetched by ONE sc1
    // load instead of whatever the streaming load of that address returned -- same values, same arithmetic, no stale line
    f32x4 k_new = {0.f, 0.f, 0.f, 0.f}, v_new = {0.f, 0.f, 0.f, 0.f};
    if constexpr (COH) {
        k_new = ld_act4_sc1(kbase + (long)(L - 1) * D);
        v_new = ld_act4_sc1(vbase + (long)(L - 1) * D);
    }
#define ATT_LOAD(dst, base, bi, fresh)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < AU; ++u) {                                                                     \
        dst[u] = ld_f4<NT>(base + (long)min((bi) * BATCH + jw + u * NW * KPI, L - 1) * D);                                \
        if constexpr (COH) { if ((bi) * BATCH + jw + u * NW * KPI >= L - 1) dst[u] = fresh; }                            \
    }

    float4 qv;
    if constexpr (COH) {
        const f32x4 q4 = ld_act4_sc1(q + (long)b * q_sB + h * D + 4 * d4);
        qv = make_float4(q4[0], q4[1], q4[2], q4[3]);
    } else {
        qv = *reinterpret_cast<const float4*>(q + (long)b * q_sB + h * D + 4 * d4);
    }
    float lmax = -INFINITY;
#pragma unroll 1   // (one batch of AU rows in flight per lane: the 48-register budget; hipcc unrolls this loop by two once the body is an inlined function)
    for (int bi = 0; bi < nbatch; ++bi) {
        f32x4 kv[AU];
        ATT_LOAD(kv, kbase, bi, k_new);
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int j = bi * BATCH + jw + u * NW * KPI;
            float s = kv[u][0] * qv.x + kv[u][1] * qv.y + kv[u][2] * qv.z + kv[u][3] * qv.w;
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
            s *= scale;
            if (j < L) {
                if (d4 == 0) ps[j] = s;
                lmax = fmaxf(lmax, s);
            }
        }
    }
    f32x4 vv[AU];
    ATT_LOAD(vv, vbase, 0, v_new);  // in flight during the softmax reductions

    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    float gmax = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) gmax = fmaxf(gmax, red[w]);
    float lsum = 0.f;
    for (int j = tid; j < L; j += NW * 64) {
        const float e = expf(ps[j] - gmax);
        ps[j] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    float tot = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += red[w];

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int bi = 0; bi < nbatch; ++bi) {
        if (bi > 0) ATT_LOAD(vv, vbase, bi, v_new);
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int j = bi * BATCH + jw + u * NW * KPI;
            const float p = (j < L) ? ps[min(j, L - 1)] : 0.f;
            acc.x += p * vv[u][0]; acc.y += p * vv[u][1]; acc.z += p * vv[u][2]; acc.w += p * vv[u][3];
        }
    }
#undef ATT_LOAD
    *reinterpret_cast<float4*>(pv + (wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (tid < D) {
        const int dd4 = tid >> 2, comp = tid & 3;
        float o = 0.f;
        for (int w = 0; w < NW; ++w)
            for (int s2 = 0; s2 < KPI; ++s2) o += pv[(w * 64 + s2 * LPK + dd4) * 4 + comp];
        st_act<COH>(out + (long)b * (H * D) + h * D + tid, o / tot);
    }
}

template <int D, bool NT>
__global__ __launch_bounds__(256) void attention_decode_kernel(const float* __restrict__ q, long q_sB, const float* __restrict__ kc,
                                                               const float* __restrict__ vc, float* __restrict__ out, int H, int pos0,
                                                               const int32_t* __restrict__ pos_dev, int grp_rows, int Tmax, float scale) {
    TOKEN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    attention_decode_item<D, NT, false>(q, q_sB, kc, vc, out, H, pos0, pos_dev, grp_rows, Tmax, scale, blockIdx.x, smem);
}

#define ATT_DECODE_LAUNCH(Dv, grid_, smem_, ...)                                                                       \
    do {                                                                                                               \
        if (decode_nt() & 1) hipLaunchKernelGGL((attention_decode_kernel<Dv, true>), grid_, dim3(256), smem_, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((attention_decode_kernel<Dv, false>), grid_, dim3(256), smem_, st, __VA_ARGS__);     \
    } while (0)

extern "C" int ccvs_attention(const float* q, int64_t q_sB, int64_t ldq, const float* kcache, const float* vcache, float* out, int32_t B,
                              int32_t H, int32_t Tq, int32_t pos0, const int32_t* pos_dev, int32_t Tmax, int32_t D, void* stream) {
    CCVS_REQUIRE(q && kcache && vcache && out, "ccvs_attention: null pointer");
    CCVS_REQUIRE(D == 64 || D == 32 || D == 16, "ccvs_attention: head dim %d unsupported (16, 32, 64)", D);
    CCVS_REQUIRE(B > 0 && H > 0 && Tq > 0 && pos0 >= 0 && pos0 + Tq <= Tmax, "ccvs_attention: bad positions");
    CCVS_REQUIRE(cdiv(Tq, 128) <= 65535, "ccvs_attention: %d queries too many for one launch", Tq);
    // with a device-side position the visible length is unknown to the host: size LDS for the whole cache
    const int maxL = pos_dev ? Tmax : pos0 + Tq;
    const float scale = 1.0f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    if (Tq == 1) {
        const size_t smem = (size_t)(16 + 4 * 256 + maxL) * sizeof(float);
        CCVS_REQUIRE(smem <= 64 * 1024, "ccvs_attention: sequence too long for the LDS score buffer");
        const dim3 grid((unsigned)(B * H));
        if (D == 64) ATT_DECODE_LAUNCH(64, grid, smem, q, (long)q_sB, kcache, vcache, out, H, pos0, pos_dev, 0, Tmax, scale);
        else if (D == 32) ATT_DECODE_LAUNCH(32, grid, smem, q, (long)q_sB, kcache, vcache, out, H, pos0, pos_dev, 0, Tmax, scale);
        else ATT_DECODE_LAUNCH(16, grid, smem, q, (long)q_sB, kcache, vcache, out, H, pos0, pos_dev, 0, Tmax, scale);
    } else {
        const dim3 grid((unsigned)(B * H), (unsigned)cdiv(Tq, 128));
        if (D == 64) hipLaunchKernelGGL((attention_prefill_kernel<64>), grid, dim3(256), 0, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, pos_dev, Tmax, scale);
        else if (D == 32) hipLaunchKernelGGL((attention_prefill_kernel<32>), grid, dim3(256), 0, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, pos_dev, Tmax, scale);
        else hipLaunchKernelGGL((attention_prefill_kernel<16>), grid, dim3(256), 0, st, q, (long)q_sB, (long)ldq, kcache, vcache, out, H, Tq, pos0, pos_dev, Tmax, scale);
    }
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

// Philox4x32-10 (Salmon et al., SC'11): counter-based, so every (row, element, step) gets its own draw
// with no generator state to advance between hipGraph replays.
__device__ __forceinline__ unsigned philox_first(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

struct Advance {       // decode-step bookkeeping folded into the sampling kernel (all null outside ccvs_gpt_decode_step)
    int64_t* codes;    // [B][codes_sB] generated sequence: codes[b][*widx] = picked token
    long codes_sB;
    int32_t* widx;     // device-resident write index, +1 per step
    int32_t* len;      // device-resident cache length, +1 per step
    int rng;           // draw the Exp(1) noise here: Philox keyed by state[4..5], counter (element, state[6] + row, step = state[0], call = state[3])
    unsigned imm[5];   // rng == 2: immediate Philox words (key0, key1, global row of row 0, step, call) instead of `state`
    int* state;        // int32[8]: [0] completed steps, [2] arrival ticket of this kernel's rows, [3] call index, [4..5] Philox key, [6] global index of row 0
    int grp_rows;      // > 0: rows [g * grp_rows, (g+1) * grp_rows) form group g with its own widx[g], len[g] and state[8g .. 8g+7]
    const float* const* noise_stream;   // host-drawn noise as a whole stream: entry g -> [steps][rows of group g][V]; the block read is state[0] of the group
};

// k-th largest key of xs[0..V) by a 4-pass radix-256 descent (LDS histogram + suffix scan per pass).
__device__ __forceinline__ unsigned kth_largest_key(const float* xs, int V, int top_k, int* hist, int* wtot, int* sel, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    unsigned prefix = 0u, mask = 0u;
    int need = top_k;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (int j = tid; j < V; j += 256) {
            const unsigned key = fkey(xs[j]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        const int h = hist[tid];
        int s = h;  // suffix sum over bins >= tid
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_down(s, o, 64);
            if (lane + o < 64) s += t;
        }
        if (lane == 0) wtot[wave] = s;
        __syncthreads();
        for (int w = wave + 1; w < 4; ++w) s += wtot[w];
        if (s >= need && s - h < need) { sel[0] = tid; sel[1] = need - (s - h); }
        __syncthreads();
        prefix |= (unsigned)sel[0] << shift;
        mask |= 255u << shift;
        need = sel[1];
    }
    return prefix;
}

// (b = the row, B_all = the rows of the launch; smem = PICK_SMEM_WORDS(V) floats.  COH: the logits were written by the head GEMM phase
//  of the same launch -- sc1 loads.)
template <bool COH>
__device__ __forceinline__ void sample_topk_row(const float* __restrict__ logits, long ld, const float* __restrict__ noise,
                                                int64_t* __restrict__ out, long out_stride, int V, int top_k, float temperature,
                                                Advance adv, int b, int B_all, float* smem) {
    float* xs = smem;                  // [V]
    float* redf = smem + V;            // [4]
    int* redj = (int*)(smem + V + 4);  // [4]
    int* wtot = (int*)(smem + V + 8);  // [4]
    int* sel = (int*)(smem + V + 12);  // [2]
    int* hist = (int*)(smem + V + 16); // [256]
    const int tid = thread_id<COH>(), lane = tid & 63, wave = tid >> 6;
    // row group of this row: its own counters / Philox words; `brow` = the row's index inside the group
    const int grp = adv.grp_rows > 0 ? b / adv.grp_rows : 0;
    const int brow = b - grp * (adv.grp_rows > 0 ? adv.grp_rows : 0);
    const int nrows = adv.grp_rows > 0 ? min(adv.grp_rows, B_all - grp * adv.grp_rows) : B_all;
    if (adv.grp_rows > 0) { adv.state += 8 * grp; adv.widx += grp; adv.len += grp; }
    const float* lr = logits + (long)b * ld;
    float lmax = -INFINITY;
    for (int j = tid; j < V; j += 256) {
        const float v = ld_act<COH>(lr + j) / temperature;
        xs[j] = v;
        lmax = fmaxf(lmax, v);
    }
    lmax = wave_max(lmax);
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));

    unsigned thr = 0u;  // key of the k-th largest value; 0 keeps everything
    if (top_k > 0 && top_k < V) thr = kth_largest_key(xs, V, top_k, hist, wtot, sel, tid);
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
    // counter = (element, GLOBAL row = state[6] + b, step, call = state[3]): a clip's draws do not depend on which rank /
    // batch slot it runs in, nor on how many other clips share the launch
    unsigned k0 = 0u, k1 = 0u, step = 0u, row0 = 0u, call = 0u;
    if (adv.rng == 2) {
        k0 = adv.imm[0]; k1 = adv.imm[1]; row0 = adv.imm[2]; step = adv.imm[3]; call = adv.imm[4];
    } else if (adv.rng) {
        k0 = (unsigned)adv.state[4]; k1 = (unsigned)adv.state[5]; step = (unsigned)adv.state[0];
        call = (unsigned)adv.state[3]; row0 = (unsigned)adv.state[6];
    }
    // host-drawn noise: the [B, V] block of an eager pick, or this row's slice of block `state[0]` (the steps its group has
    // completed) of the group's pre-drawn stream -- the captured step then consumes the reference's generator stream in order
    const float* nrow = noise ? noise + (long)b * V : nullptr;
    if (adv.noise_stream) nrow = adv.noise_stream[grp] + ((long)adv.state[0] * nrows + brow) * V;
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int j = tid; j < V; j += 256) {
        float p = xs[j] / tot;
        if (nrow) p = p / nrow[j];
        else if (adv.rng) {
            const float u = ((float)philox_first((unsigned)j, row0 + (unsigned)brow, step, call, k0, k1) + 0.5f) * 2.3283064365386963e-10f;  // (0, 1]
            p = p / fmaxf(-logf(u), 1e-30f);
        }
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
        if (adv.codes) {
            adv.codes[(long)b * adv.codes_sB + *adv.widx] = bi;
            // the last row to finish advances the device-resident counters (every row has read them by then)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const int ticket = __hip_atomic_fetch_add(adv.state + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == nrows - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                *adv.widx += 1;
                *adv.len += 1;
                adv.state[0] += 1;
                __hip_atomic_store(adv.state + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(256) void sample_topk_kernel(const float* __restrict__ logits, long ld, const float* __restrict__ noise,
                                                          int64_t* __restrict__ out, long out_stride, int V, int top_k, float temperature,
                                                          Advance adv) {
    TOKEN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    sample_topk_row<false>(logits, ld, noise, out, out_stride, V, top_k, temperature, adv, blockIdx.x, gridDim.x, smem);
}

#define PICK_SMEM_WORDS(V) ((V) + 16 + 256)

extern "C" int ccvs_sample_topk(const float* logits, int64_t ld, const float* noise, int64_t* out, int64_t out_stride, int32_t B, int32_t V,
                                int32_t top_k, float temperature, void* stream) {
    CCVS_REQUIRE(logits && out, "ccvs_sample_topk: null pointer");
    CCVS_REQUIRE(B > 0 && V > 0 && temperature > 0.f, "ccvs_sample_topk: bad arguments");
    const size_t smem = (size_t)PICK_SMEM_WORDS(V) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sample_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    CCVS_REQUIRE(smem <= 160 * 1024, "ccvs_sample_topk: vocabulary %d too large", V);
    hipLaunchKernelGGL(sample_topk_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, logits, (long)ld, noise, out, (long)out_stride, V,
                       top_k, temperature, Advance{});
    CCVS_CHECK_LAUNCH("ccvs_sample_topk");
    return CCVS_OK;
}

extern "C" int ccvs_sample_topk_philox(const float* logits, int64_t ld, int64_t* out, int64_t out_stride, int32_t B, int32_t V, int32_t top_k,
                                       float temperature, uint32_t key0, uint32_t key1, uint32_t row0, uint32_t step, uint32_t call,
                                       void* stream) {
    CCVS_REQUIRE(logits && out, "ccvs_sample_topk_philox: null pointer");
    CCVS_REQUIRE(B > 0 && V > 0 && temperature > 0.f, "ccvs_sample_topk_philox: bad arguments");
    const size_t smem = (size_t)PICK_SMEM_WORDS(V) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sample_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    CCVS_REQUIRE(smem <= 160 * 1024, "ccvs_sample_topk_philox: vocabulary %d too large", V);
    Advance adv = {};
    adv.rng = 2;
    adv.imm[0] = key0; adv.imm[1] = key1; adv.imm[2] = row0; adv.imm[3] = step; adv.imm[4] = call;
    hipLaunchKernelGGL(sample_topk_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, logits, (long)ld, (const float*)nullptr, out,
                       (long)out_stride, V, top_k, temperature, adv);
    CCVS_CHECK_LAUNCH("ccvs_sample_topk_philox");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// get_icode with n picks per row (the proposals of beam search, transformer_model.py:358-391,395-409): temperature, top-k
// mask, softmax, then the n best of p (sample = False: torch.topk) or of p / Exp(1) noise (sample = True: what
// torch.multinomial(p, n) without replacement computes), best first, ties to the lowest index; also log p of every pick.
// One workgroup per row: probabilities and scores in LDS, n rounds of a block-wide argmax.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sample_topn_kernel(const float* __restrict__ logits, long ld, const float* __restrict__ noise,
                                                          int64_t* __restrict__ out_idx, float* __restrict__ out_logp, int V, int top_k,
                                                          float temperature, int n) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // [V] scaled logits, then probabilities
    float* sc = smem + V;                   // [V] scores the picks are taken from
    float* redf = smem + 2 * V;             // [4]
    int* redj = (int*)(smem + 2 * V + 4);   // [4]
    int* wtot = (int*)(smem + 2 * V + 8);   // [4]
    int* sel = (int*)(smem + 2 * V + 12);   // [2]
    int* hist = (int*)(smem + 2 * V + 16);  // [256]
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
    if (lane == 0) redf[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    unsigned thr = 0u;
    if (top_k > 0 && top_k < V) thr = kth_largest_key(xs, V, top_k, hist, wtot, sel, tid);
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
    for (int j = tid; j < V; j += 256) {
        const float p = xs[j] / tot;
        xs[j] = p;
        sc[j] = noise ? p / noise[(long)b * V + j] : p;
    }
    __syncthreads();
    for (int r = 0; r < n; ++r) {
        float best = -1.f;
        int bi = 0x7fffffff;
        for (int j = tid; j < V; j += 256) {
            const float v = sc[j];
            if (v > best) { best = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { redf[wave] = best; redj[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (redf[w] > best || (redf[w] == best && redj[w] < bi)) { best = redf[w]; bi = redj[w]; }
            bi = min(bi, V - 1);
            out_idx[(long)b * n + r] = bi;
            out_logp[(long)b * n + r] = logf(xs[bi]);
            sc[bi] = -2.f;   // taken
        }
        __syncthreads();
    }
}

extern "C" int ccvs_sample_topn(const float* logits, int64_t ld, const float* noise, int64_t* out_idx, float* out_logp, int32_t B, int32_t V,
                                int32_t top_k, float temperature, int32_t n, void* stream) {
    CCVS_REQUIRE(logits && out_idx && out_logp, "ccvs_sample_topn: null pointer");
    CCVS_REQUIRE(B > 0 && V > 0 && temperature > 0.f && n >= 1 && n <= V, "ccvs_sample_topn: bad arguments");
    const size_t smem = (size_t)(2 * V + 16 + 256) * sizeof(float);
    CCVS_REQUIRE(smem <= 160 * 1024, "ccvs_sample_topn: vocabulary %d too large", V);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sample_topn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(sample_topn_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, logits, (long)ld, noise, out_idx, out_logp, V, top_k,
                       temperature, n);
    CCVS_CHECK_LAUNCH("ccvs_sample_topn");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// The decode step as ONE launch (ccvs_gpt_decode.persistent): 256-thread workgroups that stay resident for
// the whole step -- one per CU -- and walk its phases themselves,
//     embed | per layer: ln1+QKV+cache scatter | attention | proj+res | ln2+fc+GELU | fc2+res | ln_f+head | pick,
// instead of 5 n_layer + 3 dependent launches.  A phase is the launch it replaces, tile for tile: workgroup w takes the tiles
// w, w + G, ... of the phase's grid and runs the SAME tile body (gemm16_tile / attention_decode_item / sample_topk_row), so a row's
// arithmetic -- K slices per wave, MFMA order, slab order, reduction order -- is the launch chain's instruction for instruction and
// the tokens are bit-identical (tests/test_persistent_step_gpu.py).  What changes is how a phase's output reaches the next phase's
// readers on OTHER CUs (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": per-XCD L2s are not
// coherent, a CU's L1 is never refreshed by another CU's stores):
//   * every buffer handed over inside the launch -- x, q, the cache row of the current position, att, h, logits, the split-K slabs --
//     is written with sc1 (write-through) stores and read with sc1 loads (L1 bypassed): COH = true in the tile bodies; weights, the
//     older cache rows, the embedding tables and the device-resident counters were written by EARLIER launches and stay plain;
//   * the grid barrier between two phases: every wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets (s_barrier), ONE
//     lane adds 1 to the workgroup's shard of a monotonic arrival counter (8 shards on lines of their own: 32 agent-scope adds per
//     word instead of 256), wave 0 polls the 8 shards with sc1 loads + s_sleep until their sum reaches base + phases passed x G,
//     the workgroup meets again and goes on -- no fence instruction anywhere (the "Valid forms" table, first row: one lane of each
//     storing workgroup signals for all of that workgroup's stores after every storing wave's vmcnt(0) and the workgroup's barrier;
//     the poller loads after its poll has matched, the other waves after a barrier it then joins; sc1 stores of 4 / 16 bytes, sc1
//     loads of 4 / 16 bytes).  `base` = the arrivals counted before this launch, kept in a word of its own that workgroup 0 moves
//     forward once per launch (read by the NEXT launch only: a kernel boundary apart), so nothing is zeroed between launches and a
//     captured step replays unchanged; comparisons are on the wrapped difference.  A poll that sees nothing for 5 s gives up, sets
//     StepBar.err and lets the workgroup run on; the word is sticky -- every later barrier of this workspace gives up at once, so a lost
//     workgroup costs ONE time-out, not one per phase -- the tokens since are garbage, and ccvs_gpt_decode_status reports it and
//     starts the barrier block over.
// Residency: G = the number of CUs; a workgroup (256 threads, 64 registers, <= 18 KB of LDS at BAIR size) fits beside the 3 x 3
// convolution workgroups of the frame decoder (two waves of 208 / 214 registers per SIMD) but not beside the 1 x 1 forms' 225, and it
// HOLDS its slot for the whole step: two token chains' persistent steps cannot both be resident beside the decoder on one CU, and
// partially resident grids that wait for each other's slots stall until convolution workgroups retire -- a schedule with ONE token
// chain is what this form is for.  MEASURED (profiles/r06_persistent_step.txt): bit-identical and 1.4-1.9 x slower than the launch
// chain, alone and in the run -- ~7 us per grid barrier against 1.5-2 per kernel boundary, and one workgroup per CU where the chain
// oversubscribes the chip (the attention phases run at ~2 TB/s instead of 5.7).  An opt-in, not the default.
// ---------------------------------------------------------------------------------------
#define STEP_SPIN_TICKS 500000000ULL   // 5 s of the 100 MHz s_memrealtime counter
struct StepBar {
    unsigned shard[8][32];   // arrivals, monotonic; one 128-byte line per shard
    unsigned base[32];       // arrivals counted when the current launch started
    unsigned err[32];        // != 0: a barrier of some launch gave up (phase index + 1)
};
static_assert(sizeof(StepBar) <= GEMM_WS_BAR_BYTES, "StepBar outgrew its slot in the workspace");

// One phase of the step, built once per descriptor by the host (ccvs_gpt_decode_prepare) into a caller-provided device buffer: the
// kernel reads phase ph's operands by scalar loads when it gets there.  (Operands selected inside the kernel from a by-value argument
// block -- 24 layers x 12 pointers -- are loop-invariant to the compiler: it kept them ALL live across the phase loop, ran out of
// scalar registers (106) and parked uniform values in vector registers: 96 + 16 registers per lane instead of the tile bodies' 48.)
#define STEP_GEMM 0
#define STEP_ATTENTION 1
struct StepPhase {
    int kind, pad_;
    Gemm16 g;   // STEP_GEMM: the launch this phase replaces; STEP_ATTENTION: x = q, y = att, kcache / vcache, H, Tmax, M = rows
};

struct StepArgs {
    int B, C, V, vocab, grp_rows, pos_off, n_phases, H;
    const float *tok_emb, *pos_table;
    const int64_t* tok;
    const int32_t* len;
    float *x, *logits;
    StepBar* bar;
    const StepPhase* prog;
    float scale;
    const float* noise;
    int top_k;
    float temperature;
    Advance adv;
};

__device__ __forceinline__ void step_barrier(StepBar* bar, unsigned target, int phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave: its sc1 stores of this phase have left
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&bar->shard[blockIdx.x & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned* sp = &bar->shard[threadIdx.x & 7][0];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 1;; ++spins) {
            unsigned v = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            if ((int)(v - target) >= 0) break;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 255u) == 0) {
                // give up: after 5 s without the others, or AT ONCE when any workgroup of this workspace has given up before (the word is
                // sticky until ccvs_gpt_decode_status clears the whole barrier block) -- a lost workgroup costs one time-out, not one per phase
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > STEP_SPIN_TICKS;
                if (late || __hip_atomic_load(&bar->err[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    if (late && threadIdx.x == 0) __hip_atomic_store(&bar->err[0], (unsigned)(phase + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

template <int RB, int CB, int U>
__device__ __forceinline__ void step_gemm(const __attribute__((address_space(4))) Gemm16& g, float* smem) {
    float* red = smem;
    float* stat = red + GEMM16_RED_WORDS(RB * CB);
    float* fin = stat + GEMM16_STAT_WORDS(RB);
    const int N = g.N, M = g.M, K = g.K, ks = g.ks, kz = g.kz;
    const long ldx = g.ldx;
    const float* x = gp_<true>(g.x);
    const float* w = gp_<true>(g.w);
    const int gx = (N + 16 * CB - 1) / (16 * CB), gy = (M + 16 * RB - 1) / (16 * RB);
    const int total = gx * gy * kz;
    // tile order = the launch's dispatch order (x fastest): the row blocks of a column tile are 96 / 32 / 128 tiles apart, a multiple
    // of 8, so with workgroups dealt to the XCDs round-robin they meet in one L2, as in the launch chain
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        const int bx = vb % gx, r = vb / gx;
        gemm16_tile<0, RB, CB, U, true>(x, w, ldx, K, N, M, ks, kz, g, bx, r % gy, r / gy, red, stat, fin);
        __syncthreads();   // the tile's LDS is the next tile's
    }
}

#define STEP_GEMM_WORDS(RB, CB) (GEMM16_RED_WORDS((RB) * (CB)) + GEMM16_STAT_WORDS(RB) + GEMM16_FIN_WORDS(RB))

// The argument block and the phase table live in device memory (d->program) and are read through CONSTANT-address-space pointers:
//   * as by-value kernel arguments the whole block -- the pick's sampler words included -- was loaded at the entry and stayed live in
//     scalar registers across all 122 phases (106 SGPRs, uniform values parked in vector registers);
//   * through an ordinary pointer the compiler must assume the kernel's own stores may alias the table: it then reads the (uniform)
//     operands with FLAT vector loads into vector registers and every pointer among them is a generic pointer -- flat loads and stores
//     for the tensors too, which s_waitcnt vmcnt does not even cover alone.  address_space(4) says "constant for this launch" (scalar
//     loads), and every pointer read from the table is cast through address_space(1) so that the accesses are global_* / buffer_*.
#define STEP_CONST __attribute__((address_space(4)))
#define STEP_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(STEP_GLOBAL T*)p;
}
// amdgpu_waves_per_eu(8, 8): at most 64 registers per lane (no finer cap exists: amdgpu_num_vgpr is not honoured here).  Left alone,
// hipcc allocates 73 + 16 for this body although no phase needs more than the launch chain's 48: with 256 threads per workgroup it
// sees no reason to be frugal.  64 fit beside two convolution waves of up to 216 registers per SIMD (the 3 x 3 forms that carry
// the decoder's time: 208 / 214), not beside the 1 x 1 forms' 225 -- there the step's workgroup waits for one to retire, once per step.
template <int D, bool T2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void gpt_step_kernel(const StepArgs* __restrict__ ap_) {
    TOKEN_PRIO();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int RB = T2 ? 2 : 1, CB = T2 ? 2 : 1, U = T2 ? 1 : GEMM_U;
    const unsigned G = gridDim.x;
    const STEP_CONST StepArgs* ap = (const STEP_CONST StepArgs*)ap_;
    StepBar* bar = as_global(ap->bar);
    const STEP_CONST StepPhase* prog = (const STEP_CONST StepPhase*)ap->prog;
    const int n_phases = ap->n_phases;
    const float att_scale = ap->scale;
    unsigned target = __hip_atomic_load(&bar->base[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // embedding of the last picked token at row pos_off + *len (gpt_embed_kernel)
        const int B = ap->B, C = ap->C, vocab = ap->vocab, grp_rows = ap->grp_rows, pos_off = ap->pos_off;
        const int64_t* tok = as_global(ap->tok);
        const int32_t* len = as_global(ap->len);
        const float* tok_emb = as_global(ap->tok_emb);
        const float* pos_table = as_global(ap->pos_table);
        float* x = as_global(ap->x);
        const long total = (long)B * C;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)G * 256) {
            const long b = i / C;
            const int c = (int)(i - b * C);
            long t = tok[b];
            t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
            const long prow = pos_off + len[grp_rows > 0 ? b / grp_rows : 0];
            st_act<true>(x + i, tok_emb[t * C + c] + pos_table[prow * C + c]);
        }
    }
    target += G; step_barrier(bar, target, 0);
    // ONE call site per tile body, operands from the phase table (per layer: ln1+QKV | attention | proj | ln2+fc | fc2; then ln_f+head)
#pragma unroll 1
    for (int ph = 0; ph < n_phases; ++ph) {
        const STEP_CONST StepPhase* P = prog + ph;
        if (P->kind == STEP_ATTENTION) {
            const STEP_CONST Gemm16& g = P->g;
            const int n_bh = g.M * g.H, H = g.H, Tmax = g.Tmax, grp_rows = g.grp_rows;
            const long q_sB = g.ldx;
            const float* q = as_global(g.x);
            const float* kc = as_global(g.kcache);
            const float* vc = as_global(g.vcache);
            float* att = as_global(g.y);
            const int32_t* len = as_global(g.pos_dev);
            for (int bh = blockIdx.x; bh < n_bh; bh += G) {   // attention over the cache
                attention_decode_item<D, true, true>(q, q_sB, kc, vc, att, H, 0, len, grp_rows, Tmax, att_scale, bh, smem);
                __syncthreads();
            }
        } else {
            step_gemm<RB, CB, U>(P->g, smem);
        }
        target += G; step_barrier(bar, target, ph + 1);
    }
    // (workgroup 0, past the last barrier: the arrivals counted so far are the next launch's base -- no workgroup of THIS launch reads it again)
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&bar->base[0], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // pick + bookkeeping
        const int B = ap->B, V = ap->V;
        Advance adv;
        adv.codes = as_global(ap->adv.codes); adv.codes_sB = ap->adv.codes_sB; adv.widx = as_global(ap->adv.widx); adv.len = as_global(ap->adv.len);
        adv.rng = ap->adv.rng;
#pragma unroll
        for (int i = 0; i < 5; ++i) adv.imm[i] = 0u;
        adv.state = as_global(ap->adv.state); adv.grp_rows = ap->adv.grp_rows; adv.noise_stream = as_global(ap->adv.noise_stream);
        for (int b = blockIdx.x; b < B; b += G) {
            sample_topk_row<true>(as_global(ap->logits), (long)V, as_global(ap->noise), const_cast<int64_t*>(as_global(ap->tok)), 1L, V, ap->top_k,
                                  ap->temperature, adv, b, B, smem);
            __syncthreads();
        }
    }
}

// 0 when the persistent steps launched with this workspace so far all completed their barriers (the stream is synchronised first)
extern "C" int ccvs_gpt_decode_status(const void* workspace, void* stream) {
    CCVS_REQUIRE(workspace, "ccvs_gpt_decode_status: null pointer");
    unsigned err = 0;
    const StepBar* bar = (const StepBar*)((const char*)workspace + GEMM_WS_BAR_OFFSET);
    hipError_t e = hipMemcpyAsync(&err, &bar->err[0], sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { ccvs_set_error("ccvs_gpt_decode_status: %s", hipGetErrorString(e)); return CCVS_ERR_LAUNCH; }
    if (err) {
        // the arrival counters of the failed launch are inconsistent: start the workspace's barrier block over (every later step would
        // otherwise give up at once on the sticky word), then report
        (void)hipMemsetAsync((void*)bar, 0, sizeof(StepBar), (hipStream_t)stream);
        (void)hipStreamSynchronize((hipStream_t)stream);
        ccvs_set_error("ccvs_gpt_decode_status: a grid barrier of the persistent decode step gave up (phase %u): the tokens of the steps since the last check are invalid", err - 1);
        return CCVS_ERR_LAUNCH;
    }
    return CCVS_OK;
}

// d->program = [StepArgs, padded to STEP_HEAD_BYTES][5 n_layer + 1 phases]
#define STEP_HEAD_BYTES 512
static_assert(sizeof(StepArgs) <= STEP_HEAD_BYTES, "StepArgs outgrew the head of the program buffer");
extern "C" int64_t ccvs_gpt_program_bytes(int32_t n_layer) { return (int64_t)STEP_HEAD_BYTES + (int64_t)(5 * (n_layer > 0 ? n_layer : 0) + 1) * sizeof(StepPhase); }

static int step_check(const ccvs_gpt_decode* d, const char* who) {
    // what the persistent step covers -- what the pipelined generation uses -- and says so otherwise: no silent fall-back
    if (!(d->workspace && d->program)) { ccvs_set_error("%s: the persistent step needs `workspace` (ccvs_gemm_workspace_bytes, zeroed once) and `program` (ccvs_gpt_program_bytes)", who); return CCVS_ERR_ARG; }
    if (d->B > GEMM_DECODE_MAX_M) { ccvs_set_error("%s: the persistent step takes at most %d rows", who, GEMM_DECODE_MAX_M); return CCVS_ERR_ARG; }
    if (!(decode_nt() == 1 && getenv_int("CCVS_GEMM_TILE2", 1) == 1)) { ccvs_set_error("%s: the persistent step is built for the default cache policies and tile (CCVS_DECODE_NT=1, CCVS_GEMM_TILE2=1)", who); return CCVS_ERR_ARG; }
    if (!(d->C >= 32 && d->F >= 32 && d->V >= 32 && d->C % 16 == 0 && d->F % 16 == 0)) { ccvs_set_error("%s: the persistent step needs >= 32 output columns per GEMM and K %% 16 == 0", who); return CCVS_ERR_ARG; }
    if (!((long)d->V * d->C * 4 < (1L << 31) && (long)d->F * d->C * 4 < (1L << 31) && (long)d->B * d->F * 4 < (1L << 31))) { ccvs_set_error("%s: operand beyond 2^31 bytes (32-bit buffer offsets)", who); return CCVS_ERR_ARG; }
    return CCVS_OK;
}

// The phase table of d's persistent step, written to d->program (device memory): the Gemm16 of every launch of the chain, K slicing
// exactly as launch_gemm16 decides it.  A synchronous copy on `stream`: once per descriptor, outside graph capture.
extern "C" int ccvs_gpt_decode_prepare(const ccvs_gpt_decode* d, void* stream) {
    CCVS_REQUIRE(d && d->layers && d->n_layer > 0 && d->C > 0 && d->H > 0 && d->C % d->H == 0, "ccvs_gpt_decode_prepare: bad descriptor");
    int rc = step_check(d, "ccvs_gpt_decode_prepare");
    if (rc != CCVS_OK) return rc;
    const int D = d->C / d->H;
    const int grp_rows = d->groups > 1 ? d->B / d->groups : 0;
    float* ws_slabs = (float*)d->workspace;
    int* ws_count = (int*)((char*)d->workspace + GEMM_WS_SLAB_BYTES);
    const int n_ph = 5 * d->n_layer + 1;
    const size_t bytes = (size_t)ccvs_gpt_program_bytes(d->n_layer);
    char* host = (char*)calloc(1, bytes);
    CCVS_REQUIRE(host, "ccvs_gpt_decode_prepare: out of host memory");
    StepPhase* prog = (StepPhase*)(host + STEP_HEAD_BYTES);
    {
        StepArgs& a = *(StepArgs*)host;
        a.B = d->B; a.C = d->C; a.V = d->V; a.vocab = d->vocab; a.grp_rows = grp_rows; a.pos_off = d->pos_off; a.n_phases = n_ph; a.H = d->H;
        a.tok_emb = d->tok_emb; a.pos_table = d->pos_table; a.tok = d->tok; a.len = d->len; a.x = d->x; a.logits = d->logits;
        a.bar = (StepBar*)((char*)d->workspace + GEMM_WS_BAR_OFFSET);
        a.prog = (const StepPhase*)((const char*)d->program + STEP_HEAD_BYTES);
        a.scale = 1.0f / sqrtf((float)D);
        a.noise = d->noise; a.top_k = d->top_k; a.temperature = d->temperature;
        a.adv.codes = d->codes; a.adv.codes_sB = (long)d->codes_sB; a.adv.widx = d->widx; a.adv.len = d->len;
        a.adv.rng = (d->rng && !d->noise && !d->noise_stream) ? 1 : 0; a.adv.state = d->state; a.adv.grp_rows = grp_rows;
        a.adv.noise_stream = d->noise ? nullptr : d->noise_stream;
    }
    auto slice = [&](Gemm16& g) {
        g.kz = gemm_kz(g);
        if (g.kz > 1 && cdiv(g.N, 16) * cdiv(g.M, 16) > GEMM_WS_TILES) g.kz = 1;
        g.ks = GEMM_WAVES;
        while (g.ks > 1 && g.K % (16 * g.ks * g.kz) != 0) g.ks >>= 1;
    };
    int n = 0;
    for (int l = 0; l < d->n_layer; ++l) {
        const ccvs_gpt_layer& L = d->layers[l];
        if (!(L.qkv_w && L.qkv_b && L.qkv_s && L.proj_w && L.proj_b && L.fc_w && L.fc_b && L.fc_s && L.fc2_w && L.fc2_b && L.kcache && L.vcache)) {
            free(host);
            ccvs_set_error("ccvs_gpt_decode_prepare: null pointer in layer %d", l);
            return CCVS_ERR_ARG;
        }
        {   // ln1 + QKV + cache scatter
            Gemm16& g = prog[n].g; prog[n++].kind = STEP_GEMM;
            g.x = d->x; g.ldx = d->C; g.w = L.qkv_w; g.bias = L.qkv_b; g.y = d->q; g.ldy = d->C; g.M = d->B; g.N = 3 * d->C; g.K = d->C;
            g.ln_s = L.qkv_s; g.ln_eps = d->ln_eps;
            g.kcache = L.kcache; g.vcache = L.vcache; g.C = d->C; g.H = d->H; g.D = D; g.Tq = 1; g.Tmax = d->Tmax; g.pos0 = 0; g.pos_dev = d->len;
            g.grp_rows = grp_rows;
            slice(g);
        }
        {   // attention over the cache
            Gemm16& g = prog[n].g; prog[n++].kind = STEP_ATTENTION;
            g.x = d->q; g.ldx = d->C; g.y = d->att; g.kcache = L.kcache; g.vcache = L.vcache; g.H = d->H; g.M = d->B; g.Tmax = d->Tmax;
            g.pos_dev = d->len; g.grp_rows = grp_rows;
        }
        {   // proj + residual (in place on x)
            Gemm16& g = prog[n].g; prog[n++].kind = STEP_GEMM;
            g.x = d->att; g.ldx = d->C; g.w = L.proj_w; g.bias = L.proj_b; g.res = d->x; g.y = d->x; g.ldy = d->C; g.M = d->B; g.N = d->C; g.K = d->C; g.epi = 2;
            g.ws_slabs = ws_slabs; g.ws_count = ws_count;
            slice(g);
        }
        {   // ln2 + fc + GELU
            Gemm16& g = prog[n].g; prog[n++].kind = STEP_GEMM;
            g.x = d->x; g.ldx = d->C; g.w = L.fc_w; g.bias = L.fc_b; g.y = d->h; g.ldy = d->F; g.M = d->B; g.N = d->F; g.K = d->C; g.epi = 1;
            g.ln_s = L.fc_s; g.ln_eps = d->ln_eps;
            slice(g);
        }
        {   // fc2 + residual (in place on x)
            Gemm16& g = prog[n].g; prog[n++].kind = STEP_GEMM;
            g.x = d->h; g.ldx = d->F; g.w = L.fc2_w; g.bias = L.fc2_b; g.res = d->x; g.y = d->x; g.ldy = d->C; g.M = d->B; g.N = d->C; g.K = d->F; g.epi = 2;
            g.ws_slabs = ws_slabs; g.ws_count = ws_count;
            slice(g);
        }
    }
    {   // ln_f + head
        Gemm16& g = prog[n].g; prog[n++].kind = STEP_GEMM;
        g.x = d->x; g.ldx = d->C; g.w = d->head_w; g.bias = d->head_b; g.y = d->logits; g.ldy = d->V; g.M = d->B; g.N = d->V; g.K = d->C;
        g.ln_s = d->head_s; g.ln_eps = d->ln_eps;
        slice(g);
    }
    hipError_t e = hipMemcpyAsync(d->program, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    free(host);
    if (e != hipSuccess) { ccvs_set_error("ccvs_gpt_decode_prepare: %s", hipGetErrorString(e)); return CCVS_ERR_LAUNCH; }
    return CCVS_OK;
}

static int launch_step_persistent(const ccvs_gpt_decode* d, int D, hipStream_t st) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { ccvs_set_error("ccvs_gpt_decode_step: no device properties"); return CCVS_ERR_LAUNCH; }
        n_cu = prop.multiProcessorCount;
    }
    const StepArgs* a = (const StepArgs*)d->program;   // written by ccvs_gpt_decode_prepare
    // workgroups per CU (an experiment, CCVS_STEP_WGS_PER_CU; default 1): more of them are all resident only with the chip to the step
    // itself -- beside the frame decoder a CU has room for one -- and every grid barrier then waits for 2-4 x the arrivals
    static const int per_cu = getenv_int("CCVS_STEP_WGS_PER_CU", 1) < 1 ? 1 : (getenv_int("CCVS_STEP_WGS_PER_CU", 1) > 4 ? 4 : getenv_int("CCVS_STEP_WGS_PER_CU", 1));
    const bool t2 = d->B > 32;
    size_t words = t2 ? STEP_GEMM_WORDS(2, 2) : STEP_GEMM_WORDS(1, 1);
    const size_t w_att = 16 + 4 * 256 + (size_t)d->Tmax, w_pick = PICK_SMEM_WORDS(d->V);
    if (w_att > words) words = w_att;
    if (w_pick > words) words = w_pick;
    const size_t smem = words * sizeof(float);
#define STEP_LAUNCH(Dv, T2v)                                                                                                          \
    do {                                                                                                                              \
        static bool attr_set = false;                                                                                                 \
        if (!attr_set) {                                                                                                              \
            (void)hipFuncSetAttribute((const void*)gpt_step_kernel<Dv, T2v>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                          \
        }                                                                                                                             \
        hipLaunchKernelGGL((gpt_step_kernel<Dv, T2v>), dim3(n_cu * per_cu), dim3(256), smem, st, a);                                  \
    } while (0)
    if (D == 64) { if (t2) STEP_LAUNCH(64, true); else STEP_LAUNCH(64, false); }
    else if (D == 32) { if (t2) STEP_LAUNCH(32, true); else STEP_LAUNCH(32, false); }
    else { if (t2) STEP_LAUNCH(16, true); else STEP_LAUNCH(16, false); }
#undef STEP_LAUNCH
    CCVS_CHECK_LAUNCH("ccvs_gpt_decode_step(persistent)");
    return CCVS_OK;
}

// ---------------------------------------------------------------------------------------
// One decode step: the launch sequence of ccvs_gpt_embed / ccvs_gemm_ln_qkv / ccvs_attention /
// ccvs_gemm_nt / ccvs_gemm_ln / ccvs_sample_topk for a single new position, 5 * n_layer + 3 launches
// on one stream, every per-step quantity device-resident (hipGraph-capturable).
// ---------------------------------------------------------------------------------------
extern "C" int ccvs_gpt_decode_step(const ccvs_gpt_decode* d, void* stream) {
    CCVS_REQUIRE(d && d->layers && d->tok_emb && d->pos_table && d->head_w && d->head_b && d->head_s, "ccvs_gpt_decode_step: null pointer");
    CCVS_REQUIRE(d->tok && d->codes && d->widx && d->len && d->x && d->q && d->att && d->h && d->logits && d->state,
                 "ccvs_gpt_decode_step: null state pointer");
    CCVS_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->C % d->H == 0 && d->F > 0 && d->n_layer > 0 && d->Tmax > 0 && d->vocab > 0 && d->V > 0,
                 "ccvs_gpt_decode_step: bad shape");
    const int D = d->C / d->H;
    CCVS_REQUIRE(D == 64 || D == 32 || D == 16, "ccvs_gpt_decode_step: head dim %d unsupported (16, 32, 64)", D);
    CCVS_REQUIRE(d->temperature > 0.f, "ccvs_gpt_decode_step: bad temperature");
    CCVS_REQUIRE(d->groups >= 0 && (d->groups <= 1 || d->B % d->groups == 0), "ccvs_gpt_decode_step: %d rows do not split into %d groups", d->B, d->groups);
    // (more rows than the weight-stream form takes run the row-blocked form inside launch_gemm16; row groups cannot)
    CCVS_REQUIRE(d->groups <= 1 || d->B <= GEMM_DECODE_MAX_M, "ccvs_gpt_decode_step: at most %d rows per grouped step", GEMM_DECODE_MAX_M);
    const int grp_rows = d->groups > 1 ? d->B / d->groups : 0;   // 0: one group (widx / len / state are single words)
    const size_t smem_att = (size_t)(16 + 4 * 256 + d->Tmax) * sizeof(float);
    const size_t smem_pick = (size_t)PICK_SMEM_WORDS(d->V) * sizeof(float);
    CCVS_REQUIRE(smem_att <= 64 * 1024, "ccvs_gpt_decode_step: sequence too long for the LDS score buffer");
    CCVS_REQUIRE(smem_pick <= 160 * 1024, "ccvs_gpt_decode_step: vocabulary %d too large", d->V);
    hipStream_t st = (hipStream_t)stream;
    if (d->persistent) {   // ONE launch (gpt_step_kernel) over the phase table ccvs_gpt_decode_prepare wrote to d->program
        int rc_ = step_check(d, "ccvs_gpt_decode_step");
        if (rc_ != CCVS_OK) return rc_;
        return launch_step_persistent(d, D, st);
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)sample_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }

    {   // embedding of the last picked token at row pos_off + *len
        const long total = (long)d->B * d->C;
        hipLaunchKernelGGL(gpt_embed_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, d->tok, 1L, (const int32_t*)nullptr, d->pos_off,
                           (const int32_t*)d->len, grp_rows, 1, d->tok_emb, d->pos_table, d->x, total, d->C, d->vocab);
        CCVS_CHECK_LAUNCH("ccvs_gpt_decode_step(embed)");
    }
    float* ws_slabs = (float*)d->workspace;
    int* ws_count = d->workspace ? (int*)((char*)d->workspace + GEMM_WS_SLAB_BYTES) : nullptr;
    const float scale = 1.0f / sqrtf((float)D);
    int rc;
    for (int l = 0; l < d->n_layer; ++l) {
        const ccvs_gpt_layer& L = d->layers[l];
        CCVS_REQUIRE(L.qkv_w && L.qkv_b && L.qkv_s && L.proj_w && L.proj_b && L.fc_w && L.fc_b && L.fc_s && L.fc2_w && L.fc2_b && L.kcache && L.vcache,
                     "ccvs_gpt_decode_step: null pointer in layer %d", l);
        Gemm16 g = {};  // ln1 + QKV + cache scatter
        g.x = d->x; g.ldx = d->C; g.w = L.qkv_w; g.bias = L.qkv_b; g.y = d->q; g.ldy = d->C; g.M = d->B; g.N = 3 * d->C; g.K = d->C;
        g.ln_s = L.qkv_s; g.ln_eps = d->ln_eps;
        g.kcache = L.kcache; g.vcache = L.vcache; g.C = d->C; g.H = d->H; g.D = D; g.Tq = 1; g.Tmax = d->Tmax; g.pos0 = 0; g.pos_dev = d->len;
        g.grp_rows = grp_rows;
        if ((rc = launch_gemm16(g, st, "ccvs_gpt_decode_step(qkv)")) != CCVS_OK) return rc;
        {   // attention over the cache
            const dim3 grid((unsigned)(d->B * d->H));
            if (D == 64) ATT_DECODE_LAUNCH(64, grid, smem_att, d->q, (long)d->C, L.kcache, L.vcache, d->att, d->H, 0, (const int32_t*)d->len, grp_rows, d->Tmax, scale);
            else if (D == 32) ATT_DECODE_LAUNCH(32, grid, smem_att, d->q, (long)d->C, L.kcache, L.vcache, d->att, d->H, 0, (const int32_t*)d->len, grp_rows, d->Tmax, scale);
            else ATT_DECODE_LAUNCH(16, grid, smem_att, d->q, (long)d->C, L.kcache, L.vcache, d->att, d->H, 0, (const int32_t*)d->len, grp_rows, d->Tmax, scale);
            CCVS_CHECK_LAUNCH("ccvs_gpt_decode_step(attention)");
        }
        g = Gemm16{};  // proj + residual (in place on x)
        g.x = d->att; g.ldx = d->C; g.w = L.proj_w; g.bias = L.proj_b; g.res = d->x; g.y = d->x; g.ldy = d->C; g.M = d->B; g.N = d->C; g.K = d->C; g.epi = 2;
        g.ws_slabs = ws_slabs; g.ws_count = ws_count;
        if ((rc = launch_gemm16(g, st, "ccvs_gpt_decode_step(proj)")) != CCVS_OK) return rc;
        g = Gemm16{};  // ln2 + fc + GELU
        g.x = d->x; g.ldx = d->C; g.w = L.fc_w; g.bias = L.fc_b; g.y = d->h; g.ldy = d->F; g.M = d->B; g.N = d->F; g.K = d->C; g.epi = 1;
        g.ln_s = L.fc_s; g.ln_eps = d->ln_eps;
        if ((rc = launch_gemm16(g, st, "ccvs_gpt_decode_step(fc)")) != CCVS_OK) return rc;
        g = Gemm16{};  // fc2 + residual (in place on x)
        g.x = d->h; g.ldx = d->F; g.w = L.fc2_w; g.bias = L.fc2_b; g.res = d->x; g.y = d->x; g.ldy = d->C; g.M = d->B; g.N = d->C; g.K = d->F; g.epi = 2;
        g.ws_slabs = ws_slabs; g.ws_count = ws_count;
        if ((rc = launch_gemm16(g, st, "ccvs_gpt_decode_step(fc2)")) != CCVS_OK) return rc;
    }
    {   // ln_f + head
        Gemm16 g = {};
        g.x = d->x; g.ldx = d->C; g.w = d->head_w; g.bias = d->head_b; g.y = d->logits; g.ldy = d->V; g.M = d->B; g.N = d->V; g.K = d->C;
        g.ln_s = d->head_s; g.ln_eps = d->ln_eps;
        if ((rc = launch_gemm16(g, st, "ccvs_gpt_decode_step(head)")) != CCVS_OK) return rc;
    }
    {   // pick + bookkeeping
        Advance adv = {};
        adv.codes = d->codes; adv.codes_sB = (long)d->codes_sB; adv.widx = d->widx; adv.len = d->len;
        adv.rng = (d->rng && !d->noise && !d->noise_stream) ? 1 : 0; adv.state = d->state; adv.grp_rows = grp_rows;
        adv.noise_stream = d->noise ? nullptr : d->noise_stream;
        hipLaunchKernelGGL(sample_topk_kernel, dim3(d->B), dim3(256), smem_pick, st, d->logits, (long)d->V, d->noise, d->tok, 1L, d->V,
                           d->top_k, d->temperature, adv);
        CCVS_CHECK_LAUNCH("ccvs_gpt_decode_step(pick)");
    }
    return CCVS_OK;
}
