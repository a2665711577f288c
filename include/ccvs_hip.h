/* ccvs_hip.h -- C ABI of libccvs_hip.so, the MI355X (gfx950) kernel library behind the
 * CCVS autoregressive synthesis hot path (frame encoder -> VQ lookup -> causal
 * transformer -> flow-guided frame decoder).
 *
 * The reference has no FFI for this path: its native boundary is two pybind11 torch
 * extensions (modules/upfirdn2d.cpp:21-23, modules/fused_bias_act.cpp:18-20), cupy
 * launches of raw CUDA kernels (modules/correlation.py:299-331) and aten ops.  Each entry
 * point below names the reference interface it replaces (file:line, upstream repo root).
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers owned by the caller (PyTorch allocator);
 *     outputs are pre-allocated by the caller; nothing is allocated inside;
 *   - tensors are fp32, NCHW, rows dense (W stride 1, H stride W) unless stated; where a
 *     channel slice of a wider tensor is read or written the batch / channel strides are
 *     passed explicitly in ELEMENTS;
 *   - `stream` is a hipStream_t (passed as void*); launches are asynchronous on it;
 *   - return 0 on success, negative on error; ccvs_last_error() gives the message of the
 *     last failing call on this host thread;
 *   - no internal threads; host state = the thread-local error string, lazily set kernel attributes and the mutex-guarded
 *     table of ccvs_stream_cu_limit.  Several host threads may launch on different streams of one device once every kind
 *     of call has been made at least once from one thread (the pipelined schedule warms up that way).
 */
#ifndef CCVS_HIP_H
#define CCVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCVS_OK 0
#define CCVS_ERR_ARG (-1)
#define CCVS_ERR_LAUNCH (-2)

#define CCVS_ACT_NONE 0
#define CCVS_ACT_LRELU 1 /* LeakyReLU(0.1), skip_autoencoder.py:100 */

const char* ccvs_last_error(void);
int ccvs_abi_version(void);

/* ---- convolution -------------------------------------------------------------------
 * Replaces F.conv2d / F.conv_transpose2d as called by EqualConv2d.forward
 * (models/skip_vid_generator/models/skip_autoencoder.py:53-59) plus the bias /
 * LeakyReLU(0.1) / residual `(a+b)/sqrt2` / `flow + head(feat)` glue around it
 * (skip_autoencoder.py:99-100,116,204-205,225-226).
 *
 * w_packed: [kh*kw][Cin][CoutPad] with the 1/sqrt(Cin*k*k) scale already multiplied in
 *           (tap-major, then input channel, then output channel, CoutPad = Cout rounded up
 *           to 32, or to 64 when Cout >= 64; padding is zero).
 * transposed != 0: conv_transpose2d(stride 2, padding 0); Hout = 2*Hin + kh - 2.
 * y = [ y + ] ( act(conv [+ pre] + bias) [+ residual] ) * out_scale
 *     the leading `y +` only when accumulate != 0.  bias / residual may be NULL.
 */
typedef struct ccvs_conv_desc {
    int32_t N, Cin, Hin, Win;
    int64_t in_sN, in_sC; /* element strides of x */
    int32_t Cout, CoutPad, Hout, Wout;
    int64_t out_sN, out_sC; /* element strides of y */
    int64_t res_sN, res_sC; /* element strides of residual */
    int32_t kh, kw, stride, pad, transposed;
    int32_t act, accumulate;
    float out_scale;
    /* optional pre-activation addend: act(conv + pre[n / pre_div] + bias).  Lets the part of a convolution
     * whose input is shared by pre_div consecutive batch items (the decoder feature that
     * skip_autoencoder.py:251 repeats k times) be computed once and broadcast. */
    const float* pre;
    int64_t pre_sN, pre_sC;
    int32_t pre_div;
    /* split-bf16 packed activations ("P8", ccvs_conv2d_bf16x3 only): [N][C/8][2 = hi,lo][H][W] units of 8 bf16 (16 bytes),
     * v = hi + lo -- the same bytes per element as fp32, already in the layout the kernel stages in LDS.
     * in_p8: x is such a tensor (dense; in_sN / in_sC ignored): staged by LDS-DMA with no conversion work.
     * out_p8: y is written in that form (dense; out_sN / out_sC ignored) by the epilogue, for the next convolution.
     * A chain conv -> conv run this way is bit-identical to the fp32-activation chain. */
    int32_t in_p8, out_p8;
    /* cu_limit > 0 (ccvs_conv2d_bf16x3): never occupy more than cu_limit compute units -- the tiles are launched as
     * consecutive chunks of cu_limit x (workgroups per CU of the chosen kernel) workgroups on `stream` -- so that work on
     * another stream (the latency-bound token loop of the next batch, a few hundred small workgroups per launch) always
     * finds free CUs beside the decoder's convolutions.  0: one launch, one workgroup per tile over the whole chip.
     * Results are identical either way. */
    int32_t cu_limit;
    /* Packed K tail (ccvs_conv2d_bf16x3, 3 x 3 stride-1 layers with Cin % 16 = r in {1, 2, 3} and Cin > 16; NULL = none):
     * a second copy of w_split in which the LAST 16-channel chunk -- r real channels, 16 - r of padding, nine tap steps
     * of matrix work on mostly zeros -- is re-laid as ceil(9 r / 16) steps whose K index runs over (tap, channel) pairs:
     * in tap slot j of tap row 0 of that chunk, K position kk holds w[tap t][Cin - r + c] with 16 j + kk = t r + c; the
     * other slots of the chunk are zero.  The kernel gathers the matching (tap, channel) values of the staged tile for
     * these steps, so the 49- and 99-channel inputs of Matching / Subpixel (skip_autoencoder.py:173-177,217-221) cost
     * 28 / 56 tap steps instead of 36 / 63.  Used when the launch runs the vectorised producer / consumer kernel;
     * every other form reads w_split.  Same products, same fp32 accumulator: sums differ only in their order. */
    const void* w_ktail;
} ccvs_conv_desc;

int ccvs_conv2d(const float* x, const float* w_packed, const float* bias, const float* residual, float* y,
                const ccvs_conv_desc* d, void* stream);

/* Same contract as ccvs_conv2d on the bf16 matrix cores with fp32-class accuracy: every fp32
 * operand is split v = hi + lo (bf16, round-to-nearest-even) and a product is evaluated as
 * hi*hi + hi*lo + lo*hi with fp32 accumulation (per-product error ~2^-16 relative).
 * w_split: [kh*kw][CinPad/8][2 = hi,lo][CoutPad][8] bf16, scale multiplied in before the split,
 *          CinPad = Cin rounded up to 16, CoutPad = Cout rounded up to 32, padding zero. */
int ccvs_conv2d_bf16x3(const float* x, const void* w_split, const float* bias, const float* residual, float* y,
                       const ccvs_conv_desc* d, void* stream);

/* Measurement support (host code only, no GPU call; no reference counterpart -- the reference has no profiling hooks):
 * bytes per lane of the ACTIVATION fetches of the convolution kernel whose name (as rocprofv3 prints it, e.g.
 * "void conv2d_bf16x3_pc_kernel<32, 2, -83, 4, 1>(...)") is given: 16 = aligned dwordx4 rows / LDS-DMA, 4 = dword by dword,
 * -1 = not a convolution kernel of this library (callers must fail, not guess).  tools/pmc_widths.py applies gfx950's
 * FETCH_SIZE correction with it (a 16-byte-per-lane stream is tallied at half its bytes, MI355X_MICROARCH.md "HBM"). */
int ccvs_conv_fetch_bytes_per_lane(const char* kernel_name);

/* Which 3 x 3 stride-1 layers of ccvs_conv2d_bf16x3 run as RESIDENT workgroups that walk their tiles (conv2d_bf16_pt.h; the same bits as
 * the per-tile kernels): bit 0 the fp32-input 128-channel layers, bit 1 the packed-input layers; process-wide, returns the previous
 * mode (mode < 0: query only).  Default 1.  The host lowers it to 0 while several batches are in flight (ccvs_amd/helpers/pipeline.py):
 * alone on the chip the resident form is faster, beside the token loops of other batches it is not.  No reference
 * counterpart: the reference's convolutions are cuDNN calls (models/skip_vid_generator/models/skip_autoencoder.py:53-59). */
int ccvs_conv_persistent_tiles(int32_t mode);

/* ---- FIR resampling ------------------------------------------------------------------
 * Replaces upfirdn2d(input, kernel, up, down, pad) (modules/upfirdn2d.py:145-159, CUDA
 * upfirdn2d_kernel.cu:107-207, pybind upfirdn2d.cpp:21-23) for the 4-tap separable
 * kernel outer([1,3,3,1])/64 * gain used by Blur (skip_autoencoder.py:27-37).
 * x [N,C,H,W] dense -> y [N,C,Ho,Wo], Ho = (H*up + pad0 + pad1 - 4)/down + 1.
 * y = ( act(fir(x)) [+ residual] ) * out_scale ; residual dense like y, may be NULL.
 */
int ccvs_upfirdn2d(const float* x, float* y, const float* residual, int64_t NC, int32_t H, int32_t W, int32_t up,
                   int32_t down, int32_t pad0, int32_t pad1, float gain, int32_t act, float out_scale, void* stream);

/* Depthwise ConvTranspose2d(C, C, 4, stride 2, padding 1, groups=C, bias=False)
 * (skip_autoencoder.py:153-154,168).  w [C,1,4,4]; x [N,C,H,W] (batch stride x_sN, channel
 * planes dense) -> y [N,C,2H,2W] (batch stride y_sN). */
int ccvs_dwconvT4x4s2(const float* x, int64_t x_sN, const float* w, float* y, int64_t y_sN, int32_t N, int32_t C, int32_t H,
                      int32_t W, void* stream);

/* ---- cost volume / warping -----------------------------------------------------------
 * ccvs_correlation7x7 replaces FunctionCorrelation(first, second, stride)
 * (modules/correlation.py:279-338,405-406; kernels :11-100) fused with the
 * F.leaky_relu(., 0.1) applied to it (skip_autoencoder.py:197).
 * first is indexed by n / first_div (first_div = k lets one projected decoder feature
 * serve its k context pairs without materialising the repeat of skip_autoencoder.py:251).
 * out [N,49,ceil(H/s),ceil(W/s)].
 */
int ccvs_correlation7x7(const float* first, const float* second, float* out, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t stride, int32_t first_div, int32_t lrelu, void* stream);

/* backwarp(input, flow*flow_mult, grid) (skip_autoencoder.py:120-128): grid_sample
 * bilinear / zeros / align_corners=False on the pixel-centre grid with the flow divided by
 * ((W-1)/2, (H-1)/2).  x [N,C,H,W] with strides, flow [N,2,H,W] (batch stride flow_sN),
 * y strided. */
int ccvs_backwarp(const float* x, int64_t x_sN, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult, float* y,
                  int64_t y_sN, int64_t y_sC, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* The same two calls with the sources given as a list of k context tensors: pair b of the N (backwarp: N = pairs;
 * warp_fuse_blend: N = frames) reads context j = b % k of frame b / k at p[j] + (b / k) * sN[j] (channel planes
 * dense).  The k contexts of a decode step are slots of the per-level context ring (plus, point-to-point, one tensor
 * outside it): this form reads them in place instead of the stacked copy of skip_autoencoder.py:251. */
#define CCVS_MAX_CTX 16
typedef struct ccvs_ctx_list {
    int32_t k;
    const float* p[CCVS_MAX_CTX];
    int64_t sN[CCVS_MAX_CTX];
} ccvs_ctx_list;
int ccvs_backwarp_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult, float* y,
                      int64_t y_sN, int64_t y_sC, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int ccvs_warp_fuse_blend_ctx(float* dec, int64_t dec_sN, int64_t dec_sC, const ccvs_ctx_list* ctx, const float* flows,
                             int64_t flows_sN, const float* occs, int64_t occs_sN, float flow_mult, int32_t N, int32_t C,
                             int32_t H, int32_t W, void* stream);

/* backwarp of the k contexts straight into the packed (P8, see ccvs_conv_desc.in_p8) input of Subpixel's first convolution,
 * cat([warped, flow, occ]) of skip_autoencoder.py:222-224 without its dec block: y_p8 = [N][C/8 + 1][2 = hi,lo][H][W] units of 8
 * bf16; groups 0 .. C/8 - 1 = backwarp(ctx, flow * flow_mult) (the same samples as ccvs_backwarp_ctx), group C/8 = (flow x, flow y,
 * occ, 0, 0, 0, 0, 0) as stored in flow_occ [N,3,H,W] (batch stride fo_sN).  C % 8 == 0, W % 4 == 0.  The convolution that reads it has
 * Cin = C + 8 with zero weights for the five padding channels. */
int ccvs_backwarp_p8_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow_occ, int64_t fo_sN, float flow_mult, void* y_p8,
                         int32_t N, int32_t C, int32_t H, int32_t W, void* stream);

/* Matching's proj(backwarp(inter, flow * flow_mult)) (skip_autoencoder.py:186-190, ConvLayer(C, max(16, C/4), 1)) in one pass:
 * y[n][o] = act(bias[o] + sum_c w_t[c][o] * backwarp(ctx)[n][c]) -- the warped C-channel tensor is never written.
 * ctx as in ccvs_backwarp_ctx; w_t [Cin][CoutPad] fp32, the 1x1 weight transposed with the EqualConv2d scale multiplied in
 * and zero columns up to CoutPad in {16, 24, 48, 96}; y [N,Cout,H,W] dense.  fp32 FMAs. */
int ccvs_backwarp_proj_ctx(const ccvs_ctx_list* ctx, int64_t x_sC, const float* flow, int64_t flow_sN, float flow_mult,
                           const float* w_t, const float* bias, float* y, int32_t N, int32_t Cin, int32_t Cout, int32_t CoutPad,
                           int32_t H, int32_t W, int32_t act, void* stream);

/* Tail of InterBlock.forward (skip_autoencoder.py:254-264): final back-warp of the k
 * context features, confidence fusion over k (eps 1e-6) and occlusion blend, written in
 * place into the first C channels of the decoder feature.
 * dec [N,C,H,W] strided (in/out); ctx [N*k,C,H,W] dense; flows [N*k,2,H,W] and occs
 * [N*k,1,H,W] with batch strides flows_sN / occs_sN. */
int ccvs_warp_fuse_blend(float* dec, int64_t dec_sN, int64_t dec_sC, const float* ctx, const float* flows, int64_t flows_sN,
                         const float* occs, int64_t occs_sN, float flow_mult, int32_t N, int32_t k, int32_t C, int32_t H,
                         int32_t W, void* stream);

/* Second half of the 3-output flow / occlusion heads (skip_autoencoder.py:176-177,204-205,225-226).
 * The k x k, 3-output convolution is run as ccvs_conv2d[_bf16x3] with 3k outputs and padding k/2 on both axes, either
 *   vertical == 0: a k x 1 kernel (row kx*3+co holds tap column kx of output co), giving t [N,3k,H,W+k-1], and
 *     y[n][co][yy][x] (+)= bias[co] + sum_kx t[n][kx*3+co][yy][x+kx]; or
 *   vertical != 0: a 1 x k kernel (row ky*3+co holds tap row ky of output co), giving t [N,3k,H+k-1,W], and
 *     y[n][co][yy][x] (+)= bias[co] + sum_ky t[n][ky*3+co][yy+ky][x]   (all k taps of a row in ONE step of the
 *     convolution kernel: 9x fewer workgroup barriers than the k x 1 form for the 9 x 9 heads).
 * y batch stride y_sN, planes dense. */
int ccvs_tap_shift_add(const float* t, const float* bias, float* y, int64_t y_sN, int32_t N, int32_t k, int32_t H, int32_t W,
                       int32_t accumulate, int32_t vertical, void* stream);

/* ---- vector quantiser ----------------------------------------------------------------
 * ccvs_vq_argmin replaces VectorQuantizer.forward's distance + argmin
 * (modules/quantize.py:40-50): idx[n*HW + p] = argmin_j (|z|^2 + |e_j|^2) - 2 z.e_j,
 * lowest index on ties.  z [N,C,HW] (NCHW), codebook_t [C][n_e] (transposed embedding),
 * e_sq [n_e]; idx int64 [N*HW].  C even (n_e a multiple of 32: the MFMA stream), or C == 1 -- the scalar state
 * quantiser VectorQuantizer(state_num, 1) of state_model.py:55, any n_e. */
int ccvs_vq_argmin(const float* z, const float* codebook_t, const float* e_sq, int64_t* idx, int32_t N, int32_t C, int32_t HW,
                   int32_t n_e, void* stream);

/* VectorQuantizer.embed_code + the transposes of QVidModel.decode
 * (modules/quantize.py:76-83, quantized_video_model.py:832-833):
 * z[n][c][p] = codebook[code[n*HW+p]][c]. */
int ccvs_embed_gather(const int64_t* code, const float* codebook, float* z, int32_t N, int32_t C, int32_t HW, int32_t n_e,
                      void* stream);

/* The encoder's output normalisation, `opt.normalize_out` (models/skip_vid_generator/models/skip_autoencoder.py:348-349):
 * x[n][c][p] /= sqrt(sum_c x[n][c][p]^2), in place; x [N,C,HW] (NCHW).  A position whose channels are all zero becomes NaN,
 * as in the reference (0 / 0). */
int ccvs_l2_normalize_channels(float* x, int32_t N, int32_t C, int64_t HW, void* stream);

/* ---- transformer ---------------------------------------------------------------------
 * Together these replace GPT.forward (models/skip_vid_generator/models/mingpt.py:232-305)
 * and Transformer.get_icode (models/transformer_model.py:395-409), restructured around a
 * KV cache (the reference recomputes the whole prefix per token, transformer_model.py:350).
 * `pos_dev` (nullable) is a device-resident int32 added to `pos0`: a decode step captured in a
 * hipGraph then replays at advancing positions by incrementing that word on the device.
 */
/* x[(b,t)][:] = tok_emb[idx[b*idx_sB + t]] + pos_table[pos_off[b] + pos0 + t], t < Tq
 * (mingpt.py:234-236,242-244).  pos_table rows are the positional embeddings pre-summed by
 * the host for the call (s_emb + t_emb[+delta_length], h/w/t_emb or pos_emb:
 * mingpt.py:186-217); pos_off (int32 [B], may be NULL) selects a per-sample table. */
int ccvs_gpt_embed(const int64_t* idx, int64_t idx_sB, const int32_t* pos_off, int32_t pos0, const int32_t* pos_dev, int32_t Tq,
                   const float* tok_emb, const float* pos_table, float* x, int32_t B, int32_t C, int32_t vocab, void* stream);

/* nn.LayerNorm over the last dim (mingpt.py:103-104,168), eps 1e-5. */
int ccvs_layernorm(const float* x, const float* gamma, const float* beta, float* y, int32_t rows, int32_t C, void* stream);

/* y = epilogue(x @ W^T + bias): nn.Linear (mingpt.py:44-51,107-110,169).
 * x [M,K] row stride ldx, W [N,K] (torch Linear layout), y [M,N] row stride ldy.
 * epilogue: 0 none, 1 GELU(erf) (mingpt.py:109), 2 add residual (res [M,N], stride ldy); OR-ed with CCVS_GEMM_SEQ for a
 * whole-sequence call (prefill, teacher-forced forward, re-prefill of a slid window).  The flag picks the kernel form by the
 * KIND of call instead of by M: a row's bits then do not depend on how many other rows share the launch (a batch prefilled
 * alone and the same batch stacked into a token group agree bit for bit).  ccvs_gemm_ln_qkv derives it from Tq > 1.
 * workspace (may be NULL): ccvs_gemm_workspace_bytes() bytes of device memory, ZEROED once by the caller and
 * then owned by this call chain; with it, GEMMs with few output columns also split K across workgroups
 * (deterministic last-arriver reduction) so that every CU streams weights. */
#define CCVS_GEMM_SEQ 0x100
int64_t ccvs_gemm_workspace_bytes(void);
int ccvs_gemm_nt(const float* x, int64_t ldx, const float* w, const float* bias, const float* res, float* y, int64_t ldy,
                 int32_t M, int32_t N, int32_t K, int32_t epilogue, void* workspace, void* stream);

/* LayerNorm folded into the following Linear (mingpt.py:115-116: x + attn(ln1(x)), mlp(ln2(x))):
 *   LN(x) @ W^T + b = rstd * (x @ W'^T - mean * s) + b',  W' = W*gamma, s = rowsum(W'), b' = b + W beta
 * (w_gamma / bias_beta / w_rowsum are packed once by the host).  epilogue 0 none, 1 GELU. */
int ccvs_gemm_ln(const float* x, int64_t ldx, const float* w_gamma, const float* bias_beta, const float* w_rowsum, float eps,
                 float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t epilogue, void* stream);

/* ln1 + fused [query; key; value] projection (mingpt.py:67-69): q [B*Tq, C] is written dense,
 * the key / value column blocks go straight into kcache / vcache [B,H,Tmax,C/H] at positions
 * pos0 (+ *pos_dev) + t. */
int ccvs_gemm_ln_qkv(const float* x, int64_t ldx, const float* w_gamma, const float* bias_beta, const float* w_rowsum, float eps,
                     float* q, float* kcache, float* vcache, int32_t B, int32_t Tq, int32_t C, int32_t H, int32_t pos0,
                     const int32_t* pos_dev, int32_t Tmax, void* stream);

/* Causal attention against a KV cache (mingpt.py:67-77).
 * q [B,Tq,H*D] (batch stride q_sB, row stride ldq); kcache/vcache [B,H,Tmax,D]; query t
 * attends cache positions 0 .. pos0+t.  out [B,Tq,H*D] dense.  D in {16, 32, 64}. */
int ccvs_attention(const float* q, int64_t q_sB, int64_t ldq, const float* kcache, const float* vcache, float* out, int32_t B,
                   int32_t H, int32_t Tq, int32_t pos0, const int32_t* pos_dev, int32_t Tmax, int32_t D, void* stream);

/* Scatter new K/V rows [B,Tq,H*D] (batch stride sB, row stride ld) into the caches at pos0.. */
int ccvs_kv_append(const float* k, const float* v, int64_t sB, int64_t ld, float* kcache, float* vcache, int32_t B, int32_t H,
                   int32_t Tq, int32_t pos0, const int32_t* pos_dev, int32_t Tmax, int32_t D, void* stream);

/* get_icode: logits/temperature -> top-k mask (ties kept) -> softmax -> pick.
 * noise == NULL: greedy argmax (sample=False, topk k=1).  noise [B,V] ~ Exp(1):
 * argmax(p / noise), the algorithm behind torch.multinomial(p, 1).
 * logits [B,V] row stride ld; out int64 [B] written at out[b*out_stride]. */
int ccvs_sample_topk(const float* logits, int64_t ld, const float* noise, int64_t* out, int64_t out_stride, int32_t B, int32_t V,
                     int32_t top_k, float temperature, void* stream);
/* The same pick with the Exp(1) noise drawn in the kernel: Philox4x32-10 keyed by (key0, key1), counter
 * (element, row0 + b, step, call) -- the draw of a clip depends on its GLOBAL index row0 + b only, so a batch sharded
 * over ranks samples the same tokens for any world size (same words as ccvs_gpt_decode.state, see below). */
int ccvs_sample_topk_philox(const float* logits, int64_t ld, int64_t* out, int64_t out_stride, int32_t B, int32_t V, int32_t top_k,
                            float temperature, uint32_t key0, uint32_t key1, uint32_t row0, uint32_t step, uint32_t call, void* stream);

/* get_icode with n picks per row and their log-probabilities: the proposals of beam search (transformer_model.py:358-391 calling
 * :395-409 with n = beam_size).  noise == NULL: the n most probable tokens (torch.topk(probs, n)); noise [B,V] ~ Exp(1): the n
 * largest of probs / noise, i.e. torch.multinomial(probs, n) without replacement.  Best first, ties to the lowest index.
 * out_idx int64 [B,n], out_logp float [B,n] = log(probs[pick]).  2 V floats of LDS: V <= ~20000. */
int ccvs_sample_topn(const float* logits, int64_t ld, const float* noise, int64_t* out_idx, float* out_logp, int32_t B, int32_t V,
                     int32_t top_k, float temperature, int32_t n, void* stream);

/* One whole KV-cached decode step of the sampling loop (transformer_model.py:343-350,395-409 calling
 * mingpt.py:219-305 for ONE new position): embed `tok` -> n_layer x [ln1+QKV+cache | attention |
 * proj+res | ln2+fc+GELU | fc2+res] -> ln_f+head -> get_icode pick -> codes[b][*widx] = tok[b] = pick;
 * ++*widx; ++*len; ++state[0].  5*n_layer+3 launches on `stream`.  All per-step state is device-resident,
 * so the call is hipGraph-capturable and a captured step replays unchanged.
 * `state`: int32[8] owned by the caller: [0] steps completed (also a Philox counter word), [2] internal
 * (zero between calls), [3] call index (Philox counter word: successive token windows of one clip), [4..5] Philox key,
 * [6] GLOBAL clip index of row 0 (rank r of a sharded batch passes its first clip, so a clip's noise does not depend on
 * the world size); the caller zeroes [0] and [2] and sets the rest when a sequence starts. */
typedef struct ccvs_gpt_layer {
    const float *qkv_w, *qkv_b, *qkv_s; /* ln1 folded into [q;k;v]: W*gamma [3C,C], b + W beta [3C], rowsum(W*gamma) [3C] */
    const float *proj_w, *proj_b;       /* [C,C], [C] */
    const float *fc_w, *fc_b, *fc_s;    /* ln2 folded into mlp[0]: [F,C], [F], [F] */
    const float *fc2_w, *fc2_b;         /* mlp[3]: [C,F], [C] */
    float *kcache, *vcache;             /* [B,H,Tmax,C/H] */
} ccvs_gpt_layer;

typedef struct ccvs_gpt_decode {
    int32_t B, C, H, F, n_layer, Tmax, vocab, V; /* F = mlp width (4C); vocab = embedding rows; V = head outputs */
    float ln_eps;
    const ccvs_gpt_layer* layers;       /* host array [n_layer] */
    const float *tok_emb, *pos_table;   /* [vocab,C]; [rows,C] pre-summed positional rows (as ccvs_gpt_embed) */
    int32_t pos_off;                    /* embedding row of this step = pos_off + *len */
    const float *head_w, *head_b, *head_s; /* ln_f folded into head: W*gamma [V,C], W beta [V], rowsum [V] */
    int64_t* tok;                       /* [B] in: last token, out: picked token */
    int64_t* codes; int64_t codes_sB;   /* generated sequences */
    int32_t *widx, *len;                /* device-resident write index / cache length */
    float *x, *q, *att, *h, *logits;    /* scratch [B,C] [B,C] [B,C] [B,F] [B,V] */
    const float* noise;                 /* [B,V] Exp(1) noise (host-reproducible sampling), or NULL */
    int32_t rng;                        /* noise == NULL: 0 greedy pick, 1 draw the Exp(1) noise in the kernel (Philox4x32-10,
                                           key = state[4..5], counter = (element, state[6] + row, step = state[0], call = state[3])) */
    int32_t top_k; float temperature;
    void* workspace;                    /* ccvs_gemm_workspace_bytes(), zeroed once */
    int32_t* state;
    /* Row groups (0 or 1: none).  groups > 1: the B rows are `groups` equal blocks of B / groups consecutive rows -- the clips of
     * `groups` generation batches whose token loops advance in ONE step: every weight matrix is streamed once per step for
     * all of them (the GEMMs run B rows against one pass over W), while everything that belongs to a batch stays per
     * group: widx, len are int32[groups], state is int32[groups][8] (own step counter, call index, Philox key and first
     * global clip index), each group's cache rows are written and attended at its own length, and the last row of a GROUP
     * to be picked advances that group's words.  A row's arithmetic does not depend on the rows it shares a step with:
     * the result is bit-identical to `groups` separate steps (tests/test_pipeline_gpu.py).  B <= 256. */
    int32_t groups;
    /* Host-drawn sampling noise as a whole STREAM (ABI 5; NULL = none; used when `noise` is NULL, overrides `rng`): a DEVICE array of
     * max(groups, 1) device pointers; entry g -> float [steps][B / groups][V], the Exp(1) blocks torch.multinomial would draw for
     * group g's batch (transformer_model.py:395-409: one [rows, V] block per pick, in the order of the process generator).  The
     * step reads block state[g][0] -- the steps that group has completed -- so a captured step replays through the stream and
     * the host only rewrites the pointer table (stream-ordered) when a new sequence starts: reference-seed sampling inside
     * hipGraph replays and inside row groups. */
    const float* const* noise_stream;
    /* ABI 6.  0: the step is 5 * n_layer + 3 dependent launches (above).  1: ONE launch of resident workgroups (one per CU) that walk the
     * same phases with in-launch grid barriers (gpt.hip, gpt_step_kernel): the same tile bodies in the same order, so the tokens are
     * bit-identical to the launch chain's; needs `workspace` and `program`, B <= 256.  A step holds its CU slots for its whole
     * duration: meant for schedules with ONE token loop in flight per GPU (two persistent steps beside the frame decoder wait
     * for each other's slots).  Check ccvs_gpt_decode_status(workspace, stream) when a sequence is done. */
    int32_t persistent;
    void* program;                      /* persistent = 1: device buffer of ccvs_gpt_program_bytes(n_layer) bytes, filled by ccvs_gpt_decode_prepare */
} ccvs_gpt_decode;
int ccvs_gpt_decode_step(const ccvs_gpt_decode* d, void* stream);
/* Synchronises `stream` and returns 0 unless a grid barrier of a persistent decode step launched with this workspace gave up (a
 * workgroup that never became resident within 5 s): the tokens of that step are then invalid and the call says which phase. */
int ccvs_gpt_decode_status(const void* workspace, void* stream);
/* The phase table of the persistent step: one entry per launch of the chain it replaces (operands, K slicing as the chain's launcher
 * decides it), written into d->program by a synchronous copy on `stream` -- once per descriptor (and again when a pointer or shape
 * in it changes), outside graph capture. */
int64_t ccvs_gpt_program_bytes(int32_t n_layer);
int ccvs_gpt_decode_prepare(const ccvs_gpt_decode* d, void* stream);

/* ---- sharing the chip between two streams ------------------------------------------
 * ccvs_stream_cu_limit(stream, n): work submitted to `stream` from now on occupies at most n compute units (0 lifts the
 * budget).  The HBM-bound kernels (FIR, warps, cost volume, depthwise up-sampling, tap sums, uint8 pack) then run as a
 * persistent grid of n x (workgroups per CU) workgroups that stride over their blocks, and the convolutions default their
 * ccvs_conv_desc.cu_limit to n.  Used by the two-batches-in-flight schedule: the frame decoder of batch i (this budget) runs
 * beside the token loop of batch i+1 (a chain of ~120 small dependent launches per token on a high-priority stream, which
 * needs free CUs the moment each launch arrives).  Results do not depend on the budget.  Host state: one table of at most
 * 16 (stream, budget) pairs behind a mutex; a budget of 0 removes the stream's entry. */
int ccvs_stream_cu_limit(void* stream, int32_t cu_limit);

/* ---- output stage ---------------------------------------------------------------------
 * save_video_batch's clamp / rescale / x255 / uint8 / channels-last pack
 * (helpers/generator.py:306-309).  vid [N,3,H,W] fp32 in [lo,hi] -> out [N,H,W,3] u8. */
int ccvs_pack_u8(const float* vid, uint8_t* out, int64_t N, int32_t H, int32_t W, float lo, float hi, void* stream);
/* The imagenet_norm branch of the same stage (helpers/generator.py:303-305,309): vid *= std; vid += mean; clamp(0, 1);
 * x255; truncate -- each step rounded to fp32 on its own like the reference's separate tensor ops (byte-exact).
 * std3 / mean3: HOST pointers to the three per-channel constants. */
int ccvs_pack_u8_norm(const float* vid, uint8_t* out, int64_t N, int32_t H, int32_t W, const float* std3, const float* mean3,
                      void* stream);

/* ---- evaluation metrics on the produced clips --------------------------------
 * tools/pytorch_metrics/metrics.py:24-25 get_psnr = piq.psnr(x, y, data_range=1., reduction='mean') (piq 0.5.4): per image
 * -10 log10(mean((x/R - y/R)^2) + 1e-8); out[N] fp32 (the caller takes the mean).  x, y: [N, per_image] fp32. */
int ccvs_psnr(const float* x, const float* y, float* out, int64_t N, int64_t per_image, float data_range, void* stream);
/* tools/pytorch_metrics/metrics.py:15-22 get_ssim = skimage.metrics.structural_similarity on every 2-D plane x[i, c], y[i, c]
 * with scikit-image 0.17.2's defaults: 7 x 7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, float64 arithmetic,
 * mean over the windows inside the plane; data_range is the caller's (the reference passes float planes and no data_range,
 * for which skimage 0.17.2 takes the dtype's span: 2).  x, y: [planes, H, W] fp32; out[planes] fp64;
 * workspace: ccvs_ssim_workspace_bytes(planes, H, W) bytes of device memory. */
int64_t ccvs_ssim_workspace_bytes(int64_t planes, int32_t H, int32_t W);
int ccvs_ssim(const float* x, const float* y, double* out, void* workspace, int64_t planes, int32_t H, int32_t W, double data_range,
              void* stream);
/* tools/pytorch_metrics/metrics.py:115-124 `upscale`: F.interpolate(frames, size=(OH, OW), mode='bilinear') (align_corners
 * False) of [planes, H, W] fp32 planes, in torch's own formulation (source index, blend order). */
int ccvs_resize_bilinear(const float* x, float* out, int64_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCVS_HIP_H */
