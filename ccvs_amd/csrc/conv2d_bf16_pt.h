// The split-bf16 3 x 3 convolution as RESIDENT workgroups that walk their tiles ("persistent tiles", PT) -- the producer / consumer
// kernel of conv2d_bf16_kernels.h (same LDS images, same MFMA order per output: bit-identical results) with the tile loop INSIDE the
// workgroup, so that a tile's prologue runs under the step loop of the tile before it:
//   * one workgroup per CU (gridDim = CUs), workgroup g takes the tiles g, g + gridDim, ... of the launch's XCD-aware tile order;
//   * the two roles are separate LOOPS OVER TILES (`if (producer) { for tiles ... } else { for tiles ... }`): written as one tile
//     loop around the role branch -- rounds 2 and 5 -- the staging registers in flight across a tile boundary are live through the
//     MFMA branch as far as the register allocator can tell (the role is a run-time value), 64 registers on top of the 208 of the
//     MFMA loop: spilled.  Unswitched by hand, the kernel's allocation is the MFMA role's as before;
//   * the staging waves treat the chunks of ALL their tiles as one stream, two chunks ahead (fp32 input) or one (packed input,
//     LDS-DMA): the first chunk of tile T+1 is requested while tile T computes and lands in a THIRD halo buffer (`slot 2`) during T's
//     last chunk -- the epilogue stages its passes over buffers 0 / 1 --, its second and third chunk are in flight (registers)
//     across T's epilogue;
//   * the MFMA waves request the first tap row of T+1's weights during T's last step;
//   * so a tile starts with its first chunk and weights in LDS: what is left of the skeleton between two step loops is the epilogue
//     and one barrier.  (Per tile before: workgroup dispatch ~0.9 us + ~1400 instructions of address set-up + the first chunk's and
//     first weights' round trip = 3.5 us of a 17-26 us tile, profiles/r05_conv_ablate_cycles.txt.)
// Scope: what the decoder's large layers are -- 3 x 3, stride 1, padding 1, 32-pixel-wide tiles that divide the image, every
// output channel real, 16-byte aligned rows, at most one epilogue addend (the launcher, launch_conv_pt, checks; anything else takes
// the kernels of conv2d_bf16_kernels.h).  Reference of the arithmetic: skip_autoencoder.py:53-59 (ConvLayer), :173-177, :215-221.
#pragma once
#include "conv2d_bf16_kernels.h"

// tile coordinates of linear tile id w_ (CONV_TILE_COORDS of conv_common.h for a 1-D walk)
__device__ __forceinline__ void pt_tile_coords(const ConvK& p, int w_, int& bx, int& by, int& bz) {
    if (p.xcd_chunk > 0) w_ = (w_ & 7) * p.xcd_chunk + (w_ >> 3);
    if (p.zi > 1) {
        const int per_ = p.gx * p.gy * p.zi;
        const int g_ = w_ / per_, q_ = w_ - g_ * per_;
        const int t_ = q_ / p.zi;
        bz = g_ * p.zi + (q_ - t_ * p.zi);
        bx = t_ % p.gx;
        by = t_ / p.gx;
    } else {
        bx = w_ % p.gx;
        const int r_ = w_ / p.gx;
        by = r_ % p.gy;
        bz = r_ / p.gy;
    }
}

#ifndef PT_DMA_MID
#define PT_DMA_MID 1   // the weight DMA of the next step between the first and the second tap (0: in front of the step)
#endif
#ifndef PT_NUM_VGPR
#define PT_VGPR_ATTR
#else
#define PT_VGPR_ATTR __attribute__((amdgpu_num_vgpr(PT_NUM_VGPR)))
#endif
// MB: 32-channel blocks per workgroup; PP: 32-pixel blocks per MFMA wave (2: 8 x 32 pixels, 4: 16 x 32); P8IN: packed split-bf16
// input (LDS-DMA staging) instead of fp32 rows (aligned dwordx4 + conversion)
template <int MB, int PP, bool P8IN>
__global__ __launch_bounds__(512, 2) PT_VGPR_ATTR void conv2d_bf16x3_pt_kernel(ConvK p, const uint4* __restrict__ wsplit, int CinG, int total) {
    constexpr int TW = 32, NPIX = 128 * PP, TH = NPIX / TW, NT = 32 * MB;
    constexpr int IH = TH + 2, IW = TW + 2;
    constexpr int XSH = P8IN ? 0 : 3, NQ = 10;                 // fp32 rows: the LDS image starts 3 pixels left of the halo (a 4-pixel boundary), 10 float4 columns
    constexpr int IWS = P8IN ? IW : 4 * NQ + 1;                // LDS row pitch in pixels (odd for the fp32 form: conflict-free ds_write_b128)
    constexpr int plane = IH * IWS, in_sz = 4 * plane, w_sz = 3 * 4 * NT;
    constexpr int NI = PP / 2;                                 // fp32 staging: (4 pixels x 8 channels) items per staging thread
    constexpr int NJ = (4 * plane + 255) / 256;                // packed staging: 16-byte slots per staging thread
    constexpr int CB_WR = (3 * 4 * NT + 255) / 256;            // 16-byte units of a tap row of weights per MFMA-wave thread
    constexpr int wunits = 3 * 4 * NT;
    static_assert(P8IN || NI == 1, "fp32 staging of the 512-pixel tile would need 166 KB of LDS with the third halo buffer");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    __shared__ float bias_all[512];
    uint4* const in_buf = smem4;                       // [2][half][hi|lo][plane]   chunks >= 1 of a tile, by parity; the epilogue's stage
    uint4* const w_buf = smem4 + 2 * in_sz;            // [2][tap column][half][hi|lo][NT]
    uint4* const slot2 = smem4 + 2 * in_sz + 2 * w_sz; // chunk 0 of a tile
    float* const stage = reinterpret_cast<float*>(smem4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int rt = producer ? tid - 256 : tid, rw = wave & 3;
    for (int i = tid; i < p.CoutPad && i < 512; i += 512) bias_all[i] = (p.bias && i < p.Cout) ? p.bias[i] : 0.f;

    const int nchunks = (p.Cin + CB_CC - 1) / CB_CC;
    const bool ktail = !P8IN && p.ktail > 0;
    const int cin_ = p.Cin, cout_pad = p.CoutPad;
    const long in_sC = p.in_sC;
    const int stride_t = (int)gridDim.x;
    const int add_kind = conv_add_kind(p);
    // the one addend of the fp32 epilogue (1 pre-activation image, 2 residual, 3 accumulate).  Summed, not selected: hipcc folds a
    // select of two loads from the by-value argument block into ONE load from a selected address -- and the whole block then lives in scratch
    const float* a_ptr = add_kind == 1 ? p.pre : (add_kind == 2 ? p.res : p.y);
    const long a_sN = (add_kind == 1) * p.pre_sN + (add_kind == 2) * p.res_sN + (add_kind == 3) * p.out_sN;
    const long a_sC = (add_kind == 1) * p.pre_sC + (add_kind == 2) * p.res_sC + (add_kind == 3) * p.out_sC;

    // ---- the epilogue's store phase (all 8 waves; the MFMA waves have written the pass to `stage` and everyone has met at a barrier) ----
    auto epi_store_f32 = [&](int m, int tx, int ty, int n, int n0) __attribute__((always_inline)) {
        constexpr int NIT = 2 * PP;   // 16-byte pieces of a 32-channel pass per thread
        int tid_ = tid;   // opaque: the piece offsets are tile-invariant per thread -- hipcc would hoist them out of the TILE loop and keep them live across the MFMA loop
        asm volatile("" : "+v"(tid_));
        const float* abase = a_ptr + (long)(add_kind == 1 ? n / p.pre_div : n) * a_sN;
        float* ybase = p.y + (long)n * p.out_sN;
#define PT_EPI_OFFS(i)                                                                  \
        const int idx4_ = tid_ + 512 * (i);                                             \
        const int col_ = idx4_ / (NPIX / 4), px_ = (idx4_ % (NPIX / 4)) * 4;            \
        const int prow_ = px_ / TW, pcol_ = px_ - prow_ * TW;                           \
        const long opix_ = (long)(ty * TH + prow_) * p.Wout + tx * TW + pcol_;          \
        const int co_ = n0 + m * 32 + col_;
#define PT_EPI_FINISH(ADD1, ADD2, ADD3)                                                 \
    _Pragma("unroll") for (int i = 0; i < NIT; ++i) {                                   \
        PT_EPI_OFFS(i)                                                                  \
        const float4 a4 = *reinterpret_cast<const float4*>(stage + col_ * NPIX + px_); \
        float v[4] = {a4.x, a4.y, a4.z, a4.w};                                          \
        const float bv = bias_all[co_];                                                 \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                 \
            float t = (v[j] + (ADD1)) + bv;                                             \
            if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);                                \
            t = (t + (ADD2)) * p.out_scale;                                             \
            v[j] = t + (ADD3);                                                          \
        }                                                                               \
        cb_store16(ybase + (long)co_ * p.out_sC + opix_, f32x4{v[0], v[1], v[2], v[3]}); \
    }
        if (add_kind == 0) {
            PT_EPI_FINISH(0.f, 0.f, 0.f)
        } else {
            f32x4 ad[NIT];
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                PT_EPI_OFFS(i)
                ad[i] = *reinterpret_cast<const f32x4*>(abase + (long)co_ * a_sC + opix_);
            }
            if (add_kind == 1) { PT_EPI_FINISH(ad[i][j], 0.f, 0.f) }
            else if (add_kind == 2) { PT_EPI_FINISH(0.f, ad[i][j], 0.f) }
            else { PT_EPI_FINISH(0.f, 0.f, ad[i][j]) }
        }
#undef PT_EPI_OFFS
#undef PT_EPI_FINISH
    };
    auto epi_store_p8 = [&](int m, int tx, int ty, int n, int n0) __attribute__((always_inline)) {
        uint4* y4 = reinterpret_cast<uint4*>(p.y);
        const int gout = (p.Cout + 7) >> 3;
        const long hw_out = (long)p.Hout * p.Wout;
        constexpr int NI8 = PP;   // (pixel, 8 channels) items of a 32-channel pass per thread
        int tid_ = tid;   // (opaque, see epi_store_f32)
        asm volatile("" : "+v"(tid_));
#define PT_P8_ITEM(i)                                                                   \
        const int item_ = tid_ + 512 * (i);                                             \
        const int gq_ = item_ / NPIX, px_ = item_ - gq_ * NPIX;                         \
        const int co0_ = n0 + m * 32 + gq_ * 8;                                         \
        const int prow_ = px_ / TW, pcol_ = px_ - prow_ * TW;                           \
        const long opix_ = (long)(ty * TH + prow_) * p.Wout + tx * TW + pcol_;
#define PT_P8_FINISH(PRE)                                                               \
    _Pragma("unroll") for (int i = 0; i < NI8; ++i) {                                   \
        PT_P8_ITEM(i)                                                                   \
        float v[8];                                                                     \
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(stage + px_ * 36 + gq_ * 8);   \
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(stage + px_ * 36 + gq_ * 8 + 4); \
        _Pragma("unroll") for (int c = 0; c < 8; ++c) {                                 \
            float t = c < 4 ? s0[c] : s1[c - 4];                                        \
            t += (PRE);                                                                 \
            t += bias_all[co0_ + c];                                                    \
            if (p.act == CCVS_ACT_LRELU) t = lrelu01(t);                                \
            v[c] = t * p.out_scale;                                                     \
        }                                                                               \
        uint4 hi, lo;                                                                   \
        split8(v, hi, lo);                                                              \
        uint4* dst = y4 + ((long)n * gout + (co0_ >> 3)) * 2 * hw_out + opix_;          \
        cb_store16(dst, hi);                                                            \
        cb_store16(dst + hw_out, lo);                                                   \
    }
        if (p.pre) {
            float pv[NI8][8];
            const float* pb = p.pre + (long)(n / p.pre_div) * p.pre_sN;
#pragma unroll
            for (int i = 0; i < NI8; ++i) {
                PT_P8_ITEM(i)
#pragma unroll
                for (int c = 0; c < 8; ++c) pv[i][c] = pb[(long)(co0_ + c) * p.pre_sC + opix_];
            }
            PT_P8_FINISH(pv[i][c])
        } else {
            PT_P8_FINISH(0.f)
        }
#undef PT_P8_ITEM
#undef PT_P8_FINISH
    };
#define PT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")   /* LDS only: global stores / loads stay in flight */

    int lin = (int)blockIdx.x;   // the tile this role computes
    if (producer) {
        // =================================== staging role ===================================
        // compute cursor (lin, cc) and, ahead of it, the request cursor (llin, lc) with the address state of ITS tile
        int llin = lin, lc = 0;
        bool lvalid = llin < total;
        int ln = 0, liy0 = 0, lix0 = 0;   // image, halo origin of the request cursor's tile
        auto req_tile = [&]() __attribute__((always_inline)) {
            int bx, by, bz;
            pt_tile_coords(p, llin, bx, by, bz);
            const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
            ln = bz; liy0 = ty * TH - 1; lix0 = tx * TW - 1;
        };
        auto req_advance = [&]() __attribute__((always_inline)) {   // the chunk after (llin, lc); at a tile's end: the next tile of this workgroup
            if (++lc == nchunks) {
                lc = 0;
                llin += stride_t;
                lvalid = llin < total;
                return true;
            }
            return false;
        };
        // epilogue of the tile at the compute cursor (the staging waves only store)
        auto epilogue = [&]() __attribute__((always_inline)) {
            int bx, by, bz;
            pt_tile_coords(p, lin, bx, by, bz);
            const int ty = bx / p.tiles_x, tx = bx - ty * p.tiles_x;
            const int n0 = by * NT, n = bz;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                PT_LDS_BARRIER();
                if (p.out_p8) epi_store_p8(m, tx, ty, n, n0);
                else epi_store_f32(m, tx, ty, n, n0);
                if (m + 1 < MB) PT_LDS_BARRIER();
            }
        };
        if constexpr (P8IN) {
            // slot j of this thread: 16-byte slot rt + 256 j of the 4-plane halo image -> (plane q, row, column), the same for every tile
            int sq[NJ], srow[NJ], scol[NJ], p8off[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int s_ = rt + 256 * j;
                sq[j] = s_ < 4 * plane ? s_ / plane : -1;
                const int e = s_ - max(sq[j], 0) * plane;
                srow[j] = e / IW;
                scol[j] = e - srow[j] * IW;
                p8off[j] = -1;
            }
            const int gin = (p.Cin + 7) >> 3;
            const long hw_in = (long)p.Hin * p.Win;
            auto setup = [&]() __attribute__((always_inline)) {
                req_tile();
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int gy = liy0 + srow[j], gx = lix0 + scol[j];
                    p8off[j] = (gy >= 0 && gy < p.Hin && gx >= 0 && gx < p.Win) ? gy * p.Win + gx : -1;
                }
            };
            auto dma_x = [&](int c_, uint4* dst) __attribute__((always_inline)) {
                const uint4* xb = reinterpret_cast<const uint4*>(p.x) + (long)ln * gin * 2 * hw_in;  // uniform
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (sq[j] >= 0) {
                        const int g_ = 2 * c_ + (sq[j] >> 1);
                        const uint4* src = (p8off[j] >= 0 && g_ < gin) ? xb + ((long)g_ * 2 + (sq[j] & 1)) * hw_in + p8off[j] : &g_conv_zero16;
                        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + rw * 64 + 256 * j), 16, 0, 0);
                    }
                }
            };
            // request cursor = the chunk AFTER the compute cursor's: it is requested at the first step of the chunk before it
            if (lvalid) {
                setup();
                dma_x(0, slot2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                req_advance();
            }
            __builtin_amdgcn_s_barrier();   // B0 of the first tile
            while (lin < total) {
                for (int cc = 0; cc < nchunks; ++cc) {
                    if (lvalid) {
                        if (lc == 0) setup();   // (the tile after this one: its first chunk goes to slot 2, free since this tile's chunk 0 was consumed)
                        dma_x(lc, smem4 + (lc == 0 ? 2 * in_sz + 2 * w_sz : (lc & 1) * in_sz));
                        req_advance();
                    }
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                epilogue();
                lin += stride_t;
                if (lin >= total) break;
                PT_LDS_BARRIER();   // B0: the stage (over halo buffers 0 / 1) is free again
            }
        } else {
            // item j of this thread: half vh (8 channels), halo row vr, float4 column vq -- the same for every tile
            bool vitem[NI];
            int ve[NI], vh[NI], vr[NI], vq[NI];
            constexpr int NQP = (NQ + 1) >> 1, per_half = NQP * IH * 2, n_items = per_half * 2;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int it0_ = rt + 256 * j;
                const int it = min(it0_, n_items - 1);
                vh[j] = it / per_half;
                const int rem = it - vh[j] * per_half;
                const int t = rem >> 1;
                const int qh = t / IH;
                vr[j] = t - qh * IH;
                vq[j] = 2 * qh + (rem & 1);
                vitem[j] = it0_ < n_items && vq[j] < NQ;
                vq[j] = min(vq[j], NQ - 1);
                ve[j] = vr[j] * IWS + 4 * vq[j];
            }
            const float* vptr[NI];
            bool vin[NI];
            auto setup = [&]() __attribute__((always_inline)) {
                req_tile();
                const float* xn = p.x + (long)ln * p.in_sN;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int gy = liy0 + vr[j], gxa = lix0 - XSH + 4 * vq[j];
                    vin[j] = gy >= 0 && gy < p.Hin && gxa >= 0 && gxa < p.Win;  // aligned and Win % 4 == 0: all 4 pixels in or out
                    vptr[j] = xn + (long)(8 * vh[j]) * in_sC + (vin[j] ? gy * p.Win + gxa : 0);
                }
            };
            f32x4 xv[2][NI][8];
            int xvc0[2] = {-1, -1};      // first channel of the chunk a set holds; -1: nothing
            bool xvin[2][NI];
            int xdst[2] = {0, 0};        // LDS destination of a set's chunk (uint4 units from smem4: an offset, not a pointer -- a run-time selected pointer may lose its address space)
#pragma unroll
            for (int j = 0; j < NI; ++j) { xvin[0][j] = false; xvin[1][j] = false; vin[j] = false; vptr[j] = p.x; }
            auto load_xv = [&](int set) __attribute__((always_inline)) {   // the request cursor's chunk into `set` (a compile-time constant at every call site), then advance
                if (!lvalid) { xvc0[set] = -1; return; }
                if (lc == 0) setup();
                xvc0[set] = lc * CB_CC;
                xdst[set] = lc == 0 ? 2 * in_sz + 2 * w_sz : (lc & 1) * in_sz;
                const bool full = xvc0[set] + CB_CC <= cin_;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    xvin[set][j] = vin[j];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int c = full ? xvc0[set] + i : min(xvc0[set] + 8 * vh[j] + i, cin_ - 1) - 8 * vh[j];
                        xv[set][j][i] = *reinterpret_cast<const f32x4*>(vptr[j] + (long)c * in_sC);
                    }
                }
                req_advance();
            };
            auto store_xv = [&](int set) __attribute__((always_inline)) {
                if (xvc0[set] < 0) return;
                uint4* dst = smem4 + xdst[set];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (!vitem[j]) continue;
                    const bool plain = xvin[set][j] && (xvc0[set] + CB_CC <= cin_);
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        float v[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = xv[set][j][i][px];
                        if (!plain) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) v[i] = (xvin[set][j] && xvc0[set] + 8 * vh[j] + i < cin_) ? v[i] : 0.f;
                        }
                        uint4 hi, lo;
                        split8(v, hi, lo);
                        dst[(vh[j] * 2 + 0) * plane + ve[j] + px] = hi;
                        dst[(vh[j] * 2 + 1) * plane + ve[j] + px] = lo;
                    }
                }
            };
            // Stream chunk k of this workgroup (the chunks of its tiles one after the other) lives in register set k & 1; when the MFMA
            // waves start computing chunk k the staging waves convert + store chunk k + 1 and request chunk k + 3 into the set that
            // became free.  The tile loop below is unrolled by two chunks so that the sets stay statically indexed; a tile may end
            // after either half.
            load_xv(0);          // stream chunk 0 = chunk 0 of the first tile: synchronously into slot 2
            store_xv(0);
            load_xv(1);          // stream chunks 1 and 2
            load_xv(0);
            __syncthreads();     // B0 of the first tile
            int cc = 0;          // chunk of the compute cursor inside its tile
            bool more = lin < total;
#define PT_STAGE_CHUNK(SET)                                                                                   \
            {                                                                                                 \
                store_xv(SET);                                                                                \
                load_xv(SET);                                                                                 \
                const int nb_ = (ktail && cc == nchunks - 1) ? 1 : 3;                                         \
                for (int a = 0; a < nb_; ++a) PT_LDS_BARRIER();                                               \
                if (++cc == nchunks) {                                                                        \
                    cc = 0;                                                                                   \
                    epilogue();                                                                               \
                    lin += stride_t;                                                                          \
                    more = lin < total;                                                                       \
                    if (more) PT_LDS_BARRIER();                                                               \
                }                                                                                             \
            }
            while (more) {
                PT_STAGE_CHUNK(1)
                if (!more) break;
                PT_STAGE_CHUNK(0)
            }
#undef PT_STAGE_CHUNK
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        // ==================================== MFMA role ====================================
        int wofs[CB_WR];  // lane part of the weight address (uint4 units): tap column, half, hi|lo, cout
#pragma unroll
        for (int i = 0; i < CB_WR; ++i) {
            const int ic = min(rt + 256 * i, wunits - 1);
            const int b = ic / (4 * NT), rem = ic - b * (4 * NT);
            const int hp = rem / NT, co = rem - hp * NT;
            wofs[i] = ((b * CinG + (hp >> 1)) * 2 + (hp & 1)) * cout_pad + co;
        }
        // LDS-DMA of one tap row of weights (chunk ci_, tap row a_, channel block at n0_): wave-instruction i of wave rw fills 64 consecutive uint4
        auto dma_w = [&](int n0_, int ci_, int a_, uint4* dst) __attribute__((always_inline)) {
            const uint4* base = wsplit + ((((long)a_ * 3) * CinG + ci_ * 2) * 2) * cout_pad + n0_;  // uniform
#pragma unroll
            for (int i = 0; i < CB_WR; ++i)
                if (rt + 256 * i < wunits)
                    __builtin_amdgcn_global_load_lds((glb_void*)(base + wofs[i]), (lds_void*)(dst + rw * 64 + 256 * i), 16, 0, 0);
        };
        int bofs[PP];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) bofs[pp] = (rw * PP + pp) * IWS + (lane & 31) + XSH;
        const int khalf = lane >> 5;
        const int nsteps = ktail ? (nchunks - 1) * 3 + 1 : nchunks * 3;
        const int nloop = ktail ? nsteps - 1 : nsteps;
        int wp = 0;   // weight buffer of the step about to be computed (runs on across tiles)
        int tx = 0, ty = 0, n = 0, n0 = 0;
        auto tile_of = [&](int l_, int& tx_, int& ty_, int& n_, int& n0_) __attribute__((always_inline)) {
            int bx, by, bz;
            pt_tile_coords(p, l_, bx, by, bz);
            ty_ = bx / p.tiles_x; tx_ = bx - ty_ * p.tiles_x;
            n_ = bz; n0_ = by * NT;
        };
        if (lin < total) {
            tile_of(lin, tx, ty, n, n0);
            dma_w(n0, 0, 0, w_buf);
        }
        __syncthreads();   // B0 of the first tile (drains the DMA)
        while (lin < total) {
            const int lnext = lin + stride_t;
            const bool has_next = lnext < total;
            int txn = 0, tyn = 0, nn = 0, n0n = 0;
            if (has_next) tile_of(lnext, txn, tyn, nn, n0n);
            f32x16 acc[MB][PP];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][pp][r] = 0.f;
            int ci = 0, a = 0;
            for (int s = 0; s < nloop; ++s) {
                // The weights of step s + 1 -- or of the next tile's first step, or (last step of the last tile) once more this tile's first row,
                // never read: an UNCONDITIONAL request with selected scalars, so that it can sit between the taps (PT_MID_DMA: address arithmetic and
                // issue in the shadow of the first tap's matrix instructions instead of in front of the step, where the matrix pipe idled 360-450
                // cycles, profiles/r05_conv_step_stamps.txt) without a branch between the fragment reads and the matrix instructions
                int a1 = a + 1, c1 = ci, n0w = n0;
                if (a1 == 3) { a1 = 0; ++c1; }
                if (s + 1 >= nsteps) { a1 = 0; c1 = 0; n0w = has_next ? n0n : n0; }
                uint4* const wdst = w_buf + (wp ^ 1) * w_sz;
                if (!PT_DMA_MID) dma_w(n0w, c1, a1, wdst);
#define PT_MID_DMA()                                      \
    if (PT_DMA_MID) {                                     \
        __builtin_amdgcn_sched_barrier(0);                \
        dma_w(n0w, c1, a1, wdst);                         \
        __builtin_amdgcn_sched_barrier(0);                \
    }
                {
                    const uint4* it0 = smem4 + (ci == 0 ? 2 * in_sz + 2 * w_sz : (ci & 1) * in_sz) + (khalf * 2) * plane + a * IWS;
                    const uint4* wt0 = w_buf + wp * w_sz + (khalf * 2) * NT + (lane & 31);
                    if constexpr (PP == 4) {
                        // four pixel blocks of a tap in registers, the weight fragments of the tap's MB channel blocks in one of two sets;
                        // each pixel block is re-read for the next tap right after its last MFMA (conv2d_bf16_kernels.h, CB_TAP4)
                        bf16x8 fq[4][2], fw[2][MB][2];
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            fq[pp][0] = __builtin_bit_cast(bf16x8, it0[bofs[pp]]);
                            fq[pp][1] = __builtin_bit_cast(bf16x8, it0[plane + bofs[pp]]);
                        }
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            fw[0][m][0] = __builtin_bit_cast(bf16x8, wt0[m * 32]);
                            fw[0][m][1] = __builtin_bit_cast(bf16x8, wt0[NT + m * 32]);
                        }
#define PT_TAP4(CUR, tb, more_)                                                                                        \
    {                                                                                                                  \
        const uint4* itn_ = it0 + ((tb) + 1);                                                                          \
        if (more_) {                                                                                                   \
            const uint4* wtn_ = wt0 + ((tb) + 1) * 4 * NT;                                                             \
            _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                           \
                fw[(CUR) ^ 1][m][0] = __builtin_bit_cast(bf16x8, wtn_[m * 32]);                                        \
                fw[(CUR) ^ 1][m][1] = __builtin_bit_cast(bf16x8, wtn_[NT + m * 32]);                                   \
            }                                                                                                          \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                                               \
            _Pragma("unroll") for (int pp = 0; pp < 4; ++pp) {                                                         \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][1], fq[pp][0], acc[m][pp], 0, 0, 0);   \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][0], fq[pp][1], acc[m][pp], 0, 0, 0);   \
                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[CUR][m][0], fq[pp][0], acc[m][pp], 0, 0, 0);   \
                if (m == MB - 1 && (more_)) {                                                                          \
                    fq[pp][0] = __builtin_bit_cast(bf16x8, itn_[bofs[pp]]);                                            \
                    fq[pp][1] = __builtin_bit_cast(bf16x8, itn_[plane + bofs[pp]]);                                    \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
                        PT_TAP4(0, 0, true)
                        PT_MID_DMA()
                        PT_TAP4(1, 1, true)
                        PT_TAP4(0, 2, false)
#undef PT_TAP4
                    } else {
                        // fragment reads software-pipelined by hand: the next block's ds_reads above the current block's 6 MFMAs
                        // (conv2d_bf16_kernels.h, CB_TAP)
                        bf16x8 fa[2][2];     // [set][0 hi | 1 lo]       weights of one 32-cout block
                        bf16x8 fb[2][2][2];  // [set][pp][0 hi | 1 lo]   the two pixel blocks of one tap
#define PT_LD_B(SET, tb)                                                                        \
    {                                                                                           \
        const uint4* it_ = it0 + (tb);                                                          \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                      \
            fb[SET][pp][0] = __builtin_bit_cast(bf16x8, it_[bofs[pp]]);                         \
            fb[SET][pp][1] = __builtin_bit_cast(bf16x8, it_[plane + bofs[pp]]);                 \
        }                                                                                       \
    }
#define PT_LD_A(SET, tb, m_)                                                                    \
    {                                                                                           \
        const uint4* wt_ = wt0 + (tb) * 4 * NT + (m_) * 32;                                     \
        fa[SET][0] = __builtin_bit_cast(bf16x8, wt_[0]);                                        \
        fa[SET][1] = __builtin_bit_cast(bf16x8, wt_[NT]);                                       \
    }
#define PT_TAP(BSET, A0, tb, has_next_)                                                         \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) {                                            \
        if (m + 1 < MB) {                                                                       \
            PT_LD_A(((A0) + m + 1) & 1, tb, m + 1)                                              \
        } else if (has_next_) {                                                                 \
            PT_LD_A(((A0) + m + 1) & 1, (tb) + 1, 0)                                            \
            PT_LD_B((BSET) ^ 1, (tb) + 1)                                                       \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        _Pragma("unroll") for (int pp = 0; pp < 2; ++pp) {                                      \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][1], fb[BSET][pp][0], acc[m][pp], 0, 0, 0); \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][0], fb[BSET][pp][1], acc[m][pp], 0, 0, 0); \
            acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[((A0) + m) & 1][0], fb[BSET][pp][0], acc[m][pp], 0, 0, 0); \
        }                                                                                       \
    }
                        PT_LD_B(0, 0)
                        PT_LD_A(0, 0, 0)
                        PT_TAP(0, 0, 0, true)
                        PT_MID_DMA()
                        PT_TAP(1, (MB & 1), 1, true)
                        PT_TAP(0, 0, 2, false)
#undef PT_LD_B
#undef PT_LD_A
#undef PT_TAP
                    }
                }
#undef PT_MID_DMA
                __syncthreads();
                wp ^= 1;
                if (++a == 3) { a = 0; ++ci; }
            }
            if constexpr (!P8IN) {
                if (ktail) {
                    // packed K tail (ccvs_conv_desc.w_ktail; conv2d_bf16_kernels.h): the last chunk's nine taps x r channels contracted in
                    // ceil(9 r / 16) steps whose K index runs over (tap, channel); the pixel operand is gathered from the staged tile
                    if (has_next) dma_w(n0n, 0, 0, w_buf + (wp ^ 1) * w_sz);
                    const int r_ = p.ktail, nq_ = 9 * r_, nj_ = (nq_ + 15) >> 4;
                    const unsigned short* ih = reinterpret_cast<const unsigned short*>(in_buf + ((nchunks - 1) & 1) * in_sz);
                    const uint4* wtl = w_buf + wp * w_sz + (khalf * 2) * NT + (lane & 31);
                    for (int j = 0; j < nj_; ++j) {
                        bf16x8 gb[PP][2];
#pragma unroll
                        for (int pp = 0; pp < PP; ++pp) {
                            unsigned hw[4], lw[4];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int q = 16 * j + 8 * khalf + i;
                                const int qc = min(q, nq_ - 1);
                                const int t = qc / r_, c = qc - t * r_;
                                const int tyy = t / 3, txx = t - 3 * tyy;
                                const int e = (bofs[pp] + tyy * IWS + txx) * 8 + c;   // bf16 index inside the [pixel][8] plane
                                unsigned hv = ih[e], lv = ih[plane * 8 + e];
                                if (q >= nq_) { hv = 0; lv = 0; }
                                if (i & 1) { hw[i >> 1] |= hv << 16; lw[i >> 1] |= lv << 16; }
                                else { hw[i >> 1] = hv; lw[i >> 1] = lv; }
                            }
                            gb[pp][0] = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
                            gb[pp][1] = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
                        }
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, wtl[j * 4 * NT + m * 32]);
                            const bf16x8 al = __builtin_bit_cast(bf16x8, wtl[j * 4 * NT + NT + m * 32]);
#pragma unroll
                            for (int pp = 0; pp < PP; ++pp) {
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, gb[pp][0], acc[m][pp], 0, 0, 0);
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gb[pp][1], acc[m][pp], 0, 0, 0);
                                acc[m][pp] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, gb[pp][0], acc[m][pp], 0, 0, 0);
                            }
                        }
                    }
                    __syncthreads();
                    wp ^= 1;
                }
            }
            // ---- epilogue: the accumulators go through LDS so that all 8 waves write 16-byte pieces along x ----
            int lane_ = lane, rw_ = rw;   // (opaque: the stage addresses are tile-invariant too)
            asm volatile("" : "+v"(lane_), "+v"(rw_));
            const int khalf_ = lane_ >> 5;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (p.out_p8) {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) {
                        float* sp = stage + ((rw_ * PP + pp) * 32 + (lane_ & 31)) * 36 + 4 * khalf_;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 q4 = {acc[m][pp][4 * g], acc[m][pp][4 * g + 1], acc[m][pp][4 * g + 2], acc[m][pp][4 * g + 3]};
                            *reinterpret_cast<f32x4*>(sp + 8 * g) = q4;
                        }
                    }
                } else {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stage[((r & 3) + 8 * (r >> 2) + 4 * khalf_) * NPIX + (rw_ * PP + pp) * 32 + (lane_ & 31)] = acc[m][pp][r];
                }
                PT_LDS_BARRIER();
                if (p.out_p8) epi_store_p8(m, tx, ty, n, n0);
                else epi_store_f32(m, tx, ty, n, n0);
                if (m + 1 < MB) PT_LDS_BARRIER();
            }
            lin = lnext;
            if (!has_next) break;
            tx = txn; ty = tyn; n = nn; n0 = n0n;
            PT_LDS_BARRIER();   // B0 of the next tile
        }
    }
#undef PT_LDS_BARRIER
}

// Can this launch run as persistent tiles?  (k: the launch as launch_conv_bf16 sees it, tiles_x / tiles_y of the 256-pixel tiling.)
template <int MB>
static bool conv_pt_ok(const ConvK& k, int pp, bool kt_possible) {
    const int th = 4 * pp, nt = 32 * MB;
    if (!(k.kh == 3 && k.kw == 3 && k.stride == 1 && k.pad == 1 && !k.transposed)) return false;
    if (k.cu_limit > 0 || k.Wout % 32 != 0 || k.Hout % th != 0 || k.Hout != k.Hin || k.Wout != k.Win) return false;
    if (k.Cout != k.CoutPad || k.Cout % nt != 0 || k.CoutPad > 512) return false;
    if ((k.Cin + CB_CC - 1) / CB_CC < 3) return false;
    const int n_add = (k.pre ? 1 : 0) + (k.res ? 1 : 0) + (k.accumulate ? 1 : 0);
    if (n_add > 1) return false;
    if ((reinterpret_cast<uintptr_t>(k.y) & 15) != 0) return false;
    if (k.out_p8) {
        if (k.res || k.accumulate) return false;
    } else {
        if ((k.out_sN & 3) != 0 || (k.out_sC & 3) != 0) return false;
        if (k.pre && ((reinterpret_cast<uintptr_t>(k.pre) & 15) != 0 || (k.pre_sN & 3) != 0 || (k.pre_sC & 3) != 0)) return false;
        if (k.res && ((reinterpret_cast<uintptr_t>(k.res) & 15) != 0 || (k.res_sN & 3) != 0 || (k.res_sC & 3) != 0)) return false;
    }
    if (!k.in_p8) {
        if (k.Win % 4 != 0 || k.in_sC % 4 != 0 || k.in_sN % 4 != 0 || (reinterpret_cast<uintptr_t>(k.x) & 15) != 0) return false;
    }
    (void)kt_possible;
    return true;
}

static int conv_pt_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
        if (getenv("CCVS_CONV_PT_CUS") && atoi(getenv("CCVS_CONV_PT_CUS")) > 0) n_cu = atoi(getenv("CCVS_CONV_PT_CUS"));   // experiment: resident workgroups on fewer CUs than the chip has
        n_cu -= n_cu % 8;   // whole rounds over the XCDs: workgroup g and its tiles g + gridDim i stay on one XCD's eighth of the tile order
        if (n_cu < 8) n_cu = 8;
    }
    return n_cu;
}

// launch as persistent tiles; k.ktail / the weight pointer as for the producer / consumer kernel
template <int MB, int PP, bool P8IN>
static int launch_conv_pt(ConvK k, const void* w, int CinG, int gz, hipStream_t st) {
    constexpr int NT = 32 * MB, TH = 4 * PP, IH = TH + 2, IWS = P8IN ? 34 : 41, plane = IH * IWS;
    constexpr size_t smem = (size_t)(3 * 4 * plane + 2 * 3 * 4 * NT) * 16;
    static_assert(smem <= 156 * 1024, "persistent tiles: LDS");
    static bool attr_set = false;
    if (!attr_set) {
        CB_SET_LDS((conv2d_bf16x3_pt_kernel<MB, PP, P8IN>), (int)smem);   // (+ 2 KB static: the layer's bias values)
        attr_set = true;
    }
    k.tiles_x = k.Wout / 32;
    k.tiles_y = k.Hout / TH;
    k.gx = k.tiles_x * k.tiles_y;
    k.gy = k.CoutPad / NT;
    const long total = (long)k.gx * k.gy * gz;
    k.nwork = (int)total; k.work0 = 0;
    static const int xcd_aware = getenv("CCVS_CONV_XCD") ? atoi(getenv("CCVS_CONV_XCD")) : 1;
    k.xcd_chunk = (xcd_aware && total % 8 == 0 && total >= 64) ? (int)(total / 8) : 0;
    const int cus = conv_pt_cus();
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    hipLaunchKernelGGL((conv2d_bf16x3_pt_kernel<MB, PP, P8IN>), dim3(grid), dim3(512), smem, st, k, (const uint4*)w, CinG, (int)total);
    CCVS_CHECK_LAUNCH("ccvs_conv2d_bf16x3 (persistent tiles)");
    return CCVS_OK;
}

// the hook of launch_conv_bf16<32, MB>: CONV_PT_NOT_TAKEN = not a launch for persistent tiles
template <int MB>
static int conv_pt_try(const ConvK& k_in, const void* wsplit, const void* wktail, int CinG, int gz, hipStream_t st, int pt_on) {
    ConvK k = k_in;
    k.ktail = 0;
    const int cus = conv_pt_cus();
    if (k.in_p8) {
        if (!(pt_on & 2)) return CONV_PT_NOT_TAKEN;
        if constexpr (MB == 2) {
            if (k.out_p8 && conv_pt_ok<2>(k, 4, false) && (long)(k.Wout / 32) * (k.Hout / 16) * (k.CoutPad / 64) * gz >= 2L * cus)
                return launch_conv_pt<2, 4, true>(k, wsplit, CinG, gz, st);
        }
        if constexpr (MB == 4) {
            if (conv_pt_ok<4>(k, 2, false) && (long)(k.Wout / 32) * (k.Hout / 8) * (k.CoutPad / 128) * gz >= 2L * cus)
                return launch_conv_pt<4, 2, true>(k, wsplit, CinG, gz, st);
        }
        return CONV_PT_NOT_TAKEN;
    }
    if constexpr (MB == 4) {
        if (!(pt_on & 1)) return CONV_PT_NOT_TAKEN;
        if (conv_pt_ok<4>(k, 2, true) && (long)(k.Wout / 32) * (k.Hout / 8) * (k.CoutPad / 128) * gz >= 2L * cus) {
            static const int ktail_on = getenv("CCVS_CONV_KTAIL") ? atoi(getenv("CCVS_CONV_KTAIL")) : 1;
            const int ktail_r = k.Cin % CB_CC;
            const bool kt = ktail_on && wktail && ktail_r >= 1 && ktail_r <= 3 && k.Cin > CB_CC;
            k.ktail = kt ? ktail_r : 0;
            return launch_conv_pt<4, 2, false>(k, kt ? wktail : wsplit, CinG, gz, st);
        }
    }
    return CONV_PT_NOT_TAKEN;
}
