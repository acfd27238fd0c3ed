// train_x3.hip - the training-mode FeedForward and the token-contraction weight gradient on the f16 matrix pipe with
// 3-term split products (common.hip.h "x3": hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_f16, fp32 accumulation, ~2^-21
// per product) instead of v_mfma_f32_16x16x4_f32, which runs at 1/16 of that pipe's rate.  Same mathematics, same
// C ABI, same workspace layout as the fp32 kernels of train.hip (kept there; -DTRAIN_X3=0 selects them).
//
//   forward    y = res + 0.5 m2 (W2 (m1 Swish(W1 LN(x) + b1)) + b2)          conformer.py:54-72,136-148,211
//   backward   dz = 0.5 m2 dy;  dd1 = W2^T dz;  dh = m1 dd1 Swish'(h);  dxn = W1^T dh;  LayerNorm backward
//   wgrad      dW[i][j] = sum_m P[m][i] Q[m][j]  (token contraction, split-K partial slabs)
//
// Structure follows the inference kernel ffn_x3_kernel (conformer_x3.hip): persistent 768-thread blocks with the
// weight images resident in LDS (the fp32 training kernels streamed 8 KB of weight fragments per token from L1 / L2),
// two 16-token blocks per wave, the per-token chain in registers.  The backward is two kernels because its three
// weight images (W1, W2^T, W1^T: 192 KB) do not fit one CU's LDS: A holds W1 and W2^T (h recompute, dd1, d1 / dh
// stores), B holds W1^T (dxn, LayerNorm backward); dh [M,256] is written by A anyway (the W1 gradient contracts it).
#include "kernels.h"
#include "train.h"
#include <map>
#include <type_traits>
#include <mutex>

#define TX_WAVES 12

// row-major W [R, K] (leading dimension ldw; transpose = 1: the image of W^T) -> x3 A-operand image
// [R/16][K/32][hi | lo][64 lanes][8 halfs]: lane (c, g) slot e <-> W[16 rb + c][32 m + 16 (e >> 2) + 4 g + (e & 3)]
struct PackX3Job { const float* w; int R, K, ldw, transpose; _Float16* out; };
struct PackX3Jobs { PackX3Job j[4]; };
__global__ void pack_x3_kernel(PackX3Jobs jobs) {
    const PackX3Job& q = jobs.j[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (rb, m, lane)
    const int M32 = q.K / 32;
    if (i >= (q.R / 16) * M32 * 64) return;
    const int lane = i & 63, blk = i >> 6, rb = blk / M32, m = blk - rb * M32;
    const int row = 16 * rb + (lane & 15);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = 32 * m + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        const float v = q.transpose ? q.w[(long)col * q.ldw + row] : q.w[(long)row * q.ldw + col];
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
    _Float16* o = q.out + (long)blk * 1024 + lane * 8;
    *reinterpret_cast<f16x8*>(o) = hi;
    *reinterpret_cast<f16x8*>(o + 512) = lo;
}
void launch_pack_x3(LaunchCtx ctx, const char* label, const PackX3Jobs& jobs, int njobs) {
    int most = 0;
    for (int k = 0; k < njobs; ++k) {
        const int n = (jobs.j[k].R / 16) * (jobs.j[k].K / 32) * 64;
        most = n > most ? n : most;
    }
    LAUNCH(ctx, label, (pack_x3_kernel<<<dim3((most + 255) / 256, njobs), 256, 0, ctx.stream>>>(jobs)));
}

__device__ __forceinline__ unsigned tx_mask_word(const unsigned char* __restrict__ m, long idx) {   // idx % 4 == 0
    return *reinterpret_cast<const unsigned*>(m + idx);
}
__device__ __forceinline__ f32x4 tx_mask4(unsigned v, float ms) {
    f32x4 r;
    r[0] = (v & 0x000000ffu) ? ms : 0.f;
    r[1] = (v & 0x0000ff00u) ? ms : 0.f;
    r[2] = (v & 0x00ff0000u) ? ms : 0.f;
    r[3] = (v & 0xff000000u) ? ms : 0.f;
    return r;
}

// ---- exact power-of-two operand scaling ------------------------------------------------------------------------------
// The fp16 split represents magnitudes between ~2^-24 and 65504.  Activations of this network live there; GRADIENTS
// do not (dL/dy of a mean loss is ~1 / numel): their lo halves would flush to zero and a product would keep 11 bits.
// Every gradient operand is therefore multiplied by an exact power of two that brings the largest magnitude of its
// tile (per-token kernels) or of the running contraction (weight gradient) to [1, 2), and the result is multiplied by
// the inverse - both exact, so the arithmetic is the split product of the unscaled values with unlimited range.
__device__ __forceinline__ float tx_max_c(float v) {            // max over the 16 lanes of a DPP row
    v = fmaxf(v, dpp_perm<0xB1>(v));
    v = fmaxf(v, dpp_perm<0x4E>(v));
    v = fmaxf(v, dpp_perm<0x141>(v));
    v = fmaxf(v, dpp_perm<0x140>(v));
    return v;
}
__device__ __forceinline__ float tx_wave_max(float v) { return red_g_max(tx_max_c(v)); }
// s = 2^-e with e the binary exponent of m (m s in [1, 2)); inv = 2^e.  m = 0 (or denormal): both 1.
__device__ __forceinline__ void tx_pow2(float m, float& s, float& inv) {
    const unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
    const bool ok = e > 0u && e < 254u;
    s = ok ? __uint_as_float((254u - e) << 23) : 1.0f;
    inv = ok ? __uint_as_float(e << 23) : 1.0f;
}
__device__ __forceinline__ float tx_absmax4(const f32x4& v, float m) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// LayerNorm of a 16-token block's rows: xh = (x - mean) rstd (returned for the backward), xn = xh gamma + beta as
// split B operands of a K = 64 product (two k32 blocks)
__device__ __forceinline__ void tx_load_norm(const float* __restrict__ x, long row, int g, const float* par_l,
                                             f32x4 (&xh)[4], float& rstd, f32x4 (&xn)[4]) {
    f32x4 xv[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) xv[kb] = ldg4(x + row * 64 + 16 * kb + 4 * g);
    float mean;
    ln_stats(xv, mean, rstd);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        xh[kb] = (xv[kb] - splat4(mean)) * splat4(rstd);
        xn[kb] = xh[kb] * *reinterpret_cast<const f32x4*>(&par_l[16 * kb + 4 * g]) +
                 *reinterpret_cast<const f32x4*>(&par_l[64 + 16 * kb + 4 * g]);
    }
}

// ---------------------------------------------------------------------------------
// forward.  LDS: W1 image [16][2] + W2 image [4][8] = 128 KB, gamma | beta | b1 | b2.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(TX_WAVES * 64) void ffn_train_fwd_x3_kernel(const float* __restrict__ x, long M,
                                                                         const _Float16* __restrict__ w1i,
                                                                         const _Float16* __restrict__ w2i,
                                                                         FfnTrainParams p,
                                                                         const unsigned char* __restrict__ m1,
                                                                         const unsigned char* __restrict__ m2, float ms,
                                                                         const float* __restrict__ res, float* __restrict__ y,
                                                                         int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];          // 128 KB
    __shared__ __attribute__((aligned(16))) float par_l[448];              // gamma[64] | beta[64] | b1[256] | b2[64]
    _Float16* w1 = wlds;
    _Float16* w2 = wlds + 32768;
    stage_lds16<4096, TX_WAVES * 64>(w1i, w1);
    stage_lds16<4096, TX_WAVES * 64>(w2i, w2);
    for (int i = threadIdx.x; i < 448; i += blockDim.x)
        par_l[i] = i < 64 ? p.gamma[i] : (i < 128 ? p.beta[i - 64] : (i < 384 ? p.b1[i - 128] : p.b2[i - 384]));
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;

#pragma unroll 1
    for (int tile = blockIdx.x * TX_WAVES + wv; tile < ntiles; tile += gridDim.x * TX_WAVES) {
        long row[2];
        bool ok[2];
        f16x8 xbh[2][2], xbl[2][2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 xh[4], xn[4];
            float rstd;
            tx_load_norm(x, row[tb], g, par_l, xh, rstd, xn);
            split8(xn[0], xn[1], xbh[tb][0], xbl[tb][0]);
            split8(xn[2], xn[3], xbh[tb][1], xbl[tb][1]);
        }
        f32x4 acc[2][4];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) acc[tb][ob] = *reinterpret_cast<const f32x4*>(&par_l[384 + 16 * ob + 4 * g]);

#pragma unroll 1
        for (int m = 0; m < 8; ++m) {                                  // hidden units 32 m .. 32 m + 31
            unsigned mw[2][2];
            if (m1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) mw[j][tb] = tx_mask_word(m1, row[tb] * 256 + 16 * (2 * m + j) + 4 * g);
            }
            f32x4 h[2][2];                                             // [j][tb]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hb = 2 * m + j;
                const f32x4 b1v = *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * hb + 4 * g]);
                h[j][0] = b1v; h[j][1] = b1v;
                lin_acc_x3<2, 2>(w1 + hb * 2048 + lane * 8, xbh, xbl, h[j]);
            }
            f16x8 sh[2], sl[2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 mk = m1 ? tx_mask4(mw[j][tb], ms) : splat4(1.f);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[j][tb][r] = swishf(h[j][tb][r]) * mk[r];
                }
                split8(h[0][tb], h[1][tb], sh[tb], sl[tb]);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const _Float16* wp = w2 + (ob * 8 + m) * 1024 + lane * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
                const f16x8 al = *reinterpret_cast<const f16x8*>(wp + 512);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) acc[tb][ob] = mfma32h(ah, sh[tb], acc[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) acc[tb][ob] = mfma32l(ah, sl[tb], acc[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) acc[tb][ob] = mfma32l(al, sh[tb], acc[tb][ob]);
            }
        }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            if (ok[tb]) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    f32x4 v = acc[tb][ob] * splat4(0.5f);
                    if (m2) v = v * tx_mask4(tx_mask_word(m2, row[tb] * 64 + 16 * ob + 4 * g), ms);
                    if (res) v = v + ldg4(res + row[tb] * 64 + 16 * ob + 4 * g);
                    stg4(y + row[tb] * 64 + 16 * ob + 4 * g, v);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// backward, part A.  LDS: W1 image [16][2] (h recompute) + W2^T image [16][2] (dd1 = W2^T dz) = 128 KB.
// Writes dz [M,64], xn [M,64], d1 = m1 Swish(h) [M,256], dh = m1 dd1 Swish'(h) [M,256].
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(TX_WAVES * 64) void ffn_train_bwd_a_x3_kernel(const float* __restrict__ x,
                                                                           const float* __restrict__ dy, long M,
                                                                           const _Float16* __restrict__ w1i,
                                                                           const _Float16* __restrict__ w2ti,
                                                                           FfnTrainParams p,
                                                                           const unsigned char* __restrict__ m1,
                                                                           const unsigned char* __restrict__ m2, float ms,
                                                                           float* __restrict__ o_dz, float* __restrict__ o_xn,
                                                                           float* __restrict__ o_d1, float* __restrict__ o_dh,
                                                                           float* __restrict__ o_dhmax, float* __restrict__ o_dzc,
                                                                           int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];
    __shared__ __attribute__((aligned(16))) float par_l[448];
    _Float16* w1 = wlds;
    _Float16* w2t = wlds + 32768;
    stage_lds16<4096, TX_WAVES * 64>(w1i, w1);
    stage_lds16<4096, TX_WAVES * 64>(w2ti, w2t);
    for (int i = threadIdx.x; i < 448; i += blockDim.x)
        par_l[i] = i < 64 ? p.gamma[i] : (i < 128 ? p.beta[i - 64] : (i < 384 ? p.b1[i - 128] : p.b2[i - 384]));
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;

#pragma unroll 1
    for (int tile = blockIdx.x * TX_WAVES + wv; tile < ntiles; tile += gridDim.x * TX_WAVES) {
        long row[2];
        bool ok[2];
        f16x8 xbh[2][2], xbl[2][2], zbh[2][2], zbl[2][2];
        float zinv[2], dhmax = 0.f;                          // inverse of the tile's dz scale; largest |dh| of the tile
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 xh[4], xn[4], dz[4];
            float rstd, zmax = 0.f;
            tx_load_norm(x, row[tb], g, par_l, xh, rstd, xn);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                f32x4 v = ldg4(dy + row[tb] * 64 + 16 * ob + 4 * g) * splat4(0.5f);
                if (m2) v = v * tx_mask4(tx_mask_word(m2, row[tb] * 64 + 16 * ob + 4 * g), ms);
                if (!ok[tb]) v = splat4(0.f);               // padding tokens of the last block contribute nothing
                dz[ob] = v;
                zmax = tx_absmax4(v, zmax);
                if (ok[tb]) {
                    stg4(o_dz + row[tb] * 64 + 16 * ob + 4 * g, v);
                    stg4(o_xn + row[tb] * 64 + 16 * ob + 4 * g, xn[ob]);
                }
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {                 // db2 partial: this 16-token block's column sums, row 2 tile + tb
                f32x4 zs4;                                   // of the [2 ntiles][64] slab the bias gradient is summed from
#pragma unroll
                for (int r = 0; r < 4; ++r) zs4[r] = red_c_sum(dz[ob][r]);
                if (c == 0) stg4(o_dzc + ((long)tile * 2 + tb) * 64 + 16 * ob + 4 * g, zs4);
            }
            float zs;
            tx_pow2(tx_wave_max(zmax), zs, zinv[tb]);       // exact power-of-two scale of this 16-token block's dz
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) dz[ob] = dz[ob] * splat4(zs);
            split8(xn[0], xn[1], xbh[tb][0], xbl[tb][0]);
            split8(xn[2], xn[3], xbh[tb][1], xbl[tb][1]);
            split8(dz[0], dz[1], zbh[tb][0], zbl[tb][0]);
            split8(dz[2], dz[3], zbh[tb][1], zbl[tb][1]);
        }
#pragma unroll 1
        for (int hb = 0; hb < 16; ++hb) {
            unsigned mw[2];
            if (m1) {
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) mw[tb] = tx_mask_word(m1, row[tb] * 256 + 16 * hb + 4 * g);
            }
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * hb + 4 * g]);
            f32x4 h[2] = {b1v, b1v}, dd[2] = {splat4(0.f), splat4(0.f)};
            lin_acc_x3<2, 2>(w1 + hb * 2048 + lane * 8, xbh, xbl, h);
            lin_acc_x3<2, 2>(w2t + hb * 2048 + lane * 8, zbh, zbl, dd);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const f32x4 mk = m1 ? tx_mask4(mw[tb], ms) : splat4(1.f);
                f32x4 d1v, dhv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hv = h[tb][r], sg = sigmoidf_fast(hv);
                    d1v[r] = hv * sg * mk[r];
                    dhv[r] = (dd[tb][r] * zinv[tb]) * mk[r] * (sg * (1.f + hv * (1.f - sg)));
                }
                dhmax = tx_absmax4(dhv, dhmax);
                if (ok[tb]) {
                    stg4(o_d1 + row[tb] * 256 + 16 * hb + 4 * g, d1v);
                    stg4(o_dh + row[tb] * 256 + 16 * hb + 4 * g, dhv);
                }
            }
        }
        dhmax = tx_wave_max(dhmax);                          // part B scales this tile's dh by it
        if (lane == 0) o_dhmax[tile] = dhmax;
    }
}

// ---------------------------------------------------------------------------------
// backward, part B.  LDS: W1^T image [4][8] = 64 KB.  dxn = W1^T dh, then the LayerNorm backward:
//   dx = rstd (g dxn - mean(g dxn) - xh mean(g dxn xh)) + dres;   g1 = dxn xh, dxn -> the column sums of dgamma, dbeta
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void ffn_train_bwd_b_x3_kernel(const float* __restrict__ x, const float* __restrict__ dh,
                                                                 const float* __restrict__ dhmax, long M,
                                                                 const _Float16* __restrict__ w1ti,
                                                                 FfnTrainParams p, const float* __restrict__ dres,
                                                                 float* __restrict__ dx, float* __restrict__ o_g1,
                                                                 float* __restrict__ o_dxn, float* __restrict__ o_dhc,
                                                                 int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[32768];          // 64 KB
    __shared__ __attribute__((aligned(16))) float par_l[128];              // gamma | beta
    stage_lds16<4096, 512>(w1ti, wlds);
    for (int i = threadIdx.x; i < 128; i += blockDim.x) par_l[i] = i < 64 ? p.gamma[i] : p.beta[i - 64];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;

#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {
        long row[2];
        bool ok[2];
        f32x4 dxn[2][4];
        float hs, hinv;
        tx_pow2(dhmax[tile], hs, hinv);                      // the tile's exact power-of-two dh scale (from part A)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) dxn[tb][ob] = splat4(0.f);
        }
#pragma unroll 2
        for (int m = 0; m < 8; ++m) {
            f16x8 bh[2], bl[2];
            f32x4 sa = splat4(0.f), sb = splat4(0.f);          // db1 partial: the tile's column sums of dh -> [ntiles][256]
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                f32x4 a = ldg4(dh + row[tb] * 256 + 32 * m + 4 * g), b = ldg4(dh + row[tb] * 256 + 32 * m + 16 + 4 * g);
                if (!ok[tb]) { a = splat4(0.f); b = splat4(0.f); }
                sa = sa + a;
                sb = sb + b;
                split8(a * splat4(hs), b * splat4(hs), bh[tb], bl[tb]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sa[r] = red_c_sum(sa[r]);
                sb[r] = red_c_sum(sb[r]);
            }
            if (c == 0) {
                stg4(o_dhc + (long)tile * 256 + 32 * m + 4 * g, sa);
                stg4(o_dhc + (long)tile * 256 + 32 * m + 16 + 4 * g, sb);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const _Float16* wp = wlds + (ob * 8 + m) * 1024 + lane * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
                const f16x8 al = *reinterpret_cast<const f16x8*>(wp + 512);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) dxn[tb][ob] = mfma32h(ah, bh[tb], dxn[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) dxn[tb][ob] = mfma32l(ah, bl[tb], dxn[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) dxn[tb][ob] = mfma32l(al, bh[tb], dxn[tb][ob]);
            }
        }
        f32x4 ca[4] = {splat4(0.f), splat4(0.f), splat4(0.f), splat4(0.f)}, cb[4] = {splat4(0.f), splat4(0.f), splat4(0.f), splat4(0.f)};
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            f32x4 xh[4], xn[4], dxh[4];
            float rstd;
            tx_load_norm(x, row[tb], g, par_l, xh, rstd, xn);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) dxn[tb][kb] = dxn[tb][kb] * splat4(hinv);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                dxh[kb] = dxn[tb][kb] * *reinterpret_cast<const f32x4*>(&par_l[16 * kb + 4 * g]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1 += dxh[kb][r];
                    s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
                }
            }
            const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
            if (ok[tb]) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
                    if (dres) dv = dv + ldg4(dres + row[tb] * 64 + 16 * kb + 4 * g);
                    stg4(dx + row[tb] * 64 + 16 * kb + 4 * g, dv);
                    ca[kb] = ca[kb] + dxn[tb][kb] * xh[kb];
                    cb[kb] = cb[kb] + dxn[tb][kb];
                }
            }
        }
        ln_tile_colsums(ca, cb, c, g, tile, o_g1, o_dxn);      // row `tile` of the [ntiles][64] dgamma / dbeta partial slabs
    }
}

// ---------------------------------------------------------------------------------
// backward, part A WITH both weight gradients contracted on the chip (the default; CMGAN_FFN_BWD_FUSED=0 selects part A
// above + two wgrad_partial64_x3_kernel launches).  Part A wrote dz, xn [M,64] and d1, dh [M,256] so that the token
// contraction could read them back: 2.5 KB written and 4 KB re-read per token, and the three kernels ran at the HBM
// roof (9 KB per token and FeedForward: 2.1 ms at 32 clips).  Here only dh leaves the chip (part B needs it: W1^T does
// not fit next to anything).
//   * A block is 8 waves; WAVE w OWNS HIDDEN UNITS 32 w .. 32 w + 31 (two 16-blocks) for the whole launch: its rows of W1
//     and columns of W2 live in registers as split B operands (64 VGPRs, no weight image in LDS), and so do its
//     [32 x 64] blocks of dW1 and dW2 (64 VGPRs), accumulated over all tiles of the block and written once as slab
//     blockIdx.x of the usual [split][R][C] partial layout (reduce_partials_kernel, fixed order).
//   * The products are evaluated token-major: h^T = xn W1^T and dd1^T = dz W2 with the tile's xn / dz rows as A operands,
//     so a lane ends with hidden unit c x tokens 4 g + r of both token blocks - which IS the A operand (rows = hidden
//     units, contraction slots = the tile's 32 tokens, split8 order) of dW2 += d1^T dz and dW1 += dh^T xn.  Nothing is
//     transposed in registers or exchanged between waves.
//   * A tile (32 tokens) is staged ONCE by the block - four tokens per wave, 16 lanes x 4 channels each: LayerNorm,
//     dz = 0.5 m2 dy, split - into four row-major planes (A operands), four transposed planes (B operands of the weight
//     gradients; token slots in split8 order) and the m1 keep-mask as 16 bits per (token, 16 hidden units),
//     double-buffered: one barrier per tile, the next tile's rows are in flight under the 96 MFMAs of the current one.
//   * dz is a gradient: the staged images carry it multiplied by an exact power of two s kept in a band around the
//     running magnitude (the rule of wgrad_partial64_x3_kernel; the 8 per-wave maxima of a tile go through LDS, every
//     wave takes the same decision).  dd1 and dh are then AT SCALE s in registers: well inside fp16 range for the dW1
//     product; dh is stored multiplied by 1 / s, the accumulators are rescaled by the exact ratio when s moves.
// ---------------------------------------------------------------------------------
#define FA_PR 80                      // halfs per row of the [token][channel] planes (160 B: conflict-free ds_read_b128 under the
                                      // instruction's lane grouping {0-3,12-15,20-27} ...; 144 B was 2-way)
#define FA_PT 36                      // halfs per row of the [channel][token slot] planes (72 B: b64 reads, 2-way stores)
struct FaImg {
    _Float16 xnh[32 * FA_PR], xnl[32 * FA_PR], dzh[32 * FA_PR], dzl[32 * FA_PR];
    _Float16 xth[64 * FA_PT], xtl[64 * FA_PT], zth[64 * FA_PT], ztl[64 * FA_PT];
    unsigned short mbits[32 * 16];    // [token][hidden 16-block]: bit u = keep hidden unit 16 blk + u
};
__device__ __forceinline__ unsigned fa_pack4(unsigned w) {        // 4 keep bytes (non-zero = keep) -> 4 bits
    w |= w >> 4; w |= w >> 2; w |= w >> 1;
    w &= 0x01010101u;
    return (w * 0x10204080u) >> 28;                               // byte k -> bit k
}
__device__ __forceinline__ f16x8 fa_ld8(const _Float16* p) {      // 8 halfs at an 8-byte aligned LDS address
    const f16x4 a = *reinterpret_cast<const f16x4*>(p), b = *reinterpret_cast<const f16x4*>(p + 4);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <bool DROP>                   // both dropout keep-masks present (training) or both absent
__global__ __launch_bounds__(512) void ffn_train_bwd_aw_x3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  long M, FfnTrainParams p,
                                                                  const unsigned char* __restrict__ m1,
                                                                  const unsigned char* __restrict__ m2, float ms,
                                                                  float* __restrict__ o_dh, float* __restrict__ o_dhmax,
                                                                  float* __restrict__ o_dzc, float* __restrict__ part_w2,
                                                                  float* __restrict__ part_w1, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fa_sm[];
    FaImg* const img = reinterpret_cast<FaImg*>(fa_sm);                                   // [2]
    float* const zmax_l = reinterpret_cast<float*>(fa_sm + 2 * sizeof(FaImg));            // [2][8]
    unsigned* const dhmax_l = reinterpret_cast<unsigned*>(zmax_l + 16);                   // [2]
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x;
    const int nloc = (ntiles - (int)blockIdx.x + G - 1) / G;       // this block's tiles: blockIdx.x + i G
    if (threadIdx.x < 2) dhmax_l[threadIdx.x] = 0u;

    // resident B operands of hidden unit 32 wv + 16 hb + c: slot (g, e) of k-step ks <-> channel / feature 32 ks + 8 g + e
    f16x8 w1h[2][2], w1l[2][2], w2h[2][2], w2l[2][2];
    float b1c[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int hu = 32 * wv + 16 * hb + c;
        b1c[hb] = p.b1[hu];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wp = p.w1 + (long)hu * 64 + 32 * ks + 8 * g;
            split8(ldg4(wp), ldg4(wp + 4), w1h[hb][ks], w1l[hb][ks]);
            f32x4 a, b;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = p.w2[(long)(32 * ks + 8 * g + e) * 256 + hu];
                b[e] = p.w2[(long)(32 * ks + 8 * g + 4 + e) * 256 + hu];
            }
            split8(a, b, w2h[hb][ks], w2l[hb][ks]);
        }
    }
    // staging role: token tk = 4 wv + g of the tile, channels 4 c .. 4 c + 3 (hidden units 16 c .. 16 c + 15 of the mask)
    const int tk = 4 * wv + g;
    const int pos = 8 * ((tk & 15) >> 2) + 4 * (tk >> 4) + (tk & 3);          // split8 slot order of token tk
    const f32x4 gam = ldg4(p.gamma + 4 * c), bet = ldg4(p.beta + 4 * c);

    struct Raw { f32x4 x, dy; unsigned mk; u32x4 m1w; bool ok; };
    struct Proc { f32x4 xn, dz; unsigned mb; };
    // global accesses as buffer descriptor (whole tensor, SGPRs) + 32-bit lane offset + scalar tile offset: a uniform pointer
    // + lane offset is hoisted out of the loop as 64-bit per-lane pointers - six of them were scratch here.  (The launcher
    // takes this kernel only when the largest tensor, dh, spans < 4 GB.)
    auto rsrc = [](const void* base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t x_rs = rsrc(x, M * 256), dy_rs = rsrc(dy, M * 256), dh_rs = rsrc(o_dh, M * 1024);
    const __amdgpu_buffer_rsrc_t m2_rs = rsrc(DROP ? m2 : reinterpret_cast<const unsigned char*>(x), M * 64),
                                 m1_rs = rsrc(DROP ? m1 : reinterpret_cast<const unsigned char*>(x), M * 256);
    auto load = [&](int i) __attribute__((always_inline)) {
        long tile = (long)blockIdx.x + (long)i * G;
        const bool have = i < nloc;
        tile = have ? tile : ntiles - 1;                          // past the end: a readable dummy, contributes nothing
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
        Raw r;
        r.ok = have && tk < rem;
        const unsigned trow = (unsigned)(tk < rem ? tk : rem - 1);
        const unsigned tb_ = (unsigned)tile * 8192u;              // byte offset of the tile in x / dy (scalar)
        r.x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, trow * 256 + 16 * c, tb_, 0));
        r.dy = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dy_rs, trow * 256 + 16 * c, tb_, 0));
        r.mk = 0x01010101u;
        r.m1w = u32x4{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
        if (DROP) {
            r.mk = __builtin_amdgcn_raw_buffer_load_b32(m2_rs, trow * 64 + 4 * c, (unsigned)tile * 2048u, 0);
            r.m1w = __builtin_amdgcn_raw_buffer_load_b128(m1_rs, trow * 256 + 16 * c, tb_, 0);
        }
        return r;
    };
    // LayerNorm + dz of the staged token; posts the wave's largest |dz| in zmax_l[buf][wv]
    auto process = [&](const Raw& r, Proc& q, int buf) __attribute__((always_inline)) {
        const float mean = red_c_sum((r.x[0] + r.x[1]) + (r.x[2] + r.x[3])) * (1.0f / 64.0f);
        const f32x4 d = r.x - splat4(mean);
        const float rstd = rsqrtf(red_c_sum(fmaf(d[0], d[0], d[1] * d[1]) + fmaf(d[2], d[2], d[3] * d[3])) * (1.0f / 64.0f) + CMGAN_EPS);
        q.xn = d * splat4(rstd) * gam + bet;
        f32x4 k = splat4(0.5f);
        if (DROP) k = tx_mask4(r.mk, 0.5f * ms);
        q.dz = r.ok ? r.dy * k : splat4(0.f);
        q.mb = fa_pack4(r.m1w[0]) | (fa_pack4(r.m1w[1]) << 4) | (fa_pack4(r.m1w[2]) << 8) | (fa_pack4(r.m1w[3]) << 12);
        const float mx = tx_wave_max(tx_absmax4(q.dz, 0.f));
        if (lane == 0) zmax_l[buf * 8 + wv] = mx;
    };
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };   // -> SGPR
    float s_run = 1.f, inv_run = 1.f;
    bool fresh = true;
    auto decide = [&](int buf) __attribute__((always_inline)) {    // identical in every wave: same 8 values, same rule
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, zmax_l[buf * 8 + k]);
        const float ms_ = m * s_run;
        if (m > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {
            tx_pow2(m, s_run, inv_run);
            fresh = false;
        }
        s_run = uni(s_run);
        inv_run = uni(inv_run);
    };
    auto write_images = [&](int buf, const Proc& q, float sc) __attribute__((always_inline)) {
        FaImg& I = img[buf];
        f16x4 h, l;
        split4(q.xn, h, l);
        *reinterpret_cast<f16x4*>(&I.xnh[tk * FA_PR + 4 * c]) = h;
        *reinterpret_cast<f16x4*>(&I.xnl[tk * FA_PR + 4 * c]) = l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.xth[(4 * c + e) * FA_PT + pos] = h[e];
            I.xtl[(4 * c + e) * FA_PT + pos] = l[e];
        }
        split4(q.dz * splat4(sc), h, l);
        *reinterpret_cast<f16x4*>(&I.dzh[tk * FA_PR + 4 * c]) = h;
        *reinterpret_cast<f16x4*>(&I.dzl[tk * FA_PR + 4 * c]) = l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.zth[(4 * c + e) * FA_PT + pos] = h[e];
            I.ztl[(4 * c + e) * FA_PT + pos] = l[e];
        }
        if (DROP) I.mbits[tk * 16 + c] = (unsigned short)q.mb;
    };

    f32x4 acc2[2][4], acc1[2][4];                                 // dW2 [hidden 4 g + r][out 16 ob + c], dW1 [hidden][in 16 cb + c]
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc2[hb][k] = splat4(0.f); acc1[hb][k] = splat4(0.f); }
    float sA = 1.f, invA = 1.f;                                   // scale the accumulators are at
    auto consume = [&](int i, int buf, float sc, float inv) __attribute__((always_inline)) {
        const FaImg& I = img[buf];
        const long tile = (long)blockIdx.x + (long)i * G;
        const unsigned dh_tb = (unsigned)tile * 32768u;           // byte offset of the tile in dh (scalar)
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);      // valid tokens of the tile (>= 1)
        // store offsets of this lane's 8 tokens; a token past M gets an offset beyond the descriptor's num_records and the
        // hardware drops the store (the range check sees the VGPR offset only - not the scalar tile offset: the launcher
        // keeps dh under 2 GB)
        unsigned dh_vo[2][4];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dh_vo[tb][r] = 16 * tb + 4 * g + r < rem ? (unsigned)(((16 * tb + 4 * g + r) * 256 + 32 * wv + c) * 4) : 0x80000000u;
        f32x4 h[2][2], dd[2][2];                                  // [hb][tb]   (b1 is added below: a splat is 4 VGPRs)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) { h[hb][tb] = splat4(0.f); dd[hb][tb] = splat4(0.f); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int o = (16 * tb + c) * FA_PR + 32 * ks + 8 * g;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&I.xnh[o]), al = *reinterpret_cast<const f16x8*>(&I.xnl[o]);
                const f16x8 zh = *reinterpret_cast<const f16x8*>(&I.dzh[o]), zl = *reinterpret_cast<const f16x8*>(&I.dzl[o]);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    h[hb][tb] = mfma32h(ah, w1h[hb][ks], h[hb][tb]);
                    dd[hb][tb] = mfma32h(zh, w2h[hb][ks], dd[hb][tb]);
                }
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    h[hb][tb] = mfma32l(ah, w1l[hb][ks], h[hb][tb]);
                    dd[hb][tb] = mfma32l(zh, w2l[hb][ks], dd[hb][tb]);
                }
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    h[hb][tb] = mfma32l(al, w1h[hb][ks], h[hb][tb]);
                    dd[hb][tb] = mfma32l(zl, w2h[hb][ks], dd[hb][tb]);
                }
            }
        f16x8 d1h[2], d1l[2], dhh[2], dhl[2];
        float dhm = 0.f;
        unsigned mw[2][4];                                        // keep bits of token 16 tb + 4 g + r for this wave's two
#pragma unroll                                                    // hidden 16-blocks: one dword
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                mw[tb][r] = DROP ? *reinterpret_cast<const unsigned*>(&I.mbits[(16 * tb + 4 * g + r) * 16 + 2 * wv]) : 0xffffffffu;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {                          // one hidden block at a time: h / dd die as d1 / dh are split
            f32x4 d1v[2], dhv[2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hv = h[hb][tb][r] + b1c[hb], sg = sigmoidf_fast(hv);
                    const float sgm = DROP ? (((mw[tb][r] >> (16 * hb + c)) & 1u) ? sg * ms : 0.f) : sg;   // keep-mask x sigmoid
                    d1v[tb][r] = hv * sgm;
                    dhv[tb][r] = dd[hb][tb][r] * (sgm * fmaf(hv, 1.f - sg, 1.f));                   // dh at scale sc
                    const float dht = dhv[tb][r] * inv;
                    dhm = fmaxf(dhm, fabsf(dht));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dht), dh_rs, dh_vo[tb][r] + 64u * hb, dh_tb, 0);
                }
            split8(d1v[0], d1v[1], d1h[hb], d1l[hb]);
            split8(dhv[0], dhv[1], dhh[hb], dhl[hb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        dhm = tx_wave_max(dhm);
        if (lane == 0) atomicMax(&dhmax_l[i & 1], __float_as_uint(dhm));     // non-negative floats order as integers
        if (sc != sA) {                                           // the scale moved: exact power-of-two ratio (wave-uniform, rare)
            const float ratio = sc * invA;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc2[hb][k] = acc2[hb][k] * splat4(ratio);
                    acc1[hb][k] = acc1[hb][k] * splat4(ratio);
                }
            sA = sc; invA = inv;
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const int o = (16 * ob + c) * FA_PT + 8 * g;
            const f16x8 zh = fa_ld8(&I.zth[o]), zl = fa_ld8(&I.ztl[o]);
            const f16x8 xh = fa_ld8(&I.xth[o]), xl = fa_ld8(&I.xtl[o]);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                acc2[hb][ob] = mfma32h(d1h[hb], zh, acc2[hb][ob]);
                acc1[hb][ob] = mfma32h(dhh[hb], xh, acc1[hb][ob]);
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                acc2[hb][ob] = mfma32l(d1h[hb], zl, acc2[hb][ob]);
                acc1[hb][ob] = mfma32l(dhh[hb], xl, acc1[hb][ob]);
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                acc2[hb][ob] = mfma32l(d1l[hb], zh, acc2[hb][ob]);
                acc1[hb][ob] = mfma32l(dhl[hb], xh, acc1[hb][ob]);
            }
            if (wv == ob) {                                       // db2 partial of the tile: column sums of dz (hi + lo: 2^-22)
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += (float)zh[e] + (float)zl[e];
                sm = red_g_sum(sm) * inv;
                if (g == 0) o_dzc[tile * 64 + 16 * ob + c] = sm;
            }
        }
    };

    Raw rw = load(0);
    Proc pc;
    process(rw, pc, 0);
    __syncthreads();
    decide(0);
    float s_cur = s_run, inv_cur = inv_run;
    write_images(0, pc, s_cur);
    rw = load(1);
    process(rw, pc, 1);
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < nloc; ++i) {
        const int buf = i & 1;
        if (wv == 0 && lane == 0 && i > 0) {                      // the previous tile's |dh| maximum (part B's scale)
            o_dhmax[(long)blockIdx.x + (long)(i - 1) * G] = __uint_as_float(dhmax_l[buf ^ 1]);
            dhmax_l[buf ^ 1] = 0u;
        }
        const Raw r2 = load(i + 2);                               // in flight under the products
        decide(buf ^ 1);                                          // tile i + 1 (past the end: all-zero maxima, nothing moves)
        const float s_nxt = s_run, inv_nxt = inv_run;
        write_images(buf ^ 1, pc, s_nxt);
        consume(i, buf, s_cur, inv_cur);
        process(r2, pc, buf);                                     // tile i + 2
        __syncthreads();
        s_cur = s_nxt; inv_cur = inv_nxt;
    }
    // (lane id recomputed: threadIdx.x kept for this one test was the kernel's only scratch dword)
    if (wv == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        o_dhmax[(long)blockIdx.x + (long)(nloc - 1) * G] = __uint_as_float(dhmax_l[(nloc - 1) & 1]);
    float* s2 = part_w2 + (long)blockIdx.x * 16384;
    float* s1 = part_w1 + (long)blockIdx.x * 16384;
    int ce = c, ge = g;
    asm volatile("" : "+v"(ce), "+v"(ge));                        // (opaque: the slab offsets would otherwise be computed before
                                                                  // the loop and live - spilled - through it)
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            stg4(s2 + (unsigned)((16 * k + ce) * 256 + 32 * wv + 16 * hb + 4 * ge), acc2[hb][k] * splat4(invA));
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[(unsigned)((32 * wv + 16 * hb + 4 * ge + r) * 64 + 16 * k + ce)] = acc1[hb][k][r] * invA;
        }
}

// ---------------------------------------------------------------------------------
// out_partial[s][i][j] = sum over the s-th token range of P[m][i] Q[m][j]  (P [M,R], Q [M,C] row-major fp32): the
// token-contraction weight gradient of wgrad_partial64_kernel (train.hip) with split products.  A wave-step is 32
// tokens: lane (c, g) feeds slot e of the contraction with token 8 g + e, i.e. eight dwords of one column of P
// (resp. Q) per 16-wide block, split to fp16 hi / lo in registers; 16 output tiles x 3 products = 48 MFMAs per step
// instead of 128 fp32 ones.  Same 64 x 64 tile per block, same fixed-order LDS combine, same slab layout.
// ---------------------------------------------------------------------------------
// colp (optional): the column sums of P - a bias gradient - as [split][R] partials for free: the blocks with jc = 0 hold
// every P element of their range in registers anyway (the separate column-sum pass re-read P: 1 GB for the conv module's
// [M,256] gate gradient).
__global__ __launch_bounds__(256) void wgrad_partial64_x3_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                 long M, int R, int C, float* __restrict__ partial,
                                                                 float* __restrict__ colp) {
    __shared__ float red[2][64 * 64];
    __shared__ float cs_l[4][64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ic = blockIdx.x, jc = blockIdx.y, s = blockIdx.z;
    const int nsplit = gridDim.z;
    const long full = M / 32, per = (full + nsplit - 1) / nsplit;            // full 32-row steps, dealt to the splits
    const long st0 = (long)s * per, st1 = st0 + per < full ? st0 + per : full;
    f32x4 acc[4][4];                                  // [ib][jb]
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = splat4(0.f);
    const unsigned oa = (unsigned)(8 * g * R + 64 * ic + c), ob = (unsigned)(8 * g * C + 64 * jc + c);
    struct Raw { f32x4 a[4][2], b[4][2]; };           // [block][e >> 2][e & 3]: token 8 g + e of the step (64 dwords)
    struct Ops { f16x8 ah[4], al[4], bh[4], bl[4]; }; // the same step as split operands (64 registers)
    auto load = [&](long st, Raw& t) {
        const float* __restrict__ pa = P + st * 32 * R + oa;             // uniform base + lane offset
        const float* __restrict__ qa = Q + st * 32 * C + ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) t.a[ib][e >> 2][e & 3] = pa[e * R + 16 * ib];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) t.b[jb][e >> 2][e & 3] = qa[e * C + 16 * jb];
        }
    };
    auto split = [&](const Raw& t, Ops& o) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            split8(t.a[k][0], t.a[k][1], o.ah[k], o.al[k]);
            split8(t.b[k][0], t.b[k][1], o.bh[k], o.bl[k]);
        }
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32h(o.ah[ib], o.bh[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(o.ah[ib], o.bl[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(o.al[ib], o.bh[jb], acc[ib][jb]);
        }
    };
    // P is a gradient (unbounded below: ~1 / numel), Q an activation (bounded): P is multiplied by an exact power of two
    // sP kept - like the stale softmax reference of the attention kernels - in a band around the running magnitude of
    // this wave's steps: a step whose largest |P| leaves [2^-2, 2^13] / sP re-references (sP from that step, the
    // accumulators rescaled by the exact ratio; wave-uniform and rare); the final sums are multiplied by 1 / sP.
    float sP = 1.f, iP = 1.f;
    bool fresh = true;
    const bool want_cs = colp != nullptr && jc == 0;              // block-uniform
    f32x4 csum = splat4(0.f);                                     // [ib]: this lane's tokens of column 64 ic + 16 ib + c
    auto rescale = [&](Raw& t) {
        float m = 0.f;
        if (want_cs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 v = t.a[k][0] + t.a[k][1];
                csum[k] += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) m = tx_absmax4(t.a[k][1], tx_absmax4(t.a[k][0], m));
        m = tx_wave_max(m);
        const float ms_ = m * sP;
        if (m > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {           // wave-uniform
            float s2, i2;
            tx_pow2(m, s2, i2);
            const float ratio = s2 * iP;                                   // exact: both are powers of two
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(ratio);
            sP = s2; iP = i2; fresh = false;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t.a[k][0] = t.a[k][0] * splat4(sP);
            t.a[k][1] = t.a[k][1] * splat4(sP);
        }
    };
    // the 64 operand dwords of the next step are in flight while the 48 MFMAs of the current one run
    Raw raw;
    Ops ops;
    long st = st0 + wv;
    if (st < st1) {
        load(st, raw);
        rescale(raw);
        split(raw, ops);
    }
    while (st < st1) {
        const long sn = st + 4;
        if (sn < st1) load(sn, raw);
        mma(ops);
        if (sn < st1) {
            rescale(raw);
            split(raw, ops);
        }
        st = sn;
    }
    if ((M & 31) && s == nsplit - 1 && wv == 0) {   // the one ragged step of the tensor: rows past M contribute zeros
        const float* __restrict__ pa = P + full * 32 * R + oa;
        const float* __restrict__ qa = Q + full * 32 * C + ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool okr = full * 32 + 8 * g + e < M;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) raw.a[ib][e >> 2][e & 3] = okr ? pa[e * R + 16 * ib] : 0.f;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) raw.b[jb][e >> 2][e & 3] = okr ? qa[e * C + 16 * jb] : 0.f;
        }
        rescale(raw);
        split(raw, ops);
        mma(ops);
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(iP);
    // (wave 2 + wave 0), (wave 3 + wave 1), then (wave 1 + wave 0): element (row 16 ib + 4 g + r, col 16 jb + c)
    auto put = [&](float* dst) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(16 * ib + 4 * g + r) * 64 + 16 * jb + c] = acc[ib][jb][r];
    };
    auto add = [&](const float* src) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ib][jb][r] += src[(16 * ib + 4 * g + r) * 64 + 16 * jb + c];
    };
    if (want_cs) {                                                // lane groups, then waves 0..3 in order: a fixed order
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float v = red_g_sum(csum[ib]);
            if (g == 0) cs_l[wv][16 * ib + c] = v;
        }
    }
    if (wv >= 2) put(red[wv - 2]);
    __syncthreads();
    if (wv < 2) add(red[wv]);
    __syncthreads();
    if (wv == 1) put(red[0]);
    __syncthreads();
    if (wv == 0) {
        add(red[0]);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    partial[((long)s * R + 64 * ic + 16 * ib + 4 * g + r) * C + 64 * jc + 16 * jb + c] = acc[ib][jb][r];
        if (want_cs) colp[(long)s * R + 64 * ic + lane] = (cs_l[0][lane] + cs_l[1][lane]) + (cs_l[2][lane] + cs_l[3][lane]);
    }
}


// ---------------------------------------------------------------------------------
// Dense-block weight gradient (db_conv_wgrad_kernel of train.hip with split products):
//   partial[tap][s][co][ci] = sum over the s-th range of positions m of dz[m][co] * a[m shifted by tap][ci]
// Steps of 32 positions; lane (c, g) feeds slot e with position m0 + 8 g + e: eight dwords of one dz column and of one
// (shifted, zero outside the plane) activation column per 16-wide block.  dz is a gradient: scaled by the running
// exact power of two of wgrad_partial64_x3_kernel.  Same block shape, LDS combine and slab layout as the fp32 kernel.
// ---------------------------------------------------------------------------------
// colp (optional): the conv-bias gradient - column sums of dz - as [split][64] partials from the tap-0 blocks (the separate
// pass re-read the four dz planes of a dense block: 1 - 2 GB).
__global__ __launch_bounds__(256) void db_conv_wgrad_x3_kernel(const float* __restrict__ dz, const float* __restrict__ a,
                                                               int B, int T, int F, int dil, int nsplit,
                                                               float* __restrict__ partial, float* __restrict__ colp) {
    __shared__ float red[2][64 * 64];
    __shared__ float cs_l[4][64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    // 1-D grid of 6 nsplit blocks (nsplit a multiple of 8), dealt round-robin to the 8 XCDs: the six tap blocks of one
    // position range read the same dz / activation rows, so they are given to the SAME XCD (one L2 fill, five hits)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tap = q % 6, s = (q / 6) * 8 + xcd;
    const int kt = tap / 3, kf = tap - 3 * kt, dt = (kt - 1) * dil, df = kf - 1;
    const unsigned M = (unsigned)B * T * F, tf = (unsigned)T * F;
    const unsigned steps = (M + 31) / 32, per = (steps + nsplit - 1) / nsplit;
    const unsigned st0 = s * per, st1 = st0 + per < steps ? st0 + per : steps;
    f32x4 acc[4][4];                                  // [ib][jb]
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = splat4(0.f);
    float sP = 1.f, iP = 1.f;
    bool fresh = true;
    const bool want_cs = colp != nullptr && tap == 0;             // block-uniform
    f32x4 csum = splat4(0.f);                                     // [ib]: this lane's positions of channel 16 ib + c
    // (t, f, clip) of this lane's first position, carried from step to step (a step advances every lane by 128 positions);
    // PMC: 585 VALU instructions per wave and step against 48 MFMAs - the matrix pipe 21 % busy, the vector ALU 64 %:
    // two divisions, 16 address computations, 64 selects.  INTERIOR steps (every lane's eight positions inside one row, the
    // tap in range for all of them: ~70 % of the steps at F = 101) need none of the per-position work: both operand runs are
    // contiguous rows, fetched as base + immediate.
    unsigned m0 = (st0 + wv) * 32 + 8 * g;
    unsigned bb = m0 / tf;
    int t, f;
    {
        const unsigned rem = m0 - bb * tf;
        t = (int)(rem / (unsigned)F);
        f = (int)(rem - (unsigned)t * F);
    }
    const int adv_t = 128 / F, adv_f = 128 - adv_t * F;           // 128 positions = adv_t rows + adv_f (uniform)
    for (unsigned st = st0 + wv; st < st1; st += 4) {
        f32x4 av[4][2], bv[4][2];                     // [block][e >> 2][e & 3]
        float mx = 0.f;
        const int ts0 = t + dt;
        const bool interior = m0 + 7u < M && f + df >= 0 && f + 7 + df < F && f + 7 < F && ts0 >= 0 && ts0 < T;
        if (__builtin_amdgcn_ballot_w64(!interior) == 0ull) {
            // 32-bit BYTE offsets from the (uniform) plane bases (a plane is at most 2^32 bytes: checked by the launcher)
            const unsigned oz0 = (m0 * 64u + (unsigned)c) * 4u;
            const unsigned oa0 = ((unsigned)((int)m0 + dt * F + df) * 64u + (unsigned)c) * 4u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dz + 16 * ib) + oz0 + 256u * e);
                    av[ib][e >> 2][e & 3] = v;
                    mx = fmaxf(mx, fabsf(v));
                }
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
                    bv[jb][e >> 2][e & 3] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a + 16 * jb) + oa0 + 256u * e);
            }
        } else {
            unsigned bbe = bb;
            int te = t, fe = f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned m = m0 + e;
                const bool ok = m < M;
                const int ts = te + dt, fs = fe + df;
                const bool inb = ok && ts >= 0 && ts < T && fs >= 0 && fs < F;
                const unsigned src = inb ? ((bbe * (unsigned)T + (unsigned)ts) * (unsigned)F + (unsigned)fs) : 0u;
                const unsigned mm = ok ? m : M - 1u;
                const unsigned oz = (mm * 64u + (unsigned)c) * 4u, oa = (src * 64u + (unsigned)c) * 4u;
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const float v0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dz + 16 * ib) + oz);
                    const float v = ok ? v0 : 0.f;
                    av[ib][e >> 2][e & 3] = v;
                    mx = fmaxf(mx, fabsf(v));
                }
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a + 16 * jb) + oa);
                    bv[jb][e >> 2][e & 3] = inb ? v : 0.f;
                }
                if (++fe == F) { fe = 0; if (++te == T) { te = 0; ++bbe; } }
            }
        }
        // the same lane, four steps (128 positions) on
        m0 += 128u;
        f += adv_f; t += adv_t;
        if (f >= F) { f -= F; ++t; }
        while (t >= T) { t -= T; ++bb; }
        if (want_cs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 tsum = av[k][0] + av[k][1];
                csum[k] += (tsum[0] + tsum[1]) + (tsum[2] + tsum[3]);
            }
        }
        mx = tx_wave_max(mx);
        const float ms_ = mx * sP;
        if (mx > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {          // wave-uniform, rare (see wgrad_partial64_x3_kernel)
            float s2, i2;
            tx_pow2(mx, s2, i2);
            const float ratio = s2 * iP;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(ratio);
            sP = s2; iP = i2; fresh = false;
        }
        f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            split8(av[k][0] * splat4(sP), av[k][1] * splat4(sP), ah[k], al[k]);
            split8(bv[k][0], bv[k][1], bh[k], bl[k]);
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32h(ah[ib], bh[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(ah[ib], bl[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(al[ib], bh[jb], acc[ib][jb]);
        }
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(iP);
    auto put = [&](float* dst) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(16 * ib + 4 * g + r) * 64 + 16 * jb + c] = acc[ib][jb][r];
    };
    auto add = [&](const float* src) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ib][jb][r] += src[(16 * ib + 4 * g + r) * 64 + 16 * jb + c];
    };
    if (want_cs) {                                                // lane groups, then waves 0..3 in order: a fixed order
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float tsum = red_g_sum(csum[ib]);
            if (g == 0) cs_l[wv][16 * ib + c] = tsum;
        }
    }
    if (wv >= 2) put(red[wv - 2]);
    __syncthreads();
    if (wv < 2) add(red[wv]);
    __syncthreads();
    if (wv == 1) put(red[0]);
    __syncthreads();
    if (wv == 0) {
        add(red[0]);
        if (want_cs) colp[(long)s * 64 + lane] = (cs_l[0][lane] + cs_l[1][lane]) + (cs_l[2][lane] + cs_l[3][lane]);
        float* out = partial + ((long)tap * nsplit + s) * 4096;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * ib + 4 * g + r) * 64 + 16 * jb + c] = acc[ib][jb][r];
    }
}
void launch_db_conv_wgrad_x3(LaunchCtx ctx, const float* dz, const float* a, int B, int T, int F, int dil, int nsplit,
                             float* partial, float* colp) {
    LAUNCH(ctx, "dense_train_wgrad", (db_conv_wgrad_x3_kernel<<<6 * nsplit, 256, 0, ctx.stream>>>(dz, a, B, T, F, dil,
                                                                                                      nsplit, partial, colp)));
}

// ---------------------------------------------------------------------------------
// Row-conv weight gradient (rc_wgrad_kernel of train.hip with split products; the encoder's stride-2 1 x 3 conv, the
// decoders' sub-pixel conv):  partial[kw][s][co][ci] = sum over the s-th range of OUTPUT positions m = (b, t, fo) of
// dz[m][co] * in[(b, t, fo SF - PL + kw)][ci].  Block = (64-channel group of co, tap kw, range s), one 64 x 64 tile as in
// db_conv_wgrad_x3_kernel: steps of 32 positions, lane (c, g) feeds slot e with position m0 + 8 g + e, the (row, fo) of a
// lane's eight positions from ONE 32-bit division plus carries (the fp32 kernel: a 64-bit division per position), dz at
// the running exact power-of-two scale.  NG = 2: dz is the pixel-shuffled plane, channel group r of position m at row 2 m + r.
// ---------------------------------------------------------------------------------
struct RcGeomX3 { int B, T, Fi, Fo, KW, SF, PL; };    // = RcGeom (train.hip)
// colp (optional): the bias gradient - column sums of dz - as [split][64 NG] partials from the kw = 0 blocks, which hold every
// dz element of their range in registers (the separate pass re-read the whole gradient plane).
template <int NG>
__global__ __launch_bounds__(256) void rc_wgrad_x3_kernel(const float* __restrict__ dz, const float* __restrict__ in, RcGeomX3 gm,
                                                          int nsplit, float* __restrict__ partial, float* __restrict__ colp) {
    __shared__ float red[2][64 * 64];
    __shared__ float cs_l[4][64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int rb = blockIdx.x, kw = blockIdx.y, s = blockIdx.z;
    const unsigned Mo = (unsigned)gm.B * gm.T * gm.Fo;
    const unsigned steps = (Mo + 31) / 32, per = (steps + nsplit - 1) / nsplit;
    const unsigned st0 = s * per, st1 = st0 + per < steps ? st0 + per : steps;
    f32x4 acc[4][4];                                  // [ib][jb]
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = splat4(0.f);
    float sP = 1.f, iP = 1.f;
    bool fresh = true;
    const bool want_cs = colp != nullptr && kw == 0;              // block-uniform
    f32x4 csum = splat4(0.f);                                     // [ib]: this lane's positions of channel 64 rb + 16 ib + c
    for (unsigned st = st0 + wv; st < st1; st += 4) {
        const unsigned m0 = st * 32 + 8 * g;
        unsigned bt = m0 / (unsigned)gm.Fo;
        int fo = (int)(m0 - bt * (unsigned)gm.Fo);
        f32x4 av[4][2], bv[4][2];                     // [block][e >> 2][e & 3]
        float mx = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned m = m0 + e;
            const bool ok = m < Mo;
            const int fi = fo * gm.SF - gm.PL + kw;
            const bool inb = ok && fi >= 0 && fi < gm.Fi;
            const unsigned src = inb ? bt * (unsigned)gm.Fi + (unsigned)fi : 0u;
            const unsigned zrow = ok ? (NG == 1 ? m : 2u * m + (unsigned)rb) : 0u;
            const unsigned oz = (zrow * 64u + (unsigned)c) * 4u, oa = (src * 64u + (unsigned)c) * 4u;     // 32-bit byte offsets
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const float v0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(dz + 16 * ib) + oz);
                const float v = ok ? v0 : 0.f;
                av[ib][e >> 2][e & 3] = v;
                mx = fmaxf(mx, fabsf(v));
            }
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(in + 16 * jb) + oa);
                bv[jb][e >> 2][e & 3] = inb ? v : 0.f;
            }
            if (++fo == gm.Fo) { fo = 0; ++bt; }
        }
        if (want_cs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 t = av[k][0] + av[k][1];
                csum[k] += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        mx = tx_wave_max(mx);
        const float ms_ = mx * sP;
        if (mx > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {          // wave-uniform, rare (see wgrad_partial64_x3_kernel)
            float s2, i2;
            tx_pow2(mx, s2, i2);
            const float ratio = s2 * iP;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(ratio);
            sP = s2; iP = i2; fresh = false;
        }
        f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            split8(av[k][0] * splat4(sP), av[k][1] * splat4(sP), ah[k], al[k]);
            split8(bv[k][0], bv[k][1], bh[k], bl[k]);
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32h(ah[ib], bh[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(ah[ib], bl[jb], acc[ib][jb]);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma32l(al[ib], bh[jb], acc[ib][jb]);
        }
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = acc[ib][jb] * splat4(iP);
    auto put = [&](float* dst) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(16 * ib + 4 * g + r) * 64 + 16 * jb + c] = acc[ib][jb][r];
    };
    auto add = [&](const float* src) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ib][jb][r] += src[(16 * ib + 4 * g + r) * 64 + 16 * jb + c];
    };
    if (want_cs) {                                                // lane groups, then waves 0..3 in order: a fixed order
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const float t = red_g_sum(csum[ib]);
            if (g == 0) cs_l[wv][16 * ib + c] = t;
        }
    }
    if (wv >= 2) put(red[wv - 2]);
    __syncthreads();
    if (wv < 2) add(red[wv]);
    __syncthreads();
    if (wv == 1) put(red[0]);
    __syncthreads();
    if (wv == 0) {
        add(red[0]);
        put(partial + (((long)kw * nsplit + s) * (64 * NG) + 64 * rb) * 64);      // rows co = 64 rb + ..., 64 columns ci
        if (want_cs) colp[(long)s * (64 * NG) + 64 * rb + lane] = (cs_l[0][lane] + cs_l[1][lane]) + (cs_l[2][lane] + cs_l[3][lane]);
    }
}
// ---------------------------------------------------------------------------------
// Row-conv data gradient (rc_dgrad_kernel of train.hip with split products):
//   din[(b, t, fi)][ci] = sum_kw W_kw^T dz[(b, t, fo)]   with fo SF - PL + kw = fi
// Per-token linear maps like the conv module's: persistent 512-thread blocks, one 16-position block per wave trip, the
// three transposed tap images (rows = ci, contraction = the 64 NG output channels) built ONCE per block in LDS from the raw
// weight [Co, 64, 1, 3]; the up to three dz rows a position touches are fetched first, scaled by the exact power of two of
// the block's largest magnitude (dz is a gradient), split, and contracted with 12 NG x 3 MFMAs per output block.
// ---------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(512) void rc_dgrad_x3_kernel(const float* __restrict__ dz, const float* __restrict__ wraw, RcGeomX3 gm,
                                                          float* __restrict__ din, int ntiles) {
    constexpr int M32 = 2 * NG;                                   // k32 blocks of the contraction (64 NG output channels)
    extern __shared__ __attribute__((aligned(16))) _Float16 rd_w[];             // [kw 3][ob 4][M32][hi | lo][64][8]
    for (int u = threadIdx.x; u < 3 * 4 * M32 * 64; u += blockDim.x) {
        const int ln = u & 63, blk = u >> 6, m = blk % M32, ob = (blk / M32) & 3, kw = blk / (4 * M32);
        const int ci = 16 * ob + (ln & 15);
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = 32 * m + 16 * (e >> 2) + 4 * (ln >> 4) + (e & 3);
            const float v = wraw[((long)co * 64 + ci) * 3 + kw];
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)(v - (float)h);
        }
        *reinterpret_cast<f16x8*>(rd_w + (long)blk * 1024 + ln * 8) = hi;
        *reinterpret_cast<f16x8*>(rd_w + (long)blk * 1024 + 512 + ln * 8) = lo;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const unsigned Mi = (unsigned)gm.B * gm.T * gm.Fi;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {
        const unsigned m = (unsigned)tile * 16u + (unsigned)c;
        const bool ok = m < Mi;
        const unsigned mm = ok ? m : Mi - 1u;
        const unsigned bt = mm / (unsigned)gm.Fi;
        const int fi = (int)(mm - bt * (unsigned)gm.Fi);
        f32x4 v[3][4 * NG];
        float mx = 0.f;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int num = fi + gm.PL - kw;
            const int fo = gm.SF == 1 ? num : num >> 1;                         // SF is 1 or 2 (the launcher checks)
            const bool inb = ok && num >= 0 && fo * gm.SF == num && fo < gm.Fo;
            const unsigned srow = inb ? bt * (unsigned)gm.Fo + (unsigned)fo : 0u;
#pragma unroll
            for (int kb = 0; kb < 4 * NG; ++kb) {
                const unsigned zrow = NG == 1 ? srow : 2u * srow + (unsigned)(kb >> 2);
                f32x4 t = ldg4(dz + (size_t)zrow * 64 + 16 * (kb & 3) + 4 * g);
                if (!inb) t = splat4(0.f);
                v[kw][kb] = t;
                mx = tx_absmax4(t, mx);
            }
        }
        float zs, zinv;
        tx_pow2(tx_wave_max(mx), zs, zinv);
        f32x4 acc[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) acc[ob] = splat4(0.f);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            f16x8 bh[1][M32], bl[1][M32];
#pragma unroll
            for (int mb = 0; mb < M32; ++mb)
                split8(v[kw][2 * mb] * splat4(zs), v[kw][2 * mb + 1] * splat4(zs), bh[0][mb], bl[0][mb]);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                f32x4 a1[1] = {acc[ob]};
                lin_acc_x3<M32, 1>(rd_w + (long)((kw * 4 + ob) * M32) * 1024 + lane * 8, bh, bl, a1);
                acc[ob] = a1[0];
            }
        }
        if (ok) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) stg4(din + (size_t)mm * 64 + 16 * ob + 4 * g, acc[ob] * splat4(zinv));
        }
    }
}
// returns false when the geometry is not one this kernel covers (the caller takes the fp32 kernel)
bool launch_rc_dgrad_x3(LaunchCtx ctx, int ng, const float* dz, const float* wraw, const int* gm7, float* din) {
    const RcGeomX3 gm{gm7[0], gm7[1], gm7[2], gm7[3], gm7[4], gm7[5], gm7[6]};
    if (gm.KW != 3 || (gm.SF != 1 && gm.SF != 2)) return false;
    const long Mi = (long)gm.B * gm.T * gm.Fi;
    if (Mi * 256 >= (1l << 32) || (long)gm.B * gm.T * gm.Fo * ng * 256 >= (1l << 32)) return false;
    const int ntiles = (int)((Mi + 15) / 16);
    const size_t lds = (size_t)3 * 4 * 2 * ng * 1024 * sizeof(_Float16);        // 48 KB x NG
    const void* fn = ng == 1 ? reinterpret_cast<const void*>(&rc_dgrad_x3_kernel<1>) : reinterpret_cast<const void*>(&rc_dgrad_x3_kernel<2>);
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, bool> optin;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = optin.find({dev, fn});
        if (it == optin.end()) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) (void)hipGetLastError();
            it = optin.emplace(std::make_pair(dev, fn), e == hipSuccess).first;
        }
        if (!it->second) return false;
    }
    const int want = (ntiles + 7) / 8, cap = ng == 1 ? 768 : 256;              // 48 KB: three blocks per CU; 96 KB: one
    const int grid = want < cap ? (want > 0 ? want : 1) : cap;
    if (ng == 1)
        LAUNCH(ctx, "rowconv_train", (rc_dgrad_x3_kernel<1><<<grid, 512, lds, ctx.stream>>>(dz, wraw, gm, din, ntiles)));
    else
        LAUNCH(ctx, "rowconv_train", (rc_dgrad_x3_kernel<2><<<grid, 512, lds, ctx.stream>>>(dz, wraw, gm, din, ntiles)));
    return true;
}

// gm7: the seven ints of RcGeom.  The plane offsets are 32-bit: the caller checks rows * 256 < 2^32
void launch_rc_wgrad_x3(LaunchCtx ctx, int ng, const float* dz, const float* in, const int* gm7, int nsplit, float* partial,
                        float* colp) {
    const RcGeomX3 gm{gm7[0], gm7[1], gm7[2], gm7[3], gm7[4], gm7[5], gm7[6]};
    if (ng == 1)
        LAUNCH(ctx, "rowconv_train", (rc_wgrad_x3_kernel<1><<<dim3(1, gm.KW, nsplit), 256, 0, ctx.stream>>>(dz, in, gm, nsplit, partial,
                                                                                                            colp)));
    else
        LAUNCH(ctx, "rowconv_train", (rc_wgrad_x3_kernel<2><<<dim3(2, gm.KW, nsplit), 256, 0, ctx.stream>>>(dz, in, gm, nsplit, partial,
                                                                                                            colp)));
}

// ---------------------------------------------------------------------------------
// host side (called from train.hip's launch_ffn_train_* when TRAIN_X3 is on)
// ---------------------------------------------------------------------------------
static int tx_grid(int ntiles, int waves) {
    const int want = (ntiles + waves - 1) / waves;
    return want < 256 ? (want > 0 ? want : 1) : 256;
}

// images live where the fp32 fragment images did: four 64 KB slots at the head of the module's workspace
void ffn_x3_pack(LaunchCtx ctx, const FfnTrainParams& p, float* img) {
    _Float16* h = reinterpret_cast<_Float16*>(img);
    launch_pack_x3(ctx, "ffn_train_pack", PackX3Jobs{{{p.w1, 256, 64, 64, 0, h},                 // rows = hidden
                                                      {p.w2, 64, 256, 256, 0, h + 32768},        // rows = out
                                                      {p.w2, 256, 64, 256, 1, h + 2 * 32768},    // W2^T: rows = hidden
                                                      {p.w1, 64, 256, 64, 1, h + 3 * 32768}}},   // W1^T: rows = in
                   4);
}
void ffn_x3_forward(LaunchCtx ctx, const float* x, long M, const FfnTrainParams& p, const float* img,
                    const unsigned char* m1, const unsigned char* m2, float ms, const float* res, float* y) {
    const _Float16* h = reinterpret_cast<const _Float16*>(img);
    const int ntiles = (int)((M + 31) / 32);
    LAUNCH(ctx, "ffn_train_fwd", (ffn_train_fwd_x3_kernel<<<tx_grid(ntiles, TX_WAVES), TX_WAVES * 64, 0, ctx.stream>>>(
                                     x, M, h, h + 32768, p, m1, m2, ms, res, y, ntiles)));
}
// dhmax: scratch of ceil(M / 32) floats (one per tile), written by part A and read by part B
void ffn_x3_backward(LaunchCtx ctx, const float* x, const float* dy, long M, const FfnTrainParams& p, const float* img,
                     const unsigned char* m1, const unsigned char* m2, float ms, const float* dres, float* dx,
                     float* o_dz, float* o_d1, float* o_dh, float* o_xn, float* o_g1, float* o_dxn, float* dhmax, float* o_dzc,
                     float* o_dhc) {
    const _Float16* h = reinterpret_cast<const _Float16*>(img);
    const int ntiles = (int)((M + 31) / 32);
    LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_a_x3_kernel<<<tx_grid(ntiles, TX_WAVES), TX_WAVES * 64, 0, ctx.stream>>>(
                                     x, dy, M, h, h + 2 * 32768, p, m1, m2, ms, o_dz, o_xn, o_d1, o_dh, dhmax, o_dzc, ntiles)));
    const int gridb = (ntiles + 7) / 8 < 512 ? ((ntiles + 7) / 8 > 0 ? (ntiles + 7) / 8 : 1) : 512;
    LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_b_x3_kernel<<<gridb, 512, 0, ctx.stream>>>(
                                     x, o_dh, dhmax, M, h + 3 * 32768, p, dres, dx, o_g1, o_dxn, o_dhc, ntiles)));
}
// the fused form: part A + both weight gradients (slabs [grid][16384] of dW2 then dW1 behind `part`), then part B.
// Returns the number of slabs written per gradient, 0 if the device refuses the kernel's LDS (the caller falls back).
int ffn_x3_backward_fused(LaunchCtx ctx, const float* x, const float* dy, long M, const FfnTrainParams& p, const float* img,
                          const unsigned char* m1, const unsigned char* m2, float ms, const float* dres, float* dx,
                          float* o_dh, float* o_g1, float* o_dxn, float* dhmax, float* o_dzc, float* o_dhc, float* part_w2,
                          float* part_w1) {
    if ((m1 == nullptr) != (m2 == nullptr)) return 0;             // one mask only: not a case the trainer produces
    if (M * 1024 >= (1l << 31)) return 0;                         // the kernel addresses dh with 32-bit byte offsets (and marks
                                                                  // tokens past M with offset 2^31)
    const bool drop = m1 != nullptr;
    const void* fn = drop ? reinterpret_cast<const void*>(&ffn_train_bwd_aw_x3_kernel<true>)
                          : reinterpret_cast<const void*>(&ffn_train_bwd_aw_x3_kernel<false>);
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, bool> optin;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const size_t lds = 2 * sizeof(FaImg) + 18 * sizeof(float);
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = optin.find({dev, fn});
        if (it == optin.end()) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) (void)hipGetLastError();
            it = optin.emplace(std::make_pair(dev, fn), e == hipSuccess).first;
        }
        if (!it->second) return 0;
    }
    const _Float16* h = reinterpret_cast<const _Float16*>(img);
    const int ntiles = (int)((M + 31) / 32);
    const int grid = ntiles < 256 ? ntiles : 256;
    if (drop)
        LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_aw_x3_kernel<true><<<grid, 512, lds, ctx.stream>>>(
                                         x, dy, M, p, m1, m2, ms, o_dh, dhmax, o_dzc, part_w2, part_w1, ntiles)));
    else
        LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_aw_x3_kernel<false><<<grid, 512, lds, ctx.stream>>>(
                                         x, dy, M, p, m1, m2, ms, o_dh, dhmax, o_dzc, part_w2, part_w1, ntiles)));
    const int gridb = (ntiles + 7) / 8 < 512 ? ((ntiles + 7) / 8 > 0 ? (ntiles + 7) / 8 : 1) : 512;
    LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_b_x3_kernel<<<gridb, 512, 0, ctx.stream>>>(
                                     x, o_dh, dhmax, M, h + 3 * 32768, p, dres, dx, o_g1, o_dxn, o_dhc, ntiles)));
    return grid;
}
void launch_wgrad_partial64_x3(LaunchCtx ctx, const char* label, const float* P, const float* Q, long M, int R, int C,
                               float* partial, int nsplit, float* colp) {
    LAUNCH(ctx, label, (wgrad_partial64_x3_kernel<<<dim3(R / 64, C / 64, nsplit), 256, 0, ctx.stream>>>(P, Q, M, R, C, partial,
                                                                                                        colp)));
}

// ---------------------------------------------------------------------------------
// Conv module, per-token stages on split products (the fp32 kernels cm_pw1glu_kernel / cm_bwd2_kernel of train.hip ran
// 256 / 512 v_mfma_f32_16x16x4_f32 per 16-token tile with L2-streamed weight fragments: 35-40 % matrix-pipe busy, the
// largest fp32 items left in the step).  Same structure as the FeedForward kernels above: persistent blocks, weight
// images resident in LDS, the per-token chain in registers.
//   forward    u = a * sigmoid(gt),  [a ; gt] = W1 LN(x) + b1               conformer.py:161-164 (LayerNorm, pw conv, GLU)
//   backward   da = du sigmoid(gt),  dg = du a sigmoid'(gt)  (a, gt recomputed);  dxn = W1^T [da ; dg];  LayerNorm backward
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void cm_pw1glu_x3_kernel(const float* __restrict__ x, long M,
                                                           const _Float16* __restrict__ w1i, ConvModTrainParams p,
                                                           float* __restrict__ u, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 w1[32768];            // 64 KB: [16 output blocks][2 k32]
    __shared__ __attribute__((aligned(16))) float par_l[384];              // gamma | beta | b1[256]
    stage_lds16<4096, 512>(w1i, w1);
    for (int i = threadIdx.x; i < 384; i += blockDim.x)
        par_l[i] = i < 64 ? p.ln_w[i] : (i < 128 ? p.ln_b[i - 64] : p.pw1_b[i - 128]);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {
        long row[2];
        bool ok[2];
        f16x8 xbh[2][2], xbl[2][2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 xh[4], xn[4];
            float rstd;
            tx_load_norm(x, row[tb], g, par_l, xh, rstd, xn);
            split8(xn[0], xn[1], xbh[tb][0], xbl[tb][0]);
            split8(xn[2], xn[3], xbh[tb][1], xbl[tb][1]);
        }
#pragma unroll 2
        for (int ob = 0; ob < 8; ++ob) {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * ob + 4 * g]);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(&par_l[256 + 16 * ob + 4 * g]);
            f32x4 a[2] = {ba, ba}, gt[2] = {bg, bg};
            lin_acc_x3<2, 2>(w1 + ob * 2048 + lane * 8, xbh, xbl, a);
            lin_acc_x3<2, 2>(w1 + (ob + 8) * 2048 + lane * 8, xbh, xbl, gt);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = a[tb][e] * sigmoidf_fast(gt[tb][e]);
                if (ok[tb]) stg4(u + row[tb] * 128 + 16 * ob + 4 * g, r);
            }
        }
    }
}

// LDS: W1 image [16][2] (a, gt recompute) + W1^T image [4][8] (dxn) = 128 KB.  One 16-token block per wave trip: the
// 256 values of [da ; dg] stay in registers between the two products, scaled by the exact power of two of the
// block's largest magnitude before the split (du is a gradient).  g1c / dxc: per-block partial sums (ln_tile_colsums).
__global__ __launch_bounds__(TX_WAVES * 64) void cm_bwd2_x3_kernel(const float* __restrict__ x, const float* __restrict__ du,
                                                                   long M, const _Float16* __restrict__ w1i,
                                                                   const _Float16* __restrict__ w1ti, ConvModTrainParams p,
                                                                   const float* __restrict__ dres, float* __restrict__ dx,
                                                                   float* __restrict__ dag, float* __restrict__ xn_out,
                                                                   float* __restrict__ g1c, float* __restrict__ dxc,
                                                                   int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];
    __shared__ __attribute__((aligned(16))) float par_l[384];
    _Float16* w1 = wlds;
    _Float16* w1t = wlds + 32768;
    stage_lds16<4096, TX_WAVES * 64>(w1i, w1);
    stage_lds16<4096, TX_WAVES * 64>(w1ti, w1t);
    for (int i = threadIdx.x; i < 384; i += blockDim.x)
        par_l[i] = i < 64 ? p.ln_w[i] : (i < 128 ? p.ln_b[i - 64] : p.pw1_b[i - 128]);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * TX_WAVES + wv; tile < ntiles; tile += gridDim.x * TX_WAVES) {
        const long t = (long)tile * 16 + c;
        const bool ok = t < M;
        const long row = ok ? t : M - 1;
        f32x4 xh[4], xn[4];
        float rstd;
        tx_load_norm(x, row, g, par_l, xh, rstd, xn);
        f16x8 xbh[1][2], xbl[1][2];
        split8(xn[0], xn[1], xbh[0][0], xbl[0][0]);
        split8(xn[2], xn[3], xbh[0][1], xbl[0][1]);
        if (ok) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) stg4(xn_out + row * 64 + 16 * kb + 4 * g, xn[kb]);
        }
        f32x4 z[16];                                                   // [da (8 blocks) ; dg (8 blocks)]
        float zmax = 0.f;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * ob + 4 * g]);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(&par_l[256 + 16 * ob + 4 * g]);
            f32x4 a[1] = {ba}, gt[1] = {bg};
            lin_acc_x3<2, 1>(w1 + ob * 2048 + lane * 8, xbh, xbl, a);
            lin_acc_x3<2, 1>(w1 + (ob + 8) * 2048 + lane * 8, xbh, xbl, gt);
            f32x4 duv = ldg4(du + row * 128 + 16 * ob + 4 * g);
            if (!ok) duv = splat4(0.f);
            f32x4 da, dg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sg = sigmoidf_fast(gt[0][e]);
                da[e] = duv[e] * sg;
                dg[e] = duv[e] * a[0][e] * sg * (1.f - sg);
            }
            if (ok) {
                stg4(dag + row * 256 + 16 * ob + 4 * g, da);
                stg4(dag + row * 256 + 128 + 16 * ob + 4 * g, dg);
            }
            z[ob] = da;
            z[8 + ob] = dg;
            zmax = tx_absmax4(dg, tx_absmax4(da, zmax));
        }
        float zs, zinv;
        tx_pow2(tx_wave_max(zmax), zs, zinv);
        f16x8 zh[1][8], zl[1][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) split8(z[2 * m] * splat4(zs), z[2 * m + 1] * splat4(zs), zh[0][m], zl[0][m]);
        f32x4 dxn[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 acc[1] = {splat4(0.f)};
            lin_acc_x3<8, 1>(w1t + ob * 8 * 1024 + lane * 8, zh, zl, acc);
            dxn[ob] = acc[0] * splat4(zinv);
        }
        f32x4 dxh[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            dxh[kb] = dxn[kb] * *reinterpret_cast<const f32x4*>(&par_l[16 * kb + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1 += dxh[kb][r];
                s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
            }
        }
        const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
        f32x4 ca[4], cb[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (ok) {
                f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
                if (dres) dv = dv + ldg4(dres + row * 64 + 16 * kb + 4 * g);
                stg4(dx + row * 64 + 16 * kb + 4 * g, dv);
            }
            ca[kb] = ok ? dxn[kb] * xh[kb] : splat4(0.f);
            cb[kb] = ok ? dxn[kb] : splat4(0.f);
        }
        ln_tile_colsums(ca, cb, c, g, tile, g1c, dxc);                 // row `tile` of the [ceil(M / 16)][64] slabs
    }
}

// the conv module's pw1 images in the two 64 KB slots where the fp32 fragment images of pw1 lived
void cm_x3_pack(LaunchCtx ctx, const ConvModTrainParams& p, float* img_w1, float* img_w1t) {
    launch_pack_x3(ctx, "convmod_train_pack", PackX3Jobs{{{p.pw1_w, 256, 64, 64, 0, reinterpret_cast<_Float16*>(img_w1)},
                                                          {p.pw1_w, 64, 256, 64, 1, reinterpret_cast<_Float16*>(img_w1t)},
                                                          {}, {}}}, 2);
}
void cm_x3_pw1glu(LaunchCtx ctx, const float* x, long M, const float* img_w1, const ConvModTrainParams& p, float* u) {
    const int ntiles = (int)((M + 31) / 32);
    const int grid = (ntiles + 7) / 8 < 512 ? ((ntiles + 7) / 8 > 0 ? (ntiles + 7) / 8 : 1) : 512;
    LAUNCH(ctx, "convmod_train_fwd", (cm_pw1glu_x3_kernel<<<grid, 512, 0, ctx.stream>>>(
                                         x, M, reinterpret_cast<const _Float16*>(img_w1), p, u, ntiles)));
}
void cm_x3_bwd2(LaunchCtx ctx, const float* x, const float* du, long M, const float* img_w1, const float* img_w1t,
                const ConvModTrainParams& p, const float* dres, float* dx, float* dag, float* xn_out, float* g1c, float* dxc) {
    const int ntiles = (int)((M + 15) / 16);
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd2_x3_kernel<<<tx_grid(ntiles, TX_WAVES), TX_WAVES * 64, 0, ctx.stream>>>(
                                         x, du, M, reinterpret_cast<const _Float16*>(img_w1),
                                         reinterpret_cast<const _Float16*>(img_w1t), p, dres, dx, dag, xn_out, g1c, dxc, ntiles)));
}

// BatchNorm apply -> Swish -> pointwise 128 -> 64 + bias (+ residual).  LDS: pw2 image [4][4] = 32 KB.
__global__ __launch_bounds__(512) void cm_bn_swish_pw2_x3_kernel(const float* __restrict__ d, long M,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const _Float16* __restrict__ w2i, const float* __restrict__ b2,
                                                                 const float* __restrict__ res, float* __restrict__ y, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 w2[16384];
    __shared__ __attribute__((aligned(16))) float par_l[320];              // scale[128] | shift[128] | b2[64]
    stage_lds16<2048, 512>(w2i, w2);
    for (int i = threadIdx.x; i < 320; i += blockDim.x) par_l[i] = i < 128 ? scale[i] : (i < 256 ? shift[i - 128] : b2[i - 256]);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {
        long row[2];
        bool ok[2];
        f16x8 sh[2][4], sl[2][4];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 s[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kb = 2 * m + j;
                    const f32x4 dn = ldg4(d + row[tb] * 128 + 16 * kb + 4 * g) * *reinterpret_cast<const f32x4*>(&par_l[16 * kb + 4 * g]) +
                                     *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * kb + 4 * g]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[j][e] = swishf(dn[e]);
                }
                split8(s[0], s[1], sh[tb][m], sl[tb][m]);
            }
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&par_l[256 + 16 * ob + 4 * g]);
            f32x4 acc[2] = {bv, bv};
            lin_acc_x3<4, 2>(w2 + ob * 4 * 1024 + lane * 8, sh, sl, acc);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                if (res) acc[tb] = acc[tb] + ldg4(res + row[tb] * 64 + 16 * ob + 4 * g);
                if (ok[tb]) stg4(y + row[tb] * 64 + 16 * ob + 4 * g, acc[tb]);
            }
        }
    }
}

// backward, part 1: ds = pw2^T dy (dy scaled by the block's exact power of two), through Swish; ddn, s and the per-block
// partial sums of g2 = ddn dhat, ddn and dy (see cm_bwd1_kernel).  LDS: pw2^T image [8][2] = 32 KB.
__global__ __launch_bounds__(512) void cm_bwd1_x3_kernel(const float* __restrict__ dy, const float* __restrict__ d, long M,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const _Float16* __restrict__ w2ti, float* __restrict__ ddn,
                                                         float* __restrict__ s_out, float* __restrict__ g2c,
                                                         float* __restrict__ ddnc, float* __restrict__ dyc, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 w2t[16384];
    __shared__ __attribute__((aligned(16))) float par_l[512];              // mean | rstd | scale | shift
    stage_lds16<2048, 512>(w2ti, w2t);
    for (int i = threadIdx.x; i < 512; i += blockDim.x)
        par_l[i] = i < 128 ? mean[i] : (i < 256 ? rstd[i - 128] : (i < 384 ? scale[i - 256] : shift[i - 384]));
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {        // one 16-token block per trip
        const long t = (long)tile * 16 + c;
        const bool ok = t < M;
        const long row = ok ? t : M - 1;
        f32x4 dyv[4];
        float zmax = 0.f;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 v = ldg4(dy + row * 64 + 16 * ob + 4 * g);
            if (!ok) v = splat4(0.f);
            dyv[ob] = v;
            zmax = tx_absmax4(v, zmax);
            f32x4 sum;
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] = red_c_sum(v[r]);
            if (c == 0) stg4(dyc + (long)tile * 64 + 16 * ob + 4 * g, sum);
        }
        float zs, zinv;
        tx_pow2(tx_wave_max(zmax), zs, zinv);
        f16x8 zh[1][2], zl[1][2];
        split8(dyv[0] * splat4(zs), dyv[1] * splat4(zs), zh[0][0], zl[0][0]);
        split8(dyv[2] * splat4(zs), dyv[3] * splat4(zs), zh[0][1], zl[0][1]);
#pragma unroll 2
        for (int hb = 0; hb < 8; ++hb) {
            f32x4 ds[1] = {splat4(0.f)};
            lin_acc_x3<2, 1>(w2t + hb * 2048 + lane * 8, zh, zl, ds);
            const f32x4 dv = ldg4(d + row * 128 + 16 * hb + 4 * g);
            const f32x4 dhat = (dv - *reinterpret_cast<const f32x4*>(&par_l[16 * hb + 4 * g])) *
                               *reinterpret_cast<const f32x4*>(&par_l[128 + 16 * hb + 4 * g]);
            const f32x4 dn = dv * *reinterpret_cast<const f32x4*>(&par_l[256 + 16 * hb + 4 * g]) +
                             *reinterpret_cast<const f32x4*>(&par_l[384 + 16 * hb + 4 * g]);
            f32x4 o_ddn, o_s;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sg = sigmoidf_fast(dn[e]);
                o_s[e] = dn[e] * sg;
                o_ddn[e] = (ds[0][e] * zinv) * (sg * (1.f + dn[e] * (1.f - sg)));
            }
            if (ok) {
                stg4(ddn + row * 128 + 16 * hb + 4 * g, o_ddn);
                stg4(s_out + row * 128 + 16 * hb + 4 * g, o_s);
            }
            f32x4 ca = ok ? o_ddn * dhat : splat4(0.f), cb = ok ? o_ddn : splat4(0.f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ca[r] = red_c_sum(ca[r]);
                cb[r] = red_c_sum(cb[r]);
            }
            if (c == 0) {
                stg4(g2c + (long)tile * 128 + 16 * hb + 4 * g, ca);
                stg4(ddnc + (long)tile * 128 + 16 * hb + 4 * g, cb);
            }
        }
    }
}

// LN -> [to_q ; to_kv] (image [12][2] = 48 KB) -> q | k | v.  as_image: the attention cores' (hi, lo) operand image with
// the q part (output blocks 0 .. 3) pre-scaled by qscale (train.hip, AT_X3); else the fp32 projection.
__global__ __launch_bounds__(512) void at_qkv_x3_kernel(const float* __restrict__ x, long M, const _Float16* __restrict__ wi,
                                                        const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                        float* __restrict__ qkv, int as_image, float qscale, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 w[24576];
    __shared__ __attribute__((aligned(16))) float par_l[128];
    stage_lds16<3072, 512>(wi, w);
    for (int i = threadIdx.x; i < 128; i += blockDim.x) par_l[i] = i < 64 ? ln_w[i] : ln_b[i - 64];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {
        long row[2];
        bool ok[2];
        f16x8 xbh[2][2], xbl[2][2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const long t = ((long)tile * 2 + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 xh[4], xn[4];
            float rstd;
            tx_load_norm(x, row[tb], g, par_l, xh, rstd, xn);
            split8(xn[0], xn[1], xbh[tb][0], xbl[tb][0]);
            split8(xn[2], xn[3], xbh[tb][1], xbl[tb][1]);
        }
#pragma unroll 2
        for (int ob = 0; ob < 12; ++ob) {
            f32x4 acc[2] = {splat4(0.f), splat4(0.f)};
            lin_acc_x3<2, 2>(w + ob * 2048 + lane * 8, xbh, xbl, acc);
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                f32x4 v = acc[tb];
                if (as_image) {
                    if (ob < 4) v = v * splat4(qscale);
                    f16x4 h, l;
                    split4(v, h, l);
                    v = __builtin_bit_cast(f32x4, __builtin_shufflevector(h, l, 0, 4, 1, 5, 2, 6, 3, 7));
                }
                if (ok[tb]) stg4(qkv + row[tb] * 192 + 16 * ob + 4 * g, v);
            }
        }
    }
}

// dxn = [to_q ; to_kv]^T dqkv (image [4][6] = 48 KB; dqkv scaled by the block's exact power of two), LayerNorm backward,
// xn for the weight gradient, per-block partial sums of dgamma / dbeta
__global__ __launch_bounds__(512) void at_qkv_bwd_x3_kernel(const float* __restrict__ x, const float* __restrict__ dqkv, long M,
                                                            const _Float16* __restrict__ wti, const float* __restrict__ ln_w,
                                                            const float* __restrict__ ln_b, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ xn_out,
                                                            float* __restrict__ g1c, float* __restrict__ dxc, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wt[24576];
    __shared__ __attribute__((aligned(16))) float par_l[128];
    stage_lds16<3072, 512>(wti, wt);
    for (int i = threadIdx.x; i < 128; i += blockDim.x) par_l[i] = i < 64 ? ln_w[i] : ln_b[i - 64];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * 8 + wv; tile < ntiles; tile += gridDim.x * 8) {        // one 16-token block per trip
        const long t = (long)tile * 16 + c;
        const bool ok = t < M;
        const long row = ok ? t : M - 1;
        f32x4 z[12];
        float zmax = 0.f;
#pragma unroll
        for (int kb = 0; kb < 12; ++kb) {
            z[kb] = ldg4(dqkv + row * 192 + 16 * kb + 4 * g);
            if (!ok) z[kb] = splat4(0.f);
            zmax = tx_absmax4(z[kb], zmax);
        }
        float zs, zinv;
        tx_pow2(tx_wave_max(zmax), zs, zinv);
        f16x8 zh[1][6], zl[1][6];
#pragma unroll
        for (int m = 0; m < 6; ++m) split8(z[2 * m] * splat4(zs), z[2 * m + 1] * splat4(zs), zh[0][m], zl[0][m]);
        f32x4 xh[4], xn[4];
        float rstd;
        tx_load_norm(x, row, g, par_l, xh, rstd, xn);
        f32x4 dxn[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 acc[1] = {splat4(0.f)};
            lin_acc_x3<6, 1>(wt + ob * 6 * 1024 + lane * 8, zh, zl, acc);
            dxn[ob] = acc[0] * splat4(zinv);
        }
        f32x4 dxh[4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            dxh[kb] = dxn[kb] * *reinterpret_cast<const f32x4*>(&par_l[16 * kb + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1 += dxh[kb][r];
                s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
            }
        }
        const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
        f32x4 ca[4], cb[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (ok) {
                f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
                if (dres) dv = dv + ldg4(dres + row * 64 + 16 * kb + 4 * g);
                stg4(dx + row * 64 + 16 * kb + 4 * g, dv);
                stg4(xn_out + row * 64 + 16 * kb + 4 * g, xn[kb]);
            }
            ca[kb] = ok ? dxn[kb] * xh[kb] : splat4(0.f);
            cb[kb] = ok ? dxn[kb] : splat4(0.f);
        }
        ln_tile_colsums(ca, cb, c, g, tile, g1c, dxc);
    }
}

static int x3_grid8(int ntiles) { const int w = (ntiles + 7) / 8; return w < 512 ? (w > 0 ? w : 1) : 512; }
// pw2 (R 64, K 128) and pw2^T images: 32 KB each, in the slots of the fp32 fragment images (w2, w2t)
void cm_x3_pack_pw2(LaunchCtx ctx, const ConvModTrainParams& p, float* img_w2, float* img_w2t) {
    launch_pack_x3(ctx, "convmod_train_pack", PackX3Jobs{{{p.pw2_w, 64, 128, 128, 0, reinterpret_cast<_Float16*>(img_w2)},
                                                          {p.pw2_w, 128, 64, 128, 1, reinterpret_cast<_Float16*>(img_w2t)},
                                                          {}, {}}}, 2);
}
void cm_x3_bn_swish_pw2(LaunchCtx ctx, const float* d, long M, const float* scale, const float* shift, const float* img_w2,
                        const float* b2, const float* res, float* y) {
    const int ntiles = (int)((M + 31) / 32);
    LAUNCH(ctx, "convmod_train_fwd", (cm_bn_swish_pw2_x3_kernel<<<x3_grid8(ntiles), 512, 0, ctx.stream>>>(
                                         d, M, scale, shift, reinterpret_cast<const _Float16*>(img_w2), b2, res, y, ntiles)));
}
// ---------------------------------------------------------------------------------
// conv module backward, part 1 WITH the pointwise-2 weight gradient contracted on the chip (the default;
// CMGAN_CM_BWD1_FUSED=0 selects cm_bwd1_x3_kernel + a token-contraction launch).  Part 1 wrote s = Swish(BN(d)) [M,128]
// only so that dW_pw2 = dy^T s could read it back with dy: 1 KB per token of round trip + dy twice more.  Same design as
// ffn_train_bwd_aw_x3_kernel: 8 waves, wave w owns hidden channels 16 w .. 16 w + 15 (its column block of pw2 as a
// register-resident B operand, its [16 x 64] block of dW_pw2 as accumulators); per 32-token tile the block stages dy once
// (row-major planes: A operands of ds^T = dy W2; transposed planes: B operands of dW_pw2 += s^T dy; at the running exact
// power-of-two scale) and d transposed in fp32; ds^T lands as (hidden c) x (tokens 4 g + r of both token blocks), where s
// is computed - already the A operand of the weight gradient.  The BatchNorm partial sums (ddn dhat, ddn) per tile are a
// lane sum + one cross-group reduce instead of eight 16-lane reductions.  Out: ddn [M,128], slab blockIdx.x of dW_pw2
// [64][128], per-TILE (32 tokens) partial rows g2c / ddnc [tiles][128] and dyc [tiles][64].
// ---------------------------------------------------------------------------------
#define CB_PD 37                      // floats per row of the [hidden][token slot] plane of d (odd: conflict-free dword reads)
struct CbImg {
    _Float16 yh[32 * FA_PR], yl[32 * FA_PR];          // dy [token][out]
    _Float16 yth[64 * FA_PT], ytl[64 * FA_PT];        // dy [out][token slot]
    float dT[128 * CB_PD];                            // d  [hidden][token slot]
};
__global__ __launch_bounds__(512) void cm_bwd1_w_x3_kernel(const float* __restrict__ dy, const float* __restrict__ d, long M,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ w2, float* __restrict__ ddn,
                                                           float* __restrict__ g2c, float* __restrict__ ddnc,
                                                           float* __restrict__ dyc, float* __restrict__ part_w2, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cb_sm[];
    CbImg* const img = reinterpret_cast<CbImg*>(cb_sm);                                   // [2]
    float* const zmax_l = reinterpret_cast<float*>(cb_sm + 2 * sizeof(CbImg));            // [2][8]
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x;
    const int nloc = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int hu = 16 * wv + c;                                   // this lane's hidden channel
    // resident B operand: pw2 [64 out][128 hidden], slot (g, e) of k-step ks <-> out 32 ks + 8 g + e
    f16x8 w2h[2], w2l[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = w2[(long)(32 * ks + 8 * g + e) * 128 + hu];
            b[e] = w2[(long)(32 * ks + 8 * g + 4 + e) * 128 + hu];
        }
        split8(a, b, w2h[ks], w2l[ks]);
    }
    const float mu = mean[hu], rs = rstd[hu], sc_bn = scale[hu], sh_bn = shift[hu];
    // staging role: token tk = 4 wv + g of the tile; dy channels 4 c .. 4 c + 3, d channels 8 c .. 8 c + 7
    const int tk = 4 * wv + g;
    const int pos = 8 * ((tk & 15) >> 2) + 4 * (tk >> 4) + (tk & 3);          // split8 slot order of token tk
    auto rsrc = [](const void* base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t y_rs = rsrc(dy, M * 256), d_rs = rsrc(d, M * 512), n_rs = rsrc(ddn, M * 512);
    struct Raw { f32x4 y, d0, d1; bool ok; };
    auto load = [&](int i) __attribute__((always_inline)) {
        long tile = (long)blockIdx.x + (long)i * G;
        const bool have = i < nloc;
        tile = have ? tile : ntiles - 1;                          // past the end: a readable dummy, contributes nothing
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
        Raw r;
        r.ok = have && tk < rem;
        const unsigned trow = (unsigned)(tk < rem ? tk : rem - 1);
        r.y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(y_rs, trow * 256 + 16 * c, (unsigned)tile * 8192u, 0));
        r.d0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, trow * 512 + 32 * c, (unsigned)tile * 16384u, 0));
        r.d1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(d_rs, trow * 512 + 32 * c + 16, (unsigned)tile * 16384u, 0));
        if (!r.ok) r.y = splat4(0.f);
        return r;
    };
    auto post_max = [&](const Raw& r, int buf) __attribute__((always_inline)) {
        const float mx = tx_wave_max(tx_absmax4(r.y, 0.f));
        if (lane == 0) zmax_l[buf * 8 + wv] = mx;
    };
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    float s_run = 1.f, inv_run = 1.f;
    bool fresh = true;
    auto decide = [&](int buf) __attribute__((always_inline)) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, zmax_l[buf * 8 + k]);
        const float ms_ = m * s_run;
        if (m > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {
            tx_pow2(m, s_run, inv_run);
            fresh = false;
        }
        s_run = uni(s_run);
        inv_run = uni(inv_run);
    };
    auto write_images = [&](int buf, const Raw& r, float sc) __attribute__((always_inline)) {
        CbImg& I = img[buf];
        f16x4 h, l;
        split4(r.y * splat4(sc), h, l);
        *reinterpret_cast<f16x4*>(&I.yh[tk * FA_PR + 4 * c]) = h;
        *reinterpret_cast<f16x4*>(&I.yl[tk * FA_PR + 4 * c]) = l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.yth[(4 * c + e) * FA_PT + pos] = h[e];
            I.ytl[(4 * c + e) * FA_PT + pos] = l[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.dT[(8 * c + e) * CB_PD + pos] = r.d0[e];
            I.dT[(8 * c + 4 + e) * CB_PD + pos] = r.d1[e];
        }
    };
    f32x4 acc[4];                                                 // dW_pw2 [hidden 4 g + r][out 16 ob + c]
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = splat4(0.f);
    float sA = 1.f, invA = 1.f;
    auto consume = [&](int i, int buf, float sc, float inv) __attribute__((always_inline)) {
        const CbImg& I = img[buf];
        const long tile = (long)blockIdx.x + (long)i * G;
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
        f32x4 ds[2] = {splat4(0.f), splat4(0.f)};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int o = (16 * tb + c) * FA_PR + 32 * ks + 8 * g;
                const f16x8 zh = *reinterpret_cast<const f16x8*>(&I.yh[o]), zl = *reinterpret_cast<const f16x8*>(&I.yl[o]);
                ds[tb] = mfma32h(zh, w2h[ks], ds[tb]);
                ds[tb] = mfma32l(zh, w2l[ks], ds[tb]);
                ds[tb] = mfma32l(zl, w2h[ks], ds[tb]);
            }
        f32x4 sv[2];
        float ca = 0.f, cb = 0.f;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dv = I.dT[hu * CB_PD + 8 * g + 4 * tb + r];
                const float dhat = (dv - mu) * rs, dn = fmaf(dv, sc_bn, sh_bn), sg = sigmoidf_fast(dn);
                sv[tb][r] = dn * sg;
                const float o_ddn = (ds[tb][r] * inv) * (sg * fmaf(dn, 1.f - sg, 1.f));   // tokens past M: dy = 0 -> 0
                ca = fmaf(o_ddn, dhat, ca);
                cb += o_ddn;
                const int tok = 16 * tb + 4 * g + r;
                // a token past M gets an offset beyond num_records: the hardware drops the store (the range check sees the
                // per-lane offset only; the launcher keeps the tensor under 2 GB)
                const unsigned vo = tok < rem ? (unsigned)((tok * 128 + hu) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o_ddn), n_rs, vo, (unsigned)tile * 16384u, 0);
            }
        ca = red_g_sum(ca);
        cb = red_g_sum(cb);
        if (g == 0) {
            g2c[tile * 128 + hu] = ca;
            ddnc[tile * 128 + hu] = cb;
        }
        if (sc != sA) {
            const float ratio = sc * invA;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = acc[k] * splat4(ratio);
            sA = sc; invA = inv;
        }
        f16x8 sh, sl;
        split8(sv[0], sv[1], sh, sl);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const int o = (16 * ob + c) * FA_PT + 8 * g;
            const f16x8 zh = fa_ld8(&I.yth[o]), zl = fa_ld8(&I.ytl[o]);
            acc[ob] = mfma32h(sh, zh, acc[ob]);
            acc[ob] = mfma32l(sh, zl, acc[ob]);
            acc[ob] = mfma32l(sl, zh, acc[ob]);
            if (wv == ob) {                                       // db_pw2 partial of the tile: column sums of dy (hi + lo: 2^-22)
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += (float)zh[e] + (float)zl[e];
                sm = red_g_sum(sm) * inv;
                if (g == 0) dyc[tile * 64 + 16 * ob + c] = sm;
            }
        }
    };
    Raw pc = load(0);
    post_max(pc, 0);
    __syncthreads();
    decide(0);
    float s_cur = s_run, inv_cur = inv_run;
    write_images(0, pc, s_cur);
    pc = load(1);
    post_max(pc, 1);
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < nloc; ++i) {
        const int buf = i & 1;
        const Raw r2 = load(i + 2);
        decide(buf ^ 1);
        const float s_nxt = s_run, inv_nxt = inv_run;
        write_images(buf ^ 1, pc, s_nxt);
        consume(i, buf, s_cur, inv_cur);
        post_max(r2, buf);
        pc = r2;
        __syncthreads();
        s_cur = s_nxt; inv_cur = inv_nxt;
    }
    float* slab = part_w2 + (long)blockIdx.x * 8192;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        stg4(slab + (unsigned)((16 * k + c) * 128 + 16 * wv + 4 * g), acc[k] * splat4(invA));
}
// returns the number of slabs of dW_pw2 written (0: take cm_x3_bwd1 + the token contraction)
int cm_x3_bwd1_fused(LaunchCtx ctx, const float* dy, const float* d, long M, const float* mean, const float* rstd,
                     const float* scale, const float* shift, const float* w2raw, float* ddn, float* g2c, float* ddnc, float* dyc,
                     float* part_w2) {
    if (M * 512 >= (1l << 31)) return 0;                          // 32-bit offsets, 2^31 marks tokens past M
    const void* fn = reinterpret_cast<const void*>(&cm_bwd1_w_x3_kernel);
    const size_t lds = 2 * sizeof(CbImg) + 16 * sizeof(float);
    static std::mutex mu;
    static std::map<int, bool> optin;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = optin.find(dev);
        if (it == optin.end()) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) (void)hipGetLastError();
            it = optin.emplace(dev, e == hipSuccess).first;
        }
        if (!it->second) return 0;
    }
    const int ntiles = (int)((M + 31) / 32);
    const int grid = ntiles < 512 ? ntiles : 512;                 // 114 VGPRs, 74 KB of LDS: two blocks per CU; the slab buffer
                                                                  // (WG_SPLIT x 16384 floats) holds 512 slabs of 8192
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd1_w_x3_kernel<<<grid, 512, lds, ctx.stream>>>(dy, d, M, mean, rstd, scale, shift, w2raw,
                                                                                           ddn, g2c, ddnc, dyc, part_w2, ntiles)));
    return grid;
}
// ---------------------------------------------------------------------------------
// conv module backward, part 2 split like the FeedForward's (the default; CMGAN_CM_BWD2_FUSED=0 selects cm_bwd2_x3_kernel
// + a token-contraction launch):  part A'' (here) = LayerNorm, a / gate recomputed, GLU backward, [da ; dg] stored, the
// pointwise-1 weight gradient contracted on the chip;  part B = ffn_train_bwd_b_x3_kernel AS IT IS (dxn = pw1^T [da ; dg],
// LayerNorm backward, residual, per-tile partial sums - pw1 has the FeedForward's W1 shape).  cm_bwd2_x3_kernel wrote
// [da ; dg] [M,256] and xn [M,64] for the token contraction to read back: 1.3 KB per token of the 4.9 KB the pair moved.
// Wave w owns GLU channels 16 w .. 16 w + 15: rows 16 w + c (a) and 128 + 16 w + c (gate) of pw1 as register-resident B
// operands, their two [16 x 64] blocks of dW_pw1 as accumulators.  du [M,128] is a gradient: staged (transposed, fp32) at the
// running exact power-of-two scale; da, dg are then at that scale for the weight-gradient products and stored times its
// inverse.  Staging / pipeline as ffn_train_bwd_aw_x3_kernel.
// ---------------------------------------------------------------------------------
struct CcImg {
    _Float16 xnh[32 * FA_PR], xnl[32 * FA_PR];        // xn [token][channel]
    _Float16 xth[64 * FA_PT], xtl[64 * FA_PT];        // xn [channel][token slot]
    float duT[128 * CB_PD];                           // du (scaled) [GLU channel][token slot]
};
__global__ __launch_bounds__(512) void cm_bwd2_aw_x3_kernel(const float* __restrict__ x, const float* __restrict__ du, long M,
                                                            ConvModTrainParams p, float* __restrict__ dag,
                                                            float* __restrict__ o_dhmax, float* __restrict__ part_w1, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cc_sm[];
    CcImg* const img = reinterpret_cast<CcImg*>(cc_sm);                                   // [2]
    float* const zmax_l = reinterpret_cast<float*>(cc_sm + 2 * sizeof(CcImg));            // [2][8]
    unsigned* const dhmax_l = reinterpret_cast<unsigned*>(zmax_l + 16);                   // [2]
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x;
    const int nloc = (ntiles - (int)blockIdx.x + G - 1) / G;
    if (threadIdx.x < 2) dhmax_l[threadIdx.x] = 0u;
    // resident B operands: hb 0 = row 16 wv + c of pw1 (a), hb 1 = row 128 + 16 wv + c (gate); slot (g, e) of k-step ks <->
    // input channel 32 ks + 8 g + e
    f16x8 w1h[2][2], w1l[2][2];
    float b1c[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int hu = 128 * hb + 16 * wv + c;
        b1c[hb] = p.pw1_b[hu];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wp = p.pw1_w + (long)hu * 64 + 32 * ks + 8 * g;
            split8(ldg4(wp), ldg4(wp + 4), w1h[hb][ks], w1l[hb][ks]);
        }
    }
    const int tk = 4 * wv + g;                                    // staging role: token tk, x channels 4 c .., du channels 8 c ..
    const int pos = 8 * ((tk & 15) >> 2) + 4 * (tk >> 4) + (tk & 3);
    const f32x4 gam = ldg4(p.ln_w + 4 * c), bet = ldg4(p.ln_b + 4 * c);
    auto rsrc = [](const void* base, long bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (unsigned)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t x_rs = rsrc(x, M * 256), u_rs = rsrc(du, M * 512), o_rs = rsrc(dag, M * 1024);
    struct Raw { f32x4 x, u0, u1; bool ok; };
    struct Proc { f32x4 xn, u0, u1; };
    auto load = [&](int i) __attribute__((always_inline)) {
        long tile = (long)blockIdx.x + (long)i * G;
        const bool have = i < nloc;
        tile = have ? tile : ntiles - 1;
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
        Raw r;
        r.ok = have && tk < rem;
        const unsigned trow = (unsigned)(tk < rem ? tk : rem - 1);
        r.x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, trow * 256 + 16 * c, (unsigned)tile * 8192u, 0));
        r.u0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rs, trow * 512 + 32 * c, (unsigned)tile * 16384u, 0));
        r.u1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rs, trow * 512 + 32 * c + 16, (unsigned)tile * 16384u, 0));
        return r;
    };
    auto process = [&](const Raw& r, Proc& q, int buf) __attribute__((always_inline)) {
        const float mean = red_c_sum((r.x[0] + r.x[1]) + (r.x[2] + r.x[3])) * (1.0f / 64.0f);
        const f32x4 d = r.x - splat4(mean);
        const float rstd = rsqrtf(red_c_sum(fmaf(d[0], d[0], d[1] * d[1]) + fmaf(d[2], d[2], d[3] * d[3])) * (1.0f / 64.0f) + CMGAN_EPS);
        q.xn = d * splat4(rstd) * gam + bet;
        q.u0 = r.ok ? r.u0 : splat4(0.f);
        q.u1 = r.ok ? r.u1 : splat4(0.f);
        const float mx = tx_wave_max(tx_absmax4(q.u1, tx_absmax4(q.u0, 0.f)));
        if (lane == 0) zmax_l[buf * 8 + wv] = mx;
    };
    auto uni = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    float s_run = 1.f, inv_run = 1.f;
    bool fresh = true;
    auto decide = [&](int buf) __attribute__((always_inline)) {
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, zmax_l[buf * 8 + k]);
        const float ms_ = m * s_run;
        if (m > 0.f && (fresh || ms_ > 8192.f || ms_ < 0.25f)) {
            tx_pow2(m, s_run, inv_run);
            fresh = false;
        }
        s_run = uni(s_run);
        inv_run = uni(inv_run);
    };
    auto write_images = [&](int buf, const Proc& q, float sc) __attribute__((always_inline)) {
        CcImg& I = img[buf];
        f16x4 h, l;
        split4(q.xn, h, l);
        *reinterpret_cast<f16x4*>(&I.xnh[tk * FA_PR + 4 * c]) = h;
        *reinterpret_cast<f16x4*>(&I.xnl[tk * FA_PR + 4 * c]) = l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.xth[(4 * c + e) * FA_PT + pos] = h[e];
            I.xtl[(4 * c + e) * FA_PT + pos] = l[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            I.duT[(8 * c + e) * CB_PD + pos] = q.u0[e] * sc;
            I.duT[(8 * c + 4 + e) * CB_PD + pos] = q.u1[e] * sc;
        }
    };
    f32x4 acc[2][4];                                              // dW_pw1 [row 128 hb + 16 wv + 4 g + r][in 16 cb + c]
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[hb][k] = splat4(0.f);
    float sA = 1.f, invA = 1.f;
    auto consume = [&](int i, int buf, float sc, float inv) __attribute__((always_inline)) {
        const CcImg& I = img[buf];
        const long tile = (long)blockIdx.x + (long)i * G;
        const int rem = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
        f32x4 ag[2][2];                                           // [hb: a | gate][tb]   (the biases are added below)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) ag[hb][tb] = splat4(0.f);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
                const int o = (16 * tb + c) * FA_PR + 32 * ks + 8 * g;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&I.xnh[o]), al = *reinterpret_cast<const f16x8*>(&I.xnl[o]);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) ag[hb][tb] = mfma32h(ah, w1h[hb][ks], ag[hb][tb]);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) ag[hb][tb] = mfma32l(ah, w1l[hb][ks], ag[hb][tb]);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) ag[hb][tb] = mfma32l(al, w1h[hb][ks], ag[hb][tb]);
            }
        f32x4 dav[2], dgv[2];
        float dhm = 0.f;
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float duv = I.duT[(16 * wv + c) * CB_PD + 8 * g + 4 * tb + r];          // at scale sc; 0 past M
                const float av = ag[0][tb][r] + b1c[0], sg = sigmoidf_fast(ag[1][tb][r] + b1c[1]);
                const float da = duv * sg, dg = duv * av * sg * (1.f - sg);
                dav[tb][r] = da;
                dgv[tb][r] = dg;
                const float dat = da * inv, dgt = dg * inv;
                dhm = fmaxf(dhm, fmaxf(fabsf(dat), fabsf(dgt)));
                const int tok = 16 * tb + 4 * g + r;
                // (a token past M gets an offset beyond num_records: the store is dropped; the launcher keeps dag under 2 GB)
                const unsigned vo = tok < rem ? (unsigned)((tok * 256 + 16 * wv + c) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dat), o_rs, vo, (unsigned)tile * 32768u, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dgt), o_rs, vo + 512u, (unsigned)tile * 32768u, 0);
            }
        dhm = tx_wave_max(dhm);
        if (lane == 0) atomicMax(&dhmax_l[i & 1], __float_as_uint(dhm));
        if (sc != sA) {
            const float ratio = sc * invA;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[hb][k] = acc[hb][k] * splat4(ratio);
            sA = sc; invA = inv;
        }
        f16x8 dh_[2], dl_[2];
        split8(dav[0], dav[1], dh_[0], dl_[0]);
        split8(dgv[0], dgv[1], dh_[1], dl_[1]);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int o = (16 * cb + c) * FA_PT + 8 * g;
            const f16x8 xh = fa_ld8(&I.xth[o]), xl = fa_ld8(&I.xtl[o]);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) acc[hb][cb] = mfma32h(dh_[hb], xh, acc[hb][cb]);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) acc[hb][cb] = mfma32l(dh_[hb], xl, acc[hb][cb]);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) acc[hb][cb] = mfma32l(dl_[hb], xh, acc[hb][cb]);
        }
    };
    Raw rw = load(0);
    Proc pc;
    process(rw, pc, 0);
    __syncthreads();
    decide(0);
    float s_cur = s_run, inv_cur = inv_run;
    write_images(0, pc, s_cur);
    rw = load(1);
    process(rw, pc, 1);
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < nloc; ++i) {
        const int buf = i & 1;
        if (wv == 0 && lane == 0 && i > 0) {
            o_dhmax[(long)blockIdx.x + (long)(i - 1) * G] = __uint_as_float(dhmax_l[buf ^ 1]);
            dhmax_l[buf ^ 1] = 0u;
        }
        const Raw r2 = load(i + 2);
        decide(buf ^ 1);
        const float s_nxt = s_run, inv_nxt = inv_run;
        write_images(buf ^ 1, pc, s_nxt);
        consume(i, buf, s_cur, inv_cur);
        process(r2, pc, buf);
        __syncthreads();
        s_cur = s_nxt; inv_cur = inv_nxt;
    }
    if (wv == 0 && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        o_dhmax[(long)blockIdx.x + (long)(nloc - 1) * G] = __uint_as_float(dhmax_l[(nloc - 1) & 1]);
    float* slab = part_w1 + (long)blockIdx.x * 16384;
    int ce = c, ge = g;
    asm volatile("" : "+v"(ce), "+v"(ge));
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                slab[(unsigned)((128 * hb + 16 * wv + 4 * ge + r) * 64 + 16 * k + ce)] = acc[hb][k][r] * invA;
}
// part A'' then the FeedForward's part B on pw1's images.  Returns the number of slabs of dW_pw1 written (0: the caller takes
// cm_x3_bwd2 + the token contraction).  o_g1 / o_dxn / o_dhc: per-32-token-tile partial rows ([tiles][64], [tiles][64],
// [tiles][256] = the pw1 bias gradient's partials); dhmax: one float per tile.
int cm_x3_bwd2_fused(LaunchCtx ctx, const float* x, const float* du, long M, const float* img_w1t, const ConvModTrainParams& p,
                     const float* dres, float* dx, float* dag, float* o_g1, float* o_dxn, float* o_dhc, float* dhmax,
                     float* part_w1) {
    if (M * 1024 >= (1l << 31)) return 0;
    const void* fn = reinterpret_cast<const void*>(&cm_bwd2_aw_x3_kernel);
    const size_t lds = 2 * sizeof(CcImg) + 18 * sizeof(float);
    static std::mutex mu;
    static std::map<int, bool> optin;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = optin.find(dev);
        if (it == optin.end()) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) (void)hipGetLastError();
            it = optin.emplace(dev, e == hipSuccess).first;
        }
        if (!it->second) return 0;
    }
    const int ntiles = (int)((M + 31) / 32);
    const int grid = ntiles < 256 ? ntiles : 256;
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd2_aw_x3_kernel<<<grid, 512, lds, ctx.stream>>>(x, du, M, p, dag, dhmax, part_w1, ntiles)));
    const FfnTrainParams lnp{p.ln_w, p.ln_b, nullptr, nullptr, nullptr, nullptr};       // part B reads gamma / beta only
    const int gridb = (ntiles + 7) / 8 < 512 ? ((ntiles + 7) / 8 > 0 ? (ntiles + 7) / 8 : 1) : 512;
    LAUNCH(ctx, "convmod_train_bwd", (ffn_train_bwd_b_x3_kernel<<<gridb, 512, 0, ctx.stream>>>(
                                         x, dag, dhmax, M, reinterpret_cast<const _Float16*>(img_w1t), lnp, dres, dx, o_g1, o_dxn,
                                         o_dhc, ntiles)));
    return grid;
}
void cm_x3_bwd1(LaunchCtx ctx, const float* dy, const float* d, long M, const float* mean, const float* rstd, const float* scale,
                const float* shift, const float* img_w2t, float* ddn, float* s_out, float* g2c, float* ddnc, float* dyc) {
    const int ntiles = (int)((M + 15) / 16);
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd1_x3_kernel<<<x3_grid8(ntiles), 512, 0, ctx.stream>>>(
                                         dy, d, M, mean, rstd, scale, shift, reinterpret_cast<const _Float16*>(img_w2t), ddn, s_out,
                                         g2c, ddnc, dyc, ntiles)));
}
// [to_q ; to_kv] (R 192, K 64) and its transpose: 48 KB each, in the slots of the fp32 fragment images
void at_x3_pack(LaunchCtx ctx, const float* wraw, float* img_w, float* img_wt) {
    launch_pack_x3(ctx, "attn_train_pack", PackX3Jobs{{{wraw, 192, 64, 64, 0, reinterpret_cast<_Float16*>(img_w)},
                                                       {wraw, 64, 192, 64, 1, reinterpret_cast<_Float16*>(img_wt)},
                                                       {}, {}}}, 2);
}
void at_x3_qkv(LaunchCtx ctx, const float* x, long M, const float* img_w, const float* ln_w, const float* ln_b, float* qkv,
               int as_image, float qscale) {
    const int ntiles = (int)((M + 31) / 32);
    LAUNCH(ctx, "attn_train_fwd", (at_qkv_x3_kernel<<<x3_grid8(ntiles), 512, 0, ctx.stream>>>(
                                      x, M, reinterpret_cast<const _Float16*>(img_w), ln_w, ln_b, qkv, as_image, qscale, ntiles)));
}
void at_x3_qkv_bwd(LaunchCtx ctx, const float* x, const float* dqkv, long M, const float* img_wt, const float* ln_w,
                   const float* ln_b, const float* dres, float* dx, float* xn_out, float* g1c, float* dxc) {
    const int ntiles = (int)((M + 15) / 16);
    LAUNCH(ctx, "attn_train_bwd", (at_qkv_bwd_x3_kernel<<<x3_grid8(ntiles), 512, 0, ctx.stream>>>(
                                      x, dqkv, M, reinterpret_cast<const _Float16*>(img_wt), ln_w, ln_b, dres, dx, xn_out, g1c, dxc,
                                      ntiles)));
}
