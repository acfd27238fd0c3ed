// conformer_x3.hip - the ConformerBlock kernels of conformer.hip on the f16 matrix pipe
// with 3-term split products ("x3" mode, common.hip.h).  Same decomposition and the same
// transposed-chain trick; what changes for a ~5x faster matrix pipe:
//   * weight images live in LDS (one cooperative copy per persistent 512-thread block)
//     instead of being streamed from L2 by every wave
//   * attention packs [hi | lo] of K (resp. E) along the 32-wide contraction so that
//     d = 16 costs two MFMAs per 16x16 tile and yields all four split terms
//   * Q / K are exchanged as row-major fp16 hi/lo rows, V as ready-made A-operand images
//     (transposed through 1 KB of wave-private LDS in the producer)
#include "kernels.h"
#include <stdlib.h>
#ifndef FFN32_DEFAULT
#define FFN32_DEFAULT 1      // 1: the FeedForward branches run ffn32_x3_kernel (32x32x16 MFMAs); 0: ffn_x3_kernel
#endif

namespace X3_NS {

#define XNTB 2
#define XWAVES 8
#ifndef FFN_WAVES
#define FFN_WAVES 12          // waves per FeedForward block (one block per CU: 128 KB of weight images); three waves
                              // per SIMD at 150 VGPRs measured 7 % faster than two, a fourth would spill
#endif
#ifdef X3_SINGLE
#define FFN_POST_WAVES 8      // the single-product twin of the FINAL variant wants 194 VGPRs: two waves per SIMD, no scratch
#else
#define FFN_POST_WAVES FFN_WAVES
#endif

__device__ __forceinline__ float swish_x(float hp) { return swish_scaled(hp); }

// LN'd input rows -> B operands (hi/lo) for K = 64 (two k32 blocks)
__device__ __forceinline__ void ln_split(const f32x4 (&x)[4], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
    float mean, rstd;
    ln_stats(x, mean, rstd);
    f32x4 xh[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) xh[kb] = (x[kb] - splat4(mean)) * splat4(rstd);
    split8(xh[0], xh[1], bh[0], bl[0]);
    split8(xh[2], xh[3], bh[1], bl[1]);
}

// ---------------------------------------------------------------------------------
// FeedForward (+ post LayerNorm + TSCB residual when FINAL), see ffn_kernel.
// LDS: W1 image [16][2] + W2 image [4][8] = 128 KB.
// ---------------------------------------------------------------------------------
template <bool FINAL, int TNTB, int TWAVES>   // token blocks per wave, waves per block
__global__ __launch_bounds__(TWAVES * 64) void ffn_x3_kernel(const float* xin, float* xout, const float* x0,
                                                     const float* __restrict__ post_gb,
                                                     const _Float16* __restrict__ w1i, const float* __restrict__ b1,
                                                     const _Float16* __restrict__ w2i, const float* __restrict__ b2,
                                                     long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];          // 128 KB
    __shared__ __attribute__((aligned(16))) float bias_l[320];             // b1[256] | b2[64]
    _Float16* w1 = wlds;                 // 16*2*1024 halfs
    _Float16* w2 = wlds + 32768;         // 4*8*1024 halfs
    stage_lds16<4096, TWAVES * 64>(w1i, w1);
    stage_lds16<4096, TWAVES * 64>(w2i, w2);
    for (int i = threadIdx.x; i < 320; i += blockDim.x) bias_l[i] = i < 256 ? b1[i] : b2[i - 256];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;

// hidden units 32*M2 .. 32*M2+31 (two 16-blocks) for both token blocks: bias + W1 x (3 products)
#define FFN_GEMM1(M2, H)                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
        const int hb = 2 * (M2) + j;                                                                  \
        const f32x4 bias = *reinterpret_cast<const f32x4*>(&bias_l[16 * hb + 4 * g]);                 \
        _Pragma("unroll") for (int tb = 0; tb < TNTB; ++tb) H[j][tb] = bias;                          \
        lin_acc_x3<2, TNTB>(w1 + hb * 2048 + lane * 8, xbh, xbl, H[j]);                               \
    }

#pragma unroll 1
    for (int tile = blockIdx.x * TWAVES + wv; tile < ntiles; tile += gridDim.x * TWAVES) {
        long row[TNTB];
        bool ok[TNTB];
        f16x8 xbh[TNTB][2], xbl[TNTB][2];
        f32x4 y[TNTB][4];
#pragma unroll
        for (int tb = 0; tb < TNTB; ++tb) {
            const long t = ((long)tile * TNTB + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 x[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) x[kb] = ldg4(xin + row[tb] * 64 + 16 * kb + 4 * g);
            ln_split(x, xbh[tb], xbl[tb]);
            // The residual and the second bias are the INITIAL VALUE of the output accumulators: the B-fragment a lane
            // loaded for k-block kb is exactly its C-fragment of output block ob = kb, and y is live through both
            // GEMMs anyway.  (Re-reading the row in the epilogue - "an L2 hit" - missed 77 % of the time: 25 MB of rows
            // in flight per launch against 32 MB of L2 also holding the output; 471 MB fetched for a 266 MB input.)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) y[tb][ob] = x[ob] + *reinterpret_cast<const f32x4*>(&bias_l[256 + 16 * ob + 4 * g]);
        }

        // software pipeline over the 8 hidden k32-blocks: GEMM1(m2+1) is issued before the
        // Swish/split of block m2, so its MFMAs overlap that VALU work inside one wave
        f32x4 hcur[2][TNTB];
        FFN_GEMM1(0, hcur)
#pragma unroll 1
        for (int m2 = 0; m2 < 8; ++m2) {
            f32x4 hnext[2][TNTB];
            if (m2 < 7) { FFN_GEMM1(m2 + 1, hnext) }
            f16x8 hh[TNTB], hl[TNTB];
#pragma unroll
            for (int tb = 0; tb < TNTB; ++tb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hcur[0][tb][r] = swish_x(hcur[0][tb][r]);
                    hcur[1][tb][r] = swish_x(hcur[1][tb][r]);
                }
                split8(hcur[0][tb], hcur[1][tb], hh[tb], hl[tb]);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const _Float16* wp = w2 + (ob * 8 + m2) * 1024 + lane * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
                const f16x8 al = *reinterpret_cast<const f16x8*>(wp + 512);
#pragma unroll
                for (int tb = 0; tb < TNTB; ++tb) y[tb][ob] = mfma32h(ah, hh[tb], y[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < TNTB; ++tb) y[tb][ob] = mfma32l(ah, hl[tb], y[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < TNTB; ++tb) y[tb][ob] = mfma32l(al, hh[tb], y[tb][ob]);
            }
            if (m2 < 7) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int tb = 0; tb < TNTB; ++tb) hcur[j][tb] = hnext[j][tb];
            }
        }
#pragma unroll
        for (int tb = 0; tb < TNTB; ++tb) {
            if (FINAL) {
                float mean, rstd;
                ln_stats(y[tb], mean, rstd);
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    const f32x4 gm = ldg4(post_gb + 16 * ob + 4 * g);
                    const f32x4 bt = ldg4(post_gb + 64 + 16 * ob + 4 * g);
                    y[tb][ob] = (y[tb][ob] - splat4(mean)) * splat4(rstd) * gm + bt;
                    if (x0) y[tb][ob] += ldg4(x0 + row[tb] * 64 + 16 * ob + 4 * g);
                }
            }
            if (ok[tb]) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) stg4(xout + row[tb] * 64 + 16 * ob + 4 * g, y[tb][ob]);
            }
        }
    }
}

#ifndef XCD_ORDER
#define XCD_ORDER 1          // XCD-contiguous work order for the depthwise kernel and the attention (attn32_x3.hip); 0 = dispatch order
#endif

// ---------------------------------------------------------------------------------
// conv module part 1: LN -> pointwise 64->256 -> GLU.   LDS: [16][2] image = 64 KB.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pw1glu_x3_kernel(const float* __restrict__ x, float* __restrict__ u,
                                                        const _Float16* __restrict__ wi,
                                                        const float* __restrict__ b, long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[32768];          // 64 KB
    __shared__ __attribute__((aligned(16))) float bias_l[256];
    stage_lds16<4096, 512>(wi, wlds);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) bias_l[i] = b[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * XWAVES + wv; tile < ntiles; tile += gridDim.x * XWAVES) {
        long row[XNTB];
        bool ok[XNTB];
        f16x8 bh[XNTB][2], bl[XNTB][2];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
            const long t = ((long)tile * XNTB + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 xr[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xr[kb] = ldg4(x + row[tb] * 64 + 16 * kb + 4 * g);
            ln_split(xr, bh[tb], bl[tb]);
        }
#pragma unroll 2
        for (int ob = 0; ob < 8; ++ob) {
            f32x4 aa[XNTB], ag[XNTB];
            const f32x4 ba = *reinterpret_cast<const f32x4*>(&bias_l[16 * ob + 4 * g]);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(&bias_l[128 + 16 * ob + 4 * g]);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) { aa[tb] = ba; ag[tb] = bg; }
            lin_acc_x3<2, XNTB>(wlds + ob * 2048 + lane * 8, bh, bl, aa);
            lin_acc_x3<2, XNTB>(wlds + (ob + 8) * 2048 + lane * 8, bh, bl, ag);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) {
                if (ok[tb]) {
                    f32x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = aa[tb][e] * sigmoidf_fast(ag[tb][e]);
                    stg4(u + row[tb] * 128 + 16 * ob + 4 * g, r);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------
// conv module parts 2+3 fused: depthwise Conv1d k=31 (+folded BatchNorm) -> Swish ->
// pointwise 128->64 + bias + residual (conformer.py:165-170, 219).  The [M,128] depthwise
// output (1 GB written + read per conformer at B=32 when done as two kernels) never leaves the
// CU: a block owns 32 consecutive positions of one sequence, stages the (32+30) x 128 GLU
// tile in LDS, runs the depthwise taps channel-per-thread (34-tap sliding window, as
// dwconv_kernel), writes Swish(v) to LDS already split into fp16 hi/lo in B-operand order,
// and the four waves finish with the 128->64 product on the matrix pipe.
// The 32 KB pointwise weight image is fetched straight into registers at kernel start (its
// consumers are two barriers away, so the L2 latency is free).  Row pitch of the v tile is
// 288 B: conflict-free for the ds_read_b128 of 16 consecutive rows.
// ---------------------------------------------------------------------------------
#define DP_TL 32
#define DP_K 31
#define DP_VS 144                 // halfs per v-tile row (128 used)
// ---------------------------------------------------------------------------------
// Sliding-window form of dwpw2_x3_kernel (the default): a block owns DS_SEG consecutive 32-position tiles of
// one sequence and walks them in order.  The 62-row u window lives in LDS; after a tile its last 30 rows are
// moved to the front and only the 32 NEW rows are fetched - into registers one tile ahead, so the HBM latency
// of tile t+1 is covered by the depthwise arithmetic of tile t.  Compared with one block per tile:
//   * u is read (128 + 30) / 128 = 1.23x per segment instead of 62 / 32 = 1.94x (a frequency-axis sequence,
//     L = 101, is a single segment: exactly 1.0x),
//   * the 32 KB pointwise weight image and the 31 depthwise taps are fetched once per segment, not per tile,
//   * the depthwise window slides over all 16 outputs of a thread (46 LDS reads per 496 FMAs instead of 136).
// Work items are (sequence, segment) in XCD-contiguous order so the 30 halo rows between two segments of a
// sequence are an L2 hit.
// ---------------------------------------------------------------------------------
#ifndef DS_SEG
#define DS_SEG 4
#endif
#ifndef DS_TOEPLITZ
#define DS_TOEPLITZ 1        // 1: dwpw2t_x3_kernel (depthwise taps as banded-Toeplitz operands of the 4x4x4 MFMA); 0: this kernel
#endif
#if !DS_TOEPLITZ              // the VALU kernel is an A/B build only (-DDS_TOEPLITZ=0): it is not in the product library
#ifndef DS_OCC
#define DS_OCC 2             // blocks (= waves per SIMD) the register allocation is sized for
#endif
#ifndef DS_WRES
#define DS_WRES (DS_OCC < 3) // pointwise operands resident in registers for the whole segment (1) or fetched per tile (0)
#endif
__global__ __launch_bounds__(256, DS_OCC) void dwpw2s_x3_kernel(float* __restrict__ x, const float* __restrict__ u,
                                                        const float* __restrict__ dw_w,
                                                        const float* __restrict__ dw_b,
                                                        const _Float16* __restrict__ w2i,
                                                        const float* __restrict__ b2, TokMap m, int nseq, int nsegs) {
    constexpr int ROWS = DP_TL + DP_K - 1;                       // 62
    __shared__ __attribute__((aligned(16))) float utile[ROWS * 128];
    __shared__ __attribute__((aligned(16))) _Float16 vth[DP_TL * DP_VS];
    __shared__ __attribute__((aligned(16))) _Float16 vtl[DP_TL * DP_VS];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4, wv = tid >> 6;
    const long item = XCD_ORDER ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const int n = (int)(item / nsegs), seg = (int)(item - (long)n * nsegs);
    if (n >= nseq) return;                                       // padding blocks of the rounded-up grid
    const int l_begin = seg * DS_SEG * DP_TL;
    const int l_end = l_begin + DS_SEG * DP_TL < m.L ? l_begin + DS_SEG * DP_TL : m.L;
    const int ntiles = (l_end - l_begin + DP_TL - 1) / DP_TL;
    const long nbase = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride;

    // pointwise operands of this wave: token block tb, output blocks ob0, ob0 + 1
    const int tb = wv >> 1, ob0 = (wv & 1) * 2;
#if DS_WRES
    f16x8 ah[2][4], al[2][4];                                    // held for the whole segment (64 VGPRs)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const _Float16* wp = w2i + ((ob0 + o) * 4 + mm) * 1024 + lane * 8;
            ah[o][mm] = *reinterpret_cast<const f16x8*>(wp);
            al[o][mm] = *reinterpret_cast<const f16x8*>(wp + 512);
        }
#endif
#if DS_WRES
    f32x4 bias2[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) bias2[o] = ldg4(b2 + 16 * (ob0 + o) + 4 * g);
#endif

    // window row r of the tile at l0 <-> sequence position l0 - 15 + r; rows outside [0, L) are the conv's zero padding
    auto load_row4 = [&](int l0, int rr, int qd) -> f32x4 {
        const int l = l0 - (DP_K / 2) + rr;
        const int lc = l < 0 ? 0 : (l < m.L ? l : m.L - 1);
        f32x4 v = ldg4(u + (nbase + (long)lc * m.lstride) * 128 + qd * 4);
        if (l < 0 || l >= m.L) v = splat4(0.f);
        return v;
    };
    {
        constexpr int NLD = (ROWS * 32 + 255) / 256;             // 8: all loads first (see stage_lds16)
        f32x4 stg[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            stg[k] = load_row4(l_begin, i < ROWS * 32 ? (i >> 5) : ROWS - 1, i & 31);
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            if (i < ROWS * 32) *reinterpret_cast<f32x4*>(&utile[(i >> 5) * 128 + (i & 31) * 4]) = stg[k];
        }
    }
    const int chn = tid & 127, sub = tid >> 7;
    float wt[DP_K];
#pragma unroll
    for (int t = 0; t < DP_K; ++t) wt[t] = dw_w[t * 128 + chn];
    const float bias = dw_b[chn];
    // B-operand order inside a row: channel 32m + 16h + 4gq + r  ->  32m + 8gq + 4h + r
    const int vcol = (chn & ~31) + ((chn >> 2) & 3) * 8 + ((chn >> 4) & 1) * 4 + (chn & 3);
    const float* ucol = utile + sub * 16 * 128 + chn;
    __syncthreads();

#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        const int l0 = l_begin + t * DP_TL;
        const bool has_next = t + 1 < ntiles;
        // ---- prefetch: the 32 new rows of the next tile, and this tile's residual rows for the epilogue ----
        f32x4 nxt[4];
        if (has_next) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k;
                nxt[k] = load_row4(l0 + DP_TL, DP_K - 1 + (i >> 5), i & 31);
            }
        }
        const int lrow = l0 + 16 * tb + c;
        const bool live = lrow < m.L;
        float* xr = x + (nbase + (long)(live ? lrow : m.L - 1) * m.lstride) * 64;
        f32x4 xold[2];
#if DS_WRES
#pragma unroll
        for (int o = 0; o < 2; ++o) xold[o] = ldg4(xr + 16 * (ob0 + o) + 4 * g);
#endif

        // ---- depthwise: 16 outputs of channel chn from a 46-row sliding window ----
        // (a half tile that lies entirely beyond the sequence end is skipped: 16 of the 128 slots of a frequency-axis
        // sequence, L = 101; its v rows keep stale values that feed only outputs which are never stored)
        if (l0 + sub * 16 < m.L) {
            float acc[16];
#pragma unroll
            for (int oo = 0; oo < 16; ++oo) acc[oo] = bias;
#pragma unroll
            for (int kk = 0; kk < 16 + DP_K - 1; ++kk) {
                const float uv = ucol[kk * 128];
#pragma unroll
                for (int oo = 0; oo < 16; ++oo) {
                    const int tp = kk - oo;
                    if (tp >= 0 && tp < DP_K) acc[oo] = fmaf(wt[tp], uv, acc[oo]);
                }
                // (three waves per SIMD: keep the scheduler from hoisting all 46 window reads above the FMAs - 46 live
                // values on top of 31 taps and 16 accumulators do not fit 168 registers)
                if (!DS_WRES && (kk & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int oo = 0; oo < 16; oo += 2) {
                f16x2 hi, lo;
                split2(swishf(acc[oo]), swishf(acc[oo + 1]), hi, lo);
                vth[(sub * 16 + oo) * DP_VS + vcol] = hi[0];
                vth[(sub * 16 + oo + 1) * DP_VS + vcol] = hi[1];
                vtl[(sub * 16 + oo) * DP_VS + vcol] = lo[0];
                vtl[(sub * 16 + oo + 1) * DP_VS + vcol] = lo[1];
            }
        }
#if !DS_WRES
        f32x4 bias2[2];                                           // (after the depthwise phase: 16 VGPRs)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            xold[o] = ldg4(xr + 16 * (ob0 + o) + 4 * g);
            bias2[o] = ldg4(b2 + 16 * (ob0 + o) + 4 * g);
        }
#endif
        // the 30 rows the next window shares with this one (read before the barrier, written after it)
        f32x4 keep[4];
        if (has_next) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k;                      // 30 rows x 32 quads = 960
                const int ii = i < (DP_K - 1) * 32 ? i : (DP_K - 1) * 32 - 1;
                keep[k] = *reinterpret_cast<const f32x4*>(&utile[(DP_TL + (ii >> 5)) * 128 + (ii & 31) * 4]);
            }
        }
        __syncthreads();                                          // all u-window reads and v-tile writes are done

        // ---- pointwise 128 -> 64 on the matrix pipe + bias + residual ----
        f32x4 acc2[2] = {bias2[0], bias2[1]};
#if !DS_WRES
        // the 16 KB of pointwise operands are fetched here, per tile and per k-block (L1 / L2 hits, one k-block ahead),
        // instead of living in 64 VGPRs through the depthwise phase: the kernel then fits three waves per SIMD
        auto wfetch = [&](int mm, f16x8 (&fh)[2], f16x8 (&fl)[2]) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const _Float16* wp = w2i + ((ob0 + o) * 4 + mm) * 1024 + lane * 8;
                fh[o] = *reinterpret_cast<const f16x8*>(wp);
                fl[o] = *reinterpret_cast<const f16x8*>(wp + 512);
            }
        };
        f16x8 ch[2], cl[2];
        wfetch(0, ch, cl);
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            f16x8 nh[2] = {ch[0], ch[1]}, nl[2] = {cl[0], cl[1]};
            if (mm < 3) wfetch(mm + 1, nh, nl);
            const f16x8 bh = *reinterpret_cast<const f16x8*>(&vth[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(&vtl[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32h(ch[o], bh, acc2[o]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32l(ch[o], bl, acc2[o]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32l(cl[o], bh, acc2[o]);
#pragma unroll
            for (int o = 0; o < 2; ++o) { ch[o] = nh[o]; cl[o] = nl[o]; }
        }
#else
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(&vth[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(&vtl[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32h(ah[o][mm], bh, acc2[o]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32l(ah[o][mm], bl, acc2[o]);
#pragma unroll
            for (int o = 0; o < 2; ++o) acc2[o] = mfma32l(al[o][mm], bh, acc2[o]);
        }
#endif
        if (live) {
#pragma unroll
            for (int o = 0; o < 2; ++o) stg4(xr + 16 * (ob0 + o) + 4 * g, xold[o] + acc2[o]);
        }
        if (has_next) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = tid + 256 * k;
                if (i < (DP_K - 1) * 32) *reinterpret_cast<f32x4*>(&utile[(i >> 5) * 128 + (i & 31) * 4]) = keep[k];
                *reinterpret_cast<f32x4*>(&utile[(DP_K - 1 + (i >> 5)) * 128 + (i & 31) * 4]) = nxt[k];
            }
        }
        __syncthreads();                                          // window and v tiles are free for the next tile
    }
}
#endif  // !DS_TOEPLITZ

// ---------------------------------------------------------------------------------
// dwpw2t_x3_kernel: dwpw2s_x3_kernel with the 31-tap depthwise on the MATRIX pipe (the default).
//
// A depthwise convolution has no contraction over channels, so the GEMM-shaped MFMAs do not apply - but
// v_mfma_f32_4x4x4_16B_f16 is sixteen INDEPENDENT 4 x 4 x 4 products per wave, and a 1-D convolution is a banded
// Toeplitz product.  Block b of the instruction is one channel; with the window of that channel cut into chunks of
// four consecutive positions,
//     D_b[i][j] += sum_k A_q[i][k] * B_q[k][j],   A_q[i][k] = w_c[4q + k - i - 2],   B_q[k][j] = win_c[4 (j + q) + k]
// accumulates, over q = 0..8, out_c[4j + i] = sum_tau w_c[tau] win_c[4j + i + tau + 2]: 16 outputs x 16 channels per
// 9 instructions (27 with the three split products) instead of 124 v_fma_f32, and the instruction's lane layout
// (A: lane 4b + i holds k = 0..3; B: lane 4b + j; D: lane 4b + j holds i = 0..3; tools/probes/mfma4x4_probe.hip) makes
// every operand one aligned 8-byte unit:
//   * A_q is a constant of the weights: api.hip builds the image [channel group 8][q 9][lane 64][hi 4 | lo 4] once;
//     a wave keeps the 18 units of its two channel groups in registers;
//   * the window lives in LDS channel-major and already split: planes uh / ul of [128 channels][64 positions] halfs,
//     position p <-> sequence row l0 - 17 + p (the "- 2" above: it puts the 32 NEW rows of a tile at positions
//     32..63, chunk-aligned, and the chunks a lane reads at 4 (j + s), s = 0..12).  Row pitch 80 halfs: the
//     ds_read_b64 of 8 channels x 4 chunks of a lane group fall on 64 different banks;
//   * a thread stages 4 positions x 4 channels (four 16-byte fetches, a register transpose, split, eight
//     ds_write_b64); after a tile positions 32..63 move to 0..31 (one 64-byte row per thread);
//   * the second 16 outputs of a tile use chunks j + q + 4: one pass over s = j-relative chunk 0..12 feeds both
//     halves (26 chunk reads per channel group instead of 36).
// Everything after the depthwise (Swish, split, v tile, pointwise 128 -> 64, residual) is dwpw2s_x3_kernel's.
// ---------------------------------------------------------------------------------
// Cycle stamps of dwpw2t_x3_kernel (measurement builds only, -DDT_STAMP; tools/probes/dwpw2t_stamps.py): per-phase
// s_memtime deltas of every wave summed into per-wave-hashed slots (same-address atomics would distort the kernel).
#if defined(X3_SINGLE) && defined(DT_STAMP)
#undef DT_STAMP
#endif
#ifdef DT_STAMP
__device__ unsigned long long g_dt_stamp[2048][16];
struct DtStamp { unsigned long long t, acc[12]; };
#define DT_MARK(ph)                                             \
    do {                                                        \
        __builtin_amdgcn_sched_barrier(0);                      \
        const unsigned long long _t = __builtin_readcyclecounter(); \
        sp.acc[ph] += _t - sp.t;                                \
        sp.t = _t;                                              \
        __builtin_amdgcn_sched_barrier(0);                      \
    } while (0)
#define DT_STAMP_ARG , DtStamp& sp
#define DT_STAMP_PASS , sp
#else
#define DT_MARK(ph)
#define DT_STAMP_ARG
#define DT_STAMP_PASS
#endif
#define DT_PITCH 80
#ifndef DT_MAPB
#define DT_MAPB 1                // staging lane map: 1 = 16 position pairs x 2 channel quads per 32-lane store group
#endif
#ifndef DT_NB
#define DT_NB 3                  // chunk buffers of the depthwise loop: reads run DT_NB - 1 steps ahead of their MFMAs
#endif
__device__ __forceinline__ f32x4 mfma4h(f16x4 a, f16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4l(f16x4 a, f16x4 b, f32x4 c) {       // a term with a lo operand
    return X3_TERMS == 3 ? __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0) : c;
}
// The depthwise of one tile for this wave (one channel group): chunk step s feeds q = s of the first 16 outputs (s <= 8)
// and q = s - 4 of the second 16 (s >= 4).  Both halves always: a second half beyond the sequence end - one tile in 11 / in
// 4 - costs 27 MFMAs (its v rows feed only outputs which are never stored), but a second code path costs EVERY tile a
// vmcnt(0) at its first MFMA: the structurised CFG has an edge from the end of one path, where the residual-row fetch is
// pending, to the top of the other.  Chunk reads run DT_NB - 1 steps ahead of their MFMAs.  (Dependent 4x4x4 MFMAs issue
// every 13 cycles, independent ones every 8.5 - tools/probes/mfma4x4_probe.hip; the two halves alternate in the middle
// steps; a second accumulator per half would cost 8 of the 128 registers four waves per SIMD leave.)
__device__ __forceinline__ void dt_taps(const _Float16* bhp, const _Float16* blp, const f16x4 (&wah)[9],
                                        const f16x4 (&wal)[9], float dbias, int vcol, int dj, _Float16* vth,
                                        _Float16* vtl, __amdgpu_buffer_rsrc_t xrs, unsigned xo, f32x4& xold DT_STAMP_ARG) {
    constexpr int NS = 13, NH = 2;
    f32x4 d[2];                                                 // [half]
    d[0] = d[1] = splat4(dbias);
    f16x4 bh[DT_NB], bl[DT_NB];
    auto rd = [&](int s) {
        bh[s % DT_NB] = *reinterpret_cast<const f16x4*>(bhp + 4 * s);
        bl[s % DT_NB] = *reinterpret_cast<const f16x4*>(blp + 4 * s);
    };
#pragma unroll
    for (int s = 0; s < DT_NB - 1; ++s) rd(s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + DT_NB - 1 < NS) rd(s + DT_NB - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) {
                const int q = s - 4 * hh;
                if (q < 0 || q > 8) continue;
                if (term == 0) d[hh] = mfma4h(wah[q], bh[s % DT_NB], d[hh]);
                if (term == 1) d[hh] = mfma4l(wah[q], bl[s % DT_NB], d[hh]);
                if (term == 2) d[hh] = mfma4l(wal[q], bh[s % DT_NB], d[hh]);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // the residual row of this lane's pointwise output: requested here - the chunk registers are free, and Swish, the
    // v-tile stores and a barrier lie between the request and its use
    xold = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, 0, 0));
    DT_MARK(2);                                                 // chunk reads + 4x4x4 MFMAs
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            f16x2 hi, lo;
            split2(swishf(d[hh][i]), swishf(d[hh][i + 1]), hi, lo);
            const int row = 16 * hh + 4 * dj + i;
            vth[row * DP_VS + vcol] = hi[0];
            vth[(row + 1) * DP_VS + vcol] = hi[1];
            vtl[row * DP_VS + vcol] = lo[0];
            vtl[(row + 1) * DP_VS + vcol] = lo[1];
        }
    }
    DT_MARK(3);                                                 // Swish, split, v-tile stores
}

// 512 threads: wave w = channel group w of the depthwise (16 channels, 36 operand registers) and one (token block, output
// block) pair of the pointwise product (16 operand registers + the lo halves in LDS): everything a wave re-uses stays in
// registers at four waves per SIMD.
// The blocks are PERSISTENT: block b owns a contiguous range of the (sequence, segment) items (XCD-contiguous, so the 30
// halo rows between two segments of a sequence are an L2 hit) and walks their tiles as one stream.  The operands are
// fetched once per block, and the first window of the next item is fetched under the last tile of the current one (into
// the registers the kept-half copy does not need there) - measured with cycle stamps, the per-item prologue (19 fetches,
// their latency, one barrier) was 28 - 31 % of the one-block-per-item kernel.
struct DtItem {
    __amdgpu_buffer_rsrc_t ur, xr;      // rows of the item's sequence in u / in x (descriptor + 32-bit lane offset: a
                                        // uniform pointer + lane offset is hoisted into 64-bit per-lane pointers, 8 VGPRs)
    int l_begin, ntiles;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dt_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__global__ __launch_bounds__(512, 4) void dwpw2t_x3_kernel(float* __restrict__ x, const float* __restrict__ u,
                                                           const _Float16* __restrict__ dwi,
                                                           const float* __restrict__ dw_b,
                                                           const _Float16* __restrict__ w2i,
                                                           const float* __restrict__ b2, TokMap m, int nseq, int nsegs) {
    __shared__ __attribute__((aligned(16))) _Float16 uh[128 * DT_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 ul[128 * DT_PITCH];
    __shared__ __attribute__((aligned(16))) _Float16 vth[DP_TL * DP_VS];
    __shared__ __attribute__((aligned(16))) _Float16 vtl[DP_TL * DP_VS];
    __shared__ __attribute__((aligned(16))) _Float16 w2l[16 * 64 * 8];      // lo halves of the pointwise operand image
    __shared__ __attribute__((aligned(16))) float b2l[64];                  // pointwise bias
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DT_STAMP
    DtStamp sp;
    sp.t = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 12; ++i) sp.acc[i] = 0;
    int stamp_tiles = 0;
#endif
    const long lblk = XCD_ORDER ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long nitems = (long)nseq * nsegs;
    const int it0 = (int)(lblk * nitems / gridDim.x), it1 = (int)((lblk + 1) * nitems / gridDim.x);
    if (it0 >= it1) return;                                      // more blocks than items (block-uniform)

    const int tb = wv >> 2, ob = wv & 3;                         // pointwise work of this wave: token block, output block
    const int db = lane >> 2, dj = lane & 3;                     // depthwise: MFMA block (channel of group wv), column
    const int chn = 16 * wv + db;
    const unsigned rowb = (unsigned)m.lstride * 512u, xrowb = (unsigned)m.lstride * 256u;    // bytes per row of u / of x
    auto item_of = [&](int it) -> DtItem {
        const int n = it / nsegs, seg = it - n * nsegs;
        const int nq = n / m.inner;
        const long nbase = (long)nq * m.outer + (long)(n - nq * m.inner) * m.istride;
        DtItem d;
        d.ur = dt_rsrc(u + nbase * 128, (unsigned)(m.L - 1) * rowb + 512u);
        d.xr = dt_rsrc(x + nbase * 64, (unsigned)(m.L - 1) * xrowb + 256u);
        d.l_begin = seg * DS_SEG * DP_TL;
        const int l_end = d.l_begin + DS_SEG * DP_TL < m.L ? d.l_begin + DS_SEG * DP_TL : m.L;
        d.ntiles = (l_end - d.l_begin + DP_TL - 1) / DP_TL;
        return d;
    };

    // staging item of this thread: position PAIR s_pp (of the 16 of a half window) x channel quad s_cq.  The channel quad
    // does not move the LDS bank at this pitch (160 cq = 0 mod 32), so a 32-lane ds_write_b32 group should span as many
    // pairs as possible: 16 pairs x 2 quads is 2-way (free); 8 pairs x 4 quads, whose wave fetch is 8 rows x 128 B instead
    // of 16 rows x 64 B, is 4-way (16 cycles per store instead of 4) and measured 3 - 4 % slower (2.21 vs 2.13 ms)
#if DT_MAPB
    const int s_pp = lane & 15, s_cq = (lane >> 4) + 4 * wv;
#else
    const int s_pp = (lane & 7) + 8 * (wv & 1), s_cq = (lane >> 3) + 8 * (wv >> 1);
#endif
    // Rows 2 pp, 2 pp + 1 of a half of the window (position p <-> row l0 - 17 + p), addressed as uniform base + 32-bit
    // lane offset (a sequence spans < 2^32 bytes of u: checked by the launcher).  The fetch is unconditional (clamped
    // row); rows outside [0, L) - the convolution's zero padding - are zeroed when the values are USED (a select on the
    // fetched register here would make every boundary tile wait for its own prefetch)
    const unsigned cqb = (unsigned)s_cq * 16u;
    auto load_pair = [&](__amdgpu_buffer_rsrc_t ur, int lfirst, f32x4 (&r)[2], unsigned& okm) {   // lfirst = row of position pair 0
        okm = 0;
        int pp2 = 2 * s_pp;
        asm volatile("" : "+v"(pp2));                            // (opaque: the next item's offsets are otherwise computed at the
                                                                 // item's START and live through all its tiles - registers)
#pragma unroll
        for (int rho = 0; rho < 2; ++rho) {
            const int l = lfirst + pp2 + rho;
            const int lc = min(max(l, 0), m.L - 1);             // (arithmetic clamp: a select between two ADDRESSES becomes a branch)
            r[rho] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, (unsigned)lc * rowb + cqb, 0, 0));
            okm |= (unsigned)(l == lc) << rho;
        }
    };
    auto store_pair = [&](int half, const f32x4 (&r)[2], unsigned okm) {
        const f32x4 r0 = (okm & 1u) ? r[0] : splat4(0.f), r1 = (okm & 2u) ? r[1] : splat4(0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f16x2 hi, lo;
            split2(r0[e], r1[e], hi, lo);                        // 2 consecutive positions of channel 4 s_cq + e
            *reinterpret_cast<f16x2*>(&uh[(4 * s_cq + e) * DT_PITCH + 32 * half + 2 * s_pp]) = hi;
            *reinterpret_cast<f16x2*>(&ul[(4 * s_cq + e) * DT_PITCH + 32 * half + 2 * s_pp]) = lo;
        }
    };
    DtItem cur = item_of(it0);
    // the first window's rows are requested FIRST: everything below is in flight behind them and is waited for at its use
    f32x4 low[2], nxt[2];                                        // (low doubles as the kept-half copy inside an item)
    unsigned low_ok, nxt_ok;
    load_pair(cur.ur, cur.l_begin - 17, low, low_ok);
    load_pair(cur.ur, cur.l_begin + 15, nxt, nxt_ok);

    // pointwise operands: the hi halves stay in registers; the lo halves (16 KB for the block) live in LDS - 128 registers
    // do not hold both next to the depthwise taps
    u32x4 w2t[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int un = tid + 512 * k;                            // 16-byte unit: (ob, mm) = un >> 6, lane un & 63
        w2t[k] = *reinterpret_cast<const u32x4*>(w2i + (un >> 6) * 1024 + 512 + (un & 63) * 8);
    }
    f16x8 ah[4];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) ah[mm] = *reinterpret_cast<const f16x8*>(w2i + (ob * 4 + mm) * 1024 + lane * 8);
    const _Float16* const alp = w2l + (ob * 4 * 64 + lane) * 8;
    store_pair(0, low, low_ok);                                   // (before the 36 registers of depthwise operands are requested)
    store_pair(1, nxt, nxt_ok);
#pragma unroll
    for (int k = 0; k < 2; ++k) *reinterpret_cast<u32x4*>(&w2l[(tid + 512 * k) * 8]) = w2t[k];
    if (tid < 64) b2l[tid] = b2[tid];
    // depthwise operands: channel group wv
    f16x8 wimg[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wimg[q] = *reinterpret_cast<const f16x8*>(dwi + ((wv * 9 + q) * 64 + lane) * 8);
    float dbias = dw_b[chn];
    const _Float16* const bhp = uh + chn * DT_PITCH + 4 * dj;    // this lane's chunk 0 (chunk s is 4 s halfs further)
    const _Float16* const blp = ul + chn * DT_PITCH + 4 * dj;
    // B-operand order inside a v row: channel 32m + 16h + 4gq + r  ->  32m + 8gq + 4h + r.  The 8-half units of row r are
    // stored XOR-swizzled by (r >> 2) & 3: a depthwise lane group writes rows 4 j + i of four j's at once, which would
    // meet in one bank (row pitch 72 dwords); the swizzle costs nothing on either side (per-lane constants) and keeps the
    // pointwise product's ds_read_b128 conflict-free (brute-forced over the lane-group map)
    const int vcol = ((chn & ~31) + ((chn >> 2) & 3) * 8 + ((chn >> 4) & 1) * 4 + (chn & 3)) ^ (8 * dj);
    const int gx = g ^ ((c >> 2) & 3);
    const _Float16* const vrh = vth + (16 * tb + c) * DP_VS + 8 * gx;     // this lane's B fragments of the pointwise product
    const _Float16* const vrl = vtl + (16 * tb + c) * DP_VS + 8 * gx;
    const unsigned xcolb = (unsigned)(16 * ob + 4 * g) * 4u;
    // the window half-row this thread moves after a tile: plane tid >> 8, channel (tid >> 1) & 127, 16 positions of 32..63
    _Float16* const krow = ((tid >> 8) ? ul : uh) + ((tid >> 1) & 127) * DT_PITCH + 16 * (tid & 1);
    // Nothing may be pending at the loop header: with the operand fetches above possibly in flight there, the compiler's
    // counted waits inside the depthwise loop become vmcnt(0) - i.e. a wait for the tile's own row prefetch.  (The empty
    // asm statements make the operands live HERE: without them the fetches are sunk below the barrier, next to their
    // first use inside the loop.)
    f16x4 wah[9], wal[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        asm volatile("" : "+v"(wimg[q]));
        wah[q] = __builtin_shufflevector(wimg[q], wimg[q], 0, 1, 2, 3);
        wal[q] = __builtin_shufflevector(wimg[q], wimg[q], 4, 5, 6, 7);
    }
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) asm volatile("" : "+v"(ah[mm]));
    asm volatile("" : "+v"(dbias));
    __syncthreads();
    DT_MARK(0);                                                   // prologue: operands, first window

#pragma unroll 1
    for (int it = it0; it < it1; ++it) {
        const bool more = it + 1 < it1;
        const DtItem nx = item_of(more ? it + 1 : it);
#pragma unroll 1
        for (int t = 0; t < cur.ntiles; ++t) {
            const int l0 = cur.l_begin + t * DP_TL;
            const bool has_next = t + 1 < cur.ntiles;             // another tile of this item: its 32 new rows
            const bool new_item = !has_next && more;              // else the whole first window of the next item
            // ---- prefetch ----
            if (has_next) load_pair(cur.ur, l0 + DP_TL + 15, nxt, nxt_ok);
            if (new_item) {
                load_pair(nx.ur, nx.l_begin - 17, low, low_ok);
                load_pair(nx.ur, nx.l_begin + 15, nxt, nxt_ok);
            }
            const int lrow = l0 + 16 * tb + c;
            const bool live = lrow < m.L;
            const unsigned xo = (unsigned)min(lrow, m.L - 1) * xrowb + xcolb;
            DT_MARK(1);                                           // prefetch issue

            // ---- depthwise: outputs 4 dj + i (+ 16 for the second half) of channel chn ----
            f32x4 xold;
            dt_taps(bhp, blp, wah, wal, dbias, vcol, dj, vth, vtl, cur.xr, xo, xold DT_STAMP_PASS);
            // the half of the window the next tile shares with this one (read before the barrier, written after it)
            if (has_next) {
#pragma unroll
                for (int k = 0; k < 2; ++k) low[k] = *reinterpret_cast<const f32x4*>(krow + 32 + 8 * k);
            }
            DT_MARK(4);                                           // kept-half reads
            __syncthreads();                                      // all window reads and v-tile writes are done
            DT_MARK(5);                                           // barrier A

            // ---- pointwise 128 -> 64 on the matrix pipe + bias + residual (one accumulator per split term:
            // consecutive MFMAs never chain) ----
            f32x4 acc2 = *reinterpret_cast<const f32x4*>(&b2l[16 * ob + 4 * g]), acc2b = splat4(0.f), acc2c = splat4(0.f);
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const f16x8 bh = *reinterpret_cast<const f16x8*>(vrh + 32 * mm);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(vrl + 32 * mm);
                const f16x8 aw = *reinterpret_cast<const f16x8*>(alp + mm * 512);
                acc2 = mfma32h(ah[mm], bh, acc2);
                acc2b = mfma32l(ah[mm], bl, acc2b);
                acc2c = mfma32l(aw, bh, acc2c);
            }
            acc2 = acc2 + (acc2b + acc2c);
            DT_MARK(6);                                           // pointwise product
            // (the sum is formed and pinned outside the branch: sunk into it, the wait for the residual row is repeated
            // after the branch as vmcnt(0), which by then includes the store.  An inline-asm store, as attn_sp_out_x3_kernel
            // uses, gains nothing here - and without the s_nop the compiler puts after a 16-byte store it knows about, the
            // next VALU write of the data registers corrupts the stored row: measured, 0.3 relative error)
            f32x4 xnew = xold + acc2;
            asm volatile("" : "+v"(xnew));
            if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xnew), cur.xr, xo, 0, 0);
            if (has_next) {
#pragma unroll
                for (int k = 0; k < 2; ++k) *reinterpret_cast<f32x4*>(krow + 8 * k) = low[k];
                store_pair(1, nxt, nxt_ok);
            }
            if (new_item) {
                store_pair(0, low, low_ok);
                store_pair(1, nxt, nxt_ok);
            }
            DT_MARK(7);                                           // store, kept half, split + store of the new rows
            __syncthreads();                                      // window and v tiles are free for the next tile
            DT_MARK(8);                                           // barrier B
#ifdef DT_STAMP
            ++stamp_tiles;
#endif
        }
        cur = nx;
    }
#ifdef DT_STAMP
    if (lane == 0) {
        unsigned long long* slot = g_dt_stamp[(blockIdx.x * 8 + wv) & 2047];
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(&slot[i], sp.acc[i]);
        atomicAdd(&slot[14], 1ull);
        atomicAdd(&slot[15], (unsigned long long)stamp_tiles);
    }
#endif
}
#ifdef DT_STAMP
}  // namespace X3_NS
extern "C" int cmgan_dbg_dt_stamps(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    static unsigned long long host[2048][16];
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(X3_NS::g_dt_stamp), sizeof host);
    for (int i = 0; i < 16; ++i) {
        out[i] = 0;
        for (int sl = 0; sl < 2048; ++sl) out[i] += host[sl][i];
    }
    if (e == hipSuccess && reset) {
        for (auto& row : host) for (auto& v : row) v = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(X3_NS::g_dt_stamp), host, sizeof host);
    }
    return e == hipSuccess ? 0 : -1;
}
namespace X3_NS {
#endif

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
}  // namespace X3_NS
using namespace X3_NS;

static int persistent_grid(int ntiles, int blocks_per_cu) {
    const int want = (ntiles + XWAVES - 1) / XWAVES;
    const int cap = 256 * blocks_per_cu;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

static int ffn_grid(int ntiles, int waves = FFN_WAVES) {  // one persistent FeedForward block per CU
    const int want = (ntiles + waves - 1) / waves;
    return want < 256 ? (want > 0 ? want : 1) : 256;
}

bool conformer_x3_addressable(const TokMap& seq) {
    const long N = seq.nblocks / seq.Lb;
    const long span = (long)(seq.L - 1) * seq.lstride + 1;       // rows between a sequence's first and last token
    return span * 512 < (1l << 32) && N * ((seq.L + 31) / 32) < (1l << 31);
}

// ---- the stages of a ConformerBlock in THIS build (split-f16, or its single-product twin): ConfStageTbl, kernels.h ----
static void stage_ffn(LaunchCtx ctx, int ff, bool plain, const float* xin, float* xout, const float* x0, const ConfWeights& w,
                      const ConfWeightsX3& w16, long M) {
    // FeedForward on 32x32x16 MFMAs (ffn32_x3.hip) unless CMGAN_FFN32=0 (the 16x16x32 kernel above: same-session A/B)
    static const bool k_ffn32 = env_knob("CMGAN_FFN32", FFN32_DEFAULT, 0, 1) != 0;
    const int flat_tiles = (int)(((M + 15) / 16 + XNTB - 1) / XNTB);
    hipStream_t s = ctx.stream;
    const bool final_ = ff == 2 && !plain;
    const _Float16 *w1 = ff == 1 ? w16.ff1_w1 : w16.ff2_w1, *w2 = ff == 1 ? w16.ff1_w2 : w16.ff2_w2;
    const _Float16 *w1_32 = ff == 1 ? w16.ff1_w1_32 : w16.ff2_w1_32, *w2_32 = ff == 1 ? w16.ff1_w2_32 : w16.ff2_w2_32;
    const float *b1 = ff == 1 ? w.ff1_b1 : w.ff2_b1, *b2 = ff == 1 ? w.ff1_b2 : w.ff2_b2;
    if (k_ffn32) { launch_ffn32_x3(ctx, final_, xin, xout, final_ ? x0 : nullptr, final_ ? w.post_gb : nullptr, w1_32, b1, w2_32, b2, M); return; }
    if (final_)
        LAUNCH(ctx, "ffn_post", (ffn_x3_kernel<true, 2, FFN_POST_WAVES><<<ffn_grid(flat_tiles, FFN_POST_WAVES), 64 * FFN_POST_WAVES, 0, s>>>(
                                    xin, xout, x0, w.post_gb, w1, b1, w2, b2, M, flat_tiles)));
    else
        LAUNCH(ctx, "ffn", (ffn_x3_kernel<false, 2, FFN_WAVES><<<ffn_grid(flat_tiles), 64 * FFN_WAVES, 0, s>>>(
                               xin, xout, nullptr, nullptr, w1, b1, w2, b2, M, flat_tiles)));
}

// attn32_x3.hip: LN -> q, k, v tile images, then attention + to_out + residual (masked calls take the un-pipelined
// kernel, which carries the mask logic)
static void stage_qkv(LaunchCtx ctx, const float* x, const TokMap& seq, const ConfWeights& w, const ConfWeightsX3& w16,
                      const ConfBuffers& b) {
    launch_qkv32_x3(ctx, x, seq, w16.qkv_w, w.qkv_b, reinterpret_cast<_Float16*>(b.q), reinterpret_cast<_Float16*>(b.k),
                    reinterpret_cast<_Float16*>(b.v));
}

static void stage_attn(LaunchCtx ctx, float* x, const TokMap& seq, const ConfWeights& w, const ConfWeightsX3& w16,
                       const ConfBuffers& b, const unsigned char* mask) {
    const _Float16 *qimg = reinterpret_cast<const _Float16*>(b.q), *kimg = reinterpret_cast<const _Float16*>(b.k),
                   *vimg = reinterpret_cast<const _Float16*>(b.v);
    if (!mask) launch_attn_sp_out_x3(ctx, qimg, kimg, vimg, w16.rel_planes, w.max_pos, x, seq, w16.wo, w.bo);
    else launch_attn32_out_x3(ctx, qimg, kimg, vimg, w16.rel_planes, w.max_pos, x, seq, w16.wo, w.bo, mask);
}

static void stage_pw1glu(LaunchCtx ctx, const float* x, const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b,
                         long M) {
    const int flat_tiles = (int)(((M + 15) / 16 + XNTB - 1) / XNTB);
    LAUNCH(ctx, "pw1glu", (pw1glu_x3_kernel<<<persistent_grid(flat_tiles, 2), 512, 0, ctx.stream>>>(
                              x, b.u, w16.pw1_w, w.pw1_b, M, flat_tiles)));
}

static void stage_dwpw2(LaunchCtx ctx, float* x, const TokMap& seq, const ConfWeights& w, const ConfWeightsX3& w16,
                        const ConfBuffers& b) {
    const int N = seq.nblocks / seq.Lb;
    const int ntl = (seq.L + DP_TL - 1) / DP_TL;
    const int nsegs = (ntl + DS_SEG - 1) / DS_SEG;
    const long items = (long)N * nsegs;
#if DS_TOEPLITZ
    // persistent blocks (two per CU) over contiguous ranges of the items.  The kernel addresses the rows of a sequence
    // with 32-bit byte offsets: conformer_x3_addressable rejects shapes whose sequences span 4 GB of u
    const unsigned pgrid = items < 512 ? (unsigned)(((items + 7) / 8) * 8) : 512u;
    LAUNCH(ctx, "dwpw2", (dwpw2t_x3_kernel<<<pgrid, 512, 0, ctx.stream>>>(x, b.u, w16.dw_img, w.dw_b, w16.pw2_w, w.pw2_b, seq, N,
                                                                         nsegs)));
#else
    const unsigned grid = XCD_ORDER ? (unsigned)(((items + 7) / 8) * 8) : (unsigned)items;
    LAUNCH(ctx, "dwpw2", (dwpw2s_x3_kernel<<<grid, 256, 0, ctx.stream>>>(x, b.u, w.dw_w, w.dw_b, w16.pw2_w, w.pw2_b, seq, N,
                                                                        nsegs)));
#endif
}

const ConfStageTbl& conf_stages_x3() {
    static const ConfStageTbl t{stage_ffn, stage_qkv, stage_attn, stage_pw1glu, stage_dwpw2};
    return t;
}

bool conformer_forward_x3(LaunchCtx ctx, const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b,
                          const TokMap& seq, long M, float* taps, bool outer_residual, const unsigned char* mask) {
    if (!conformer_x3_addressable(seq)) return false;
    const ConfStageTbl& t = conf_stages_x3();
    return conformer_forward_tbl(ctx, t, t, t, t, t, t, w, w16, b, seq, M, taps, outer_residual, mask);
}

#ifndef X3_SINGLE
// f16 MFMA convention self-test: D = A(16 x 32KB2) * B with x3 images built like the loader does.
__global__ void selftest_x3_kernel(const _Float16* __restrict__ a_img, const float* __restrict__ b_fm,
                                   float* __restrict__ d, int M32) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    f32x4 acc = splat4(0.f);
    for (int m = 0; m < M32; ++m) {
        f16x8 bh, bl;
        split8(ldg4(b_fm + (2 * m) * 256 + lane * 4), ldg4(b_fm + (2 * m + 1) * 256 + lane * 4), bh, bl);
        const f16x8 ah = *reinterpret_cast<const f16x8*>(a_img + m * 1024 + lane * 8);
        const f16x8 al = *reinterpret_cast<const f16x8*>(a_img + m * 1024 + 512 + lane * 8);
        acc = mfma32h(ah, bh, acc);
        acc = mfma32h(ah, bl, acc);
        acc = mfma32h(al, bh, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + c] = acc[r];
}

void launch_selftest_x3(hipStream_t s, const void* a_img, const float* b_fm, float* d, int M32) {
    selftest_x3_kernel<<<1, 64, 0, s>>>(reinterpret_cast<const _Float16*>(a_img), b_fm, d, M32);
}

#endif
