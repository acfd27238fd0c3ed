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

#define XNTB 2
#define XWAVES 8

__device__ __forceinline__ float swish_x(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

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
template <bool FINAL>
__global__ __launch_bounds__(512) void ffn_x3_kernel(const float* xin, float* xout, const float* x0,
                                                     const float* __restrict__ post_gb,
                                                     const _Float16* __restrict__ w1i, const float* __restrict__ b1,
                                                     const _Float16* __restrict__ w2i, const float* __restrict__ b2,
                                                     long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];          // 128 KB
    _Float16* w1 = wlds;                 // 16*2*1024 halfs
    _Float16* w2 = wlds + 32768;         // 4*8*1024 halfs
    stage_lds16(w1i, w1, 4096);
    stage_lds16(w2i, w2, 4096);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;

#pragma unroll 1
    for (int tile = blockIdx.x * XWAVES + wv; tile < ntiles; tile += gridDim.x * XWAVES) {
        long row[XNTB];
        bool ok[XNTB];
        f32x4 x[XNTB][4];
        f16x8 xbh[XNTB][2], xbl[XNTB][2];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
            const long t = ((long)tile * XNTB + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) x[tb][kb] = ldg4(xin + row[tb] * 64 + 16 * kb + 4 * g);
            ln_split(x[tb], xbh[tb], xbl[tb]);
        }
        f32x4 y[XNTB][4];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) y[tb][ob] = splat4(0.f);

#pragma unroll 1
        for (int m2 = 0; m2 < 8; ++m2) {              // hidden k32 blocks = pairs of 16-unit blocks
            f32x4 hacc[2][XNTB];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hb = 2 * m2 + j;
                const f32x4 bias = ldg4(b1 + 16 * hb + 4 * g);
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) hacc[j][tb] = bias;
                lin_acc_x3<2, XNTB>(w1 + hb * 2048 + lane * 8, xbh, xbl, hacc[j]);
            }
            f16x8 hh[XNTB], hl[XNTB];
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hacc[0][tb][r] = swish_x(hacc[0][tb][r]);
                    hacc[1][tb][r] = swish_x(hacc[1][tb][r]);
                }
                split8(hacc[0][tb], hacc[1][tb], hh[tb], hl[tb]);
            }
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const _Float16* wp = w2 + (ob * 8 + m2) * 1024 + lane * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(wp);
                const f16x8 al = *reinterpret_cast<const f16x8*>(wp + 512);
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) y[tb][ob] = mfma32h(ah, hh[tb], y[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) y[tb][ob] = mfma32h(ah, hl[tb], y[tb][ob]);
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) y[tb][ob] = mfma32h(al, hh[tb], y[tb][ob]);
            }
        }
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) y[tb][ob] = y[tb][ob] + ldg4(b2 + 16 * ob + 4 * g) + x[tb][ob];
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

// ---------------------------------------------------------------------------------
// LN -> q (x0.25 folded), k, v.  A wave owns a PAIR of 16-token blocks (32 consecutive
// positions of one sequence).  Outputs per (sequence n, head h):
//   qh/ql, kh/kl : fp16 rows [Lp = 32*Lb2][16]      (hi and lo planes)
//   vimg         : [Lb2][hi|lo][64 lanes][8 halfs]   A operand of O^T = V^T P^T:
//                  lane (d, g) slot e <-> key 32*ip + 16*(e>>2) + 4*g + (e&3)
// LDS: weight image [12][2] = 48 KB + 1 KB transposition scratch per wave.
// ---------------------------------------------------------------------------------
struct QkvOut {
    _Float16 *qh, *ql, *kh, *kl, *vimg;
};

__global__ __launch_bounds__(512) void qkv_x3_kernel(const float* __restrict__ x, TokMap m, int Lb2,
                                                     const _Float16* __restrict__ wi, const float* __restrict__ b,
                                                     QkvOut o, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[24576 + 4352];   // 48 KB image + 8.5 KB scratch
    _Float16* w = wlds;                                        // 12*2*1024 halfs = 48 KB
    float* scratch = reinterpret_cast<float*>(wlds + 24576);   // 8 waves x 16 x 17 floats
    stage_lds16(wi, w, 3072);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    float* T = scratch + wv * 272;
    const int Lp = Lb2 * 32;

#pragma unroll 1
    for (int tile = blockIdx.x * XWAVES + wv; tile < ntiles; tile += gridDim.x * XWAVES) {
        const int n = tile / Lb2, ip = tile - n * Lb2;
        f16x8 xbh[XNTB][2], xbl[XNTB][2];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
            int l = ip * 32 + tb * 16 + c;
            if (l >= m.L) l = m.L - 1;
            const long row = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride + (long)l * m.lstride;
            f32x4 xr[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) xr[kb] = ldg4(x + row * 64 + 16 * kb + 4 * g);
            ln_split(xr, xbh[tb], xbl[tb]);
        }
#pragma unroll 1
        for (int ob = 0; ob < 12; ++ob) {
            const f32x4 bias = ldg4(b + 16 * ob + 4 * g);
            f32x4 acc[XNTB];
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) acc[tb] = bias;
            lin_acc_x3<2, XNTB>(w + ob * 2048 + lane * 8, xbh, xbl, acc);
            const int which = ob >> 2, h = ob & 3;
            const long nh = (long)n * 4 + h;
            if (which < 2) {
                _Float16* ph = which == 0 ? o.qh : o.kh;
                _Float16* pl = which == 0 ? o.ql : o.kl;
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) {
                    f16x4 hi, lo;
                    split4(acc[tb], hi, lo);
                    const long off = (nh * Lp + ip * 32 + tb * 16 + c) * 16 + 4 * g;
                    *reinterpret_cast<f16x4*>(ph + off) = hi;
                    *reinterpret_cast<f16x4*>(pl + off) = lo;
                }
            } else {
                f32x4 vt[XNTB];
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) {
                    wave_lds_fence();
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[c * 17 + 4 * g + r] = acc[tb][r];      // T[token][d]
                    wave_lds_fence();
#pragma unroll
                    for (int r = 0; r < 4; ++r) vt[tb][r] = T[(4 * g + r) * 17 + c];     // V[token 4g+r][d = c]
                }
                f16x8 vh, vl;
                split8(vt[0], vt[1], vh, vl);
                _Float16* base = o.vimg + ((nh * Lb2 + ip) * 2) * 512 + lane * 8;
                *reinterpret_cast<f16x8*>(base) = vh;
                *reinterpret_cast<f16x8*>(base + 512) = vl;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Attention core, one (sequence, head, 16-query block) per wave - see attn_kernel for the
// algorithm.  Contraction slots of the 32-wide MFMA: lane groups g = 0,1 carry the hi half
// of K (resp. E), g = 2,3 the lo half, both over d = 8*(g&1) + e; B carries Q_hi in both
// halves (MFMA 1) then Q_lo (MFMA 2): two MFMAs give (K_hi + K_lo) . (Q_hi + Q_lo).
// ---------------------------------------------------------------------------------
#define RSTRIDE_X 20
__global__ __launch_bounds__(256) void attn_x3_kernel(QkvOut io, const _Float16* __restrict__ eh,
                                                      const _Float16* __restrict__ el, int max_pos,
                                                      float* __restrict__ o, int L, int Lb, int Lb2, long total) {
    __shared__ float rbuf[4][80 * RSTRIDE_X];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wv;          // ((n*4 + h) * Lb + ib)
    if (item >= total) return;
    const int ib = (int)(item % Lb);
    const long nh = item / Lb;
    float* R = rbuf[wv];
    const int Lp = Lb2 * 32;
    const int dsel = 8 * (g & 1);
    const bool lo_half = g >= 2;

    const long qoff = (nh * Lp + ib * 16 + c) * 16 + dsel;
    const f16x8 q1 = *reinterpret_cast<const f16x8*>(io.qh + qoff);
    const f16x8 q2 = *reinterpret_cast<const f16x8*>(io.ql + qoff);
    const _Float16* kp = (lo_half ? io.kl : io.kh) + (nh * Lp + c) * 16 + dsel;
    const _Float16* ep = (lo_half ? el : eh) + dsel;
    const _Float16* vp = io.vimg + nh * Lb2 * 1024 + lane * 8;
    const int i0 = ib * 16;

    float mrun = -INFINITY, lrun = 0.f;
    f32x4 oacc = splat4(0.f);

#pragma unroll 1
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int rem = (L - j0 + 15) >> 4;
        const int nb = rem < 4 ? rem : 4;
        f32x4 s[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            s[jb] = splat4(0.f);
            if (jb < nb) {
                const f16x8 kf = *reinterpret_cast<const f16x8*>(kp + (long)(j0 + 16 * jb) * 16);
                s[jb] = mfma32h(kf, q1, s[jb]);
                s[jb] = mfma32h(kf, q2, s[jb]);
            }
        }
        const int rmin = i0 - j0 - 63;
#pragma unroll 1
        for (int cb = 4 - nb; cb < 5; ++cb) {
            int rl = rmin + 16 * cb + c;
            rl = rl < -max_pos ? -max_pos : (rl > max_pos ? max_pos : rl);
            const f16x8 ef = *reinterpret_cast<const f16x8*>(ep + (long)(rl + max_pos) * 16);
            f32x4 rt = splat4(0.f);
            rt = mfma32h(ef, q1, rt);
            rt = mfma32h(ef, q2, rt);
#pragma unroll
            for (int r = 0; r < 4; ++r) R[(16 * cb + 4 * g + r) * RSTRIDE_X + c] = rt[r];
        }
        wave_lds_fence();
        float mx = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + 16 * jb + 4 * g + r;
                float sv = -INFINITY;
                if (jb < nb && key < L) sv = s[jb][r] + R[(c - 16 * jb - 4 * g - r + 63) * RSTRIDE_X + c];
                s[jb][r] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        wave_lds_fence();
        mx = red_g_max(mx);
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[jb][r] - mnew);
                s[jb][r] = p;
                psum += p;
            }
        psum = red_g_sum(psum);
        lrun = lrun * alpha + psum;
        oacc = oacc * splat4(alpha);
        mrun = mnew;
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            if (2 * mp < nb) {
                f16x8 ph, pl;
                split8(s[2 * mp], s[2 * mp + 1], ph, pl);
                const _Float16* vb = vp + (long)((j0 >> 5) + mp) * 1024;
                const f16x8 vh = *reinterpret_cast<const f16x8*>(vb);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(vb + 512);
                oacc = mfma32h(vh, ph, oacc);
                oacc = mfma32h(vh, pl, oacc);
                oacc = mfma32h(vl, ph, oacc);
            }
        }
    }
    const float inv = 1.0f / lrun;
    stg4(o + item * 256 + lane * 4, oacc * splat4(inv));
}

// ---------------------------------------------------------------------------------
// to_out + bias + residual (in place); O arrives as fp32 C-fragments per head.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void outproj_x3_kernel(float* __restrict__ x, TokMap m,
                                                         const float* __restrict__ o,
                                                         const _Float16* __restrict__ wi,
                                                         const float* __restrict__ bo, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[8192];
    stage_lds16(wi, wlds, 1024);                             // [4][2] image = 16 KB
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const long hstride = (long)m.Lb * 256;
#pragma unroll 1
    for (int tile = blockIdx.x * XWAVES + wv; tile < ntiles; tile += gridDim.x * XWAVES) {
        bool ok[XNTB];
        long row[XNTB];
        f16x8 bh[XNTB][2], bl[XNTB][2];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
            int blk = tile * XNTB + tb;
            const bool live = blk < m.nblocks;
            if (!live) blk = m.nblocks - 1;
            ok[tb] = tok_row(m, blk, c, row[tb]) && live;
            const int n = blk / m.Lb, ib = blk - n * m.Lb;
            const long base = ((long)n * 4 * m.Lb + ib) * 256 + lane * 4;
            split8(ldg4(o + base), ldg4(o + base + hstride), bh[tb][0], bl[tb][0]);
            split8(ldg4(o + base + 2 * hstride), ldg4(o + base + 3 * hstride), bh[tb][1], bl[tb][1]);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 bias = ldg4(bo + 16 * ob + 4 * g);
            f32x4 acc[XNTB];
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) acc[tb] = bias;
            lin_acc_x3<2, XNTB>(wlds + ob * 2048 + lane * 8, bh, bl, acc);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) {
                if (ok[tb]) {
                    float* p = x + row[tb] * 64 + 16 * ob + 4 * g;
                    stg4(p, ldg4(p) + acc[tb]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// conv module part 1: LN -> pointwise 64->256 -> GLU.   LDS: [16][2] image = 64 KB.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pw1glu_x3_kernel(const float* __restrict__ x, float* __restrict__ u,
                                                        const _Float16* __restrict__ wi,
                                                        const float* __restrict__ b, long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[32768];          // 64 KB
    stage_lds16(wi, wlds, 4096);
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
#pragma unroll 1
        for (int ob = 0; ob < 8; ++ob) {
            f32x4 aa[XNTB], ag[XNTB];
            const f32x4 ba = ldg4(b + 16 * ob + 4 * g), bg = ldg4(b + 128 + 16 * ob + 4 * g);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) { aa[tb] = ba; ag[tb] = bg; }
            lin_acc_x3<2, XNTB>(wlds + ob * 2048 + lane * 8, bh, bl, aa);
            lin_acc_x3<2, XNTB>(wlds + (ob + 8) * 2048 + lane * 8, bh, bl, ag);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) {
                if (ok[tb]) {
                    f32x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = aa[tb][e] * __frcp_rn(1.0f + __expf(-ag[tb][e]));
                    stg4(u + row[tb] * 128 + 16 * ob + 4 * g, r);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// conv module part 3: pointwise 128->64 + bias + residual.   LDS: [4][4] image = 32 KB.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pw2_x3_kernel(float* __restrict__ x, const float* __restrict__ vin,
                                                     const _Float16* __restrict__ wi, const float* __restrict__ b,
                                                     long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[16384];          // 32 KB
    stage_lds16(wi, wlds, 2048);
    __syncthreads();
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
#pragma unroll 1
    for (int tile = blockIdx.x * XWAVES + wv; tile < ntiles; tile += gridDim.x * XWAVES) {
        long row[XNTB];
        bool ok[XNTB];
        f16x8 bh[XNTB][4], bl[XNTB][4];
#pragma unroll
        for (int tb = 0; tb < XNTB; ++tb) {
            const long t = ((long)tile * XNTB + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm)
                split8(ldg4(vin + row[tb] * 128 + 32 * mm + 4 * g), ldg4(vin + row[tb] * 128 + 32 * mm + 16 + 4 * g),
                       bh[tb][mm], bl[tb][mm]);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 bias = ldg4(b + 16 * ob + 4 * g);
            f32x4 acc[XNTB];
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) acc[tb] = bias;
            lin_acc_x3<4, XNTB>(wlds + ob * 4096 + lane * 8, bh, bl, acc);
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) {
                if (ok[tb]) {
                    float* p = x + row[tb] * 64 + 16 * ob + 4 * g;
                    stg4(p, ldg4(p) + acc[tb]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static int persistent_grid(int ntiles, int blocks_per_cu) {
    const int want = (ntiles + XWAVES - 1) / XWAVES;
    const int cap = 256 * blocks_per_cu;
    return want < cap ? (want > 0 ? want : 1) : cap;
}

void conformer_forward_x3(LaunchCtx ctx, const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b,
                          const TokMap& seq, long M, float* taps, bool outer_residual) {
    hipStream_t s = ctx.stream;
    const int N = seq.nblocks / seq.Lb;
    const int Lb2 = (seq.Lb + 1) / 2;
    const int flat_blocks = (int)((M + 15) / 16);
    const int flat_tiles = (flat_blocks + XNTB - 1) / XNTB;
    const size_t tap_bytes = (size_t)M * 64 * sizeof(float);
    const size_t plane = (size_t)N * 4 * Lb2 * 32 * 16;          // halfs in one hi (or lo) Q/K plane
    QkvOut io;
    io.qh = reinterpret_cast<_Float16*>(b.q); io.ql = io.qh + plane;
    io.kh = reinterpret_cast<_Float16*>(b.k); io.kl = io.kh + plane;
    io.vimg = reinterpret_cast<_Float16*>(b.v);

    LAUNCH(ctx, "ffn", (ffn_x3_kernel<false><<<persistent_grid(flat_tiles, 1), 512, 0, s>>>(
                           b.xa, b.xb, nullptr, nullptr, w16.ff1_w1, w.ff1_b1, w16.ff1_w2, w.ff1_b2, M, flat_tiles)));
    if (taps) hipMemcpyAsync(taps, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    const int qtiles = N * Lb2;
    LAUNCH(ctx, "qkv", (qkv_x3_kernel<<<persistent_grid(qtiles, 2), 512, 0, s>>>(
                           b.xb, seq, Lb2, w16.qkv_w, w.qkv_b, io, qtiles)));
    const long items = (long)N * 4 * seq.Lb;
    LAUNCH(ctx, "attn", (attn_x3_kernel<<<(unsigned)((items + 3) / 4), 256, 0, s>>>(
                            io, w16.rel_h, w16.rel_l, w.max_pos, b.o, seq.L, seq.Lb, Lb2, items)));
    const int otiles = (seq.nblocks + XNTB - 1) / XNTB;
    LAUNCH(ctx, "outproj", (outproj_x3_kernel<<<persistent_grid(otiles, 2), 512, 0, s>>>(b.xb, seq, b.o, w16.wo,
                                                                                           w.bo, otiles)));
    if (taps) hipMemcpyAsync(taps + (size_t)M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    LAUNCH(ctx, "pw1glu", (pw1glu_x3_kernel<<<persistent_grid(flat_tiles, 2), 512, 0, s>>>(
                              b.xb, b.u, w16.pw1_w, w.pw1_b, M, flat_tiles)));
    launch_dwconv(ctx, b.u, b.w, w.dw_w, w.dw_b, seq);
    LAUNCH(ctx, "pw2", (pw2_x3_kernel<<<persistent_grid(flat_tiles, 2), 512, 0, s>>>(b.xb, b.w, w16.pw2_w,
                                                                                       w.pw2_b, M, flat_tiles)));
    if (taps) {
        hipMemcpyAsync(taps + (size_t)2 * M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);
        LAUNCH(ctx, "ffn", (ffn_x3_kernel<false><<<persistent_grid(flat_tiles, 1), 512, 0, s>>>(
                               b.xb, taps + (size_t)3 * M * 64, nullptr, nullptr, w16.ff2_w1, w.ff2_b1, w16.ff2_w2,
                               w.ff2_b2, M, flat_tiles)));
    }
    LAUNCH(ctx, "ffn_post", (ffn_x3_kernel<true><<<persistent_grid(flat_tiles, 1), 512, 0, s>>>(
                                b.xb, b.xa, outer_residual ? b.xa : nullptr, w.post_gb, w16.ff2_w1, w.ff2_b1,
                                w16.ff2_w2, w.ff2_b2, M, flat_tiles)));
}

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
