// conformer.hip - the macaron ConformerBlock of CMGAN (reference:
// src/models/conformer.py:182-222, instantiated at src/models/generator.py:75-90 with
// dim 64, 4 heads x 16, ff_mult 4, conv expansion 2, kernel 31) as CDNA4 kernels.
//
// Residual stream: channels-last rows of 64 fp32 (256 B).  One kernel per sub-module
// boundary that needs data from OTHER tokens (attention, depthwise conv); everything
// that is per-token is fused and register-resident (see common.hip.h for the
// transposed-MFMA convention):
//   ffn_kernel      LN -> W1 -> Swish -> W2 -> +x            (hidden never leaves VGPRs)
//                   (+ post LayerNorm + TSCB outer residual when FINAL)
//   qkv_kernel      LN -> [Wq|Wkv]  -> fragment-major Q, K, V^T per (sequence, head)
//   attn_kernel     S^T = K Q^T + Shaw rel-pos (Toeplitz skew through wave-private LDS)
//                   -> online softmax -> O^T = V^T P^T       (scores never leave VGPRs)
//   outproj_kernel  Wo + bias + residual
//   pw1glu_kernel   LN -> Conv1d(64,256,1) -> GLU
//   dwconv_kernel   depthwise k=31 (+folded BatchNorm) -> Swish       (VALU, LDS tile)
//   pw2_kernel      Conv1d(128,64,1) + bias + residual
#include "kernels.h"

#define NTB 2   // token blocks (of 16) per wave in the per-token kernels

// ---------------------------------------------------------------------------------
// FeedForward: x + 0.5*FF(LN(x))            conformer.py:136-148, 211-212, 217, 220
// FINAL: additionally y = LN_post(.) + x0   conformer.py:221, generator.py:95,97
// LN affine is folded into W1/b1, the 0.5 into W2/b2 (packer.py).
// ---------------------------------------------------------------------------------
template <bool FINAL>
__global__ __launch_bounds__(256) void ffn_kernel(const float* xin, float* xout,   // xout may alias x0 (FINAL)
                                                  const float* x0,
                                                  const float* __restrict__ post_gb,
                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                  long M, int nblocks) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk0 = wave * NTB;
    if (blk0 >= nblocks) return;

    long row[NTB];
    bool ok[NTB];
    f32x4 x[NTB][4], xh[NTB][4];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const long t = (long)(blk0 + tb) * 16 + c;
        ok[tb] = t < M;
        row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) x[tb][kb] = ldg4(xin + row[tb] * 64 + 16 * kb + 4 * g);
    }
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        float mean, rstd;
        ln_stats(x[tb], mean, rstd);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) xh[tb][kb] = (x[tb][kb] - splat4(mean)) * splat4(rstd);
    }

    f32x4 y[NTB][4];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) y[tb][ob] = splat4(0.f);

#pragma unroll 1
    for (int hc = 0; hc < 4; ++hc) {          // 4 chunks of 64 hidden units
        f32x4 h[NTB][4];
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            const int ob = hc * 4 + hb;
            const f32x4 bias = ldg4(b1 + 16 * ob + 4 * g);
            f32x4 acc[NTB];
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) acc[tb] = bias;
            lin_acc<4, NTB>(w1 + (long)ob * 4 * 256 + lane * 4, xh, acc);
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[tb][hb][r] = swish_scaled(acc[tb][r]);
        }
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                const f32x4 a = ldg4(w2 + ((long)ob * 16 + hc * 4 + hb) * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int tb = 0; tb < NTB; ++tb) y[tb][ob] = mfma16(a[r], h[tb][hb][r], y[tb][ob]);
            }
        }
    }

#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
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
                if (x0) y[tb][ob] += ldg4(x0 + row[tb] * 64 + 16 * ob + 4 * g);   // TSCB outer residual
            }
        }
        if (ok[tb]) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) stg4(xout + row[tb] * 64 + 16 * ob + 4 * g, y[tb][ob]);
        }
    }
}

// ---------------------------------------------------------------------------------
// LN -> q = Wq x (x0.25 folded), k, v = Wkv x          conformer.py:100-101, 68
// outputs per (sequence n, head h, 16-token block ib) as 1 KiB register images:
//   Q, K : [lane][4] = row (16ib + c), dims 4g..4g+3      (B / A fragment of S^T = K Q^T)
//   V^T  : [(key>>2 & 3)*16 + d][key & 3]                  (A fragment of O^T = V^T P^T)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qkv_kernel(const float* __restrict__ x, TokMap m,
                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                  float* __restrict__ q, float* __restrict__ k,
                                                  float* __restrict__ v) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk0 = wave * NTB;
    if (blk0 >= m.nblocks) return;

    bool live[NTB];
    long obase[NTB];
    f32x4 xh[NTB][4];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        int blk = blk0 + tb;
        live[tb] = blk < m.nblocks;
        if (!live[tb]) blk = m.nblocks - 1;
        long row;
        tok_row(m, blk, c, row);
        f32x4 xr[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) xr[kb] = ldg4(x + row * 64 + 16 * kb + 4 * g);
        float mean, rstd;
        ln_stats(xr, mean, rstd);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) xh[tb][kb] = (xr[kb] - splat4(mean)) * splat4(rstd);
        const int n = blk / m.Lb, ib = blk - n * m.Lb;
        obase[tb] = ((long)n * 4 * m.Lb + ib) * 256;       // + h * Lb * 256
    }
    const long hstride = (long)m.Lb * 256;

#pragma unroll 1
    for (int ob = 0; ob < 12; ++ob) {
        const f32x4 bias = ldg4(b + 16 * ob + 4 * g);
        f32x4 acc[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[tb] = bias;
        lin_acc<4, NTB>(w + (long)ob * 4 * 256 + lane * 4, xh, acc);
        const int which = ob >> 2, h = ob & 3;
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (!live[tb]) continue;
            const long base = obase[tb] + h * hstride;
            if (which == 0) {
                stg4(q + base + lane * 4, acc[tb]);
            } else if (which == 1) {
                stg4(k + base + lane * 4, acc[tb]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[base + (((c >> 2) * 16) + 4 * g + r) * 4 + (c & 3)] = acc[tb][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Attention core (fp32 MFMA).  Every wave is independent (no block barriers) and owns a
// PAIR of consecutive 16-query blocks of one (sequence, head), so two independent
// MFMA -> LDS skew -> softmax -> MFMA chains are in flight and K / V / E operand fragments
// are shared by both.
//   dots = (q k^T + q E[clamp(i-j)]^T) * scale ; softmax_j ; out = attn v
//                                                          conformer.py:103-130
// Transposed: S^T[key][query] so a lane owns one query column -> softmax reductions
// are in-lane + two permlane swaps, and P^T is already the B fragment of O^T = V^T P^T.
// Relative positions: R^T[rel][query] = E_window q^T is a second MFMA product over the
// 79 relative offsets a (16 query x 64 key) tile can see; bias[key][query] =
// R^T[query - key - rmin][query] is a Toeplitz read-back through 6.4 KB of wave-private
// LDS per query block (row stride 20 floats: conflict-free for both the write and the read).
// Scores are in log2 units (log2(e) folded into the q projection): softmax is a bare v_exp_f32.
// ---------------------------------------------------------------------------------
#define RSTRIDE 20

struct AttnState {
    float m, run, l;      // reference level, running max relative to it, denominator (relative to m)
    f32x4 o;
};

// Online softmax with a stale reference kept in the band (-4, +12] around the running maximum; see
// the header of attn32_x3.hip for the derivation.  The rel-pos accumulator starts at -m_ref and
// the skewed tile is read straight into the score accumulators, so the common path has no per-score
// add or subtract: p = exp2(K q + E q - m_ref).
#define ATTN_HI 12.0f
#define ATTN_LO -4.0f
// MASK: the [b, n] mask of ConformerBlock.forward(x, mask) (conformer.py:113-126), see att_softmax in conformer_x3.hip.
template <bool FULL, bool MASK = false>
__device__ __forceinline__ void attn_softmax(f32x4 (&s)[4], int c, int g, int j0, int nb, int L, AttnState& st,
                                             const unsigned char* __restrict__ mk = nullptr, bool qvalid = true) {
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (FULL || jb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + 16 * jb + 4 * g + r;
                if (!FULL) s[jb][r] = key < L ? s[jb][r] : -INFINITY;
                if (MASK && (FULL || key < L)) s[jb][r] = qvalid ? (mk[key] ? s[jb][r] : -INFINITY) : 0.f;
                mx = fmaxf(mx, s[jb][r]);
            }
        }
    }
    const float run = fmaxf(st.run, red_g_max(mx));
    const bool dead = MASK && run == -INFINITY;           // every key so far was masked for this (unmasked) query
    const bool drift = !dead && (run > ATTN_HI || run < ATTN_LO);
    float psum = 0.f;
    if (__any(drift)) {
        const float alpha = st.l > 0.f ? __builtin_amdgcn_exp2f(-run) : 1.0f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            if (FULL || jb < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = dead ? 0.f : __builtin_amdgcn_exp2f(s[jb][r] - run);
                    s[jb][r] = p;
                    psum += p;
                }
            } else {
                s[jb] = splat4(0.f);
            }
        }
        st.l *= alpha;
        st.o = st.o * splat4(alpha);
        if (!dead) {
            st.m += run;
            st.run = 0.f;
        }
    } else {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            if (FULL || jb < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[jb][r]);
                    s[jb][r] = p;
                    psum += p;
                }
            } else {
                s[jb] = splat4(0.f);
            }
        }
        st.run = run;
    }
    st.l += red_g_sum(psum);
}

struct AttnCtx {
    const float *qp, *kp, *vp, *rel;
    float *RA, *RB;
    int Lb, L, max_pos, c, g, lane;
    const unsigned char* mk;          // this sequence's attention mask row (MASK variant only)
};

template <bool FULL, bool MASK = false>
__device__ __forceinline__ void attn_chunk(const AttnCtx& a, int ibA, int j0, const f32x4& qA, const f32x4& qB,
                                           AttnState& sa, AttnState& sb) {
    const int nb = FULL ? 4 : ((a.L - j0 + 15) >> 4);
    const int c = a.c, g = a.g;
    // window of the pair: 6 row blocks from rminA = 16 ibA - j0 - 63; A uses blocks 0..4, B blocks 1..5
    const int rminA = ibA * 16 - j0 - 63;
    wave_lds_fence();                                     // previous chunk's skew reads are done
#pragma unroll
    for (int we = 0; we < 6; ++we) {
        const bool useA = we < 5 && (FULL || we >= 4 - nb);
        const bool useB = we >= 1 && (FULL || we - 1 >= 4 - nb);
        if (useA || useB) {
            int rl = rminA + 16 * we + c;
            rl = rl < -a.max_pos ? -a.max_pos : (rl > a.max_pos ? a.max_pos : rl);
            const f32x4 ef = ldg4(a.rel + (long)(rl + a.max_pos) * 16 + 4 * g);
            if (useA) {
                f32x4 rt = splat4(-sa.m);
#pragma unroll
                for (int r = 0; r < 4; ++r) rt = mfma16(ef[r], qA[r], rt);
#pragma unroll
                for (int r = 0; r < 4; ++r) a.RA[(16 * we + 4 * g + r) * RSTRIDE + c] = rt[r];
            }
            if (useB) {
                f32x4 rt = splat4(-sb.m);
#pragma unroll
                for (int r = 0; r < 4; ++r) rt = mfma16(ef[r], qB[r], rt);
#pragma unroll
                for (int r = 0; r < 4; ++r) a.RB[(16 * (we - 1) + 4 * g + r) * RSTRIDE + c] = rt[r];
            }
        }
    }
    wave_lds_fence();
    // skewed (E q - m_ref) tile read straight into the score accumulators; K q accumulates on top
    f32x4 sA[4], sB[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (FULL || jb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sA[jb][r] = a.RA[(c - 16 * jb - 4 * g - r + 63) * RSTRIDE + c];
                sB[jb][r] = a.RB[(c - 16 * jb - 4 * g - r + 63) * RSTRIDE + c];
            }
            const f32x4 kf = ldg4(a.kp + (long)((j0 >> 4) + jb) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sA[jb] = mfma16(kf[r], qA[r], sA[jb]);
                sB[jb] = mfma16(kf[r], qB[r], sB[jb]);
            }
        } else {
            sA[jb] = splat4(0.f);
            sB[jb] = splat4(0.f);
        }
    }
    if (MASK) {
        const int ibB = ibA + 1 < a.Lb ? ibA + 1 : a.Lb - 1;
        const int la = ibA * 16 + c, lb = ibB * 16 + c;
        attn_softmax<FULL, true>(sA, c, g, j0, nb, a.L, sa, a.mk, a.mk[la < a.L ? la : a.L - 1] != 0);
        attn_softmax<FULL, true>(sB, c, g, j0, nb, a.L, sb, a.mk, a.mk[lb < a.L ? lb : a.L - 1] != 0);
    } else {
        attn_softmax<FULL>(sA, c, g, j0, nb, a.L, sa);
        attn_softmax<FULL>(sB, c, g, j0, nb, a.L, sb);
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (FULL || jb < nb) {
            const f32x4 vf = ldg4(a.vp + (long)((j0 >> 4) + jb) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sa.o = mfma16(vf[r], sA[jb][r], sa.o);
                sb.o = mfma16(vf[r], sB[jb][r], sb.o);
            }
        }
    }
}

template <bool MASK>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                   const float* __restrict__ v, const float* __restrict__ rel,
                                                   int max_pos, float* __restrict__ o, int L, int Lb, int npairs,
                                                   long total, const unsigned char* __restrict__ mask) {
    __shared__ float rbuf[4][2][80 * RSTRIDE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long item = (long)blockIdx.x * 4 + wv;          // (n*4 + h) * npairs + pair
    if (item >= total) return;                            // waves are independent: no barriers below
    const long nh = item / npairs;
    const int ibA = (int)(item % npairs) * 2;
    const int ibB = ibA + 1 < Lb ? ibA + 1 : Lb - 1;      // odd Lb: the last wave's B is a clamped duplicate
    AttnCtx a;
    a.lane = lane; a.c = lane & 15; a.g = lane >> 4;
    a.RA = rbuf[wv][0]; a.RB = rbuf[wv][1];
    a.Lb = Lb; a.L = L; a.max_pos = max_pos; a.rel = rel;
    a.mk = MASK ? mask + (nh >> 2) * L : nullptr;
    a.qp = q + nh * Lb * 256 + lane * 4;
    a.kp = k + nh * Lb * 256 + lane * 4;
    a.vp = v + nh * Lb * 256 + lane * 4;
    const f32x4 qA = ldg4(a.qp + (long)ibA * 256), qB = ldg4(a.qp + (long)ibB * 256);

    AttnState sa, sb;
    sa.m = sb.m = 0.f; sa.run = sb.run = -INFINITY; sa.l = sb.l = 0.f; sa.o = sb.o = splat4(0.f);
    const int nfull = L >> 6;
#pragma unroll 1
    for (int ch = 0; ch < nfull; ++ch) attn_chunk<true, MASK>(a, ibA, ch * 64, qA, qB, sa, sb);
    if (L & 63) attn_chunk<false, MASK>(a, ibA, nfull * 64, qA, qB, sa, sb);

    stg4(o + (nh * Lb + ibA) * 256 + lane * 4, sa.o * splat4(__builtin_amdgcn_rcpf(sa.l)));
    if (ibA + 1 < Lb) stg4(o + (nh * Lb + ibA + 1) * 256 + lane * 4, sb.o * splat4(__builtin_amdgcn_rcpf(sb.l)));
}

// ---------------------------------------------------------------------------------
// to_out + bias + residual (in place)                        conformer.py:131-132, 218
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void outproj_kernel(float* __restrict__ x, TokMap m,
                                                      const float* __restrict__ o,
                                                      const float* __restrict__ wo,
                                                      const float* __restrict__ bo) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk0 = wave * NTB;
    if (blk0 >= m.nblocks) return;
    bool ok[NTB];
    long row[NTB];
    f32x4 xf[NTB][4];
    const long hstride = (long)m.Lb * 256;
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        int blk = blk0 + tb;
        const bool live = blk < m.nblocks;
        if (!live) blk = m.nblocks - 1;
        ok[tb] = tok_row(m, blk, c, row[tb]) && live;
        const int n = blk / m.Lb, ib = blk - n * m.Lb;
        const long base = ((long)n * 4 * m.Lb + ib) * 256 + lane * 4;
#pragma unroll
        for (int h = 0; h < 4; ++h) xf[tb][h] = ldg4(o + base + h * hstride);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const f32x4 bias = ldg4(bo + 16 * ob + 4 * g);
        f32x4 acc[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[tb] = bias;
        lin_acc<4, NTB>(wo + (long)ob * 4 * 256 + lane * 4, xf, acc);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (ok[tb]) {
                float* p = x + row[tb] * 64 + 16 * ob + 4 * g;
                stg4(p, ldg4(p) + acc[tb]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// conv module, part 1: LN -> pointwise 64->256 -> GLU           conformer.py:161-164, 30-37
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pw1glu_kernel(const float* __restrict__ x, float* __restrict__ u,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ b, long M, int nblocks) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk0 = wave * NTB;
    if (blk0 >= nblocks) return;
    long row[NTB];
    bool ok[NTB];
    f32x4 xh[NTB][4];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const long t = (long)(blk0 + tb) * 16 + c;
        ok[tb] = t < M;
        row[tb] = ok[tb] ? t : M - 1;
        f32x4 xr[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) xr[kb] = ldg4(x + row[tb] * 64 + 16 * kb + 4 * g);
        float mean, rstd;
        ln_stats(xr, mean, rstd);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) xh[tb][kb] = (xr[kb] - splat4(mean)) * splat4(rstd);
    }
#pragma unroll 1
    for (int ob = 0; ob < 8; ++ob) {
        f32x4 aa[NTB], ag[NTB];
        const f32x4 ba = ldg4(b + 16 * ob + 4 * g), bg = ldg4(b + 128 + 16 * ob + 4 * g);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) { aa[tb] = ba; ag[tb] = bg; }
        lin_acc<4, NTB>(w + (long)ob * 4 * 256 + lane * 4, xh, aa);
        lin_acc<4, NTB>(w + (long)(ob + 8) * 4 * 256 + lane * 4, xh, ag);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (ok[tb]) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = aa[tb][e] * sigmoidf_fast(ag[tb][e]);
                stg4(u + row[tb] * 128 + 16 * ob + 4 * g, r);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// conv module, part 2: depthwise Conv1d k=31 (same padding) with eval-mode
// BatchNorm1d folded into taps/bias, then Swish.               conformer.py:40-48,165-169
// block = (32 positions of one sequence) x 128 channels; the (32+30) x 128 input tile is
// staged in LDS; each thread slides a 34-tap window over 4 outputs at a time.
// ---------------------------------------------------------------------------------
#define DW_TL 32
#define DW_K 31
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ u, float* __restrict__ out,
                                                     const float* __restrict__ dw_w,
                                                     const float* __restrict__ dw_b, TokMap m) {
    __shared__ __attribute__((aligned(16))) float tile[(DW_TL + DW_K - 1) * 128];
    const int n = blockIdx.x;
    const int l0 = blockIdx.y * DW_TL;
    const long nbase = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride;
    constexpr int ROWS = DW_TL + DW_K - 1;
    constexpr int NLD = (ROWS * 32 + 255) / 256;
    f32x4 stg[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {                         // all loads first, then the LDS stores (see stage_lds16)
        const int i = threadIdx.x + 256 * k, rr = i >> 5, qd = i & 31;
        const int l = l0 - (DW_K / 2) + rr;
        const bool inb = i < ROWS * 32 && l >= 0 && l < m.L;
        const int lc = l < 0 ? 0 : (l < m.L ? l : m.L - 1);
        stg[k] = ldg4(u + (nbase + (long)lc * m.lstride) * 128 + qd * 4);
        if (!inb) stg[k] = splat4(0.f);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < ROWS * 32) *reinterpret_cast<f32x4*>(&tile[(i >> 5) * 128 + (i & 31) * 4]) = stg[k];
    }
    const int ch = threadIdx.x & 127, sub = threadIdx.x >> 7;
    float wt[DW_K];
#pragma unroll
    for (int t = 0; t < DW_K; ++t) wt[t] = dw_w[t * 128 + ch];
    const float bias = dw_b[ch];
    __syncthreads();
#pragma unroll 1
    for (int og = 0; og < 4; ++og) {
        const int base = sub * 16 + og * 4;
        float acc[4] = {bias, bias, bias, bias};
#pragma unroll
        for (int kk = 0; kk < DW_K + 3; ++kk) {
            const float uv = tile[(base + kk) * 128 + ch];
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) {
                const int t = kk - oo;
                if (t >= 0 && t < DW_K) acc[oo] = fmaf(wt[t], uv, acc[oo]);
            }
        }
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) {
            const int l = l0 + base + oo;
            if (l < m.L) out[(nbase + (long)l * m.lstride) * 128 + ch] = swishf(acc[oo]);
        }
    }
}

// ---------------------------------------------------------------------------------
// conv module, part 3: pointwise 128->64 + bias + residual (in place)  conformer.py:170, 219
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pw2_kernel(float* __restrict__ x, const float* __restrict__ vin,
                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                  long M, int nblocks) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk0 = wave * NTB;
    if (blk0 >= nblocks) return;
    long row[NTB];
    bool ok[NTB];
    f32x4 xf[NTB][8];
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) {
        const long t = (long)(blk0 + tb) * 16 + c;
        ok[tb] = t < M;
        row[tb] = ok[tb] ? t : M - 1;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xf[tb][kb] = ldg4(vin + row[tb] * 128 + 16 * kb + 4 * g);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        const f32x4 bias = ldg4(b + 16 * ob + 4 * g);
        f32x4 acc[NTB];
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[tb] = bias;
        lin_acc<8, NTB>(w + (long)ob * 8 * 256 + lane * 4, xf, acc);
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
            if (ok[tb]) {
                float* p = x + row[tb] * 64 + 16 * ob + 4 * g;
                stg4(p, ldg4(p) + acc[tb]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
TokMap make_flat_map(long M) {
    TokMap m{};
    m.seq = 0;
    m.M = M;
    m.nblocks = (int)((M + 15) / 16);
    m.L = 0; m.Lb = 1; m.inner = 1; m.outer = 0; m.istride = 0; m.lstride = 0;
    return m;
}

TokMap make_seq_map(int N, int L, int inner, long outer, long istride, long lstride) {
    TokMap m{};
    m.seq = 1;
    m.L = L;
    m.Lb = (L + 15) / 16;
    m.nblocks = N * m.Lb;
    m.M = 0;
    m.inner = inner; m.outer = outer; m.istride = istride; m.lstride = lstride;
    return m;
}

size_t conf_qkv_floats(int N, int L) { return (size_t)N * 4 * (2 * ((L + 31) / 32)) * 256; }

static inline int grid_for_blocks(int nblocks) {
    const int waves = (nblocks + NTB - 1) / NTB;
    return (waves + 3) / 4;
}

void launch_dwconv(LaunchCtx ctx, const float* u, float* out, const float* dw_w, const float* dw_b, const TokMap& seq) {
    const int N = seq.nblocks / seq.Lb;
    dim3 dwgrid(N, (seq.L + DW_TL - 1) / DW_TL);
    LAUNCH(ctx, "dwconv", (dwconv_kernel<<<dwgrid, 256, 0, ctx.stream>>>(u, out, dw_w, dw_b, seq)));
}

void conformer_forward(LaunchCtx ctx, const ConfWeights& w, const ConfBuffers& b, const TokMap& seq, long M,
                       float* taps, bool outer_residual, const unsigned char* mask) {
    const TokMap flat = make_flat_map(M);
    const int N = seq.nblocks / seq.Lb;
    hipStream_t s = ctx.stream;
    const size_t tap_bytes = (size_t)M * 64 * sizeof(float);

    LAUNCH(ctx, "ffn", (ffn_kernel<false><<<grid_for_blocks(flat.nblocks), 256, 0, s>>>(
                           b.xa, b.xb, nullptr, nullptr, w.ff1_w1, w.ff1_b1, w.ff1_w2, w.ff1_b2, M,
                           flat.nblocks)));
    if (taps) hipMemcpyAsync(taps, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    LAUNCH(ctx, "qkv", (qkv_kernel<<<grid_for_blocks(seq.nblocks), 256, 0, s>>>(b.xb, seq, w.qkv_w, w.qkv_b,
                                                                                  b.q, b.k, b.v)));
    const int npairs = (seq.Lb + 1) / 2;
    const long items = (long)N * 4 * npairs;
    if (mask)
        LAUNCH(ctx, "attn", (attn_kernel<true><<<(unsigned)((items + 3) / 4), 256, 0, s>>>(
                                b.q, b.k, b.v, w.rel, w.max_pos, b.o, seq.L, seq.Lb, npairs, items, mask)));
    else
        LAUNCH(ctx, "attn", (attn_kernel<false><<<(unsigned)((items + 3) / 4), 256, 0, s>>>(
                                b.q, b.k, b.v, w.rel, w.max_pos, b.o, seq.L, seq.Lb, npairs, items, nullptr)));
    LAUNCH(ctx, "outproj",
           (outproj_kernel<<<grid_for_blocks(seq.nblocks), 256, 0, s>>>(b.xb, seq, b.o, w.wo, w.bo)));
    if (taps) hipMemcpyAsync(taps + (size_t)M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    LAUNCH(ctx, "pw1glu", (pw1glu_kernel<<<grid_for_blocks(flat.nblocks), 256, 0, s>>>(b.xb, b.u, w.pw1_w, w.pw1_b,
                                                                                        M, flat.nblocks)));
    launch_dwconv(ctx, b.u, b.w, w.dw_w, w.dw_b, seq);
    LAUNCH(ctx, "pw2", (pw2_kernel<<<grid_for_blocks(flat.nblocks), 256, 0, s>>>(b.xb, b.w, w.pw2_w, w.pw2_b, M,
                                                                                  flat.nblocks)));
    if (taps) hipMemcpyAsync(taps + (size_t)2 * M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    if (taps) {
        // test hook: the residual stream after ff2 but before post_norm (conformer.py:220)
        LAUNCH(ctx, "ffn", (ffn_kernel<false><<<grid_for_blocks(flat.nblocks), 256, 0, s>>>(
                               b.xb, taps + (size_t)3 * M * 64, nullptr, nullptr, w.ff2_w1, w.ff2_b1, w.ff2_w2,
                               w.ff2_b2, M, flat.nblocks)));
    }
    LAUNCH(ctx, "ffn_post", (ffn_kernel<true><<<grid_for_blocks(flat.nblocks), 256, 0, s>>>(
                                b.xb, b.xa, outer_residual ? b.xa : nullptr, w.post_gb, w.ff2_w1, w.ff2_b1, w.ff2_w2, w.ff2_b2, M,
                                flat.nblocks)));
}

// ---------------------------------------------------------------------------------
// MFMA convention self-test: D[16][16] = A[16][16*KB] * B[16*KB][16], operands given
// fragment-major exactly as the packer produces them (A = fm(A), B = fm(B^T)).
// ---------------------------------------------------------------------------------
__global__ void selftest_mfma_kernel(const float* __restrict__ a_fm, const float* __restrict__ b_fm,
                                     float* __restrict__ d, int KB) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    f32x4 acc = splat4(0.f);
    for (int kb = 0; kb < KB; ++kb) {
        const f32x4 a = ldg4(a_fm + kb * 256 + lane * 4);
        const f32x4 b = ldg4(b_fm + kb * 256 + lane * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(a[r], b[r], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + c] = acc[r];
}

void launch_selftest_mfma(hipStream_t s, const float* a_fm, const float* b_fm, float* d, int KB) {
    selftest_mfma_kernel<<<1, 64, 0, s>>>(a_fm, b_fm, d, KB);
}

// ---------------------------------------------------------------------------------
// ConformerBlock.forward (conformer.py:216-222) on a table of per-stage launchers (ConfStageTbl, kernels.h): the stage
// order, tap copies and buffer roles of the split-f16 path; which build each stage comes from is the caller's choice.
// residual stream: xa (block input, kept for the TSCB residual) -> ff1 -> xb, then attention / conv module in place on
// xb, ff2 + post norm (+ xa) -> xa.
// ---------------------------------------------------------------------------------
bool conformer_forward_tbl(LaunchCtx ctx, const ConfStageTbl& ff1, const ConfStageTbl& qkv, const ConfStageTbl& attn,
                           const ConfStageTbl& pw1, const ConfStageTbl& dwpw2, const ConfStageTbl& ff2,
                           const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b, const TokMap& seq, long M,
                           float* taps, bool outer_residual, const unsigned char* mask) {
    hipStream_t s = ctx.stream;
    const size_t tap_bytes = (size_t)M * 64 * sizeof(float);
    ff1.ffn(ctx, 1, false, b.xa, b.xb, nullptr, w, w16, M);
    if (taps) hipMemcpyAsync(taps, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);
    qkv.qkv(ctx, b.xb, seq, w, w16, b);
    attn.attn(ctx, b.xb, seq, w, w16, b, mask);
    if (taps) hipMemcpyAsync(taps + (size_t)M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);
    pw1.pw1glu(ctx, b.xb, w, w16, b, M);
    dwpw2.dwpw2(ctx, b.xb, seq, w, w16, b);
    if (taps) {
        hipMemcpyAsync(taps + (size_t)2 * M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);
        ff2.ffn(ctx, 2, true, b.xb, taps + (size_t)3 * M * 64, nullptr, w, w16, M);     // the ff2 tap: before post_norm
    }
    ff2.ffn(ctx, 2, false, b.xb, b.xa, outer_residual ? b.xa : nullptr, w, w16, M);
    return true;
}
