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

namespace X3_NS {

#define XNTB 2
#define XWAVES 8
#ifndef FFN_WAVES
#define FFN_WAVES 12          // waves per FeedForward block (one block per CU: 128 KB of weight images); three waves
                              // per SIMD at 150 VGPRs measured 7 % faster than two, a fourth would spill
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
#pragma unroll
        for (int tb = 0; tb < TNTB; ++tb) {
            const long t = ((long)tile * TNTB + tb) * 16 + c;
            ok[tb] = t < M;
            row[tb] = ok[tb] ? t : M - 1;
            f32x4 x[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) x[kb] = ldg4(xin + row[tb] * 64 + 16 * kb + 4 * g);
            ln_split(x, xbh[tb], xbl[tb]);      // the residual is re-read in the epilogue (L2 hit): 32 VGPRs saved
        }
        f32x4 y[TNTB][4];
#pragma unroll
        for (int tb = 0; tb < TNTB; ++tb)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) y[tb][ob] = splat4(0.f);

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
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
                y[tb][ob] = y[tb][ob] + *reinterpret_cast<const f32x4*>(&bias_l[256 + 16 * ob + 4 * g]) +
                            ldg4(xin + row[tb] * 64 + 16 * ob + 4 * g);
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
//   qimg, kimg   : [2*Lb2 blocks of 16 tokens][part 0..3][16 tokens][8 halfs]
//                  (parts 0,1 = hi of d 0..7 / 8..15, parts 2,3 = lo)
//   vimg         : [Lb2][hi|lo][64 lanes][8 halfs]   A operand of O^T = V^T P^T:
//                  lane (d, g) slot e <-> key 32*ip + 16*(e>>2) + 4*g + (e&3)
// LDS: weight image [12][2] = 48 KB + 1 KB transposition scratch per wave.
// ---------------------------------------------------------------------------------
struct QkvOut {
    _Float16 *qimg, *kimg, *vimg;
};

__global__ __launch_bounds__(512) void qkv_x3_kernel(const float* __restrict__ x, TokMap m, int Lb2,
                                                     const _Float16* __restrict__ wi, const float* __restrict__ b,
                                                     QkvOut o, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[24576 + 4352];   // 48 KB image + 8.5 KB scratch
    _Float16* w = wlds;                                        // 12*2*1024 halfs = 48 KB
    float* scratch = reinterpret_cast<float*>(wlds + 24576);   // 8 waves x 16 x 17 floats
    __shared__ __attribute__((aligned(16))) float bias_l[192];
    stage_lds16<3072, 512>(wi, w);
    for (int i = threadIdx.x; i < 192; i += blockDim.x) bias_l[i] = b[i];
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
#pragma unroll 2
        for (int ob = 0; ob < 12; ++ob) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(&bias_l[16 * ob + 4 * g]);
            f32x4 acc[XNTB];
#pragma unroll
            for (int tb = 0; tb < XNTB; ++tb) acc[tb] = bias;
            lin_acc_x3<2, XNTB>(w + ob * 2048 + lane * 8, xbh, xbl, acc);
            const int which = ob >> 2, h = ob & 3;
            const long nh = (long)n * 4 + h;
            if (which < 2) {
                // Q / K image: per 16-token block [part 0..3][token 0..15][8 halfs], parts 0,1 = hi of
                // d 0..7 / 8..15, parts 2,3 = lo: a 16-token operand is one lane-linear 1 KiB read
                _Float16* img = which == 0 ? o.qimg : o.kimg;
#pragma unroll
                for (int tb = 0; tb < XNTB; ++tb) {
                    f16x4 hi, lo;
                    split4(acc[tb], hi, lo);
                    _Float16* blk = img + ((nh * 2 * Lb2) + 2 * ip + tb) * 512 + 4 * (g & 1);
                    *reinterpret_cast<f16x4*>(blk + (((g >> 1)) * 16 + c) * 8) = hi;
                    *reinterpret_cast<f16x4*>(blk + (((g >> 1) + 2) * 16 + c) * 8) = lo;
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
// Attention core (see attn_kernel in conformer.hip for the algorithm).  Barrier-free:
// every wave is independent and owns ATT_NQ consecutive 16-query blocks of one (sequence,
// head), processed as pairs so that two independent dependency chains (MFMA -> LDS skew
// -> softmax -> MFMA) are in flight per wave.  K / V / E operand images are read straight
// from L2 as lane-linear 1 KiB fragments and shared by the wave's query blocks; LDS is only the wave-private
// Toeplitz skew scratch (2 x 6.4 KB per wave).
// Contraction slots of the 32-wide MFMA: lane groups g = 0,1 carry the hi half of K (resp. E),
// g = 2,3 the lo half, both over d = 8*(g&1) + e; B carries Q_hi in both halves (MFMA 1) then
// Q_lo (MFMA 2): two MFMAs give (K_hi + K_lo) . (Q_hi + Q_lo).
// ---------------------------------------------------------------------------------
#ifndef XCD_ORDER
#define XCD_ORDER 1          // XCD-contiguous work order for attention and the depthwise kernel (0 = dispatch order)
#endif
#ifndef ATT_SKIP_DEAD
#define ATT_SKIP_DEAD 1      // do not compute the non-existent second query block of a sequence's last pair
#endif
#ifndef ATTN_FUSE_OUT
#define ATTN_FUSE_OUT 1      // attention + to_out + residual in one kernel (0 = attn_x3_kernel, then outproj_x3_kernel)
#endif
#ifndef ATTN32
#define ATTN32 2             // 2 = software-pipelined attention on 32x32x16 MFMAs (attn32_x3.hip: attn_sp_out_x3_kernel; masked
                             // calls take attn32_out_x3_kernel<true>); 1 = the un-pipelined 32x32x16 kernel; 0 = the 16x16x32 kernels below
#endif
#ifndef DWPW2_SLIDE
#define DWPW2_SLIDE 1        // sliding-window depthwise + pointwise kernel (0 = one block per 32-position tile)
#endif
#define RSTRIDE_X 20
// Query blocks per wave / waves per SIMD the kernel is compiled for.  Measured in one session: a lone wave per
// SIMD 9.8 ms, two waves (4 query blocks each, 216 VGPRs) 6.0 ms, three waves (2 query blocks = one pair each,
// 168 VGPRs, no spill) 5.4 ms: the kernel is a long dependency chain (MFMA -> LDS skew -> MFMA -> softmax -> MFMA)
// that more resident waves hide better than more work per wave does, even though K / V / E operand reuse halves.
#ifndef ATT_NQ
#define ATT_NQ 2
#endif
#ifndef ATT_WAVES
#define ATT_WAVES 3
#endif

struct AttState {
    float m, run, l;      // reference level, running max relative to it, denominator (relative to m)
    f32x4 o;
};

// Online softmax with a STALE reference (T13-style), arranged so the common path has no per-score
// subtract or add at all:
//   * the rel-pos accumulator starts at -m_ref, so R' = E q - m_ref comes out of the MFMA for free;
//   * the skewed R' is read from LDS straight INTO the score accumulators, and the K q MFMAs
//     accumulate on top: s = K q + E q - m_ref;
//   * p = exp2(s) directly (scores are in log2 units: log2(e) is folded into the q projection).
// m_ref (st.m) is the reference level of the query block (0 before the first chunk); st.run is the
// true running maximum RELATIVE to it.  The reference is kept inside the band
//   ATT_LO < run <= ATT_HI      (-4, +12]
// +12 bounds p <= 2^12 (inside fp16 range for the split-product P V MFMAs), -4 keeps the largest p
// >= 2^-4 so the fp16 lo half of P stays normal (a reference that is too HIGH would silently cost
// mantissa bits).  Leaving the band takes the re-reference path (m_ref += run, p = exp2(s - run),
// o and l rescaled by exp2(-run)); the branch is wave-uniform (__any) and exact for every lane.
// Scaling an empty accumulator is skipped (0 * exp2(+big) would be NaN).
#define ATT_HI 12.0f
#define ATT_LO -4.0f
// MASK (ConformerBlock.forward(x, mask), conformer.py:113-126): mk = the sequence's [L] byte mask, qvalid = this lane's
// query is unmasked.  A pair keeps its score only if query AND key are unmasked; the reference fills every other score
// with -finfo.max, so an unmasked query ignores masked keys (p = 0) and a masked query attends uniformly to all L keys
// (all its scores equal: 0 here).  A query whose keys so far were all masked has run = -inf ("dead"): it contributes
// p = 0 and keeps its reference level untouched.
template <bool FULL, bool MASK = false>
__device__ __forceinline__ void att_softmax(f32x4 (&s)[4], int c, int g, int j0, int nb, int L, AttState& st,
                                            const unsigned char* __restrict__ mk = nullptr, bool qvalid = true) {
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        if (FULL || jb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = j0 + 16 * jb + 4 * g + r;
                if (!FULL) s[jb][r] = key < L ? s[jb][r] : -INFINITY;   // select, no branch
                if (MASK && (FULL || key < L)) s[jb][r] = qvalid ? (mk[key] ? s[jb][r] : -INFINITY) : 0.f;
                mx = fmaxf(mx, s[jb][r]);
            }
        }
    }
    const float run = fmaxf(st.run, red_g_max(mx));
    const bool dead = MASK && run == -INFINITY;
    const bool drift = !dead && (run > ATT_HI || run < ATT_LO);
    float psum = 0.f;
    if (__any(drift)) {                                  // rare: re-reference this query block to its running maximum
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

// Operand fetches are raw buffer loads (wave-uniform descriptor in SGPRs + 32-bit lane offset + scalar block offset):
// no 64-bit VGPR address arithmetic per fetch.  The distance table is read from its four-plane image (kernels.h:
// rel_planes; plane g = lane group g's slice, rows in reversed distance order), so the 16 consecutive distances of an
// operand block are 256 contiguous bytes per lane group instead of a gather over 64-byte rows.
typedef unsigned att_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t att_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f16x8 att_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
struct AttCtx {
    __amdgpu_buffer_rsrc_t qr, kr, vr, er;
    unsigned lane16, eoff;            // lane * 16 bytes; g * plane bytes
    float *RA, *RB;
    int qoff1, qoff2, nblk16, Lb, Lb2, L, max_pos, c, g;
    const unsigned char* mk;          // this sequence's attention mask row (MASK variants only)
};

// one 64-key chunk for the wave's (up to) four query blocks; FULL = all 64 keys exist; HASB = the second block of
// the pair exists (false only for the last pair of a sequence with an odd number of 16-token blocks - L = 321 -> 21,
// L = 101 -> 7: one of 8 query-block slots of a frequency-axis sequence - whose B half is then not computed at all)
template <bool FULL, bool HASB = true, bool MASK = false>
__device__ __forceinline__ void att_chunk(const AttCtx& a, int ibb, int j0, AttState (&st)[ATT_NQ]) {
    const int nb = FULL ? 4 : ((a.L - j0 + 15) >> 4);     // live 16-key blocks (tail chunk: 1..4)
    const int c = a.c, g = a.g;
    f16x8 kf[4], vh[2], vl[2];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        int kb = (j0 >> 4) + jb;
        if (!FULL) kb = kb < a.nblk16 ? kb : a.nblk16 - 1;
        kf[jb] = att_ld(a.kr, a.lane16, (unsigned)kb * 1024u);
    }
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
        int pr = (j0 >> 5) + mp;
        if (!FULL) pr = pr < a.Lb2 ? pr : a.Lb2 - 1;
        vh[mp] = att_ld(a.vr, a.lane16, (unsigned)pr * 2048u);
        vl[mp] = att_ld(a.vr, a.lane16 + 1024u, (unsigned)pr * 2048u);
    }
#pragma unroll
    for (int pair = 0; pair < ATT_NQ / 2; ++pair) {
        const int ibA = ibb + 2 * pair;
        if (ibA >= a.Lb) break;
        const int qB = ibA + 1 < a.Lb ? ibA + 1 : a.Lb - 1;
        const f16x8 qA1 = att_ld(a.qr, (unsigned)a.qoff1 * 2u, (unsigned)ibA * 1024u);
        const f16x8 qA2 = att_ld(a.qr, (unsigned)a.qoff2 * 2u, (unsigned)ibA * 1024u);
        f16x8 qB1 = qA1, qB2 = qA2;
        if (HASB) {
            qB1 = att_ld(a.qr, (unsigned)a.qoff1 * 2u, (unsigned)qB * 1024u);
            qB2 = att_ld(a.qr, (unsigned)a.qoff2 * 2u, (unsigned)qB * 1024u);
        }
        // relative-position window of the pair: 6 row blocks starting at rminA = 16 ibA - j0 - 63;
        // block A uses window blocks 0..4 as its cb 0..4, block B (16 queries later) blocks 1..5
        f16x8 ef[6];
        const int rminA = ibA * 16 - j0 - 63;
#pragma unroll
        for (int we = 0; we < 6; ++we) {
            int rw = a.max_pos - (rminA + 16 * we + c);                    // row of the reversed-order planes
            rw = rw < 0 ? 0 : (rw > 2 * a.max_pos ? 2 * a.max_pos : rw);
            ef[we] = att_ld(a.er, (unsigned)rw * 16u + a.eoff, 0);
        }
        AttState& sa = st[2 * pair];
        AttState& sb = st[2 * pair + 1];
        wave_lds_fence();                                 // previous pair's skew reads are done
#pragma unroll
        for (int we = 0; we < 6; ++we) {
            if (we < 5 && (FULL || we >= 4 - nb)) {       // cb = we for block A
                f32x4 rt = mfma32h(ef[we], qA1, splat4(-sa.m));
                rt = mfma32l(ef[we], qA2, rt);
#pragma unroll
                for (int r = 0; r < 4; ++r) a.RA[(16 * we + 4 * g + r) * RSTRIDE_X + c] = rt[r];
            }
            if (HASB && we >= 1 && (FULL || we - 1 >= 4 - nb)) {  // cb = we - 1 for block B
                f32x4 rt = mfma32h(ef[we], qB1, splat4(-sb.m));
                rt = mfma32l(ef[we], qB2, rt);
#pragma unroll
                for (int r = 0; r < 4; ++r) a.RB[(16 * (we - 1) + 4 * g + r) * RSTRIDE_X + c] = rt[r];
            }
        }
        wave_lds_fence();
        // the skewed (E q - m_ref) tile is read straight into the score accumulators (every index is
        // in range, so the read is unconditional); K q accumulates on top of it
        f32x4 sA[4], sB[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            if (FULL || jb < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sA[jb][r] = a.RA[(c - 16 * jb - 4 * g - r + 63) * RSTRIDE_X + c];
                    if (HASB) sB[jb][r] = a.RB[(c - 16 * jb - 4 * g - r + 63) * RSTRIDE_X + c];
                }
            } else {
                sA[jb] = splat4(0.f);
                sB[jb] = splat4(0.f);
            }
            if (!HASB) sB[jb] = splat4(0.f);
        }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
            if (FULL || jb < nb) { sA[jb] = mfma32h(kf[jb], qA1, sA[jb]); if (HASB) sB[jb] = mfma32h(kf[jb], qB1, sB[jb]); }
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
            if (FULL || jb < nb) { sA[jb] = mfma32l(kf[jb], qA2, sA[jb]); if (HASB) sB[jb] = mfma32l(kf[jb], qB2, sB[jb]); }
        if (MASK) {
            const int la = ibA * 16 + c, lb = qB * 16 + c;
            att_softmax<FULL, true>(sA, c, g, j0, nb, a.L, sa, a.mk, a.mk[la < a.L ? la : a.L - 1] != 0);
            if (HASB) att_softmax<FULL, true>(sB, c, g, j0, nb, a.L, sb, a.mk, a.mk[lb < a.L ? lb : a.L - 1] != 0);
        } else {
            att_softmax<FULL>(sA, c, g, j0, nb, a.L, sa);
            if (HASB) att_softmax<FULL>(sB, c, g, j0, nb, a.L, sb);
        }
#pragma unroll
        for (int mp = 0; mp < 2; ++mp) {
            if (FULL || 2 * mp < nb) {
                f16x8 pAh, pAl, pBh, pBl;
                split8(sA[2 * mp], sA[2 * mp + 1], pAh, pAl);
                if (HASB) split8(sB[2 * mp], sB[2 * mp + 1], pBh, pBl);
                st[2 * pair].o = mfma32h(vh[mp], pAh, st[2 * pair].o);
                if (HASB) st[2 * pair + 1].o = mfma32h(vh[mp], pBh, st[2 * pair + 1].o);
                st[2 * pair].o = mfma32l(vh[mp], pAl, st[2 * pair].o);
                if (HASB) st[2 * pair + 1].o = mfma32l(vh[mp], pBl, st[2 * pair + 1].o);
                st[2 * pair].o = mfma32l(vl[mp], pAh, st[2 * pair].o);
                if (HASB) st[2 * pair + 1].o = mfma32l(vl[mp], pBh, st[2 * pair + 1].o);
            }
        }
    }
}

__global__ __launch_bounds__(256, ATT_WAVES) void attn_x3_kernel(QkvOut io, const _Float16* __restrict__ eimg, int max_pos,
                                                         float* __restrict__ o, int L, int Lb, int Lb2, int nqg,
                                                         long total) {
    __shared__ float rbuf[4][2][80 * RSTRIDE_X + 4];   // +4: keeps RA/RB 1604 dwords apart, which no ds_read2* form can span, so each skew read
                                                       // lands directly in its accumulator register (paired A/B reads cost a v_mov per value)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroup b runs on XCD b % 8 (each XCD has its own L2).  The query-block waves of one (sequence, head)
    // read the same K / V images, so every XCD walks a CONTIGUOUS range of work items: the grid is a multiple
    // of 8 blocks and block b takes logical block (b % 8) * (grid / 8) + b / 8.
    const long lblk = XCD_ORDER ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long item = lblk * 4 + wv;
    if (item >= total) return;                            // no block-level synchronisation below
    const int nh = __builtin_amdgcn_readfirstlane((int)((unsigned)item / (unsigned)nqg));   // wave-uniform: SGPR bases
    const int ibb = ((int)item - nh * nqg) * ATT_NQ;      // first query block of this wave
    AttCtx a;
    a.c = lane & 15; a.g = lane >> 4;
    a.RA = rbuf[wv][0]; a.RB = rbuf[wv][1];
    a.nblk16 = 2 * Lb2; a.Lb = Lb; a.Lb2 = Lb2; a.L = L; a.max_pos = max_pos;
    a.qr = att_rsrc(io.qimg + (long)nh * a.nblk16 * 512, (unsigned)a.nblk16 * 1024u);
    a.kr = att_rsrc(io.kimg + (long)nh * a.nblk16 * 512, (unsigned)a.nblk16 * 1024u);
    a.vr = att_rsrc(io.vimg + (long)nh * Lb2 * 1024, (unsigned)Lb2 * 2048u);
    a.er = att_rsrc(eimg, (unsigned)(2 * max_pos + 1) * 64u);
    a.lane16 = (unsigned)lane * 16u;
    a.eoff = (unsigned)a.g * (unsigned)(2 * max_pos + 1) * 16u;
    a.qoff1 = ((a.g & 1) * 16 + a.c) * 8; a.qoff2 = ((2 + (a.g & 1)) * 16 + a.c) * 8;
    a.mk = nullptr;

    AttState st[ATT_NQ];
#pragma unroll
    for (int i = 0; i < ATT_NQ; ++i) {
        st[i].m = 0.f; st[i].run = -INFINITY; st[i].l = 0.f; st[i].o = splat4(0.f);
    }

    const int nfull = L >> 6;
#pragma unroll 1
    for (int ch = 0; ch < nfull; ++ch) att_chunk<true>(a, ibb, ch * 64, st);
    if (L & 63) att_chunk<false>(a, ibb, nfull * 64, st);

#pragma unroll
    for (int i = 0; i < ATT_NQ; ++i) {
        const int ib = ibb + i;
        if (ib < Lb) stg4(o + (nh * Lb + ib) * 256 + lane * 4, st[i].o * splat4(__builtin_amdgcn_rcpf(st[i].l)));
    }
}

// ---------------------------------------------------------------------------------
// Attention core + to_out + bias + residual in ONE kernel (the default): a block is the four heads of one
// (sequence, query-block pair), one head per wave, same barrier-free chunk loop as attn_x3_kernel.  When a wave's
// head is done its normalised O tile (2 x 1 KB of fp32 C-fragments, which ARE the B-fragments of k-block h of
// to_out) goes into its own - now idle - skew scratch; after the block's only barrier wave w evaluates output
// block w of  x += Wo . concat_h(O_h) + bo  for the pair's 32 tokens and updates the residual stream in place.
// O never touches HBM (1 row written + 1 row read per token and conformer before), the to_out launch and its
// second pass over the residual are gone, and the arithmetic is bit-identical to attn_x3 + outproj_x3.
// Blocks run in XCD-contiguous order: the query pairs of a sequence follow each other on one XCD, so the
// K / V images of its four heads are fetched into that L2 once.
// ---------------------------------------------------------------------------------
template <bool MASK>
__global__ __launch_bounds__(256, ATT_WAVES) void attn_out_x3_kernel(QkvOut io, const _Float16* __restrict__ eimg,
                                                             int max_pos, float* __restrict__ x, TokMap m,
                                                             const _Float16* __restrict__ woi,
                                                             const float* __restrict__ bo, int Lb2, int nqg,
                                                             long nblocks, const unsigned char* __restrict__ mask) {
    __shared__ __attribute__((aligned(16))) float rbuf[4][2][80 * RSTRIDE_X + 4];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long lblk = XCD_ORDER ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    if (lblk >= nblocks) return;                          // padding blocks of the rounded-up grid (block-uniform)
    // (the division runs on the VALU; readfirstlane returns the - uniform - quotient to an SGPR so that every base
    // derived from it stays scalar)
    const int n = __builtin_amdgcn_readfirstlane((int)((unsigned)lblk / (unsigned)nqg));
    const int ibb = ((int)lblk - n * nqg) * ATT_NQ;       // first query block of the pair
    const long nh = (long)n * 4 + wv;                     // this wave's head
    const int L = m.L, Lb = m.Lb;
    AttCtx a;
    a.c = lane & 15; a.g = lane >> 4;
    a.RA = rbuf[wv][0]; a.RB = rbuf[wv][1];
    a.nblk16 = 2 * Lb2; a.Lb = Lb; a.Lb2 = Lb2; a.L = L; a.max_pos = max_pos;
    a.qr = att_rsrc(io.qimg + nh * a.nblk16 * 512, (unsigned)a.nblk16 * 1024u);
    a.kr = att_rsrc(io.kimg + nh * a.nblk16 * 512, (unsigned)a.nblk16 * 1024u);
    a.vr = att_rsrc(io.vimg + nh * Lb2 * 1024, (unsigned)Lb2 * 2048u);
    a.er = att_rsrc(eimg, (unsigned)(2 * max_pos + 1) * 64u);
    a.lane16 = (unsigned)lane * 16u;
    a.eoff = (unsigned)a.g * (unsigned)(2 * max_pos + 1) * 16u;
    a.qoff1 = ((a.g & 1) * 16 + a.c) * 8; a.qoff2 = ((2 + (a.g & 1)) * 16 + a.c) * 8;
    a.mk = MASK ? mask + (long)n * L : nullptr;

    AttState st[ATT_NQ];
#pragma unroll
    for (int i = 0; i < ATT_NQ; ++i) {
        st[i].m = 0.f; st[i].run = -INFINITY; st[i].l = 0.f; st[i].o = splat4(0.f);
    }
    const int nfull = L >> 6;
    if (!ATT_SKIP_DEAD || ibb + 1 < Lb) {                  // block-uniform: both query blocks of the pair exist
#pragma unroll 1
        for (int ch = 0; ch < nfull; ++ch) att_chunk<true, true, MASK>(a, ibb, ch * 64, st);
        if (L & 63) att_chunk<false, true, MASK>(a, ibb, nfull * 64, st);
    } else {
#pragma unroll 1
        for (int ch = 0; ch < nfull; ++ch) att_chunk<true, false, MASK>(a, ibb, ch * 64, st);
        if (L & 63) att_chunk<false, false, MASK>(a, ibb, nfull * 64, st);
        st[1].l = 1.f;                                     // never accumulated: keep the (unused) stash finite
    }

    // the to_out operands of this wave's output block are fetched now: their L2 latency hides behind the barrier
    const _Float16* wp = woi + wv * 2048 + lane * 8;       // [ob = wv][m][hi | lo][64][8]
    const f16x8 ah0 = *reinterpret_cast<const f16x8*>(wp), al0 = *reinterpret_cast<const f16x8*>(wp + 512);
    const f16x8 ah1 = *reinterpret_cast<const f16x8*>(wp + 1024), al1 = *reinterpret_cast<const f16x8*>(wp + 1536);
    const f32x4 bias = ldg4(bo + 16 * wv + 4 * a.g);

    wave_lds_fence();                                     // this wave's last skew reads are done
    f32x4* stash = reinterpret_cast<f32x4*>(&rbuf[0][0][0]);       // O tile of (head h, block i) at [(h * 2 + i) * 64 + lane]
    constexpr int HSTRIDE = 2 * (80 * RSTRIDE_X + 4) / 4;          // float4s between two waves' scratch areas
#pragma unroll
    for (int i = 0; i < ATT_NQ; ++i)
        stash[wv * HSTRIDE + i * 64 + lane] = st[i].o * splat4(__builtin_amdgcn_rcpf(st[i].l));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ATT_NQ; ++i) {
        const int ib = ibb + i, l = ib * 16 + a.c;
        f16x8 bh[1][2], bl[1][2];
        split8(stash[0 * HSTRIDE + i * 64 + lane], stash[1 * HSTRIDE + i * 64 + lane], bh[0][0], bl[0][0]);
        split8(stash[2 * HSTRIDE + i * 64 + lane], stash[3 * HSTRIDE + i * 64 + lane], bh[0][1], bl[0][1]);
        f32x4 acc = bias;                                  // same product order as lin_acc_x3 / outproj_x3_kernel
        acc = mfma32h(ah0, bh[0][0], acc);
        acc = mfma32l(ah0, bl[0][0], acc);
        acc = mfma32l(al0, bh[0][0], acc);
        acc = mfma32h(ah1, bh[0][1], acc);
        acc = mfma32l(ah1, bl[0][1], acc);
        acc = mfma32l(al1, bh[0][1], acc);
        if (ib < Lb && l < L) {
            const long row = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride + (long)l * m.lstride;
            float* p = x + row * 64 + 16 * wv + 4 * a.g;
            stg4(p, ldg4(p) + acc);
        }
    }
}

// ---------------------------------------------------------------------------------
// to_out + bias + residual (in place); O arrives as fp32 C-fragments per head.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void outproj_x3_kernel(float* __restrict__ x, TokMap m,
                                                         const float* __restrict__ o,
                                                         const _Float16* __restrict__ wi,
                                                         const float* __restrict__ bo, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[8192];
    __shared__ __attribute__((aligned(16))) float bias_l[64];
    stage_lds16<1024, 512>(wi, wlds);                        // [4][2] image = 16 KB
    for (int i = threadIdx.x; i < 64; i += blockDim.x) bias_l[i] = bo[i];
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
            const f32x4 bias = *reinterpret_cast<const f32x4*>(&bias_l[16 * ob + 4 * g]);
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
#ifndef DP_WAVES
#define DP_WAVES 4              // waves per block: 4 (two 16-token groups x 128 channels); 8 (four 8-token groups, 108
                                // VGPRs, twice the resident waves) measured 23 % slower: the per-thread tap loads and
                                // window warm-up are amortised over half as many outputs
#endif
__global__ __launch_bounds__(64 * DP_WAVES) void dwpw2_x3_kernel(float* __restrict__ x, const float* __restrict__ u,
                                                       const float* __restrict__ dw_w,
                                                       const float* __restrict__ dw_b,
                                                       const _Float16* __restrict__ w2i,
                                                       const float* __restrict__ b2, TokMap m, int nseq, int ntl) {
    __shared__ __attribute__((aligned(16))) float utile[(DP_TL + DP_K - 1) * 128];
    __shared__ __attribute__((aligned(16))) _Float16 vth[DP_TL * DP_VS];
    __shared__ __attribute__((aligned(16))) _Float16 vtl[DP_TL * DP_VS];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4, wv = tid >> 6;
    // XCD-contiguous order with the l-tile fastest: consecutive tiles of a sequence (which share 30 halo rows of
    // u) run back to back on one XCD, so the halo is an L2 hit instead of a second HBM read
    int n, l0;
    if (XCD_ORDER) {
        const int lt = (int)(((long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) % ntl);
        n = (int)(((long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) / ntl);
        if (n >= nseq) return;                              // padding blocks of the rounded-up grid (block-uniform)
        l0 = lt * DP_TL;
    } else {
        n = blockIdx.x;
        l0 = blockIdx.y * DP_TL;
    }
    const long nbase = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride;

    // pointwise operands for this wave: token block tb, output blocks ob0 .. ob0 + NOB - 1
    constexpr int NTHR = 64 * DP_WAVES, NOB = 8 / DP_WAVES, TOK = 4096 / NTHR;   // tokens per thread in the depthwise phase
    const int tb = wv / (DP_WAVES / 2), ob0 = (wv % (DP_WAVES / 2)) * NOB;
    f16x8 ah[NOB][4], al[NOB][4];
#pragma unroll
    for (int o = 0; o < NOB; ++o)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const _Float16* wp = w2i + ((ob0 + o) * 4 + mm) * 1024 + lane * 8;
            ah[o][mm] = *reinterpret_cast<const f16x8*>(wp);
            al[o][mm] = *reinterpret_cast<const f16x8*>(wp + 512);
        }

    constexpr int ROWS = DP_TL + DP_K - 1;
    constexpr int NLD = (ROWS * 32 + NTHR - 1) / NTHR;
    f32x4 stg[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {                         // all loads first (see stage_lds16)
        const int i = tid + NTHR * k, rr = i >> 5, qd = i & 31;
        const int l = l0 - (DP_K / 2) + rr;
        const bool inb = i < ROWS * 32 && l >= 0 && l < m.L;   // 'same' zero padding outside the sequence
        const int lc = l < 0 ? 0 : (l < m.L ? l : m.L - 1);
        stg[k] = ldg4(u + (nbase + (long)lc * m.lstride) * 128 + qd * 4);
        if (!inb) stg[k] = splat4(0.f);
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + NTHR * k;
        if (i < ROWS * 32) *reinterpret_cast<f32x4*>(&utile[(i >> 5) * 128 + (i & 31) * 4]) = stg[k];
    }
    const int chn = tid & 127, sub = tid >> 7;
    float wt[DP_K];
#pragma unroll
    for (int t = 0; t < DP_K; ++t) wt[t] = dw_w[t * 128 + chn];
    const float bias = dw_b[chn];
    // B-operand order inside a row: channel 32m + 16h + 4gq + r  ->  32m + 8gq + 4h + r
    const int vcol = (chn & ~31) + ((chn >> 2) & 3) * 8 + ((chn >> 4) & 1) * 4 + (chn & 3);
    __syncthreads();
#pragma unroll 1
    for (int og = 0; og < TOK / 4; ++og) {
        const int base = sub * TOK + og * 4;
        float acc[4] = {bias, bias, bias, bias};
#pragma unroll
        for (int kk = 0; kk < DP_K + 3; ++kk) {
            const float uv = utile[(base + kk) * 128 + chn];
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) {
                const int t = kk - oo;
                if (t >= 0 && t < DP_K) acc[oo] = fmaf(wt[t], uv, acc[oo]);
            }
        }
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) {
            const float v = swishf(acc[oo]);
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            vth[(base + oo) * DP_VS + vcol] = hi;
            vtl[(base + oo) * DP_VS + vcol] = lo;
        }
    }
    __syncthreads();

    f32x4 acc2[NOB];
#pragma unroll
    for (int o = 0; o < NOB; ++o) acc2[o] = ldg4(b2 + 16 * (ob0 + o) + 4 * g);
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        const f16x8 bh = *reinterpret_cast<const f16x8*>(&vth[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(&vtl[(16 * tb + c) * DP_VS + 32 * mm + 8 * g]);
#pragma unroll
        for (int o = 0; o < NOB; ++o) acc2[o] = mfma32h(ah[o][mm], bh, acc2[o]);
#pragma unroll
        for (int o = 0; o < NOB; ++o) acc2[o] = mfma32l(ah[o][mm], bl, acc2[o]);
#pragma unroll
        for (int o = 0; o < NOB; ++o) acc2[o] = mfma32l(al[o][mm], bh, acc2[o]);
    }
    const int l = l0 + 16 * tb + c;
    if (l < m.L) {
        float* xr = x + (nbase + (long)l * m.lstride) * 64;
#pragma unroll
        for (int o = 0; o < NOB; ++o) {
            float* p = xr + 16 * (ob0 + o) + 4 * g;
            stg4(p, ldg4(p) + acc2[o]);
        }
    }
}

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
#ifndef DS_OCC
#define DS_OCC 2             // blocks (= waves per SIMD) the register allocation is sized for
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

    // pointwise operands of this wave: token block tb, output blocks ob0, ob0 + 1 (held for the whole segment)
    const int tb = wv >> 1, ob0 = (wv & 1) * 2;
    f16x8 ah[2][4], al[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const _Float16* wp = w2i + ((ob0 + o) * 4 + mm) * 1024 + lane * 8;
            ah[o][mm] = *reinterpret_cast<const f16x8*>(wp);
            al[o][mm] = *reinterpret_cast<const f16x8*>(wp + 512);
        }
    f32x4 bias2[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) bias2[o] = ldg4(b2 + 16 * (ob0 + o) + 4 * g);

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
#pragma unroll
        for (int o = 0; o < 2; ++o) xold[o] = ldg4(xr + 16 * (ob0 + o) + 4 * g);

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

static int ffn_grid(int ntiles) {                         // one persistent FeedForward block per CU
    const int want = (ntiles + FFN_WAVES - 1) / FFN_WAVES;
    return want < 256 ? (want > 0 ? want : 1) : 256;
}

void conformer_forward_x3(LaunchCtx ctx, const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b,
                          const TokMap& seq, long M, float* taps, bool outer_residual, const unsigned char* mask) {
    hipStream_t s = ctx.stream;
    const int N = seq.nblocks / seq.Lb;
    const int Lb2 = (seq.Lb + 1) / 2;
    const int flat_blocks = (int)((M + 15) / 16);
    const int flat_tiles = (flat_blocks + XNTB - 1) / XNTB;
    const size_t tap_bytes = (size_t)M * 64 * sizeof(float);
    QkvOut io;
    io.qimg = reinterpret_cast<_Float16*>(b.q);
    io.kimg = reinterpret_cast<_Float16*>(b.k);
    io.vimg = reinterpret_cast<_Float16*>(b.v);

    LAUNCH(ctx, "ffn", (ffn_x3_kernel<false, 2, FFN_WAVES><<<ffn_grid(flat_tiles), 64 * FFN_WAVES, 0, s>>>(
                           b.xa, b.xb, nullptr, nullptr, w16.ff1_w1, w.ff1_b1, w16.ff1_w2, w.ff1_b2, M, flat_tiles)));
    if (taps) hipMemcpyAsync(taps, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

#if ATTN32
    launch_qkv32_x3(ctx, b.xb, seq, w16.qkv_w, w.qkv_b, io.qimg, io.kimg, io.vimg);
    if (ATTN32 == 2 && !mask)
        launch_attn_sp_out_x3(ctx, io.qimg, io.kimg, io.vimg, w16.rel_planes, w.max_pos, b.xb, seq, w16.wo, w.bo);
    else
        launch_attn32_out_x3(ctx, io.qimg, io.kimg, io.vimg, w16.rel_planes, w.max_pos, b.xb, seq, w16.wo, w.bo, mask);
#else
    const int qtiles = N * Lb2;
    LAUNCH(ctx, "qkv", (qkv_x3_kernel<<<persistent_grid(qtiles, 2), 512, 0, s>>>(
                           b.xb, seq, Lb2, w16.qkv_w, w.qkv_b, io, qtiles)));
    {
        const int nqg = (seq.Lb + ATT_NQ - 1) / ATT_NQ;
#if ATTN_FUSE_OUT
        static_assert(ATT_NQ == 2, "the fused to_out epilogue stashes one pair of O tiles per head");
        const long nb = (long)N * nqg;
        const unsigned agrid = XCD_ORDER ? (unsigned)(((nb + 7) / 8) * 8) : (unsigned)nb;
        if (mask)
            LAUNCH(ctx, "attn_out", (attn_out_x3_kernel<true><<<agrid, 256, 0, s>>>(io, w16.rel_planes, w.max_pos, b.xb, seq, w16.wo,
                                                                                  w.bo, Lb2, nqg, nb, mask)));
        else
            LAUNCH(ctx, "attn_out", (attn_out_x3_kernel<false><<<agrid, 256, 0, s>>>(io, w16.rel_planes, w.max_pos, b.xb, seq, w16.wo,
                                                                                   w.bo, Lb2, nqg, nb, nullptr)));
    }
#else
        const long waves = (long)N * 4 * nqg;
        const unsigned ablk = (unsigned)((waves + 3) / 4);
        LAUNCH(ctx, "attn", (attn_x3_kernel<<<XCD_ORDER ? ((ablk + 7) / 8) * 8 : ablk, 256, 0, s>>>(
                                io, w16.rel_planes, w.max_pos, b.o, seq.L, seq.Lb, Lb2, nqg, waves)));
    }
    const int otiles = (seq.nblocks + XNTB - 1) / XNTB;
    LAUNCH(ctx, "outproj", (outproj_x3_kernel<<<persistent_grid(otiles, 2), 512, 0, s>>>(b.xb, seq, b.o, w16.wo,
                                                                                           w.bo, otiles)));
#endif
#endif
    if (taps) hipMemcpyAsync(taps + (size_t)M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);

    LAUNCH(ctx, "pw1glu", (pw1glu_x3_kernel<<<persistent_grid(flat_tiles, 2), 512, 0, s>>>(
                              b.xb, b.u, w16.pw1_w, w.pw1_b, M, flat_tiles)));
    const int ntl = (seq.L + DP_TL - 1) / DP_TL;
#if DWPW2_SLIDE
    {
        const int nsegs = (ntl + DS_SEG - 1) / DS_SEG;
        const long items = (long)N * nsegs;
        const unsigned grid = XCD_ORDER ? (unsigned)(((items + 7) / 8) * 8) : (unsigned)items;
        LAUNCH(ctx, "dwpw2", (dwpw2s_x3_kernel<<<grid, 256, 0, s>>>(b.xb, b.u, w.dw_w, w.dw_b, w16.pw2_w, w.pw2_b, seq, N,
                                                                   nsegs)));
    }
#else
    dim3 dgrid(N, ntl);
    if (XCD_ORDER) dgrid = dim3((unsigned)((((long)N * ntl + 7) / 8) * 8), 1);
    LAUNCH(ctx, "dwpw2", (dwpw2_x3_kernel<<<dgrid, 64 * DP_WAVES, 0, s>>>(b.xb, b.u, w.dw_w, w.dw_b, w16.pw2_w, w.pw2_b, seq,
                                                                          N, ntl)));
#endif
    if (taps) {
        hipMemcpyAsync(taps + (size_t)2 * M * 64, b.xb, tap_bytes, hipMemcpyDeviceToDevice, s);
        LAUNCH(ctx, "ffn", (ffn_x3_kernel<false, 2, FFN_WAVES><<<ffn_grid(flat_tiles), 64 * FFN_WAVES, 0, s>>>(
                               b.xb, taps + (size_t)3 * M * 64, nullptr, nullptr, w16.ff2_w1, w.ff2_b1, w16.ff2_w2,
                               w.ff2_b2, M, flat_tiles)));
    }
    LAUNCH(ctx, "ffn_post", (ffn_x3_kernel<true, 2, FFN_WAVES><<<ffn_grid(flat_tiles), 64 * FFN_WAVES, 0, s>>>(
                                b.xb, b.xa, outer_residual ? b.xa : nullptr, w.post_gb, w16.ff2_w1, w.ff2_b1,
                                w16.ff2_w2, w.ff2_b2, M, flat_tiles)));
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
