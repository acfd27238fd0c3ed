// ffn32_x3.hip - the ConformerBlock's FeedForward branches (conformer.py:54-72, 136-148, 216-221:
// `Scale(0.5, PreNorm(FeedForward))`, for ff2 followed by post_norm and the TSCB residual) on v_mfma_f32_32x32x16_f16,
// three split-f16 products per contraction.  x3 / x1 modes.  Lane-level CPU model: tests/test_ffn32_tile_model.py.
//
// Why a second FeedForward kernel.  ffn_x3_kernel (conformer_x3.hip) uses the 4-pass 16x16x32 shape, whose issue rate
// leaves no slot for another instruction: its 384 MFMAs (6.1 k cycles per 32-token tile) and its 128 Swish per lane
// (two quarter-rate transcendentals each, 5 - 6 k cycles) ADD UP - 16.4 k cycles per tile and SIMD measured, 0.33 -
// 0.44 matrix-pipe busy.  The 8-pass 32x32x16 shape holds the pipe for 32 cycles per instruction and lets VALU work
// issue underneath, provided the instruction STREAM alternates them: the body below is written as 17 slots of four
// MFMA triples with the Swish / fp16-split VALU work of the neighbouring hidden tile placed between the MFMAs by hand
// (scheduling barriers keep the order).
//
// Layout.  A wave owns 32 tokens; lane (tok = lane & 31, hh = lane >> 5).  Every 64-vector of a token is kept in the
// PERMUTED channel order  channel(q, r, hh) = 8 q + 4 hh + r  (q = 0..7 float4s, r = element): the eight float4s a
// lane loads are exactly the channels its output accumulators hold (accumulator register v of output tile u <->
// channel(4 u + (v >> 2), v & 3, hh)), so residual + bias are the accumulators' initial value, stores are float4s, and
// nothing crosses lanes except the two LayerNorm sums (one v_permlane32_swap each).  The hidden tile t (32 units) leaves
// GEMM 1 as accumulator registers v <-> hidden 32 t + 8 (v >> 2) + 4 hh + (v & 3); registers 8 jp .. 8 jp + 7 ARE the
// B operand of k-step 2 t + jp of GEMM 2 - the operand images api.hip packs (x3_image_ffn32_*) order W1's columns
// and W2's columns to match:
//   W1 image [t 8][kk 4][hi | lo][64 lanes][8]: lane (row, hh) slot e = W1[32 t + row][channel(2 kk + (e >> 2), e & 3, hh)]
//   W2 image [u 2][ks 16][hi | lo][64 lanes][8]: lane (row, hh) slot e = W2[32 u + row][32 (ks >> 1) + 16 (ks & 1) + 8 (e >> 2) + 4 hh + (e & 3)]
// (the packer's folds - LayerNorm affine into W1 / b1, -log2 e into W1 / b1, -ln 2 / 2 into W2 - are in the fp32
// fragment-major blob the images are derived from).
#include "kernels.h"

namespace X3_NS {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 fmfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 fmfmal(f16x8 a, f16x8 b, f32x16 c) {        // a term with a lo operand
    return X3_TERMS == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0) : c;
}
#define F32_SB() __builtin_amdgcn_sched_barrier(0)

#ifndef FFN32_X0_EARLY
#define FFN32_X0_EARLY 0      // 1: the TSCB residual rows of the FINAL variant are requested in slot 16 (32 more live registers:
                              // spills 5 dwords at the 168-register budget of 12 waves)
#endif
#ifndef FFN32_X0_HALF
#define FFN32_X0_HALF 1       // FINAL: the u = 0 half of the TSCB residual rows is requested in slot 16 (16 registers; the whole
                              // row set spills at the 168-register budget): ffn_post 1.93 -> 1.87 ms
#endif
#ifndef FFN32_AHEAD
#define FFN32_AHEAD 1         // MFMA groups between an operand fragment's LDS read and its use (1 / 2 / 3 measured equal)
#endif
#ifndef FFN32_WAVES
#define FFN32_WAVES 12        // waves per block (one persistent block per CU: 128 KB of weight images)
#endif

template <bool FINAL, int TWAVES>
__global__ __launch_bounds__(TWAVES * 64) void ffn32_x3_kernel(const float* xin, float* xout, const float* x0,
                                                               const float* __restrict__ post_gb,
                                                               const _Float16* __restrict__ w1i, const float* __restrict__ b1,
                                                               const _Float16* __restrict__ w2i, const float* __restrict__ b2,
                                                               long M, int ntiles) {
    __shared__ __attribute__((aligned(16))) _Float16 wlds[65536];          // W1 image 64 KB | W2 image 64 KB
    __shared__ __attribute__((aligned(16))) float bias_l[448];             // b1[256] | b2[64] | post_norm gamma[64] | beta[64]
    stage_lds16<4096, TWAVES * 64>(w1i, wlds);
    stage_lds16<4096, TWAVES * 64>(w2i, wlds + 32768);
    for (int i = threadIdx.x; i < 320; i += blockDim.x) bias_l[i] = i < 256 ? b1[i] : b2[i - 256];
    if (FINAL)          // (through LDS: as global fetches the 16 gamma / beta float4s cost two more hoisted 64-bit lane pointers -> spills)
        for (int i = threadIdx.x; i < 128; i += blockDim.x) bias_l[320 + i] = post_gb[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, tok = lane & 31, hh = lane >> 5, wv = threadIdx.x >> 6;
    const _Float16* const w1 = wlds + lane * 8;                  // + (t * 4 + kk) * 1024 (+ 512: lo)
    const _Float16* const w2 = wlds + 32768 + lane * 8;          // + (u * 16 + ks) * 1024 (+ 512: lo)
    const float* const bl = bias_l + 4 * hh;

    // rows of the wave's first tile; every later tile's rows are requested inside the previous tile's last slots, into
    // the 32 registers the GEMM 1 operands (xh / xl) free there
    // element offset of this lane's first float4 in a token row of tile_ (clamped to the last token).  Opaque to the
    // optimiser on purpose: otherwise `base + 4 hh` is hoisted as a 64-bit per-lane pointer for every base (xin, xout,
    // x0: 6 VGPRs held across the whole loop - the FINAL instantiation spilled them at its 168-register budget)
    auto off_of = [&](int tile_) -> long {
        const long t0_ = (long)tile_ * 32 + tok;
        long o_ = (t0_ < M ? t0_ : M - 1) * 64 + 4 * hh;
        asm volatile("" : "+v"(o_));
        return o_;
    };
    f32x4 x[8];
    {
        const int tile0 = blockIdx.x * TWAVES + wv;
        const long o0 = off_of(tile0 < ntiles ? tile0 : 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = ldg4(xin + o0 + 8 * q);
    }
#pragma unroll 1
    for (int tile = blockIdx.x * TWAVES + wv; tile < ntiles; tile += gridDim.x * TWAVES) {
        // LayerNorm statistics: 32 channels in the lane, the other 32 in lane ^ 32
        float mean, rstd;
        {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += (x[q][0] + x[q][1]) + (x[q][2] + x[q][3]);
            float o;
            const float a = xchg32(s, o);
            mean = (a + o) * (1.0f / 64.0f);
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = x[q][r] - mean;
                    v = fmaf(d, d, v);
                }
            const float a2 = xchg32(v, o);
            rstd = rsqrtf((a2 + o) * (1.0f / 64.0f) + CMGAN_EPS);
        }
        f16x8 xh[4], xl[4];                                      // B operands of GEMM 1: k-step kk <-> float4s 2 kk, 2 kk + 1
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            split8((x[2 * kk] - splat4(mean)) * splat4(rstd), (x[2 * kk + 1] - splat4(mean)) * splat4(rstd), xh[kk], xl[kk]);
        // output accumulators start from residual + second bias
        f32x16 y[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 256 + 32 * u + 8 * j);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[u][4 * j + r] = x[4 * u + j][r] + b[r];
            }

        f32x16 h[2];                                             // hidden tiles t (even / odd)
        auto h_init = [&](int t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * t + 8 * j);
#pragma unroll
                for (int r = 0; r < 4; ++r) h[t & 1][4 * j + r] = b[r];
            }
        };
        // GEMM 1 of hidden tile 0 (nothing to overlap with yet)
        h_init(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(w1 + kk * 1024);
            const f16x8 al = *reinterpret_cast<const f16x8*>(w1 + kk * 1024 + 512);
            h[0] = fmfma(ah, xh[kk], h[0]);
            h[0] = fmfmal(ah, xl[kk], h[0]);
            h[0] = fmfmal(al, xh[kk], h[0]);
        }

        // Slots s = 0 .. 16.  Slot s = 2 t + jp:
        //   VALU : fp16 split of slot s - 1's activations, Swish of hidden registers 8 jp .. 8 jp + 7 of tile t
        //   MFMA : GEMM 1 of tile t + 1, k-steps 2 jp and 2 jp + 1 (two triples), GEMM 2 of slot s - 1 (two triples: u = 0, 1)
        // laid out as four groups { triple | VALU chunk } so that VALU issues while the matrix pipe is busy.
        f32x4 xe[4];                                             // (FFN32_X0_HALF) early half of the residual rows
        float av[2][8];                                          // activations of the current / previous slot
        f16x8 ph, pl;
        // Operand fragments (hi, lo) of MFMA group i = 4 s + g are read from LDS FFN32_AHEAD groups ahead of their triple
        // into a ring of FFN32_AHEAD + 1 register pairs (twelve waves share the CU's LDS: one group of lead was not enough -
        // half of a wave's cycles sat in s_waitcnt lgkmcnt)
        auto gvalid = [&](int i) -> bool {
            const int s_ = i >> 2, g_ = i & 3;
            return s_ <= 16 && (g_ < 2 ? (s_ >> 1) + 1 < 8 : s_ >= 1);
        };
        auto gptr = [&](int i) -> const _Float16* {
            const int s_ = i >> 2, g_ = i & 3;
            return g_ < 2 ? w1 + (((s_ >> 1) + 1) * 4 + 2 * (s_ & 1) + g_) * 1024 : w2 + ((g_ - 2) * 16 + s_ - 1) * 1024;
        };
        constexpr int RING = FFN32_AHEAD + 1;
        f16x8 fh[RING], fl[RING];
#pragma unroll
        for (int i = 0; i < FFN32_AHEAD; ++i)
            if (gvalid(i)) {
                fh[i % RING] = *reinterpret_cast<const f16x8*>(gptr(i));
                fl[i % RING] = *reinterpret_cast<const f16x8*>(gptr(i) + 512);
            }
#define FFN32_FETCH(I)                                                                        \
    if (gvalid((I) + FFN32_AHEAD)) {                                                          \
        fh[((I) + FFN32_AHEAD) % RING] = *reinterpret_cast<const f16x8*>(gptr((I) + FFN32_AHEAD));       \
        fl[((I) + FFN32_AHEAD) % RING] = *reinterpret_cast<const f16x8*>(gptr((I) + FFN32_AHEAD) + 512); \
    }
#pragma unroll
        for (int s = 0; s <= 16; ++s) {
            const int t = s >> 1, jp = s & 1, cur = s & 1, prv = cur ^ 1;
            const bool g1 = t + 1 < 8;                            // GEMM 1 triples of tile t + 1 exist
            const bool g2 = s >= 1;                               // GEMM 2 triples of slot s - 1 exist
            const bool sw = s < 16;                               // this slot has activations to compute
            if (g1 && jp == 0) h_init(t + 1);
#if FFN32_X0_EARLY
            if (FINAL && s == 16 && x0) {                         // TSCB residual rows: under the last GEMM 2 triples
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = ldg4(x0 + off_of(tile) + 8 * q);      // (recomputed: two registers less across the slots)
            }
#endif
#if FFN32_X0_HALF
            if (FINAL && s == 16 && x0) {                         // the u = 0 half of the TSCB residual rows: under the last GEMM 2 triples
                const long oh = off_of(tile);
#pragma unroll
                for (int j = 0; j < 4; ++j) xe[j] = ldg4(x0 + oh + 8 * j);
            }
#endif
            if (s == 14) {                                        // xh / xl are dead from here on: 32 registers for the rows needed next
                if (!(FINAL && FFN32_X0_EARLY)) {
                    const int nt = tile + gridDim.x * TWAVES;
                    const long on = off_of(nt < ntiles ? nt : tile);
#pragma unroll
                    for (int q = 0; q < 8; ++q) x[q] = ldg4(xin + on + 8 * q);
                }
            }
            f32x16& hn = h[(t + 1) & 1];
            const f32x16& hc = h[t & 1];
            f16x2 sh[4], sl[4];
            float e[8], d[8];
            const int i0 = 4 * s;
            // ---- group 0: GEMM 1 triple kk = 2 jp | split pairs 0, 1 of slot s - 1, Swish of values 0, 1
            FFN32_FETCH(i0)
            F32_SB();
            if (g1) hn = fmfma(fh[i0 % RING], xh[2 * jp], hn);
            if (g2) split2(av[prv][0], av[prv][1], sh[0], sl[0]);
            F32_SB();
            if (g1) hn = fmfmal(fh[i0 % RING], xl[2 * jp], hn);
            if (g2) split2(av[prv][2], av[prv][3], sh[1], sl[1]);
            if (sw) { e[0] = __builtin_amdgcn_exp2f(hc[8 * jp + 0]); e[1] = __builtin_amdgcn_exp2f(hc[8 * jp + 1]); }
            F32_SB();
            if (g1) hn = fmfmal(fl[i0 % RING], xh[2 * jp], hn);
            if (sw) {
                d[0] = __builtin_amdgcn_rcpf(1.0f + e[0]); d[1] = __builtin_amdgcn_rcpf(1.0f + e[1]);
                av[cur][0] = hc[8 * jp + 0] * d[0]; av[cur][1] = hc[8 * jp + 1] * d[1];
            }
            // ---- group 1: GEMM 1 triple kk = 2 jp + 1 | split pairs 2, 3, Swish of values 2, 3
            FFN32_FETCH(i0 + 1)
            F32_SB();
            if (g1) hn = fmfma(fh[(i0 + 1) % RING], xh[2 * jp + 1], hn);
            if (g2) split2(av[prv][4], av[prv][5], sh[2], sl[2]);
            F32_SB();
            if (g1) hn = fmfmal(fh[(i0 + 1) % RING], xl[2 * jp + 1], hn);
            if (g2) split2(av[prv][6], av[prv][7], sh[3], sl[3]);
            if (sw) { e[2] = __builtin_amdgcn_exp2f(hc[8 * jp + 2]); e[3] = __builtin_amdgcn_exp2f(hc[8 * jp + 3]); }
            F32_SB();
            if (g1) hn = fmfmal(fl[(i0 + 1) % RING], xh[2 * jp + 1], hn);
            if (sw) {
                d[2] = __builtin_amdgcn_rcpf(1.0f + e[2]); d[3] = __builtin_amdgcn_rcpf(1.0f + e[3]);
                av[cur][2] = hc[8 * jp + 2] * d[2]; av[cur][3] = hc[8 * jp + 3] * d[3];
            }
            if (g2) {
                ph = __builtin_bit_cast(f16x8, (u32x4){__builtin_bit_cast(unsigned, sh[0]), __builtin_bit_cast(unsigned, sh[1]),
                                                        __builtin_bit_cast(unsigned, sh[2]), __builtin_bit_cast(unsigned, sh[3])});
                pl = __builtin_bit_cast(f16x8, (u32x4){__builtin_bit_cast(unsigned, sl[0]), __builtin_bit_cast(unsigned, sl[1]),
                                                        __builtin_bit_cast(unsigned, sl[2]), __builtin_bit_cast(unsigned, sl[3])});
            }
            // ---- group 2: GEMM 2 triple u = 0 of slot s - 1 | Swish of values 4, 5
            FFN32_FETCH(i0 + 2)
            F32_SB();
            if (g2) y[0] = fmfma(fh[(i0 + 2) % RING], ph, y[0]);
            if (sw) { e[4] = __builtin_amdgcn_exp2f(hc[8 * jp + 4]); e[5] = __builtin_amdgcn_exp2f(hc[8 * jp + 5]); }
            F32_SB();
            if (g2) y[0] = fmfmal(fh[(i0 + 2) % RING], pl, y[0]);
            if (sw) { d[4] = __builtin_amdgcn_rcpf(1.0f + e[4]); d[5] = __builtin_amdgcn_rcpf(1.0f + e[5]); }
            F32_SB();
            if (g2) y[0] = fmfmal(fl[(i0 + 2) % RING], ph, y[0]);
            if (sw) { av[cur][4] = hc[8 * jp + 4] * d[4]; av[cur][5] = hc[8 * jp + 5] * d[5]; }
            // ---- group 3: GEMM 2 triple u = 1 | Swish of values 6, 7
            FFN32_FETCH(i0 + 3)
            F32_SB();
            if (g2) y[1] = fmfma(fh[(i0 + 3) % RING], ph, y[1]);
            if (sw) { e[6] = __builtin_amdgcn_exp2f(hc[8 * jp + 6]); e[7] = __builtin_amdgcn_exp2f(hc[8 * jp + 7]); }
            F32_SB();
            if (g2) y[1] = fmfmal(fh[(i0 + 3) % RING], pl, y[1]);
            if (sw) { d[6] = __builtin_amdgcn_rcpf(1.0f + e[6]); d[7] = __builtin_amdgcn_rcpf(1.0f + e[7]); }
            F32_SB();
            if (g2) y[1] = fmfmal(fl[(i0 + 3) % RING], ph, y[1]);
            if (sw) { av[cur][6] = hc[8 * jp + 6] * d[6]; av[cur][7] = hc[8 * jp + 7] * d[7]; }
            F32_SB();
        }
#undef FFN32_FETCH

        // ---- epilogue: (post LayerNorm + TSCB residual), store ----
        if (FINAL) {
            const long oe = off_of(tile);
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += y[u][v];
            float o;
            const float a = xchg32(s, o);
            const float pm = (a + o) * (1.0f / 64.0f);
            float vv = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float dd = y[u][v] - pm;
                    vv = fmaf(dd, dd, vv);
                }
            const float a2 = xchg32(vv, o);
            const float pr = rsqrtf((a2 + o) * (1.0f / 64.0f) + CMGAN_EPS);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(bl + 320 + 32 * u + 8 * j);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(bl + 384 + 32 * u + 8 * j);
                    f32x4 r0 = {y[u][4 * j], y[u][4 * j + 1], y[u][4 * j + 2], y[u][4 * j + 3]};
                    r0 = (r0 - splat4(pm)) * splat4(pr) * gm + bt;
#if FFN32_X0_EARLY
                    if (x0) r0 += x[4 * u + j];                  // TSCB residual rows: requested in slot 16
#elif FFN32_X0_HALF
                    if (x0) r0 += u == 0 ? xe[j] : ldg4(x0 + oe + 32 * u + 8 * j);
#else
                    if (x0) r0 += ldg4(x0 + oe + 32 * u + 8 * j);
#endif
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[u][4 * j + r] = r0[r];
                }
        }
        if ((long)tile * 32 + tok < M) {
            const long oo = off_of(tile);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 r0 = {y[u][4 * j], y[u][4 * j + 1], y[u][4 * j + 2], y[u][4 * j + 3]};
                    stg4(xout + oo + 32 * u + 8 * j, r0);
                }
        }
    }
}

}  // namespace X3_NS
using namespace X3_NS;

static int ffn32_grid(int ntiles, int waves) {                  // one persistent block per CU
    const int want = (ntiles + waves - 1) / waves;
    return want < 256 ? (want > 0 ? want : 1) : 256;
}

void launch_ffn32_x3(LaunchCtx ctx, bool final_, const float* xin, float* xout, const float* x0, const float* post_gb,
                     const _Float16* w1i, const float* b1, const _Float16* w2i, const float* b2, long M) {
    const int ntiles = (int)((M + 31) / 32);
    const int grid = ffn32_grid(ntiles, FFN32_WAVES);
    if (final_)
        LAUNCH(ctx, "ffn_post", (ffn32_x3_kernel<true, FFN32_WAVES><<<grid, 64 * FFN32_WAVES, 0, ctx.stream>>>(
                                    xin, xout, x0, post_gb, w1i, b1, w2i, b2, M, ntiles)));
    else
        LAUNCH(ctx, "ffn", (ffn32_x3_kernel<false, FFN32_WAVES><<<grid, 64 * FFN32_WAVES, 0, ctx.stream>>>(
                               xin, xout, x0, post_gb, w1i, b1, w2i, b2, M, ntiles)));
}
