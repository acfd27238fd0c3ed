// conv_x3.hip - the implicit-GEMM convolutions of conv.hip on the f16 matrix pipe with
// 3-term split products (see common.hip.h, "x3" mode).  Same tiling idea (positions of the
// (t, f') plane flattened with one virtual zero column per row, halo-shared f taps, two
// time planes, normalise-on-load, InstanceNorm partials in the epilogue), re-balanced for a
// pipe that is ~5x faster:
//   * tile = 256 positions x all output channels, 4 waves, each wave 64 positions x COUT
//     (A fragments re-used across 4 position blocks -> LDS read traffic per MFMA halves)
//   * a stage = one time plane of one 32-channel chunk: activations are normalised, split
//     into fp16 hi/lo and written to LDS ONCE, then read by 3 taps x COUT/16 x 3 products
//   * stage s+1's global loads are issued into registers before stage s's MFMAs (T14-style
//     issue-early / write-late), so HBM/L2 latency hides under the matrix work.
#include "kernels.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CX_STRIDE 48          // halfs per LDS activation row (32 used): 96 B = 6 x 16 B slots, the row pitch at which
                              // the B-operand ds_read_b128 (16 consecutive rows x 4 lane groups) is bank-conflict free
                              // for every base row (80 B was 2-way; brute-forced over the b128 lane-group map)

__device__ __forceinline__ f32x4 norm_prelu4x(f32x4 v, f32x4 sc, f32x4 sh, f32x4 al) {
    f32x4 y = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = y[e] >= 0.f ? y[e] : al[e] * y[e];
    return y;
}

// kernarg pointer tables are indexed with compile-time constants only (a runtime index would
// make the compiler spill the whole ConvArgs struct to scratch)
__device__ __forceinline__ const float* sel4(const float* const (&p)[4], int i) {
    return i == 0 ? p[0] : (i == 1 ? p[1] : (i == 2 ? p[2] : p[3]));
}

template <int NT, int COUT, int NPB>
__global__ __launch_bounds__(256) void conv3x_kernel(ConvArgs a, const _Float16* __restrict__ w16) {
    constexpr int CX_TILE = 64 * NPB;                         // positions per block (4 waves x NPB x 16)
    constexpr int CX_ROWS = CX_TILE + 2;
    constexpr int CB = COUT / 16;
    constexpr int TAPS = NT * 3;
    constexpr int NACT = (CX_ROWS * 8 + 255) / 256;          // float4 loads per thread per stage
    constexpr int W16 = 3 * CB * 2 * 64;                      // 16-byte units of weights per stage
    constexpr int NW = W16 / 256;
    __shared__ __attribute__((aligned(16))) _Float16 act_h[CX_ROWS * CX_STRIDE];
    __shared__ __attribute__((aligned(16))) _Float16 act_l[CX_ROWS * CX_STRIDE];
    __shared__ __attribute__((aligned(16))) _Float16 wl[W16 * 8];
    __shared__ float red[4][COUT][2];

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4, wv = tid >> 6;
    // XCD-aware block order: the dispatcher deals consecutive workgroups round-robin to the 8
    // XCDs (private L2 each).  Re-map so XCD x walks a CONTIGUOUS range of (clip, tile): the
    // rows a tile reads as its t-dil plane were read moments earlier as the t plane of a
    // neighbouring tile on the SAME XCD -> second read is an L2 hit, not a second HBM fetch.
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int qn = nwg >> 3, rn = nwg & 7;
    const int logical = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    const int b = logical / a.ntiles, tile = logical - b * a.ntiles;
    const int Fp = a.F + 1;
    const int q0 = tile * CX_TILE;

    // per-thread staging map: row p = (tid >> 3) + 32 e, channel quad qd = tid & 7 of the 32-chunk
    const int qd = tid & 7;
    const int lds_col = 8 * (qd & 3) + 4 * (qd >> 2);       // chain slot order: [4g..4g+3 | 16+4g..]
    int apos1[NACT], apos0[NACT];
#pragma unroll
    for (int e = 0; e < NACT; ++e) {
        const int p = (tid >> 3) + 32 * e;
        apos1[e] = apos0[e] = -1;
        if (p < CX_ROWS) {
            const int q = q0 - 1 + p;
            if (q >= 0) {
                const int t = q / Fp, f = q - t * Fp;
                if (f < a.F && t < a.T) {
                    apos1[e] = (b * a.T + t) * a.F + f;
                    if (NT == 2 && t >= a.dil) apos0[e] = apos1[e] - a.dil * a.F;
                }
            }
        }
    }

    f32x4 acc[CB][NPB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = splat4(0.f);

    const int nst = a.nslots * 2 * NT;
    f32x4 pre[NACT];
    u32x4 wpre[NW];
    f32x4 sc = splat4(1.f), sh = splat4(0.f), al = splat4(1.f);
    bool tr = false;

#define CX_PREFETCH(S)                                                                                   \
    do {                                                                                                 \
        const int chunk_ = (S) / NT, kt_ = (S) - chunk_ * NT;                                            \
        const int slot_ = chunk_ >> 1, half_ = chunk_ & 1;                                               \
        const float* src_ = sel4(a.in, slot_) + half_ * 32 + qd * 4;                                                 \
        _Pragma("unroll") for (int e = 0; e < NACT; ++e) {                                               \
            const int ap_ = (NT == 2 && kt_ == 0) ? apos0[e] : apos1[e];                                 \
            pre[e] = ldg4(src_ + (long)(ap_ >= 0 ? ap_ : 0) * 64);   /* unconditional; masked at write */ \
        }                                                                                                \
        const float* nsc_ = sel4(a.nscale, slot_);                                                       \
        tr = nsc_ != nullptr;                                                                            \
        if (tr) {                                                                                        \
            sc = ldg4(nsc_ + b * 64 + half_ * 32 + qd * 4);                                              \
            sh = ldg4(sel4(a.nshift, slot_) + b * 64 + half_ * 32 + qd * 4);                             \
            al = ldg4(sel4(a.nalpha, slot_) + half_ * 32 + qd * 4);                                      \
        }                                                                                                \
        const u32x4* wsrc_ = reinterpret_cast<const u32x4*>(w16) + ((long)chunk_ * TAPS + kt_ * 3) * (CB * 128); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) wpre[i] = wsrc_[tid + 256 * i];                   \
    } while (0)

    CX_PREFETCH(0);
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
        const int kt = s % NT;
        __syncthreads();                                  // stage s-1 fully consumed
#pragma unroll
        for (int e = 0; e < NACT; ++e) {
            const int p = (tid >> 3) + 32 * e;
            if (p < CX_ROWS) {
                const int ap = (NT == 2 && kt == 0) ? apos0[e] : apos1[e];
                f32x4 v = pre[e];
                if (tr) v = norm_prelu4x(v, sc, sh, al);
                if (ap < 0) v = splat4(0.f);              // zero padding (select, no branch)
                f16x4 hi, lo;
                split4(v, hi, lo);
                *reinterpret_cast<f16x4*>(&act_h[p * CX_STRIDE + lds_col]) = hi;
                *reinterpret_cast<f16x4*>(&act_l[p * CX_STRIDE + lds_col]) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) reinterpret_cast<u32x4*>(wl)[tid + 256 * i] = wpre[i];
        __syncthreads();
        if (s + 1 < nst) CX_PREFETCH(s + 1);              // in flight during the MFMAs below

#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
            f16x8 bh[NPB], bl[NPB];
#pragma unroll
            for (int tb = 0; tb < NPB; ++tb) {
                const int row = 16 * NPB * wv + 16 * tb + c + kf;
                bh[tb] = *reinterpret_cast<const f16x8*>(&act_h[row * CX_STRIDE + 8 * g]);
                bl[tb] = *reinterpret_cast<const f16x8*>(&act_l[row * CX_STRIDE + 8 * g]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&wl[((kf * CB + cb) * 2 + 0) * 512 + lane * 8]);
                const f16x8 alo = *reinterpret_cast<const f16x8*>(&wl[((kf * CB + cb) * 2 + 1) * 512 + lane * 8]);
#pragma unroll
                for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32h(ah, bh[tb], acc[cb][tb]);
#pragma unroll
                for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32h(ah, bl[tb], acc[cb][tb]);
#pragma unroll
                for (int tb = 0; tb < NPB; ++tb) acc[cb][tb] = mfma32h(alo, bh[tb], acc[cb][tb]);
            }
        }
    }

    // ---- epilogue: bias, store, InstanceNorm partial sums (same contract as conv3_kernel) ----
    bool ok[NPB];
    long obase[NPB];
#pragma unroll
    for (int tb = 0; tb < NPB; ++tb) {
        const int q = q0 + 16 * NPB * wv + 16 * tb + c;
        const int t = q / Fp, f = q - t * Fp;
        ok[tb] = (t < a.T) && (f < a.F);
        if (a.mode == 1) {
            ok[tb] = ok[tb] && ((f & 1) == 0);
            const int F2 = (a.F + 1) >> 1;
            obase[tb] = ((long)(b * a.T + t) * F2 + (f >> 1)) * 64;
        } else if (a.mode == 2) {
            obase[tb] = ((long)(b * a.T + t) * (2 * a.F) + 2 * f) * 64;
        } else {
            obase[tb] = ((long)(b * a.T + t) * a.F + f) * 64;
        }
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const f32x4 bias = ldg4(a.bias + 16 * cb + 4 * g);
        f32x4 s1 = splat4(0.f), s2 = splat4(0.f);
#pragma unroll
        for (int tb = 0; tb < NPB; ++tb) {
            const f32x4 v = acc[cb][tb] + bias;
            if (ok[tb]) {
                long off = obase[tb] + 16 * (cb & 3) + 4 * g;
                if (a.mode == 2) off += (cb >> 2) * 64;
                stg4(a.out + off, v);
                s1 += v;
                s2 += v * v;
            }
        }
        if (a.partials) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t1 = red_c_sum(s1[r]), t2 = red_c_sum(s2[r]);
                if (c == 0) {
                    red[wv][16 * cb + 4 * g + r][0] = t1;
                    red[wv][16 * cb + 4 * g + r][1] = t2;
                }
            }
        }
    }
    if (a.partials) {
        __syncthreads();
        for (int i = tid; i < COUT * 2; i += 256) {
            const int co = i >> 1, wh = i & 1;
            const float t = (red[0][co][wh] + red[1][co][wh]) + (red[2][co][wh] + red[3][co][wh]);
            a.partials[(((long)b * a.ntiles + tile) * COUT + co) * 2 + wh] = t;
        }
    }
}

// the dense / 1x3 convs use 256-position tiles; the 128-channel sub-pixel conv 128-position tiles
int conv3x_ntiles(int T, int F, int cout) {
    const int tile = cout == 128 ? 128 : 256;
    return (T * (F + 1) + tile - 1) / tile;
}

void launch_conv3_x3(LaunchCtx ctx, const ConvArgs& a, const void* w16, int B, int time_taps, int cout) {
    dim3 grid(a.ntiles * B);                              // 1-D: see the XCD re-map in the kernel
    const _Float16* w = reinterpret_cast<const _Float16*>(w16);
    if (time_taps == 2 && cout == 64)
        LAUNCH(ctx, "conv_dense", (conv3x_kernel<2, 64, 4><<<grid, 256, 0, ctx.stream>>>(a, w)));
    else if (time_taps == 1 && cout == 64)
        LAUNCH(ctx, "conv_1x3", (conv3x_kernel<1, 64, 4><<<grid, 256, 0, ctx.stream>>>(a, w)));
    else
        LAUNCH(ctx, "conv_subpixel", (conv3x_kernel<1, 128, 2><<<grid, 256, 0, ctx.stream>>>(a, w)));
}
