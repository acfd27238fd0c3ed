// stft.hip - waveform <-> power-compressed spectrogram front/back end (reference:
// src/evaluation.py:21-23,36-39,41-51 and src/utils.py:20-39; torch.stft/istft with
// n_fft=400, hop=100, periodic Hamming window, center=True/reflect, onesided).
//
// n_fft = 400 = 2^4 * 5^2 is not a power of two; at these sizes the transform is a tiny
// fraction of the path, so it is evaluated as a dense (windowed) DFT-matrix product on the
// fp32 MFMA pipe with the power-law compression fused into the epilogue: one kernel reads
// the waveform once and writes the model input [B,2,T,F] once.  The inverse is the
// mirror image (uncompress fused into the operand staging, synthesis window folded into
// the inverse DFT matrix) followed by a small overlap-add / envelope / un-scale kernel.
#include "kernels.h"
#include <stdlib.h>

// m2^p for the power-law (de)compression: 2^(p log2 m2) on the transcendental unit (v_log_f32 / v_exp_f32,
// relative error ~1e-7 |log2 m2|) instead of libm powf (~60 instructions; 52 of them per lane used to be most
// of the STFT kernel).  v_log_f32 flushes denormal inputs, so the argument is pre-scaled by 2^60 (exact) and
// the 60 subtracted again: branch-free and correct down to the smallest denormal; m2 = 0 -> 0.
__device__ __forceinline__ float pow_pos(float m2, float p) {
    const float l2 = __builtin_amdgcn_logf(m2 * 1152921504606846976.0f) - 60.0f;
    const float r = __builtin_amdgcn_exp2f(p * l2);
    return m2 > 0.f ? r : 0.f;
}

// ---------------------------------------------------------------------------------
// c[b] = sqrt(L / sum x^2)                    evaluation.py:21, train.py:75-79
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rms_scale_kernel(const float* __restrict__ wav, int L,
                                                         float* __restrict__ scale) {
    __shared__ double red[1024];
    const float* x = wav + (long)blockIdx.x * L;
    double s = 0.0;
    for (int i = threadIdx.x; i < L; i += 1024) s += (double)x[i] * (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) scale[blockIdx.x] = (float)sqrt((double)L / red[0]);
}

void launch_rms_scale(LaunchCtx ctx, const float* wav, int B, int L, float* scale) {
    LAUNCH(ctx, "rms_scale", (rms_scale_kernel<<<B, 1024, 0, ctx.stream>>>(wav, L, scale)));
}

// ---------------------------------------------------------------------------------
// STFT + power compression.  block = (16 frames of one clip), 4 waves; the 16 frames'
// samples (15*hop + n_fft, reflect-padded, times the RMS scale) are staged in LDS once.
// Orientation out[frame][bin]: A = frames (LDS), B = windowed DFT matrix (fragment-major,
// global/L2), so for a fixed accumulator register 16 lanes hold 16 consecutive bins of
// one frame and the store is contiguous.  Each wave owns bin blocks wv, wv+4, ... and
// accumulates the re and im rows of a bin block together, so the compression
// X * (re^2+im^2)^-0.35  (== mag^0.3 * (cos, sin)(phase), utils.py:20-29) is lane-local.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_compress_kernel(SpectralTables tb, const float* __restrict__ wav,
                                                            const float* __restrict__ scale, int L, int T,
                                                            float* __restrict__ spec) {
    extern __shared__ __attribute__((aligned(16))) float seg[];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, fb = blockIdx.x;
    const int seglen = 15 * tb.hop + tb.n_fft;
    const int start = fb * 16 * tb.hop - tb.n_fft / 2;
    const float sc = scale ? scale[b] : 1.0f;
    const float* x = wav + (long)b * L;
    for (int i0 = threadIdx.x; i0 < seglen; i0 += 256 * 8) {      // 8 independent loads in flight per thread
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int s = start + i0 + 256 * k;
            if (s < 0) s = -s;
            if (s >= L) s = 2 * (L - 1) - s;
            s = s < 0 ? 0 : (s >= L ? L - 1 : s);  // frames past T in the last block: any finite value
            v[k] = x[s];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + 256 * k < seglen) seg[i0 + 256 * k] = v[k] * sc;
    }
    __syncthreads();
    const int KB = tb.n_fft / 16;
    const long P = (long)T * tb.F;
    for (int bb = wv; bb < tb.FB; bb += 4) {
        f32x4 are = splat4(0.f), aim = splat4(0.f);
        const float* wre = tb.fwd_fm + (long)bb * KB * 256 + lane * 4;
        const float* wim = tb.fwd_fm + (long)(tb.FB + bb) * KB * 256 + lane * 4;
#pragma unroll 5
        for (int kb = 0; kb < KB; ++kb) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&seg[c * tb.hop + 16 * kb + 4 * g]);
            const f32x4 br = ldg4(wre + kb * 256), bi = ldg4(wim + kb * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                are = mfma16(a[r], br[r], are);
                aim = mfma16(a[r], bi[r], aim);
            }
        }
        const int bin = bb * 16 + c;
        if (bin < tb.F) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = fb * 16 + 4 * g + r;
                if (t < T) {
                    const float m2 = are[r] * are[r] + aim[r] * aim[r];
                    const float s = pow_pos(m2, -0.35f);
                    spec[((long)b * 2 + 0) * P + (long)t * tb.F + bin] = are[r] * s;
                    spec[((long)b * 2 + 1) * P + (long)t * tb.F + bin] = aim[r] * s;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// F16X3 mode, n_fft 400 / hop 100: the same STFT + compression as above with
//   * the real DFT FOLDED about n = N/2: with u_c[n] = x[n] + x[N-n], u_s[n] = x[n] - x[N-n] (n = 1..N/2-1;
//     u_c[0] = x[0], u_c[N/2] = x[N/2]) and the symmetric Hamming window moved into the matrices,
//     re X[k] = sum_{n <= N/2} (w[n] cos(2 pi k n / N)) u_c[n],   im X[k] = sum (-w[n] sin(..)) u_s[n]:
//     contraction 201 -> 7 k32 blocks instead of 400, so half the matrix bytes and half the MFMAs;
//   * split-f16 products on the 16x16x32 pipe (3 MFMAs per k32 block instead of 8 fp32 ones);
//   * a block = 64 frames of one clip (4 waves x 16 frames): the folded operands of a wave's frames
//     stay in registers (A fragments), the 364 KB matrix image streams through a double-buffered
//     28 KB LDS chunk per 16-bin block, shared by the four waves.
// Per launch at B = 32: 6 x 32 blocks, 546 MFMAs per wave, 87 MB of matrix reads from L2.
// ---------------------------------------------------------------------------------
template <int NFFT, int HOP, int BSPLIT>
__global__ __launch_bounds__(256, 2) void stft_fold_x3_kernel(SpectralTables tb, const float* __restrict__ wav,
                                                              const float* __restrict__ scale, int L, int T,
                                                              float* __restrict__ spec) {
    constexpr int H = NFFT / 2, M32 = (H + 1 + 31) / 32, SEG = 63 * HOP + NFFT;
    constexpr int CHUNK = 2 * M32 * 1024;                    // halfs per bin block: [cos | -sin][M32][hi | lo][64][8]
    constexpr int NLD = CHUNK / 8 / 256;                     // 16-byte loads per thread per chunk
    static_assert(CHUNK % (8 * 256) == 0, "chunk must split evenly over the block");
    // LDS: 2 x 28 KB = two blocks per CU.  The staged waveform segment is only needed until the folded operands
    // are in registers, so it shares its bytes with the second matrix buffer (first written after bin block 0).
    constexpr int RAWB = (SEG + 4) * 4 > CHUNK * 2 ? (SEG + 4) * 4 : CHUNK * 2;
    __shared__ __attribute__((aligned(16))) _Float16 chunk0[CHUNK];
    __shared__ __attribute__((aligned(16))) unsigned char raw[RAWB];
    float* const seg = reinterpret_cast<float*>(raw);
    _Float16* const chunk1 = reinterpret_cast<_Float16*>(raw);
    __shared__ float wmax[4];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, fb = blockIdx.x;               // 64-frame tile of clip b
    const int start = fb * 64 * HOP - NFFT / 2;
    const float sc = scale ? scale[b] : 1.0f;
    const float* x = wav + (long)b * L;
    const u32x4* img = reinterpret_cast<const u32x4*>(tb.fold_fwd16);
    constexpr int FBC = (NFFT / 2 + 1 + 15) / 16;           // bin blocks
    // Small batches leave most CUs idle and the 13-step chain of one block IS the kernel time: blockIdx.z then
    // splits the bin blocks BSPLIT ways (every bin is computed by exactly the same instructions either way).
    constexpr int NBPER = (FBC + BSPLIT - 1) / BSPLIT;
    const int bb0 = BSPLIT > 1 ? (int)blockIdx.z * NBPER : 0;
    auto chunk_of = [&](int i) { const int bb = bb0 + i; return bb < FBC ? bb : FBC - 1; };   // clamped: any valid chunk
    // matrix chunks are prefetched DEPTH bin blocks ahead into registers (an L2 round trip is ~3 chunks of MFMA work)
    constexpr int DEPTH = 3;
    u32x4 pre[DEPTH][NLD];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < NLD; ++i) pre[d][i] = img[(long)chunk_of(d) * (CHUNK / 8) + threadIdx.x + 256 * i];
    {
        constexpr int NSEG = (SEG + 255) / 256;             // every segment load of the thread in flight at once
        float v[NSEG];
#pragma unroll
        for (int k = 0; k < NSEG; ++k) {
            int sidx = start + threadIdx.x + 256 * k;
            if (sidx < 0) sidx = -sidx;
            if (sidx >= L) sidx = 2 * (L - 1) - sidx;
            sidx = sidx < 0 ? 0 : (sidx >= L ? L - 1 : sidx);   // frames past T in the last tile: any finite value
            v[k] = x[sidx];
        }
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < NSEG; ++k)
            if (threadIdx.x + 256 * k < SEG) {
                seg[threadIdx.x + 256 * k] = v[k] * sc;
                amax = fmaxf(amax, fabsf(v[k] * sc));
            }
        amax = red_g_max(amax);                               // over the 4 lane groups ...
        amax = fmaxf(amax, dpp_perm<0xB1>(amax));            // ... and the 16 lanes of a row
        amax = fmaxf(amax, dpp_perm<0x4E>(amax));
        amax = fmaxf(amax, dpp_perm<0x141>(amax));
        amax = fmaxf(amax, dpp_perm<0x140>(amax));
        if (lane == 0) wmax[wv] = amax;
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) reinterpret_cast<u32x4*>(chunk0)[threadIdx.x + 256 * i] = pre[0][i];
    __syncthreads();
    // fp16 hi/lo operands only cover ~2^-24 .. 2^16 around 1: bring the tile's samples to [0.5, 1) with an exact
    // power-of-two factor (and undo it on the fp32 result), so the transform works at any input amplitude
    const float tile_max = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    int ex = (int)((__float_as_uint(tile_max) >> 23) & 0xff);
    ex = ex < 1 ? 1 : (ex > 252 ? 252 : ex);
    const float up = __uint_as_float((unsigned)(253 - ex) << 23);       // 2^(126 - ex)
    const float down = __uint_as_float((unsigned)(ex + 1) << 23);        // 2^(ex - 126)

    // folded operands of this wave's 16 frames: lane (frame c, group g), k32 block m, slot e <-> n = 32 m + 8 g + e
    f16x8 uch[M32], ucl[M32], ush[M32], usl[M32];
    const float* fr = seg + (wv * 16 + c) * HOP;
#pragma unroll
    for (int m = 0; m < M32; ++m) {
        const int n0 = 32 * m + 8 * g;
        f32x4 uc[2], us[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = n0 + e;
            // unconditional (clamped) LDS reads + selects: a per-lane `cond ? lds[..] : 0` becomes a branch per element
            const int na = n <= H ? n : H, nb = (n >= 1 && n < H) ? NFFT - n : H;
            float a = fr[na], bm = fr[nb];
            if (32 * m + 31 > H) a = n <= H ? a : 0.f;                      // only the last k32 block crosses N/2
            const bool pair = (32 * m > 0 || n >= 1) && (32 * m + 31 < H || n < H);
            bm = pair ? bm : 0.f;
            uc[e >> 2][e & 3] = (a + bm) * up;
            us[e >> 2][e & 3] = pair ? (a - bm) * up : 0.f;
        }
        split8(uc[0], uc[1], uch[m], ucl[m]);
        split8(us[0], us[1], ush[m], usl[m]);
    }

    __syncthreads();                                          // every wave has folded: the segment's bytes become chunk1

    const long P = (long)T * tb.F;
#pragma unroll
    for (int i = 0; i < NBPER; ++i) {                        // fully unrolled: the prefetch ring is statically indexed
        const int bb = bb0 + i;
        if (i >= 1 && i + DEPTH - 1 < NBPER) {               // slot of chunk i-1 is free again: fetch chunk i+DEPTH-1
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                pre[(i + DEPTH - 1) % DEPTH][k] = img[(long)chunk_of(i + DEPTH - 1) * (CHUNK / 8) + threadIdx.x + 256 * k];
        }
        if (BSPLIT == 1 || bb < FBC) {                        // block-uniform
            const _Float16* cb = ((i & 1) ? chunk1 : chunk0) + lane * 8;
            f32x4 are = splat4(0.f), aim = splat4(0.f);
#pragma unroll
            for (int m = 0; m < M32; ++m) {
                const f16x8 ch = *reinterpret_cast<const f16x8*>(cb + m * 1024);
                const f16x8 cl = *reinterpret_cast<const f16x8*>(cb + m * 1024 + 512);
                const f16x8 sh = *reinterpret_cast<const f16x8*>(cb + (M32 + m) * 1024);
                const f16x8 sl = *reinterpret_cast<const f16x8*>(cb + (M32 + m) * 1024 + 512);
                are = mfma32h(uch[m], ch, are);
                aim = mfma32h(ush[m], sh, aim);
                are = mfma32l(uch[m], cl, are);
                aim = mfma32l(ush[m], sl, aim);
                are = mfma32l(ucl[m], ch, are);
                aim = mfma32l(usl[m], sh, aim);
            }
            const int bin = bb * 16 + c;
            if (bin < tb.F) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = fb * 64 + wv * 16 + 4 * g + r;
                    if (t < T) {
                        const float xr = are[r] * down, xi = aim[r] * down;
                        const float m2 = xr * xr + xi * xi;
                        const float s = pow_pos(m2, -0.35f);
                        spec[((long)b * 2 + 0) * P + (long)t * tb.F + bin] = xr * s;
                        spec[((long)b * 2 + 1) * P + (long)t * tb.F + bin] = xi * s;
                    }
                }
            }
        }
        if (i + 1 < NBPER) {
#pragma unroll
            for (int k = 0; k < NLD; ++k)
                reinterpret_cast<u32x4*>(((i + 1) & 1) ? chunk1 : chunk0)[threadIdx.x + 256 * k] = pre[(i + 1) % DEPTH][k];
        }
        __syncthreads();
    }
}

#ifndef STFT_BSPLIT
#define STFT_BSPLIT 3
#endif
#ifndef STFT_FFT
#define STFT_FFT 1             // 1: n_fft 400 / hop 100 runs the real-FFT kernels (stft_fft400_kernel); 0: the folded split-f16 DFT products
#endif
void launch_stft_compress(LaunchCtx ctx, const SpectralTables& tb, const float* wav, const float* scale, int B,
                          int L, int T, float* spec) {
    static const int k_fft = env_knob("CMGAN_STFT_FFT", STFT_FFT, 0, 1);
    if (k_fft && tb.fold_fwd16 && tb.n_fft == 400 && tb.hop == 100 && tb.F == 201) {
        launch_stft_fft400(ctx, wav, scale, tb.window, B, L, T, spec);
        return;
    }
    if (tb.fold_fwd16 && tb.n_fft == 400 && tb.hop == 100) {
        const int tiles = (T + 63) / 64;
        if ((long)tiles * B >= 512) {                         // two blocks per CU already cover the chip
            dim3 grid64(tiles, B);
            LAUNCH(ctx, "stft_compress",
                   (stft_fold_x3_kernel<400, 100, 1><<<grid64, 256, 0, ctx.stream>>>(tb, wav, scale, L, T, spec)));
        } else {
            // bin blocks per thread block: the fewer, the shorter the dependent chain of one block (CMGAN_STFT_BSPLIT:
            // same-session sweep knob; every bin is computed by the same instructions whatever the split)
            static const int k_split = env_knob("CMGAN_STFT_BSPLIT", STFT_BSPLIT, 1, 13);
#define STFT_LAUNCH(S)                                                                                             \
    do {                                                                                                           \
        dim3 grid64(tiles, B, S);                                                                                  \
        LAUNCH(ctx, "stft_compress",                                                                               \
               (stft_fold_x3_kernel<400, 100, S><<<grid64, 256, 0, ctx.stream>>>(tb, wav, scale, L, T, spec)));    \
    } while (0)
            if (k_split >= 13) STFT_LAUNCH(13);
            else if (k_split >= 7) STFT_LAUNCH(7);
            else STFT_LAUNCH(3);
#undef STFT_LAUNCH
        }
        return;
    }
    dim3 grid((T + 15) / 16, B);
    const size_t shm = (size_t)(15 * tb.hop + tb.n_fft) * sizeof(float);
    LAUNCH(ctx, "stft_compress",
           (stft_compress_kernel<<<grid, 256, shm, ctx.stream>>>(tb, wav, scale, L, T, spec)));
}

// ---------------------------------------------------------------------------------
// power uncompress + inverse real DFT + synthesis window, per 16 frames.
// est[B,1,T,F] (re, im) -> frames[B,T,n_fft].  The uncompressed spectrum
// Y * (re^2+im^2)^(7/6)  (== mag^(1/0.3) with phase kept, utils.py:32-39) of 16 frames is
// staged in LDS as [16][2*FB*16] (re bins | im bins, zero padded); the inverse matrix has
// irfft's Hermitian weights (1, 2, ..., 2, 1; imag of DC/Nyquist ignored), 1/N and the
// Hamming window folded in (evaluation.py:44-50).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void uncompress_irfft_kernel(SpectralTables tb, const float* __restrict__ re,
                                                               const float* __restrict__ im, int T,
                                                               float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float ysp[];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, fb = blockIdx.x;
    const int K = 2 * tb.FB * 16;              // padded contraction length
    const int ld = K + 4;                      // row pitch (floats), keeps 16 B alignment
    const long P = (long)T * tb.F;
    const int nstage = 16 * tb.FB * 16;
    for (int i0 = threadIdx.x; i0 < nstage; i0 += 256 * 4) {       // 4 independent (re, im) loads in flight per thread
        float ar[4], ai[4];
        bool okv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            const int fr = i / (tb.FB * 16), bin = i - fr * (tb.FB * 16);
            const int t = fb * 16 + fr;
            okv[k] = i < nstage && t < T && bin < tb.F;
            const long off = (long)b * P + (long)(okv[k] ? t : 0) * tb.F + (okv[k] ? bin : 0);
            ar[k] = re[off];
            ai[k] = im[off];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i < nstage) {
                const int fr = i / (tb.FB * 16), bin = i - fr * (tb.FB * 16);
                float yr = 0.f, yi = 0.f;
                if (okv[k]) {
                    const float m2 = ar[k] * ar[k] + ai[k] * ai[k];
                    const float sc = pow_pos(m2, 7.0f / 6.0f);
                    yr = ar[k] * sc;
                    yi = ai[k] * sc;
                }
                ysp[fr * ld + bin] = yr;
                ysp[fr * ld + tb.FB * 16 + bin] = yi;
            }
        }
    }
    __syncthreads();
    const int KB = K / 16;
    const int NB = tb.n_fft / 16;
    for (int nb = wv; nb < NB; nb += 4) {
        f32x4 acc0 = splat4(0.f), acc1 = splat4(0.f);
        const float* wp = tb.inv_fm + (long)nb * KB * 256 + lane * 4;
        for (int kb = 0; kb < KB; kb += 2) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&ysp[c * ld + 16 * kb + 4 * g]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&ysp[c * ld + 16 * (kb + 1) + 4 * g]);
            const f32x4 b0 = ldg4(wp + kb * 256), b1 = ldg4(wp + (kb + 1) * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0 = mfma16(a0[r], b0[r], acc0);
                acc1 = mfma16(a1[r], b1[r], acc1);
            }
        }
        const int n = nb * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = fb * 16 + 4 * g + r;
            if (t < T) frames[((long)b * T + t) * tb.n_fft + n] = acc0[r] + acc1[r];
        }
    }
}

// overlap-add, window-envelope division, centre trim, '/ c'   (torch.istft; evaluation.py:51)
__global__ __launch_bounds__(256) void ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                  const float* __restrict__ scale, int n_fft, int hop, int T,
                                                  int Lout, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= Lout) return;
    const int p = s + n_fft / 2;
    int t1 = p / hop;
    if (t1 > T - 1) t1 = T - 1;
    int t0 = (p - n_fft + hop) / hop;          // ceil((p - n_fft + 1) / hop) for p >= n_fft - 1
    if (p - n_fft + 1 <= 0) t0 = 0;
    float acc = 0.f, env = 0.f;
    for (int t = t0; t <= t1; ++t) {
        const int n = p - t * hop;
        if (n >= 0 && n < n_fft) {
            acc += frames[((long)b * T + t) * n_fft + n];
            env = fmaf(window[n], window[n], env);
        }
    }
    float v = acc / env;
    if (scale) v /= scale[b];
    wav[(long)b * Lout + s] = v;
}

// ---------------------------------------------------------------------------------
// F16X3 mode, n_fft 400 / hop 100: power uncompress + inverse real DFT, the mirror image of
// stft_fold_x3_kernel.  With theta = 2 pi k n / N and Hermitian weights c_k (1 at DC / Nyquist, else 2)
//   C[n] = sum_k (c_k cos(theta) / N) Yr_k,   S[n] = sum_k (c_k sin(theta) / N) Yi_k     for n = 0 .. N/2
//   x[n] = w[n] (C[n] - S[n]),   x[N - n] = w[N - n] (C[n] + S[n])                         (n = 1 .. N/2 - 1)
// so the contraction is over the 201 bins (7 k32 blocks) and only 201 output columns are computed for the
// 400 samples.  A block = 64 frames of one clip: the uncompressed re (then im) tile is staged in LDS as fp32,
// scaled to [0.5, 1) by an exact power of two per tile and part, split into fp16 hi/lo A fragments kept in
// registers; the two matrix images stream through the same double-buffered 28 KB chunks as the forward kernel.
// ---------------------------------------------------------------------------------
template <int NFFT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void irfft_fold_x3_kernel(SpectralTables tb, const float* __restrict__ re,
                                                            const float* __restrict__ im, int T,
                                                            float* __restrict__ frames) {
    constexpr int H = NFFT / 2, M32 = (H + 1 + 31) / 32, KP = 32 * M32, LD = KP + 4;
    constexpr int CHUNK = 2 * M32 * 1024, NLD = CHUNK / 8 / 256, NBC = (H + 1 + 15) / 16, DEPTH = 3;
    __shared__ __attribute__((aligned(16))) float ytile[64 * LD];
    __shared__ __attribute__((aligned(16))) _Float16 chunk[2][CHUNK];
    __shared__ float wmax[4];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, fb = blockIdx.x;
    const long P = (long)T * tb.F;
    const u32x4* img = reinterpret_cast<const u32x4*>(tb.fold_inv16);
    u32x4 pre[DEPTH][NLD];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < NLD; ++i) pre[d][i] = img[(long)d * (CHUNK / 8) + threadIdx.x + 256 * i];

    f16x8 uh[2][M32], ul[2][M32];                            // [re | im][k32 block]
    float down[2];
    // one pass over the 64 x 201 (re, im) pairs with every load of the thread in flight at once; the uncompressed
    // real parts go to the LDS tile, the imaginary parts wait in registers for the second use of the tile
    constexpr int NST = 64 * KP / 256;
    float vre[NST], vim[NST];
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int fr = i / KP, bin = i - fr * KP;
        const int t = fb * 64 + fr;
        const long off = (long)b * P + (long)(t < T ? t : 0) * tb.F + (bin <= H ? bin : 0);
        vre[k] = re[off];
        vim[k] = im[off];
    }
    float amax_r = 0.f, amax_i = 0.f;
#pragma unroll
    for (int k = 0; k < NST; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int fr = i / KP, bin = i - fr * KP;
        const int t = fb * 64 + fr;
        const bool ok = t < T && bin <= H;
        const float sc = pow_pos(vre[k] * vre[k] + vim[k] * vim[k], 7.0f / 6.0f);
        vre[k] = ok ? vre[k] * sc : 0.f;
        vim[k] = (ok && bin >= 1 && bin < H) ? vim[k] * sc : 0.f;          // imag of DC / Nyquist is ignored
        amax_r = fmaxf(amax_r, fabsf(vre[k]));
        amax_i = fmaxf(amax_i, fabsf(vim[k]));
    }
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        if (part) __syncthreads();                            // every wave has built its re fragments
        float amax = part ? amax_i : amax_r;
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int fr = i / KP, bin = i - fr * KP;
            ytile[fr * LD + bin] = part ? vim[k] : vre[k];
        }
        amax = red_g_max(amax);
        amax = fmaxf(amax, dpp_perm<0xB1>(amax));
        amax = fmaxf(amax, dpp_perm<0x4E>(amax));
        amax = fmaxf(amax, dpp_perm<0x141>(amax));
        amax = fmaxf(amax, dpp_perm<0x140>(amax));
        if (lane == 0) wmax[wv] = amax;
        if (part == 0) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) reinterpret_cast<u32x4*>(chunk[0])[threadIdx.x + 256 * i] = pre[0][i];
        }
        __syncthreads();
        const float tile_max = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        int ex = (int)((__float_as_uint(tile_max) >> 23) & 0xff);
        ex = ex < 1 ? 1 : (ex > 252 ? 252 : ex);
        const float up = __uint_as_float((unsigned)(253 - ex) << 23);
        down[part] = __uint_as_float((unsigned)(ex + 1) << 23);
        const float* row = ytile + (wv * 16 + c) * LD + 8 * g;
#pragma unroll
        for (int m = 0; m < M32; ++m) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(row + 32 * m) * splat4(up);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + 32 * m + 4) * splat4(up);
            split8(a0, a1, uh[part][m], ul[part][m]);
        }
    }

#pragma unroll
    for (int nb = 0; nb < NBC; ++nb) {
        if (nb >= 1 && nb + DEPTH - 1 < NBC) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                pre[(nb + DEPTH - 1) % DEPTH][i] = img[(long)(nb + DEPTH - 1) * (CHUNK / 8) + threadIdx.x + 256 * i];
        }
        const _Float16* cb = chunk[nb & 1] + lane * 8;
        f32x4 ac = splat4(0.f), as = splat4(0.f);
#pragma unroll
        for (int m = 0; m < M32; ++m) {
            const f16x8 ch = *reinterpret_cast<const f16x8*>(cb + m * 1024);
            const f16x8 cl = *reinterpret_cast<const f16x8*>(cb + m * 1024 + 512);
            const f16x8 sh = *reinterpret_cast<const f16x8*>(cb + (M32 + m) * 1024);
            const f16x8 sl = *reinterpret_cast<const f16x8*>(cb + (M32 + m) * 1024 + 512);
            ac = mfma32h(uh[0][m], ch, ac);
            as = mfma32h(uh[1][m], sh, as);
            ac = mfma32l(uh[0][m], cl, ac);
            as = mfma32l(uh[1][m], sl, as);
            ac = mfma32l(ul[0][m], ch, ac);
            as = mfma32l(ul[1][m], sh, as);
        }
        const int n = nb * 16 + c;
        if (n <= H) {
            const float w0 = tb.window[n], w1 = tb.window[n >= 1 && n < H ? NFFT - n : n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = fb * 64 + wv * 16 + 4 * g + r;
                if (t < T) {
                    const float C = ac[r] * down[0], S = as[r] * down[1];
                    float* fr = frames + ((long)b * T + t) * NFFT;
                    fr[n] = (C - S) * w0;
                    if (n >= 1 && n < H) fr[NFFT - n] = (C + S) * w1;
                }
            }
        }
        if (nb + 1 < NBC) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                reinterpret_cast<u32x4*>(chunk[(nb + 1) & 1])[threadIdx.x + 256 * i] = pre[(nb + 1) % DEPTH][i];
        }
        __syncthreads();
    }
}

void launch_uncompress_istft(LaunchCtx ctx, const SpectralTables& tb, const float* re, const float* im,
                             const float* scale, int B, int T, float* frames_ws, float* wav_out) {
    static const int k_fft = env_knob("CMGAN_STFT_FFT", STFT_FFT, 0, 1);
    if (k_fft && tb.fold_inv16 && tb.n_fft == 400 && tb.hop == 100 && tb.F == 201) {
        launch_istft_fft400(ctx, re, im, scale, tb.window, B, T, wav_out);
        return;
    }
    if (tb.fold_inv16 && tb.n_fft == 400 && tb.hop == 100) {
        dim3 grid64((T + 63) / 64, B);
        LAUNCH(ctx, "uncompress_irfft",
               (irfft_fold_x3_kernel<400><<<grid64, 256, 0, ctx.stream>>>(tb, re, im, T, frames_ws)));
    } else {
        dim3 grid((T + 15) / 16, B);
        const size_t shm = (size_t)16 * (2 * tb.FB * 16 + 4) * sizeof(float);
        LAUNCH(ctx, "uncompress_irfft",
               (uncompress_irfft_kernel<<<grid, 256, shm, ctx.stream>>>(tb, re, im, T, frames_ws)));
    }
    const int Lout = tb.hop * (T - 1);
    dim3 g2((Lout + 255) / 256, B);
    LAUNCH(ctx, "ola", (ola_kernel<<<g2, 256, 0, ctx.stream>>>(frames_ws, tb.window, scale, tb.n_fft, tb.hop, T, Lout,
                                                              wav_out)));
}

// ---------------------------------------------------------------------------------
// stand-alone utils.power_compress / power_uncompress (src/utils.py:20-39) in the
// reference's own layouts, for drop-in use outside the fused pipeline.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void power_compress_kernel(const float* __restrict__ x, long FT, long total,
                                                             float* __restrict__ y) {
    // x[B,F,T,2] -> y[B,2,F,T]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / FT, p = i - b * FT;
        const float re = x[i * 2], im = x[i * 2 + 1];
        const float m2 = re * re + im * im;
        const float s = m2 > 0.f ? powf(m2, -0.35f) : 0.f;
        y[(b * 2 + 0) * FT + p] = re * s;
        y[(b * 2 + 1) * FT + p] = im * s;
    }
}

__global__ __launch_bounds__(256) void power_uncompress_kernel(const float* __restrict__ re,
                                                               const float* __restrict__ im, long total,
                                                               float* __restrict__ y) {
    // real, imag [B,1,F,T] -> y[B,1,F,T,2]
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float a = re[i], b = im[i];
        const float m2 = a * a + b * b;
        const float s = m2 > 0.f ? powf(m2, 7.0f / 6.0f) : 0.f;
        y[i * 2] = a * s;
        y[i * 2 + 1] = b * s;
    }
}

void launch_power_compress(LaunchCtx ctx, const float* x, int B, int F, int T, float* y) {
    const long FT = (long)F * T, total = (long)B * FT;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    LAUNCH(ctx, "power_compress", (power_compress_kernel<<<grid, 256, 0, ctx.stream>>>(x, FT, total, y)));
}

void launch_power_uncompress(LaunchCtx ctx, const float* re, const float* im, int B, int F, int T, float* y) {
    const long total = (long)B * F * T;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    LAUNCH(ctx, "power_uncompress", (power_uncompress_kernel<<<grid, 256, 0, ctx.stream>>>(re, im, total, y)));
}
