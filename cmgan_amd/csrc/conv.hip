// conv.hip - the convolutional encoder / decoders of TSCNet (reference:
// src/models/generator.py:6-69, 102-156, 174-196) as CDNA4 kernels.
//
// Activations are channels-last [B, T*F, 64] fp32.  InstanceNorm2d is a global (T x F)
// reduction between every pair of convs, so each conv writes its RAW output once and
// accumulates per-(b, channel) sum / sum-of-squares partials in its epilogue; a tiny
// finalize kernel turns them into scale/shift, and the NEXT consumer applies
// scale/shift + PReLU while staging its input ("normalise on load").  The dense block's
// concat (generator.py:46) is never materialised: a conv takes up to 4 input slots.
//
// conv3_kernel is an implicit GEMM on v_mfma_f32_16x16x4_f32, evaluated transposed
// (out^T[co][pos] = sum_{tap,ci} W[tap][co][ci] * in[pos+shift(tap)][ci]): weights are the
// A operand (fragment-major, staged through LDS as a linear copy), activations the B
// operand.  A tile is 128 consecutive positions of the (t, f') plane flattened with ONE
// virtual zero column per row (pitch F+1), so the f-1 / f+1 taps are just the staged rows
// shifted by one and need no per-MFMA masking; the time taps t-dil / t are two staged planes.
#include "kernels.h"

#define CONV_TILE 128
#define CONV_ROWS (CONV_TILE + 2)
#define CONV_RS 24            // floats per LDS activation row (16 used): 96 B pitch = conflict-free ds_read_b128

__device__ __forceinline__ f32x4 norm_prelu4(f32x4 v, f32x4 sc, f32x4 sh, f32x4 al) {
    f32x4 y = v * sc + sh;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = y[e] >= 0.f ? y[e] : al[e] * y[e];
    return y;
}

template <int NT, int COUT>
__global__ __launch_bounds__(256) void conv3_kernel(ConvArgs a) {
    constexpr int CB = COUT / 16;
    constexpr int TAPS = NT * 3;
    constexpr int NSTAGE = (NT * CONV_ROWS * 4 + 255) / 256;   // float4 per thread per chunk
    constexpr int WF4 = TAPS * CB * 64;                         // float4 of weights per chunk
    __shared__ __attribute__((aligned(16))) float act[NT * CONV_ROWS * CONV_RS];
    __shared__ __attribute__((aligned(16))) float wl[TAPS * CB * 256];
    __shared__ float red[4][COUT][2];

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4, wv = tid >> 6;
    const int b = blockIdx.y;
    const int Fp = a.F + 1;
    const int q0 = blockIdx.x * CONV_TILE;

    // staging metadata: which input position each of this thread's float4 comes from
    int apos[NSTAGE];
#pragma unroll
    for (int e = 0; e < NSTAGE; ++e) {
        const int idx = tid + 256 * e;
        apos[e] = -1;
        if (idx < NT * CONV_ROWS * 4) {
            const int rr = idx >> 2;
            const int kt = rr / CONV_ROWS, p = rr - kt * CONV_ROWS;
            const int q = q0 - 1 + p;
            if (q >= 0) {
                const int t = q / Fp, f = q - t * Fp;
                const int tin = (NT == 2) ? t - a.dil * (1 - kt) : t;
                if (f < a.F && t < a.T && tin >= 0) apos[e] = (b * a.T + tin) * a.F + f;
            }
        }
    }
    const int qd = tid & 3;

    f32x4 acc[CB][2];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) { acc[cb][0] = splat4(0.f); acc[cb][1] = splat4(0.f); }

    const int nchunks = a.nslots * 4;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int s = chunk >> 2, cc = chunk & 3;
        const float* src = a.in[s];
        const bool tr = a.nscale[s] != nullptr;
        f32x4 sc = splat4(1.f), sh = splat4(0.f), al = splat4(1.f);
        if (tr) {
            sc = ldg4(a.nscale[s] + b * 64 + cc * 16 + qd * 4);
            sh = ldg4(a.nshift[s] + b * 64 + cc * 16 + qd * 4);
            al = ldg4(a.nalpha[s] + cc * 16 + qd * 4);
        }
        // all global loads of the chunk are issued before the barrier / any LDS store (no per-iteration waits)
        f32x4 aval[NSTAGE], wval[WF4 / 256];
#pragma unroll
        for (int e = 0; e < NSTAGE; ++e)
            aval[e] = ldg4(src + (long)(apos[e] >= 0 ? apos[e] : 0) * 64 + cc * 16 + qd * 4);
        const float* wsrc = a.w + (long)chunk * WF4 * 4;
#pragma unroll
        for (int i = 0; i < WF4 / 256; ++i) wval[i] = ldg4(wsrc + (tid + 256 * i) * 4);
        __syncthreads();          // previous chunk's fragments are consumed
#pragma unroll
        for (int e = 0; e < NSTAGE; ++e) {
            const int idx = tid + 256 * e;
            if (idx < NT * CONV_ROWS * 4) {
                f32x4 val = aval[e];
                if (tr) val = norm_prelu4(val, sc, sh, al);
                if (apos[e] < 0) val = splat4(0.f);
                *reinterpret_cast<f32x4*>(&act[(idx >> 2) * CONV_RS + (idx & 3) * 4]) = val;
            }
        }
#pragma unroll
        for (int i = 0; i < WF4 / 256; ++i) *reinterpret_cast<f32x4*>(&wl[(tid + 256 * i) * 4]) = wval[i];
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int kt = tap / 3, kf = tap - kt * 3;
            f32x4 bf[2];
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
                bf[tb] = *reinterpret_cast<const f32x4*>(
                    &act[((kt * CONV_ROWS) + 32 * wv + 16 * tb + c + kf) * CONV_RS + 4 * g]);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(&wl[(tap * CB + cb) * 256 + lane * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[cb][0] = mfma16(af[r], bf[0][r], acc[cb][0]);
                    acc[cb][1] = mfma16(af[r], bf[1][r], acc[cb][1]);
                }
            }
        }
    }

    // ---- epilogue: bias, store, InstanceNorm partial sums ---------------------------
    bool ok[2];
    long obase[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
        const int q = q0 + 32 * wv + 16 * tb + c;
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
        for (int tb = 0; tb < 2; ++tb) {
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
            a.partials[(((long)b * a.ntiles + blockIdx.x) * COUT + co) * 2 + wh] = t;
        }
    }
}

int conv3_ntiles(int T, int F) { return (T * (F + 1) + CONV_TILE - 1) / CONV_TILE; }

void launch_conv3(LaunchCtx ctx, const ConvArgs& a, int B, int time_taps, int cout) {
    dim3 grid(a.ntiles, B);
    if (time_taps == 2 && cout == 64)
        LAUNCH(ctx, "conv_dense", (conv3_kernel<2, 64><<<grid, 256, 0, ctx.stream>>>(a)));
    else if (time_taps == 1 && cout == 64)
        LAUNCH(ctx, "conv_1x3", (conv3_kernel<1, 64><<<grid, 256, 0, ctx.stream>>>(a)));
    else
        LAUNCH(ctx, "conv_subpixel", (conv3_kernel<1, 128><<<grid, 256, 0, ctx.stream>>>(a)));
}

// ---------------------------------------------------------------------------------
// conv_1: mag/re/im -> 64 channels (1x1 conv, K = 3) + partial sums.
// generator.py:175-179 (prologue) and :54.  spec is the planar model input [B,2,T,F].
// lane = output channel; a wave walks 64 positions, broadcasting each position's
// (mag, re, im) with a shuffle, so every store is one full 256 B row.
// ---------------------------------------------------------------------------------
#define CIN_TILE 256
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ spec, const float* __restrict__ w,
                                                      float* __restrict__ out, float* __restrict__ partials,
                                                      int P, int ntiles) {
    __shared__ float red[4][64][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * CIN_TILE + wv * 64;
    const int pm = p0 + lane;
    float re_m = 0.f, im_m = 0.f;
    if (pm < P) {
        re_m = spec[((long)b * 2 + 0) * P + pm];
        im_m = spec[((long)b * 2 + 1) * P + pm];
    }
    const float mg_m = sqrtf(re_m * re_m + im_m * im_m);
    const float w0 = w[lane], w1 = w[64 + lane], w2 = w[128 + lane], bb = w[192 + lane];
    float s1 = 0.f, s2 = 0.f;
    const int n = min(64, P - p0);
    for (int i = 0; i < n; ++i) {
        const float re = __shfl(re_m, i), im = __shfl(im_m, i), mg = __shfl(mg_m, i);
        const float v = fmaf(w0, mg, fmaf(w1, re, fmaf(w2, im, bb)));
        out[((long)b * P + p0 + i) * 64 + lane] = v;
        s1 += v;
        s2 = fmaf(v, v, s2);
    }
    red[wv][lane][0] = s1;
    red[wv][lane][1] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int co = threadIdx.x >> 1, wh = threadIdx.x & 1;
        const float t = (red[0][co][wh] + red[1][co][wh]) + (red[2][co][wh] + red[3][co][wh]);
        partials[(((long)b * ntiles + blockIdx.x) * 64 + co) * 2 + wh] = t;
    }
}

int conv_in_ntiles(int P) { return (P + CIN_TILE - 1) / CIN_TILE; }

void launch_conv_in(LaunchCtx ctx, const float* spec, const float* w, float* out, float* partials, int B, int P) {
    const int nt = conv_in_ntiles(P);
    LAUNCH(ctx, "conv_in", (conv_in_kernel<<<dim3(nt, B), 256, 0, ctx.stream>>>(spec, w, out, partials, P, nt)));
}

// ---------------------------------------------------------------------------------
// InstanceNorm2d(affine) statistics -> per-(b, c) scale / shift.  Partials are reduced in
// a FIXED order in fp64 so results are bit-reproducible (needed for the sharded == single
// GPU guarantee).  Biased variance, eps 1e-5 (generator.py:35,55,61,148).
// fold2: channel c also owns partial column c + 64 (pixel-shuffled sub-pixel conv).
// ---------------------------------------------------------------------------------
#define INF_NP 16              // partial sums per (clip, channel): 16 x 64 threads walk the tiles 16 apart, 8 loads in flight each
__global__ __launch_bounds__(64 * INF_NP) void in_finalize_kernel(const float* __restrict__ partials, int ntiles,
                                                                  int cstride, int fold2, double count,
                                                                  const float* __restrict__ gb,
                                                                  float* __restrict__ nscale, float* __restrict__ nshift) {
    __shared__ double acc[INF_NP][64][2];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int b = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int t0 = part; t0 < ntiles; t0 += INF_NP * 8) {      // 8 independent loads in flight, fixed summation order
        float v1[8], v2[8], f1[8], f2[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + INF_NP * k < ntiles ? t0 + INF_NP * k : ntiles - 1;
            const float* p = partials + (((long)b * ntiles + t) * cstride + c) * 2;
            v1[k] = p[0];
            v2[k] = p[1];
            f1[k] = fold2 ? p[128] : 0.f;
            f2[k] = fold2 ? p[129] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (t0 + INF_NP * k < ntiles) {
                s1 += (double)v1[k];
                s2 += (double)v2[k];
                s1 += (double)f1[k];
                s2 += (double)f2[k];
            }
        }
    }
    acc[part][c][0] = s1;
    acc[part][c][1] = s2;
    __syncthreads();
    if (part == 0) {
        s1 = 0.0; s2 = 0.0;
#pragma unroll
        for (int q = 0; q < INF_NP; q += 4) {                 // fixed order: groups of four, then the groups
            s1 += (acc[q][c][0] + acc[q + 1][c][0]) + (acc[q + 2][c][0] + acc[q + 3][c][0]);
            s2 += (acc[q][c][1] + acc[q + 1][c][1]) + (acc[q + 2][c][1] + acc[q + 3][c][1]);
        }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + 1e-5);
        const double sc = (double)gb[c] * rstd;
        nscale[b * 64 + c] = (float)sc;
        nshift[b * 64 + c] = (float)((double)gb[64 + c] - mean * sc);
    }
}

void launch_in_finalize(LaunchCtx ctx, const float* partials, int B, int ntiles, int cstride, int fold2,
                        double count, const float* gb, float* nscale, float* nshift) {
    LAUNCH(ctx, "in_finalize", (in_finalize_kernel<<<B, 64 * INF_NP, 0, ctx.stream>>>(partials, ntiles, cstride, fold2, count,
                                                                              gb, nscale, nshift)));
}

// materialise InstanceNorm + PReLU (the encoder output feeds the conformer residual stream)
__global__ __launch_bounds__(256) void in_apply_kernel(const float* __restrict__ in, const float* __restrict__ nscale,
                                                       const float* __restrict__ nshift,
                                                       const float* __restrict__ alpha, float* __restrict__ out,
                                                       long P, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        const long pos = i >> 4;
        const int b = (int)(pos / P);
        const f32x4 v = ldg4(in + i * 4);
        stg4(out + i * 4, norm_prelu4(v, ldg4(nscale + b * 64 + c4 * 4), ldg4(nshift + b * 64 + c4 * 4),
                                      ldg4(alpha + c4 * 4)));
    }
}

void launch_in_apply(LaunchCtx ctx, const float* in, const float* nscale, const float* nshift, const float* alpha,
                     float* out, int B, long P) {
    const long total4 = (long)B * P * 16;
    const int grid = (int)std::min<long>((total4 + 255) / 256, 8192);
    LAUNCH(ctx, "in_apply", (in_apply_kernel<<<grid, 256, 0, ctx.stream>>>(in, nscale, nshift, alpha, out, P, total4)));
}

// ---------------------------------------------------------------------------------
// Decoder tails.  Both end in a (1,2) conv over the pixel-shuffled tensor SP[B,T,W,64]
// (W = 2F'): mask conv_1 64->1 (generator.py:127,136), complex conv 64->2 (:149,155).
// tail_proj computes the per-position projections d[pos][row] = sum_c Wt[row][c] * sp[pos][c]
// (rows = (out, kf) pairs, <= 4) with one MFMA chain per 16 positions; the two kf taps are
// summed across neighbouring positions by the consumers below.
// The complex path normalises on load (IN + PReLU, generator.py:153-154).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tail_proj_kernel(const float* __restrict__ sp,
                                                        const float* __restrict__ nscale,
                                                        const float* __restrict__ nshift,
                                                        const float* __restrict__ alpha,
                                                        const float* __restrict__ tailw, float* __restrict__ d,
                                                        long P2, long total) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blk * 16 >= total) return;
    const long pos = blk * 16 + c;
    const bool ok = pos < total;
    const long row = ok ? pos : total - 1;
    const int b = (int)(row / P2);
    f32x4 xf[1][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        f32x4 v = ldg4(sp + row * 64 + 16 * kb + 4 * g);
        if (nscale)
            v = norm_prelu4(v, ldg4(nscale + b * 64 + 16 * kb + 4 * g), ldg4(nshift + b * 64 + 16 * kb + 4 * g),
                            ldg4(alpha + 16 * kb + 4 * g));
        xf[0][kb] = v;
    }
    f32x4 acc[1] = {splat4(0.f)};
    lin_acc<4, 1>(tailw + lane * 4, xf, acc);
    if (ok && g == 0) stg4(d + row * 4, acc[0]);
}

void launch_tail_proj(LaunchCtx ctx, const float* sp, const float* nscale, const float* nshift, const float* alpha,
                      const float* tailw, float* d, int B, long P2) {
    const long total = (long)B * P2;
    const long nblk = (total + 15) / 16;
    LAUNCH(ctx, "tail_proj", (tail_proj_kernel<<<(unsigned)((nblk + 3) / 4), 256, 0, ctx.stream>>>(
                                 sp, nscale, nshift, alpha, tailw, d, P2, total)));
}

// mask value before its InstanceNorm: conv_1 over (f, f+1) + bias
__device__ __forceinline__ float mask_raw(const float* __restrict__ dm, long base, float bias) {
    return dm[base * 4 + 0] + dm[(base + 1) * 4 + 1] + bias;
}

// InstanceNorm2d(1) statistics of the mask branch: one block per clip, fixed-order fp64 tree
__global__ __launch_bounds__(1024) void mask_stats_kernel(const float* __restrict__ dm,
                                                          const float* __restrict__ scalars, int T, int F,
                                                          float* __restrict__ mstat) {
    __shared__ double r1[1024], r2[1024];
    const int b = blockIdx.x;
    const int W = F + 1;
    const float bias = scalars[0];
    double s1 = 0.0, s2 = 0.0;
    const long n = (long)T * F;
    for (long i0 = threadIdx.x; i0 < n; i0 += 1024 * 4) {           // 4 independent loads in flight per thread
        float mv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long i = i0 + 1024 * k < n ? i0 + 1024 * k : n - 1;
            const int t = (int)(i / F), f = (int)(i - (long)t * F);
            mv[k] = mask_raw(dm, ((long)b * T + t) * W + f, bias);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + 1024 * k < n) {
                s1 += (double)mv[k];
                s2 += (double)mv[k] * (double)mv[k];
            }
        }
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            r1[threadIdx.x] += r1[threadIdx.x + st];
            r2[threadIdx.x] += r2[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = r1[0] / (double)n;
        double var = r2[0] / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        mstat[b * 2 + 0] = (float)mean;
        mstat[b * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

void launch_mask_stats(LaunchCtx ctx, const float* dm, const float* scalars, int B, int T, int F, float* mstat) {
    LAUNCH(ctx, "mask_stats", (mask_stats_kernel<<<B, 1024, 0, ctx.stream>>>(dm, scalars, T, F, mstat)));
}

// ---------------------------------------------------------------------------------
// Output stage: mask tail (IN(1) -> PReLU -> 1x1 conv -> per-frequency PReLU,
// generator.py:128-131,137-139), complex tail bias, and the recombination
// final = mask * mag * (cos, sin)(angle(x)) + complex  ==  mask * x + complex
// (generator.py:188-194; mag*cos(angle) is the input's real part).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void final_combine_kernel(const float* __restrict__ spec,
                                                            const float* __restrict__ dm,
                                                            const float* __restrict__ dc,
                                                            const float* __restrict__ mstat,
                                                            const float* __restrict__ sca,
                                                            const float* __restrict__ prelu_out,
                                                            const float* __restrict__ cx_bias, int T, int F,
                                                            long total, float* __restrict__ out_re,
                                                            float* __restrict__ out_im, float* __restrict__ tap_mask,
                                                            float* __restrict__ tap_cplx) {
    const long P = (long)T * F;
    const int W = F + 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / P);
        const long p = i - (long)b * P;
        const int t = (int)(p / F), f = (int)(p - (long)t * F);
        const long base = ((long)b * T + t) * W + f;
        float m = mask_raw(dm, base, sca[0]);
        m = (m - mstat[b * 2]) * mstat[b * 2 + 1] * sca[1] + sca[2];
        m = m >= 0.f ? m : sca[3] * m;
        m = m * sca[4] + sca[5];
        m = m >= 0.f ? m : prelu_out[f] * m;
        const float c0 = dc[base * 4 + 0] + dc[(base + 1) * 4 + 1] + cx_bias[0];
        const float c1 = dc[base * 4 + 2] + dc[(base + 1) * 4 + 3] + cx_bias[1];
        const float re = spec[((long)b * 2 + 0) * P + p], im = spec[((long)b * 2 + 1) * P + p];
        out_re[i] = fmaf(m, re, c0);
        out_im[i] = fmaf(m, im, c1);
        if (tap_mask) tap_mask[i] = m;
        if (tap_cplx) {
            tap_cplx[((long)b * 2 + 0) * P + p] = c0;
            tap_cplx[((long)b * 2 + 1) * P + p] = c1;
        }
    }
}

void launch_final_combine(LaunchCtx ctx, const float* spec, const float* dm, const float* dc, const float* mstat,
                          const float* mk_scalars, const float* prelu_out, const float* cx_bias, int B, int T,
                          int F, float* out_re, float* out_im, float* tap_mask, float* tap_cplx) {
    const long total = (long)B * T * F;
    const int grid = (int)std::min<long>((total + 255) / 256, 8192);
    LAUNCH(ctx, "final_combine",
           (final_combine_kernel<<<grid, 256, 0, ctx.stream>>>(spec, dm, dc, mstat, mk_scalars, prelu_out, cx_bias, T,
                                                               F, total, out_re, out_im, tap_mask, tap_cplx)));
}

// channels-last [B,P,64] -> NCHW [B,64,P] (test taps only)
__global__ __launch_bounds__(256) void cl_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         long P) {
    __shared__ float tile[64][65];
    const int b = blockIdx.y;
    const long p0 = (long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int pp = i >> 6, ch = i & 63;
        tile[pp][ch] = (p0 + pp < P) ? in[((long)b * P + p0 + pp) * 64 + ch] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int ch = i >> 6, pp = i & 63;
        if (p0 + pp < P) out[((long)b * 64 + ch) * P + p0 + pp] = tile[pp][ch];
    }
}

void launch_cl_to_nchw(LaunchCtx ctx, const float* in, float* out, int B, long P) {
    dim3 grid((unsigned)((P + 63) / 64), B);
    LAUNCH(ctx, "cl_to_nchw", (cl_to_nchw_kernel<<<grid, 256, 0, ctx.stream>>>(in, out, P)));
}
