// train.hip - device pieces of the reference's training / validation step (src/train.py) that sit directly on the
// generator forward path: the non-adversarial loss terms of Trainer.calculate_generator_loss (train.py:124-151)
// as deterministic reductions (the scalars the data-parallel step all-reduces over RCCL).
#include "train.h"
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

// ---------------------------------------------------------------------------------
// loss_ri  = mse(est_real, clean_real) + mse(est_imag, clean_imag)        train.py:135-137
// loss_mag = mse(|est|, |clean|)                                           train.py:132-134 (mags: train.py:100-101)
// time     = mean |est_audio - clean_audio|                                train.py:139-141
// plus mean (est_audio - clean_audio)^2 for logging.
// Two fixed-shape passes (LOSS_BLOCKS partial sums in fp64, then one block) so the result does not depend on
// the batch split or on atomics' arrival order: bit-reproducible, like every other reduction on the path.
// Spectral operands are in the model layout ([B,1,T,F] estimates, [B,2,T,F] compressed clean spectrogram);
// the reference permutes to [B,1,F,T] first, which an elementwise mean does not see.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum4(double (&v)[4], double* red /* [4][256] */) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) red[k * 256 + tid] = v[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k * 256 + tid] += red[k * 256 + tid + st];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = red[k * 256];
}

__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ er, const float* __restrict__ ei,
                                                           const float* __restrict__ clean_spec, long P, long nspec,
                                                           const float* __restrict__ ea, const float* __restrict__ ca,
                                                           long naudio, double* __restrict__ partials) {
    __shared__ double red[4 * 256];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};            // ri, mag, |d|, d^2
    const long stride = (long)gridDim.x * 256;
    if (er && ei && clean_spec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nspec; i += stride) {
            const long b = i / P, p = i - b * P;
            const float cr = clean_spec[(b * 2) * P + p], ci = clean_spec[(b * 2 + 1) * P + p];
            const float xr = er[i], xi = ei[i];
            const float dr = xr - cr, di = xi - ci;
            const float dm = sqrtf(xr * xr + xi * xi) - sqrtf(cr * cr + ci * ci);
            acc[0] += (double)dr * dr + (double)di * di;
            acc[1] += (double)dm * dm;
        }
    }
    if (ea && ca) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < naudio; i += stride) {
            const float d = ea[i] - ca[i];
            acc[2] += (double)fabsf(d);
            acc[3] += (double)d * d;
        }
    }
    block_sum4(acc, red);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) partials[(long)blockIdx.x * 4 + k] = acc[k];
    }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const double* __restrict__ partials, int nblocks, double nspec,
                                                         double naudio, float* __restrict__ out4) {
    __shared__ double red[4 * 256];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblocks; i += 256) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += partials[(long)i * 4 + k];
    }
    block_sum4(acc, red);
    if (threadIdx.x == 0) {
        out4[0] = nspec > 0 ? (float)(acc[0] / nspec) : 0.f;
        out4[1] = nspec > 0 ? (float)(acc[1] / nspec) : 0.f;
        out4[2] = naudio > 0 ? (float)(acc[2] / naudio) : 0.f;
        out4[3] = naudio > 0 ? (float)(acc[3] / naudio) : 0.f;
    }
}

void launch_loss_terms(LaunchCtx ctx, const float* est_real, const float* est_imag, const float* clean_spec, int B,
                       long P, const float* est_audio, const float* clean_audio, long naudio, double* partials,
                       float* out4) {
    const bool spec = est_real && est_imag && clean_spec;
    const long nspec = spec ? (long)B * P : 0;
    const bool audio = est_audio && clean_audio;
    LAUNCH(ctx, "loss_terms", (loss_partial_kernel<<<LOSS_BLOCKS, 256, 0, ctx.stream>>>(
                                  est_real, est_imag, clean_spec, P, nspec, est_audio, clean_audio,
                                  audio ? naudio : 0, partials)));
    LAUNCH(ctx, "loss_terms", (loss_final_kernel<<<1, 256, 0, ctx.stream>>>(partials, LOSS_BLOCKS, (double)nspec,
                                                                            audio ? (double)naudio : 0.0, out4)));
}

// =====================================================================================
// Training-mode FeedForward (first backward slice of SURVEY.md N2).
//   y = 0.5 * m2 * (W2 (m1 * Swish(W1 LN(x) + b1)) + b2)
// = Scale(0.5, PreNorm(dim, FeedForward(dim, mult=4, dropout)))(x) of src/models/conformer.py:54-72,136-148,211 in
// TRAIN mode, with the two nn.Dropout layers expressed as caller-supplied byte keep-masks m1 [M,256], m2 [M,64]
// (non-zero = keep, kept values scaled by 1/(1-p); NULL = no dropout) so that the result is reproducible and comparable
// with autograd on the oracle.
// The residual add of ConformerBlock.forward (conformer.py:217) stays with the caller (dx_total = dy + dx).
//
// Same per-token transposed MFMA chain as the inference kernels (out^T = W x^T on v_mfma_f32_16x16x4_f32, a wave owns
// 16 tokens, C-fragments of one layer are the B-fragments of the next), on RAW parameters: the four A-operand images
// (W1, W2, W2^T, W1^T, fragment-major) are re-packed on the device at every call because an optimiser step changes them.
// Backward recomputes LN / W1 / Swish from x instead of saving the [M,256] hidden activations of the forward
// (1 GB at B = 32): on this machine the extra 64x256 GEMM is cheaper than the HBM round trip.
//   dz = 0.5 m2 dy              dW2 = dz^T d1     db2 = colsum dz
//   dd1 = W2^T dz               dh = m1 dd1 Swish'(h),  Swish'(h) = s(h) (1 + h (1 - s(h)))
//   dW1 = dh^T xn   db1 = colsum dh               dxn = W1^T dh
//   dgamma = colsum(dxn xh)  dbeta = colsum dxn   dx = rstd (g dxn - mean(g dxn) - xh mean(g dxn xh))
// Weight gradients are token-contractions (K = M): split-K MFMA products into fixed-shape partial slabs that a
// second kernel adds in a fixed order - deterministic, no atomics.
// =====================================================================================
// row-major W [R, K] (leading dimension ldw; transpose = 1: the image of W^T) -> fragment-major image [R/16][K/16][64][4]
// (common.hip.h); the four images of one module in ONE launch (blockIdx.y = image): a training step re-packs 128 images
struct PackJob { const float* w; int R, K, ldw, transpose; float* out; };
struct PackJobs { PackJob j[4]; };
__global__ void pack_fm4_kernel(PackJobs jobs) {
    const PackJob& q = jobs.j[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.R * q.K) return;
    const int r = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
    const int KB = q.K / 16, rb = blk / KB, kb = blk - rb * KB;
    const int row = 16 * rb + (lane & 15), col = 16 * kb + 4 * (lane >> 4) + r;
    q.out[i] = q.transpose ? q.w[(long)col * q.ldw + row] : q.w[(long)row * q.ldw + col];
}
static void launch_pack4(LaunchCtx ctx, const char* label, const PackJobs& jobs, int njobs = 4) {
    int most = 0;
    for (int k = 0; k < njobs; ++k) most = jobs.j[k].R * jobs.j[k].K > most ? jobs.j[k].R * jobs.j[k].K : most;
    LAUNCH(ctx, label, (pack_fm4_kernel<<<dim3((most + 255) / 256, njobs), 256, 0, ctx.stream>>>(jobs)));
}

// Dropout keep-masks are BYTES (non-zero = keep; the kept values are scaled by `ms` = 1 / (1 - p)): a quarter of the
// traffic of float masks - per conformer block and token 704 mask values are read in the forward and again in the backward
__device__ __forceinline__ unsigned mask_word(const unsigned char* __restrict__ m, long idx) {   // idx % 4 == 0
    return *reinterpret_cast<const unsigned*>(m + idx);
}
__device__ __forceinline__ f32x4 mask4w(unsigned v, float ms) {
    f32x4 r;
    r[0] = (v & 0x000000ffu) ? ms : 0.f;
    r[1] = (v & 0x0000ff00u) ? ms : 0.f;
    r[2] = (v & 0x00ff0000u) ? ms : 0.f;
    r[3] = (v & 0xff000000u) ? ms : 0.f;
    return r;
}
__device__ __forceinline__ f32x4 mask4(const unsigned char* __restrict__ m, long idx, float ms) {
    const unsigned v = *reinterpret_cast<const unsigned*>(m + idx);          // idx is a multiple of 4
    f32x4 r;
    r[0] = (v & 0x000000ffu) ? ms : 0.f;
    r[1] = (v & 0x0000ff00u) ? ms : 0.f;
    r[2] = (v & 0x00ff0000u) ? ms : 0.f;
    r[3] = (v & 0xff000000u) ? ms : 0.f;
    return r;
}

struct FfnTrainImg {
    const float *w1, *w2, *w2t, *w1t;     // fm [16][4], [4][16], [16][4], [4][16]
    const float *gamma, *beta, *b1, *b2;
};

__device__ __forceinline__ bool ffn_load_norm(const float* __restrict__ x, long M, long t0, int c, int g,
                                              const FfnTrainImg& w, f32x4 (&xh)[4], f32x4 (&xn)[1][4], float& rstd,
                                              long& row) {
    const long t = t0 + c;
    const bool ok = t < M;
    row = ok ? t : M - 1;
    f32x4 xv[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) xv[kb] = ldg4(x + row * 64 + 16 * kb + 4 * g);
    float mean;
    ln_stats(xv, mean, rstd);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        xh[kb] = (xv[kb] - splat4(mean)) * splat4(rstd);
        xn[0][kb] = xh[kb] * ldg4(w.gamma + 16 * kb + 4 * g) + ldg4(w.beta + 16 * kb + 4 * g);
    }
    return ok;
}

// four weight fragments = the A operands of 16 MFMAs.  rows: fragments [blk][0..3] of an image whose k-blocks are
// contiguous (W1 [16][4], W2^T [16][4]); cols: fragments [0..3][blk] of an image with 16 k-blocks per row (W2, W1^T [4][16])
struct FfnFrag { f32x4 a[4]; };
__device__ __forceinline__ FfnFrag ffn_frag_rows(const float* __restrict__ img, int blk, int lane) {
    FfnFrag f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) f.a[kb] = ldg4(img + ((long)blk * 4 + kb) * 256 + lane * 4);
    __builtin_amdgcn_sched_barrier(0);    // keep the loads HERE: the scheduler otherwise sinks them next to their use
    return f;
}
__device__ __forceinline__ FfnFrag ffn_frag_cols(const float* __restrict__ img, int blk, int lane) {
    FfnFrag f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) f.a[ob] = ldg4(img + ((long)ob * 16 + blk) * 256 + lane * 4);
    __builtin_amdgcn_sched_barrier(0);
    return f;
}
// acc + sum_kb frag[kb] x xf[kb]: one output block of a per-token linear layer (lin_acc<4, 1> on prefetched fragments)
__device__ __forceinline__ f32x4 ffn_frag_mma(const FfnFrag& f, const f32x4 (&xf)[4], f32x4 acc) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(f.a[kb][r], xf[kb][r], acc);
    return acc;
}

__global__ __launch_bounds__(256) void ffn_train_fwd_kernel(const float* __restrict__ x, long M, FfnTrainImg w,
                                                            const unsigned char* __restrict__ m1,
                                                            const unsigned char* __restrict__ m2, float ms,
                                                            const float* __restrict__ res, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = ffn_load_norm(x, M, t0, c, g, w, xh, xn, rstd, row);
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = ldg4(w.b2 + 16 * ob + 4 * g);
    // the weight fragments of the NEXT 16 MFMAs are fetched while the current 16 run (as written by the compiler, every
    // group of 4 MFMAs waited for its own fragment: the kernel ran at L1 latency, not at the matrix rate)
    // (loads retire in order: the small mask / bias loads of a block are issued BEFORE the fragment loads that are
    // still in flight when they are needed)
    unsigned mw = m1 ? mask_word(m1, row * 256 + 4 * g) : 0u;
    f32x4 b1v = ldg4(w.b1 + 4 * g);
    FfnFrag g1 = ffn_frag_rows(w.w1, 0, lane), g2;
    for (int hb = 0; hb < 16; ++hb) {
        g2 = ffn_frag_cols(w.w2, hb, lane);
        const f32x4 h = ffn_frag_mma(g1, xn[0], b1v);
        const f32x4 mk = m1 ? mask4w(mw, ms) : splat4(1.f);
        const int hn = hb < 15 ? hb + 1 : 15;
        if (m1) mw = mask_word(m1, row * 256 + 16 * hn + 4 * g);
        b1v = ldg4(w.b1 + 16 * hn + 4 * g);
        g1 = ffn_frag_rows(w.w1, hn, lane);
        f32x4 s;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = swishf(h[r]) * mk[r];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ob] = mfma16(g2.a[ob][r], s[r], acc[ob]);
        }
    }
    if (ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 v = acc[ob] * splat4(0.5f);
            if (m2) v = v * mask4(m2, row * 64 + 16 * ob + 4 * g, ms);
            if (res) v = v + ldg4(res + row * 64 + 16 * ob + 4 * g);      // the block's residual add, fused
            stg4(y + row * 64 + 16 * ob + 4 * g, v);
        }
    }
}

struct FfnBwdBufs {
    float *dz, *d1, *dh, *xn, *g1, *dxn;      // [M,64] [M,256] [M,256] [M,64] [M,64] [M,64]
};

__global__ __launch_bounds__(256) void ffn_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            long M, FfnTrainImg w, const unsigned char* __restrict__ m1,
                                                            const unsigned char* __restrict__ m2, float ms,
                                                            const float* __restrict__ dres, float* __restrict__ dx,
                                                            FfnBwdBufs o) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = ffn_load_norm(x, M, t0, c, g, w, xh, xn, rstd, row);
    f32x4 dz[1][4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x4 v = ldg4(dy + row * 64 + 16 * ob + 4 * g) * splat4(0.5f);
        if (m2) v = v * mask4(m2, row * 64 + 16 * ob + 4 * g, ms);
        if (!ok) v = splat4(0.f);                       // padding tokens of the last block contribute nothing
        dz[0][ob] = v;
        if (ok) {
            stg4(o.dz + row * 64 + 16 * ob + 4 * g, v);
            stg4(o.xn + row * 64 + 16 * ob + 4 * g, xn[0][ob]);
        }
    }
    f32x4 dxn[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) dxn[kb] = splat4(0.f);
    // three groups of 16 MFMAs per hidden block (W1 recompute, W2^T, W1^T); the next group's fragments are in flight
    // while one runs
    unsigned mw = m1 ? mask_word(m1, row * 256 + 4 * g) : 0u;
    f32x4 b1v = ldg4(w.b1 + 4 * g);
    FfnFrag g1 = ffn_frag_rows(w.w1, 0, lane), g2, g3;
    for (int hb = 0; hb < 16; ++hb) {
        g2 = ffn_frag_rows(w.w2t, hb, lane);
        const f32x4 h = ffn_frag_mma(g1, xn[0], b1v);
        g3 = ffn_frag_cols(w.w1t, hb, lane);
        const f32x4 dd1 = ffn_frag_mma(g2, dz[0], splat4(0.f));
        const f32x4 mk = m1 ? mask4w(mw, ms) : splat4(1.f);
        const int hn = hb < 15 ? hb + 1 : 15;
        if (m1) mw = mask_word(m1, row * 256 + 16 * hn + 4 * g);
        b1v = ldg4(w.b1 + 16 * hn + 4 * g);
        g1 = ffn_frag_rows(w.w1, hn, lane);
        f32x4 d1v, dhv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float hv = h[r], sg = sigmoidf_fast(hv);
            d1v[r] = hv * sg * mk[r];
            dhv[r] = dd1[r] * mk[r] * (sg * (1.f + hv * (1.f - sg)));
        }
        if (ok) {
            stg4(o.d1 + row * 256 + 16 * hb + 4 * g, d1v);
            stg4(o.dh + row * 256 + 16 * hb + 4 * g, dhv);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dxn[kb] = mfma16(g3.a[kb][r], dhv[r], dxn[kb]);
        }
    }
    // LayerNorm backward (per token: the 64 channels live in the 4 lanes c, c+16, c+32, c+48)
    f32x4 dxh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        dxh[kb] = dxn[kb] * ldg4(w.gamma + 16 * kb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += dxh[kb][r];
            s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
        }
    }
    const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
    if (ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
            if (dres) dv = dv + ldg4(dres + row * 64 + 16 * kb + 4 * g);   // + the gradient of the residual path
            stg4(dx + row * 64 + 16 * kb + 4 * g, dv);
        }
    }
    f32x4 ca[4], cb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        ca[kb] = ok ? dxn[kb] * xh[kb] : splat4(0.f);
        cb[kb] = ok ? dxn[kb] : splat4(0.f);
    }
    ln_tile_colsums(ca, cb, c, g, t0 >> 4, o.g1, o.dxn);
}

// out_partial[s][i][j] = sum over the s-th token range of P[m][i] * Q[m][j]   (P [M,R], Q [M,C], row-major), a 64 x 64
// output tile per block (all four 16-row blocks of a 64-row band): a first version with 16 x 64 tiles
// re-read Q once per row block and P once per column block - 664 MB of L2 / HBM reads for the 166 MB of
// operands of one FeedForward weight gradient, which is what bounded it (83 us at 8 TB/s of cache traffic).  Here a
// wave-step loads 16 + 16 operand dwords for 64 MFMAs; the four waves' tiles are combined through 32 KB of LDS in a
// fixed order.  Grid (R / 64, C / 64, WG_SPLIT).
#define WG_SPLIT 256                  // largest split (sizes the slab buffers); a launch uses wg_split(tiles) <= WG_SPLIT
__global__ __launch_bounds__(256) void wgrad_partial64_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                              long M, int R, int C, float* __restrict__ partial) {
    __shared__ float red[2][64 * 64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ic = blockIdx.x, jc = blockIdx.y, s = blockIdx.z;
    const int nsplit = gridDim.z;
    const long full = M / 16, per = (full + nsplit - 1) / nsplit;            // full 16-row steps, dealt to the splits
    const long st0 = (long)s * per, st1 = st0 + per < full ? st0 + per : full;
    f32x4 acc[4][4];                                  // [ib][jb]
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = splat4(0.f);
    // two steps per trip: the 32 operand dwords of the other step are in flight while the 64 MFMAs of one run (a wave
    // that waited for its own loads before every step spent more time on HBM / L2 latency than on products)
    unsigned oa[4], ob[4];                            // lane offsets of row 4g + r inside a step (P: + 16 ib, Q: + 16 jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        oa[r] = (unsigned)((4 * g + r) * R + 64 * ic + c);
        ob[r] = (unsigned)((4 * g + r) * C + 64 * jc + c);
    }
    auto load = [&](long st, f32x4 (&a)[4], f32x4 (&b)[4]) {       // [ib][r], [jb][r]; a full step (16 rows < M)
        const float* __restrict__ Pp = P + st * 16 * R;             // uniform
        const float* __restrict__ Qp = Q + st * 16 * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* __restrict__ pa = Pp + oa[r];              // + 16 ib / + 16 jb are instruction immediates
            const float* __restrict__ qa = Qp + ob[r];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) a[ib][r] = pa[16 * ib];
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) b[jb][r] = qa[16 * jb];
        }
    };
    auto mma = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma16(a[ib][r], b[jb][r], acc[ib][jb]);
    };
    f32x4 a0[4], b0[4], a1[4], b1[4];
    long st = st0 + wv;
    if (st < st1) load(st, a0, b0);
    while (st < st1) {
        const long sn = st + 4;
        if (sn < st1) load(sn, a1, b1);
        mma(a0, b0);
        st = sn + 4;
        if (st < st1) load(st, a0, b0);
        if (sn < st1) mma(a1, b1);
    }
    if ((M & 15) && s == nsplit - 1 && wv == 0) {   // the one ragged step of the tensor: rows past M contribute zeros
        const float* __restrict__ Pp = P + full * 16 * R;
        const float* __restrict__ Qp = Q + full * 16 * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = full * 16 + 4 * g + r < M;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) a0[ib][r] = ok ? Pp[oa[r] + 16 * ib] : 0.f;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) b0[jb][r] = ok ? Qp[ob[r] + 16 * jb] : 0.f;
        }
        mma(a0, b0);
    }
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
    }
}
// the kernel holds two steps of operands next to its 64 x 64 accumulators: 184 registers, two waves per SIMD.  The
// split is chosen so that one launch is at most those 2048 waves (tiles x split x 4), i.e. a single round
static int wg_split(int tiles) { return tiles >= 3 ? 128 : 256; }

// column sums of X [M,C] over NB fixed row ranges -> partial [NB][C]; four independent accumulators per thread keep
// four loads in flight (the loop is latency-bound otherwise), combined in a fixed order
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, long M, int C,
                                                             float* __restrict__ partial) {
    __shared__ float red[256];
    const int col = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;     // C divides 256
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long m = m0 + sub;
    for (; m + 3L * nsub < m1; m += 4L * nsub) {
        s0 += X[m * C + col];
        s1 += X[(m + nsub) * C + col];
        s2 += X[(m + 2L * nsub) * C + col];
        s3 += X[(m + 3L * nsub) * C + col];
    }
    for (; m < m1; m += nsub) s0 += X[m * C + col];
    float s = (s0 + s1) + (s2 + s3);
    red[threadIdx.x] = s;
    __syncthreads();
    if (sub == 0) {
        for (int k = 1; k < nsub; ++k) s += red[k * C + col];
        partial[(long)blockIdx.x * C + col] = s;
    }
}

// out[e] = sum_s partial[s][e]: a block owns 64 consecutive outputs at a time, its blockDim / 64 thread groups each
// add every G-th slab (four loads in flight), the groups are combined in group order - a fixed order for a given
// launch shape, coalesced 256-byte reads.  Final column sums (n <= 256) are launched as <<<4, 1024>>>.
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nsplit, long n, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6, G = blockDim.x >> 6;
    for (long chunk = blockIdx.x; chunk * 64 < n; chunk += gridDim.x) {
        const long e = chunk * 64 + col;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < n) {
            int k = grp;
            for (; k + 3 * G < nsplit; k += 4 * G) {
                s0 += partial[(long)k * n + e];
                s1 += partial[(long)(k + G) * n + e];
                s2 += partial[(long)(k + 2 * G) * n + e];
                s3 += partial[(long)(k + 3 * G) * n + e];
            }
            for (; k < nsplit; k += G) s0 += partial[(long)k * n + e];
        }
        red[grp][col] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (grp == 0 && e < n) {
            float s = red[0][col];
            for (int g = 1; g < G; ++g) s += red[g][col];
            out[e] = s;
        }
        __syncthreads();
    }
}

// Several column sums over the SAME M rows in two launches instead of two per sum (a conformer block's backward has
// twenty of them): blockIdx.y selects the job, each job has its own 512-slab region of `cpart`.
#define COLSUM_MAX_JOBS 4
struct ColsumJobs {
    const float* X[COLSUM_MAX_JOBS];
    float* out[COLSUM_MAX_JOBS];
    int C[COLSUM_MAX_JOBS];
    long rows[COLSUM_MAX_JOBS];     // 0: the launch's M; else this job's own row count (per-tile partial-sum slabs)
};
__global__ __launch_bounds__(256) void colsum_multi_partial_kernel(ColsumJobs jobs, long M, float* __restrict__ cpart) {
    __shared__ float red[256];
    const int job = blockIdx.y;
    const float* __restrict__ X = jobs.X[job];
    const int C = jobs.C[job];
    if (jobs.rows[job] > 0) M = jobs.rows[job];
    float* partial = cpart + (size_t)job * FFN_COLSUM_BLOCKS * 256;
    const int col = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long m = m0 + sub;
    for (; m + 3L * nsub < m1; m += 4L * nsub) {
        s0 += X[m * C + col];
        s1 += X[(m + nsub) * C + col];
        s2 += X[(m + 2L * nsub) * C + col];
        s3 += X[(m + 3L * nsub) * C + col];
    }
    for (; m < m1; m += nsub) s0 += X[m * C + col];
    float sum = (s0 + s1) + (s2 + s3);
    red[threadIdx.x] = sum;
    __syncthreads();
    if (sub == 0) {
        for (int k = 1; k < nsub; ++k) sum += red[k * C + col];
        partial[(long)blockIdx.x * C + col] = sum;
    }
}
__global__ __launch_bounds__(1024) void colsum_multi_reduce_kernel(ColsumJobs jobs, const float* __restrict__ cpart) {
    __shared__ float red[16][64];
    const int job = blockIdx.y, n = jobs.C[job];
    const float* partial = cpart + (size_t)job * FFN_COLSUM_BLOCKS * 256;
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + col;
    if ((long)blockIdx.x * 64 >= n) return;                               // block-uniform
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int k = grp;
        for (; k + 48 < FFN_COLSUM_BLOCKS; k += 64) {
            s0 += partial[(long)k * n + e];
            s1 += partial[(long)(k + 16) * n + e];
            s2 += partial[(long)(k + 32) * n + e];
            s3 += partial[(long)(k + 48) * n + e];
        }
        for (; k < FFN_COLSUM_BLOCKS; k += 16) s0 += partial[(long)k * n + e];
    }
    red[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && e < n) {
        float sum = red[0][col];
        for (int g = 1; g < 16; ++g) sum += red[g][col];
        jobs.out[job][e] = sum;
    }
}
// cpart must hold njobs * FFN_COLSUM_BLOCKS * 256 floats
static void colsum_batch(LaunchCtx ctx, const char* label, const ColsumJobs& jobs, int njobs, long M, float* cpart) {
    LAUNCH(ctx, label, (colsum_multi_partial_kernel<<<dim3(FFN_COLSUM_BLOCKS, njobs), 256, 0, ctx.stream>>>(jobs, M, cpart)));
    LAUNCH(ctx, label, (colsum_multi_reduce_kernel<<<dim3(4, njobs), 1024, 0, ctx.stream>>>(jobs, cpart)));
}

#ifndef TRAIN_X3
#define TRAIN_X3 1          // 1: FeedForward forward / backward and the token-contraction weight gradients on split-f16
#endif                      //    products (train_x3.hip); 0: the fp32-MFMA kernels of this file
// train_x3.hip
void ffn_x3_pack(LaunchCtx, const FfnTrainParams& p, float* img);
void ffn_x3_forward(LaunchCtx, const float* x, long M, const FfnTrainParams& p, const float* img, const unsigned char* m1,
                    const unsigned char* m2, float ms, const float* res, float* y);
int ffn_x3_backward_fused(LaunchCtx, const float* x, const float* dy, long M, const FfnTrainParams& p, const float* img,
                          const unsigned char* m1, const unsigned char* m2, float ms, const float* dres, float* dx,
                          float* o_dh, float* o_g1, float* o_dxn, float* dhmax, float* o_dzc, float* o_dhc, float* part_w2,
                          float* part_w1);
void ffn_x3_backward(LaunchCtx, const float* x, const float* dy, long M, const FfnTrainParams& p, const float* img,
                     const unsigned char* m1, const unsigned char* m2, float ms, const float* dres, float* dx, float* o_dz,
                     float* o_d1, float* o_dh, float* o_xn, float* o_g1, float* o_dxn, float* dhmax, float* o_dzc, float* o_dhc);
void launch_wgrad_partial64_x3(LaunchCtx, const char* label, const float* P, const float* Q, long M, int R, int C,
                               float* partial, int nsplit, float* colp);
void launch_db_conv_wgrad_x3(LaunchCtx, const float* dz, const float* a, int B, int T, int F, int dil, int nsplit,
                             float* partial, float* colp);
void cm_x3_pack(LaunchCtx, const ConvModTrainParams& p, float* img_w1, float* img_w1t);
void cm_x3_pw1glu(LaunchCtx, const float* x, long M, const float* img_w1, const ConvModTrainParams& p, float* u);
void cm_x3_bwd2(LaunchCtx, const float* x, const float* du, long M, const float* img_w1, const float* img_w1t,
                const ConvModTrainParams& p, const float* dres, float* dx, float* dag, float* xn_out, float* g1c, float* dxc);
void cm_x3_pack_pw2(LaunchCtx, const ConvModTrainParams& p, float* img_w2, float* img_w2t);
void cm_x3_bn_swish_pw2(LaunchCtx, const float* d, long M, const float* scale, const float* shift, const float* img_w2,
                        const float* b2, const float* res, float* y);
int cm_x3_bwd2_fused(LaunchCtx, const float* x, const float* du, long M, const float* img_w1t, const ConvModTrainParams& p,
                     const float* dres, float* dx, float* dag, float* o_g1, float* o_dxn, float* o_dhc, float* dhmax,
                     float* part_w1);
int cm_x3_bwd1_fused(LaunchCtx, const float* dy, const float* d, long M, const float* mean, const float* rstd,
                     const float* scale, const float* shift, const float* w2raw, float* ddn, float* g2c, float* ddnc, float* dyc,
                     float* part_w2);
void cm_x3_bwd1(LaunchCtx, const float* dy, const float* d, long M, const float* mean, const float* rstd, const float* scale,
                const float* shift, const float* img_w2t, float* ddn, float* s_out, float* g2c, float* ddnc, float* dyc);
void at_x3_pack(LaunchCtx, const float* wraw, float* img_w, float* img_wt);
void at_x3_qkv(LaunchCtx, const float* x, long M, const float* img_w, const float* ln_w, const float* ln_b, float* qkv,
               int as_image, float qscale);
void at_x3_qkv_bwd(LaunchCtx, const float* x, const float* dqkv, long M, const float* img_wt, const float* ln_w,
                   const float* ln_b, const float* dres, float* dx, float* xn_out, float* g1c, float* dxc);
// the token-contraction weight gradient in either mode: grid (R / 64, C / 64, nsplit)
// colp / colsum_out (split-f16 build only): also the column sums of P -> colsum_out[R], through [nsplit][R] partials at colp
static void wgrad_partial64(LaunchCtx ctx, const char* label, const float* P, const float* Q, long M, int R, int C,
                            float* partial, int nsplit, float* colp = nullptr, float* colsum_out = nullptr,
                            const char* reduce_label = nullptr) {
#if TRAIN_X3
    launch_wgrad_partial64_x3(ctx, label, P, Q, M, R, C, partial, nsplit, colp);
    if (colp) LAUNCH(ctx, reduce_label, (reduce_partials_kernel<<<4, 1024, 0, ctx.stream>>>(colp, nsplit, R, colsum_out)));
#else
    (void)colp; (void)colsum_out; (void)reduce_label;
    LAUNCH(ctx, label, (wgrad_partial64_kernel<<<dim3(R / 64, C / 64, nsplit), 256, 0, ctx.stream>>>(P, Q, M, R, C, partial)));
#endif
}

// pack = false: the images the forward of the SAME parameters left in the workspace are reused (backward passes)
static FfnTrainImg ffn_pack_images(LaunchCtx ctx, const FfnTrainParams& p, float* img, bool pack = true) {
    hipStream_t s = ctx.stream;
    float *w1 = img, *w2 = img + 16384, *w2t = img + 2 * 16384, *w1t = img + 3 * 16384;
    if (!pack) return FfnTrainImg{w1, w2, w2t, w1t, p.gamma, p.beta, p.b1, p.b2};
    launch_pack4(ctx, "ffn_train_pack", PackJobs{{{p.w1, 256, 64, 64, 0, w1},       // rows = hidden
                                                  {p.w2, 64, 256, 256, 0, w2},      // rows = out
                                                  {p.w2, 256, 64, 256, 1, w2t},     // rows = hidden
                                                  {p.w1, 64, 256, 64, 1, w1t}}});   // rows = in
    return FfnTrainImg{w1, w2, w2t, w1t, p.gamma, p.beta, p.b1, p.b2};
}

// The fused backward (train_x3.hip, the default) keeps dh [M,256] and per-tile partial rows only: its workspace is COMPACT -
// 270 floats per token instead of the 768 of dz | d1 | dh | xn | g1 | dxn (3.2 GB -> 1.1 GB per FeedForward at 32 clips,
// sixteen of them in the generator).  Compact <=> this build, CMGAN_FFN_BWD_FUSED != 0 and a dh tensor under 2 GB; a call the
// fused kernel cannot take then (one keep-mask without the other) is an ERROR, not a fallback into buffers that are not there.
static bool ffn_ws_compact(long M) {
    static const bool k_fused = env_knob("CMGAN_FFN_BWD_FUSED", 1, 0, 1) != 0;
    return TRAIN_X3 && k_fused && M * 1024 < (1l << 31);
}
size_t ffn_train_ws_floats(long M) {
    const size_t tiles = (size_t)((M + 31) / 32);
    const size_t act = ffn_ws_compact(M) ? (size_t)M * 256 + tiles * 448 : (size_t)M * 768;
    return (size_t)4 * 16384 + act + (size_t)WG_SPLIT * 16384 * 2 + (size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256;
}

void launch_ffn_train_forward(LaunchCtx ctx, const float* x, long M, const FfnTrainParams& p, const unsigned char* m1,
                              const unsigned char* m2, float ms, const float* res, float* y, float* ws) {
#if TRAIN_X3
    ffn_x3_pack(ctx, p, ws);
    ffn_x3_forward(ctx, x, M, p, ws, m1, m2, ms, res, y);
#else
    const FfnTrainImg w = ffn_pack_images(ctx, p, ws);
    const unsigned grid = (unsigned)((M + 63) / 64);
    LAUNCH(ctx, "ffn_train_fwd", (ffn_train_fwd_kernel<<<grid, 256, 0, ctx.stream>>>(x, M, w, m1, m2, ms, res, y)));
#endif
}

bool launch_ffn_train_backward(LaunchCtx ctx, const float* x, const float* dy, long M, const FfnTrainParams& p,
                               const unsigned char* m1, const unsigned char* m2, float ms, const float* dres, float* dx,
                               const FfnTrainParams& grad, float* ws) {
    hipStream_t s = ctx.stream;
    const FfnTrainImg w = ffn_pack_images(ctx, p, ws, false);
    float* act = ws + 4 * 16384;
    const bool compact = ffn_ws_compact(M);
    const long xrows = (M + 31) / 32;
    // compact: dh | g1 rows | dz sums | dxn rows | dh sums (ffn_train_ws_floats); else dz | d1 | dh | xn | g1 | dxn
    FfnBwdBufs o = compact ? FfnBwdBufs{nullptr, nullptr, act, nullptr, act + M * 256, act + M * 256 + xrows * 128}
                           : FfnBwdBufs{act, act + M * 64, act + M * 320, act + M * 576, act + M * 640, act + M * 704};
    float* part = act + (compact ? M * 256 + xrows * 448 : M * 768);      // [SPLIT][16384] x 2, then colsum slabs
    float* cpart = part + (size_t)WG_SPLIT * 16384 * 2;
#if TRAIN_X3
    (void)w;
    // per-tile partial sums for the four column sums: [tiles][64] dgamma | dz (db2) in the g1 region, [tiles][64] dbeta |
    // [tiles][256] dh (db1) in the dxn region (the full [M,64] g1 / dxn tensors of round 2 are no longer written)
    float *dzc = o.g1 + xrows * 64, *dhc = o.dxn + xrows * 64;   // dzc: [2 tiles][64] (fused form: [tiles][64])
    // default: part A with both weight gradients contracted on the chip (train_x3.hip); CMGAN_FFN_BWD_FUSED=0: A/B
    static const bool k_fused = env_knob("CMGAN_FFN_BWD_FUSED", 1, 0, 1) != 0;
    const int fused_slabs = !k_fused ? 0
        : ffn_x3_backward_fused(ctx, x, dy, M, p, ws, m1, m2, ms, dres, dx, o.dh, o.g1, o.dxn, cpart, dzc, dhc, part,
                                part + (size_t)WG_SPLIT * 16384);
    if (!fused_slabs && compact) return false;                    // (one mask without the other / LDS refused: see ffn_ws_compact)
    if (!fused_slabs)
        ffn_x3_backward(ctx, x, dy, M, p, ws, m1, m2, ms, dres, dx, o.dz, o.d1, o.dh, o.xn, o.g1, o.dxn,
                        cpart,                                    // per-tile |dh| maxima: the column-sum slabs are free until colsum_batch
                        dzc, dhc);
#else
    const int fused_slabs = 0;
    const unsigned grid = (unsigned)((M + 63) / 64);
    LAUNCH(ctx, "ffn_train_bwd", (ffn_train_bwd_kernel<<<grid, 256, 0, s>>>(x, dy, M, w, m1, m2, ms, dres, dx, o)));
#endif
    // dW2 [64,256] = dz^T d1 ; dW1 [256,64] = dh^T xn
    if (!fused_slabs) {
        wgrad_partial64(ctx, "ffn_train_wgrad", o.dz, o.d1, M, 64, 256, part, wg_split(4));
        wgrad_partial64(ctx, "ffn_train_wgrad", o.dh, o.xn, M, 256, 64, part + (size_t)WG_SPLIT * 16384, wg_split(4));
    }
    const int nslab = fused_slabs ? fused_slabs : wg_split(4);
    LAUNCH(ctx, "ffn_train_reduce", (reduce_partials_kernel<<<256, 1024, 0, s>>>(part, nslab, 16384, grad.w2)));
    LAUNCH(ctx, "ffn_train_reduce", (reduce_partials_kernel<<<256, 1024, 0, s>>>(part + (size_t)WG_SPLIT * 16384, nslab, 16384,
                                                                                 grad.w1)));
    // o.g1 / o.dxn hold per-tile partial sums (ln_tile_colsums): one row per 32-token tile (x3) / 16-token tile (fp32)
    const long trows = TRAIN_X3 ? (M + 31) / 32 : (M + 15) / 16;
#if TRAIN_X3
    const ColsumJobs jobs{{dhc, dzc, o.g1, o.dxn}, {grad.b1, grad.b2, grad.gamma, grad.beta}, {256, 64, 64, 64},
                          {trows, fused_slabs ? trows : 2 * trows, trows, trows}};      // dz sums: per 16-token block (per tile: fused)
#else
    const ColsumJobs jobs{{o.dh, o.dz, o.g1, o.dxn}, {grad.b1, grad.b2, grad.gamma, grad.beta}, {256, 64, 64, 64},
                          {0, 0, trows, trows}};
#endif
    colsum_batch(ctx, "ffn_train_reduce", jobs, 4, M, cpart);
    return true;
}

// ---------------------------------------------------------------------------------
// torch.optim.AdamW (src/train.py:63-66: AdamW(parameters, lr), defaults betas (0.9, 0.999), eps 1e-8,
// weight_decay 0.01) as ONE launch over a flat parameter bucket - parameters, gradients and both moment buffers are
// flat fp32 arrays (the same bucket the gradient all-reduce runs over), so the update is a single coalesced pass:
//   p <- p (1 - lr wd);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g^2;
//   p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (the operation order of torch's single-tensor implementation, so results agree to rounding).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1,
                                                    float rsqrt_bc2) {
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

// the same update with the step counter and the learning rate in device memory (state[0] = lr, state[1] = update count
// as a float), so that the launch can be replayed from a captured hipGraph: a one-thread kernel advances the count,
// the update kernel derives the bias corrections from it
__global__ void adamw_tick_kernel(float* __restrict__ state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1.0f;
}
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const float* __restrict__ state, float b1, float b2, float eps,
                                                        float wd) {
    const float lr = state[0];
    const double t = (double)state[1];
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        pi -= step_size * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}
void launch_adamw_dev(LaunchCtx ctx, float* p, const float* g, float* m, float* v, long n, float* state, float b1, float b2,
                      float eps, float wd) {
    const long want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? (want > 0 ? want : 1) : 2048);
    LAUNCH(ctx, "adamw", (adamw_tick_kernel<<<1, 64, 0, ctx.stream>>>(state)));
    LAUNCH(ctx, "adamw", (adamw_dev_kernel<<<grid, 256, 0, ctx.stream>>>(p, g, m, v, n, state, b1, b2, eps, wd)));
}
void launch_adamw(LaunchCtx ctx, float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                  float eps, float wd, int step) {
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const long want = (n + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? (want > 0 ? want : 1) : 2048);
    LAUNCH(ctx, "adamw", (adamw_kernel<<<grid, 256, 0, ctx.stream>>>(p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1,
                                                                     (float)(1.0 / sqrt(bc2)))));
}

// =====================================================================================
// Training-mode ConformerConvModule (second backward slice of SURVEY.md N2; conformer.py:151-176):
//   LayerNorm -> Conv1d(64,256,1) -> GLU -> DepthWiseConv1d(k=31, 'same') -> BatchNorm1d(128, BATCH statistics)
//   -> Swish -> Conv1d(128,64,1)            (its Dropout has p = conv_dropout = 0)
// on contiguous sequences x [N, L, 64] (token m = n L + l).  The pointwise stages are the per-token fp32-MFMA chain of
// the FeedForward slice; the depthwise conv (and its data-gradient, the same kernel with flipped taps) is a 16-output
// sliding window per thread; BatchNorm statistics and every parameter gradient are fixed-order two-pass reductions.
// The forward leaves u (GLU output), d (depthwise output) and the batch statistics in the workspace for the backward.
// =====================================================================================
struct CmImg {
    const float *w1, *w1t, *w2, *w2t;       // fm: pw1 [16][4], pw1^T [4][16], pw2 [4][8], pw2^T [8][4]
};
struct CmStats { float *mean, *rstd, *scale, *shift; };   // [128] each: batch mean, 1/sqrt(var+eps), gamma rstd, beta - mean gamma rstd

__device__ __forceinline__ bool cm_load_norm(const float* __restrict__ x, long M, long t0, int c, int g,
                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                             f32x4 (&xh)[4], f32x4 (&xn)[1][4], float& rstd, long& row) {
    const long t = t0 + c;
    const bool ok = t < M;
    row = ok ? t : M - 1;
    f32x4 xv[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) xv[kb] = ldg4(x + row * 64 + 16 * kb + 4 * g);
    float mean;
    ln_stats(xv, mean, rstd);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        xh[kb] = (xv[kb] - splat4(mean)) * splat4(rstd);
        xn[0][kb] = xh[kb] * ldg4(gamma + 16 * kb + 4 * g) + ldg4(beta + 16 * kb + 4 * g);
    }
    return ok;
}

// LN -> pointwise 64 -> 256 -> GLU:  u [M,128]
__global__ __launch_bounds__(256) void cm_pw1glu_kernel(const float* __restrict__ x, long M, const float* __restrict__ w1fm,
                                                        ConvModTrainParams p, float* __restrict__ u) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, p.ln_w, p.ln_b, xh, xn, rstd, row);
    // value half, then gate half of an output block; the other half's weight fragments are in flight meanwhile
    f32x4 ba = ldg4(p.pw1_b + 4 * g), bg = ldg4(p.pw1_b + 128 + 4 * g);
    FfnFrag fa = ffn_frag_rows(w1fm, 0, lane), fg;
    for (int ob = 0; ob < 8; ++ob) {
        fg = ffn_frag_rows(w1fm, ob + 8, lane);
        const f32x4 a = ffn_frag_mma(fa, xn[0], ba);
        const int on = ob < 7 ? ob + 1 : 7;
        ba = ldg4(p.pw1_b + 16 * on + 4 * g);
        fa = ffn_frag_rows(w1fm, on, lane);
        const f32x4 gt = ffn_frag_mma(fg, xn[0], bg);
        bg = ldg4(p.pw1_b + 128 + 16 * on + 4 * g);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = a[e] * sigmoidf_fast(gt[e]);
        if (ok) stg4(u + row * 128 + 16 * ob + 4 * g, r);
    }
}

// Staging of one 32-position tile of a [N, L, 128] tensor for the depthwise kernels: the 62 rows l0 - 15 .. l0 + 46
// (zero outside [0, L)) as 16-byte loads - 8 per thread, coalesced 512-byte rows - held in registers one tile ahead and
// dropped into LDS when the previous tile has been consumed.  (Round 2 had every thread fetch its 46 window values as
// scalar loads, 2.9 x halo re-read, one tile in flight per block: 56 / 174 us per launch at batch 4 against a
// bandwidth floor of ~27 us.)
#define CMD_ROWS 62
struct CmWin { f32x4 r[8]; };
__device__ __forceinline__ void cm_win_load(const float* __restrict__ src, long n, int L, int l0, int tid, CmWin& w) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + 256 * k;
        const int rr = i < CMD_ROWS * 32 ? (i >> 5) : CMD_ROWS - 1;
        const int l = l0 - 15 + rr, lc = l < 0 ? 0 : (l < L ? l : L - 1);
        f32x4 v = ldg4(src + (n * L + lc) * 128 + (i & 31) * 4);
        if (l < 0 || l >= L) v = splat4(0.f);
        w.r[k] = v;
    }
}
__device__ __forceinline__ void cm_win_store(float* win, int tid, const CmWin& w) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = tid + 256 * k;
        if (i < CMD_ROWS * 32) *reinterpret_cast<f32x4*>(win + (i >> 5) * 128 + (i & 31) * 4) = w.r[k];
    }
}

// out[(n,l)][ch] = bias[ch] + sum_t taps[ch][flip ? 30 - t : t] * in[(n, l + t - 15)][ch], zero outside [0, L).
// flip = 0: the forward depthwise conv; flip = 1 (bias = NULL): its gradient w.r.t. the input.
// A block walks `tpb` consecutive (sequence, 32-position) tiles; thread = (channel, 16-output half tile).
// stats (forward only): per block and channel (sum, sum of squares) of the outputs -> partial [blk][128][2].
__global__ __launch_bounds__(256) void cm_depthwise_kernel(const float* __restrict__ in, const float* __restrict__ taps,
                                                           const float* __restrict__ bias, int flip, int L, int ntile_l,
                                                           long ntile, int tpb, float* __restrict__ out,
                                                           float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float win[CMD_ROWS * 128];
    __shared__ float tl[128 * 31];
    const int tid = threadIdx.x, ch = tid & 127, sub = tid >> 7;
    const long t_begin = (long)blockIdx.x * tpb, t_end = t_begin + tpb < ntile ? t_begin + tpb : ntile;
    for (int i = tid; i < 128 * 31; i += 256) tl[i] = taps[i];
    CmWin pre;
    if (t_begin < t_end) {
        const long n = t_begin / ntile_l;
        cm_win_load(in, n, L, (int)(t_begin - n * ntile_l) * 32, tid, pre);
    }
    __syncthreads();
    float w[31];
#pragma unroll
    for (int t = 0; t < 31; ++t) w[t] = tl[ch * 31 + (flip ? 30 - t : t)];
    const float b = bias ? bias[ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const float* ucol = win + sub * 16 * 128 + ch;
    for (long tile = t_begin; tile < t_end; ++tile) {
        const long n = tile / ntile_l;
        const int l0 = (int)(tile - n * ntile_l) * 32 + sub * 16;
        cm_win_store(win, tid, pre);
        __syncthreads();
        if (tile + 1 < t_end) {
            const long n1 = (tile + 1) / ntile_l;
            cm_win_load(in, n1, L, (int)(tile + 1 - n1 * ntile_l) * 32, tid, pre);
        }
        if (l0 < L) {
            float acc[16];
#pragma unroll
            for (int oo = 0; oo < 16; ++oo) acc[oo] = b;
#pragma unroll
            for (int kk = 0; kk < 46; ++kk) {
                const float v = ucol[kk * 128];
#pragma unroll
                for (int oo = 0; oo < 16; ++oo) {
                    const int t = kk - oo;
                    if (t >= 0 && t < 31) acc[oo] = fmaf(w[t], v, acc[oo]);
                }
            }
#pragma unroll
            for (int oo = 0; oo < 16; ++oo) {
                const int l = l0 + oo;
                if (l < L) {
                    out[(n * L + l) * 128 + ch] = acc[oo];
                    s1 += acc[oo];
                    s2 = fmaf(acc[oo], acc[oo], s2);
                }
            }
        }
        __syncthreads();                                          // the window is free for the next tile
    }
    if (stats) {
        float* red = win;                                         // [2][128][2]
        red[(sub * 128 + ch) * 2 + 0] = s1;
        red[(sub * 128 + ch) * 2 + 1] = s2;
        __syncthreads();
        if (sub == 0) {
            stats[((long)blockIdx.x * 128 + ch) * 2 + 0] = red[ch * 2] + red[(128 + ch) * 2];
            stats[((long)blockIdx.x * 128 + ch) * 2 + 1] = red[ch * 2 + 1] + red[(128 + ch) * 2 + 1];
        }
    }
}

// batch statistics of BatchNorm1d(128) in train mode (biased variance, eps 1e-5) from the per-block partial sums, in
// fp64: 64 blocks each add every 64-th slab of 256 floats (channel, {sum, sum of squares}), the finalize adds the 64
// partials in order; running statistics updated like torch (momentum 0.1, unbiased variance)   conformer.py:168
#define CM_BN_RED 64
__global__ __launch_bounds__(256) void cm_bn_reduce_kernel(const float* __restrict__ stats, long nblk,
                                                           double* __restrict__ part) {
    double a0 = 0.0, a1 = 0.0;
    long k = blockIdx.x;
    for (; k + CM_BN_RED < nblk; k += 2 * CM_BN_RED) {
        a0 += (double)stats[k * 256 + threadIdx.x];
        a1 += (double)stats[(k + CM_BN_RED) * 256 + threadIdx.x];
    }
    if (k < nblk) a0 += (double)stats[k * 256 + threadIdx.x];
    part[(long)blockIdx.x * 256 + threadIdx.x] = a0 + a1;
}
__global__ void cm_bn_finalize_kernel(const double* __restrict__ part, double count,
                                      const float* __restrict__ bn_w, const float* __restrict__ bn_b, CmStats st,
                                      float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int ch = threadIdx.x;
    if (ch >= 128) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < CM_BN_RED; ++k) {
        s1 += part[(long)k * 256 + ch * 2];
        s2 += part[(long)k * 256 + ch * 2 + 1];
    }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + 1e-5);
    st.mean[ch] = (float)mean;
    st.rstd[ch] = (float)rstd;
    st.scale[ch] = (float)((double)bn_w[ch] * rstd);
    st.shift[ch] = (float)((double)bn_b[ch] - mean * (double)bn_w[ch] * rstd);
    if (running_mean && running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[ch] = (float)(0.9 * (double)running_mean[ch] + 0.1 * mean);
        running_var[ch] = (float)(0.9 * (double)running_var[ch] + 0.1 * unbiased);
    }
}

// BatchNorm apply -> Swish -> pointwise 128 -> 64 + bias
__global__ __launch_bounds__(256) void cm_bn_swish_pw2_kernel(const float* __restrict__ d, long M, CmStats st,
                                                              const float* __restrict__ w2fm,
                                                              const float* __restrict__ b2, const float* __restrict__ res,
                                                              float* __restrict__ y) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    const long t = t0 + c;
    const bool ok = t < M;
    const long row = ok ? t : M - 1;
    f32x4 s[1][8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const f32x4 dn = ldg4(d + row * 128 + 16 * kb + 4 * g) * ldg4(st.scale + 16 * kb + 4 * g) +
                         ldg4(st.shift + 16 * kb + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[0][kb][e] = swishf(dn[e]);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x4 acc[1] = {ldg4(b2 + 16 * ob + 4 * g)};
        lin_acc<8, 1>(w2fm + (long)ob * 8 * 256 + lane * 4, s, acc);
        if (res) acc[0] = acc[0] + ldg4(res + row * 64 + 16 * ob + 4 * g);
        if (ok) stg4(y + row * 64 + 16 * ob + 4 * g, acc[0]);
    }
}

// backward, part 1 (per token): ds = pw2^T dy, through Swish; writes ddn = dL/d(bn output), s (for dW_pw2) and the
// per-tile column sums the three reductions of this stage are finished from (ln_tile_colsums' scheme): rows of
// [tiles][128] for g2 = ddn * dhat (BatchNorm dgamma) and ddn (dbeta), [tiles][64] for dy (the pointwise bias)
__global__ __launch_bounds__(256) void cm_bwd1_kernel(const float* __restrict__ dy, const float* __restrict__ d, long M,
                                                      CmStats st, const float* __restrict__ w2tfm,
                                                      float* __restrict__ ddn, float* __restrict__ s_out,
                                                      float* __restrict__ g2c, float* __restrict__ ddnc,
                                                      float* __restrict__ dyc) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    const long t = t0 + c;
    const bool ok = t < M;
    const long row = ok ? t : M - 1;
    f32x4 dyf[1][4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) dyf[0][ob] = ldg4(dy + row * 64 + 16 * ob + 4 * g);
    const long tile = t0 >> 4;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x4 v = ok ? dyf[0][ob] : splat4(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = red_c_sum(v[r]);
        if (c == 0) stg4(dyc + tile * 64 + 16 * ob + 4 * g, v);
    }
#pragma unroll 2
    for (int hb = 0; hb < 8; ++hb) {
        f32x4 ds[1] = {splat4(0.f)};
        lin_acc<4, 1>(w2tfm + (long)hb * 4 * 256 + lane * 4, dyf, ds);
        const f32x4 dv = ldg4(d + row * 128 + 16 * hb + 4 * g);
        const f32x4 dhat = (dv - ldg4(st.mean + 16 * hb + 4 * g)) * ldg4(st.rstd + 16 * hb + 4 * g);
        const f32x4 dn = dv * ldg4(st.scale + 16 * hb + 4 * g) + ldg4(st.shift + 16 * hb + 4 * g);
        f32x4 o_ddn, o_s;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = sigmoidf_fast(dn[e]);
            o_s[e] = dn[e] * sg;
            o_ddn[e] = ds[0][e] * (sg * (1.f + dn[e] * (1.f - sg)));
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
            stg4(g2c + tile * 128 + 16 * hb + 4 * g, ca);
            stg4(ddnc + tile * 128 + 16 * hb + 4 * g, cb);
        }
    }
}

// BatchNorm backward (batch statistics): dd = gamma rstd (ddn - mean(ddn) - dhat mean(ddn dhat)), in place on ddn
// bpart (optional): [gridDim.x][128] per-block sums of dd - the depthwise BIAS gradient's partials - for free in this HBM-bound
// pass: a thread's channel is fixed (the stride is a multiple of 128), two threads per channel are combined through LDS.
// (A separate column-sum pass re-read the [M,128] plane.)
__global__ __launch_bounds__(256) void cm_bn_bwd_kernel(float* __restrict__ ddn, const float* __restrict__ d, long total,
                                                        CmStats st, const float* __restrict__ sum_ddn,
                                                        const float* __restrict__ sum_g2, float inv_count,
                                                        float* __restrict__ bpart) {
    __shared__ float red[128];
    const int ch = threadIdx.x & 127;                             // = i & 127 for every i of this thread
    const float mu = st.mean[ch], rs = st.rstd[ch], sc = st.scale[ch];
    const float c1 = sum_ddn[ch] * inv_count, c2 = sum_g2[ch] * inv_count;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float dhat = (d[i] - mu) * rs;
        const float v = sc * (ddn[i] - c1 - dhat * c2);
        ddn[i] = v;
        acc += v;
    }
    if (bpart) {
        if (threadIdx.x >= 128) red[ch] = acc;
        __syncthreads();
        if (threadIdx.x < 128) bpart[(long)blockIdx.x * 128 + ch] = acc + red[ch];
    }
}

// depthwise weight gradient: dw[ch][t] = sum_{n,l} dd[(n,l)][ch] u[(n, l + t - 15)][ch] -> partial [blk][128*31].
// At most CM_DW_SLABS blocks each walk `tpb` consecutive (sequence, 32-token) tiles and keep their 31 taps in registers
// across tiles, so the second-stage reduction reads a few hundred slabs instead of one per tile; the u window and the
// dd rows of a tile are staged through LDS one tile ahead (cm_win_load).
#define CM_DW_SLABS 768
__global__ __launch_bounds__(256) void cm_dw_wgrad_kernel(const float* __restrict__ dd, const float* __restrict__ u, int L,
                                                          int ntile_l, long ntile, int tpb, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float win[CMD_ROWS * 128];
    __shared__ __attribute__((aligned(16))) float dt[32 * 128];
    const int tid = threadIdx.x, ch = tid & 127, sub = tid >> 7;
    const long t_begin = (long)blockIdx.x * tpb, t_end = t_begin + tpb < ntile ? t_begin + tpb : ntile;
    float acc[31];
#pragma unroll
    for (int t = 0; t < 31; ++t) acc[t] = 0.f;
    CmWin pre;
    f32x4 pd[4];
    auto load_dd = [&](long tile) {
        const long n = tile / ntile_l;
        const int l0 = (int)(tile - n * ntile_l) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k, l = l0 + (i >> 5), lc = l < L ? l : L - 1;
            f32x4 v = ldg4(dd + (n * L + lc) * 128 + (i & 31) * 4);
            if (l >= L) v = splat4(0.f);
            pd[k] = v;
        }
        cm_win_load(u, n, L, l0, tid, pre);
    };
    if (t_begin < t_end) load_dd(t_begin);
    const float* ucol = win + sub * 16 * 128 + ch;
    const float* dcol = dt + sub * 16 * 128 + ch;
    for (long tile = t_begin; tile < t_end; ++tile) {
        cm_win_store(win, tid, pre);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + 256 * k;
            *reinterpret_cast<f32x4*>(dt + (i >> 5) * 128 + (i & 31) * 4) = pd[k];
        }
        __syncthreads();
        if (tile + 1 < t_end) load_dd(tile + 1);
        // tap-major loops with compile-time indices only: the output-major form with `acc[kk - oo]` left the 46-step
        // loop rolled and indexed registers dynamically
        float g[16], uw[46];
#pragma unroll
        for (int oo = 0; oo < 16; ++oo) g[oo] = dcol[oo * 128];
#pragma unroll
        for (int kk = 0; kk < 46; ++kk) uw[kk] = ucol[kk * 128];
#pragma unroll
        for (int t = 0; t < 31; ++t) {
            float a = acc[t];
#pragma unroll
            for (int oo = 0; oo < 16; ++oo) a = fmaf(g[oo], uw[oo + t], a);
            acc[t] = a;
        }
        __syncthreads();
    }
    float* red = win;                                             // [128 * 31] of the 62 x 128 window
    if (sub == 1) {
#pragma unroll
        for (int t = 0; t < 31; ++t) red[ch * 31 + t] = acc[t];
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
        for (int t = 0; t < 31; ++t) partial[(long)blockIdx.x * 3968 + ch * 31 + t] = acc[t] + red[ch * 31 + t];
    }
}

// backward, part 2 (per token): GLU backward with a, g recomputed from x, dxn = pw1^T [da; dg], LayerNorm backward
__global__ __launch_bounds__(256) void cm_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ du, long M,
                                                      const float* __restrict__ w1fm, const float* __restrict__ w1tfm,
                                                      ConvModTrainParams p, const float* __restrict__ dres,
                                                      float* __restrict__ dx, float* __restrict__ dag,
                                                      float* __restrict__ xn_out, float* __restrict__ g1,
                                                      float* __restrict__ dxn_out) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, p.ln_w, p.ln_b, xh, xn, rstd, row);
    f32x4 dxn[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) dxn[kb] = splat4(0.f);
    // four groups of 16 MFMAs per output block (value, gate, and their two transposed products); the fragments of a
    // group are fetched while the previous groups run
    f32x4 ba = ldg4(p.pw1_b + 4 * g), bg = ldg4(p.pw1_b + 128 + 4 * g), dun = ldg4(du + row * 128 + 4 * g);
    FfnFrag fa = ffn_frag_rows(w1fm, 0, lane), fg, ta, tg;
    for (int ob = 0; ob < 8; ++ob) {
        fg = ffn_frag_rows(w1fm, ob + 8, lane);
        const f32x4 a = ffn_frag_mma(fa, xn[0], ba);
        ta = ffn_frag_cols(w1tfm, ob, lane);
        const f32x4 gt = ffn_frag_mma(fg, xn[0], bg);
        tg = ffn_frag_cols(w1tfm, ob + 8, lane);
        f32x4 duv = dun;
        if (!ok) duv = splat4(0.f);
        const int on = ob < 7 ? ob + 1 : 7;
        ba = ldg4(p.pw1_b + 16 * on + 4 * g);
        bg = ldg4(p.pw1_b + 128 + 16 * on + 4 * g);
        dun = ldg4(du + row * 128 + 16 * on + 4 * g);
        f32x4 da, dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sg = sigmoidf_fast(gt[e]);
            da[e] = duv[e] * sg;
            dg[e] = duv[e] * a[e] * sg * (1.f - sg);
        }
        if (ok) {
            stg4(dag + row * 256 + 16 * ob + 4 * g, da);
            stg4(dag + row * 256 + 128 + 16 * ob + 4 * g, dg);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) dxn[kb] = mfma16(ta.a[kb][r], da[r], dxn[kb]);
        fa = ffn_frag_rows(w1fm, on, lane);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) dxn[kb] = mfma16(tg.a[kb][r], dg[r], dxn[kb]);
    }
    f32x4 dxh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        dxh[kb] = dxn[kb] * ldg4(p.ln_w + 16 * kb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += dxh[kb][r];
            s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
        }
    }
    const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
    if (ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
            if (dres) dv = dv + ldg4(dres + row * 64 + 16 * kb + 4 * g);
            stg4(dx + row * 64 + 16 * kb + 4 * g, dv);
            stg4(xn_out + row * 64 + 16 * kb + 4 * g, xn[0][kb]);
        }
    }
    f32x4 ca[4], cb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        ca[kb] = ok ? dxn[kb] * xh[kb] : splat4(0.f);
        cb[kb] = ok ? dxn[kb] : splat4(0.f);
    }
    ln_tile_colsums(ca, cb, c, g, t0 >> 4, g1, dxn_out);            // [tiles][64] slabs: dgamma / dbeta partial sums
}

// tiles per block of the depthwise kernels: at most `cap` blocks (three 48 KB blocks fit a CU: 768 per round)
#define CM_DW_BLOCKS 1536
static int cm_dw_tpb(long ntile, int cap) { return (int)((ntile + cap - 1) / cap); }

// The fused backward parts (train_x3.hip, the default) never write s = Swish(BN(d)) [M,128] (part 1) nor xn [M,64] (part 2): the
// workspace leaves them out.  Compact <=> this build, the part's knob != 0 and a tensor size its 32-bit offsets cover; a call
// the fused part cannot take then is an ERROR, not a fallback into buffers that are not there.
static bool cm_ws_compact1(long M) {
    static const bool k = env_knob("CMGAN_CM_BWD1_FUSED", 1, 0, 1) != 0;
    return TRAIN_X3 && k && M * 512 < (1l << 31);
}
static bool cm_ws_compact2(long M) {
    static const bool k = env_knob("CMGAN_CM_BWD2_FUSED", 1, 0, 1) != 0;
    return TRAIN_X3 && k && M * 1024 < (1l << 31);
}
// workspace layout (floats): images | u | d | stats (4 x 128) | bwd buffers | partials
struct CmPlan {
    size_t img, u, d, st, ddn, s, g2, du, dag, xn, g1, dxn, wpart, dwpart, bnpart, bnred, cpart, sums, total;
};
static CmPlan cm_plan(int N, int L) {
    CmPlan p;
    const size_t M = (size_t)N * L, nblk = (size_t)N * ((L + 31) / 32);
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    p.img = take(16384 * 2 + 8192 * 2);
    p.u = take(M * 128); p.d = take(M * 128); p.st = take(512);
    // g2 / g1 / dxn hold per-TILE partial rows only ([tiles16][128] x 3; [tiles16][64]; [tiles16][64] + [tiles32][256]); s and xn
    // exist only for the un-fused backward parts (cm_ws_compact1 / 2: the fused parts - the default - never write them)
    const size_t t16 = (M + 15) / 16, t32 = (M + 31) / 32;
    p.ddn = take(M * 128); p.s = take(cm_ws_compact1((long)M) ? 0 : M * 128); p.g2 = take(3 * t16 * 128); p.du = take(M * 128);
    p.dag = take(M * 256); p.xn = take(cm_ws_compact2((long)M) ? 0 : M * 64); p.g1 = take(t16 * 64); p.dxn = take(t16 * 64 + t32 * 256);
    p.wpart = take((size_t)WG_SPLIT * 16384);
    p.dwpart = take((size_t)CM_DW_SLABS * 3968);
    p.bnpart = take(nblk * 256);
    p.bnred = take((size_t)CM_BN_RED * 256 * 2);        // doubles
    p.cpart = take((size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256);
    p.sums = take(256);
    p.total = cur;
    return p;
}
size_t convmod_train_ws_floats(int N, int L) { return cm_plan(N, L).total; }

static CmImg cm_pack_images(LaunchCtx ctx, const ConvModTrainParams& p, float* img, bool pack = true) {
    hipStream_t s = ctx.stream;
    float *w1 = img, *w1t = img + 16384, *w2 = img + 32768, *w2t = img + 32768 + 8192;
    if (!pack) return CmImg{w1, w1t, w2, w2t};
#if TRAIN_X3
    // every per-token stage of the module runs on split products: split-f16 images in the four slots
    cm_x3_pack(ctx, p, w1, w1t);
    cm_x3_pack_pw2(ctx, p, w2, w2t);
#else
    launch_pack4(ctx, "convmod_train_pack", PackJobs{{{p.pw1_w, 256, 64, 64, 0, w1}, {p.pw1_w, 64, 256, 64, 1, w1t},
                                                      {p.pw2_w, 64, 128, 128, 0, w2}, {p.pw2_w, 128, 64, 128, 1, w2t}}});
#endif
    return CmImg{w1, w1t, w2, w2t};
}

void launch_convmod_train_forward(LaunchCtx ctx, const float* x, int N, int L, const ConvModTrainParams& p,
                                  float* running_mean, float* running_var, const float* res, float* y, float* ws) {
    hipStream_t s = ctx.stream;
    const CmPlan pl = cm_plan(N, L);
    const long M = (long)N * L;
    const CmImg im = cm_pack_images(ctx, p, ws + pl.img);
    const CmStats st{ws + pl.st, ws + pl.st + 128, ws + pl.st + 256, ws + pl.st + 384};
    const unsigned grid = (unsigned)((M + 63) / 64);
    const dim3 dgrid(N, (L + 31) / 32);
#if TRAIN_X3
    cm_x3_pw1glu(ctx, x, M, im.w1, p, ws + pl.u);
#else
    LAUNCH(ctx, "convmod_train_fwd", (cm_pw1glu_kernel<<<grid, 256, 0, s>>>(x, M, im.w1, p, ws + pl.u)));
#endif
    const long ntile = (long)dgrid.x * dgrid.y;
    const int tpb = cm_dw_tpb(ntile, CM_DW_BLOCKS), dblocks = (int)((ntile + tpb - 1) / tpb);
    LAUNCH(ctx, "convmod_train_fwd", (cm_depthwise_kernel<<<dblocks, 256, 0, s>>>(ws + pl.u, p.dw_w, p.dw_b, 0, L, (int)dgrid.y, ntile,
                                                                                  tpb, ws + pl.d, ws + pl.bnpart)));
    double* bnred = (double*)(ws + pl.bnred);
    LAUNCH(ctx, "convmod_train_fwd", (cm_bn_reduce_kernel<<<CM_BN_RED, 256, 0, s>>>(ws + pl.bnpart, (long)dblocks, bnred)));
    LAUNCH(ctx, "convmod_train_fwd", (cm_bn_finalize_kernel<<<1, 128, 0, s>>>(bnred, (double)M, p.bn_w, p.bn_b, st,
                                                                              running_mean, running_var)));
#if TRAIN_X3
    cm_x3_bn_swish_pw2(ctx, ws + pl.d, M, st.scale, st.shift, im.w2, p.pw2_b, res, y);
#else
    LAUNCH(ctx, "convmod_train_fwd", (cm_bn_swish_pw2_kernel<<<grid, 256, 0, s>>>(ws + pl.d, M, st, im.w2, p.pw2_b, res, y)));
#endif
}

bool launch_convmod_train_backward(LaunchCtx ctx, const float* x, const float* dy, int N, int L,
                                   const ConvModTrainParams& p, const float* dres, float* dx,
                                   const ConvModTrainParams& grad, float* ws) {
    hipStream_t s = ctx.stream;
    const CmPlan pl = cm_plan(N, L);
    const long M = (long)N * L;
    const CmImg im = cm_pack_images(ctx, p, ws + pl.img, false);   // the forward's images (same parameters)
    const CmStats st{ws + pl.st, ws + pl.st + 128, ws + pl.st + 256, ws + pl.st + 384};
    const unsigned grid = (unsigned)((M + 63) / 64);
    const dim3 dgrid(N, (L + 31) / 32);
    const long nblk = (long)dgrid.x * dgrid.y;
    float* cpart = ws + pl.cpart;
    long trows = (M + 15) / 16;                                   // per-tile partial sums inside the g2 region [M,128]
    float *g2c = ws + pl.g2, *ddnc = g2c + trows * 128, *dyc = ddnc + trows * 128;
    int ns_pw2 = 0;                                               // > 0: dW_pw2's slabs already written (fused part 1)
#if TRAIN_X3
    // default: part 1 with the pointwise-2 weight gradient contracted on the chip (train_x3.hip; s never leaves it; its
    // partial-sum rows are per 32-token tile); CMGAN_CM_BWD1_FUSED=0: A/B
    static const bool k_fused1 = env_knob("CMGAN_CM_BWD1_FUSED", 1, 0, 1) != 0;
    if (k_fused1)
        ns_pw2 = cm_x3_bwd1_fused(ctx, dy, ws + pl.d, M, st.mean, st.rstd, st.scale, st.shift, p.pw2_w, ws + pl.ddn, g2c, ddnc,
                                  dyc, ws + pl.wpart);
    if (ns_pw2) trows = (M + 31) / 32;
    else if (cm_ws_compact1(M)) return false;                     // (the device refused the fused kernel's LDS)
    else cm_x3_bwd1(ctx, dy, ws + pl.d, M, st.mean, st.rstd, st.scale, st.shift, im.w2t, ws + pl.ddn, ws + pl.s, g2c, ddnc, dyc);
#else
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd1_kernel<<<grid, 256, 0, s>>>(dy, ws + pl.d, M, st, im.w2t, ws + pl.ddn,
                                                                          ws + pl.s, g2c, ddnc, dyc)));
#endif
    // pointwise-2 gradients: dW_pw2 [64,128] = dy^T s, db_pw2 = colsum dy
    if (!ns_pw2) {
        wgrad_partial64(ctx, "convmod_train_wgrad", dy, ws + pl.s, M, 64, 128, ws + pl.wpart, wg_split(2));
        ns_pw2 = wg_split(2);
    }
    LAUNCH(ctx, "convmod_train_reduce", (reduce_partials_kernel<<<128, 1024, 0, s>>>(ws + pl.wpart, ns_pw2, 8192,
                                                                                   grad.pw2_w)));
    // BatchNorm: dbeta = sum ddn, dgamma = sum ddn dhat; then dd in place   (+ db_pw2 = colsum dy in the same pair of launches)
    {
        // (sum ddn = dbeta and sum ddn dhat = dgamma land in the gradient tensors themselves; the BatchNorm backward reads them there)
        const ColsumJobs jobs{{dyc, ddnc, g2c}, {grad.pw2_b, grad.bn_b, grad.bn_w}, {64, 128, 128}, {trows, trows, trows}};
        colsum_batch(ctx, "convmod_train_reduce", jobs, 3, M, cpart);
    }
    LAUNCH(ctx, "convmod_train_bwd", (cm_bn_bwd_kernel<<<2048, 256, 0, s>>>(ws + pl.ddn, ws + pl.d, M * 128, st, grad.bn_b,
                                                                            grad.bn_w, (float)(1.0 / (double)M), cpart)));
    LAUNCH(ctx, "convmod_train_reduce", (reduce_partials_kernel<<<4, 1024, 0, s>>>(cpart, 2048, 128, grad.dw_b)));
    float* dd = ws + pl.ddn;
    // depthwise: weight gradient (the bias gradient rode on the BatchNorm backward above), then the data gradient (same kernel,
    // flipped taps)
    const int wtpb = cm_dw_tpb(nblk, CM_DW_SLABS), nslab = (int)((nblk + wtpb - 1) / wtpb);
    LAUNCH(ctx, "convmod_train_wgrad", (cm_dw_wgrad_kernel<<<nslab, 256, 0, s>>>(dd, ws + pl.u, L, (int)dgrid.y, nblk, wtpb,
                                                                                 ws + pl.dwpart)));
    LAUNCH(ctx, "convmod_train_reduce", (reduce_partials_kernel<<<62, 1024, 0, s>>>(ws + pl.dwpart, nslab, 3968, grad.dw_w)));
    {
        const int tpb = cm_dw_tpb(nblk, CM_DW_BLOCKS), dblocks = (int)((nblk + tpb - 1) / tpb);
        LAUNCH(ctx, "convmod_train_bwd", (cm_depthwise_kernel<<<dblocks, 256, 0, s>>>(dd, p.dw_w, nullptr, 1, L, (int)dgrid.y, nblk, tpb,
                                                                                      ws + pl.du, nullptr)));
    }
#if TRAIN_X3
    // default: part 2 split like the FeedForward's backward - LN / GLU backward / [da ; dg] + dW_pw1 on the chip, then the
    // FeedForward's own part B on pw1's image (train_x3.hip); CMGAN_CM_BWD2_FUSED=0: A/B
    static const bool k_fused2 = env_knob("CMGAN_CM_BWD2_FUSED", 1, 0, 1) != 0;
    const long t32 = (M + 31) / 32;
    float* dhc = ws + pl.dxn + t32 * 64;                          // [tiles][256] partials of db_pw1 behind the [tiles][64] dxn rows
    const int ns_pw1 = !k_fused2 ? 0
        : cm_x3_bwd2_fused(ctx, x, ws + pl.du, M, im.w1t, p, dres, dx, ws + pl.dag, ws + pl.g1, ws + pl.dxn, dhc,
                           cpart + (size_t)3 * FFN_COLSUM_BLOCKS * 256, ws + pl.wpart);
    if (ns_pw1) {
        LAUNCH(ctx, "convmod_train_reduce", (reduce_partials_kernel<<<256, 1024, 0, s>>>(ws + pl.wpart, ns_pw1, 16384, grad.pw1_w)));
        const ColsumJobs jobs{{dhc, ws + pl.g1, ws + pl.dxn}, {grad.pw1_b, grad.ln_w, grad.ln_b}, {256, 64, 64}, {t32, t32, t32}};
        colsum_batch(ctx, "convmod_train_reduce", jobs, 3, M, cpart);
        return true;
    }
    if (cm_ws_compact2(M)) return false;
    cm_x3_bwd2(ctx, x, ws + pl.du, M, im.w1, im.w1t, p, dres, dx, ws + pl.dag, ws + pl.xn, ws + pl.g1, ws + pl.dxn);
#else
    LAUNCH(ctx, "convmod_train_bwd", (cm_bwd2_kernel<<<grid, 256, 0, s>>>(x, ws + pl.du, M, im.w1, im.w1t, p, dres, dx,
                                                                          ws + pl.dag, ws + pl.xn, ws + pl.g1,
                                                                          ws + pl.dxn)));
#endif
    // pointwise-1 and LayerNorm gradients
#if TRAIN_X3
    // db_pw1 = colsum [da ; dg] comes out of the weight-gradient kernel, which holds every element of dag anyway (the column-sum
    // pass re-read the [M,256] tensor); its [split][256] partials use the fourth job's region of cpart
    wgrad_partial64(ctx, "convmod_train_wgrad", ws + pl.dag, ws + pl.xn, M, 256, 64, ws + pl.wpart, wg_split(4),
                    cpart + (size_t)3 * FFN_COLSUM_BLOCKS * 256, grad.pw1_b, "convmod_train_reduce");
#else
    wgrad_partial64(ctx, "convmod_train_wgrad", ws + pl.dag, ws + pl.xn, M, 256, 64, ws + pl.wpart, wg_split(4));
#endif
    LAUNCH(ctx, "convmod_train_reduce", (reduce_partials_kernel<<<256, 1024, 0, s>>>(ws + pl.wpart, wg_split(4), 16384,
                                                                                   grad.pw1_w)));
#if TRAIN_X3
    const ColsumJobs jobs{{ws + pl.g1, ws + pl.dxn}, {grad.ln_w, grad.ln_b}, {64, 64},
                          {(M + 15) / 16, (M + 15) / 16}};                        // g1 / dxn: per-tile partial sums
    colsum_batch(ctx, "convmod_train_reduce", jobs, 2, M, cpart);
#else
    const ColsumJobs jobs{{ws + pl.dag, ws + pl.g1, ws + pl.dxn}, {grad.pw1_b, grad.ln_w, grad.ln_b}, {256, 64, 64},
                          {0, (M + 15) / 16, (M + 15) / 16}};                     // g1 / dxn: per-tile partial sums
    colsum_batch(ctx, "convmod_train_reduce", jobs, 3, M, cpart);
#endif
    return true;
}

// =====================================================================================
// Training-mode PreNorm(Attention) (third backward slice of SURVEY.md N2; conformer.py:54-72, 75-133):
//   LayerNorm -> to_q / to_kv (no bias) -> per head softmax((q k^T + q E[clamp(i - j)]^T) / 4) v -> to_out + bias
//   -> Dropout (keep-mask on the [M,64] output)
// on contiguous sequences x [N, L, 64], L <= AT_MAX_L.  The projections are the per-token fp32-MFMA chain; the
// attention core runs on the fp32 matrix pipe, one wave per (sequence, head, 16-row block) task (see below), each
// backward core recomputing the probabilities from q, k, E and the saved row log-sum-exp, so no task ever accumulates
// into another task's output and every sum has a fixed order.
// =====================================================================================
#define AT_MAX_L 4096                 // a workspace-size bound only: the cores have no length-dependent resource

#ifndef AT_X3
#define AT_X3 1             // 1: the attention cores on split-f16 products fed from PRE-SPLIT operand images (below);
#endif                      // 0: exact fp32 products (v_mfma_f32_16x16x4_f32), operands straight from the fp32 tensors
// AT_X3 operand images.  Every tensor the cores read (q | k | v, dO, the relative-position window) is stored split into
// fp16 hi / lo halves by its PRODUCER: the 4 bytes of element i hold (hi_i, lo_i), so that a core's operand is its
// load plus four v_perm_b32 (at_row_a) instead of a 6-VALU split per float4 and product (round 3's first split-f16
// form split inside the cores, which took back what the matrix pipe gave - VALU and MFMA issue add up on this
// machine).  ONE interleaving serves both fragment types (an A-type float4 = 16 bytes of one row, a row-type fragment
// = four dwords down a column): a second, MFMA-ready [hi x 4 | lo x 4] image for the A-type loads saved the perms but
// doubled the bytes a (sequence, head) keeps in L2, which cost far more (at_bwd_fused_kernel).
// q is stored PRE-SCALED by dim_head^-0.5 log2(e) (the factor every score product applies; dk and dE, which contract
// with q, are multiplied by ln 2 instead of 0.25 at the end); dO is stored pre-scaled by the exact power of two of
// at_scale.  fp32 build: both "images" are the fp32 tensor itself.
struct AtBufs {
    float *qkv;      // [M,192]  q | k | v  (features 16 h + d inside each 64): fp32, or the (hi, lo) image
    float *qkvp;     //          = qkv (a second interleaving lived here)
    float *o;        // [M,64]   softmax(.) v, heads concatenated
    float *lse;      // [N,4,L]  row log-sum-exp of the scaled scores
};
// Sequences of up to ATF_MAX_NB blocks run the fused backward core (at_bwd_fused_kernel, below): 44 product steps per
// tile instead of the 76 of the three cores.  CMGAN_ATTN_BWD=cores in the environment forces the three cores (which
// longer sequences always use); both paths are held to the same gradients by the tests.  Per step at batch 32
// (rocprofv3): time axis 32.7 -> 30.0 ms, frequency axis 14.3 -> 13.9 ms.  Its first form was SLOWER (63 vs 54 ms):
// one 7-wave block per CU, and operands read from both image interleavings - 164 KB per (sequence, head), 5.2 MB per
// XCD against a 4 MB L2, 67 % TCC misses (PMC) - see the load lambda.
#define AT_LOG2E 1.4426950408889634f  // scores are kept in log2 units: p = v_exp_f32(s - lse) without a multiply
#define AT_QSCALE (0.25f * AT_LOG2E)  // dim_head^-0.5 * log2(e), folded into the q fragment of every score product
// one float4 -> its image form (hi0, lo0, hi1, lo1, hi2, lo2, hi3, lo3)
__device__ __forceinline__ f32x4 at_img4(const f32x4& v) {
    f16x4 h, l;
    split4(v, h, l);
    return __builtin_bit_cast(f32x4, __builtin_shufflevector(h, l, 0, 4, 1, 5, 2, 6, 3, 7));
}

// fp32 build: the projection as is.  AT_X3: both operand images, the q part (output blocks 0 .. 3) pre-scaled
__device__ __forceinline__ void at_store_qkv(float* __restrict__ qkv, float* __restrict__ qkvp, long off, f32x4 v, bool is_q,
                                             bool ok) {
#if AT_X3
    (void)qkvp;
    if (is_q) v = v * splat4(AT_QSCALE);
    if (ok) stg4(qkv + off, at_img4(v));
#else
    (void)qkvp; (void)is_q;
    if (ok) stg4(qkv + off, v);
#endif
}
// LN -> [to_q ; to_kv] (A image [12][4]) -> qkv
__global__ __launch_bounds__(256) void at_qkv_kernel(const float* __restrict__ x, long M, const float* __restrict__ wfm,
                                                     const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                     float* __restrict__ qkv, float* __restrict__ qkvp) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, ln_w, ln_b, xh, xn, rstd, row);
    FfnFrag f0 = ffn_frag_rows(wfm, 0, lane), f1;     // the next output block's fragments in flight
    for (int ob = 0; ob < 12; ob += 2) {
        f1 = ffn_frag_rows(wfm, ob + 1, lane);
        const f32x4 q0 = ffn_frag_mma(f0, xn[0], splat4(0.f));
        at_store_qkv(qkv, qkvp, row * 192 + 16 * ob + 4 * g, q0, ob < 4, ok);
        f0 = ffn_frag_rows(wfm, ob + 2 < 12 ? ob + 2 : 11, lane);
        const f32x4 q1 = ffn_frag_mma(f1, xn[0], splat4(0.f));
        at_store_qkv(qkv, qkvp, row * 192 + 16 * (ob + 1) + 4 * g, q1, ob + 1 < 4, ok);
    }
}

// ---------------------------------------------------------------------------------
// Attention cores on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32), one WAVE per task, no block barriers:
//   forward  task (n, h, query block)  : S^T = K q^T, E q^T -> skew -> online softmax down the key blocks -> o^T += V^T P^T
//   dq       task (n, h, query block)  : recompute P^T, dS^T;  dq += dS K + unskew(dS) E_band
//   dk / dv  task (n, h, key block)    : recompute P, dS;      dk += dS^T q, dv += P^T dO
//   dE       task (n, h, tile diagonal): recompute dS;         dE_band += unskew(dS)^T q   (all tiles of a diagonal
//                                        share their 31 distances, so the band stays in registers)
// A 16 x 16 tile of (query, key) pairs costs 4 MFMAs for q.k, 8 for the relative-position term (the 31 distances of
// the tile, as q . E_band^T) and 4 for dO.v; the band result is turned into the tile ("skew": R[i][j] =
// QE[i][15 + i - j]) through a wave-private LDS patch, and dS goes back the same way for the two E-side products.
// Operand fragments (common.hip.h convention, lane = (g, c)):
//   "A-type" XA[s] = X[row0 + c][4g + s]   one float4 per lane: at_dot(XA, YA)[r] = sum_d X[4g + r][d] Y[c][d]
//   "row-type" XB[r] = X[row0 + 4g + r][c] : B operand of a product that contracts over the 16 rows of the tile
// and an accumulator f32x4 holds D[4g + r][c].  Choosing which of the two operands of at_dot is the key side gives
// the tile (S) or its transpose (S^T) without any data movement, and the accumulator layout of one product is the
// A-operand layout of the next: dS^T feeds dq, dS feeds dk, P^T feeds o^T, P feeds dv straight from registers.
// Every sum has a fixed order (a task owns its outputs; dE partial slabs are reduced in (n, h) order afterwards).
// ---------------------------------------------------------------------------------
#define AT_PA 20                      // LDS pitch of the [32][16] band patch (E q^T)
#define AT_PB 36                      // LDS pitch of the [16][32] band patch (q E^T): conflict-free writes, <= 2-way reads
#define AT_PS 48                      // dS patch row: [16 zeros | 16 keys | 16 zeros] - out-of-tile reads of the unskew are 0

struct AtTask { int nh, blk; };       // wave-uniform (SGPRs)
// blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of tasks so that the waves sharing one
// (n, h)'s q / k / v / dO rows sit behind the same L2
__device__ __forceinline__ bool at_task(long ntask, int nb, AtTask& t) {
    const long per = (gridDim.x + 7) / 8;
    const long blk = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const long task = blk * 4 + (threadIdx.x >> 6);
    if (task >= ntask) return false;
    const int nh = (int)(task / nb);
    t.nh = __builtin_amdgcn_readfirstlane(nh);
    t.blk = __builtin_amdgcn_readfirstlane((int)(task - (long)nh * nb));
    return true;
}
// ---- operand types of the cores ------------------------------------------------------------------------------------
// at_mma(a, b, acc) = acc + sum over the 16 contraction indices (g, s) of a(lane (i, g))[s] * b(lane (j, g))[s].
// fp32 build: operands are the lanes' float4s, four v_mfma_f32_16x16x4_f32 steps.  AT_X3: the A side is the lane's
// [hi(4) | lo(4)] along the 32-wide contraction of v_mfma_f32_16x16x32_f16, the B side [hi | hi], then [lo | lo]: two
// MFMAs give all four split terms.  Loaded operands arrive in these forms from the images (at_lda / at_row_a, no VALU
// but register shuffles); operands computed in the core (probabilities, dS, unskewed bands) are split by at_a / at_b.
#if AT_X3
typedef f16x8 AtA;
struct AtB { f16x8 hh, ll; };
__device__ __forceinline__ AtA at_a(const f32x4& x) {
    f16x4 h, l;
    split4(x, h, l);
    return __builtin_shufflevector(h, l, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ AtB at_b_of(const AtA& a) {
    return AtB{__builtin_shufflevector(a, a, 0, 1, 2, 3, 0, 1, 2, 3), __builtin_shufflevector(a, a, 4, 5, 6, 7, 4, 5, 6, 7)};
}
__device__ __forceinline__ AtB at_b(const f32x4& x) { return at_b_of(at_a(x)); }
// four (hi, lo) dwords of the image - an A-type float4 (four consecutive features of a row) or a row-type fragment
// (four scalar loads down a column) - -> [hi(4) | lo(4)]
__device__ __forceinline__ AtA at_row_a(const f32x4& raw) {
    const unsigned d0 = __float_as_uint(raw[0]), d1 = __float_as_uint(raw[1]), d2 = __float_as_uint(raw[2]),
                   d3 = __float_as_uint(raw[3]);
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ v = {__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u),
                      __builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ AtA at_lda(const float* __restrict__ p) { return at_row_a(ldg4(p)); }
__device__ __forceinline__ f32x4 at_mma(const AtA& a, const AtB& b, f32x4 acc) {
    acc = mfma32h(a, b.hh, acc);
    return mfma32l(a, b.ll, acc);
}
#define AT_QBACK 0.6931471805599453f      // dk, dE contract with the PRE-SCALED q: 0.25 = (0.25 log2 e) ln 2
#else
typedef f32x4 AtA;
typedef f32x4 AtB;
__device__ __forceinline__ AtA at_a(const f32x4& x) { return x; }
__device__ __forceinline__ AtB at_b_of(const AtA& a) { return a; }
__device__ __forceinline__ AtB at_b(const f32x4& x) { return x; }
__device__ __forceinline__ AtA at_lda(const float* __restrict__ p) { return ldg4(p); }
__device__ __forceinline__ AtA at_row_a(const f32x4& raw) { return raw; }
__device__ __forceinline__ f32x4 at_mma(const AtA& a, const AtB& b, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma16(a[s], b[s], acc);
    return acc;
}
#define AT_QBACK 0.25f
#endif
__device__ __forceinline__ AtB at_row_b(const f32x4& raw) { return at_b_of(at_row_a(raw)); }
__device__ __forceinline__ f32x4 at_dot(const AtA& a, const AtB& b) { return at_mma(a, b, splat4(0.f)); }
// the q fragment of a score product (A-type rows of the q image / tensor): fp32 build applies the score scale here
__device__ __forceinline__ AtA at_ldq(const float* __restrict__ p) {
#if AT_X3
    return at_lda(p);
#else
    return ldg4(p) * splat4(AT_QSCALE);
#endif
}
// a q fragment that did not come through at_ldq (fp32 build: the score scale is still to be applied)
__device__ __forceinline__ AtA at_qfix(const AtA& q) {
#if AT_X3
    return q;
#else
    return q * splat4(AT_QSCALE);
#endif
}
// loads as uniform base + 32-bit per-lane BYTE offset: the global_load "saddr" form (no 64-bit VALU address arithmetic)
__device__ __forceinline__ f32x4 at_ld4b(const float* __restrict__ base, unsigned boff) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ float at_ld1b(const float* __restrict__ base, unsigned boff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
}
// exact power-of-two scale of the backward cores' gradient operands (dO and D = rowsum(dO o)): s brings the largest
// |dO| of the whole tensor (*amax: per-block maxima of at_out_bwd_kernel reduced by at_amax_kernel - order-independent)
// to [1, 2); the dO images are stored scaled, D is scaled on load, and every output of a core - linear in dO - is
// multiplied by inv = 1 / s at the end.  fp32 build: 1, 1.
__device__ __forceinline__ void at_scale(const float* __restrict__ amax, float& s, float& inv) {
    s = 1.0f; inv = 1.0f;
#if AT_X3
    const unsigned e = (__float_as_uint(amax[0]) >> 23) & 0xffu;
    if (e > 0u && e < 254u) {
        s = __uint_as_float((254u - e) << 23);
        inv = __uint_as_float(e << 23);
    }
#endif
}
// lane offsets (floats) of the fragments of the 16 rows that start at row R0 of a sequence of L rows, relative to row
// R0: rows past the end read row L - 1 (finite; masked or never stored).  Full blocks use the affine forms
// c * stride + 4g / (4g + r) * stride + c, which cost no VALU inside the loops (uniform row pointer + immediates).
__device__ __forceinline__ unsigned at_off_a(int R0, int L, int stride, int c, int g) {
    const int r = R0 + c < L ? c : L - 1 - R0;
    return (unsigned)(r * stride + 4 * g);
}
__device__ __forceinline__ unsigned at_off_b(int R0, int L, int stride, int c, int g, int r) {
    const int rr = R0 + 4 * g + r < L ? 4 * g + r : L - 1 - R0;
    return (unsigned)(rr * stride + c);
}
// relative-position window: row w <-> distance w - W (W = 16 nb + 16 covers every band a tile can ask for), table row
// clamp(distance, +-max_pos) + max_pos - the clamp of conformer.py:104 is applied once here, the cores index affinely
__global__ void at_window_kernel(const float* __restrict__ rel, int W, int max_pos, float* __restrict__ ewin,
                                 float* __restrict__ ewinp) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;             // one float4 (four features of one distance) per thread
    if (q >= (2 * W + 1) * 4) return;
    int dist = (q >> 2) - W;
    dist = dist < -max_pos ? -max_pos : (dist > max_pos ? max_pos : dist);
    const f32x4 v = ldg4(rel + (long)(dist + max_pos) * 16 + (q & 3) * 4);
#if AT_X3
    (void)ewinp;
    stg4(ewin + 4 * q, at_img4(v));
#else
    (void)ewinp;
    stg4(ewin + 4 * q, v);
#endif
}
// band^T [32 distances][16 queries] (two accumulators) -> R^T[key 4g + r][query c] = band[15 + c - (4g + r)][c]
__device__ __forceinline__ f32x4 at_skew_t(float* buf, const f32x4& eq0, const f32x4& eq1, int c, int g) {
    wave_lds_fence();                     // the previous tile's reads of the patch are done
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        buf[(4 * g + r) * AT_PA + c] = eq0[r];
        buf[(16 + 4 * g + r) * AT_PA + c] = eq1[r];
    }
    wave_lds_fence();
    f32x4 rt;
#pragma unroll
    for (int r = 0; r < 4; ++r) rt[r] = buf[(15 + c - 4 * g) * AT_PA + c - r * AT_PA];
    return rt;
}
// band [16 queries][32 distances] -> R[query 4g + r][key c] = band[4g + r][15 + 4g + r - c]
__device__ __forceinline__ f32x4 at_skew(float* buf, const f32x4& qe0, const f32x4& qe1, int c, int g) {
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        buf[(4 * g + r) * AT_PB + c] = qe0[r];
        buf[(4 * g + r) * AT_PB + 16 + c] = qe1[r];
    }
    wave_lds_fence();
    f32x4 rr;
#pragma unroll
    for (int r = 0; r < 4; ++r) rr[r] = buf[4 * g * (AT_PB + 1) + 15 - c + r * (AT_PB + 1)];
    return rr;
}
__device__ __forceinline__ void at_zero_pads(float* patch, int lane) {
    for (int i = lane; i < 16 * 32; i += 64) patch[(i >> 5) * AT_PS + ((i & 31) < 16 ? (i & 31) : 16 + (i & 31))] = 0.f;
}

// forward: o = softmax((q k^T + q E^T) / 4) v and the row log-sum-exp, online over the key blocks.  Everything a query
// owns (running max, denominator, its o^T column) lives in the lanes with c = its index: the rescale needs no transpose.
// The band product of a tile's upper 16 distances is the lower one of the previous key block: 4 new MFMAs per tile.
__global__ __launch_bounds__(256) void at_fwd_kernel(AtBufs b, const float* __restrict__ ewin, int L, int nb, long ntask) {
    __shared__ float sm[4][32 * AT_PA];
    AtTask t;
    if (!at_task(ntask, nb, t)) return;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* buf = sm[threadIdx.x >> 6];
    const int h = t.nh & 3, I0 = 16 * t.blk, W = 16 * nb + 16, nfull = L >> 4;
    const long base = (long)(t.nh >> 2) * L;
    const float* __restrict__ qh = b.qkv + base * 192 + 16 * h;                    // uniform: this head's q | k | v rows
    const float* __restrict__ qhp = b.qkvp + base * 192 + 16 * h;                  // the same rows of the pair image
    const float* __restrict__ e_blk = ewin + (long)(I0 - 15 + W) * 16;             // band of key block 0; block jb: - 256 jb
    const AtB qa = at_b_of(at_ldq(qh + (long)I0 * 192 + at_off_a(I0, L, 192, c, g)));
    const unsigned la = (c * 192 + 4 * g) * 4, lb = (4 * g * 192 + c) * 4, le = c * 16 + 4 * g;      // la, lb: BYTE offsets
    const unsigned la_t = at_off_a(16 * nfull, L, 192, c, g) * 4;
    unsigned lb_t[4];
    bool vt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lb_t[r] = at_off_b(16 * nfull, L, 192, c, g, r) * 4; vt[r] = 16 * nfull + 4 * g + r < L; }
    f32x4 ot = splat4(0.f);               // o^T[d = 4g + r][query c]
    float m = -1e30f, l = 0.f;
    f32x4 eq1 = at_dot(at_lda(e_blk + 256 + le), qa);
    struct Frag { AtA ka, ea; f32x4 vb; };
    auto load = [&](auto tail, int jb) {
        constexpr bool TAIL = decltype(tail)::value;
        const float* __restrict__ kp = qh + 64 + (long)jb * (16 * 192);
        const float* __restrict__ vp = qhp + 128 + (long)jb * (16 * 192);
        Frag f;
        f.ka = at_row_a(at_ld4b(kp, TAIL ? la_t : la));
#pragma unroll
        for (int r = 0; r < 4; ++r) f.vb[r] = TAIL ? at_ld1b(vp, lb_t[r]) : at_ld1b(vp + r * 192, lb);
        f.ea = at_row_a(at_ld4b(e_blk - (long)jb * 256, le * 4));
        return f;
    };
    auto tile = [&](auto tail, const Frag& f) {
        constexpr bool TAIL = decltype(tail)::value;
        const f32x4 eq0 = at_dot(f.ea, qa);
        const f32x4 st = at_dot(f.ka, qa);
        const f32x4 rt = at_skew_t(buf, eq0, eq1, c, g);
        eq1 = eq0;
        f32x4 sc;
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[r] = (!TAIL || vt[r]) ? st[r] + rt[r] : -1e30f;
        const float mx = red_g_max(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
        if (__builtin_amdgcn_ballot_w64(mx > m) != 0) {   // some query's running maximum moves (rare after the first blocks)
            const float mn = fmaxf(m, mx), corr = __builtin_amdgcn_exp2f(m - mn);
            l *= corr;
            ot = ot * splat4(corr);
            m = mn;
        }
        f32x4 p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(sc[r] - m);
        l += (p[0] + p[1]) + (p[2] + p[3]);               // this lane's four keys; the four lane groups are summed at the end
        ot = at_mma(at_row_a(f.vb), at_b(p), ot);
    };
    if (nfull > 0) {
        // two blocks per trip: the other block's operands are in flight while one is being worked on
        Frag fa = load(std::false_type{}, 0), fb;
        int jb = 0;
        for (; jb + 1 < nfull; jb += 2) {
            fb = load(std::false_type{}, jb + 1);
            tile(std::false_type{}, fa);
            fa = load(std::false_type{}, jb + 2 < nfull ? jb + 2 : jb + 1);
            tile(std::false_type{}, fb);
        }
        if (jb < nfull) tile(std::false_type{}, fa);
    }
    if (nfull < nb) tile(std::true_type{}, load(std::true_type{}, nfull));
    l = red_g_sum(l);
    if (I0 + c < L) {
        stg4(b.o + (base + I0 + c) * 64 + 16 * h + 4 * g, ot * splat4(__builtin_amdgcn_rcpf(l)));
        if (g == 0) b.lse[(long)t.nh * L + I0 + c] = (m + __log2f(l)) * 0.6931471805599453f;
    }
}

// y = mask * (Wo O + bo)
__global__ __launch_bounds__(256) void at_out_kernel(const float* __restrict__ o, long M, const float* __restrict__ wofm,
                                                     const float* __restrict__ bo, const unsigned char* __restrict__ mask, float ms,
                                                     const float* __restrict__ res, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    const long t = t0 + c;
    const bool ok = t < M;
    const long row = ok ? t : M - 1;
    f32x4 of[1][4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) of[0][kb] = ldg4(o + row * 64 + 16 * kb + 4 * g);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x4 acc[1] = {ldg4(bo + 16 * ob + 4 * g)};
        lin_acc<4, 1>(wofm + (long)ob * 4 * 256 + lane * 4, of, acc);
        if (mask) acc[0] = acc[0] * mask4(mask, row * 64 + 16 * ob + 4 * g, ms);
        if (res) acc[0] = acc[0] + ldg4(res + row * 64 + 16 * ob + 4 * g);
        if (ok) stg4(y + row * 64 + 16 * ob + 4 * g, acc[0]);
    }
}

// backward of to_out: dout = mask dy (kept for dWo / dbo), dO = Wo^T dout, D[token][head] = sum_d dO O
__global__ __launch_bounds__(256) void at_out_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ mask, float ms,
                                                         const float* __restrict__ o, long M,
                                                         const float* __restrict__ wotfm, float* __restrict__ dout,
                                                         float* __restrict__ dO, float* __restrict__ D,
                                                         float* __restrict__ maxslot) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    const long t = t0 + c;
    const bool ok = t < M;
    const long row = ok ? t : M - 1;
    float mx = 0.f;
    f32x4 dyf[1][4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x4 v = ldg4(dy + row * 64 + 16 * ob + 4 * g);
        if (mask) v = v * mask4(mask, row * 64 + 16 * ob + 4 * g, ms);
        dyf[0][ob] = v;
        if (ok) stg4(dout + row * 64 + 16 * ob + 4 * g, v);
    }
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {                       // feature block hb = head hb
        f32x4 acc[1] = {splat4(0.f)};
        lin_acc<4, 1>(wotfm + (long)hb * 4 * 256 + lane * 4, dyf, acc);
        const f32x4 ov = ldg4(o + row * 64 + 16 * hb + 4 * g);
        float part = acc[0][0] * ov[0] + acc[0][1] * ov[1] + acc[0][2] * ov[2] + acc[0][3] * ov[3];
        part = red_g_sum(part);
        if (ok) {
            stg4(dO + row * 64 + 16 * hb + 4 * g, acc[0]);
            if (g == 0) D[row * 4 + hb] = part;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc[0][0]), fabsf(acc[0][1]))), fmaxf(fabsf(acc[0][2]), fabsf(acc[0][3])));
        }
    }
#if AT_X3
    // the wave's largest |dO|, one slot per 16-token tile (at_amax_kernel reduces them): the exact power-of-two scale of
    // the split-f16 cores (at_scale).  (One atomicMax per wave on a single word cost ~70 us per launch.)
    mx = red_g_max(mx);
    mx = fmaxf(mx, dpp_perm<0xB1>(mx));
    mx = fmaxf(mx, dpp_perm<0x4E>(mx));
    mx = fmaxf(mx, dpp_perm<0x141>(mx));
    mx = fmaxf(mx, dpp_perm<0x140>(mx));
    if (lane == 0) maxslot[t0 >> 4] = mx;
#else
    (void)mx; (void)maxslot;
#endif
}
#if AT_X3
// out[0] = max of n non-negative floats (one block; max is order-independent)
__global__ __launch_bounds__(1024) void at_amax_kernel(const float* __restrict__ slots, long n, float* __restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, slots[i]);
    m = red_g_max(m);
    m = fmaxf(m, dpp_perm<0xB1>(m));
    m = fmaxf(m, dpp_perm<0x4E>(m));
    m = fmaxf(m, dpp_perm<0x141>(m));
    m = fmaxf(m, dpp_perm<0x140>(m));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
        out[0] = m;
    }
}
// dO (fp32, as at_out_bwd_kernel left it) -> the scaled image IN PLACE (each float4 becomes its own 16 bytes)
__global__ __launch_bounds__(256) void at_dO_split_kernel(float* __restrict__ dO, float* __restrict__ dOp,
                                                          const float* __restrict__ amax, long n4) {
    float gs, ginv;
    at_scale(amax, gs, ginv);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        (void)dOp;
        stg4(dO + 4 * i, at_img4(ldg4(dO + 4 * i) * splat4(gs)));
    }
}
#endif

// p_ij and ds_ij of one (query i, key j) pair; scores are recomputed, never stored.  q is the RAW query row (the
// 16^-0.5 scale is applied to the score), so that rows can come straight from wave-uniform scalar loads.
// dq: task (n, h, query block); P^T / dS^T tiles (key 4g + r, query c).   dq_i = scale sum_j ds_ij (k_j + E[i - j])
__global__ __launch_bounds__(256) void at_dq_kernel(AtBufs b, const float* __restrict__ ewin, const float* __restrict__ ewinp,
                                                    const float* __restrict__ dO, const float* __restrict__ D,
                                                    const float* __restrict__ amax, int L, int nb, long ntask,
                                                    float* __restrict__ dqkv) {
    __shared__ float sm[4][32 * AT_PA + 16 * AT_PS];
    AtTask t;
    if (!at_task(ntask, nb, t)) return;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* buf = sm[threadIdx.x >> 6];
    float* buf2 = buf + 32 * AT_PA;       // dS patch [query][16 + key]
    at_zero_pads(buf2, lane);
    const int h = t.nh & 3, I0 = 16 * t.blk, W = 16 * nb + 16, nfull = L >> 4;
    const long base = (long)(t.nh >> 2) * L;
    const float* __restrict__ qh = b.qkv + base * 192 + 16 * h;
    const float* __restrict__ qhp = b.qkvp + base * 192 + 16 * h;
    const float* __restrict__ e_blk = ewin + (long)(I0 - 15 + W) * 16;
    const float* __restrict__ e_blkp = ewinp + (long)(I0 - 15 + W) * 16;
    const int ri = I0 + c < L ? I0 + c : L - 1;
    const AtB qa = at_b_of(at_ldq(qh + (long)ri * 192 + 4 * g));
    float gs, ginv;
    at_scale(amax, gs, ginv);
    const AtB ga = at_b_of(at_lda(dO + (base + ri) * 64 + 16 * h + 4 * g));        // the dO image is stored scaled by gs
    const float lse = b.lse[(long)t.nh * L + ri] * AT_LOG2E, Di = D[(base + ri) * 4 + h] * gs;
    const unsigned la = c * 192 + 4 * g, lb = 4 * g * 192 + c, le = c * 16 + 4 * g, lg = g * 16 + c;
    const unsigned la_t = at_off_a(16 * nfull, L, 192, c, g);
    unsigned lb_t[4];
    bool vt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lb_t[r] = at_off_b(16 * nfull, L, 192, c, g, r); vt[r] = 16 * nfull + 4 * g + r < L; }
    f32x4 dq = splat4(0.f);               // dq[query 4g + r][d = c]
    f32x4 eq1 = at_dot(at_lda(e_blk + 256 + le), qa);
    AtB eb_prev;                          // E rows of the previous key block's lower 16 distances = this block's upper 16
    {
        f32x4 raw;
#pragma unroll
        for (int s = 0; s < 4; ++s) raw[s] = e_blkp[256 + lg + 64 * s];
        eb_prev = at_row_b(raw);
    }
    struct Frag { AtA ka, va, ea; f32x4 kb, eb; };
    auto load = [&](auto tail, int jb) {
        constexpr bool TAIL = decltype(tail)::value;
        const float* __restrict__ kp = qh + 64 + (long)jb * (16 * 192);
        const float* __restrict__ kpp = qhp + 64 + (long)jb * (16 * 192);
        const float* __restrict__ ep = e_blk - (long)jb * 256;
        const float* __restrict__ epp = e_blkp - (long)jb * 256;
        Frag f;
        f.ka = at_lda(kp + (TAIL ? la_t : la));
        f.va = at_lda(kp + 64 + (TAIL ? la_t : la));
#pragma unroll
        for (int r = 0; r < 4; ++r) f.kb[r] = kpp[TAIL ? lb_t[r] : lb + r * 192];
        f.ea = at_lda(ep + le);
#pragma unroll
        for (int s = 0; s < 4; ++s) f.eb[s] = epp[lg + 64 * s];
        return f;
    };
    auto tile = [&](auto tail, const Frag& f) {
        constexpr bool TAIL = decltype(tail)::value;
        const AtB kb = at_row_b(f.kb), eb_lo = at_row_b(f.eb);
        const f32x4 eq0 = at_dot(f.ea, qa);
        const f32x4 st = at_dot(f.ka, qa), dpt = at_dot(f.va, ga);
        const f32x4 rt = at_skew_t(buf, eq0, eq1, c, g);
        eq1 = eq0;
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (!TAIL || vt[r]) ? __builtin_amdgcn_exp2f(st[r] + rt[r] - lse) : 0.f;
            ds[r] = p * (dpt[r] - Di);
        }
        dq = at_mma(at_a(ds), kb, dq);
        // unskew: dSE[query c][distance 4s + g] = dS[c][15 + c - (4s + g)], zero outside the tile (the pads)
        wave_lds_fence();
        *reinterpret_cast<f32x4*>(buf2 + c * AT_PS + 16 + 4 * g) = ds;
        wave_lds_fence();
        f32x4 a_lo, a_hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a_lo[s] = buf2[c * (AT_PS + 1) + 3 - g + 4 * (7 - s)];
            a_hi[s] = buf2[c * (AT_PS + 1) + 3 - g + 4 * (3 - s)];
        }
        dq = at_mma(at_a(a_lo), eb_lo, dq);
        dq = at_mma(at_a(a_hi), eb_prev, dq);
        eb_prev = eb_lo;
    };
    if (nfull > 0) {
        // two blocks per trip: the other block's operands are in flight while one is being worked on
        Frag fa = load(std::false_type{}, 0), fb;
        int jb = 0;
        for (; jb + 1 < nfull; jb += 2) {
            fb = load(std::false_type{}, jb + 1);
            tile(std::false_type{}, fa);
            fa = load(std::false_type{}, jb + 2 < nfull ? jb + 2 : jb + 1);
            tile(std::false_type{}, fb);
        }
        if (jb < nfull) tile(std::false_type{}, fa);
    }
    if (nfull < nb) tile(std::true_type{}, load(std::true_type{}, nfull));
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (I0 + 4 * g + r < L) dqkv[(base + I0 + 4 * g + r) * 192 + 16 * h + c] = dq[r] * (0.25f * ginv);
}

// one tile in the (query 4g + r, key c) orientation: P and dS.  lse (log2 units) / D are per query ROW here.
__device__ __forceinline__ void at_tile_pds(float* buf, const AtA& qa, const AtA& ga, const AtB& ka, const AtB& va,
                                            const AtB& e0, const AtB& e1, const f32x4& lse, const f32x4& Dr, int c, int g,
                                            f32x4& p, f32x4& ds) {
    const f32x4 sc = at_dot(qa, ka), dp = at_dot(ga, va);
    const f32x4 rr = at_skew(buf, at_dot(qa, e0), at_dot(qa, e1), c, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p[r] = __builtin_amdgcn_exp2f(sc[r] + rr[r] - lse[r]);
        ds[r] = p[r] * (dp[r] - Dr[r]);
    }
}

// dk, dv: task (n, h, key block).          dk_j = scale sum_i ds_ij q_i,  dv_j = sum_i p_ij dO_i
// (key columns past the end of the sequence only feed their own, never stored, rows: no mask for them)
__global__ __launch_bounds__(256) void at_dkv_kernel(AtBufs b, const float* __restrict__ ewin, const float* __restrict__ dO,
                                                     const float* __restrict__ dOp, const float* __restrict__ D,
                                                     const float* __restrict__ amax, int L, int nb, long ntask,
                                                     float* __restrict__ dqkv) {
    __shared__ float sm[4][16 * AT_PB];
    AtTask t;
    if (!at_task(ntask, nb, t)) return;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* buf = sm[threadIdx.x >> 6];
    const int h = t.nh & 3, J0 = 16 * t.blk, W = 16 * nb + 16, nfull = L >> 4;
    const long base = (long)(t.nh >> 2) * L;
    const float* __restrict__ qh = b.qkv + base * 192 + 16 * h;
    const float* __restrict__ qhp = b.qkvp + base * 192 + 16 * h;
    const float* __restrict__ gh = dO + base * 64 + 16 * h;
    const float* __restrict__ ghp = dOp + base * 64 + 16 * h;
    const float* __restrict__ lh = b.lse + (long)t.nh * L;
    const float* __restrict__ Dh = D + base * 4 + h;
    const float* __restrict__ e_blk = ewin + (long)(W - J0 - 15) * 16;             // band of query block 0; block ib: + 256 ib
    const long rj = (long)J0 * 192 + at_off_a(J0, L, 192, c, g);
    const AtB ka = at_b_of(at_lda(qh + 64 + rj)), va = at_b_of(at_lda(qh + 128 + rj));
    const unsigned le = c * 16 + 4 * g;
    float gs, ginv;
    at_scale(amax, gs, ginv);
    f32x4 dk = splat4(0.f), dv = splat4(0.f);        // [key 4g + r][d = c]
    AtB e0 = at_b_of(at_lda(e_blk + le));
    struct Frag { AtA qa, ga, e1; f32x4 qb, gb, lse, Dr; int I0; };
    auto load = [&](auto tail, int ib) {
        constexpr bool TAIL = decltype(tail)::value;
        const int I0 = 16 * ib;
        const float* __restrict__ qp = qh + (long)ib * (16 * 192);
        const float* __restrict__ gp = gh + (long)ib * (16 * 64);
        const float* __restrict__ qpp = qhp + (long)ib * (16 * 192);
        const float* __restrict__ gpp = ghp + (long)ib * (16 * 64);
        Frag f;
        f.I0 = I0;
        f.qa = at_ldq(qp + (TAIL ? at_off_a(I0, L, 192, c, g) : c * 192 + 4 * g));
        f.ga = at_lda(gp + (TAIL ? at_off_a(I0, L, 64, c, g) : c * 64 + 4 * g));
        f.e1 = at_lda(e_blk + (long)ib * 256 + 256 + le);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = TAIL ? (I0 + 4 * g + r < L ? 4 * g + r : L - 1 - I0) : 4 * g + r;
            f.qb[r] = qpp[rr * 192 + c];
            f.gb[r] = gpp[rr * 64 + c];
            f.lse[r] = lh[I0 + rr];
            f.Dr[r] = Dh[(long)(I0 + rr) * 4];
        }
        return f;
    };
    auto tile = [&](auto tail, const Frag& f) {
        constexpr bool TAIL = decltype(tail)::value;
        const int I0 = f.I0;
        const AtB qb = at_row_b(f.qb), gb = at_row_b(f.gb), e1 = at_b_of(f.e1);
        f32x4 p, ds;
        at_tile_pds(buf, f.qa, f.ga, ka, va, e0, e1, f.lse * splat4(AT_LOG2E), f.Dr * splat4(gs), c, g, p, ds);
        e0 = e1;
        if (TAIL) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (I0 + 4 * g + r >= L) { p[r] = 0.f; ds[r] = 0.f; }
        }
        dk = at_mma(at_a(ds), qb, dk);
        dv = at_mma(at_a(p), gb, dv);
    };
    if (nfull > 0) {
        // two blocks per trip: the other block's operands are in flight while one is being worked on
        Frag fa = load(std::false_type{}, 0), fb;
        int ib = 0;
        for (; ib + 1 < nfull; ib += 2) {
            fb = load(std::false_type{}, ib + 1);
            tile(std::false_type{}, fa);
            fa = load(std::false_type{}, ib + 2 < nfull ? ib + 2 : ib + 1);
            tile(std::false_type{}, fb);
        }
        if (ib < nfull) tile(std::false_type{}, fa);
    }
    if (nfull < nb) tile(std::true_type{}, load(std::true_type{}, nfull));
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (J0 + 4 * g + r < L) {
            dqkv[(base + J0 + 4 * g + r) * 192 + 64 + 16 * h + c] = dk[r] * (AT_QBACK * ginv);
            dqkv[(base + J0 + 4 * g + r) * 192 + 128 + 16 * h + c] = dv[r] * ginv;
        }
}

// dE: task (n, h, t) walks tile diagonal delta = t (nb - t tiles) and then delta = t - nb (t tiles): nb tiles per task.
// All tiles of a diagonal cover the same 31 distances 16 delta - 15 .. 16 delta + 15, so the band gradient
//   dEband[dist][:] = scale sum_{i - j = dist} ds_ij q_i
// stays in two accumulators and is written once per diagonal: slab [(n, h)][delta + nb - 1][32][16].
__global__ __launch_bounds__(256) void at_de_kernel(AtBufs b, const float* __restrict__ ewin, const float* __restrict__ dO,
                                                    const float* __restrict__ D, const float* __restrict__ amax, int L, int nb,
                                                    long ntask, float* __restrict__ partial) {
    __shared__ float sm[4][16 * AT_PB + 16 * AT_PS];
    AtTask t;
    if (!at_task(ntask, nb, t)) return;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float* buf = sm[threadIdx.x >> 6];
    float* buf2 = buf + 16 * AT_PB;       // dS patch [query][16 + key]
    at_zero_pads(buf2, lane);
    const int h = t.nh & 3, W = 16 * nb + 16, nfull = L >> 4;
    const long base = (long)(t.nh >> 2) * L;
    const float* __restrict__ qh = b.qkv + base * 192 + 16 * h;
    const float* __restrict__ qhp = b.qkvp + base * 192 + 16 * h;
    const float* __restrict__ gh = dO + base * 64 + 16 * h;
    const float* __restrict__ lh = b.lse + (long)t.nh * L;
    const float* __restrict__ Dh = D + base * 4 + h;
    const unsigned le = c * 16 + 4 * g;
    float gs, ginv;
    at_scale(amax, gs, ginv);
    for (int seg = 0; seg < 2; ++seg) {
        if (seg == 1 && t.blk == 0) break;
        const int delta = seg == 0 ? t.blk : t.blk - nb;
        const int ntile = seg == 0 ? nb - t.blk : t.blk;
        const int ib0 = seg == 0 ? t.blk : 0, jb0 = seg == 0 ? 0 : nb - t.blk;
        const float* __restrict__ ep = ewin + (long)(16 * delta - 15 + W) * 16;
        const AtB e0 = at_b_of(at_lda(ep + le)), e1 = at_b_of(at_lda(ep + 256 + le));
        f32x4 de0 = splat4(0.f), de1 = splat4(0.f);  // [distance 16 blk + 4g + r][d = c]
        struct Frag { AtA qa, ga, ka, va; f32x4 qb, lse, Dr; int I0, J0; };
        auto load = [&](auto tail, int k) {
            constexpr bool TAIL = decltype(tail)::value;
            const int I0 = 16 * (ib0 + k), J0 = 16 * (jb0 + k);
            const float* __restrict__ qp = qh + (long)I0 * 192;
            const float* __restrict__ qpp = qhp + (long)I0 * 192;
            const float* __restrict__ gp = gh + (long)I0 * 64;
            const float* __restrict__ kp = qh + 64 + (long)J0 * 192;
            Frag f;
            f.I0 = I0; f.J0 = J0;
            f.qa = at_ldq(qp + (TAIL ? at_off_a(I0, L, 192, c, g) : c * 192 + 4 * g));
            f.ga = at_lda(gp + (TAIL ? at_off_a(I0, L, 64, c, g) : c * 64 + 4 * g));
            const unsigned oj = TAIL ? at_off_a(J0, L, 192, c, g) : c * 192 + 4 * g;
            f.ka = at_lda(kp + oj);
            f.va = at_lda(kp + 64 + oj);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = TAIL ? (I0 + 4 * g + r < L ? 4 * g + r : L - 1 - I0) : 4 * g + r;
                f.qb[r] = qpp[rr * 192 + c];
                f.lse[r] = lh[I0 + rr];
                f.Dr[r] = Dh[(long)(I0 + rr) * 4];
            }
            return f;
        };
        auto tile = [&](auto tail, const Frag& f) {
            constexpr bool TAIL = decltype(tail)::value;
            const int I0 = f.I0, J0 = f.J0;
            const AtB qb = at_row_b(f.qb);
            f32x4 p, ds;
            at_tile_pds(buf, f.qa, f.ga, at_b_of(f.ka), at_b_of(f.va), e0, e1, f.lse * splat4(AT_LOG2E), f.Dr * splat4(gs),
                        c, g, p, ds);
            if (TAIL) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (I0 + 4 * g + r >= L || J0 + c >= L) ds[r] = 0.f;
            }
            // unskew: dSE^T[distance 16 blk + c][query 4g + r] = dS[4g + r][15 + 4g + r - 16 blk - c], zero outside
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 4; ++r) buf2[(4 * g + r) * AT_PS + 16 + c] = ds[r];
            wave_lds_fence();
            f32x4 r0, r1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* row = buf2 + 4 * g * (AT_PS + 1) + 15 - c + r * (AT_PS + 1);
                r0[r] = row[16];
                r1[r] = row[0];
            }
            de0 = at_mma(at_a(r0), qb, de0);
            de1 = at_mma(at_a(r1), qb, de1);
        };
        // a diagonal's tiles are full except (when L is not a multiple of 16) its last one
        const int nfl = (ib0 + ntile > nfull || jb0 + ntile > nfull) ? ntile - 1 : ntile;
        if (nfl > 0) {
            // two blocks per trip: the other block's operands are in flight while one is being worked on
            Frag fa = load(std::false_type{}, 0), fb;
            int k = 0;
            for (; k + 1 < nfl; k += 2) {
                fb = load(std::false_type{}, k + 1);
                tile(std::false_type{}, fa);
                fa = load(std::false_type{}, k + 2 < nfl ? k + 2 : k + 1);
                tile(std::false_type{}, fb);
            }
            if (k < nfl) tile(std::false_type{}, fa);
        }
        if (nfl < ntile) tile(std::true_type{}, load(std::true_type{}, nfl));
        float* out = partial + ((long)t.nh * (2 * nb - 1) + (delta + nb - 1)) * 512;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[(4 * g + r) * 16 + c] = de0[r] * (AT_QBACK * ginv);
            out[(16 + 4 * g + r) * 16 + c] = de1[r] * (AT_QBACK * ginv);
        }
    }
}

// ---------------------------------------------------------------------------------
// Fused backward core (sequences of at most ATF_MAX_NB blocks): ONE block per (sequence, head) computes dq, dk, dv and
// the band gradient from a single recomputation of P and dS per 16 x 16 tile, instead of three kernels that each
// recompute them.  The obstacle to fusing is ownership: dq is summed along tile rows, dk / dv along tile columns and
// dE along tile diagonals.  Here a wave owns WRAPPED DIAGONALS t (tiles with i - j = t mod nbp, nbp = the odd number
// of blocks nb or nb + 1): in round r it works on tile (i, j) = ((r + 2t) mod nbp, (r + t) mod nbp), so that within a
// round all waves hold distinct query blocks AND distinct key blocks (t -> 2t and t -> t are injective mod an odd
// nbp: a Latin square).  dq / dk / dv tiles are accumulated in LDS without conflicts or atomics - a block barrier
// separates the rounds, every sum keeps a fixed order - and the band gradient of the wave's diagonal stays in two
// accumulators (a wrapped diagonal is two real ones, delta = t and t - nbp; the accumulators are flushed to the
// diagonal's slab when delta changes - pos -> neg -> pos or neg -> pos -> neg - and the second visit of a real
// diagonal adds to what the first one stored: same wave, fixed order).
// Per tile: 16 + 8 + 8 + 12 product steps instead of 28 + 24 + 24, one skew, one exp pass, one set of operand loads.
// LDS: three [16][16 nbp + 4] fp32 accumulators (transposed: a lane's four rows are one b128) + the two wave-private
// patches.  SLOTS = wrapped diagonals per wave: 1 up to 8 blocks, else 2 (waves = ceil(nbp / SLOTS) <= 12).
// ---------------------------------------------------------------------------------
#define ATF_MAX_NB 22                 // L <= 352: accumulators 3 x 16 x 372 x 4 B = 71 KB
#define ATF_PATCH (16 * AT_PB + 16 * AT_PS)
struct AtfStep { int i, j, delta, t; bool valid; };
template <int SLOTS>
__device__ __forceinline__ AtfStep atf_step(int r, int k, int NW, int wv, int nb, int nbp) {
    AtfStep s;
    s.t = wv + k * NW;
    int i = r + 2 * s.t, j = r + s.t;
    i -= i >= nbp ? nbp : 0;
    i -= i >= nbp ? nbp : 0;
    j -= j >= nbp ? nbp : 0;
    s.valid = s.t < nbp && i < nb && j < nb;
    s.i = s.valid ? i : 0;
    s.j = s.valid ? j : 0;
    s.delta = s.i - s.j;
    return s;
}
template <int SLOTS>
__global__ __launch_bounds__(SLOTS == 1 ? 512 : 768) void at_bwd_fused_kernel(AtBufs b, const float* __restrict__ ewin, const float* __restrict__ ewinp,
                                                           const float* __restrict__ dO, const float* __restrict__ dOp,
                                                           const float* __restrict__ D, const float* __restrict__ amax, int N,
                                                           int L, int nb, int nbp, float* __restrict__ dqkv,
                                                           float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float atf_sm[];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
    const int ROWP = 16 * nbp + 4;
    float* accq = atf_sm;
    float* acck = accq + 16 * ROWP;
    float* accv = acck + 16 * ROWP;
    float* buf = accv + 16 * ROWP + wv * ATF_PATCH;       // band patch [16][AT_PB]
    float* buf2 = buf + 16 * AT_PB;                       // dS patch [query][16 + key]
    // blocks are dealt round-robin to the 8 XCDs: the four heads of a sequence (whose 64-byte row segments interleave
    // in 128-byte lines) go to the SAME XCD - blocks b, b + 8, b + 16, b + 24 - so a line fetched into that L2 serves two
    // heads instead of being fetched by two L2s (the grid is ceil(N / 8) * 32 blocks)
    const int nseq = (int)(blockIdx.x >> 5) * 8 + (int)(blockIdx.x & 7);
    if (nseq >= N) return;
    const int nh = nseq * 4 + (int)((blockIdx.x >> 3) & 3);
    for (int e = threadIdx.x; e < 3 * 16 * ROWP; e += blockDim.x) atf_sm[e] = 0.f;
    at_zero_pads(buf2, lane);
    const int h = nh & 3, W = 16 * nb + 16;
    const bool ragged = (L & 15) != 0;
    const long base = (long)(nh >> 2) * L;
    const float* __restrict__ qh = b.qkv + base * 192 + 16 * h;
    const float* __restrict__ qhp = b.qkvp + base * 192 + 16 * h;
    const float* __restrict__ gh = dO + base * 64 + 16 * h;
    const float* __restrict__ ghp = dOp + base * 64 + 16 * h;
    const float* __restrict__ lh = b.lse + (long)nh * L;
    const float* __restrict__ Dh = D + base * 4 + h;
    const unsigned le = c * 16 + 4 * g, lg = g * 16 + c;
    float gs, ginv;
    at_scale(amax, gs, ginv);
    __syncthreads();

    // Frag: what a tile needs at its START, fetched one step ahead.  Rows: the row-type fragments it needs from its
    // middle on (dk / dv / dE / dq products), fetched at the start of the tile itself - their latency hides under the
    // score phase and they do not occupy registers across steps.
    struct Frag { AtA qa, ga, ka, va, e0, e1; f32x4 lse, Dr; };
    struct Rows { f32x4 qb, gb, kb, eb_lo, eb_hi; };
    // operand loads as uniform base (SGPRs, the row / distance block folded in with scalar adds) + a loop-invariant 32-bit
    // per-lane BYTE offset: the global_load "saddr" form, no 64-bit VALU address arithmetic per load
    const unsigned oa192 = (c * 192 + 4 * g) * 4, oa64 = (c * 64 + 4 * g) * 4, ob192 = (4 * g * 192 + c) * 4,
                   ob64 = (4 * g * 64 + c) * 4, ole = le * 4, olg = lg * 4;
    auto ld4 = [](const float* base, unsigned off) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off);
    };
    auto ld1 = [](const float* base, unsigned off) {
        return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off);
    };
    auto load = [&](const AtfStep& s) {
        const int I0 = 16 * s.i, J0 = 16 * s.j;
        const bool tq = ragged && s.i == nb - 1, tk = ragged && s.j == nb - 1;
        // A-type fragments from the PAIR images too (one 16-byte load + the four v_perm of at_row_a): reading both
        // interleavings made a (sequence, head)'s working set 164 KB - 5.2 MB per XCD with one block per CU, more than
        // its 4 MB L2, and the tile sweeps of the rounds missed it 67 % of the time (TCC counters; 17 % at L = 101)
        const float* __restrict__ qp = qhp + (long)I0 * 192;
        const float* __restrict__ gp = ghp + (long)I0 * 64;
        const float* __restrict__ kp = qhp + 64 + (long)J0 * 192;
        const float* __restrict__ ep = ewin + (long)(16 * s.delta - 15 + W) * 16;
        Frag f;
        if (!(tq || tk)) {
            f.qa = at_qfix(at_row_a(ld4(qp, oa192)));
            f.ga = at_row_a(ld4(gp, oa64));
            f.ka = at_row_a(ld4(kp, oa192));
            f.va = at_row_a(ld4(kp + 64, oa192));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f.lse[r] = ld1(lh + I0 + r, 16 * g);
                f.Dr[r] = ld1(Dh + (long)(I0 + r) * 4, 64 * g);
            }
        } else {                                                  // ragged last block: rows past the end read row L - 1
            const int ci = tq ? (I0 + c < L ? c : L - 1 - I0) : c;
            const int cj = tk ? (J0 + c < L ? c : L - 1 - J0) : c;
            f.qa = at_qfix(at_row_a(ldg4(qp + ci * 192 + 4 * g)));
            f.ga = at_row_a(ldg4(gp + ci * 64 + 4 * g));
            f.ka = at_row_a(ldg4(kp + cj * 192 + 4 * g));
            f.va = at_row_a(ldg4(kp + 64 + cj * 192 + 4 * g));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = tq ? (I0 + 4 * g + r < L ? 4 * g + r : L - 1 - I0) : 4 * g + r;
                f.lse[r] = lh[I0 + ri];
                f.Dr[r] = Dh[(long)(I0 + ri) * 4];
            }
        }
        f.e0 = at_row_a(ld4(ep, ole));
        f.e1 = at_row_a(ld4(ep + 256, ole));
        return f;
    };
    auto load_rows = [&](const AtfStep& s) {
        const int I0 = 16 * s.i, J0 = 16 * s.j;
        const bool tq = ragged && s.i == nb - 1, tk = ragged && s.j == nb - 1;
        const float* __restrict__ qpp = qhp + (long)I0 * 192;
        const float* __restrict__ gpp = ghp + (long)I0 * 64;
        const float* __restrict__ kpp = qhp + 64 + (long)J0 * 192;
        const float* __restrict__ epp = ewinp + (long)(16 * s.delta - 15 + W) * 16;
        Rows w;
        if (!(tq || tk)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w.qb[r] = ld1(qpp + r * 192, ob192);
                w.gb[r] = ld1(gpp + r * 64, ob64);
                w.kb[r] = ld1(kpp + r * 192, ob192);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = tq ? (I0 + 4 * g + r < L ? 4 * g + r : L - 1 - I0) : 4 * g + r;
                const int rj = tk ? (J0 + 4 * g + r < L ? 4 * g + r : L - 1 - J0) : 4 * g + r;
                w.qb[r] = qpp[ri * 192 + c];
                w.gb[r] = gpp[ri * 64 + c];
                w.kb[r] = kpp[rj * 192 + c];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.eb_lo[r] = ld1(epp + 64 * r, olg);
            w.eb_hi[r] = ld1(epp + 256 + 64 * r, olg);
        }
        return w;
    };
    auto tile = [&](const Frag& f, const AtfStep& s, f32x4& de0, f32x4& de1) {
        const int I0 = 16 * s.i, J0 = 16 * s.j;
        const Rows w = load_rows(s);
        f32x4 p, ds;
        at_tile_pds(buf, f.qa, f.ga, at_b_of(f.ka), at_b_of(f.va), at_b_of(f.e0), at_b_of(f.e1), f.lse * splat4(AT_LOG2E),
                    f.Dr * splat4(gs), c, g, p, ds);
        if (ragged && (s.i == nb - 1 || s.j == nb - 1)) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (I0 + 4 * g + r >= L || J0 + c >= L) { p[r] = 0.f; ds[r] = 0.f; }
        }
        // dk_j += dS^T q,  dv_j += P^T dO       [key 4g + r][d = c], accumulated in the transposed LDS arrays
        const AtB qb = at_row_b(w.qb);
        {
            const f32x4 dk = at_dot(at_a(ds), qb), dv = at_dot(at_a(p), at_row_b(w.gb));
            f32x4* pk = reinterpret_cast<f32x4*>(acck + c * ROWP + J0 + 4 * g);
            f32x4* pv = reinterpret_cast<f32x4*>(accv + c * ROWP + J0 + 4 * g);
            *pk = *pk + dk;
            *pv = *pv + dv;
        }
        // the dS patch serves the band gradient (unskew^T), dq's band term (unskew) and dq's key term (transpose)
        wave_lds_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) buf2[(4 * g + r) * AT_PS + 16 + c] = ds[r];
        wave_lds_fence();
        f32x4 r0, r1, a_lo, a_hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* row = buf2 + 4 * g * (AT_PS + 1) + 15 - c + r * (AT_PS + 1);
            r0[r] = row[16];
            r1[r] = row[0];
            a_lo[r] = buf2[c * (AT_PS + 1) + 3 - g + 4 * (7 - r)];
            a_hi[r] = buf2[c * (AT_PS + 1) + 3 - g + 4 * (3 - r)];
        }
        const f32x4 dst = *reinterpret_cast<const f32x4*>(buf2 + c * AT_PS + 16 + 4 * g);   // dS[query c][key 4g + r]
        de0 = at_mma(at_a(r0), qb, de0);
        de1 = at_mma(at_a(r1), qb, de1);
        f32x4 dq = at_dot(at_a(dst), at_row_b(w.kb));
        dq = at_mma(at_a(a_lo), at_row_b(w.eb_lo), dq);
        dq = at_mma(at_a(a_hi), at_row_b(w.eb_hi), dq);
        f32x4* pq = reinterpret_cast<f32x4*>(accq + c * ROWP + I0 + 4 * g);
        *pq = *pq + dq;
    };
    // band-gradient slabs [(n, h)][delta + nb - 1][32][16]: first visit of a real diagonal stores, the second adds
    auto flush = [&](int delta, const f32x4& de0, const f32x4& de1, bool add) {
        float* out = partial + ((long)nh * (2 * nb - 1) + (delta + nb - 1)) * 512;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* o0 = out + (4 * g + r) * 16 + c;
            float* o1 = out + (16 + 4 * g + r) * 16 + c;
            const float v0 = de0[r] * (AT_QBACK * ginv), v1 = de1[r] * (AT_QBACK * ginv);
            *o0 = add ? *o0 + v0 : v0;
            *o1 = add ? *o1 + v1 : v1;
        }
    };
    f32x4 de0[SLOTS], de1[SLOTS];
    int cur[SLOTS];
    bool posdone[SLOTS], negdone[SLOTS];      // a wrapped diagonal visits one of its two real diagonals twice
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        de0[k] = splat4(0.f); de1[k] = splat4(0.f); cur[k] = -4096; posdone[k] = false; negdone[k] = false;
    }

    // one step = one (round, slot) tile of this wave; the next step's operands are in flight while a tile is worked on.
    // Two rounds per trip so that the two operand buffers alternate with compile-time parity (no register copies).
    auto work = [&](const Frag& f, const AtfStep& s, int k) {
        if (!s.valid) return;
        if (s.delta != cur[k]) {
            if (cur[k] != -4096) {
                flush(cur[k], de0[k], de1[k], cur[k] >= 0 ? posdone[k] : negdone[k]);
                if (cur[k] >= 0) posdone[k] = true;
                else negdone[k] = true;
            }
            cur[k] = s.delta;
            de0[k] = splat4(0.f);
            de1[k] = splat4(0.f);
        }
        tile(f, s, de0[k], de1[k]);
    };
    auto step_at = [&](int q) {                                       // q = round * SLOTS + slot; past the end: a dummy
        const int r = q / SLOTS, k = q - r * SLOTS;
        return atf_step<SLOTS>(r < nbp ? r : 0, k, NW, wv, nb, nbp);
    };
    AtfStep sa = step_at(0), sb;
    Frag fa = load(sa), fb;
    for (int r = 0; r < nbp; r += 2) {
        const bool second = r + 1 < nbp;                              // nbp is odd: the last trip has one round
#pragma unroll
        for (int p = 0; p < 2 * SLOTS; p += 2) {
            // steps p (buffer a) and p + 1 (buffer b) of this trip; step p + 2 goes back into buffer a
            const int q = r * SLOTS + p;
            sb = step_at(q + 1);
            if (!second && p + 1 >= SLOTS) sb.valid = false;
            fb = load(sb);
            if (second || p < SLOTS) work(fa, sa, p % SLOTS);
            if (p % SLOTS == SLOTS - 1) __syncthreads();              // end of a round: its dq / dk / dv tiles are in LDS
            sa = step_at(q + 2);
            if (!second && p + 2 >= SLOTS) sa.valid = false;
            fa = load(sa);
            if (second || p + 1 < SLOTS) work(fb, sb, (p + 1) % SLOTS);
            if ((p + 1) % SLOTS == SLOTS - 1) __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k)
        if (cur[k] != -4096) flush(cur[k], de0[k], de1[k], cur[k] >= 0 ? posdone[k] : negdone[k]);
    for (int e = threadIdx.x; e < L * 16; e += blockDim.x) {
        const int row = e >> 4, d = e & 15;
        float* o = dqkv + (base + row) * 192 + 16 * h + d;
        o[0] = accq[d * ROWP + row] * (0.25f * ginv);
        o[64] = acck[d * ROWP + row] * (AT_QBACK * ginv);
        o[128] = accv[d * ROWP + row] * ginv;
    }
}
// > 64 KB of dynamic LDS is an opt-in per kernel AND per device; the answer is remembered per (device, kernel), a refusal
// included (the caller then takes the three-kernel path)
static bool atf_optin(const void* fn) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, bool> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    auto it = done.find({dev, fn});
    if (it != done.end()) return it->second;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();               // clear the sticky error: the fallback path is valid
    done[{dev, fn}] = (e == hipSuccess);
    return e == hipSuccess;
}
static size_t atf_lds_bytes(int nbp, int nw) { return ((size_t)3 * 16 * (16 * nbp + 4) + (size_t)nw * ATF_PATCH) * sizeof(float); }

// rel_pos_emb gradient [2 max_pos + 1][16] from the (n, h)-summed diagonal slabs [2 nb - 1][32][16]: row e sums, in
// distance then diagonal order, every band row whose clamped distance is e - max_pos (a distance lies in the bands of
// one or two neighbouring diagonals)
__global__ void at_de_scatter_kernel(const float* __restrict__ slabs, int L, int nb, int max_pos, float* __restrict__ drel) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int rows = 2 * max_pos + 1;
    if (idx >= rows * 16) return;
    const int e = idx >> 4, d = idx & 15;
    const int dist = e - max_pos;
    int lo = dist, hi = dist;
    if (dist == -max_pos) lo = -(L - 1);                  // every distance <= -max_pos
    if (dist == max_pos) hi = L - 1;                      // every distance >= +max_pos
    lo = lo < -(L - 1) ? -(L - 1) : lo;
    hi = hi > L - 1 ? L - 1 : hi;
    float s = 0.f;
    for (int dd = lo; dd <= hi; ++dd) {
        // diagonals delta with 16 delta - 15 <= dd <= 16 delta + 15
        int d_lo = (dd - 15 + 16 * nb + 15) / 16 - nb, d_hi = (dd + 15 + 16 * nb) / 16 - nb;    // ceil / floor, shifted positive
        d_lo = d_lo < -(nb - 1) ? -(nb - 1) : d_lo;
        d_hi = d_hi > nb - 1 ? nb - 1 : d_hi;
        for (int delta = d_lo; delta <= d_hi; ++delta)
            s += slabs[((long)(delta + nb - 1) * 32 + (dd - (16 * delta - 15))) * 16 + d];
    }
    drel[idx] = s;
}


// backward of the projections: dxn = [to_q ; to_kv]^T dqkv (A image [4][12]), LayerNorm backward
__global__ __launch_bounds__(256) void at_qkv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dqkv, long M,
                                                         const float* __restrict__ wtfm, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, const float* __restrict__ dres,
                                                         float* __restrict__ dx, float* __restrict__ xn_out,
                                                         float* __restrict__ g1, float* __restrict__ dxn_out) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, ln_w, ln_b, xh, xn, rstd, row);
    f32x4 df[1][12];
#pragma unroll
    for (int kb = 0; kb < 12; ++kb) {
        df[0][kb] = ldg4(dqkv + row * 192 + 16 * kb + 4 * g);
        if (!ok) df[0][kb] = splat4(0.f);
    }
    f32x4 dxn[4];
    // 12 groups of 4 fragments (image [4 row blocks][12 k-blocks] = group index rb * 3 + third); next group in flight
    {
        FfnFrag f0 = ffn_frag_rows(wtfm, 0, lane), f1;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 acc = splat4(0.f);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int gi = rb * 3 + t;
                FfnFrag& cur = (gi & 1) ? f1 : f0;
                FfnFrag& nxt = (gi & 1) ? f0 : f1;
                nxt = ffn_frag_rows(wtfm, gi + 1 < 12 ? gi + 1 : 11, lane);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = mfma16(cur.a[kb][r], df[0][4 * t + kb][r], acc);
            }
            dxn[rb] = acc;
        }
    }
    f32x4 dxh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        dxh[kb] = dxn[kb] * ldg4(ln_w + 16 * kb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += dxh[kb][r];
            s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
        }
    }
    const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
    if (ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 dv = (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd);
            if (dres) dv = dv + ldg4(dres + row * 64 + 16 * kb + 4 * g);
            stg4(dx + row * 64 + 16 * kb + 4 * g, dv);
            stg4(xn_out + row * 64 + 16 * kb + 4 * g, xn[0][kb]);
        }
    }
    f32x4 ca[4], cb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        ca[kb] = ok ? dxn[kb] * xh[kb] : splat4(0.f);
        cb[kb] = ok ? dxn[kb] : splat4(0.f);
    }
    ln_tile_colsums(ca, cb, c, g, t0 >> 4, g1, dxn_out);            // [tiles][64] slabs: dgamma / dbeta partial sums
}

// dynamic LDS above the 64 KB default needs an explicit opt-in per kernel
static int at_blocks(int L) { return (L + 15) / 16; }
static int at_window(int L) { return 16 * at_blocks(L) + 16; }      // half-width W of the relative-position window
// launch width of the core kernels: four one-wave tasks per block, blocks rounded up to a multiple of the 8 XCDs
static unsigned at_core_grid(long ntask) { return (unsigned)(((ntask + 3) / 4 + 7) / 8 * 8); }
struct AtPlan { size_t raw, wqkv, wqkvt, wo, wot, ewin, ewinp, qkv, qkvp, o, lse, dout, dO, dOp, D, dqkv, xn, g1, dxn, depart, dewin, wpart, cpart, total; };
static AtPlan at_plan(int N, int L) {
    AtPlan p;
    const size_t M = (size_t)N * L;
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    p.raw = take(12288); p.wqkv = take(12288); p.wqkvt = take(12288); p.wo = take(4096); p.wot = take(4096);
    p.ewin = take((size_t)(2 * at_window(L) + 1) * 16);
    p.qkv = take(M * 192); p.o = take(M * 64); p.lse = take((size_t)N * 4 * L);
    p.dout = take(M * 64); p.dO = take(M * 64); p.D = take(M * 4); p.dqkv = take(M * 192);
    p.ewinp = p.ewin; p.qkvp = p.qkv; p.dOp = p.dO;             // one image per tensor (or the fp32 tensor itself)
    p.xn = take(M * 64); p.g1 = take((M + 15) / 16 * 64); p.dxn = take((M + 15) / 16 * 64);      // g1 / dxn: per-tile partial rows
    const size_t slabs = (size_t)(2 * at_blocks(L) - 1) * 512;       // dE band slabs [2 nb - 1][32][16]
    p.depart = take((size_t)N * 4 * slabs);
    p.dewin = take(slabs);
    p.wpart = take((size_t)WG_SPLIT * 12288);
    p.cpart = take((size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256);
    p.total = cur;
    return p;
}
size_t attn_train_ws_floats(int N, int L) { return at_plan(N, L).total; }
int attn_train_max_len() { return AT_MAX_L; }

static void at_pack_images(LaunchCtx ctx, const AttnTrainParams& p, float* ws, const AtPlan& pl, bool pack = true) {
    if (!pack) return;
    hipStream_t s = ctx.stream;
    // [to_q ; to_kv] as one [192,64] matrix: in place when the two parameters are adjacent (views of one flat bucket), else a copy
    const float* raw = p.wq;
    if (p.wkv != p.wq + 4096) {
        hipMemcpyAsync(ws + pl.raw, p.wq, 4096 * sizeof(float), hipMemcpyDeviceToDevice, s);            // rows 0..63
        hipMemcpyAsync(ws + pl.raw + 4096, p.wkv, 8192 * sizeof(float), hipMemcpyDeviceToDevice, s);    // rows 64..191
        raw = ws + pl.raw;
    }
#if TRAIN_X3
    at_x3_pack(ctx, raw, ws + pl.wqkv, ws + pl.wqkvt);              // the projections run on split products
    launch_pack4(ctx, "attn_train_pack", PackJobs{{{p.wo, 64, 64, 64, 0, ws + pl.wo}, {p.wo, 64, 64, 64, 1, ws + pl.wot},
                                                   {}, {}}}, 2);
#else
    launch_pack4(ctx, "attn_train_pack", PackJobs{{{raw, 192, 64, 64, 0, ws + pl.wqkv},
                                                   {raw, 64, 192, 64, 1, ws + pl.wqkvt},
                                                   {p.wo, 64, 64, 64, 0, ws + pl.wo}, {p.wo, 64, 64, 64, 1, ws + pl.wot}}});
#endif
}

void launch_attn_train_forward(LaunchCtx ctx, const float* x, int N, int L, const AttnTrainParams& p, int max_pos,
                               const unsigned char* mask, float ms, const float* res, float* y, float* ws) {
    hipStream_t s = ctx.stream;
    const AtPlan pl = at_plan(N, L);
    const long M = (long)N * L;
    at_pack_images(ctx, p, ws, pl);
    const AtBufs b{ws + pl.qkv, ws + pl.qkvp, ws + pl.o, ws + pl.lse};
    const unsigned grid = (unsigned)((M + 63) / 64);
#if TRAIN_X3
    at_x3_qkv(ctx, x, M, ws + pl.wqkv, p.ln_w, p.ln_b, b.qkv, AT_X3, AT_QSCALE);
#else
    LAUNCH(ctx, "attn_train_fwd", (at_qkv_kernel<<<grid, 256, 0, s>>>(x, M, ws + pl.wqkv, p.ln_w, p.ln_b, b.qkv, b.qkvp)));
#endif
    const int nb = at_blocks(L);
    const long ntask = (long)N * 4 * nb;
    const int W = at_window(L);
    LAUNCH(ctx, "attn_train_pack", (at_window_kernel<<<((2 * W + 1) * 4 + 255) / 256, 256, 0, s>>>(p.rel, W, max_pos, ws + pl.ewin,
                                                                                                   ws + pl.ewinp)));
    LAUNCH(ctx, "attn_train_fwd", (at_fwd_kernel<<<at_core_grid(ntask), 256, 0, s>>>(b, ws + pl.ewin, L, nb, ntask)));
    LAUNCH(ctx, "attn_train_fwd", (at_out_kernel<<<grid, 256, 0, s>>>(b.o, M, ws + pl.wo, p.bo, mask, ms, res, y)));
}

void launch_attn_train_backward(LaunchCtx ctx, const float* x, const float* dy, int N, int L, const AttnTrainParams& p,
                                int max_pos, const unsigned char* mask, float ms, const float* dres, float* dx,
                                const AttnTrainParams& grad, float* ws) {
    hipStream_t s = ctx.stream;
    const AtPlan pl = at_plan(N, L);
    const long M = (long)N * L;
    at_pack_images(ctx, p, ws, pl, false);
    const AtBufs b{ws + pl.qkv, ws + pl.qkvp, ws + pl.o, ws + pl.lse};
    const unsigned grid = (unsigned)((M + 63) / 64);
    float* cpart = ws + pl.cpart;                         // [0]: max |dO|; [64 ..): its per-tile maxima (the column-sum slabs
                                                          // are idle until the end of this function)
    LAUNCH(ctx, "attn_train_bwd", (at_out_bwd_kernel<<<grid, 256, 0, s>>>(dy, mask, ms, b.o, M, ws + pl.wot, ws + pl.dout,
                                                                          ws + pl.dO, ws + pl.D, cpart + 64)));
#if AT_X3
    LAUNCH(ctx, "attn_train_bwd", (at_amax_kernel<<<1, 1024, 0, s>>>(cpart + 64, (M + 15) / 16, cpart)));
    LAUNCH(ctx, "attn_train_bwd", (at_dO_split_kernel<<<2048, 256, 0, s>>>(ws + pl.dO, ws + pl.dOp, cpart, M * 16)));
#endif
    // to_out gradients: dWo [64,64] = dout^T O, dbo = colsum dout
#if TRAIN_X3
    // (cpart holds the per-block |dO| maxima in its first rows until at_amax_kernel / at_dO_split_kernel above have run; the
    // column-sum partials of dout go to the fourth job's region, which colsum_batch below - three jobs at most - never uses)
    wgrad_partial64(ctx, "attn_train_wgrad", ws + pl.dout, b.o, M, 64, 64, ws + pl.wpart, wg_split(1),
                    cpart + (size_t)3 * FFN_COLSUM_BLOCKS * 256, grad.bo, "attn_train_reduce");
#else
    wgrad_partial64(ctx, "attn_train_wgrad", ws + pl.dout, b.o, M, 64, 64, ws + pl.wpart, wg_split(1));
#endif
    LAUNCH(ctx, "attn_train_reduce", (reduce_partials_kernel<<<64, 1024, 0, s>>>(ws + pl.wpart, wg_split(1), 4096,
                                                                                grad.wo)));
    // attention core: dq (query blocks), dk / dv (key blocks), dE (tile diagonals)
    const int nb = at_blocks(L);
    const long ntask = (long)N * 4 * nb;
    const unsigned cgrid = at_core_grid(ntask);
    const char* bwd_env = getenv("CMGAN_ATTN_BWD");               // read per launch: tests switch it inside one process
    const bool fused_env = !(bwd_env != nullptr && strcmp(bwd_env, "cores") == 0);
    // two wrapped diagonals per wave at EVERY length (CMGAN_ATF_SLOTS_SHORT=1: one per wave up to 8 blocks, the form of rounds
    // 3 - 5).  At L = 101 that is 4 waves and 44 KB of LDS per block instead of 7 waves and 60 KB: three blocks per CU instead of
    // one (the kernel's 160 VGPRs allow 12 waves) - attention backward 45.8 -> 41.9 ms per step, same-session.
    static const int k_short_slots = env_knob("CMGAN_ATF_SLOTS_SHORT", 2, 1, 2);
    const int nbp = nb | 1, slots = nbp <= 8 ? k_short_slots : 2, nw = (nbp + slots - 1) / slots;       // <= 12 waves
    // the fused kernel needs more than 64 KB of dynamic LDS: opted into once per (device, instantiation); if the runtime
    // refuses, the three cores run instead (same results)
    const bool fused_ok = fused_env && nb <= ATF_MAX_NB &&
                          atf_optin(slots == 1 ? reinterpret_cast<const void*>(&at_bwd_fused_kernel<1>)
                                               : reinterpret_cast<const void*>(&at_bwd_fused_kernel<2>));
    if (fused_ok) {
        const size_t lds = atf_lds_bytes(nbp, nw);
#define ATF_LAUNCH(SL)                                                                                                     \
        LAUNCH(ctx, "attn_train_bwd", (at_bwd_fused_kernel<SL><<<(N + 7) / 8 * 32, 64 * nw, lds, s>>>(                      \
                                          b, ws + pl.ewin, ws + pl.ewinp, ws + pl.dO, ws + pl.dOp, ws + pl.D, cpart, N, L,  \
                                          nb, nbp, ws + pl.dqkv, ws + pl.depart)))
        if (slots == 1) ATF_LAUNCH(1);
        else ATF_LAUNCH(2);
#undef ATF_LAUNCH
    } else {
    LAUNCH(ctx, "attn_train_bwd", (at_dq_kernel<<<cgrid, 256, 0, s>>>(b, ws + pl.ewin, ws + pl.ewinp, ws + pl.dO, ws + pl.D, cpart, L,
                                                                      nb, ntask, ws + pl.dqkv)));
    LAUNCH(ctx, "attn_train_bwd", (at_dkv_kernel<<<cgrid, 256, 0, s>>>(b, ws + pl.ewin, ws + pl.dO, ws + pl.dOp, ws + pl.D, cpart, L,
                                                                       nb, ntask, ws + pl.dqkv)));
    LAUNCH(ctx, "attn_train_bwd", (at_de_kernel<<<cgrid, 256, 0, s>>>(b, ws + pl.ewin, ws + pl.dO, ws + pl.D, cpart, L, nb, ntask,
                                                                      ws + pl.depart)));
    }
    // rel_pos_emb gradient: sum the band slabs over (n, h) first (grouped, coalesced), then fold the band rows onto
    // the clamped table rows
    const int slab_elems = (2 * nb - 1) * 512;
    LAUNCH(ctx, "attn_train_reduce", (reduce_partials_kernel<<<(slab_elems + 63) / 64, 1024, 0, s>>>(ws + pl.depart, N * 4,
                                                                                                    slab_elems, ws + pl.dewin)));
    const int rel_elems = (2 * max_pos + 1) * 16;
    LAUNCH(ctx, "attn_train_reduce", (at_de_scatter_kernel<<<(rel_elems + 255) / 256, 256, 0, s>>>(ws + pl.dewin, L, nb, max_pos,
                                                                                                  grad.rel)));
    // projections + LayerNorm
#if TRAIN_X3
    at_x3_qkv_bwd(ctx, x, ws + pl.dqkv, M, ws + pl.wqkvt, p.ln_w, p.ln_b, dres, dx, ws + pl.xn, ws + pl.g1, ws + pl.dxn);
#else
    LAUNCH(ctx, "attn_train_bwd", (at_qkv_bwd_kernel<<<grid, 256, 0, s>>>(x, ws + pl.dqkv, M, ws + pl.wqkvt, p.ln_w, p.ln_b,
                                                                          dres, dx, ws + pl.xn, ws + pl.g1, ws + pl.dxn)));
#endif
    wgrad_partial64(ctx, "attn_train_wgrad", ws + pl.dqkv, ws + pl.xn, M, 192, 64, ws + pl.wpart, wg_split(3));
    // [192,64] = [dW_q ; dW_kv]: straight into the gradient tensors when they are adjacent (views of one flat bucket), else split
    const bool adjacent = grad.wkv == grad.wq + 4096;
    LAUNCH(ctx, "attn_train_reduce", (reduce_partials_kernel<<<192, 1024, 0, s>>>(ws + pl.wpart, wg_split(3), 12288,
                                                                                adjacent ? grad.wq : ws + pl.raw)));
    if (!adjacent) {
        hipMemcpyAsync(grad.wq, ws + pl.raw, 4096 * sizeof(float), hipMemcpyDeviceToDevice, s);
        hipMemcpyAsync(grad.wkv, ws + pl.raw + 4096, 8192 * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
#if TRAIN_X3
    const ColsumJobs jobs{{ws + pl.g1, ws + pl.dxn}, {grad.ln_w, grad.ln_b}, {64, 64},
                          {(M + 15) / 16, (M + 15) / 16}};                        // g1 / dxn: per-tile partial sums (dbo: above)
    colsum_batch(ctx, "attn_train_reduce", jobs, 2, M, cpart);
#else
    const ColsumJobs jobs{{ws + pl.dout, ws + pl.g1, ws + pl.dxn}, {grad.bo, grad.ln_w, grad.ln_b}, {64, 64, 64},
                          {0, (M + 15) / 16, (M + 15) / 16}};                     // g1 / dxn: per-tile partial sums
    colsum_batch(ctx, "attn_train_reduce", jobs, 3, M, cpart);
#endif
}

// ---------------------------------------------------------------------------------
// glue of ConformerBlock.forward in train mode (conformer.py:216-222): residual add, and the closing
// post_norm LayerNorm(64) with its backward (same wave-level statistics as the branch kernels).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        stg4(out + 4 * i, ldg4(a + 4 * i) + ldg4(b + 4 * i));
}
// acc += b: the accumulate form (out == a).  Its own kernel because add_kernel promises the compiler that its three
// pointers do not alias (__restrict__), which an in-place call would break.
__global__ __launch_bounds__(256) void add_inplace_kernel(float* acc, const float* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        stg4(acc + 4 * i, ldg4(acc + 4 * i) + ldg4(b + 4 * i));
}
void launch_add(LaunchCtx ctx, const float* a, const float* b, float* out, long n) {
    const long n4 = n / 4, want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
    if (a == out)
        LAUNCH(ctx, "residual_add", (add_inplace_kernel<<<grid, 256, 0, ctx.stream>>>(out, b, n4)));
    else if (b == out)
        LAUNCH(ctx, "residual_add", (add_inplace_kernel<<<grid, 256, 0, ctx.stream>>>(out, a, n4)));
    else
        LAUNCH(ctx, "residual_add", (add_kernel<<<grid, 256, 0, ctx.stream>>>(a, b, out, n4)));
}

__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const float* __restrict__ x, long M,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ res, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, gamma, beta, xh, xn, rstd, row);
    if (ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 v = xn[0][kb];
            if (res) v = v + ldg4(res + row * 64 + 16 * kb + 4 * g);
            stg4(y + row * 64 + 16 * kb + 4 * g, v);
        }
    }
}

__global__ __launch_bounds__(256) void ln_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long M,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ dx,
                                                           float* __restrict__ g1) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long t0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (t0 >= M) return;
    f32x4 xh[4], xn[1][4];
    float rstd;
    long row;
    const bool ok = cm_load_norm(x, M, t0, c, g, gamma, beta, xh, xn, rstd, row);
    f32x4 dyv[4], dxh[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        dyv[kb] = ldg4(dy + row * 64 + 16 * kb + 4 * g);
        dxh[kb] = dyv[kb] * ldg4(gamma + 16 * kb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1 += dxh[kb][r];
            s2 = fmaf(dxh[kb][r], xh[kb][r], s2);
        }
    }
    const float mu1 = red_g_sum(s1) * (1.0f / 64.0f), mu2 = red_g_sum(s2) * (1.0f / 64.0f);
    if (ok) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
            stg4(dx + row * 64 + 16 * kb + 4 * g, (dxh[kb] - splat4(mu1) - xh[kb] * splat4(mu2)) * splat4(rstd));
    }
    f32x4 ca[4], cb[4];                                             // dgamma = colsum(dy xhat), dbeta = colsum(dy)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        ca[kb] = ok ? dyv[kb] * xh[kb] : splat4(0.f);
        cb[kb] = ok ? dyv[kb] : splat4(0.f);
    }
    ln_tile_colsums(ca, cb, c, g, t0 >> 4, g1, g1 + ((M + 15) / 16) * 64);
}

size_t ln_train_ws_floats(long M) { return (size_t)M * 64 + (size_t)2 * FFN_COLSUM_BLOCKS * 256; }

void launch_ln_train_forward(LaunchCtx ctx, const float* x, long M, const float* gamma, const float* beta,
                             const float* res, float* y) {
    LAUNCH(ctx, "ln_train", (ln_train_fwd_kernel<<<(unsigned)((M + 63) / 64), 256, 0, ctx.stream>>>(x, M, gamma, beta, res, y)));
}

void launch_ln_train_backward(LaunchCtx ctx, const float* x, const float* dy, long M, const float* gamma,
                              const float* beta, float* dx, float* dgamma, float* dbeta, float* ws) {
    hipStream_t s = ctx.stream;
    float* g1 = ws;
    float* cpart = ws + (size_t)M * 64;
    LAUNCH(ctx, "ln_train", (ln_train_bwd_kernel<<<(unsigned)((M + 63) / 64), 256, 0, s>>>(x, dy, M, gamma, beta, dx, g1)));
    const long trows = (M + 15) / 16;                              // per-tile partial sums of dy xhat | dy
    const ColsumJobs jobs{{g1, g1 + trows * 64}, {dgamma, dbeta}, {64, 64}, {trows, trows}};
    colsum_batch(ctx, "ln_train", jobs, 2, M, cpart);
}

// [B, A, C, 64] -> [B, C, A, 64]: the layout flip between the time-axis and frequency-axis sequences of a TSCB
// (the reference's permute(...).contiguous(), generator.py:94,96) on channels-last rows of 256 B; `add` (optional, laid
// out like `in`) is summed in on the way: the residual `x_t = time_conformer(x_t) + x_t` of :95 rides on the flip of :96
__global__ __launch_bounds__(256) void swap_axes_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                        float* __restrict__ out, int A, int C, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int q = (int)(i & 15);
        long r = i >> 4;                       // output row (b, c, a)
        const int a = (int)(r % A);
        r /= A;
        const int c = (int)(r % C);
        const long b = r / C;
        const long src = ((((b * A + a) * C + c) << 4) + q) * 4;
        f32x4 v = ldg4(in + src);
        if (add) v = v + ldg4(add + src);
        stg4(out + 4 * i, v);
    }
}
// ---------------------------------------------------------------------------------
// Keep-masks of the step's Dropout layers as bytes (1 = keep, probability `keep`), drawn by Philox4x32-10 - the
// counter-based generator torch's own dropout uses - from (seed, offset) in DEVICE memory: counter = offset + the
// index of a 16-byte group, each group = two Philox blocks = sixteen 16-bit uniforms compared with
// round(keep * 65536).  The offset is advanced by a one-thread kernel after the draw, so a captured graph replays with
// fresh masks and nothing in the launches changes from step to step.  (torch's bernoulli_ on the 5.8 GB of masks of a
// 32-clip step took 4.7 ms; this writes them at the HBM rate.)   state = {seed, offset} as two 64-bit words.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ out, long ngroups, unsigned thresh,
                                                           const unsigned long long* __restrict__ state) {
    const unsigned long long seed = state[0], offset = state[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ngroups; i += (long)gridDim.x * 256) {
        const unsigned long long ctr = offset + (unsigned long long)i;
        unsigned w[4];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)half, 0u};
            philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
            for (int j = 0; j < 2; ++j) {                          // two 32-bit words -> four mask bytes
                const unsigned a = c[2 * j], b = c[2 * j + 1];
                w[2 * half + j] = ((a & 0xffffu) < thresh ? 1u : 0u) | ((a >> 16) < thresh ? 0x100u : 0u) |
                                  ((b & 0xffffu) < thresh ? 0x10000u : 0u) | ((b >> 16) < thresh ? 0x1000000u : 0u);
            }
        }
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        reinterpret_cast<u32x4_*>(out)[i] = u32x4_{w[0], w[1], w[2], w[3]};
    }
}
__global__ void dropout_tick_kernel(unsigned long long* __restrict__ state, unsigned long long ngroups) { state[1] += ngroups; }
void launch_dropout_masks(LaunchCtx ctx, unsigned char* out, long nbytes, float keep, unsigned long long* state) {
    const long ngroups = nbytes / 16;
    double th = (double)keep * 65536.0 + 0.5;
    const unsigned thresh = th < 0.0 ? 0u : (th > 65536.0 ? 65536u : (unsigned)th);
    const unsigned grid = (unsigned)((ngroups + 255) / 256 < 8192 ? (ngroups + 255) / 256 : 8192);
    LAUNCH(ctx, "dropout_masks", (dropout_mask_kernel<<<grid, 256, 0, ctx.stream>>>(out, ngroups, thresh, state)));
    LAUNCH(ctx, "dropout_masks", (dropout_tick_kernel<<<1, 1, 0, ctx.stream>>>(state, (unsigned long long)ngroups)));
}

void launch_swap_axes(LaunchCtx ctx, const float* in, const float* add, float* out, int B, int A, int C) {
    const long total4 = (long)B * A * C * 16, want = (total4 + 255) / 256;
    LAUNCH(ctx, "swap_axes", (swap_axes_kernel<<<(unsigned)(want < 8192 ? (want > 0 ? want : 1) : 8192), 256, 0, ctx.stream>>>(
                                 in, add, out, A, C, total4)));
}

// =====================================================================================
// Training-mode DilatedDenseNet (fourth backward slice of SURVEY.md N2; generator.py:6-47): four layers of
//   pad(top = dil, left/right = 1) -> Conv2d(64 (i+1) -> 64, kernel (2,3), dilation (2^i, 1)) -> InstanceNorm2d(affine)
//   -> PReLU(64) -> concat newest-first
// on channels-last activations [B, T, F, 64].  The concat is never materialised: layer i reads "slots" a_0 = x,
// a_s = output of layer s-1, and the reference's newest-first channel order is an index map into the weight
// (slot s <-> input channels [64 (i - s), 64 (i - s + 1)) of conv{i+1}).  Convolutions are the per-position fp32-MFMA
// chain (16 positions per wave, one 64x64 A image per (slot, tap), B fragments = the tap-shifted rows, zero outside
// the plane); dgrad is the same chain with the transposed images and the opposite shifts, accumulated into per-slot
// gradient planes; wgrad is a split-K token contraction with a shifted second operand.  InstanceNorm statistics and
// its backward sums are per-(clip, channel) two-pass reductions in fp64.  A conv bias in front of an InstanceNorm
// has an exactly zero gradient; it is still computed (column sum of dz) so that the ten "bias" tensors are written.
// =====================================================================================
#define DB_NCH 256                     // position chunks per clip of the per-(b, c) reductions: B x 256 blocks fill the chip
                                       // (32 chunks = 128 blocks at batch 4 ran these streaming sums at 0.2 TB/s)
#define MT_NCH 32                      // the single-channel mask head's planes are 64 x smaller
#define DB_MAXBLK 2048                 // blocks of db_in_bwd_kernel = per-block |dz| maxima it leaves for the dgrad scale

struct DbSlots { const float* p[5]; };

__device__ __forceinline__ float wave_max_all(float v) {        // maximum over the 64 lanes
    v = fmaxf(v, dpp_perm<0xB1>(v));
    v = fmaxf(v, dpp_perm<0x4E>(v));
    v = fmaxf(v, dpp_perm<0x141>(v));
    v = fmaxf(v, dpp_perm<0x140>(v));
    return red_g_max(v);
}

// (b, t, f) of the 16 positions of a wave; `src` = flat row of the position shifted by (dt, df) or -1 outside the plane
struct DbPos { long m; int b, t, f; bool ok; };
__device__ __forceinline__ DbPos db_pos(long m0, int c, long M, int T, int F) {
    DbPos q;
    q.m = m0 + c;
    q.ok = q.m < M;
    const long mm = q.ok ? q.m : M - 1;
    const long tf = (long)T * F;
    q.b = (int)(mm / tf);
    const int rem = (int)(mm - (long)q.b * tf);
    q.t = rem / F;
    q.f = rem - q.t * F;
    return q;
}
__device__ __forceinline__ long db_shift(const DbPos& q, int dt, int df, int T, int F) {
    const int t = q.t + dt, f = q.f + df;
    return (q.ok && t >= 0 && t < T && f >= 0 && f < F) ? ((long)q.b * T + t) * F + f : -1;
}

// z[m][co] = bias[co] + sum_{slot, tap} W_{slot,tap} a_slot[m shifted by tap]      images: [slot][tap][4 ob][4 kb][64][4]
template <int NS>
__global__ __launch_bounds__(256) void db_conv_fwd_kernel(DbSlots in, const float* __restrict__ wimg,
                                                          const float* __restrict__ bias, int B, int T, int F, int dil,
                                                          float* __restrict__ z) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long M = (long)B * T * F;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (m0 >= M) return;
    const DbPos q = db_pos(m0, c, M, T, F);
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = ldg4(bias + 16 * ob + 4 * g);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int tap = 0; tap < 6; ++tap) {
            const int kt = tap / 3, kf = tap - 3 * kt;
            const long src = db_shift(q, (kt - 1) * dil, kf - 1, T, F);
            f32x4 bfr[1][4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                bfr[0][kb] = ldg4(in.p[s] + (src < 0 ? 0 : src) * 64 + 16 * kb + 4 * g);
                if (src < 0) bfr[0][kb] = splat4(0.f);
            }
            const float* wp = wimg + ((long)(s * 6 + tap) * 16) * 256 + lane * 4;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                f32x4 a1[1] = {acc[ob]};
                lin_acc<4, 1>(wp + (long)ob * 4 * 256, bfr, a1);
                acc[ob] = a1[0];
            }
        }
    }
    if (q.ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) stg4(z + q.m * 64 + 16 * ob + 4 * g, acc[ob]);
    }
}

// per-(clip, chunk, channel) sums over positions: MODE 0: (sum z, sum z^2) of z;  MODE 1 (backward): from z, the
// incoming gradient ga and the layer's statistics / affine / slope: dn = ga * PReLU'(n) (written over ga) and
// (sum dn, sum dn zhat, sum ga n [n < 0])
// gin (MODE 1, optional): the incoming gradient is read from gin instead of ga (out of place: the dense block's last layer
// reads the caller's dy directly - no copy of the plane into the workspace first)
template <int MODE>
__global__ __launch_bounds__(256) void db_sums_kernel(const float* __restrict__ z, float* __restrict__ ga, int P,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ alpha, float* __restrict__ partial,
                                                      const float* __restrict__ gin = nullptr) {
    __shared__ float red[4][64][3];
    const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int per = (P + DB_NCH - 1) / DB_NCH, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float mu = 0.f, rs = 0.f, gm = 0.f, bt = 0.f, al = 0.f;
    if (MODE == 1) { mu = mean[b * 64 + c]; rs = rstd[b * 64 + c]; gm = gamma[c]; bt = beta[c]; al = alpha[c]; }
    for (int p = p0 + sub; p < p1; p += 4) {
        const long i = ((long)b * P + p) * 64 + c;
        const float zv = z[i];
        if (MODE == 0) {
            s0 += zv;
            s1 = fmaf(zv, zv, s1);
        } else {
            const float zh = (zv - mu) * rs, n = zh * gm + bt, gv = gin ? gin[i] : ga[i];
            const float dn = n < 0.f ? gv * al : gv;
            ga[i] = dn;
            s0 += dn;
            s1 = fmaf(dn, zh, s1);
            s2 += n < 0.f ? gv * n : 0.f;
        }
    }
    red[sub][c][0] = s0; red[sub][c][1] = s1; red[sub][c][2] = s2;
    __syncthreads();
    if (sub == 0) {
        const long o = (((long)b * DB_NCH + chunk) * 64 + c) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) partial[o + k] = (red[0][c][k] + red[1][c][k]) + (red[2][c][k] + red[3][c][k]);
    }
}

// fp64 sum of up to 256 chunk partials per thread block: thread k holds chunk k, fixed-shape tree
__device__ __forceinline__ double db_tree_sum(double v, double* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (t < d) red[t] += red[t + d];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
// forward statistics: one block per (b, c) adds the DB_NCH chunk partials in fp64 -> mean, rstd
__global__ __launch_bounds__(256) void db_stats_finalize_kernel(const float* __restrict__ partial, int B, double count,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[256];
    const int i = blockIdx.x, b = i >> 6, c = i & 63, k = threadIdx.x;
    const long o = (((long)b * DB_NCH + k) * 64 + c) * 3;
    const double s1 = db_tree_sum(k < DB_NCH ? (double)partial[o] : 0.0, red);
    const double s2 = db_tree_sum(k < DB_NCH ? (double)partial[o + 1] : 0.0, red);
    if (k == 0) {
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        var = var > 0.0 ? var : 0.0;
        mean[i] = (float)mu;
        rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// the same from the convolution kernel's own per-(clip, tile) partial sums [b][ntiles][64][2] (conv3x_kernel's epilogue,
// the inference path's InstanceNorm statistics): no separate pass over z
__global__ __launch_bounds__(256) void db_stats_finalize_tiles_kernel(const float* __restrict__ partial, int ntiles,
                                                                      double count, float* __restrict__ mean,
                                                                      float* __restrict__ rstd) {
    __shared__ double red[256];
    const int i = blockIdx.x, b = i >> 6, c = i & 63, k = threadIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int t = k; t < ntiles; t += 256) {
        const long o = (((long)b * ntiles + t) * 64 + c) * 2;
        a1 += (double)partial[o];
        a2 += (double)partial[o + 1];
    }
    const double s1 = db_tree_sum(a1, red);
    const double s2 = db_tree_sum(a2, red);
    if (k == 0) {
        const double mu = s1 / count;
        double var = s2 / count - mu * mu;
        var = var > 0.0 ? var : 0.0;
        mean[i] = (float)mu;
        rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// a = PReLU(InstanceNorm(z))
__global__ __launch_bounds__(256) void db_norm_prelu_kernel(const float* __restrict__ z, long total, int P,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ alpha, float* __restrict__ a) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i & 63);
        const long bc = (i >> 6) / P * 64 + c;
        const float n = (z[i] - mean[bc]) * rstd[bc] * gamma[c] + beta[c];
        a[i] = n >= 0.f ? n : alpha[c] * n;
    }
}

// backward means per (b, c) and the per-channel parameter gradients (summed over clips in clip order): one block per
// channel; the DB_NCH = 256 chunk partials of a clip are added in fp64 by a butterfly inside each wave and the four waves
// in wave order - one barrier per FOUR clips (the shared fixed-shape tree of db_tree_sum took 27 barriers per clip: 99 us per
// launch at 32 clips, fifteen launches per step)
__global__ __launch_bounds__(1024) void db_bwd_finalize_kernel(const float* __restrict__ partial, int B, double count,
                                                               float* __restrict__ m1, float* __restrict__ m2,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dalpha) {
    static_assert(DB_NCH == 256, "one chunk per thread");
    // 1024 threads = FOUR clips per trip (clip lane q = wave >> 2, 256 chunk threads each): a barrier per four clips; the
    // per-channel sums still add the clips in clip order (thread 0 walks the four results of a trip in order)
    __shared__ double red[2][4][3][4];                            // [trip parity][clip lane][sum][wave of the lane]
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, q = tid >> 8, k = tid & 255;
    double g = 0.0, bsum = 0.0, a = 0.0;
    for (int b0 = 0; b0 < B; b0 += 4) {
        const int b = b0 + q;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        if (b < B) {
            const long o = (((long)b * DB_NCH + k) * 64 + c) * 3;
            v0 = (double)partial[o]; v1 = (double)partial[o + 1]; v2 = (double)partial[o + 2];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            v0 += __shfl_xor(v0, d);
            v1 += __shfl_xor(v1, d);
            v2 += __shfl_xor(v2, d);
        }
        double (*r)[3][4] = red[(b0 >> 2) & 1];                   // (alternating buffers: the next trip's writes need no barrier)
        if (lane == 0) { r[q][0][wv] = v0; r[q][1][wv] = v1; r[q][2][wv] = v2; }
        __syncthreads();
        if (tid == 0) {
            for (int qq = 0; qq < 4 && b0 + qq < B; ++qq) {
                const double s0 = (r[qq][0][0] + r[qq][0][1]) + (r[qq][0][2] + r[qq][0][3]);
                const double s1 = (r[qq][1][0] + r[qq][1][1]) + (r[qq][1][2] + r[qq][1][3]);
                const double s2 = (r[qq][2][0] + r[qq][2][1]) + (r[qq][2][2] + r[qq][2][3]);
                m1[(b0 + qq) * 64 + c] = (float)(s0 / count);
                m2[(b0 + qq) * 64 + c] = (float)(s1 / count);
                bsum += s0; g += s1; a += s2;
            }
        }
    }
    if (tid == 0) { dgamma[c] = (float)g; dbeta[c] = (float)bsum; dalpha[c] = (float)a; }
}

// dz = gamma rstd (dn - mean(dn) - zhat mean(dn zhat)), in place on dn
// absmax (may be NULL): [gridDim.x] per-block maxima of |dz| (no atomics: 8192 of them on one word cost 70 us per
// launch), reduced by db_dgrad_setup_kernel - the split-f16 data gradient scales dz by the plane's maximum
__global__ __launch_bounds__(256) void db_in_bwd_kernel(float* __restrict__ dn, const float* __restrict__ z, long total, int P,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ m1,
                                                        const float* __restrict__ m2, float* __restrict__ absmax = nullptr) {
    __shared__ float wmax[4];
    float mx = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i & 63);
        const long bc = (i >> 6) / P * 64 + c;
        const float zh = (z[i] - mean[bc]) * rstd[bc];
        const float v = gamma[c] * rstd[bc] * (dn[i] - m1[bc] - zh * m2[bc]);
        dn[i] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    if (absmax) {
        mx = wave_max_all(mx);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) absmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    }
}

// ga_slot[m][ci] += sum_tap W_{slot,tap}^T dz[m shifted by -tap]                images (transposed): [tap][4][4][64][4]
__global__ __launch_bounds__(256) void db_conv_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wimgT,
                                                            int B, int T, int F, int dil, float* __restrict__ ga) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long M = (long)B * T * F;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (m0 >= M) return;
    const DbPos q = db_pos(m0, c, M, T, F);
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = splat4(0.f);
#pragma unroll
    for (int tap = 0; tap < 6; ++tap) {
        const int kt = tap / 3, kf = tap - 3 * kt;
        const long src = db_shift(q, -(kt - 1) * dil, -(kf - 1), T, F);     // the output position that read this input
        f32x4 bfr[1][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bfr[0][kb] = ldg4(dz + (src < 0 ? 0 : src) * 64 + 16 * kb + 4 * g);
            if (src < 0) bfr[0][kb] = splat4(0.f);
        }
        const float* wp = wimgT + ((long)tap * 16) * 256 + lane * 4;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 a1[1] = {acc[ob]};
            lin_acc<4, 1>(wp + (long)ob * 4 * 256, bfr, a1);
            acc[ob] = a1[0];
        }
    }
    if (q.ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            float* p = ga + q.m * 64 + 16 * ob + 4 * g;
            stg4(p, ldg4(p) + acc[ob]);
        }
    }
}

// partial[tap][s][co][ci] = sum over the s-th range of tokens of dz[m][co] * a[m shifted by tap][ci]
// One block per (tap, token range) computes the whole 64 x 64 tile (every operand row is loaded once per tap instead
// of once per 16 output channels), and the (clip, t, f) coordinates of a lane's four tokens come from ONE 32-bit
// division per step plus carries - the first version spent as many VALU cycles on 64-bit divisions as the matrix pipe
// spent on the products.
#define DB_WG_SPLIT 256
__global__ __launch_bounds__(256) void db_conv_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ a, int B,
                                                            int T, int F, int dil, float* __restrict__ partial) {
    __shared__ float red[2][64 * 64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int tap = blockIdx.x, s = blockIdx.y;
    const int kt = tap / 3, kf = tap - 3 * kt, dt = (kt - 1) * dil, df = kf - 1;
    const unsigned M = (unsigned)B * T * F, tf = (unsigned)T * F;
    const unsigned steps = (M + 15) / 16, per = (steps + DB_WG_SPLIT - 1) / DB_WG_SPLIT;
    const unsigned st0 = s * per, st1 = st0 + per < steps ? st0 + per : steps;
    f32x4 acc[4][4];                                  // [ib][jb]
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = splat4(0.f);
    for (unsigned st = st0 + wv; st < st1; st += 4) {
        const unsigned m0 = st * 16 + 4 * g;
        unsigned bb = m0 / tf;
        const unsigned rem = m0 - bb * tf;
        int t = (int)(rem / (unsigned)F), f = (int)(rem - (unsigned)t * F);
        f32x4 av[4], bv[4];                           // [ib][r], [jb][r]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned m = m0 + r;
            const bool ok = m < M;
            const int ts = t + dt, fs = f + df;
            const bool inb = ok && ts >= 0 && ts < T && fs >= 0 && fs < F;
            const long src = inb ? ((long)bb * T + ts) * F + fs : 0;
            const long mm = ok ? m : M - 1;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) av[ib][r] = ok ? dz[mm * 64 + 16 * ib + c] : 0.f;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float v = a[src * 64 + 16 * jb + c];
                bv[jb][r] = inb ? v : 0.f;
            }
            if (++f == F) { f = 0; if (++t == T) { t = 0; ++bb; } }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = mfma16(av[ib][r], bv[jb][r], acc[ib][jb]);
    }
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
    if (wv >= 2) put(red[wv - 2]);
    __syncthreads();
    if (wv < 2) add(red[wv]);
    __syncthreads();
    if (wv == 1) put(red[0]);
    __syncthreads();
    if (wv == 0) {
        add(red[0]);
        float* out = partial + ((long)tap * DB_WG_SPLIT + s) * 4096;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * ib + 4 * g + r) * 64 + 16 * jb + c] = acc[ib][jb][r];
    }
}
// dW[co][cbase + ci][tap] = sum_s partial[tap][s][co][ci]: a block owns 64 consecutive (co, ci) of one tap, its four
// thread groups add every fourth slab (coalesced), combined in group order
__global__ __launch_bounds__(256) void db_wgrad_scatter_kernel(const float* __restrict__ partial, int Cin, int cbase,
                                                               float* __restrict__ dW) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int tap = blockIdx.x >> 6, e = (blockIdx.x & 63) * 64 + col;
    float s0 = 0.f, s1 = 0.f;
    for (int k = grp; k < DB_WG_SPLIT; k += 8) {
        s0 += partial[((long)tap * DB_WG_SPLIT + k) * 4096 + e];
        s1 += partial[((long)tap * DB_WG_SPLIT + k + 4) * 4096 + e];
    }
    red[grp][col] = s0 + s1;
    __syncthreads();
    if (grp == 0) {
        const float sum = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        const int co = e >> 6, ci = e & 63;
        dW[((long)co * Cin + cbase + ci) * 6 + tap] = sum;
    }
}

struct DbPlan { size_t img, imgT, a, z, ga, mean, rstd, part, m1, m2, wpart, cpart, total; };
static DbPlan db_plan(int B, int T, int F) {
    DbPlan p;
    const size_t M = (size_t)B * T * F;
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    p.img = take(60 * 4096); p.imgT = take(60 * 4096);
    p.a = take(4 * M * 64);            // a_1 .. a_4 (a_0 = x is the caller's)
    p.z = take(4 * M * 64);
    p.ga = take(4 * M * 64);           // gradients w.r.t. a_1 .. a_4 (a_0's is the caller's dx)
    p.mean = take((size_t)4 * B * 64); p.rstd = take((size_t)4 * B * 64);
    {
        const size_t chunks = (size_t)B * DB_NCH * 64 * 3, tiles = (size_t)B * conv3x_ntiles(T, F, 64) * 128;
        p.part = take(chunks > tiles ? chunks : tiles);     // chunk partials, or the convolution kernel's tile partials
    }
    p.m1 = take((size_t)B * 64); p.m2 = take((size_t)B * 64);
    p.wpart = take((size_t)6 * DB_WG_SPLIT * 4096);
    p.cpart = take((size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256);
    p.total = cur;
    return p;
}
size_t dense_train_ws_floats(int B, int T, int F) { return db_plan(B, T, F).total; }

// image index of (layer i, slot s): layers own 6 (i+1) images each, in slot-major order
static int db_img_index(int i, int s) { return 6 * (i * (i + 1) / 2 + s); }

// all 6 (i+1) tap images of layer i, plain (blockIdx.z = 0) and transposed (1), in one launch
__global__ void db_pack_layer_kernel(const float* __restrict__ w, int i, float* __restrict__ img, float* __restrict__ imgT,
                                     int zbase = 0) {       // zbase = 1 with gridDim.z = 1: the transposed images only
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // element of one 64x64 image
    if (e >= 4096) return;
    const int st = blockIdx.y, s = st / 6, tap = st - 6 * s, Cin = 64 * (i + 1);
    const int cbase = 64 * (i - s);                                 // newest-first concat: slot s sits at channel block i - s
    const int r = e & 3, lane = (e >> 2) & 63, blk = e >> 8, rb = blk >> 2, kb = blk & 3;
    const int row = 16 * rb + (lane & 15), col = 16 * kb + 4 * (lane >> 4) + r;
    const int tr = blockIdx.z + zbase;
    const int co = tr ? col : row, ci = tr ? row : col;
    const float v = w[((long)co * Cin + cbase + ci) * 6 + tap];
    (tr ? imgT : img)[(long)st * 4096 + e] = v;
}

#if TRAIN_X3
// The forward convolution of the training step IS the inference kernel (conv3x_kernel<2, 64>, conv_x3.hip: 256-position
// tiles staged through LDS, split-f16 products): the training plane layout, the causal time taps and the raw output
// are the same; only the operand image has to be rebuilt from the raw weight after every optimiser step.
// x3 conv image of layer i: [32-channel chunk in SLOT order][tap = kt * 3 + kf][4 cb][hi | lo][64][8 halfs], lane (c, g)
// slot e <-> w[co = 16 cb + c][slot-order channel 32 chunk + 16 (e >> 2) + 4 g + (e & 3)][tap]   (api.hip: x3_conv_image)
__global__ void db_pack_x3_layer_kernel(const float* __restrict__ w, int i, _Float16* __restrict__ img) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // (chunk, tap, cb, lane)
    const int nch = 2 * (i + 1), Cin = 64 * (i + 1);
    if (t >= nch * 6 * 4 * 64) return;
    const int lane = t & 63, cb = (t >> 6) & 3, rest = t >> 8, tap = rest % 6, chunk = rest / 6;
    const int co = 16 * cb + (lane & 15), s = chunk >> 1;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cs = 32 * (chunk & 1) + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);       // channel inside slot s
        const float v = w[((long)co * Cin + 64 * (i - s) + cs) * 6 + tap];                 // newest-first concat order
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
    _Float16* o = img + ((long)(chunk * 6 + tap) * 4 + cb) * 1024 + lane * 8;
    *reinterpret_cast<f16x8*>(o) = hi;
    *reinterpret_cast<f16x8*>(o + 512) = lo;
}
// halfs offset of layer i's x3 image inside the (re-used) fp32 image area: 2 (i + 1) * 6 * 4 * 1024 halfs per layer
static long db_x3_img_off(int i) { return (long)(i * (i + 1)) * 24576; }

// The DATA GRADIENT of the same convolution is that kernel again (launch_conv3_x3_dgrad): on the time-reversed plane the
// anti-causal tap of the transposed conv (dz at t + dil) is the causal one, the frequency taps are mirrored in the
// image, the contraction runs over layer i's 64 output channels and the 64 results accumulate into slot s's gradient.
// Image of (layer i, slot s): [2 chunks of 32 dz channels][tap = kt * 3 + kfc][4 cb][hi | lo][64][8], lane (c, g) slot e
// <-> w[co = 32 chunk + 16 (e >> 2) + 4 g + (e & 3)][64 (i - s) + 16 cb + c][kt][kf = 2 - kfc]
__global__ void db_pack_x3_dgrad_kernel(const float* __restrict__ w, int i, _Float16* __restrict__ img) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // (slot, chunk, tap, cb, lane)
    const int Cin = 64 * (i + 1);
    if (t >= (i + 1) * 2 * 6 * 4 * 64) return;
    const int lane = t & 63, cb = (t >> 6) & 3, rest = t >> 8, tap = rest % 6, chunk = (rest / 6) & 1, s = rest / 12;
    const int ci = 64 * (i - s) + 16 * cb + (lane & 15), kt = tap / 3, kf = 2 - (tap - 3 * kt);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = 32 * chunk + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        const float v = w[((long)co * Cin + ci) * 6 + kt * 3 + kf];
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
    _Float16* o = img + ((long)((s * 2 + chunk) * 6 + tap) * 4 + cb) * 1024 + lane * 8;
    *reinterpret_cast<f16x8*>(o) = hi;
    *reinterpret_cast<f16x8*>(o + 512) = lo;
}
// halfs offset of (layer i, slot s) in the (re-used) transposed-image area: 49152 halfs per (i, s), layers in order
static long db_x3_dgrad_off(int i, int s) { return (long)(i * (i + 1) / 2 + s) * 49152; }

// scratch of the dgrad launches, carved out of the per-(b, c) partial-sum area (idle between a layer's InstanceNorm
// backward and the next layer's sums): [64] 1 / scale, [128..191] ones, [192..255] zeros, [256 ..] scale per (b, c),
// then zeros per (b, c), then the DB_MAXBLK per-block maxima of |dz| written by db_in_bwd_kernel.  One block.
__global__ __launch_bounds__(256) void db_dgrad_setup_kernel(float* __restrict__ sc, int B) {
    __shared__ float red[256];
    const float* slots = sc + 256 + 2 * B * 64;
    float m = 0.f;
    for (int k = threadIdx.x; k < DB_MAXBLK; k += 256) m = fmaxf(m, slots[k]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    const unsigned e = (__float_as_uint(red[0]) >> 23) & 0xffu;
    const bool ok = e > 0u && e < 254u;
    const float s = ok ? __uint_as_float((254u - e) << 23) : 1.0f, inv = ok ? __uint_as_float(e << 23) : 1.0f;
    if (threadIdx.x == 0) sc[64] = inv;
    if (threadIdx.x < 64) { sc[128 + threadIdx.x] = 1.0f; sc[192 + threadIdx.x] = 0.0f; }
    for (int i = threadIdx.x; i < B * 64; i += 256) { sc[256 + i] = s; sc[256 + B * 64 + i] = 0.0f; }
}
#endif

// the inference conv kernel steps its staging rows by 64 per load and assumes a plane row (F + 1 positions with the
// virtual zero column) is at least that long: true for every model size (F = 201 / 101 / 601 / 301); tiny test planes
// take the per-position fp32 kernel
static bool db_x3_forward(int F) { return TRAIN_X3 && F + 1 >= 64; }

static void db_pack_images(LaunchCtx ctx, const DenseTrainParams& p, float* ws, const DbPlan& pl, int F, bool pack = true) {
    if (!pack) return;
    for (int i = 0; i < 4; ++i) {
        const long off = (long)db_img_index(i, 0) * 4096;
#if TRAIN_X3
        if (db_x3_forward(F)) {
            // forward and data-gradient operand images in split-f16 form (exactly the space of the fp32 ones)
            const int nthr = 2 * (i + 1) * 6 * 4 * 64;
            LAUNCH(ctx, "dense_train_pack", (db_pack_x3_layer_kernel<<<(nthr + 255) / 256, 256, 0, ctx.stream>>>(
                                                p.conv_w[i], i, reinterpret_cast<_Float16*>(ws + pl.img) + db_x3_img_off(i))));
            LAUNCH(ctx, "dense_train_pack", (db_pack_x3_dgrad_kernel<<<(nthr + 255) / 256, 256, 0, ctx.stream>>>(
                                                p.conv_w[i], i, reinterpret_cast<_Float16*>(ws + pl.imgT) + db_x3_dgrad_off(i, 0))));
            (void)off;
            continue;
        }
#endif
        LAUNCH(ctx, "dense_train_pack", (db_pack_layer_kernel<<<dim3(16, 6 * (i + 1), 2), 256, 0, ctx.stream>>>(
                                            p.conv_w[i], i, ws + pl.img + off, ws + pl.imgT + off)));
    }
}

template <int NS>
static void db_launch_conv(LaunchCtx ctx, const DbSlots& in, const float* wimg, const float* bias, int B, int T, int F,
                           int dil, float* z) {
    const long M = (long)B * T * F;
    LAUNCH(ctx, "dense_train_fwd", (db_conv_fwd_kernel<NS><<<(unsigned)((M + 63) / 64), 256, 0, ctx.stream>>>(
                                       in, wimg, bias, B, T, F, dil, z)));
}

void launch_dense_train_forward(LaunchCtx ctx, const float* x, int B, int T, int F, const DenseTrainParams& p, float* y,
                                float* ws) {
    hipStream_t st = ctx.stream;
    const DbPlan pl = db_plan(B, T, F);
    const long M = (long)B * T * F;
    const int P = T * F;
    db_pack_images(ctx, p, ws, pl, F);
    DbSlots in{};
    in.p[0] = x;
    for (int i = 0; i < 4; ++i) {
        float* z = ws + pl.z + (size_t)i * M * 64;
        float* a = i == 3 ? y : ws + pl.a + (size_t)i * M * 64;
        const int dil = 1 << i;
        bool tile_stats = false;
#if TRAIN_X3
        if (db_x3_forward(F)) {
            ConvArgs ca{};                                  // slots arrive normalised + activated (or raw: slot 0): identity on load
            for (int sl = 0; sl <= i; ++sl) ca.in[sl] = in.p[sl];
            ca.nslots = i + 1;
            ca.bias = p.conv_b[i];
            ca.out = z;
            ca.partials = ws + pl.part;                     // per-(clip, tile) (sum, sum of squares): the InstanceNorm statistics
            ca.T = T; ca.F = F; ca.dil = dil; ca.mode = 0; ca.ntiles = conv3x_ntiles(T, F, 64);
            launch_conv3_x3(ctx, ca, reinterpret_cast<const _Float16*>(ws + pl.img) + db_x3_img_off(i), B, 2, 64);
            tile_stats = true;
        } else
#endif
        {
            const float* wimg = ws + pl.img + (long)db_img_index(i, 0) * 4096;
            switch (i) {
                case 0: db_launch_conv<1>(ctx, in, wimg, p.conv_b[i], B, T, F, dil, z); break;
                case 1: db_launch_conv<2>(ctx, in, wimg, p.conv_b[i], B, T, F, dil, z); break;
                case 2: db_launch_conv<3>(ctx, in, wimg, p.conv_b[i], B, T, F, dil, z); break;
                default: db_launch_conv<4>(ctx, in, wimg, p.conv_b[i], B, T, F, dil, z); break;
            }
        }
        float* mean = ws + pl.mean + (size_t)i * B * 64;
        float* rstd = ws + pl.rstd + (size_t)i * B * 64;
        if (tile_stats) {
            LAUNCH(ctx, "dense_train_fwd", (db_stats_finalize_tiles_kernel<<<B * 64, 256, 0, st>>>(
                                               ws + pl.part, conv3x_ntiles(T, F, 64), (double)P, mean, rstd)));
        } else {
            LAUNCH(ctx, "dense_train_fwd", (db_sums_kernel<0><<<dim3(B, DB_NCH), 256, 0, st>>>(z, nullptr, P, nullptr, nullptr,
                                                                                             nullptr, nullptr, nullptr,
                                                                                             ws + pl.part)));
            LAUNCH(ctx, "dense_train_fwd", (db_stats_finalize_kernel<<<B * 64, 256, 0, st>>>(ws + pl.part, B,
                                                                                                           (double)P, mean, rstd)));
        }
        LAUNCH(ctx, "dense_train_fwd", (db_norm_prelu_kernel<<<2048, 256, 0, st>>>(z, M * 64, P, mean, rstd, p.norm_w[i],
                                                                                   p.norm_b[i], p.prelu_w[i], a)));
        if (i < 3) in.p[i + 1] = a;
    }
}

void launch_dense_train_backward(LaunchCtx ctx, const float* x, const float* dy, int B, int T, int F,
                                 const DenseTrainParams& p, float* dx, const DenseTrainParams& grad, float* ws) {
    hipStream_t st = ctx.stream;
    const DbPlan pl = db_plan(B, T, F);
    const long M = (long)B * T * F;
    const int P = T * F;
    db_pack_images(ctx, p, ws, pl, F, false);
    // gradient planes: ga_0 IS the caller's dx (accumulated in place - dx must not alias x or dy), ga_1 .. ga_4 in the
    // workspace; ga_4 = dy is never materialised: layer 3 reads dy and writes its dn into ga_4 (db_sums_kernel gin).
    // (Two plane copies per backward - 2 x 0.53 GB moved at the encoder's shape - are gone this way.)
    auto ga = [&](int s) { return s == 0 ? dx : ws + pl.ga + (size_t)(s - 1) * M * 64; };
    auto aslot = [&](int s) -> const float* { return s == 0 ? x : ws + pl.a + (size_t)(s - 1) * M * 64; };
    hipMemsetAsync(dx, 0, (size_t)M * 64 * sizeof(float), st);                                   // ga_0
    hipMemsetAsync(ga(1), 0, (size_t)3 * M * 64 * sizeof(float), st);                            // ga_1 .. ga_3
    float* cpart = ws + pl.cpart;
    for (int i = 3; i >= 0; --i) {
        const float* z = ws + pl.z + (size_t)i * M * 64;
        float* g = ga(i + 1);                              // dL/da_{i+1} -> dn -> dz, in place
        const float* mean = ws + pl.mean + (size_t)i * B * 64;
        const float* rstd = ws + pl.rstd + (size_t)i * B * 64;
        LAUNCH(ctx, "dense_train_bwd", (db_sums_kernel<1><<<dim3(B, DB_NCH), 256, 0, st>>>(z, g, P, mean, rstd, p.norm_w[i],
                                                                                         p.norm_b[i], p.prelu_w[i],
                                                                                         ws + pl.part, i == 3 ? dy : nullptr)));
        LAUNCH(ctx, "dense_train_bwd", (db_bwd_finalize_kernel<<<64, 1024, 0, st>>>(ws + pl.part, B, (double)P, ws + pl.m1,
                                                                                ws + pl.m2, grad.norm_w[i], grad.norm_b[i],
                                                                                grad.prelu_w[i])));
        const bool x3d = db_x3_forward(F);
        float* sc = ws + pl.part;                          // dgrad scratch (see db_dgrad_setup_kernel)
        LAUNCH(ctx, "dense_train_bwd", (db_in_bwd_kernel<<<DB_MAXBLK, 256, 0, st>>>(g, z, M * 64, P, mean, rstd, p.norm_w[i],
                                                                                    ws + pl.m1, ws + pl.m2,
                                                                                    x3d ? sc + 256 + (size_t)2 * B * 64 : nullptr)));
        const int dil = 1 << i, Cin = 64 * (i + 1);
#if TRAIN_X3
        if (x3d) LAUNCH(ctx, "dense_train_bwd", (db_dgrad_setup_kernel<<<1, 256, 0, st>>>(sc, B)));
#endif
        for (int s = 0; s <= i; ++s) {
            // the split-f16 kernel addresses a plane with 32-bit byte offsets: planes of 4 GB and more (16.7 M positions,
            // 250 clips of the encoder's shape) take the fp32 kernel
            if (TRAIN_X3 && (long)M * 256 < (1L << 32)) {
                // (the conv-bias gradient of layer i = colsum of its dz rides on the first of its weight-gradient launches)
                launch_db_conv_wgrad_x3(ctx, g, aslot(s), B, T, F, dil, DB_WG_SPLIT, ws + pl.wpart, s == 0 ? cpart : nullptr);
                if (s == 0)
                    LAUNCH(ctx, "dense_train_reduce", (reduce_partials_kernel<<<4, 1024, 0, st>>>(cpart, DB_WG_SPLIT, 64,
                                                                                                  grad.conv_b[i])));
            } else
                LAUNCH(ctx, "dense_train_wgrad", (db_conv_wgrad_kernel<<<dim3(6, DB_WG_SPLIT), 256, 0, st>>>(
                                                     g, aslot(s), B, T, F, dil, ws + pl.wpart)));
            LAUNCH(ctx, "dense_train_reduce", (db_wgrad_scatter_kernel<<<6 * 64, 256, 0, st>>>(ws + pl.wpart, Cin, 64 * (i - s),
                                                                                               grad.conv_w[i])));
#if TRAIN_X3
            if (x3d) {
                ConvArgs ca{};                              // dz scaled by an exact power of two on load, ga(s) += result / scale
                ca.in[0] = g; ca.nscale[0] = sc + 256; ca.nshift[0] = sc + 256 + (size_t)B * 64; ca.nalpha[0] = sc + 128;
                ca.nslots = 1;
                ca.bias = sc + 192;
                ca.out = ga(s);
                ca.T = T; ca.F = F; ca.dil = dil; ca.mode = 0; ca.ntiles = conv3x_ntiles(T, F, 64);
                ca.revt = 1; ca.accum = 1; ca.oscale = sc + 64;
                launch_conv3_x3_dgrad(ctx, ca, reinterpret_cast<const _Float16*>(ws + pl.imgT) + db_x3_dgrad_off(i, s), B);
                continue;
            }
#endif
            LAUNCH(ctx, "dense_train_bwd", (db_conv_dgrad_kernel<<<(unsigned)((M + 63) / 64), 256, 0, st>>>(
                                               g, ws + pl.imgT + (long)db_img_index(i, s) * 4096, B, T, F, dil, ga(s))));
        }
    }
    // the four conv-bias gradients (column sums of the layers' dz planes, which stay untouched once written) in one batch
    if (!(TRAIN_X3 && (long)M * 256 < (1L << 32))) {              // (split-f16 build: done inside the weight-gradient launches)
        const ColsumJobs jobs{{ga(1), ga(2), ga(3), ga(4)}, {grad.conv_b[0], grad.conv_b[1], grad.conv_b[2], grad.conv_b[3]},
                              {64, 64, 64, 64}};
        colsum_batch(ctx, "dense_train_reduce", jobs, 4, M, cpart);
    }
}

// =====================================================================================
// Encoder / decoder heads (generator.py:50-69, 102-156), training mode.
//
// "Row conv": Conv2d(64 -> 64 NG, kernel (1, KW), stride (1, SF), left padding PL) on channels-last planes, on the
// per-position fp32-MFMA chain (tap images [kw][4 NG ob][4 kb]).  NG = 2 is the sub-pixel conv: its 128 output
// channels are written pixel-shuffled, channel 64 r + c of position f -> channel c of position 2 f + r of a plane
// that is twice as wide (SPConvTranspose2d, generator.py:102-119) - the shuffle is an index map, never a copy.
// =====================================================================================
struct RcGeom { int B, T, Fi, Fo, KW, SF, PL; };      // Fo = output positions per row (before the pixel shuffle)
#define RC_WG_SPLIT 256               // position ranges of the split-f16 weight gradient (sizes the slab buffers; >= FFN_WGRAD_SPLIT)
void launch_rc_wgrad_x3(LaunchCtx, int ng, const float* dz, const float* in, const int* gm7, int nsplit, float* partial,
                        float* colp);
bool launch_rc_dgrad_x3(LaunchCtx, int ng, const float* dz, const float* wraw, const int* gm7, float* din);
// the data gradient on split products (train_x3.hip) when the build and the geometry allow; CMGAN_RC_DGRAD_X3=0: A/B
static bool rc_dgrad_x3(LaunchCtx ctx, int ng, const float* dz, const float* wraw, const RcGeom& gm, float* din) {
#if TRAIN_X3
    static const bool k_on = env_knob("CMGAN_RC_DGRAD_X3", 1, 0, 1) != 0;
    const int gm7[7] = {gm.B, gm.T, gm.Fi, gm.Fo, gm.KW, gm.SF, gm.PL};
    return k_on && launch_rc_dgrad_x3(ctx, ng, dz, wraw, gm7, din);
#else
    (void)ctx; (void)ng; (void)dz; (void)wraw; (void)gm; (void)din;
    return false;
#endif
}

template <int NG>
__global__ __launch_bounds__(256) void rc_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wimg,
                                                     const float* __restrict__ bias, RcGeom gm, float* __restrict__ z) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long Mo = (long)gm.B * gm.T * gm.Fo;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (m0 >= Mo) return;
    const long m = m0 + c;
    const bool ok = m < Mo;
    const long mm = ok ? m : Mo - 1;
    const long bt = mm / gm.Fo;
    const int fo = (int)(mm - bt * gm.Fo);
    f32x4 acc[4 * NG];
#pragma unroll
    for (int ob = 0; ob < 4 * NG; ++ob) acc[ob] = ldg4(bias + 16 * ob + 4 * g);
    for (int kw = 0; kw < gm.KW; ++kw) {
        const int fi = fo * gm.SF - gm.PL + kw;
        const bool inb = ok && fi >= 0 && fi < gm.Fi;
        const long src = inb ? bt * gm.Fi + fi : 0;
        f32x4 bfr[1][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bfr[0][kb] = ldg4(in + src * 64 + 16 * kb + 4 * g);
            if (!inb) bfr[0][kb] = splat4(0.f);
        }
        const float* wp = wimg + ((long)kw * 4 * NG * 4) * 256 + lane * 4;
#pragma unroll
        for (int ob = 0; ob < 4 * NG; ++ob) {
            f32x4 a1[1] = {acc[ob]};
            lin_acc<4, 1>(wp + (long)ob * 4 * 256, bfr, a1);
            acc[ob] = a1[0];
        }
    }
    if (ok) {
#pragma unroll
        for (int ob = 0; ob < 4 * NG; ++ob) {
            const int r = ob >> 2;                                        // pixel-shuffle phase (0 when NG = 1)
            const long orow = NG == 1 ? mm : (bt * gm.Fo + fo) * 2 + r;   // (b, t, 2 fo + r) of the 2 Fo wide plane
            stg4(z + orow * 64 + 16 * (ob & 3) + 4 * g, acc[ob]);
        }
    }
}

// din[(b,t,fi)][ci] = sum_kw W_kw^T dz[(b,t,fo)] with fo SF - PL + kw = fi          images: [kw][4 ob(ci)][4 NG kb(co)]
template <int NG>
__global__ __launch_bounds__(256) void rc_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wimgT,
                                                       RcGeom gm, float* __restrict__ din) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long Mi = (long)gm.B * gm.T * gm.Fi;
    const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (m0 >= Mi) return;
    const long m = m0 + c;
    const bool ok = m < Mi;
    const long mm = ok ? m : Mi - 1;
    const long bt = mm / gm.Fi;
    const int fi = (int)(mm - bt * gm.Fi);
    f32x4 acc[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc[ob] = splat4(0.f);
    for (int kw = 0; kw < gm.KW; ++kw) {
        const int num = fi + gm.PL - kw;
        const int fo = num / gm.SF;
        const bool inb = ok && num >= 0 && fo * gm.SF == num && fo < gm.Fo;
        f32x4 bfr[1][4 * NG];
#pragma unroll
        for (int kb = 0; kb < 4 * NG; ++kb) {
            const long srow = NG == 1 ? bt * gm.Fo + fo : (bt * gm.Fo + fo) * 2 + (kb >> 2);
            bfr[0][kb] = ldg4(dz + (inb ? srow : 0) * 64 + 16 * (kb & 3) + 4 * g);
            if (!inb) bfr[0][kb] = splat4(0.f);
        }
        const float* wp = wimgT + ((long)kw * 4 * 4 * NG) * 256 + lane * 4;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x4 a1[1] = {acc[ob]};
            lin_acc<4 * NG, 1>(wp + (long)ob * 4 * NG * 256, bfr, a1);
            acc[ob] = a1[0];
        }
    }
    if (ok) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) stg4(din + mm * 64 + 16 * ob + 4 * g, acc[ob]);
    }
}

// partial[kw][s][co][ci] = sum over the s-th range of output tokens of dz[.][co] * in[(b, t, fo SF - PL + kw)][ci]
template <int NG>
__global__ __launch_bounds__(256) void rc_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ in, RcGeom gm,
                                                       int nsplit, float* __restrict__ partial) {
    __shared__ float red[4][16 * 64];
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    const int ib = blockIdx.x, kw = blockIdx.y, s = blockIdx.z;            // ib: output-channel block of 16
    const long Mo = (long)gm.B * gm.T * gm.Fo;
    const long steps = (Mo + 15) / 16, per = (steps + nsplit - 1) / nsplit;
    const long st0 = (long)s * per, st1 = st0 + per < steps ? st0 + per : steps;
    f32x4 acc[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) acc[jb] = splat4(0.f);
    for (long st = st0 + wv; st < st1; st += 4) {
        float av[4];
        f32x4 bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = st * 16 + 4 * g + r;
            const bool ok = m < Mo;
            const long mm = ok ? m : Mo - 1;
            const long bt = mm / gm.Fo;
            const int fo = (int)(mm - bt * gm.Fo), fi = fo * gm.SF - gm.PL + kw;
            const bool inb = ok && fi >= 0 && fi < gm.Fi;
            const long src = inb ? bt * gm.Fi + fi : 0;
            const long zrow = NG == 1 ? mm : (bt * gm.Fo + fo) * 2 + (ib >> 2);
            av[r] = ok ? dz[zrow * 64 + 16 * (ib & 3) + c] : 0.f;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const float v = in[src * 64 + 16 * jb + c];
                bv[jb][r] = inb ? v : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[jb] = mfma16(av[r], bv[jb][r], acc[jb]);
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][(4 * g + r) * 64 + 16 * jb + c] = acc[jb][r];
    __syncthreads();
    const int Co = 64 * NG;
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        const int i = e >> 6, j = e & 63;
        partial[(((long)kw * nsplit + s) * Co + 16 * ib + i) * 64 + j] = v;
    }
}

// dW[co][ci][0][kw] = sum_s partial[kw][s][co][ci]                     (conv weight [Co, 64, 1, KW])
// a block is 64 elements x 4 slab groups (group g adds slabs g, g + 4, ... two at a time; the groups are combined in group order:
// a fixed order for a given launch shape) - one thread walking all the slabs of its element was a dependent fetch per slab
__global__ __launch_bounds__(256) void rc_wgrad_scatter_kernel(const float* __restrict__ partial, int nsplit, int Co, int KW,
                                                               float* __restrict__ dW) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col, n = KW * Co * 64;
    float s0 = 0.f, s1 = 0.f;
    int kw = 0, e = 0;
    if (idx < n) {
        kw = idx / (Co * 64);
        e = idx - kw * Co * 64;                                            // e = co * 64 + ci
        int k = grp;
        for (; k + 4 < nsplit; k += 8) {
            s0 += partial[((long)kw * nsplit + k) * Co * 64 + e];
            s1 += partial[((long)kw * nsplit + k + 4) * Co * 64 + e];
        }
        for (; k < nsplit; k += 4) s0 += partial[((long)kw * nsplit + k) * Co * 64 + e];
    }
    red[grp][col] = s0 + s1;
    __syncthreads();
    if (grp == 0 && idx < n) dW[(long)e * KW + kw] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

// tap images of a row conv: plain [kw][Co/16][4] and transposed [kw][4][Co/16]
__global__ void rc_pack_kernel(const float* __restrict__ w, int Co, int KW, float* __restrict__ img, float* __restrict__ imgT) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = Co * 64;
    if (e >= n) return;
    const int kw = blockIdx.y;
    const int r = e & 3, lane = (e >> 2) & 63, blk = e >> 8;
    {   // plain: rows = co (Co), cols = ci (64): KB = 4
        const int rb = blk >> 2, kb = blk & 3;
        const int co = 16 * rb + (lane & 15), ci = 16 * kb + 4 * (lane >> 4) + r;
        img[(long)kw * n + e] = w[((long)co * 64 + ci) * KW + kw];
    }
    {   // transposed: rows = ci (64), cols = co (Co): KB = Co / 16
        const int KB = Co / 16, rb = blk / KB, kb = blk - rb * KB;
        const int ci = 16 * rb + (lane & 15), co = 16 * kb + 4 * (lane >> 4) + r;
        imgT[(long)kw * n + e] = w[((long)co * 64 + ci) * KW + kw];
    }
}

// ---- conv_1 of the encoder: Conv2d(3 -> 64, 1x1) on the [mag, re, im] planes (generator.py:54) --------------------
__global__ __launch_bounds__(256) void c1_fwd_kernel(const float* __restrict__ xin, long M, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ z) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < M * 64; i += (long)gridDim.x * 256) {
        const long m = i >> 6;
        const int co = (int)(i & 63);
        z[i] = b[co] + w[co * 3] * xin[m * 3] + w[co * 3 + 1] * xin[m * 3 + 1] + w[co * 3 + 2] * xin[m * 3 + 2];
    }
}
// partial[blk][co][ci] = sum over the block's positions of dz[m][co] * xin[m][ci]
// bpart: [blk][64] partial sums of dz itself (the conv_1 bias gradient) from the same pass
__global__ __launch_bounds__(256) void c1_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ xin, long M,
                                                       float* __restrict__ partial, float* __restrict__ bpart) {
    __shared__ float red[4][64][4];
    const int co = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const long per = (M + gridDim.x - 1) / gridDim.x, m0 = (long)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, sb = 0.f;
    long m = m0 + sub;
    for (; m + 12 < m1; m += 16) {                                // four positions' operands in flight (same sum order)
        float d[4], x[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            d[u] = dz[(m + 4 * u) * 64 + co];
#pragma unroll
            for (int k = 0; k < 3; ++k) x[u][k] = xin[(m + 4 * u) * 3 + k];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 = fmaf(d[u], x[u][0], s0); s1 = fmaf(d[u], x[u][1], s1); s2 = fmaf(d[u], x[u][2], s2); sb += d[u]; }
    }
    for (; m < m1; m += 4) {
        const float d = dz[m * 64 + co];
        s0 = fmaf(d, xin[m * 3], s0); s1 = fmaf(d, xin[m * 3 + 1], s1); s2 = fmaf(d, xin[m * 3 + 2], s2);
        sb += d;
    }
    red[sub][co][0] = s0; red[sub][co][1] = s1; red[sub][co][2] = s2; red[sub][co][3] = sb;
    __syncthreads();
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            partial[((long)blockIdx.x * 64 + co) * 3 + k] = (red[0][co][k] + red[1][co][k]) + (red[2][co][k] + red[3][co][k]);
        bpart[(long)blockIdx.x * 64 + co] = (red[0][co][3] + red[1][co][3]) + (red[2][co][3] + red[3][co][3]);
    }
}

// InstanceNorm2d(64, affine) + PReLU(64) on a channels-last plane: forward (statistics -> mean, rstd) ...
static void in_prelu_forward(LaunchCtx ctx, const float* z, int B, int P, const float* gamma, const float* beta,
                             const float* alpha, float* mean, float* rstd, float* part, float* a) {
    hipStream_t st = ctx.stream;
    LAUNCH(ctx, "in_prelu_train", (db_sums_kernel<0><<<dim3(B, DB_NCH), 256, 0, st>>>(z, nullptr, P, nullptr, nullptr, nullptr,
                                                                                    nullptr, nullptr, part)));
    LAUNCH(ctx, "in_prelu_train", (db_stats_finalize_kernel<<<B * 64, 256, 0, st>>>(part, B, (double)P, mean, rstd)));
    LAUNCH(ctx, "in_prelu_train", (db_norm_prelu_kernel<<<2048, 256, 0, st>>>(z, (long)B * P * 64, P, mean, rstd, gamma, beta,
                                                                              alpha, a)));
}
// ... and backward: g holds dL/da on entry (or gin does: then g is output only) and dL/dz on exit; the three parameter
// gradients are written
static void in_prelu_backward(LaunchCtx ctx, const float* z, float* g, int B, int P, const float* gamma, const float* beta,
                              const float* alpha, const float* mean, const float* rstd, float* part, float* m1, float* m2,
                              float* dgamma, float* dbeta, float* dalpha, const float* gin = nullptr) {
    hipStream_t st = ctx.stream;
    LAUNCH(ctx, "in_prelu_train", (db_sums_kernel<1><<<dim3(B, DB_NCH), 256, 0, st>>>(z, g, P, mean, rstd, gamma, beta, alpha,
                                                                                    part, gin)));
    LAUNCH(ctx, "in_prelu_train", (db_bwd_finalize_kernel<<<64, 1024, 0, st>>>(part, B, (double)P, m1, m2, dgamma, dbeta, dalpha)));
    LAUNCH(ctx, "in_prelu_train", (db_in_bwd_kernel<<<2048, 256, 0, st>>>(g, z, (long)B * P * 64, P, mean, rstd, gamma, m1, m2)));
}

#if TRAIN_X3
// The forward 1 x 3 convolutions of the training step (the encoder's stride-2 conv, the decoders' sub-pixel conv) ARE the
// inference kernels (conv3x_kernel<1, 64> mode 1 / <1, 128> mode 2, conv_x3.hip), as the dense blocks' convs are: same
// planes, same zero padding, raw output; only the operand image is rebuilt from the raw weight [Co, 64, 1, 3] every step.
// Image: [2 chunks of 32 input channels][tap 3][Co / 16 cb][hi | lo][64][8 halfs], lane (c, g) slot e <->
// w[co = 16 cb + c][ci = 32 chunk + 16 (e >> 2) + 4 g + (e & 3)][tap]   (api.hip: x3_conv_image(src, 4, 3, Co / 16))
__global__ void rc_pack_x3_kernel(const float* __restrict__ w, int Co, _Float16* __restrict__ img) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, CB = Co / 16;
    if (t >= 2 * 3 * CB * 64) return;
    const int lane = t & 63, cb = (t >> 6) % CB, rest = (t >> 6) / CB, tap = rest % 3, chunk = rest / 3;
    const int co = 16 * cb + (lane & 15);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = 32 * chunk + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        const float v = w[((long)co * 64 + ci) * 3 + tap];
        const _Float16 h = (_Float16)v;
        hi[e] = h;
        lo[e] = (_Float16)(v - (float)h);
    }
    _Float16* o = img + ((long)(chunk * 3 + tap) * CB + cb) * 1024 + lane * 8;
    *reinterpret_cast<f16x8*>(o) = hi;
    *reinterpret_cast<f16x8*>(o + 512) = lo;
}
// the plane must be long enough for the kernel's tiles (db_x3_forward) and the geometry one of the two the modes express
static bool rc_x3_forward(const RcGeom& gm, int NG) {
    static const bool k_on = env_knob("CMGAN_RC_FWD_X3", 1, 0, 1) != 0;
    return k_on && gm.KW == 3 && gm.PL == 1 && gm.Fi + 1 >= 64 &&
           ((NG == 2 && gm.SF == 1 && gm.Fo == gm.Fi) || (NG == 1 && gm.SF == 2 && gm.Fo == (gm.Fi - 1) / 2 + 1));
}
#endif
// img: the fp32 fragment image of rc_pack_kernel; the split-f16 path rebuilds it in place (same size) from the raw weight
template <int NG>
static void rc_forward(LaunchCtx ctx, const float* in, float* img, const float* wraw, const float* bias, const RcGeom& gm,
                       float* z) {
#if TRAIN_X3
    if (rc_x3_forward(gm, NG)) {
        const int nthr = 2 * 3 * 4 * NG * 64;
        LAUNCH(ctx, "rowconv_train", (rc_pack_x3_kernel<<<(nthr + 255) / 256, 256, 0, ctx.stream>>>(
                                         wraw, 64 * NG, reinterpret_cast<_Float16*>(img))));
        ConvArgs ca{};
        ca.in[0] = in;
        ca.nslots = 1;
        ca.bias = bias;
        ca.out = z;
        ca.T = gm.T; ca.F = gm.Fi; ca.dil = 1; ca.mode = NG == 2 ? 2 : 1; ca.ntiles = conv3x_ntiles(gm.T, gm.Fi, 64 * NG);
        launch_conv3_x3(ctx, ca, img, gm.B, 1, 64 * NG);
        return;
    }
#endif
    (void)wraw;
    const long Mo = (long)gm.B * gm.T * gm.Fo;
    LAUNCH(ctx, "rowconv_train", (rc_fwd_kernel<NG><<<(unsigned)((Mo + 63) / 64), 256, 0, ctx.stream>>>(in, img, bias, gm, z)));
}
// db / cpart (optional): the bias gradient from the weight-gradient kernel ([RC_WG_SPLIT][64 NG] partials at cpart); returns
// false if the caller still has to compute it (fp32 build / planes of 4 GB)
template <int NG>
static bool rc_backward(LaunchCtx ctx, const float* dz, const float* in, const float* imgT, const RcGeom& gm, float* din,
                        float* dW, float* wpart, float* db = nullptr, float* cpart = nullptr) {
    bool db_done = false;
    hipStream_t st = ctx.stream;
    const long Mi = (long)gm.B * gm.T * gm.Fi;
    const int nw = gm.KW * 64 * NG * 64;
    const long rows = (long)gm.B * gm.T * (gm.Fi > NG * gm.Fo ? gm.Fi : NG * gm.Fo);
#if TRAIN_X3
    static const bool k_wx3 = env_knob("CMGAN_RC_WGRAD_X3", 1, 0, 1) != 0;
    if (k_wx3 && rows * 256 < (1l << 32)) {           // split products (train_x3.hip), RC_WG_SPLIT position ranges
        const int gm7[7] = {gm.B, gm.T, gm.Fi, gm.Fo, gm.KW, gm.SF, gm.PL};
        launch_rc_wgrad_x3(ctx, NG, dz, in, gm7, RC_WG_SPLIT, wpart, db ? cpart : nullptr);
        LAUNCH(ctx, "rowconv_train", (rc_wgrad_scatter_kernel<<<(nw + 63) / 64, 256, 0, st>>>(wpart, RC_WG_SPLIT, 64 * NG,
                                                                                                gm.KW, dW)));
        if (db) {
            LAUNCH(ctx, "rowconv_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(cpart, RC_WG_SPLIT, 64 * NG, db)));
            db_done = true;
        }
    } else
#endif
    {
        (void)rows;
        LAUNCH(ctx, "rowconv_train", (rc_wgrad_kernel<NG><<<dim3(4 * NG, gm.KW, FFN_WGRAD_SPLIT), 256, 0, st>>>(
                                         dz, in, gm, FFN_WGRAD_SPLIT, wpart)));
        LAUNCH(ctx, "rowconv_train", (rc_wgrad_scatter_kernel<<<(nw + 63) / 64, 256, 0, st>>>(wpart, FFN_WGRAD_SPLIT, 64 * NG,
                                                                                                gm.KW, dW)));
    }
    if (din)
        LAUNCH(ctx, "rowconv_train", (rc_dgrad_kernel<NG><<<(unsigned)((Mi + 63) / 64), 256, 0, st>>>(dz, imgT, gm, din)));
    return db_done;
}

// ---- DenseEncoder (generator.py:50-69) --------------------------------------------------------------------------
struct EncPlan { size_t img2, img2T, z1, a1, d, z2, g, st, part, m, wpart, cpart, dense, total; };
static EncPlan enc_plan(int B, int T, int F) {
    EncPlan p;
    const size_t M = (size_t)B * T * F, F2 = (F - 1) / 2 + 1, M2 = (size_t)B * T * F2;
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    p.img2 = take(3 * 4096); p.img2T = take(3 * 4096);
    p.z1 = take(M * 64); p.a1 = take(M * 64); p.d = take(M * 64); p.z2 = take(M2 * 64);
    p.g = take(M * 64);                         // gradient plane (dd, then da1 / dz1)
    p.st = take((size_t)4 * B * 64);            // mean1, rstd1, mean2, rstd2
    p.part = take((size_t)B * DB_NCH * 64 * 3);
    p.m = take((size_t)2 * B * 64);
    p.wpart = take((size_t)3 * RC_WG_SPLIT * 4096);
    p.cpart = take((size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256);
    p.dense = take(dense_train_ws_floats(B, T, F));
    p.total = cur;
    return p;
}
size_t encoder_train_ws_floats(int B, int T, int F) { return enc_plan(B, T, F).total; }

void launch_encoder_train_forward(LaunchCtx ctx, const float* xin, int B, int T, int F, const EncoderTrainParams& p,
                                  float* y, float* ws) {
    hipStream_t st = ctx.stream;
    const EncPlan pl = enc_plan(B, T, F);
    const long M = (long)B * T * F;
    const int F2 = (F - 1) / 2 + 1;
    const RcGeom g2{B, T, F, F2, 3, 2, 1};
    LAUNCH(ctx, "encoder_train", (rc_pack_kernel<<<dim3(16, 3), 256, 0, st>>>(p.c2_w, 64, 3, ws + pl.img2, ws + pl.img2T)));
    LAUNCH(ctx, "encoder_train", (c1_fwd_kernel<<<2048, 256, 0, st>>>(xin, M, p.c1_w, p.c1_b, ws + pl.z1)));
    float* stt = ws + pl.st;
    in_prelu_forward(ctx, ws + pl.z1, B, T * F, p.n1_w, p.n1_b, p.p1_w, stt, stt + B * 64, ws + pl.part, ws + pl.a1);
    launch_dense_train_forward(ctx, ws + pl.a1, B, T, F, p.dense, ws + pl.d, ws + pl.dense);
    rc_forward<1>(ctx, ws + pl.d, ws + pl.img2, p.c2_w, p.c2_b, g2, ws + pl.z2);
    in_prelu_forward(ctx, ws + pl.z2, B, T * F2, p.n2_w, p.n2_b, p.p2_w, stt + 2 * B * 64, stt + 3 * B * 64, ws + pl.part, y);
}

void launch_encoder_train_backward(LaunchCtx ctx, const float* xin, const float* dy, int B, int T, int F,
                                   const EncoderTrainParams& p, const EncoderTrainParams& grad, float* ws) {
    hipStream_t st = ctx.stream;
    const EncPlan pl = enc_plan(B, T, F);
    const long M = (long)B * T * F;
    const int F2 = (F - 1) / 2 + 1;
    const long M2 = (long)B * T * F2;
    const RcGeom g2{B, T, F, F2, 3, 2, 1};
    float* stt = ws + pl.st;
    float* cpart = ws + pl.cpart;
    // conv_2 + IN + PReLU: dy -> dz2 (kept in the front of the [M,64] gradient plane)
    float* dz2 = ws + pl.g;
    in_prelu_backward(ctx, ws + pl.z2, dz2, B, T * F2, p.n2_w, p.n2_b, p.p2_w, stt + 2 * B * 64, stt + 3 * B * 64, ws + pl.part,
                      ws + pl.m, ws + pl.m + B * 64, grad.n2_w, grad.n2_b, grad.p2_w, dy);      // (reads dy: no copy first)
    // the weight gradient reads conv_2's input d; only then is d overwritten by its own gradient dd.  (db_c2 = colsum dz2
    // comes out of the split-f16 weight-gradient kernel; otherwise the column-sum pass)
    if (!rc_backward<1>(ctx, dz2, ws + pl.d, ws + pl.img2T, g2, nullptr, grad.c2_w, ws + pl.wpart, grad.c2_b, cpart)) {
        LAUNCH(ctx, "encoder_train", (colsum_partial_kernel<<<FFN_COLSUM_BLOCKS, 256, 0, st>>>(dz2, M2, 64, cpart)));
        LAUNCH(ctx, "encoder_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(cpart, FFN_COLSUM_BLOCKS, 64, grad.c2_b)));
    }
    if (!rc_dgrad_x3(ctx, 1, dz2, p.c2_w, g2, ws + pl.d))
        LAUNCH(ctx, "rowconv_train", (rc_dgrad_kernel<1><<<(unsigned)((M + 63) / 64), 256, 0, st>>>(dz2, ws + pl.img2T, g2,
                                                                                                   ws + pl.d)));
    // dilated dense block: x = a1, dy = dd (in pl.d) -> da1 (into pl.g; dz2 is dead)
    launch_dense_train_backward(ctx, ws + pl.a1, ws + pl.d, B, T, F, p.dense, ws + pl.g, grad.dense, ws + pl.dense);
    // conv_1 + IN + PReLU
    float* dz1 = ws + pl.g;
    in_prelu_backward(ctx, ws + pl.z1, dz1, B, T * F, p.n1_w, p.n1_b, p.p1_w, stt, stt + B * 64, ws + pl.part, ws + pl.m,
                      ws + pl.m + B * 64, grad.n1_w, grad.n1_b, grad.p1_w);
    LAUNCH(ctx, "encoder_train", (c1_wgrad_kernel<<<FFN_COLSUM_BLOCKS, 256, 0, st>>>(dz1, xin, M, ws + pl.wpart, cpart)));
    LAUNCH(ctx, "encoder_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(cpart, FFN_COLSUM_BLOCKS, 64, grad.c1_b)));
    LAUNCH(ctx, "encoder_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(ws + pl.wpart, FFN_COLSUM_BLOCKS, 192, grad.c1_w)));
}

// row of position m in rows of `w` positions.  The [R, W, 64] fp32 plane of these kernels fits the GPU's memory, so positions are
// far below 2^31 and a 32-bit division does (a 64-bit one is ~100 VALU instructions per lane: tail_wgrad / tail_dgrad spent
// most of their time in it - 0.55 / 0.40 ms per launch against 0.11 ms of HBM time)
__device__ __forceinline__ long tail_row(long m, int w) { return (long)((unsigned)m / (unsigned)w); }
// ---- decoder tails: Conv2d(64 -> NO, (1,2)) with NO = 1 (mask head) or 2 (complex head) on a [R, W, 64] plane -------
// out[(r, f)][o] = b[o] + sum_{kw, ci} w[o][ci][0][kw] in[(r, f + kw)][ci],  f < W - 1.       (generator.py:126,148)
// 16 lanes share one output position: the 128-float window (two adjacent rows) is read once, coalesced.
template <int NO>
__global__ __launch_bounds__(256) void tail_fwd_kernel(const float* __restrict__ in, long R, int W, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out) {
    const int l = threadIdx.x & 15;
    const long Mo = R * (W - 1);
    float wv[NO][8];
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * l + e, kw = j >> 6, ci = j & 63;
            wv[o][e] = w[(o * 64 + ci) * 2 + kw];
        }
    for (long m = (long)blockIdx.x * 16 + (threadIdx.x >> 4); m < Mo; m += (long)gridDim.x * 16) {
        const long r = tail_row(m, W - 1);
        const long base = (m + r) * 64;                         // position (r, f) of the W wide plane = m + r
        const f32x4 v0 = ldg4(in + base + 8 * l), v1 = ldg4(in + base + 8 * l + 4);
        float s[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a = fmaf(wv[o][e], v0[e], a); a = fmaf(wv[o][4 + e], v1[e], a); }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) a += __shfl_xor(a, d, 16);
            s[o] = a;
        }
        if (l == 0) {
#pragma unroll
            for (int o = 0; o < NO; ++o) out[m * NO + o] = s[o] + b[o];
        }
    }
}
// din[(r, fi)][ci] = sum_{kw, o} w[o][ci][0][kw] dz[(r, fi - kw)][o]
template <int NO>
__global__ __launch_bounds__(256) void tail_dgrad_kernel(const float* __restrict__ dz, long R, int W, const float* __restrict__ w,
                                                         float* __restrict__ din) {
    const long total = R * W * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ci = (int)(i & 63);
        const long pos = i >> 6, r = tail_row(pos, W);
        const int fi = (int)(pos - r * W);
        float s = 0.f;
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
            const int f = fi - kw;
            if (f < 0 || f >= W - 1) continue;
#pragma unroll
            for (int o = 0; o < NO; ++o) s = fmaf(w[(o * 64 + ci) * 2 + kw], dz[(r * (W - 1) + f) * NO + o], s);
        }
        din[i] = s;
    }
}
// partial[blk][o][ci][kw] = sum over the block's output positions of dz[m][o] in[(r, f + kw)][ci]
template <int NO>
__global__ __launch_bounds__(256) void tail_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ in, long R, int W,
                                                         float* __restrict__ partial) {
    __shared__ float red[4][NO * 128];
    const int ci = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const long Mo = R * (W - 1);
    const long per = (Mo + gridDim.x - 1) / gridDim.x, m0 = (long)blockIdx.x * per, m1 = m0 + per < Mo ? m0 + per : Mo;
    float s[NO][2];
#pragma unroll
    for (int o = 0; o < NO; ++o) s[o][0] = s[o][1] = 0.f;
    // four positions per trip, their operands fetched before the first FMA (the loop was one dependent fetch per position:
    // 1 000 round trips per thread at eight waves per CU - 0.55 ms per launch); the sums keep their order
    long m = m0 + sub;
    for (; m + 12 < m1; m += 16) {
        float x0[4], x1[4], d[4][NO];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long mu = m + 4 * u, r = tail_row(mu, W - 1);
            x0[u] = in[(mu + r) * 64 + ci];
            x1[u] = in[(mu + r + 1) * 64 + ci];
#pragma unroll
            for (int o = 0; o < NO; ++o) d[u][o] = dz[mu * NO + o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                s[o][0] = fmaf(d[u][o], x0[u], s[o][0]);
                s[o][1] = fmaf(d[u][o], x1[u], s[o][1]);
            }
    }
    for (; m < m1; m += 4) {
        const long r = tail_row(m, W - 1);
        const float x0 = in[(m + r) * 64 + ci], x1 = in[(m + r + 1) * 64 + ci];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const float d = dz[m * NO + o];
            s[o][0] = fmaf(d, x0, s[o][0]);
            s[o][1] = fmaf(d, x1, s[o][1]);
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) { red[sub][(o * 64 + ci) * 2] = s[o][0]; red[sub][(o * 64 + ci) * 2 + 1] = s[o][1]; }
    __syncthreads();
    for (int e = threadIdx.x; e < NO * 128; e += 256)
        partial[(long)blockIdx.x * NO * 128 + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

template <int NO>
static void tail_forward(LaunchCtx ctx, const float* in, long R, int W, const float* w, const float* b, float* out) {
    LAUNCH(ctx, "tail_train", (tail_fwd_kernel<NO><<<2048, 256, 0, ctx.stream>>>(in, R, W, w, b, out)));
}
// dW, db and the data gradient of a tail conv; wpart needs FFN_COLSUM_BLOCKS * NO * 128 floats, cpart FFN_COLSUM_BLOCKS * 256
template <int NO>
static void tail_backward(LaunchCtx ctx, const float* dz, const float* in, long R, int W, const float* w, float* din, float* dW,
                          float* db, float* wpart, float* cpart) {
    hipStream_t st = ctx.stream;
    LAUNCH(ctx, "tail_train", (tail_wgrad_kernel<NO><<<FFN_COLSUM_BLOCKS, 256, 0, st>>>(dz, in, R, W, wpart)));
    LAUNCH(ctx, "tail_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(wpart, FFN_COLSUM_BLOCKS, NO * 128, dW)));
    LAUNCH(ctx, "tail_train", (colsum_partial_kernel<<<FFN_COLSUM_BLOCKS, 256, 0, st>>>(dz, R * (W - 1), NO, cpart)));
    LAUNCH(ctx, "tail_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(cpart, FFN_COLSUM_BLOCKS, NO, db)));
    LAUNCH(ctx, "tail_train", (tail_dgrad_kernel<NO><<<2048, 256, 0, st>>>(dz, R, W, w, din)));
}

// ---- mask head after conv_1: InstanceNorm2d(1, affine) + PReLU(1) + Conv2d(1,1,1x1) + PReLU(num_features) -----------
//   n = gamma zhat + beta,  a = PReLU_alpha(n),  u = wf a + bf,  mask = PReLU_{alphaf[f]}(u)        (generator.py:127-138)
struct MaskTailP { const float *gamma, *beta, *alpha, *wf, *bf, *alphaf; };
struct MaskTailG { float *gamma, *beta, *alpha, *wf, *bf, *alphaf; };

__global__ __launch_bounds__(256) void mt_stats_kernel(const float* __restrict__ t1, int P, float* __restrict__ partial) {
    __shared__ float red[256][2];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int per = (P + MT_NCH - 1) / MT_NCH, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    float s0 = 0.f, s1 = 0.f;
    for (int p = p0 + threadIdx.x; p < p1; p += 256) { const float v = t1[(long)b * P + p]; s0 += v; s1 = fmaf(v, v, s1); }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { red[threadIdx.x][0] += red[threadIdx.x + d][0]; red[threadIdx.x][1] += red[threadIdx.x + d][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[((long)b * MT_NCH + chunk) * 2] = red[0][0]; partial[((long)b * MT_NCH + chunk) * 2 + 1] = red[0][1]; }
}
__global__ void mt_stats_finalize_kernel(const float* __restrict__ partial, int B, double count, float* __restrict__ mean,
                                         float* __restrict__ rstd) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < MT_NCH; ++k) { s1 += (double)partial[((long)b * MT_NCH + k) * 2]; s2 += (double)partial[((long)b * MT_NCH + k) * 2 + 1]; }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[b] = (float)mu;
    rstd[b] = (float)(1.0 / sqrt(var + 1e-5));
}
__global__ __launch_bounds__(256) void mt_apply_kernel(const float* __restrict__ t1, long total, int P, int F,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       MaskTailP p, float* __restrict__ mask) {
    const float gm = p.gamma[0], bt = p.beta[0], al = p.alpha[0], wf = p.wf[0], bf = p.bf[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P;
        const int f = (int)(i % F);
        const float n = (t1[i] - mean[b]) * rstd[b] * gm + bt;
        const float a = n >= 0.f ? n : al * n;
        const float u = fmaf(wf, a, bf);
        mask[i] = u >= 0.f ? u : p.alphaf[f] * u;
    }
}
// backward pass 1: g = dL/dmask on entry, dn on exit; partial[b][chunk][5] = (sum dn, sum dn zhat, sum da n [n<0],
// sum du a, sum du)
__global__ __launch_bounds__(256) void mt_bwd_sums_kernel(const float* __restrict__ t1, float* __restrict__ g, int P, int F,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          MaskTailP p, float* __restrict__ partial) {
    __shared__ float red[256][5];
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int per = (P + MT_NCH - 1) / MT_NCH, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    const float gm = p.gamma[0], bt = p.beta[0], al = p.alpha[0], wf = p.wf[0], bf = p.bf[0], mu = mean[b], rs = rstd[b];
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = p0 + threadIdx.x; q < p1; q += 256) {
        const long i = (long)b * P + q;
        const int f = q % F;
        const float zh = (t1[i] - mu) * rs, n = zh * gm + bt, a = n >= 0.f ? n : al * n, u = fmaf(wf, a, bf);
        const float du = u >= 0.f ? g[i] : g[i] * p.alphaf[f];
        const float da = du * wf, dn = n >= 0.f ? da : da * al;
        g[i] = dn;
        s[0] += dn; s[1] = fmaf(dn, zh, s[1]); s[2] += n < 0.f ? da * n : 0.f; s[3] = fmaf(du, a, s[3]); s[4] += du;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[threadIdx.x][k] = s[k];
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d)
#pragma unroll
            for (int k = 0; k < 5; ++k) red[threadIdx.x][k] += red[threadIdx.x + d][k];
        __syncthreads();
    }
    if (threadIdx.x < 5) partial[((long)b * MT_NCH + chunk) * 5 + threadIdx.x] = red[0][threadIdx.x];
}
__global__ void mt_bwd_finalize_kernel(const float* __restrict__ partial, int B, double count, float* __restrict__ m1,
                                       float* __restrict__ m2, MaskTailG gr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double t[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
        double s[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < MT_NCH; ++k)
            for (int j = 0; j < 5; ++j) s[j] += (double)partial[((long)b * MT_NCH + k) * 5 + j];
        m1[b] = (float)(s[0] / count);
        m2[b] = (float)(s[1] / count);
        for (int j = 0; j < 5; ++j) t[j] += s[j];
    }
    gr.beta[0] = (float)t[0]; gr.gamma[0] = (float)t[1]; gr.alpha[0] = (float)t[2]; gr.wf[0] = (float)t[3]; gr.bf[0] = (float)t[4];
}
// backward pass 2: dt1 = gamma rstd (dn - mean dn - zhat mean(dn zhat)), in place
__global__ __launch_bounds__(256) void mt_in_bwd_kernel(float* __restrict__ g, const float* __restrict__ t1, long total, int P,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ m1,
                                                        const float* __restrict__ m2) {
    const float gm = gamma[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P;
        const float zh = (t1[i] - mean[b]) * rstd[b];
        g[i] = gm * rstd[b] * (g[i] - m1[b] - zh * m2[b]);
    }
}
// dalphaf[f] = sum_{b,t} dL/dmask u [u < 0]: one block per frequency, rows in (b, t) order
__global__ __launch_bounds__(256) void mt_dalphaf_kernel(const float* __restrict__ t1, const float* __restrict__ dmask, int B,
                                                         int T, int F, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, MaskTailP p, float* __restrict__ dalphaf) {
    __shared__ float red[256];
    const int f = blockIdx.x;
    const float gm = p.gamma[0], bt = p.beta[0], al = p.alpha[0], wf = p.wf[0], bf = p.bf[0];
    float s = 0.f;
    for (long r = threadIdx.x; r < (long)B * T; r += 256) {
        const long b = r / T, i = r * F + f;
        const float n = (t1[i] - mean[b]) * rstd[b] * gm + bt, a = n >= 0.f ? n : al * n, u = fmaf(wf, a, bf);
        s += u < 0.f ? dmask[i] * u : 0.f;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) dalphaf[f] = red[0];
}

// ---- MaskDecoder / ComplexDecoder (generator.py:121-156) ----------------------------------------------------------
struct DecPlan { size_t img, imgT, d, s, a, t1, g, g2, st, part, m, wpart, cpart, dense, total; };
static DecPlan dec_plan(int B, int T, int Fe) {
    DecPlan p;
    const size_t Me = (size_t)B * T * Fe, Ms = 2 * Me;
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    p.img = take(3 * 8192); p.imgT = take(3 * 8192);
    p.d = take(Me * 64);                        // dense block output (sub-pixel input); later its gradient
    p.s = take(Ms * 64);                        // sub-pixel output [B,T,2 Fe,64] (pre-norm for the complex head)
    p.a = take(Ms * 64);                        // complex head: PReLU(IN(s))
    p.t1 = take(Ms);                            // mask head: conv_1 output [B,T,2 Fe - 1]
    p.g = take(Ms * 64);                        // gradient plane on the 2 Fe wide grid
    p.g2 = take(Ms * 2);                        // gradient of the head output (dn / dt1, or a copy of dc)
    p.st = take((size_t)2 * B * 64);
    p.part = take((size_t)B * DB_NCH * 64 * 3);
    p.m = take((size_t)2 * B * 64);
    p.wpart = take((size_t)3 * RC_WG_SPLIT * 8192);
    p.cpart = take((size_t)COLSUM_MAX_JOBS * FFN_COLSUM_BLOCKS * 256);
    p.dense = take(dense_train_ws_floats(B, T, Fe));
    p.total = cur;
    return p;
}
size_t decoder_train_ws_floats(int B, int T, int Fe) { return dec_plan(B, T, Fe).total; }

static MaskTailP mask_tail_params(const DecoderTrainParams& p) { return {p.n_w, p.n_b, p.p_w, p.f_w, p.f_b, p.po_w}; }

// kind 0: mask head -> out [B,T,2 Fe - 1];  kind 1: complex head -> out [B,T,2 Fe - 1,2]
void launch_decoder_train_forward(LaunchCtx ctx, int kind, const float* x, int B, int T, int Fe, const DecoderTrainParams& p,
                                  float* out, float* ws) {
    hipStream_t st = ctx.stream;
    const DecPlan pl = dec_plan(B, T, Fe);
    const int W = 2 * Fe, F = W - 1;
    const long R = (long)B * T;
    const RcGeom gs{B, T, Fe, Fe, 3, 1, 1};
    LAUNCH(ctx, "decoder_train", (rc_pack_kernel<<<dim3(32, 3), 256, 0, st>>>(p.sp_w, 128, 3, ws + pl.img, ws + pl.imgT)));
    launch_dense_train_forward(ctx, x, B, T, Fe, p.dense, ws + pl.d, ws + pl.dense);
    rc_forward<2>(ctx, ws + pl.d, ws + pl.img, p.sp_w, p.sp_b, gs, ws + pl.s);
    float* stt = ws + pl.st;
    if (kind == 0) {
        tail_forward<1>(ctx, ws + pl.s, R, W, p.c_w, p.c_b, ws + pl.t1);
        LAUNCH(ctx, "decoder_train", (mt_stats_kernel<<<dim3(B, MT_NCH), 256, 0, st>>>(ws + pl.t1, T * F, ws + pl.part)));
        LAUNCH(ctx, "decoder_train", (mt_stats_finalize_kernel<<<(B + 63) / 64, 64, 0, st>>>(ws + pl.part, B, (double)T * F, stt,
                                                                                             stt + B)));
        LAUNCH(ctx, "decoder_train", (mt_apply_kernel<<<1024, 256, 0, st>>>(ws + pl.t1, R * F, T * F, F, stt, stt + B,
                                                                            mask_tail_params(p), out)));
    } else {
        in_prelu_forward(ctx, ws + pl.s, B, T * W, p.n_w, p.n_b, p.p_w, stt, stt + B * 64, ws + pl.part, ws + pl.a);
        tail_forward<2>(ctx, ws + pl.a, R, W, p.c_w, p.c_b, out);
    }
}

void launch_decoder_train_backward(LaunchCtx ctx, int kind, const float* x, const float* dout, int B, int T, int Fe,
                                   const DecoderTrainParams& p, float* dx, const DecoderTrainParams& grad, float* ws) {
    hipStream_t st = ctx.stream;
    const DecPlan pl = dec_plan(B, T, Fe);
    const int W = 2 * Fe, F = W - 1;
    const long R = (long)B * T, Me = R * Fe;
    const RcGeom gs{B, T, Fe, Fe, 3, 1, 1};
    float* stt = ws + pl.st;
    float* g = ws + pl.g;
    float* g2 = ws + pl.g2;
    if (kind == 0) {
        const MaskTailP mp = mask_tail_params(p);
        const MaskTailG mg{grad.n_w, grad.n_b, grad.p_w, grad.f_w, grad.f_b, grad.po_w};
        LAUNCH(ctx, "decoder_train", (mt_dalphaf_kernel<<<F, 256, 0, st>>>(ws + pl.t1, dout, B, T, F, stt, stt + B, mp,
                                                                          grad.po_w)));
        hipMemcpyAsync(g2, dout, (size_t)R * F * sizeof(float), hipMemcpyDeviceToDevice, st);
        LAUNCH(ctx, "decoder_train", (mt_bwd_sums_kernel<<<dim3(B, MT_NCH), 256, 0, st>>>(ws + pl.t1, g2, T * F, F, stt, stt + B,
                                                                                        mp, ws + pl.part)));
        LAUNCH(ctx, "decoder_train", (mt_bwd_finalize_kernel<<<1, 64, 0, st>>>(ws + pl.part, B, (double)T * F, ws + pl.m,
                                                                              ws + pl.m + B, mg)));
        LAUNCH(ctx, "decoder_train", (mt_in_bwd_kernel<<<1024, 256, 0, st>>>(g2, ws + pl.t1, R * F, T * F, stt, stt + B, p.n_w,
                                                                             ws + pl.m, ws + pl.m + B)));
        tail_backward<1>(ctx, g2, ws + pl.s, R, W, p.c_w, g, grad.c_w, grad.c_b, ws + pl.wpart, ws + pl.cpart);
    } else {
        tail_backward<2>(ctx, dout, ws + pl.a, R, W, p.c_w, g, grad.c_w, grad.c_b, ws + pl.wpart, ws + pl.cpart);
        in_prelu_backward(ctx, ws + pl.s, g, B, T * W, p.n_w, p.n_b, p.p_w, stt, stt + B * 64, ws + pl.part, ws + pl.m,
                          ws + pl.m + B * 64, grad.n_w, grad.n_b, grad.p_w);
    }
    // sub-pixel conv: g = dL/ds on the 2 Fe wide grid = [Me, 128] rows in conv-channel order (64 r + c)
    if (!rc_backward<2>(ctx, g, ws + pl.d, ws + pl.imgT, gs, nullptr, grad.sp_w, ws + pl.wpart, grad.sp_b, ws + pl.cpart)) {
        LAUNCH(ctx, "decoder_train", (colsum_partial_kernel<<<FFN_COLSUM_BLOCKS, 256, 0, st>>>(g, Me, 128, ws + pl.cpart)));
        LAUNCH(ctx, "decoder_train", (reduce_partials_kernel<<<4, 1024, 0, st>>>(ws + pl.cpart, FFN_COLSUM_BLOCKS, 128, grad.sp_b)));
    }
    if (!rc_dgrad_x3(ctx, 2, g, p.sp_w, gs, ws + pl.d))
        LAUNCH(ctx, "rowconv_train", (rc_dgrad_kernel<2><<<(unsigned)((Me + 63) / 64), 256, 0, st>>>(g, ws + pl.imgT, gs, ws + pl.d)));
    launch_dense_train_backward(ctx, x, ws + pl.d, B, T, Fe, p.dense, dx, grad.dense, ws + pl.dense);
}

// ---- TSCNet.forward glue (generator.py:176-201) --------------------------------------------------------------------
// prologue: spec [B,2,T,F] (compressed re, im) -> xin [B,T,F,3] = (|spec|, re, im), the encoder's channels-last input
__global__ __launch_bounds__(256) void tsc_prologue_kernel(const float* __restrict__ spec, int B, long P, float* __restrict__ xin) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P, q = i - b * P;
        const float re = spec[(2 * b) * P + q], im = spec[(2 * b + 1) * P + q];
        xin[i * 3] = sqrtf(re * re + im * im);
        xin[i * 3 + 1] = re;
        xin[i * 3 + 2] = im;
    }
}
// epilogue: est = mask |spec| (cos, sin)(angle spec) + complex_out = mask (re, im) + complex_out     generator.py:192-199
__global__ __launch_bounds__(256) void tsc_epilogue_fwd_kernel(const float* __restrict__ spec, const float* __restrict__ mask,
                                                               const float* __restrict__ cplx, int B, long P,
                                                               float* __restrict__ est_real, float* __restrict__ est_imag) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P, q = i - b * P;
        const float re = spec[(2 * b) * P + q], im = spec[(2 * b + 1) * P + q], m = mask[i];
        est_real[i] = fmaf(m, re, cplx[2 * i]);
        est_imag[i] = fmaf(m, im, cplx[2 * i + 1]);
    }
}
__global__ __launch_bounds__(256) void tsc_epilogue_bwd_kernel(const float* __restrict__ spec, const float* __restrict__ d_real,
                                                               const float* __restrict__ d_imag, int B, long P,
                                                               float* __restrict__ dmask, float* __restrict__ dcplx) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P, q = i - b * P;
        const float re = spec[(2 * b) * P + q], im = spec[(2 * b + 1) * P + q], gr = d_real[i], gi = d_imag[i];
        dmask[i] = fmaf(gr, re, gi * im);
        dcplx[2 * i] = gr;
        dcplx[2 * i + 1] = gi;
    }
}
void launch_tsc_prologue(LaunchCtx ctx, const float* spec, int B, int T, int F, float* xin) {
    LAUNCH(ctx, "tsc_glue_train", (tsc_prologue_kernel<<<1024, 256, 0, ctx.stream>>>(spec, B, (long)T * F, xin)));
}
void launch_tsc_epilogue_forward(LaunchCtx ctx, const float* spec, const float* mask, const float* cplx, int B, int T, int F,
                                 float* est_real, float* est_imag) {
    LAUNCH(ctx, "tsc_glue_train", (tsc_epilogue_fwd_kernel<<<1024, 256, 0, ctx.stream>>>(spec, mask, cplx, B, (long)T * F, est_real,
                                                                                         est_imag)));
}
void launch_tsc_epilogue_backward(LaunchCtx ctx, const float* spec, const float* d_real, const float* d_imag, int B, int T, int F,
                                  float* dmask, float* dcplx) {
    LAUNCH(ctx, "tsc_glue_train", (tsc_epilogue_bwd_kernel<<<1024, 256, 0, ctx.stream>>>(spec, d_real, d_imag, B, (long)T * F, dmask,
                                                                                         dcplx)));
}

// ---- gradient of the non-adversarial generator loss (train.py:100-112, 133-148) with respect to est_real / est_imag ---
//   L = w_ri (mse(er, cr) + mse(ei, ci)) + w_mag mse(|e|, |c|) + w_time mean |istft(uncompress(e)) - clean|
// Time term: dL/d audio = w_time sign(est - clean) / (B La); the adjoint of torch.istft (window, overlap-add, envelope
// division, centre trim) followed by the adjoint of the one-sided inverse real DFT is a windowed forward real DFT of
// the envelope-normalised gradient: gU[f] = (c_f / N) sum_k df[k] e^{-2 pi i f k / N}, c_f = 1 at DC / Nyquist, else 2.
// Then the uncompress (utils.py:32-39) U = e |e|^(p-1), p = 1/0.3, in Cartesian form:
//   dL/der = s gUr + (p-1) |e|^(p-3) er (er gUr + ei gUi),   s = |e|^(p-1)   (same for ei).
__device__ __forceinline__ float hamming_periodic(int k, int n) { return 0.54f - 0.46f * cospif(2.0f * (float)k / (float)n); }

__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ est_real, const float* __restrict__ est_imag,
                                                       const float* __restrict__ clean_spec, const float* __restrict__ est_audio,
                                                       const float* __restrict__ clean_audio, int B, int T, int F, int nfft,
                                                       int hop, float w_ri, float w_mag, float w_time,
                                                       float* __restrict__ d_real, float* __restrict__ d_imag) {
    extern __shared__ float sm[];
    float* df = sm;                 // [nfft] windowed, envelope-normalised audio gradient of this frame
    float* tc = sm + nfft;          // [nfft] cos(2 pi j / nfft)
    float* ts = sm + 2 * nfft;      // [nfft] sin(2 pi j / nfft)
    const int b = blockIdx.x / T, t = blockIdx.x - b * T;
    const long La = (long)hop * (T - 1);
    const bool has_time = est_audio && clean_audio && w_time != 0.f;
    if (has_time) {
        const float ga = w_time / (float)((double)B * (double)La);
        for (int k = threadIdx.x; k < nfft; k += 256) {
            float sn, cs;
            sincospif(2.0f * (float)k / (float)nfft, &sn, &cs);
            tc[k] = cs; ts[k] = sn;
            const long n = (long)t * hop + k;                   // index in the centre-padded signal
            const long a = n - nfft / 2;                        // index in the returned audio
            float v = 0.f;
            if (a >= 0 && a < La) {
                float env = 0.f;
                int t0 = (int)((n - nfft + hop) / hop); if (t0 < 0) t0 = 0;
                int t1 = (int)(n / hop); if (t1 > T - 1) t1 = T - 1;
                for (int tt = t0; tt <= t1; ++tt) { const float w = hamming_periodic((int)(n - (long)tt * hop), nfft); env = fmaf(w, w, env); }
                const float d = est_audio[b * La + a] - clean_audio[b * La + a];
                const float sg = d > 0.f ? ga : (d < 0.f ? -ga : 0.f);
                v = sg * hamming_periodic(k, nfft) / env;
            }
            df[k] = v;
        }
        __syncthreads();
    }
    const long P = (long)T * F;
    const float invn = 1.0f / (float)((double)B * (double)P);
    const float p = 1.0f / 0.3f;
    for (int f = threadIdx.x; f < F; f += 256) {
        const long i = (long)b * P + (long)t * F + f;
        const float er = est_real[i], ei = est_imag[i];
        const float cr = clean_spec[(2L * b) * P + (long)t * F + f], ci = clean_spec[(2L * b + 1) * P + (long)t * F + f];
        const float me = sqrtf(er * er + ei * ei), mc = sqrtf(cr * cr + ci * ci);
        float gr = 2.f * w_ri * invn * (er - cr), gi = 2.f * w_ri * invn * (ei - ci);
        if (me > 0.f) {
            const float q = 2.f * w_mag * invn * (me - mc) / me;
            gr = fmaf(q, er, gr); gi = fmaf(q, ei, gi);
        }
        if (has_time) {
            float sr = 0.f, si = 0.f;
            int j = 0;                                          // (f k) mod nfft, advanced incrementally
            for (int k = 0; k < nfft; ++k) {
                sr = fmaf(df[k], tc[j], sr);
                si = fmaf(df[k], ts[j], si);
                j += f; if (j >= nfft) j -= nfft;
            }
            const float cf = (f == 0 || 2 * f == nfft) ? 1.f : 2.f;
            const float gur = cf * sr / (float)nfft, gui = (f == 0 || 2 * f == nfft) ? 0.f : -cf * si / (float)nfft;
            const float s = powf(me, p - 1.f), s3 = (p - 1.f) * powf(me, p - 3.f), dot = er * gur + ei * gui;
            gr += s * gur + s3 * er * dot;
            gi += s * gui + s3 * ei * dot;
        }
        d_real[i] = gr; d_imag[i] = gi;
    }
}
void launch_loss_backward(LaunchCtx ctx, const float* est_real, const float* est_imag, const float* clean_spec,
                          const float* est_audio, const float* clean_audio, int B, int T, int F, int nfft, int hop, float w_ri,
                          float w_mag, float w_time, float* d_real, float* d_imag) {
    LAUNCH(ctx, "loss_backward", (loss_bwd_kernel<<<B * T, 256, 3 * nfft * sizeof(float), ctx.stream>>>(
                                     est_real, est_imag, clean_spec, est_audio, clean_audio, B, T, F, nfft, hop, w_ri, w_mag, w_time,
                                     d_real, d_imag)));
}
