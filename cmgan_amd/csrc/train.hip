// train.hip - device pieces of the reference's training / validation step (src/train.py) that sit directly on the
// generator forward path: the non-adversarial loss terms of Trainer.calculate_generator_loss (train.py:124-151)
// as deterministic reductions (the scalars the data-parallel step all-reduces over RCCL).
#include "kernels.h"

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
