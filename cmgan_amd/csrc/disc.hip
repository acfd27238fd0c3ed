// disc.hip - the metric discriminator of the reference trainer (src/models/discriminator.py:29-64) with its backward,
// for the adversarial term of the generator loss and the discriminator step (src/train.py:124-171, 185-205).
//
//   D(x, y):  cat -> 4 x [ spectral-norm Conv2d(4x4, stride 2, pad 1, no bias) -> InstanceNorm2d(affine) -> PReLU ]
//             (2 -> 16 -> 32 -> 64 -> 128 channels) -> global max pool -> spectral-norm Linear(128, 64) -> Dropout(0.3)
//             -> PReLU(64) -> spectral-norm Linear(64, 1) -> LearnableSigmoid(1)
//
// The network is 0.2 % of the generator's arithmetic (105 M MACs per clip), so the kernels are plain and correctness
// first: channels-last activations [B, T', F', C] (the reference's [B, C, F', T'] with the two spatial axes swapped, so
// tap (kt, kf) here is weight[co][ci][kh = kf][kw = kt]), direct convolutions with one thread per output element and
// weights re-laid so that the fastest thread index is contiguous, split-K weight gradients into fixed-shape partial
// slabs, fp32 everywhere, fixed reduction orders (deterministic).
//
// Spectral norm (torch.nn.utils.spectral_norm, the legacy hook the reference uses): every TRAIN-mode forward runs one
// power iteration v <- normalize(W^T u), u <- normalize(W v) outside the graph and uses W / sigma, sigma = u^T W v.
// The forward keeps the (u, v, sigma) it used in the workspace; the backward is
//   dL/dW = (G - (sum G . W_eff) u v^T) / sigma,   G = dL/dW_eff.
#include "train.h"

#define DC_NCH 8             // position chunks of the InstanceNorm partial sums
#define DC_SPLIT 32          // position chunks of the conv weight gradient

// ---- spectral norm ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dc_block_sum(float v, float* red) {
    const int t = threadIdx.x;
    red[t] = v;
    __syncthreads();
    for (int d = blockDim.x >> 1; d >= 1; d >>= 1) {
        if (t < d) red[t] += red[t + d];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// one block of 1024 threads; W [out, K] row-major (out <= 128, K <= 1024).  W^T u: one column per thread (coalesced
// rows); W v: one row per wave at a time with a DPP wave reduction - no block barrier per row (the 256-thread form with
// a block reduction per row took ~100 us per launch, 18 launches per step).
__global__ __launch_bounds__(1024) void sn_power_kernel(const float* __restrict__ W, int out, int K, float* __restrict__ u_state,
                                                        float* __restrict__ v_state, int update, float* __restrict__ u_used,
                                                        float* __restrict__ v_used, float* __restrict__ sigma) {
    __shared__ float su[128], sv[1024], swv[128], red[1024];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (t < out) su[t] = u_state[t];
    if (t < K) sv[t] = v_state[t];
    __syncthreads();
    if (update) {
        float mine = 0.f;
        if (t < K) {
            float s0 = 0.f, s1 = 0.f;
            int o = 0;
            for (; o + 1 < out; o += 2) {
                s0 = fmaf(W[(long)o * K + t], su[o], s0);
                s1 = fmaf(W[(long)(o + 1) * K + t], su[o + 1], s1);
            }
            if (o < out) s0 = fmaf(W[(long)o * K + t], su[o], s0);
            mine = s0 + s1;
        }
        const float nrm = sqrtf(dc_block_sum(mine * mine, red));
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);
        if (t < K) sv[t] = mine * inv;
        __syncthreads();
    }
    for (int o = wv; o < out; o += 16) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s = fmaf(W[(long)o * K + k], sv[k], s);
        s = wave_sum(s);
        if (lane == 0) swv[o] = s;
    }
    __syncthreads();
    if (update) {
        const float nrm = sqrtf(dc_block_sum(t < out ? swv[t] * swv[t] : 0.f, red));
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);
        if (t < out) su[t] = swv[t] * inv;
        __syncthreads();
    }
    const float sg = dc_block_sum(t < out ? su[t] * swv[t] : 0.f, red);
    if (t == 0) sigma[0] = sg;
    if (t < out) { u_used[t] = su[t]; if (update) u_state[t] = su[t]; }
    if (t < K) { v_used[t] = sv[t]; if (update) v_state[t] = sv[t]; }
}

// effective conv weights W / sigma in the two thread-friendly layouts: wf [tap][ci][co], wb [tap][co][ci]
__global__ void dc_pack_kernel(const float* __restrict__ W, int Co, int Ci, const float* __restrict__ sigma,
                               float* __restrict__ wf, float* __restrict__ wb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Co * Ci * 16) return;
    const int kw = e & 3, kh = (e >> 2) & 3, ci = (e >> 4) % Ci, co = (e >> 4) / Ci;
    const float v = W[e] / sigma[0];
    const int tap = kw * 4 + kh;                             // (kt, kf) = (kw, kh)
    wf[((long)tap * Ci + ci) * Co + co] = v;
    wb[((long)tap * Co + co) * Ci + ci] = v;
}

// ---- 4x4 stride-2 pad-1 convolution ----------------------------------------------------------------------------------
struct DcGeom { int B, Ti, Fi, Ci, To, Fo, Co; };

// Forward convolution and data gradient as one LDS-tiled register-blocked GEMM over gathered rows:
//   MODE 0 (forward):  out[pos][co] = sum_m xcol[pos][m] wf[m][co],  m = tap * Ci + ci  (K = 16 Ci, N = Co), pos = (b, to, fo)
//   MODE 1 (dgrad):    din[pos][ci] = sum_m dcol[pos][m] wb[row(m)][ci], m = j * Co + co (K = 4 Co, N = Ci): an input
//                      position (ti, fi) = (2a + rt, 2c + rf) receives the 2 x 2 taps kt = 1 - rt + 2 jt, kf = 1 - rf + 2 jf
//                      from the outputs (a + rt - jt, c + rf - jf); blockIdx.y = the parity class (rt, rf), pos = (b, a, c)
// A block owns TPF positions x all N columns (TPF * N = 4096), a thread RP positions x RN columns; K is walked in chunks of KC
// rows staged once per chunk (consecutive m = consecutive floats of the source row).  (Round 2: one thread per output
// element, 9 ms per step at batch 32 for 10 GMAC.)
template <int MODE, int TPF, int RP, int RN, int KC>
__global__ __launch_bounds__(256) void dc_conv_gemm_kernel(const float* __restrict__ src, const float* __restrict__ wmat, DcGeom gm,
                                                           int lgC, float* __restrict__ dst) {
    __shared__ __attribute__((aligned(16))) float xs[TPF * (KC + 1)];
    __shared__ __attribute__((aligned(16))) float wsm[KC * 128];
    __shared__ int p_b[TPF], p_x[TPF], p_y[TPF];
    const int tid = threadIdx.x;
    const int N = MODE == 0 ? gm.Co : gm.Ci, Kt = MODE == 0 ? 16 * gm.Ci : 4 * gm.Co, C = 1 << lgC;
    const int rt = MODE == 1 ? (int)(blockIdx.y >> 1) : 0, rf = MODE == 1 ? (int)(blockIdx.y & 1) : 0;
    const int X = MODE == 0 ? gm.To : (gm.Ti - rt + 1) / 2, Y = MODE == 0 ? gm.Fo : (gm.Fi - rf + 1) / 2;
    const long P = (long)gm.B * X * Y, p0 = (long)blockIdx.x * TPF;
    if (p0 >= P) return;
    for (int p = tid; p < TPF; p += 256) {
        const long pos = p0 + p;
        int b = -1, x = 0, y = 0;
        if (pos < P) {
            y = (int)(pos % Y);
            const long r = pos / Y;
            x = (int)(r % X);
            b = (int)(r / X);
        }
        p_b[p] = b; p_x[p] = x; p_y[p] = y;
    }
    const int ncol = N / RN, tn = tid % ncol, tp = tid / ncol;
    float acc[RP][RN];
#pragma unroll
    for (int i = 0; i < RP; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < Kt; k0 += KC) {
        __syncthreads();                                          // position table ready / previous chunk consumed
        for (int e = tid; e < TPF * KC; e += 256) {
            const int p = e / KC, kk = e - p * KC, m = k0 + kk;
            const int hi = m >> lgC, ch = m & (C - 1);            // tap | 2x2 tap index, channel
            const int b = p_b[p];
            float v = 0.f;
            if (MODE == 0) {
                const int ti = 2 * p_x[p] - 1 + (hi >> 2), fi = 2 * p_y[p] - 1 + (hi & 3);
                if (b >= 0 && ti >= 0 && ti < gm.Ti && fi >= 0 && fi < gm.Fi)
                    v = src[(((long)b * gm.Ti + ti) * gm.Fi + fi) * gm.Ci + ch];
            } else {
                const int to = p_x[p] + rt - (hi >> 1), fo = p_y[p] + rf - (hi & 1);
                if (b >= 0 && to >= 0 && to < gm.To && fo >= 0 && fo < gm.Fo)
                    v = src[(((long)b * gm.To + to) * gm.Fo + fo) * gm.Co + ch];
            }
            xs[p * (KC + 1) + kk] = v;
        }
        for (int e = tid; e < KC * N; e += 256) {
            const int kk = e / N, n = e - kk * N, m = k0 + kk;
            long row = m;
            if (MODE == 1) {
                const int j = m >> lgC, co = m & (C - 1);
                row = (long)((1 - rt + 2 * (j >> 1)) * 4 + (1 - rf + 2 * (j & 1))) * gm.Co + co;
            }
            wsm[kk * N + n] = wmat[row * N + n];
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < KC; ++k) {
            float xr[RP], wr[RN];
#pragma unroll
            for (int i = 0; i < RP; ++i) xr[i] = xs[(tp * RP + i) * (KC + 1) + k];
#pragma unroll
            for (int j = 0; j < RN; ++j) wr[j] = wsm[k * N + tn * RN + j];
#pragma unroll
            for (int i = 0; i < RP; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = fmaf(xr[i], wr[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
        const int p = tp * RP + i, b = p_b[p];
        if (b < 0) continue;
        long o;
        if (MODE == 0) o = (p0 + p) * gm.Co;
        else o = (((long)b * gm.Ti + 2 * p_x[p] + rt) * gm.Fi + 2 * p_y[p] + rf) * gm.Ci;
#pragma unroll
        for (int j = 0; j < RN; ++j) dst[o + tn * RN + j] = acc[i][j];
    }
}

// tile shapes per layer (TPF * N = 4096; the two-channel data gradient of the first layer: 512 x 2)
static void dc_launch_fwd(LaunchCtx ctx, const float* in, const float* wf, const DcGeom& gm, float* out) {
    const long P = (long)gm.B * gm.To * gm.Fo;
    int lg = 0;
    while ((1 << lg) < gm.Ci) ++lg;
    hipStream_t st = ctx.stream;
#define DC_FWD(TPF) LAUNCH(ctx, "disc_conv_fwd", (dc_conv_gemm_kernel<0, TPF, 4, 4, 32><<<(unsigned)((P + TPF - 1) / TPF), 256, 0, st>>>(in, wf, gm, lg, out)))
    if (gm.Co == 16) DC_FWD(256);
    else if (gm.Co == 32) DC_FWD(128);
    else if (gm.Co == 64) DC_FWD(64);
    else DC_FWD(32);
#undef DC_FWD
}
static void dc_launch_dgrad(LaunchCtx ctx, const float* dout, const float* wb, const DcGeom& gm, float* din) {
    const long P = (long)gm.B * ((gm.Ti + 1) / 2) * ((gm.Fi + 1) / 2);   // the largest parity class
    int lg = 0;
    while ((1 << lg) < gm.Co) ++lg;
    hipStream_t st = ctx.stream;
#define DC_DG(TPF, RP, RN, KC) LAUNCH(ctx, "disc_conv_bwd", (dc_conv_gemm_kernel<1, TPF, RP, RN, KC><<<dim3((unsigned)((P + TPF - 1) / TPF), 4), 256, 0, st>>>(dout, wb, gm, lg, din)))
    if (gm.Ci == 2) DC_DG(512, 2, 2, 16);
    else if (gm.Ci == 16) DC_DG(256, 4, 4, 32);
    else if (gm.Ci == 32) DC_DG(128, 4, 4, 32);
    else DC_DG(64, 4, 4, 32);
#undef DC_DG
}

// partial[s][tap][ci][co] = sum over the s-th chunk of output positions of dout[pos][co] * in[pos shifted by tap][ci]:
// the token contraction G = X_col^T dz (X_col = the im2col rows, m = tap * Ci + ci) as an LDS-tiled register-blocked
// product.  A block owns MB rows x all Co columns of G (MB * Co = 8192 except for the first layer: 32 x 16) and one
// chunk of DW_TP-position tiles; a thread owns RM x RN accumulators, the tile's im2col slice and dz rows are staged once
// (consecutive m = consecutive (kf, ci) = consecutive floats of the input row) and read back as broadcast b128s.
// (Round 2 had one thread per (ci, co) pair walking all positions: 43 ms per step at batch 32 for 7 GMAC.)
#define DW_TP 32
template <int RM, int RN>
__global__ __launch_bounds__(256) void dc_conv_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ in,
                                                            DcGeom gm, int MB, int lgCi, int tiles_total, int tiles_per_chunk,
                                                            float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float xs[DW_TP * 256];       // [p][MB]
    __shared__ __attribute__((aligned(16))) float dz[DW_TP * 128];       // [p][Co]
    __shared__ int p_b[DW_TP], p_ti[DW_TP], p_fi[DW_TP];
    const int tid = threadIdx.x, s = blockIdx.x, mg = blockIdx.y;
    const int ncol = gm.Co / RN, tn = tid % ncol, tm = tid / ncol;
    const long P = (long)gm.B * gm.To * gm.Fo;
    float acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) acc[i][j] = 0.f;
    const int t_begin = s * tiles_per_chunk;
    const int t_end = t_begin + tiles_per_chunk < tiles_total ? t_begin + tiles_per_chunk : tiles_total;
    for (int tile = t_begin; tile < t_end; ++tile) {
        const long p0 = (long)tile * DW_TP;
        __syncthreads();                                              // the previous tile is consumed
        if (tid < DW_TP) {
            const long pos = p0 + tid;
            int b = -1, ti = 0, fi = 0;
            if (pos < P) {
                const int fo = (int)(pos % gm.Fo);
                const long r = pos / gm.Fo;
                const int to = (int)(r % gm.To);
                b = (int)(r / gm.To);
                ti = 2 * to - 1;
                fi = 2 * fo - 1;
            }
            p_b[tid] = b; p_ti[tid] = ti; p_fi[tid] = fi;
        }
        __syncthreads();
        for (int e = tid; e < DW_TP * MB; e += 256) {
            const int p = e / MB, ml = e - p * MB, m = mg * MB + ml;
            const int tap = m >> lgCi, ci = m & (gm.Ci - 1);
            const int b = p_b[p], ti = p_ti[p] + (tap >> 2), fi = p_fi[p] + (tap & 3);
            float v = 0.f;
            if (b >= 0 && ti >= 0 && ti < gm.Ti && fi >= 0 && fi < gm.Fi)
                v = in[(((long)b * gm.Ti + ti) * gm.Fi + fi) * gm.Ci + ci];
            xs[p * MB + ml] = v;
        }
        for (int e = tid; e < DW_TP * gm.Co; e += 256) {
            const int p = e / gm.Co;
            dz[e] = p_b[p] >= 0 ? dout[p0 * gm.Co + e] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < DW_TP; ++k) {
            float xr[RM], dr[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) xr[i] = xs[k * MB + tm * RM + i];
#pragma unroll
            for (int j = 0; j < RN; ++j) dr[j] = dz[k * gm.Co + tn * RN + j];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = fmaf(xr[i], dr[j], acc[i][j]);
        }
    }
    float* out = partial + ((long)s * 16 * gm.Ci + (long)mg * MB + tm * RM) * gm.Co + tn * RN;
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) out[(long)i * gm.Co + j] = acc[i][j];
}
// G[e] = sum_s partial[s][e]: a block is 64 elements x 16 slab groups (group g adds slabs g, g + 16, ... with four loads in
// flight, the groups are combined in group order: a fixed order for a given launch shape).  One thread per element walking up to
// 1024 slabs was one dependent fetch per slab: 105 us per launch for 6 MB, twelve launches per step.
__global__ __launch_bounds__(1024) void dc_reduce_kernel(const float* __restrict__ partial, int nsplit, long n, float* __restrict__ G) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int k = grp;
        for (; k + 48 < nsplit; k += 64) {
            s0 += partial[(long)k * n + e];
            s1 += partial[(long)(k + 16) * n + e];
            s2 += partial[(long)(k + 32) * n + e];
            s3 += partial[(long)(k + 48) * n + e];
        }
        for (; k < nsplit; k += 16) s0 += partial[(long)k * n + e];
    }
    red[grp][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && e < n) {
        float s = red[0][col];
#pragma unroll
        for (int g = 1; g < 16; ++g) s += red[g][col];
        G[e] = s;
    }
}
// spectral-norm backward for a conv layer: G and W_eff in the [tap][ci][co] layout, dW in the parameter's layout
__global__ __launch_bounds__(1024) void sn_conv_finish_kernel(const float* __restrict__ G, const float* __restrict__ wf, int Co,
                                                              int Ci, const float* __restrict__ u, const float* __restrict__ v,
                                                              const float* __restrict__ sigma, float* __restrict__ dW) {
    __shared__ float red[1024];
    const int n = Co * Ci * 16;
    float part = 0.f;
    for (int e = threadIdx.x; e < n; e += 1024) part = fmaf(G[e], wf[e], part);
    const float dot = dc_block_sum(part, red);
    const float inv = 1.0f / sigma[0];
    for (int e = threadIdx.x; e < n; e += 1024) {                 // e in the PARAMETER layout [co][ci][kh][kw]
        const int kw = e & 3, kh = (e >> 2) & 3, ci = (e >> 4) % Ci, co = (e >> 4) / Ci;
        const int tap = kw * 4 + kh;
        const float g = G[((long)tap * Ci + ci) * Co + co];
        dW[e] = (g - dot * u[co] * v[e - co * Ci * 16]) * inv;
    }
}

// ---- InstanceNorm2d(C, affine) + PReLU(C), C in {16, 32, 64, 128}, planes [B, P, C] ------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void gin_sums_kernel(const float* __restrict__ z, float* __restrict__ ga, int P, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ alpha, float* __restrict__ partial) {
    __shared__ float red[256][3];
    const int c = threadIdx.x % C, sub = threadIdx.x / C, nsub = 256 / C;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int per = (P + DC_NCH - 1) / DC_NCH, p0 = chunk * per, p1 = p0 + per < P ? p0 + per : P;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float mu = 0.f, rs = 0.f, gm = 0.f, bt = 0.f, al = 0.f;
    if (MODE == 1) { mu = mean[b * C + c]; rs = rstd[b * C + c]; gm = gamma[c]; bt = beta[c]; al = alpha[c]; }
    for (int p = p0 + sub; p < p1; p += nsub) {
        const long i = ((long)b * P + p) * C + c;
        const float zv = z[i];
        if (MODE == 0) {
            s0 += zv;
            s1 = fmaf(zv, zv, s1);
        } else {
            const float zh = (zv - mu) * rs, n = zh * gm + bt, gv = ga[i];
            const float dn = n < 0.f ? gv * al : gv;
            ga[i] = dn;
            s0 += dn;
            s1 = fmaf(dn, zh, s1);
            s2 += n < 0.f ? gv * n : 0.f;
        }
    }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2;
    __syncthreads();
    if (sub == 0) {
        for (int k = 1; k < nsub; ++k) { s0 += red[k * C + c][0]; s1 += red[k * C + c][1]; s2 += red[k * C + c][2]; }
        const long o = (((long)b * DC_NCH + chunk) * C + c) * 3;
        partial[o] = s0; partial[o + 1] = s1; partial[o + 2] = s2;
    }
}
__global__ void gin_stats_finalize_kernel(const float* __restrict__ partial, int B, int C, double count, float* __restrict__ mean,
                                          float* __restrict__ rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < DC_NCH; ++k) {
        const long o = (((long)b * DC_NCH + k) * C + c) * 3;
        s1 += (double)partial[o];
        s2 += (double)partial[o + 1];
    }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
}
__global__ __launch_bounds__(256) void gin_norm_prelu_kernel(const float* __restrict__ z, long total, int P, int C,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ alpha, float* __restrict__ a) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long bc = (i / C) / P * C + c;
        const float n = (z[i] - mean[bc]) * rstd[bc] * gamma[c] + beta[c];
        a[i] = n >= 0.f ? n : alpha[c] * n;
    }
}
// one block of 1024 threads = C channels x 1024 / C clip groups (C in {16, 32, 64, 128}): group g finalises clips g, g + G, ...
// (means of (b, c) from the DC_NCH chunk partials in fp64), the parameter gradients add the groups' sums in group order - a
// fixed order for a given launch shape.  (One thread per channel walking all clips and chunks: 40 us per launch, twelve per step.)
__global__ __launch_bounds__(1024) void gin_bwd_finalize_kernel(const float* __restrict__ partial, int B, int C, double count,
                                                                float* __restrict__ m1, float* __restrict__ m2,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dalpha) {
    __shared__ double red[3][1024];
    const int tid = threadIdx.x, c = tid % C, grp = tid / C, G = 1024 / C;
    double g = 0.0, bsum = 0.0, a = 0.0;
    for (int b = grp < G ? grp : B; b < B; b += G) {              // (threads past the last full group idle: any C <= 1024 works)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < DC_NCH; ++k) {
            const long o = (((long)b * DC_NCH + k) * C + c) * 3;
            s0 += (double)partial[o]; s1 += (double)partial[o + 1]; s2 += (double)partial[o + 2];
        }
        m1[b * C + c] = (float)(s0 / count);
        m2[b * C + c] = (float)(s1 / count);
        bsum += s0; g += s1; a += s2;
    }
    red[0][tid] = g; red[1][tid] = bsum; red[2][tid] = a;
    __syncthreads();
    if (grp == 0) {
        for (int gg = 1; gg < G; ++gg) { g += red[0][gg * C + c]; bsum += red[1][gg * C + c]; a += red[2][gg * C + c]; }
        dgamma[c] = (float)g; dbeta[c] = (float)bsum; dalpha[c] = (float)a;
    }
}
__global__ __launch_bounds__(256) void gin_in_bwd_kernel(float* __restrict__ dn, const float* __restrict__ z, long total, int P,
                                                         int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ m1,
                                                         const float* __restrict__ m2) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long bc = (i / C) / P * C + c;
        const float zh = (z[i] - mean[bc]) * rstd[bc];
        dn[i] = gamma[c] * rstd[bc] * (dn[i] - m1[bc] - zh * m2[bc]);
    }
}

// ---- global max pool over the plane (AdaptiveMaxPool2d(1)), 128 channels ---------------------------------------------
__global__ __launch_bounds__(128) void dc_maxpool_kernel(const float* __restrict__ a, int P, float* __restrict__ pooled,
                                                         int* __restrict__ idx) {
    const int b = blockIdx.x, c = threadIdx.x;
    float best = a[((long)b * P) * 128 + c];
    int bi = 0;
    for (int p = 1; p < P; ++p) {
        const float v = a[((long)b * P + p) * 128 + c];
        if (v > best) { best = v; bi = p; }
    }
    pooled[b * 128 + c] = best;
    idx[b * 128 + c] = bi;
}
__global__ __launch_bounds__(128) void dc_maxpool_bwd_kernel(const float* __restrict__ dpool, const int* __restrict__ idx, int P,
                                                             float* __restrict__ da) {
    const int b = blockIdx.x, c = threadIdx.x;
    da[((long)b * P + idx[b * 128 + c]) * 128 + c] = dpool[b * 128 + c];
}

// ---- head: SN-Linear(128, 64) -> Dropout mask -> PReLU(64) -> SN-Linear(64, 1) -> sigmoid(slope .) --------------------
struct DcHead {
    const float *w1, *b1, *sig1, *alpha, *w2, *b2, *sig2, *slope;
};
__global__ __launch_bounds__(64) void dc_head_fwd_kernel(const float* __restrict__ pooled, DcHead hd, const float* __restrict__ mask,
                                                         float* __restrict__ h1, float* __restrict__ x2, float* __restrict__ score) {
    __shared__ float red[64];
    const int b = blockIdx.x, j = threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < 128; ++k) s = fmaf(hd.w1[j * 128 + k], pooled[b * 128 + k], s);
    const float h = s / hd.sig1[0] + hd.b1[j];
    h1[b * 64 + j] = h;
    const float d5 = mask ? h * mask[b * 64 + j] : h;
    const float a5 = d5 >= 0.f ? d5 : hd.alpha[j] * d5;
    red[j] = hd.w2[j] * a5;
    __syncthreads();
    for (int d = 32; d >= 1; d >>= 1) { if (j < d) red[j] += red[j + d]; __syncthreads(); }
    if (j == 0) {
        const float x = red[0] / hd.sig2[0] + hd.b2[0];
        x2[b] = x;
        score[b] = 1.0f / (1.0f + __expf(-hd.slope[0] * x));
    }
}
// per sample: dh1 [B,64], ds2 [B], dslope_b [B], da5n [B,64] (= da5 d5 [d5 < 0]) and dL/dpooled [B,128]
__global__ __launch_bounds__(128) void dc_head_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ h1,
                                                          const float* __restrict__ x2, const float* __restrict__ dscore, DcHead hd,
                                                          const float* __restrict__ mask, float* __restrict__ dh1,
                                                          float* __restrict__ ds2, float* __restrict__ dslope_b,
                                                          float* __restrict__ dalpha_b, float* __restrict__ dpool) {
    __shared__ float sdh[64];
    const int b = blockIdx.x, t = threadIdx.x;
    const float sl = hd.slope[0], x = x2[b];
    const float y = 1.0f / (1.0f + __expf(-sl * x));
    const float dy = dscore[b] * y * (1.0f - y);
    const float s2 = dy * sl;
    if (t == 0) { ds2[b] = s2; dslope_b[b] = dy * x; }
    if (t < 64) {
        const float mk = mask ? mask[b * 64 + t] : 1.0f;
        const float d5 = h1[b * 64 + t] * mk;
        const float da5 = s2 * hd.w2[t] / hd.sig2[0];
        const float dd5 = d5 >= 0.f ? da5 : da5 * hd.alpha[t];
        dalpha_b[b * 64 + t] = d5 < 0.f ? da5 * d5 : 0.f;
        const float v = dd5 * mk;
        dh1[b * 64 + t] = v;
        sdh[t] = v;
    }
    __syncthreads();
    float s = 0.f;
    for (int j = 0; j < 64; ++j) s = fmaf(sdh[j], hd.w1[j * 128 + t], s);
    dpool[b * 128 + t] = s / hd.sig1[0];
}
// sums over the batch (in sample order) + the spectral-norm backward of the two Linear layers; one block of 1024
struct DcHeadGrads { float *w1, *b1, *alpha, *w2, *b2, *slope; };
__global__ __launch_bounds__(1024) void dc_head_finish_kernel(const float* __restrict__ pooled, const float* __restrict__ h1,
                                                              const float* __restrict__ dh1, const float* __restrict__ ds2,
                                                              const float* __restrict__ dslope_b, const float* __restrict__ dalpha_b,
                                                              const float* __restrict__ mask, int B, DcHead hd,
                                                              const float* __restrict__ u1, const float* __restrict__ v1,
                                                              const float* __restrict__ u2, const float* __restrict__ v2,
                                                              DcHeadGrads gr) {
    __shared__ float red[1024];
    __shared__ float G2[64];
    const int t = threadIdx.x;
    // fc1: G1 [64,128] = sum_b dh1_b pooled_b^T
    float g1[8];
    float part = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = t + 1024 * q, j = e >> 7, k = e & 127;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(dh1[b * 64 + j], pooled[b * 128 + k], s);
        g1[q] = s;
        part = fmaf(s, hd.w1[e] / hd.sig1[0], part);
    }
    const float dot1 = dc_block_sum(part, red);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int e = t + 1024 * q, j = e >> 7, k = e & 127;
        gr.w1[e] = (g1[q] - dot1 * u1[j] * v1[k]) / hd.sig1[0];
    }
    // fc2: G2 [1,64] = sum_b ds2_b a5_b^T ; biases, PReLU slope, LearnableSigmoid slope
    float p2 = 0.f;
    if (t < 64) {
        float s = 0.f, sb = 0.f, sa = 0.f;
        for (int b = 0; b < B; ++b) {
            const float mk = mask ? mask[b * 64 + t] : 1.0f;
            const float d5 = h1[b * 64 + t] * mk;
            const float a5 = d5 >= 0.f ? d5 : hd.alpha[t] * d5;
            s = fmaf(ds2[b], a5, s);
            sb += dh1[b * 64 + t];
            sa += dalpha_b[b * 64 + t];
        }
        G2[t] = s;
        gr.b1[t] = sb;
        gr.alpha[t] = sa;
        p2 = s * hd.w2[t] / hd.sig2[0];
    }
    const float dot2 = dc_block_sum(p2, red);
    if (t < 64) gr.w2[t] = (G2[t] - dot2 * u2[0] * v2[t]) / hd.sig2[0];
    if (t == 0) {
        float sb = 0.f, ss = 0.f;
        for (int b = 0; b < B; ++b) { sb += ds2[b]; ss += dslope_b[b]; }
        gr.b2[0] = sb;
        gr.slope[0] = ss;
    }
}

// ---- input / loss glue -------------------------------------------------------------------------------------------------
// xy [B,T,F,2] = (|clean|, |est|); est = clean when est_real is NULL   (train.py:102-103, 126-128, 163-167)
__global__ __launch_bounds__(256) void mag_pair_kernel(const float* __restrict__ clean_spec, const float* __restrict__ est_real,
                                                       const float* __restrict__ est_imag, int B, long P, float* __restrict__ xy) {
    const long total = (long)B * P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / P, q = i - b * P;
        const float cr = clean_spec[(2 * b) * P + q], ci = clean_spec[(2 * b + 1) * P + q];
        const float mc = sqrtf(cr * cr + ci * ci);
        float me = mc;
        if (est_real) { const float er = est_real[i], ei = est_imag[i]; me = sqrtf(er * er + ei * ei); }
        xy[2 * i] = mc;
        xy[2 * i + 1] = me;
    }
}
// d_real / d_imag += scale * dxy[..., 1] * (er, ei) / |e|
__global__ __launch_bounds__(256) void mag_pair_bwd_kernel(const float* __restrict__ est_real, const float* __restrict__ est_imag,
                                                           const float* __restrict__ dxy, long total, float scale,
                                                           float* __restrict__ d_real, float* __restrict__ d_imag) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float er = est_real[i], ei = est_imag[i], me = sqrtf(er * er + ei * ei);
        if (me > 0.f) {
            const float q = scale * dxy[2 * i + 1] / me;
            d_real[i] = fmaf(q, er, d_real[i]);
            d_imag[i] = fmaf(q, ei, d_imag[i]);
        }
    }
}
// loss = mean (score - target)^2 (target = 1 when NULL), dscore = scale * 2 (score - target) / B      train.py:129-131, 168-170
__global__ void score_mse_kernel(const float* __restrict__ score, const float* __restrict__ target, int B, float scale,
                                 float* __restrict__ loss, float* __restrict__ dscore) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
        const float d = score[b] - (target ? target[b] : 1.0f);
        s += (double)d * d;
        if (dscore) dscore[b] = scale * 2.0f * d / (float)B;
    }
    loss[0] = (float)(s / B);
}

// ---- host side -----------------------------------------------------------------------------------------------------------
struct DcLayer { int Ti, Fi, Ci, To, Fo, Co; };
struct DcPlan {
    DcLayer L[4];
    size_t wf[4], wb[4], z[4], a[4], g[4], mean[4], rstd[4];
    size_t uu[6], vv[6], sigma;                  // the (u, v, sigma) this forward used, per spectral-norm layer
    size_t pooled, idx, h1, x2, dh1, ds2, dslope, dalpha, dpool, part, m12, wpart, G, total;
};
static DcPlan dc_plan(int B, int T, int F) {
    DcPlan p;
    const int ch[5] = {2, 16, 32, 64, 128};
    int t = T, f = F;
    size_t cur = 0;
    auto take = [&](size_t n) { const size_t o = cur; cur += (n + 63) & ~(size_t)63; return o; };
    size_t maxw = 0;
    for (int i = 0; i < 4; ++i) {
        p.L[i] = {t, f, ch[i], t / 2, f / 2, ch[i + 1]};
        t /= 2; f /= 2;
        const size_t nw = (size_t)16 * ch[i] * ch[i + 1], na = (size_t)B * p.L[i].To * p.L[i].Fo * ch[i + 1];
        p.wf[i] = take(nw); p.wb[i] = take(nw);
        p.z[i] = take(na); p.a[i] = take(na); p.g[i] = take(na);
        p.mean[i] = take((size_t)B * ch[i + 1]); p.rstd[i] = take((size_t)B * ch[i + 1]);
        p.uu[i] = take(ch[i + 1]); p.vv[i] = take((size_t)16 * ch[i]);
        if (nw > maxw) maxw = nw;
    }
    p.uu[4] = take(64); p.vv[4] = take(128); p.uu[5] = take(1); p.vv[5] = take(64);
    p.sigma = take(8);
    p.pooled = take((size_t)B * 128); p.idx = take((size_t)B * 128);
    p.h1 = take((size_t)B * 64); p.x2 = take(B);
    p.dh1 = take((size_t)B * 64); p.ds2 = take(B); p.dslope = take(B); p.dalpha = take((size_t)B * 64);
    p.dpool = take((size_t)B * 128);
    p.part = take((size_t)B * DC_NCH * 128 * 3);
    p.m12 = take((size_t)2 * B * 128);
    p.wpart = take((size_t)DC_SPLIT * maxw);
    p.G = take(maxw);
    p.total = cur;
    return p;
}
size_t disc_ws_floats(int B, int T, int F) { return dc_plan(B, T, F).total; }
bool disc_shape_ok(int T, int F) { return T >= 16 && F >= 16; }

static DcHead dc_head(const DiscParams& p, float* ws, const DcPlan& pl) {
    return DcHead{p.fc1_w, p.fc1_b, ws + pl.sigma + 4, p.prelu5_w, p.fc2_w, p.fc2_b, ws + pl.sigma + 5, p.slope};
}

void launch_disc_forward(LaunchCtx ctx, const float* xy, int B, int T, int F, const DiscParams& p, const float* mask,
                         int update_uv, float* score, float* ws) {
    hipStream_t st = ctx.stream;
    const DcPlan pl = dc_plan(B, T, F);
    const float* in = xy;
    for (int i = 0; i < 4; ++i) {
        const DcLayer& L = pl.L[i];
        const DcGeom gm{B, L.Ti, L.Fi, L.Ci, L.To, L.Fo, L.Co};
        LAUNCH(ctx, "disc_spectral_norm", (sn_power_kernel<<<1, 1024, 0, st>>>(p.conv_w[i], L.Co, L.Ci * 16, p.conv_u[i], p.conv_v[i],
                                                                             update_uv, ws + pl.uu[i], ws + pl.vv[i],
                                                                             ws + pl.sigma + i)));
        const int nw = 16 * L.Ci * L.Co;
        LAUNCH(ctx, "disc_spectral_norm", (dc_pack_kernel<<<(nw + 255) / 256, 256, 0, st>>>(p.conv_w[i], L.Co, L.Ci, ws + pl.sigma + i,
                                                                                           ws + pl.wf[i], ws + pl.wb[i])));
        const long total = (long)B * L.To * L.Fo * L.Co;
        const int P = L.To * L.Fo;
        const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        dc_launch_fwd(ctx, in, ws + pl.wf[i], gm, ws + pl.z[i]);
        LAUNCH(ctx, "disc_norm", (gin_sums_kernel<0><<<dim3(B, DC_NCH), 256, 0, st>>>(ws + pl.z[i], nullptr, P, L.Co, nullptr, nullptr,
                                                                                     nullptr, nullptr, nullptr, ws + pl.part)));
        LAUNCH(ctx, "disc_norm", (gin_stats_finalize_kernel<<<(B * L.Co + 255) / 256, 256, 0, st>>>(ws + pl.part, B, L.Co, (double)P,
                                                                                                   ws + pl.mean[i], ws + pl.rstd[i])));
        LAUNCH(ctx, "disc_norm", (gin_norm_prelu_kernel<<<grid, 256, 0, st>>>(ws + pl.z[i], total, P, L.Co, ws + pl.mean[i],
                                                                              ws + pl.rstd[i], p.norm_w[i], p.norm_b[i],
                                                                              p.prelu_w[i], ws + pl.a[i])));
        in = ws + pl.a[i];
    }
    const int P4 = pl.L[3].To * pl.L[3].Fo;
    LAUNCH(ctx, "disc_head", (dc_maxpool_kernel<<<B, 128, 0, st>>>(ws + pl.a[3], P4, ws + pl.pooled, (int*)(ws + pl.idx))));
    LAUNCH(ctx, "disc_spectral_norm", (sn_power_kernel<<<1, 1024, 0, st>>>(p.fc1_w, 64, 128, p.fc1_u, p.fc1_v, update_uv, ws + pl.uu[4],
                                                                         ws + pl.vv[4], ws + pl.sigma + 4)));
    LAUNCH(ctx, "disc_spectral_norm", (sn_power_kernel<<<1, 1024, 0, st>>>(p.fc2_w, 1, 64, p.fc2_u, p.fc2_v, update_uv, ws + pl.uu[5],
                                                                         ws + pl.vv[5], ws + pl.sigma + 5)));
    LAUNCH(ctx, "disc_head", (dc_head_fwd_kernel<<<B, 64, 0, st>>>(ws + pl.pooled, dc_head(p, ws, pl), mask, ws + pl.h1, ws + pl.x2,
                                                                  score)));
}

void launch_disc_backward(LaunchCtx ctx, const float* xy, const float* dscore, int B, int T, int F, const DiscParams& p,
                          const float* mask, float* dxy, const DiscParams& grad, float* ws) {
    hipStream_t st = ctx.stream;
    const DcPlan pl = dc_plan(B, T, F);
    const DcHead hd = dc_head(p, ws, pl);
    LAUNCH(ctx, "disc_head", (dc_head_bwd_kernel<<<B, 128, 0, st>>>(ws + pl.pooled, ws + pl.h1, ws + pl.x2, dscore, hd, mask,
                                                                   ws + pl.dh1, ws + pl.ds2, ws + pl.dslope, ws + pl.dalpha,
                                                                   ws + pl.dpool)));
    const DcHeadGrads hg{grad.fc1_w, grad.fc1_b, grad.prelu5_w, grad.fc2_w, grad.fc2_b, grad.slope};
    LAUNCH(ctx, "disc_head", (dc_head_finish_kernel<<<1, 1024, 0, st>>>(ws + pl.pooled, ws + pl.h1, ws + pl.dh1, ws + pl.ds2,
                                                                       ws + pl.dslope, ws + pl.dalpha, mask, B, hd, ws + pl.uu[4],
                                                                       ws + pl.vv[4], ws + pl.uu[5], ws + pl.vv[5], hg)));
    const int P4 = pl.L[3].To * pl.L[3].Fo;
    hipMemsetAsync(ws + pl.g[3], 0, (size_t)B * P4 * 128 * sizeof(float), st);
    LAUNCH(ctx, "disc_head", (dc_maxpool_bwd_kernel<<<B, 128, 0, st>>>(ws + pl.dpool, (const int*)(ws + pl.idx), P4, ws + pl.g[3])));
    for (int i = 3; i >= 0; --i) {
        const DcLayer& L = pl.L[i];
        const DcGeom gm{B, L.Ti, L.Fi, L.Ci, L.To, L.Fo, L.Co};
        const int P = L.To * L.Fo;
        const long total = (long)B * P * L.Co;
        const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        float* g = ws + pl.g[i];                                  // dL/da_i -> dn -> dz, in place
        float* m1 = ws + pl.m12;
        float* m2 = ws + pl.m12 + (size_t)B * 128;
        LAUNCH(ctx, "disc_norm", (gin_sums_kernel<1><<<dim3(B, DC_NCH), 256, 0, st>>>(ws + pl.z[i], g, P, L.Co, ws + pl.mean[i],
                                                                                     ws + pl.rstd[i], p.norm_w[i], p.norm_b[i],
                                                                                     p.prelu_w[i], ws + pl.part)));
        LAUNCH(ctx, "disc_norm", (gin_bwd_finalize_kernel<<<1, 1024, 0, st>>>(ws + pl.part, B, L.Co, (double)P, m1, m2, grad.norm_w[i],
                                                                            grad.norm_b[i], grad.prelu_w[i])));
        LAUNCH(ctx, "disc_norm", (gin_in_bwd_kernel<<<grid, 256, 0, st>>>(g, ws + pl.z[i], total, P, L.Co, ws + pl.mean[i],
                                                                          ws + pl.rstd[i], p.norm_w[i], m1, m2)));
        const float* in = i == 0 ? xy : ws + pl.a[i - 1];
        const int nw = 16 * L.Ci * L.Co;
        {
            // rows per block: MB * Co = 8192 (first layer: all 32 rows); chunks so that ~1024 blocks run and the partial
            // slabs fit the DC_SPLIT * maxw floats of the workspace
            const int Mtot = 16 * L.Ci, MB = Mtot < 8192 / L.Co ? Mtot : 8192 / L.Co, groups = Mtot / MB;
            const int tiles_total = (int)(((long)B * P + DW_TP - 1) / DW_TP);
            int ns = 1024 / groups;
            const long cap = ((long)DC_SPLIT * 16 * 64 * 128) / nw;
            if (ns > cap) ns = (int)cap;
            if (ns > tiles_total) ns = tiles_total;
            const int tpc = (tiles_total + ns - 1) / ns;
            int lg = 0;
            while ((1 << lg) < L.Ci) ++lg;
            if (L.Ci * L.Co == 32)
                LAUNCH(ctx, "disc_conv_wgrad", (dc_conv_wgrad_kernel<2, 1><<<dim3(ns, groups), 256, 0, st>>>(g, in, gm, MB, lg, tiles_total,
                                                                                                          tpc, ws + pl.wpart)));
            else
                LAUNCH(ctx, "disc_conv_wgrad", (dc_conv_wgrad_kernel<8, 4><<<dim3(ns, groups), 256, 0, st>>>(g, in, gm, MB, lg, tiles_total,
                                                                                                          tpc, ws + pl.wpart)));
            LAUNCH(ctx, "disc_conv_wgrad", (dc_reduce_kernel<<<(nw + 63) / 64, 1024, 0, st>>>(ws + pl.wpart, ns, nw, ws + pl.G)));
        }
        LAUNCH(ctx, "disc_spectral_norm", (sn_conv_finish_kernel<<<1, 1024, 0, st>>>(ws + pl.G, ws + pl.wf[i], L.Co, L.Ci, ws + pl.uu[i],
                                                                                    ws + pl.vv[i], ws + pl.sigma + i, grad.conv_w[i])));
        float* din = i == 0 ? dxy : ws + pl.g[i - 1];
        if (din) {
            dc_launch_dgrad(ctx, g, ws + pl.wb[i], gm, din);
        }
    }
}

void launch_mag_pair(LaunchCtx ctx, const float* clean_spec, const float* est_real, const float* est_imag, int B, int T, int F,
                     float* xy) {
    LAUNCH(ctx, "disc_glue", (mag_pair_kernel<<<1024, 256, 0, ctx.stream>>>(clean_spec, est_real, est_imag, B, (long)T * F, xy)));
}
void launch_mag_pair_backward(LaunchCtx ctx, const float* est_real, const float* est_imag, const float* dxy, int B, int T, int F,
                              float scale, float* d_real, float* d_imag) {
    LAUNCH(ctx, "disc_glue", (mag_pair_bwd_kernel<<<1024, 256, 0, ctx.stream>>>(est_real, est_imag, dxy, (long)B * T * F, scale, d_real,
                                                                                d_imag)));
}
void launch_score_mse(LaunchCtx ctx, const float* score, const float* target, int B, float scale, float* loss, float* dscore) {
    LAUNCH(ctx, "disc_glue", (score_mse_kernel<<<1, 64, 0, ctx.stream>>>(score, target, B, scale, loss, dscore)));
}
