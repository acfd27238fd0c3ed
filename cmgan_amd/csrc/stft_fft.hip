// stft_fft.hip - the front / back end at n_fft 400 / hop 100 as REAL transforms (round 6): torch.stft + power_compress and
// power_uncompress + torch.istft (src/evaluation.py:36-51, src/utils.py:20-39) with the 400-point real DFT of a frame
// evaluated as a 16 x 25 Cooley-Tukey factorisation in fp32 on the VALU, instead of the dense (folded) 201 x 200 matrix
// products of stft.hip, whose 364 KB operand image every block streams from L2 (87 MB of L2 reads per launch at 32 clips
// for 20.6 MB of HBM traffic):  ~3 k flops per frame instead of 80 k MACs x 3 split products.  Twiddles are fp32
// roundings of float64 values (stft_fft_tables.h, tools/gen_fft_tables.py); the accuracy is that of an fp32 FFT (~1e-7
// of a frame's largest bin) at any input amplitude.  Index maps and butterflies: tests/test_stft_fft_model.py.
//
//   n = 16 m + r,  k = k' + 25 k1:
//   X[k' + 25 k1] = sum_r W16^{r k1} ( W400^{r k'} sum_m x[16 m + r] W25^{m k'} )
//
// This file is compiled WITH packed fp32 operations (cmgan_amd/build.py): a complex add is one v_pk_add_f32, a complex
// multiply a v_pk_mul_f32 + a v_pk_fma_f32.  (Beside MFMAs those cost issue slots - the rest of the library is built
// without them - but these kernels are pure VALU work and VALU-issue bound: measured, the instruction count IS the time.)
#include "kernels.h"
#include "stft_fft_tables.h"

typedef float f2 __attribute__((ext_vector_type(2)));        // (re, im)

// |X|^p x sign-preserving helper of stft.hip: m2^p for m2 >= 0, branch-free, correct down to the smallest denormal
__device__ __forceinline__ float fft_pow_pos(float m2, float p) {
    const float l2 = __builtin_amdgcn_logf(m2 * 1152921504606846976.0f) - 60.0f;
    const float r = __builtin_amdgcn_exp2f(p * l2);
    return m2 > 0.f ? r : 0.f;
}
__device__ __forceinline__ f2 cmul(f2 a, f2 b) { const f2 bs = {-b.y, b.x}; return a.xx * b + a.yy * bs; }
__device__ __forceinline__ f2 cmulc(f2 a, f2 b) { const f2 bs = {b.y, b.x}; return a.xx * f2{b.x, -b.y} + a.yy * bs; }   // a conj(b)
__device__ __forceinline__ f2 mul_mi(f2 a) { return f2{a.y, -a.x}; }          // -i a
__device__ __forceinline__ f2 mul_pi(f2 a) { return f2{-a.y, a.x}; }          // +i a
__device__ __forceinline__ f2 cconj(f2 a) { return f2{a.x, -a.y}; }
__device__ __forceinline__ f2 cld(const float2* p) { const float2 v = *p; return f2{v.x, v.y}; }
#define FFT_C1 0.30901699437494745f
#define FFT_C2 -0.8090169943749475f
#define FFT_S1 0.9510565162951535f
#define FFT_S2 0.5877852522924731f
// 5-point DFT in place; INV = false: e^{-2 pi i n k / 5}, true: e^{+...}
template <bool INV>
__device__ __forceinline__ void dft5(f2 (&x)[5]) {
    const f2 t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
    const f2 a1 = x[0] + FFT_C1 * t1 + FFT_C2 * t2, a2 = x[0] + FFT_C2 * t1 + FFT_C1 * t2;
    const f2 b1 = FFT_S1 * t3 + FFT_S2 * t4, b2 = FFT_S2 * t3 - FFT_S1 * t4;
    const f2 r1 = INV ? mul_pi(b1) : mul_mi(b1), r2 = INV ? mul_pi(b2) : mul_mi(b2);
    x[0] = x[0] + t1 + t2;
    x[1] = a1 + r1; x[4] = a1 - r1;
    x[2] = a2 + r2; x[3] = a2 - r2;
}
// forward 5-point DFT of REAL input (half the work of the complex butterfly: no imaginary parts to carry)
__device__ __forceinline__ void dft5_real(const float (&x)[5], f2 (&y)[5]) {
    const float t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
    const float a1 = x[0] + FFT_C1 * t1 + FFT_C2 * t2, a2 = x[0] + FFT_C2 * t1 + FFT_C1 * t2;
    const float b1 = FFT_S1 * t3 + FFT_S2 * t4, b2 = FFT_S2 * t3 - FFT_S1 * t4;
    y[0] = f2{x[0] + t1 + t2, 0.f};
    y[1] = f2{a1, -b1}; y[4] = f2{a1, b1};
    y[2] = f2{a2, -b2}; y[3] = f2{a2, b2};
}
template <bool INV>
__device__ __forceinline__ void dft4(f2 (&x)[4]) {
    const f2 t0 = x[0] + x[2], t1 = x[0] - x[2], t2 = x[1] + x[3], t3 = x[1] - x[3];
    const f2 r3 = INV ? mul_pi(t3) : mul_mi(t3);
    x[0] = t0 + t2; x[2] = t0 - t2;
    x[1] = t1 + r3; x[3] = t1 - r3;
}
#define FFT_HP 17                 // float2 pitch of a (frame, channel) row of 16 residues in the forward exchange buffer
#define IFFT_HP 13                // float2 pitch of a (frame, residue) row of 13 channels in the inverse exchange buffer

// ---------------------------------------------------------------------------------
// stft_fft400_kernel: a wave owns FOUR consecutive frames (lane = frame j x residue r):
//   pass 1  lane (j, r): the 25 windowed samples x[16 m + r] -> 25-point DFT in registers (5 x 5, real input), channels
//           k' = 0..12 only (the others are their conjugates), times W400^{r k'}  -> wave-private LDS
//   pass 2  lane (j, k'), 52 lanes: 16-point DFT over r in registers (4 x 4): Y[k1] = X[k' + 25 k1]; bins above N/2 are
//           the conjugates of the missing channels 25 - k': together exactly the 201 bins, each once;
//           |X|^-0.7 compression on the spot; staged so that
//   store   the four frames' rows (4 x 201 contiguous floats per part) leave as coalesced rows.
// Nothing is shared between waves: no block barrier; a block is four such waves (16 frames of one clip).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stft_fft400_kernel(const float* __restrict__ wav, const float* __restrict__ scale,
                                                          const float* __restrict__ window, int L, int T,
                                                          float* __restrict__ spec) {
    // per wave: the segment (704 floats: samples 100 t0 - 200 + i, i < 700) and the exchange buffer (52 x 17 float2) side
    // by side; the output staging (2 x 4 x 201 floats + one dump slot) re-uses the same bytes once pass 2 holds its
    // inputs in registers: 10 KB per wave, four blocks per CU
    constexpr int WREG = 704 + 2 * 52 * FFT_HP;
    static_assert(WREG >= 2 * 4 * 201 + 1, "the output staging must fit the wave's region");
    __shared__ __attribute__((aligned(16))) float wreg[4][WREG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* const sg = wreg[wv];
    float2* const hx_w = reinterpret_cast<float2*>(wreg[wv] + 704);
    float* const out_w = wreg[wv];
    const int b = blockIdx.y, t0 = (blockIdx.x * 4 + wv) * 4;  // this wave's first frame
    if (t0 >= T) return;                                      // (wave-uniform; no block barrier anywhere below)
    const float sc = scale ? scale[b] : 1.0f;
    const float* x = wav + (long)b * L;
    {
        const int start = 100 * t0 - 200;
        if (start >= 0 && start + 704 <= L) {                 // interior group (wave-uniform): plain coalesced loads
#pragma unroll
            for (int k = 0; k < 11; ++k) sg[lane + 64 * k] = x[start + lane + 64 * k] * sc;
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const int i = lane + 64 * k;
                int sidx = start + i;
                sidx = sidx < 0 ? -sidx : sidx;                // reflect padding (torch.stft center=True)
                sidx = sidx >= L ? 2 * (L - 1) - sidx : sidx;
                sidx = sidx < 0 ? 0 : (sidx >= L ? L - 1 : sidx);   // frames past T in the last group: any finite value
                sg[i] = x[sidx] * sc;
            }
        }
    }
    wave_lds_fence();
    // ---- pass 1: lane (j, r) ----
    {
        const int j = lane >> 4, r = lane & 15;
        float v[25];
#pragma unroll
        for (int m = 0; m < 25; ++m) v[m] = sg[100 * j + 16 * m + r] * window[16 * m + r];
        f2 C[5][5];                                           // C[b][c] = sum_a v[5 a + b] W5^{a c}
#pragma unroll
        for (int bb = 0; bb < 5; ++bb) {
            const float xr[5] = {v[bb], v[5 + bb], v[10 + bb], v[15 + bb], v[20 + bb]};
            dft5_real(xr, C[bb]);
        }
        float2* hw = hx_w + (j * 13) * FFT_HP + r;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            f2 z[5];
            z[0] = C[0][c];
#pragma unroll
            for (int bb = 1; bb < 5; ++bb) z[bb] = c == 0 ? C[bb][c] : cmul(C[bb][c], cld(&fft_tw25[bb * 5 + c]));
            dft5<false>(z);                                   // z[d] = G[c + 5 d]
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int kp = c + 5 * d;
                if (kp < 13) {
                    const f2 h = cmul(z[d], cld(&fft_tw400[r * 13 + kp]));
                    // (the compiler barrier keeps the 8-byte stores apart: merged pairwise into ds_write2_b64, the later data
                    // dwords are still being read by the LDS path when the next VALU instruction overwrites them - tools/isa_lint.py;
                    // a volatile store would do too, but it goes out as a system-scope FLAT store)
                    hw[kp * FFT_HP] = make_float2(h.x, h.y);
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    wave_lds_fence();
    // ---- pass 2: lane (j, k'), 52 lanes ----
    {
        const int l2 = lane < 52 ? lane : 51;
        const int j = l2 / 13, kp = l2 - 13 * j;
        const float2* hr = hx_w + l2 * FFT_HP;
        f2 inner[4][4];                                       // inner[q][s] = sum_p h[4 p + q] W4^{p s}
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f2 z[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) z[pp] = cld(hr + 4 * pp + q);
            dft4<false>(z);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) inner[q][s_] = z[s_];
        }
        // the staging below re-uses the bytes of the segment and of the exchange buffer: every lane's reads above are
        // issued before any lane's writes (one instruction stream, LDS operations of a wave execute in order) - the
        // compiler only has to keep that order
        asm volatile("" ::: "memory");
        // Where channel k' puts Y[k1] (k = k' + 25 k1): k1 <= 7 is always a direct bin, k1 >= 9 always a mirror bin
        // 400 - k (conjugated), k1 = 8 is bin 200 - k' either way (k' = 0: the Nyquist bin itself; else mirrored) - so the
        // addresses are two per-lane bases plus compile-time offsets.  Channel 0's mirror bins (k1 >= 9) are bins it also
        // produces directly: dropped (dump slot).  Lanes 52..63 recompute lane 51 and store the same values again.
        constexpr int DUMP = 2 * 804;
        // (indices into ONE LDS array: a select between two LDS POINTERS loses the address space and becomes a flat store)
        const int id_ = j * 201 + kp;                         // direct bins: + 25 k1
        const int im_ = j * 201 - kp;                         // mirror bins: + 400 - 25 k1
        const float sgn8 = kp == 0 ? 1.f : -1.f;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            f2 z[4];
            z[0] = inner[0][s_];
#pragma unroll
            for (int q = 1; q < 4; ++q) z[q] = s_ == 0 ? inner[q][s_] : cmul(inner[q][s_], cld(&fft_tw16[q * 4 + s_]));
            dft4<false>(z);                                   // z[u] = Y[s + 4 u] = X[k' + 25 (s + 4 u)]
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k1 = s_ + 4 * u;
                const float m2 = z[u].x * z[u].x + z[u].y * z[u].y;
                const f2 c_ = z[u] * fft_pow_pos(m2, -0.35f);  // power compression (utils.py:20-29)
                if (k1 <= 7) { out_w[id_ + 25 * k1] = c_.x; out_w[id_ + 804 + 25 * k1] = c_.y; }
                else if (k1 == 8) { out_w[im_ + 200] = c_.x; out_w[im_ + 804 + 200] = sgn8 * c_.y; }
                else {
                    const int o_ = kp == 0 ? DUMP : im_ + (400 - 25 * k1);
                    out_w[o_] = c_.x;
                    out_w[kp == 0 ? DUMP : o_ + 804] = -c_.y;
                }
            }
        }
    }
    wave_lds_fence();
    // ---- store: rows t0 .. t0 + 3 of both parts are contiguous in [B,2,T,F] ----
    const int nfr = T - t0 < 4 ? T - t0 : 4;
    const long P = (long)T * 201;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        float* dst = spec + ((long)b * 2 + part) * P + (long)t0 * 201;
        const float* src = out_w + 804 * part;
        if (nfr == 4) {                                       // (wave-uniform) 804 floats = 12 full trips + 36
#pragma unroll
            for (int k = 0; k < 12; ++k) dst[lane + 64 * k] = src[lane + 64 * k];
            if (lane < 36) dst[768 + lane] = src[768 + lane];
        } else {
            for (int i = lane; i < nfr * 201; i += 64) dst[i] = src[i];
        }
    }
}

// ---------------------------------------------------------------------------------
// istft_fft400_kernel - the mirror image, ONE launch for power uncompress + inverse real FFT + synthesis window +
// overlap-add + envelope division + centre trim + '/ c' (utils.py:32-39, evaluation.py:44-51):
//   x[16 m + r] = (1 / N) sum_k' W25^{-m k'} ( W400^{-r k'} sum_k1 Y[k' + 25 k1] W16^{-r k1} )
// A block produces 1600 output samples (16 hops) of one clip; they are covered by the 19 frames 16 i - 1 .. 16 i + 17, so
// five waves of four frames each (1.2 x the transforms, in exchange for which no frame ever leaves the chip: the
// [B, T, 400] frame tensor of the two-kernel form was 16.4 MB written and read back per launch):
//   pass A  lane (j, k'), 52 lanes: channel inputs Y[k' + 25 k1] straight from the rows of est_real / est_imag (bins above
//           N/2 = conjugates of the mirror bins; the imaginary parts of DC / Nyquist ignored, as irfft does), |Y|^(7/3)
//           un-compression on the spot, inverse 16-point DFT (4 x 4), times W400^{-r k'} -> wave-private LDS
//   pass B  lane (j, r): the 13 channels -> 25 by Hermitian symmetry, inverse 25-point DFT (5 x 5), real parts,
//           times w[n] / N -> the block's frame buffer
//   OLA     after ONE barrier: every thread sums the (up to) four frames over five output samples, divides by the
//           window envelope of the frames that exist (torch.istft) and by the clip's scale.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(320) void istft_fft400_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                           const float* __restrict__ scale, const float* __restrict__ window,
                                                           int T, int Lout, float* __restrict__ wav) {
    // per wave: the exchange buffer [(frame, residue)][channel] (64 x 13 float2), re-used - once pass B holds its inputs in
    // registers - for the wave's four windowed time-domain frames (4 x 400 floats): frame slot fs = 4 w + j (frame
    // 16 i - 1 + fs) lives at hx[w] + 400 j.  35 KB per block: four blocks per CU.
    constexpr int HXF = 2 * 64 * IFFT_HP;                     // floats per wave region
    static_assert(HXF >= 4 * 400, "a wave's four frames must fit its exchange buffer");
    __shared__ __attribute__((aligned(16))) float2 hx[5][64 * IFFT_HP];
    __shared__ float win[400];
    const float* const fr = reinterpret_cast<const float*>(&hx[0][0]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int tw0 = 16 * blk - 1 + 4 * wv;                    // this wave's first frame (may be -1 or beyond T)
    float2* const hx_w = hx[wv];
    // (every wave stages the whole window: its own stores are visible to it behind its own wave fence; 7 fetches per lane)
    for (int i = lane; i < 400; i += 64) win[i] = window[i];
    // frames of this wave that exist and are needed (slot 19 = frame 16 i + 18 covers no sample of this block): [ja, jb)
    const int ja = tw0 < 0 ? -tw0 : 0;
    int jb = T - tw0 < 4 ? (T - tw0 < 0 ? 0 : T - tw0) : 4;
    if (wv == 4 && jb > 3) jb = 3;
    if (jb > ja) {                                            // (wave-uniform)
        {
            // ---- pass A: lane (j, k') (lanes outside the live (frame, channel) set work on a clamped copy and are dropped) ----
            const int l2 = lane < 52 ? lane : 51;
            const int jr = l2 / 13, kp = l2 - 13 * jr;
            const int j = jr < ja ? ja : (jr >= jb ? jb - 1 : jr);
            const bool act = lane < 52 && jr >= ja && jr < jb;
            // channel inputs Y[k' + 25 k1]: direct bins for k1 <= 7, bin 200 - k' for k1 = 8 (mirrored unless k' = 0), mirror bins
            // 400 - 25 k1 - k' for k1 >= 9: two per-lane row pointers plus compile-time offsets
            const long row = ((long)b * T + (tw0 + j)) * 201;
            const float *rd = re + row + kp, *id = im + row + kp, *rm = re + row - kp, *imr = im + row - kp;
            f2 y[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                if (k1 <= 7) y[k1] = f2{rd[25 * k1], id[25 * k1]};
                else if (k1 == 8) y[k1] = f2{rm[200], imr[200]};
                else y[k1] = f2{rm[400 - 25 * k1], imr[400 - 25 * k1]};
            }
            const float s0 = kp == 0 ? 0.f : 1.f;             // the imaginary parts of DC (k1 = 0) and Nyquist (k1 = 8) are ignored
            f2 inner[4][4];                                   // inner[q][s] = sum_p Y[k' + 25 (4 p + q)] W4^{-p s}
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f2 z[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int k1 = 4 * pp + q;
                    const f2 v = y[k1];
                    const float f = fft_pow_pos(v.x * v.x + v.y * v.y, 7.0f / 6.0f);   // mag^(1/0.3), phase kept (utils.py:32-39)
                    const float sy = k1 == 0 ? s0 : (k1 == 8 ? -s0 : (k1 <= 7 ? 1.f : -1.f));   // conj for mirror bins
                    z[pp] = f2{v.x, sy * v.y} * f;
                }
                dft4<true>(z);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) inner[q][s_] = z[s_];
            }
            float2* const hbase = hx_w + (j * 16) * IFFT_HP + kp;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                f2 z[4];
                z[0] = inner[0][s_];
#pragma unroll
                for (int q = 1; q < 4; ++q) z[q] = s_ == 0 ? inner[q][s_] : cmulc(inner[q][s_], cld(&fft_tw16[q * 4 + s_]));
                dft4<true>(z);                                // z[u] = h[r = s + 4 u]
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = s_ + 4 * u;
                    const f2 h = cmulc(z[u], cld(&fft_tw400[r * 13 + kp]));
                    if (act) {
                        hbase[r * IFFT_HP] = make_float2(h.x, h.y);
                        asm volatile("" ::: "memory");        // (keeps the 8-byte stores apart: see stft_fft400_kernel)
                    }
                }
            }
        }
        wave_lds_fence();
        // ---- pass B: lane (j, r) ----
        const int j2 = lane >> 4, r = lane & 15;
        if (j2 >= ja && j2 < jb) {
            const float2* hr = hx_w + lane * IFFT_HP;
            f2 full[25];
#pragma unroll
            for (int k = 0; k < 13; ++k) full[k] = cld(hr + k);
#pragma unroll
            for (int k = 13; k < 25; ++k) full[k] = cconj(full[25 - k]);
            f2 C[5][5];                                       // C[b][c] = sum_a full[5 a + b] W5^{-a c}
#pragma unroll
            for (int bb = 0; bb < 5; ++bb) {
                f2 z[5];
#pragma unroll
                for (int a = 0; a < 5; ++a) z[a] = full[5 * a + bb];
                dft5<true>(z);
#pragma unroll
                for (int c = 0; c < 5; ++c) C[bb][c] = z[c];
            }
            // (the frames re-use the exchange buffer's bytes: every lane's reads above are issued before any lane's writes -
            // one instruction stream, LDS operations of a wave execute in order; the compiler only has to keep that order)
            asm volatile("" ::: "memory");
            float* fw = reinterpret_cast<float*>(hx_w) + j2 * 400 + r;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                f2 z[5];
                z[0] = C[0][c];
#pragma unroll
                for (int bb = 1; bb < 5; ++bb) z[bb] = c == 0 ? C[bb][c] : cmulc(C[bb][c], cld(&fft_tw25[bb * 5 + c]));
                dft5<true>(z);                                // z[d].re = N x[16 (c + 5 d) + r]
#pragma unroll
                for (int d = 0; d < 5; ++d) {
                    const int m = c + 5 * d;
                    fw[16 * m] = z[d].x * (1.0f / 400.0f) * win[16 * m + r];
                }
            }
        }
    }
    __syncthreads();
    // ---- overlap-add over the block's 1600 samples (branch-free: frames that do not exist read slot 0 with weight 0) ----
    const float cdiv = scale ? scale[b] : 1.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sl = threadIdx.x + 320 * i, s_ = 1600 * blk + sl;
        const int fs0 = sl / 100, n0 = 300 + (sl - 100 * fs0);
        float acc = 0.f, env = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fs = fs0 + q, t = 16 * blk - 1 + fs, n = n0 - 100 * q;
            const bool ok = t >= 0 && t < T;
            const float v = fr[ok ? (fs >> 2) * HXF + (fs & 3) * 400 + n : 0];
            const float w = win[n];
            acc += ok ? v : 0.f;
            env = fmaf(ok ? w : 0.f, w, env);
        }
        float v = acc * __builtin_amdgcn_rcpf(env);           // (1 ulp: the envelope is in [1.29, 1.6] wherever a sample is kept)
        if (scale) v /= cdiv;                                 // (a true division, as ola_kernel and the reference's '/ c')
        if (s_ < Lout) wav[(long)b * Lout + s_] = v;
    }
}

void launch_stft_fft400(LaunchCtx ctx, const float* wav, const float* scale, const float* window, int B, int L, int T,
                        float* spec) {
    dim3 grid((T + 15) / 16, B);
    LAUNCH(ctx, "stft_compress", (stft_fft400_kernel<<<grid, 256, 0, ctx.stream>>>(wav, scale, window, L, T, spec)));
}

void launch_istft_fft400(LaunchCtx ctx, const float* re, const float* im, const float* scale, const float* window, int B,
                         int T, float* wav_out) {
    const int Lo = 100 * (T - 1);
    dim3 grid((Lo + 1599) / 1600, B);
    LAUNCH(ctx, "uncompress_irfft", (istft_fft400_kernel<<<grid, 320, 0, ctx.stream>>>(re, im, scale, window, T, Lo, wav_out)));
}
