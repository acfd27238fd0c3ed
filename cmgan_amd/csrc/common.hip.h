// common.hip.h - device helpers shared by all kernels (gfx950 only, wave64).
//
// MFMA convention used everywhere (v_mfma_f32_16x16x4_f32, exact fp32):
//   lane l: c = l & 15, g = l >> 4
//   A[i = c][k = g]   one f32 per lane        B[k = g][j = c]   one f32 per lane
//   D[row = 4*g + reg][col = c], reg = 0..3   (f32x4 per lane)
// A "fragment" is the float4 a lane feeds over four consecutive k-steps; k-step r of
// block kb consumes k = 16*kb + 4*g + r (the contraction order is ours to choose, so
// operands are fetched as one float4 per lane instead of four strided scalars).
//
// Per-token linear layers are evaluated TRANSPOSED: out^T = W * x^T, with W as the A
// operand (rows = output features) and the activations as B (columns = 16 tokens).
// Then lane (token c, group g) of the result holds out[token][16*ob + 4*g + reg] -
// exactly the B-fragment layout the next layer needs for k-block ob - so chains of
// layers (LN -> W1 -> Swish -> W2, attention S -> softmax -> PV) never leave registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CMGAN_EPS 1e-5f

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }

// reduce over the 4 lane groups (lanes c, c+16, c+32, c+48) with v_permlane16/32_swap: two VALU
// ops per step instead of a ds_bpermute round trip through the LDS crossbar.
// permlane16_swap(v, v): {r0, r1} = {rows (0,0,2,2), rows (1,1,3,3)} of v; permlane32_swap: halves.
__device__ __forceinline__ float xchg16(float v, float& other) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    other = __uint_as_float(r[1]);
    return __uint_as_float(r[0]);
}
__device__ __forceinline__ float xchg32(float v, float& other) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    other = __uint_as_float(r[1]);
    return __uint_as_float(r[0]);
}
__device__ __forceinline__ float red_g_sum(float v) {
    float b;
    float a = xchg16(v, b);
    v = a + b;
    a = xchg32(v, b);
    return a + b;
}
__device__ __forceinline__ float red_g_max(float v) {
    float b;
    float a = xchg16(v, b);
    v = fmaxf(a, b);
    a = xchg32(v, b);
    return fmaxf(a, b);
}
// reduce over the 16 lanes of one lane group (same g, c = 0..15)
// acc (2 x f32) += w (2 x f32) * broadcast(u.lo) / broadcast(u.hi): packed FMA with the scalar operand
// selected by op_sel from either half of a register pair
__device__ __forceinline__ void pk_fma_lo(f32x2& acc, f32x2 w, f32x2 u) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(u));
}
__device__ __forceinline__ void pk_fma_hi(f32x2& acc, f32x2 w, f32x2 u) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(u));
}

// sum over the 16 lanes that share lane>>4 (one DPP row): four DPP adds (quad swaps, then the
// half-row and row mirrors), no LDS permute traffic.  (As __shfl_xor this was 4 ds_bpermute
// round trips per value: 128 of them made the dense-conv epilogue cost 14k cycles per tile.)
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float red_c_sum(float v) {
    v += dpp_perm<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);      // row_half_mirror: other quad of the 8-lane half
    v += dpp_perm<0x140>(v);      // row_mirror: other half of the row
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return red_g_sum(red_c_sum(v)); }

// sigmoid / Swish on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each).  NOT
// __frcp_rn / 1.0f/x: those expand to the ~12-instruction correctly-rounded division sequence
// and made the FFN VALU-bound.
__device__ __forceinline__ float sigmoidf_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float swishf(float x) { return x * sigmoidf_fast(x); }
// FeedForward activation on the pre-scaled hidden value h' = -log2(e) h (weights.h, CF_FF1_W1):
// Swish(h) = -ln2 * h' / (1 + 2^h'); the -ln2 lives in the second Linear.
__device__ __forceinline__ float swish_scaled(float hp) {
    return hp * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(hp));
}

// orders a wave's LDS writes before its later LDS reads (cross-lane, same wave)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------
// Token maps: which activation row each of the 16 tokens of a token block is.
// flat : block b covers tokens 16b .. 16b+15 of [M]
// seq  : block b = (n, ib) covers positions 16ib .. 16ib+15 of sequence n;
//        row(n, l) = (n / inner) * outer + (n % inner) * istride + l * lstride
//        time axis (n = (b,f'), l = t): inner = F', outer = T*F', istride = 1, lstride = F'
//        freq axis (n = (b,t),  l = f'): inner = 1,  outer = F',   istride = 0, lstride = 1
// ---------------------------------------------------------------------------------
struct TokMap {
    int seq;
    int nblocks;
    long M;
    int L, Lb;
    int inner;
    long outer, istride, lstride;
};

// returns validity; `row` is always a readable row (clamped) so loads need no predicate
__device__ __forceinline__ bool tok_row(const TokMap& m, int blk, int c, long& row) {
    if (!m.seq) {
        long t = (long)blk * 16 + c;
        bool ok = t < m.M;
        row = ok ? t : m.M - 1;
        return ok;
    }
    int n = blk / m.Lb, ib = blk - n * m.Lb;
    int l = ib * 16 + c;
    bool ok = l < m.L;
    if (!ok) l = m.L - 1;
    row = (long)(n / m.inner) * m.outer + (long)(n % m.inner) * m.istride + (long)l * m.lstride;
    return ok;
}

// acc[tb] += Wfm[ob][0..KB) (A operand) x xf[tb][0..KB) (B operand)
// wp = W_fm + ob*KB*256 + lane*4
template <int KB, int NTB>
__device__ __forceinline__ void lin_acc(const float* __restrict__ wp, const f32x4 (&xf)[NTB][KB],
                                        f32x4 (&acc)[NTB]) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const f32x4 a = ldg4(wp + kb * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int tb = 0; tb < NTB; ++tb) acc[tb] = mfma16(a[r], xf[tb][kb][r], acc[tb]);
        }
    }
}

// LayerNorm statistics of a 64-channel row held as 4 fragments (16 values per lane,
// the 4 lanes c, c+16, c+32, c+48 together hold the row).  Biased variance, eps 1e-5
// (nn.LayerNorm, conformer.py:68).
__device__ __forceinline__ void ln_stats(const f32x4 (&x)[4], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) s += (x[kb][0] + x[kb][1]) + (x[kb][2] + x[kb][3]);
    mean = red_g_sum(s) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = x[kb][r] - mean;
            v = fmaf(d, d, v);
        }
    }
    rstd = rsqrtf(red_g_sum(v) * (1.0f / 64.0f) + CMGAN_EPS);
}

// =====================================================================================
// "x3" mode: fp32-accurate products on the f16 matrix pipe.
// gfx950 has no TF32/xf32 and its fp32 MFMA runs at 1/16 of the f16/bf16 rate, so every
// operand x is split as x = hi + lo with hi = fp16(x), lo = fp16(x - hi) and a product is
// evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation:
// ~2^-21 relative error per product (fp32-class) at ~5x the fp32-MFMA throughput.
// All MFMA operands on this path are bounded (post-LayerNorm / InstanceNorm activations,
// softmax probabilities, O(0.1) weights), far inside fp16 range; the unbounded STFT/ISTFT
// stay on the fp32 pipe.
//
// 16x16x32 f16 convention: lane l (c = l & 15, g = l >> 4) feeds 8 halfs; k-slot (g, e),
// e = 0..7.  Which contraction index a slot means is ours to choose, consistently for A
// and B.  Chain convention: slot (g, e) of k32-block m  <->  feature 32m + 16(e>>2) + 4g + (e&3),
// i.e. the concatenation of the two f32x4 C-fragments (blocks 2m, 2m+1) a lane already
// holds - so layer outputs feed the next layer's B operand without any data movement.
// A weight image is [ob][m][hi|lo][64 lanes][8 halfs] (built from the fp32 fragment-major
// weights on the host at load time, api.hip).
// =====================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma32h(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// X3_TERMS = 3 (the product): hi*hi + hi*lo + lo*hi.  X3_TERMS = 1 (with -DX3_SINGLE; cmgan_amd/build.py compiles the
// three x3 kernel files a second time this way into the `_x1` entry points) is the shipped, OPT-IN, reduced-precision
// mode CMGAN_MFMA_F16X1: every product term that involves a lo half is compiled out, i.e. plain fp16 operands with
// fp32 accumulation - the "single-product half precision" mode BASELINE.json's configs[1] names ("TSCNet bf16"; fp16
// has 3 more mantissa bits than bf16).  6e-4 .. 9e-4 from the reference: inside the 1e-3 gate without margin, not
// fp32-class, never the default (include/cmgan_hip.h; tests/test_gpu_parity.py::test_f16x1_mode_error_bands).
#ifndef X3_TERMS
#define X3_TERMS 3
#endif
__device__ __forceinline__ f32x4 mfma32l(f16x8 a, f16x8 b, f32x4 c) {      // a term with a lo operand
    return X3_TERMS == 3 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0) : c;
}
__device__ __forceinline__ f16x2 pkrtz(float a, float b) {
    return __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a, b));
}
// 4 floats -> hi / lo halves (hi by round-toward-zero, lo = x - hi exactly representable residual, RTZ)
// hi = fp16 RTZ of (a, b); lo = fp16(a - hi.x, b - hi.y).  The f16 -> f32 widening of hi, the
// subtraction and the fp16 rounding of the residual are ONE v_fma_mix{lo,hi}_f16 per value
// (mixed-precision FMA: f16 source, f32 addend, f16 result into one half of the destination):
// 1.5 VALU per value instead of the 2.5-3 the compiler emits for cvt / sub / pack.  VALU issue
// is not hidden behind MFMAs on this machine, so this is kernel time.
__device__ __forceinline__ void split2(float a, float b, f16x2& hi, f16x2& lo) {
    if (X3_TERMS == 1) {                       // single-product experiment: hi alone carries the value -> round to nearest
        hi[0] = (_Float16)a; hi[1] = (_Float16)b;
        lo[0] = (_Float16)0.f; lo[1] = (_Float16)0.f;
        return;
    }
    hi = pkrtz(a, b);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(l)
        : "v"(hi), "v"(a), "v"(b));
    lo = __builtin_bit_cast(f16x2, l);
}
__device__ __forceinline__ void split4(f32x4 v, f16x4& hi, f16x4& lo) {
    f16x2 h0, h1, l0, l1;
    split2(v[0], v[1], h0, l0);
    split2(v[2], v[3], h1, l1);
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1];
}
// two C-fragments (k-blocks 2m, 2m+1) -> one k32 B/A operand pair
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, f16x8& hi, f16x8& lo) {
    f16x4 ha, la, hb, lb;
    split4(a, ha, la);
    split4(b, hb, lb);
    hi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
}
// acc[tb] += W(ob, m = 0..M32) x B[tb][m]  with the three split products.
// wp = image + ob*M32*1024 + lane*8 (halfs); image may live in LDS or global.
template <int M32, int NTB_>
__device__ __forceinline__ void lin_acc_x3(const _Float16* wp, const f16x8 (&bh)[NTB_][M32],
                                           const f16x8 (&bl)[NTB_][M32], f32x4 (&acc)[NTB_]) {
#pragma unroll
    for (int m = 0; m < M32; ++m) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(wp + m * 1024);
        const f16x8 al = *reinterpret_cast<const f16x8*>(wp + m * 1024 + 512);
#pragma unroll
        for (int tb = 0; tb < NTB_; ++tb) acc[tb] = mfma32h(ah, bh[tb][m], acc[tb]);
#pragma unroll
        for (int tb = 0; tb < NTB_; ++tb) acc[tb] = mfma32l(ah, bl[tb][m], acc[tb]);
#pragma unroll
        for (int tb = 0; tb < NTB_; ++tb) acc[tb] = mfma32l(al, bh[tb][m], acc[tb]);
    }
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// block-cooperative linear copy global -> LDS of N16 16-byte units by NTHR threads.  All loads are
// issued before the first LDS store: a plain `for (i = tid; i < n; i += nthr) lds[i] = g[i]` loop has a
// runtime trip count, so each load is waited for before its store and the copy pays one full memory
// round trip PER ITERATION (measured: 8 serial round trips = ~24k cycles per 32-token dwconv block).
template <int N16, int NTHR>
__device__ __forceinline__ void stage_lds16(const void* __restrict__ g, void* l) {
    constexpr int IT = (N16 + NTHR - 1) / NTHR;
    const u32x4* src = reinterpret_cast<const u32x4*>(g);
    u32x4* dst = reinterpret_cast<u32x4*>(l);
    u32x4 t[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = threadIdx.x + NTHR * k;
        if (N16 % NTHR == 0 || i < N16) t[k] = src[i];
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = threadIdx.x + NTHR * k;
        if (N16 % NTHR == 0 || i < N16) dst[i] = t[k];
    }
}
