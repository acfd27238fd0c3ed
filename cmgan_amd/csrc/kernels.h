// kernels.h - internal launcher interface between api.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "common.hip.h"

// Optional per-launch timing with HIP events on the caller's stream (bench.py's
// live roofline figure).  Disabled by default: then LAUNCH() is a plain launch.
struct Profiler {
    bool enabled = false;
    struct Rec { const char* name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void reset() { recs.clear(); used = 0; }
};

// cmgan_enhance_branched: the second half-batch branch starts on another stream once the first branch has issued `at`
// launches (an event recorded behind that launch), so that the two branches are in different kernels at any time.
struct Fork {
    int count = 0, at = 0;
    bool fired = false;
    hipEvent_t ev = nullptr;
    void tick(hipStream_t s) {
        if (!fired && ++count >= at) { hipEventRecord(ev, s); fired = true; }
    }
};

struct LaunchCtx {
    hipStream_t stream;
    Profiler* prof;
    Fork* fork = nullptr;
};

#define LAUNCH(ctx, name, ...)                                         \
    do {                                                               \
        if ((ctx).prof && (ctx).prof->enabled) {                       \
            hipEvent_t _a = (ctx).prof->get(), _b = (ctx).prof->get(); \
            hipEventRecord(_a, (ctx).stream);                          \
            __VA_ARGS__;                                               \
            hipEventRecord(_b, (ctx).stream);                          \
            (ctx).prof->recs.push_back({name, _a, _b});                \
        } else {                                                       \
            __VA_ARGS__;                                               \
        }                                                              \
        if ((ctx).fork) (ctx).fork->tick((ctx).stream);                \
    } while (0)

// Launch-shape / kernel-choice overrides read from the environment ONCE per process (same-session A/B sweeps,
// tools/knob_sweep.sh; listed in include/cmgan_hip.h, "Environment").  Every value is validated: anything outside
// [lo, hi] (or not a number) falls back to the built-in default, so no setting can produce an invalid launch.  None of
// them changes a result: they select between kernels / launch shapes that are each parity-tested.
#include <stdlib.h>
inline int env_knob(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    const long x = strtol(v, &end, 10);
    return (end && *end == 0 && x >= lo && x <= hi) ? (int)x : dflt;
}

// ------------------------------- stft.hip ---------------------------------------
struct SpectralTables {
    int n_fft, hop, F, FB;          // FB = ceil(F/16) bin blocks
    const float* fwd_fm;            // fm [2*FB][n_fft/16][64][4]: rows = re bins | im bins, window folded
    const float* inv_fm;            // fm [n_fft/16][2*FB][64][4]: rows = output sample n, cols = re|im bins, window/N folded
    const float* window;            // [n_fft]
    // F16X3 mode, n_fft = 400 / hop = 100 only (else null): forward real DFT folded about n = N/2
    // (u_c[n] = x[n] + x[N-n], u_s[n] = x[n] - x[N-n]; the symmetric window lives in the matrix), as
    // B-operand images [FB][cos | -sin][7 k32 blocks][hi | lo][64 lanes][8 halfs]   (stft.hip)
    const void* fold_fwd16;
    // inverse counterpart: [13 sample blocks of 16 (n = 0..207)][cos | sin][7 k32 blocks][hi | lo][64][8],
    // Hermitian weights and 1/N folded in, window applied in the epilogue
    const void* fold_inv16;
};
void launch_rms_scale(LaunchCtx, const float* wav, int B, int L, float* scale);
void launch_stft_compress(LaunchCtx, const SpectralTables&, const float* wav, const float* scale,
                          int B, int L, int T, float* spec);
void launch_uncompress_istft(LaunchCtx, const SpectralTables&, const float* re, const float* im,
                             const float* scale, int B, int T, float* frames_ws, float* wav_out);
// stft_fft.hip: n_fft 400 / hop 100 as real FFTs (16 x 25 factorisation on the VALU); window = SpectralTables::window
void launch_stft_fft400(LaunchCtx, const float* wav, const float* scale, const float* window, int B, int L, int T, float* spec);
void launch_istft_fft400(LaunchCtx, const float* re, const float* im, const float* scale, const float* window, int B, int T,
                         float* wav_out);
void launch_power_compress(LaunchCtx, const float* x, int B, int F, int T, float* y);
void launch_power_uncompress(LaunchCtx, const float* re, const float* im, int B, int F, int T, float* y);

// ------------------------------- conv.hip ---------------------------------------
struct ConvArgs {
    const float* in[4];        // channels-last [B, T*F, 64] slots, slot order = concat order (oldest first)
    const float* nscale[4];    // per slot [B][64] InstanceNorm scale (NULL = identity)
    const float* nshift[4];    // per slot [B][64]
    const float* nalpha[4];    // per slot [64] PReLU slopes (NULL = none)
    int nslots;
    const float* w;            // fm chunks [4*nslots][taps][COUT/16][64][4]
    const float* bias;         // [COUT]
    float* out;
    float* partials;           // [B][ntiles][COUT][2] (sum, sum of squares) or NULL
    int T, F, dil;
    int mode;                  // 0 plain, 1 keep even f only (stride-2 conv), 2 pixel shuffle (COUT = 128)
    int ntiles;
    // training only (launch_conv3_x3_dgrad; ignored by the inference instantiations): the data gradient of a dense-block
    // conv is the same conv on the TIME-REVERSED plane (the causal tap t - dil becomes the anti-causal t + dil) with
    // transposed, frequency-mirrored weights, accumulated into the slot's gradient plane
    // x3 dense blocks only (conv_x3.hip): slot s with bit s of img_mask set is not raw fp32 but the NORMALISED, PReLU'd
    // activation already split into fp16 (hi, lo) - per 4 channels 8 bytes of hi then 8 bytes of lo, i.e. the same 16
    // bytes per channel quad and the same addressing as the raw slot - written by the slot's FIRST consumer (img_out:
    // the newest slot's image, stored from the t-plane stage by the tile that owns the position), so that the later
    // layers of the block stage it with two LDS stores instead of normalise -> PReLU -> split for every re-read
    unsigned img_mask;
    void* img_out;
    int revt;                  // 1: logical frame t is physical frame T - 1 - t, for inputs and outputs alike
    int accum;                 // 1: out += oscale * result
    const float* oscale;       // device scalar (the inverse of the power-of-two input scale carried by nscale)
};
int  conv3_ntiles(int T, int F);
void launch_conv3(LaunchCtx, const ConvArgs&, int B, int time_taps, int cout);
int  conv_in_ntiles(int P);
void launch_conv_in(LaunchCtx, const float* spec, const float* w, float* out, float* partials, int B, int P);
void launch_in_finalize(LaunchCtx, const float* partials, int B, int ntiles, int cstride, int fold2,
                        double count, const float* gb, float* nscale, float* nshift);
void launch_in_apply(LaunchCtx, const float* in, const float* nscale, const float* nshift,
                     const float* alpha, float* out, int B, long P);
void launch_tail_proj(LaunchCtx, const float* sp, const float* nscale, const float* nshift,
                      const float* alpha, const float* tailw, float* d, int B, long P2);
void launch_mask_stats(LaunchCtx, const float* dm, const float* scalars, int B, int T, int F, float* mstat);
void launch_final_combine(LaunchCtx, const float* spec, const float* dm, const float* dc, const float* mstat,
                          const float* mk_scalars, const float* prelu_out, const float* cx_bias,
                          int B, int T, int F, float* out_re, float* out_im, float* tap_mask, float* tap_cplx);
void launch_cl_to_nchw(LaunchCtx, const float* in, float* out, int B, long P);

// ----------------------------- conformer.hip ------------------------------------
struct ConfWeights {
    const float *ff1_w1, *ff1_b1, *ff1_w2, *ff1_b2;
    const float *qkv_w, *qkv_b, *wo, *bo, *rel;
    const float *pw1_w, *pw1_b, *dw_w, *dw_b, *pw2_w, *pw2_b;
    const float *ff2_w1, *ff2_b1, *ff2_w2, *ff2_b2, *post_gb;
    int max_pos;
};
struct ConfBuffers {
    float *xa, *xb;        // residual stream ping/pong [M,64]
    float *q, *k, *v, *o;  // fragment-major per (sequence, head)
    float *u, *w;          // conv module [M,128]
};
TokMap make_flat_map(long M);
TokMap make_seq_map(int N, int L, int inner, long outer, long istride, long lstride);
size_t conf_qkv_floats(int N, int L);   // floats of one of q/k/v/o for N sequences of length L (Lb rounded up to even)
// one ConformerBlock on the residual stream in bufs.xa (in place); taps (may be NULL) -> 4 x [M,64].
// outer_residual: add the block's input again after post_norm (what TSCB does, generator.py:95,97).
// mask (may be NULL): [N, L] bytes, non-zero = keep - ConformerBlock.forward(x, mask), conformer.py:113-126, 217.
void conformer_forward(LaunchCtx, const ConfWeights&, const ConfBuffers&, const TokMap& seq, long M, float* taps,
                       bool outer_residual, const unsigned char* mask = nullptr);

// --------------------------- x3 mode (f16 split products) ------------------------
struct ConfWeightsX3 {
    const _Float16 *ff1_w1, *ff1_w2, *qkv_w, *wo, *pw1_w, *pw2_w, *ff2_w1, *ff2_w2;
    const _Float16* rel_img;    // [2*max_pos+1][hi 16 | lo 16] halfs
    const _Float16* dw_img;     // [8 channel groups][9][64 lanes][hi 4 | lo 4]: Toeplitz operands of the depthwise taps (dwpw2t_x3_kernel)
    const _Float16* rel_planes; // 4 x [2*max_pos+1 rows, reversed][8 halfs]: hi d0-7 | hi d8-15 | lo d0-7 | lo d8-15 (attn32_x3.hip)
    const _Float16 *ff1_w1_32, *ff1_w2_32, *ff2_w1_32, *ff2_w2_32;   // FeedForward operand images of ffn32_x3_kernel (ffn32_x3.hip)
};
// ffn32_x3.hip: FeedForward (+ post LayerNorm + TSCB residual when final_) on 32x32x16 MFMAs; x0 / post_gb as ffn_x3_kernel
void launch_ffn32_x3(LaunchCtx, bool final_, const float* xin, float* xout, const float* x0, const float* post_gb,
                     const _Float16* w1i, const float* b1, const _Float16* w2i, const float* b2, long M);
// The six launches of one ConformerBlock as a table of per-stage entry points.  conformer_x3.hip is compiled twice (the
// split-f16 build and its single-product twin, cmgan_amd/build.py): each build exports its own table, and
// conformer_forward_tbl (conformer.hip) walks ANY table - the pure ones behind conformer_forward_x3 / _x1, or one mixed
// per stage from the two (CMGAN_MFMA_F16MIX).  Stages exchange only fp32 rows and hi | lo fp16 images whose layout is
// the same in both builds, so any mix is well-formed.
struct ConfStageTbl {
    // ff: 1 = ff1 (xin -> xout), 2 = ff2 + post LayerNorm (+ x0 = the TSCB residual); plain = ff2 WITHOUT post norm (tap)
    void (*ffn)(LaunchCtx, int ff, bool plain, const float* xin, float* xout, const float* x0, const ConfWeights&,
                const ConfWeightsX3&, long M);
    void (*qkv)(LaunchCtx, const float* x, const TokMap& seq, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&);
    void (*attn)(LaunchCtx, float* x, const TokMap& seq, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&,
                 const unsigned char* mask);
    void (*pw1glu)(LaunchCtx, const float* x, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&, long M);
    void (*dwpw2)(LaunchCtx, float* x, const TokMap& seq, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&);
};
const ConfStageTbl& conf_stages_x3();
const ConfStageTbl& conf_stages_x1();
// stage order, tap copies and buffer roles of conformer_forward_x3, on any table (ff1 / ff2 may come from different builds)
bool conformer_forward_tbl(LaunchCtx, const ConfStageTbl& ff1, const ConfStageTbl& qkv, const ConfStageTbl& attn,
                           const ConfStageTbl& pw1, const ConfStageTbl& dwpw2, const ConfStageTbl& ff2,
                           const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&, const TokMap& seq, long M,
                           float* taps, bool outer_residual, const unsigned char* mask);
// Returns false WITHOUT launching anything when the shape is outside what the split-f16 conv-module kernel can address
// (dwpw2t_x3_kernel: buffer descriptor + 32-bit lane byte offsets over a sequence's rows of the GLU output, 512 B each;
// the 32-position tiles of a call counted in an int) - the limit lives with the kernel, every caller gets it.
bool conformer_x3_addressable(const TokMap& seq);
bool conformer_forward_x3(LaunchCtx, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&, const TokMap& seq,
                          long M, float* taps, bool outer_residual, const unsigned char* mask = nullptr);
void launch_dwconv(LaunchCtx, const float* u, float* out, const float* dw_w, const float* dw_b, const TokMap& seq);
// attn32_x3.hip: the x3 attention on 32x32x16 MFMAs (32-token Q / K / V tile images) + to_out + residual
void launch_qkv32_x3(LaunchCtx, const float* x, const TokMap& seq, const _Float16* wi, const float* b,
                     _Float16* qimg, _Float16* kimg, _Float16* vimg);
void launch_attn32_out_x3(LaunchCtx, const _Float16* qimg, const _Float16* kimg, const _Float16* vimg,
                          const _Float16* rel_img, int max_pos, float* x, const TokMap& seq, const _Float16* woi,
                          const float* bo, const unsigned char* mask);
void launch_attn_sp_out_x3(LaunchCtx, const _Float16* qimg, const _Float16* kimg, const _Float16* vimg,
                           const _Float16* rel_img, int max_pos, float* x, const TokMap& seq, const _Float16* woi,
                           const float* bo);
int  conv3x_ntiles(int T, int F, int cout);
void launch_conv3_x3(LaunchCtx, const ConvArgs&, const void* w16, int B, int time_taps, int cout);
void launch_conv3_x3_dgrad(LaunchCtx, const ConvArgs&, const void* w16, int B);      // 2 time taps, 64 -> 64, revt / accum honoured
void launch_selftest_x3(hipStream_t, const void* a_img, const float* b_fm, float* d, int M32);

// ------------------------------- selftest ---------------------------------------
void launch_selftest_mfma(hipStream_t, const float* a_fm, const float* b_fm, float* d, int KB);

// ----------------------------- single-product twins (F16X1 mode) ------------------------------
// conformer_x3.hip, attn32_x3.hip and conv_x3.hip are compiled a second time with -DX3_SINGLE -DX3_TERMS=1
// (cmgan_amd/build.py): every product term with a lo operand is compiled out and operands are rounded to nearest
// (common.hip.h), kernels live in their own namespace, and the host entry points take the _x1 names below.
bool conformer_forward_x1(LaunchCtx, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&, const TokMap& seq,
                          long M, float* taps, bool outer_residual, const unsigned char* mask = nullptr);
void launch_qkv32_x1(LaunchCtx, const float* x, const TokMap& seq, const _Float16* wi, const float* b,
                     _Float16* qimg, _Float16* kimg, _Float16* vimg);
void launch_attn32_out_x1(LaunchCtx, const _Float16* qimg, const _Float16* kimg, const _Float16* vimg,
                          const _Float16* rel_img, int max_pos, float* x, const TokMap& seq, const _Float16* woi,
                          const float* bo, const unsigned char* mask);
void launch_attn_sp_out_x1(LaunchCtx, const _Float16* qimg, const _Float16* kimg, const _Float16* vimg,
                           const _Float16* rel_img, int max_pos, float* x, const TokMap& seq, const _Float16* woi,
                           const float* bo);
void launch_conv3_x1(LaunchCtx, const ConvArgs&, const void* w16, int B, int time_taps, int cout);
void launch_ffn32_x1(LaunchCtx, bool final_, const float* xin, float* xout, const float* x0, const float* post_gb,
                     const _Float16* w1i, const float* b1, const _Float16* w2i, const float* b2, long M);
#ifdef X3_SINGLE
#define X3_NS x1k
#define conformer_forward_x3 conformer_forward_x1
#define conformer_x3_addressable conformer_x1_addressable
#define conf_stages_x3 conf_stages_x1
#define launch_qkv32_x3 launch_qkv32_x1
#define launch_attn32_out_x3 launch_attn32_out_x1
#define launch_attn_sp_out_x3 launch_attn_sp_out_x1
#define launch_conv3_x3 launch_conv3_x1
#define launch_ffn32_x3 launch_ffn32_x1
#else
#define X3_NS x3k
#endif
