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

struct LaunchCtx {
    hipStream_t stream;
    Profiler* prof;
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
    } while (0)

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
void conformer_forward(LaunchCtx, const ConfWeights&, const ConfBuffers&, const TokMap& seq, long M, float* taps,
                       bool outer_residual);

// --------------------------- x3 mode (f16 split products) ------------------------
struct ConfWeightsX3 {
    const _Float16 *ff1_w1, *ff1_w2, *qkv_w, *wo, *pw1_w, *pw2_w, *ff2_w1, *ff2_w2;
    const _Float16* rel_img;    // [2*max_pos+1][hi 16 | lo 16] halfs
};
void conformer_forward_x3(LaunchCtx, const ConfWeights&, const ConfWeightsX3&, const ConfBuffers&, const TokMap& seq,
                          long M, float* taps, bool outer_residual);
void launch_dwconv(LaunchCtx, const float* u, float* out, const float* dw_w, const float* dw_b, const TokMap& seq);
int  conv3x_ntiles(int T, int F, int cout);
void launch_conv3_x3(LaunchCtx, const ConvArgs&, const void* w16, int B, int time_taps, int cout);
void launch_selftest_x3(hipStream_t, const void* a_img, const float* b_fm, float* d, int M32);

// ------------------------------- train.hip ---------------------------------------
#define LOSS_BLOCKS 256                       // fixed partial-sum shape: results do not depend on the batch split
void launch_loss_terms(LaunchCtx, const float* est_real, const float* est_imag, const float* clean_spec, int B,
                       long P, const float* est_audio, const float* clean_audio, long naudio, double* partials,
                       float* out4);

// training-mode FeedForward (forward with dropout masks, full backward) on raw parameters
struct FfnTrainParams {
    float *gamma, *beta;      // PreNorm LayerNorm(64)                      conformer.py:68
    float *w1, *b1;           // Linear(64, 256): weight [256,64], bias     conformer.py:140
    float *w2, *b2;           // Linear(256, 64): weight [64,256], bias     conformer.py:143
};
#define FFN_WGRAD_SPLIT 64
#define FFN_COLSUM_BLOCKS 128
size_t ffn_train_ws_floats(long M);
void launch_ffn_train_forward(LaunchCtx, const float* x, long M, const FfnTrainParams& p, const float* m1,
                              const float* m2, float* y, float* ws);
void launch_ffn_train_backward(LaunchCtx, const float* x, const float* dy, long M, const FfnTrainParams& p,
                               const float* m1, const float* m2, float* dx, const FfnTrainParams& grad, float* ws);

// training-mode ConformerConvModule (BatchNorm1d on batch statistics) forward + backward on raw parameters
struct ConvModTrainParams {
    float *ln_w, *ln_b;       // net.0  LayerNorm(64)                          conformer.py:161
    float *pw1_w, *pw1_b;     // net.2  Conv1d(64, 256, 1): [256,64], [256]    conformer.py:163
    float *dw_w, *dw_b;       // net.4  depthwise Conv1d k=31: [128,31], [128] conformer.py:165-167
    float *bn_w, *bn_b;       // net.5  BatchNorm1d(128) gamma, beta           conformer.py:168
    float *pw2_w, *pw2_b;     // net.7  Conv1d(128, 64, 1): [64,128], [64]     conformer.py:170
};
size_t convmod_train_ws_floats(int N, int L);
void launch_convmod_train_forward(LaunchCtx, const float* x, int N, int L, const ConvModTrainParams& p,
                                  float* running_mean, float* running_var, float* y, float* ws);
void launch_convmod_train_backward(LaunchCtx, const float* x, const float* dy, int N, int L,
                                   const ConvModTrainParams& p, float* dx, const ConvModTrainParams& grad, float* ws);
// training-mode PreNorm(Attention) forward + backward on raw parameters
struct AttnTrainParams {
    float *ln_w, *ln_b;       // attn.norm               LayerNorm(64)          conformer.py:68
    float *wq, *wkv;          // attn.fn.to_q [64,64], attn.fn.to_kv [128,64]   conformer.py:81-82 (no bias)
    float *wo, *bo;           // attn.fn.to_out [64,64], [64]                   conformer.py:83
    float *rel;               // attn.fn.rel_pos_emb [2 max_pos + 1, 16]        conformer.py:86
};
size_t attn_train_ws_floats(int N, int L);
int attn_train_max_len();
void launch_attn_train_forward(LaunchCtx, const float* x, int N, int L, const AttnTrainParams& p, int max_pos,
                               const float* mask, float* y, float* ws);
void launch_attn_train_backward(LaunchCtx, const float* x, const float* dy, int N, int L, const AttnTrainParams& p,
                                int max_pos, const float* mask, float* dx, const AttnTrainParams& grad, float* ws);
void launch_swap_axes(LaunchCtx, const float* in, float* out, int B, int A, int C);
void launch_add(LaunchCtx, const float* a, const float* b, float* out, long n);
size_t ln_train_ws_floats(long M);
void launch_ln_train_forward(LaunchCtx, const float* x, long M, const float* gamma, const float* beta, float* y);
void launch_ln_train_backward(LaunchCtx, const float* x, const float* dy, long M, const float* gamma, const float* beta,
                              float* dx, float* dgamma, float* dbeta, float* ws);
void launch_adamw(LaunchCtx, float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                  float eps, float wd, int step);

// ------------------------------- selftest ---------------------------------------
void launch_selftest_mfma(hipStream_t, const float* a_fm, const float* b_fm, float* d, int KB);
