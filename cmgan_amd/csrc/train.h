// train.h - launcher interface between api_train.hip and train.hip (training-step slices).
#pragma once
#include "kernels.h"

// ------------------------------- train.hip ---------------------------------------
#ifndef LOSS_BLOCKS
#define LOSS_BLOCKS 256
#endif                       // fixed partial-sum shape: results do not depend on the batch split
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
#define FFN_COLSUM_BLOCKS 512
size_t ffn_train_ws_floats(long M);
// dropout keep-masks: bytes (non-zero = keep), kept values scaled by mask_scale = 1 / (1 - p); NULL = no dropout.
// res / dres (here and in the two modules below): optional [M,64] tensor added in the final store, i.e. the residual
// connection of conformer.py:216-219 fused into the branch (y = res + f(x); dx = dres + f'(dy)); NULL = none.
void launch_ffn_train_forward(LaunchCtx, const float* x, long M, const FfnTrainParams& p, const unsigned char* m1,
                              const unsigned char* m2, float mask_scale, const float* res, float* y, float* ws);
// false: the call cannot be served with the compact workspace (train.hip ffn_ws_compact)
bool launch_ffn_train_backward(LaunchCtx, const float* x, const float* dy, long M, const FfnTrainParams& p,
                               const unsigned char* m1, const unsigned char* m2, float mask_scale, const float* dres,
                               float* dx, const FfnTrainParams& grad, float* ws);

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
                                  float* running_mean, float* running_var, const float* res, float* y, float* ws);
// false: the call cannot be served with the compact workspace (train.hip cm_ws_compact1 / 2)
bool launch_convmod_train_backward(LaunchCtx, const float* x, const float* dy, int N, int L,
                                   const ConvModTrainParams& p, const float* dres, float* dx,
                                   const ConvModTrainParams& grad, float* ws);
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
                               const unsigned char* mask, float mask_scale, const float* res, float* y, float* ws);
void launch_attn_train_backward(LaunchCtx, const float* x, const float* dy, int N, int L, const AttnTrainParams& p,
                                int max_pos, const unsigned char* mask, float mask_scale, const float* dres, float* dx,
                                const AttnTrainParams& grad, float* ws);
void launch_swap_axes(LaunchCtx, const float* in, const float* add, float* out, int B, int A, int C);
// nbytes % 16 == 0; state = {seed, offset} (device); advances the offset by nbytes / 16
void launch_dropout_masks(LaunchCtx, unsigned char* out, long nbytes, float keep, unsigned long long* state);
void launch_add(LaunchCtx, const float* a, const float* b, float* out, long n);
size_t ln_train_ws_floats(long M);
void launch_ln_train_forward(LaunchCtx, const float* x, long M, const float* gamma, const float* beta, const float* res,
                             float* y);
void launch_ln_train_backward(LaunchCtx, const float* x, const float* dy, long M, const float* gamma, const float* beta,
                              float* dx, float* dgamma, float* dbeta, float* ws);
// training-mode DilatedDenseNet (generator.py:6-47) forward + backward on raw parameters, channels-last [B,T,F,64]
struct DenseTrainParams {
    float *conv_w[4], *conv_b[4];     // conv{i+1}.weight [64, 64 (i+1), 2, 3], .bias [64]
    float *norm_w[4], *norm_b[4];     // norm{i+1}  InstanceNorm2d(64, affine)
    float *prelu_w[4];                // prelu{i+1} PReLU(64)
};
size_t dense_train_ws_floats(int B, int T, int F);
void launch_dense_train_forward(LaunchCtx, const float* x, int B, int T, int F, const DenseTrainParams& p, float* y,
                                float* ws);
void launch_dense_train_backward(LaunchCtx, const float* x, const float* dy, int B, int T, int F,
                                 const DenseTrainParams& p, float* dx, const DenseTrainParams& grad, float* ws);
// training-mode DenseEncoder (generator.py:50-69): conv_1 (1x1, 3 -> 64) + IN + PReLU, DilatedDenseNet, conv_2
// ((1,3), stride (1,2), padding (0,1)) + IN + PReLU; input [B,T,F,3] = (mag, re, im) per bin, output [B,T,F',64]
struct EncoderTrainParams {
    float *c1_w, *c1_b, *n1_w, *n1_b, *p1_w;      // conv_1.{0.weight [64,3,1,1], 0.bias, 1.weight, 1.bias, 2.weight}
    DenseTrainParams dense;                       // dilated_dense.*
    float *c2_w, *c2_b, *n2_w, *n2_b, *p2_w;      // conv_2.{0.weight [64,64,1,3], 0.bias, 1.weight, 1.bias, 2.weight}
};
size_t encoder_train_ws_floats(int B, int T, int F);
void launch_encoder_train_forward(LaunchCtx, const float* xin, int B, int T, int F, const EncoderTrainParams& p, float* y,
                                  float* ws);
void launch_encoder_train_backward(LaunchCtx, const float* xin, const float* dy, int B, int T, int F,
                                   const EncoderTrainParams& p, const EncoderTrainParams& grad, float* ws);
// training-mode MaskDecoder (kind 0, generator.py:121-138) and ComplexDecoder (kind 1, generator.py:141-156):
// DilatedDenseNet -> sub-pixel conv ((1,3), 64 -> 128, pixel shuffle x2 along frequency) -> head.
//   kind 0: conv_1 (1,2) 64 -> 1, InstanceNorm2d(1) + PReLU(1), final_conv 1x1, PReLU(num_features)  -> [B,T,F]
//   kind 1: InstanceNorm2d(64) + PReLU(64), conv (1,2) 64 -> 2                                        -> [B,T,F,2]
struct DecoderTrainParams {
    DenseTrainParams dense;          // dense_block.*
    float *sp_w, *sp_b;              // sub_pixel.conv.{weight [128,64,1,3], bias [128]}
    float *c_w, *c_b;                // conv_1 / conv: weight [NO,64,1,2], bias [NO]
    float *n_w, *n_b, *p_w;          // norm.{weight,bias} and prelu.weight ([1] for kind 0, [64] for kind 1)
    float *f_w, *f_b, *po_w;         // kind 0 only: final_conv.{weight [1,1,1,1], bias [1]}, prelu_out.weight [F]
};
size_t decoder_train_ws_floats(int B, int T, int Fe);
void launch_decoder_train_forward(LaunchCtx, int kind, const float* x, int B, int T, int Fe, const DecoderTrainParams& p,
                                  float* out, float* ws);
void launch_decoder_train_backward(LaunchCtx, int kind, const float* x, const float* dout, int B, int T, int Fe,
                                   const DecoderTrainParams& p, float* dx, const DecoderTrainParams& grad, float* ws);
// TSCNet.forward glue (generator.py:176-201): spec [B,2,T,F] -> xin [B,T,F,3]; est = mask * spec + complex_out
void launch_tsc_prologue(LaunchCtx, const float* spec, int B, int T, int F, float* xin);
void launch_tsc_epilogue_forward(LaunchCtx, const float* spec, const float* mask, const float* cplx, int B, int T, int F,
                                 float* est_real, float* est_imag);
void launch_tsc_epilogue_backward(LaunchCtx, const float* spec, const float* d_real, const float* d_imag, int B, int T, int F,
                                  float* dmask, float* dcplx);
// gradient of w_ri loss_ri + w_mag loss_mag + w_time time_loss (train.py:133-148) with respect to est_real / est_imag
void launch_loss_backward(LaunchCtx, const float* est_real, const float* est_imag, const float* clean_spec,
                          const float* est_audio, const float* clean_audio, int B, int T, int F, int nfft, int hop, float w_ri,
                          float w_mag, float w_time, float* d_real, float* d_imag);
// Column sums of one 16-token tile's LayerNorm-gradient rows - a = dxn * xhat (dgamma), b = dxn (dbeta), 16 features of
// each per lane in the chain layout - reduced over the tile's tokens with DPP row sums and written as row `tile` of two
// [tiles][64] slabs: the column-sum pass then reads 1/16 of what the full [M,64] tensors were (which existed only to
// be summed).  Callers mask rows past the end of the batch before the call.
__device__ __forceinline__ void ln_tile_colsums(f32x4 (&a)[4], f32x4 (&b)[4], int c, int g, long tile, float* __restrict__ g1c,
                                                float* __restrict__ dxc) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[kb][r] = red_c_sum(a[kb][r]);
            b[kb][r] = red_c_sum(b[kb][r]);
        }
        if (c == 0) {
            stg4(g1c + tile * 64 + 16 * kb + 4 * g, a[kb]);
            stg4(dxc + tile * 64 + 16 * kb + 4 * g, b[kb]);
        }
    }
}
// ------------------------------- disc.hip ----------------------------------------
// the metric discriminator Discriminator(ndf=16) (src/models/discriminator.py:29-64) on RAW parameters
struct DiscParams {
    float *conv_w[4], *conv_u[4], *conv_v[4];     // layers.{0,3,6,9}.weight_orig [Co,Ci,4,4], weight_u [Co], weight_v [16 Ci]
    float *norm_w[4], *norm_b[4], *prelu_w[4];    // layers.{1,4,7,10}.{weight,bias}, layers.{2,5,8,11}.weight
    float *fc1_w, *fc1_b, *fc1_u, *fc1_v;         // layers.14.{weight_orig [64,128], bias, weight_u, weight_v}
    float *prelu5_w;                              // layers.16.weight [64]
    float *fc2_w, *fc2_b, *fc2_u, *fc2_v;         // layers.17.{weight_orig [1,64], bias, weight_u, weight_v}
    float *slope;                                 // layers.18.slope [1]
};
size_t disc_ws_floats(int B, int T, int F);
bool disc_shape_ok(int T, int F);
// xy [B,T,F,2] = (|clean|, |est|) channels-last; score [B]; mask [B,64] = Dropout(0.3) keep-mask or NULL;
// update_uv != 0: one power iteration per spectral norm, written back to the u / v buffers (train mode)
void launch_disc_forward(LaunchCtx, const float* xy, int B, int T, int F, const DiscParams& p, const float* mask,
                         int update_uv, float* score, float* ws);
void launch_disc_backward(LaunchCtx, const float* xy, const float* dscore, int B, int T, int F, const DiscParams& p,
                          const float* mask, float* dxy, const DiscParams& grad, float* ws);
void launch_mag_pair(LaunchCtx, const float* clean_spec, const float* est_real, const float* est_imag, int B, int T, int F,
                     float* xy);
void launch_mag_pair_backward(LaunchCtx, const float* est_real, const float* est_imag, const float* dxy, int B, int T, int F,
                              float scale, float* d_real, float* d_imag);
void launch_score_mse(LaunchCtx, const float* score, const float* target, int B, float scale, float* loss, float* dscore);

void launch_adamw_dev(LaunchCtx, float* p, const float* g, float* m, float* v, long n, float* state, float b1, float b2,
                      float eps, float wd);
void launch_adamw(LaunchCtx, float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                  float eps, float wd, int step);

