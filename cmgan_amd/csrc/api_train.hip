// api_train.hip - C ABI of the training-step slices (include/cmgan_hip.h, "training" entry points): loss terms,
// train-mode FeedForward / ConformerConvModule / Attention forward + backward, block glue, AdamW.  Kept apart from
// api.hip so that the inference path's sources (and the digest bench.py ties its PMC evidence to) do not change
// while the training side grows.
#include "api_internal.h"
#include "train.h"

// ------------------------------------------------------------------------------------
// training / validation step pieces (src/train.py)
// ------------------------------------------------------------------------------------
extern "C" int cmgan_loss_terms(cmgan_handle* h, const float* est_real, const float* est_imag,
                                const float* clean_spec, int B, int T, const float* est_audio,
                                const float* clean_audio, int L_audio, float* out4, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    const bool spec = est_real || est_imag || clean_spec, audio = est_audio || clean_audio;
    if (!out4 || B <= 0 || (!spec && !audio)) return fail(h, CMGAN_E_BADARG, "cmgan_loss_terms: bad argument");
    if (spec && (!est_real || !est_imag || !clean_spec || T <= 0))
        return fail(h, CMGAN_E_BADARG, "cmgan_loss_terms: spectral terms need est_real, est_imag, clean_spec and T > 0");
    if (audio && (!est_audio || !clean_audio || L_audio <= 0))
        return fail(h, CMGAN_E_BADARG, "cmgan_loss_terms: time term needs est_audio, clean_audio and L_audio > 0");
    launch_loss_terms(begin(h, stream), est_real, est_imag, clean_spec, B, (long)T * h->cfg.num_features, est_audio,
                      clean_audio, (long)B * L_audio, h->d_loss, out4);
    return check_launch(h, "loss_terms");
}

extern "C" size_t cmgan_ffn_train_workspace_bytes(const cmgan_handle* h, long long M) {
    if (!h || M <= 0) return 0;
    return ffn_train_ws_floats((long)M) * sizeof(float);
}

static bool ffn_params_ok(const cmgan_ffn_params* p) {
    return p && p->ln_weight && p->ln_bias && p->w1 && p->b1 && p->w2 && p->b2;
}
static FfnTrainParams ffn_params(const cmgan_ffn_params* p) {
    return FfnTrainParams{p->ln_weight, p->ln_bias, p->w1, p->b1, p->w2, p->b2};
}

extern "C" int cmgan_ffn_train_forward(cmgan_handle* h, const float* x, long long M, const cmgan_ffn_params* params,
                                       const unsigned char* mask1, const unsigned char* mask2, float mask_scale,
                                       const float* residual, float* y, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || M <= 0 || !ffn_params_ok(params)) return fail(h, CMGAN_E_BADARG, "cmgan_ffn_train_forward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, ffn_train_ws_floats((long)M) * sizeof(float))) return rc;
    launch_ffn_train_forward(begin(h, stream), x, (long)M, ffn_params(params), mask1, mask2, mask_scale, residual, y,
                             (float*)ws);
    return check_launch(h, "ffn_train_forward");
}

extern "C" int cmgan_ffn_train_backward(cmgan_handle* h, const float* x, const float* dy, long long M,
                                        const cmgan_ffn_params* params, const unsigned char* mask1,
                                        const unsigned char* mask2, float mask_scale, const float* dresidual,
                                        float* dx, const cmgan_ffn_params* grads, void* ws, size_t ws_bytes,
                                        void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !dy || !dx || M <= 0 || !ffn_params_ok(params) || !ffn_params_ok(grads))
        return fail(h, CMGAN_E_BADARG, "cmgan_ffn_train_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, ffn_train_ws_floats((long)M) * sizeof(float))) return rc;
    if (!launch_ffn_train_backward(begin(h, stream), x, dy, (long)M, ffn_params(params), mask1, mask2, mask_scale,
                                   dresidual, dx, ffn_params(grads), (float*)ws))
        return fail(h, CMGAN_E_BADARG, "cmgan_ffn_train_backward: one keep-mask without the other (or a device that refuses the "
                                       "fused kernel's LDS) needs the un-fused workspace: set CMGAN_FFN_BWD_FUSED=0");
    return check_launch(h, "ffn_train_backward");
}

extern "C" size_t cmgan_convmod_train_workspace_bytes(const cmgan_handle* h, int N, int L) {
    if (!h || N <= 0 || L <= 0) return 0;
    return convmod_train_ws_floats(N, L) * sizeof(float);
}

static bool convmod_params_ok(const cmgan_convmod_params* p) {
    return p && p->ln_weight && p->ln_bias && p->pw1_weight && p->pw1_bias && p->dw_weight && p->dw_bias &&
           p->bn_weight && p->bn_bias && p->pw2_weight && p->pw2_bias;
}
static ConvModTrainParams convmod_params(const cmgan_convmod_params* p) {
    return ConvModTrainParams{p->ln_weight, p->ln_bias, p->pw1_weight, p->pw1_bias, p->dw_weight, p->dw_bias,
                              p->bn_weight, p->bn_bias, p->pw2_weight, p->pw2_bias};
}

extern "C" int cmgan_convmod_train_forward(cmgan_handle* h, const float* x, int N, int L,
                                           const cmgan_convmod_params* params, float* running_mean,
                                           float* running_var, const float* residual, float* y, void* ws,
                                           size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || N <= 0 || L <= 0 || !convmod_params_ok(params) || (!running_mean != !running_var))
        return fail(h, CMGAN_E_BADARG, "cmgan_convmod_train_forward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, convmod_train_ws_floats(N, L) * sizeof(float))) return rc;
    launch_convmod_train_forward(begin(h, stream), x, N, L, convmod_params(params), running_mean, running_var,
                                 residual, y, (float*)ws);
    return check_launch(h, "convmod_train_forward");
}

extern "C" int cmgan_convmod_train_backward(cmgan_handle* h, const float* x, const float* dy, int N, int L,
                                            const cmgan_convmod_params* params, const float* dresidual, float* dx,
                                            const cmgan_convmod_params* grads, void* ws, size_t ws_bytes,
                                            void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !dy || !dx || N <= 0 || L <= 0 || !convmod_params_ok(params) || !convmod_params_ok(grads))
        return fail(h, CMGAN_E_BADARG, "cmgan_convmod_train_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, convmod_train_ws_floats(N, L) * sizeof(float))) return rc;
    if (!launch_convmod_train_backward(begin(h, stream), x, dy, N, L, convmod_params(params), dresidual, dx,
                                  convmod_params(grads), (float*)ws))
        return fail(h, CMGAN_E_BADARG, "cmgan_convmod_train_backward: the device refused a fused backward kernel's LDS; set "
                                       "CMGAN_CM_BWD1_FUSED=0 CMGAN_CM_BWD2_FUSED=0 (un-fused workspace)");
    return check_launch(h, "convmod_train_backward");
}

extern "C" size_t cmgan_attn_train_workspace_bytes(const cmgan_handle* h, int N, int L) {
    if (!h || N <= 0 || L <= 0 || L > attn_train_max_len()) return 0;
    return attn_train_ws_floats(N, L) * sizeof(float);
}

static bool attn_params_ok(const cmgan_attn_params* p) {
    return p && p->ln_weight && p->ln_bias && p->to_q_weight && p->to_kv_weight && p->to_out_weight &&
           p->to_out_bias && p->rel_pos_emb;
}
static AttnTrainParams attn_params(const cmgan_attn_params* p) {
    return AttnTrainParams{p->ln_weight, p->ln_bias, p->to_q_weight, p->to_kv_weight, p->to_out_weight, p->to_out_bias,
                           p->rel_pos_emb};
}

extern "C" int cmgan_attn_train_forward(cmgan_handle* h, const float* x, int N, int L, const cmgan_attn_params* params,
                                        const unsigned char* mask, float mask_scale, const float* residual, float* y,
                                        void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || N <= 0 || L <= 0 || !attn_params_ok(params))
        return fail(h, CMGAN_E_BADARG, "cmgan_attn_train_forward: bad argument");
    if (L > attn_train_max_len())
        return fail(h, CMGAN_E_UNSUPPORTED, "cmgan_attn_train: sequences up to %d positions (got %d)", attn_train_max_len(), L);
    if (int rc = check_ws(h, ws, ws_bytes, attn_train_ws_floats(N, L) * sizeof(float))) return rc;
    launch_attn_train_forward(begin(h, stream), x, N, L, attn_params(params), h->cfg.max_pos_emb, mask, mask_scale,
                              residual, y, (float*)ws);
    return check_launch(h, "attn_train_forward");
}

extern "C" int cmgan_attn_train_backward(cmgan_handle* h, const float* x, const float* dy, int N, int L,
                                         const cmgan_attn_params* params, const unsigned char* mask, float mask_scale,
                                         const float* dresidual, float* dx, const cmgan_attn_params* grads, void* ws,
                                         size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !dy || !dx || N <= 0 || L <= 0 || !attn_params_ok(params) || !attn_params_ok(grads))
        return fail(h, CMGAN_E_BADARG, "cmgan_attn_train_backward: bad argument");
    if (L > attn_train_max_len())
        return fail(h, CMGAN_E_UNSUPPORTED, "cmgan_attn_train: sequences up to %d positions (got %d)", attn_train_max_len(), L);
    if (int rc = check_ws(h, ws, ws_bytes, attn_train_ws_floats(N, L) * sizeof(float))) return rc;
    launch_attn_train_backward(begin(h, stream), x, dy, N, L, attn_params(params), h->cfg.max_pos_emb, mask, mask_scale,
                               dresidual, dx, attn_params(grads), (float*)ws);
    return check_launch(h, "attn_train_backward");
}

extern "C" int cmgan_swap_axes(cmgan_handle* h, const float* in, const float* add, float* out, int B, int A, int C,
                               void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!in || !out || in == out || add == out || B <= 0 || A <= 0 || C <= 0)
        return fail(h, CMGAN_E_BADARG, "cmgan_swap_axes: bad argument");
    launch_swap_axes(begin(h, stream), in, add, out, B, A, C);
    return check_launch(h, "swap_axes");
}

extern "C" int cmgan_dropout_masks(cmgan_handle* h, unsigned char* masks, long long nbytes, float keep_prob,
                                   unsigned long long* state, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!masks || !state || nbytes <= 0 || (nbytes & 15) || !(keep_prob >= 0.f && keep_prob <= 1.f))
        return fail(h, CMGAN_E_BADARG, "cmgan_dropout_masks: bad argument (nbytes must be a positive multiple of 16)");
    launch_dropout_masks(begin(h, stream), masks, (long)nbytes, keep_prob, state);
    return check_launch(h, "dropout_masks");
}

extern "C" int cmgan_add(cmgan_handle* h, const float* a, const float* b, float* out, long long n, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!a || !b || !out || n <= 0 || (n & 3)) return fail(h, CMGAN_E_BADARG, "cmgan_add: bad argument (n must be a multiple of 4)");
    launch_add(begin(h, stream), a, b, out, (long)n);
    return check_launch(h, "add");
}

extern "C" size_t cmgan_layernorm_train_workspace_bytes(const cmgan_handle* h, long long M) {
    if (!h || M <= 0) return 0;
    return ln_train_ws_floats((long)M) * sizeof(float);
}

extern "C" int cmgan_layernorm_train_forward(cmgan_handle* h, const float* x, long long M, const float* weight,
                                             const float* bias, const float* residual, float* y, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || !weight || !bias || M <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_layernorm_train_forward: bad argument");
    launch_ln_train_forward(begin(h, stream), x, (long)M, weight, bias, residual, y);
    return check_launch(h, "layernorm_train_forward");
}

extern "C" int cmgan_layernorm_train_backward(cmgan_handle* h, const float* x, const float* dy, long long M,
                                              const float* weight, const float* bias, float* dx, float* dweight,
                                              float* dbias, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !dy || !dx || !weight || !bias || !dweight || !dbias || M <= 0)
        return fail(h, CMGAN_E_BADARG, "cmgan_layernorm_train_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, ln_train_ws_floats((long)M) * sizeof(float))) return rc;
    launch_ln_train_backward(begin(h, stream), x, dy, (long)M, weight, bias, dx, dweight, dbias, (float*)ws);
    return check_launch(h, "layernorm_train_backward");
}

extern "C" size_t cmgan_dense_train_workspace_bytes(const cmgan_handle* h, int B, int T, int F) {
    if (!h || B <= 0 || T <= 0 || F <= 0) return 0;
    return dense_train_ws_floats(B, T, F) * sizeof(float);
}

static bool dense_params_ok(const cmgan_dense_params* p) {
    if (!p) return false;
    for (int i = 0; i < 4; ++i)
        if (!p->conv_weight[i] || !p->conv_bias[i] || !p->norm_weight[i] || !p->norm_bias[i] || !p->prelu_weight[i]) return false;
    return true;
}
static DenseTrainParams dense_params(const cmgan_dense_params* p) {
    DenseTrainParams d;
    for (int i = 0; i < 4; ++i) {
        d.conv_w[i] = p->conv_weight[i]; d.conv_b[i] = p->conv_bias[i];
        d.norm_w[i] = p->norm_weight[i]; d.norm_b[i] = p->norm_bias[i]; d.prelu_w[i] = p->prelu_weight[i];
    }
    return d;
}

extern "C" int cmgan_dense_train_forward(cmgan_handle* h, const float* x, int B, int T, int F,
                                         const cmgan_dense_params* params, float* y, void* ws, size_t ws_bytes,
                                         void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || B <= 0 || T <= 0 || F <= 0 || !dense_params_ok(params))
        return fail(h, CMGAN_E_BADARG, "cmgan_dense_train_forward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, dense_train_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_dense_train_forward(begin(h, stream), x, B, T, F, dense_params(params), y, (float*)ws);
    return check_launch(h, "dense_train_forward");
}

extern "C" int cmgan_dense_train_backward(cmgan_handle* h, const float* x, const float* dy, int B, int T, int F,
                                          const cmgan_dense_params* params, float* dx, const cmgan_dense_params* grads,
                                          void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !dy || !dx || B <= 0 || T <= 0 || F <= 0 || !dense_params_ok(params) || !dense_params_ok(grads))
        return fail(h, CMGAN_E_BADARG, "cmgan_dense_train_backward: bad argument");
    if (dx == x || dx == dy)                              // dx is accumulated in place while x and dy are still being read
        return fail(h, CMGAN_E_BADARG, "cmgan_dense_train_backward: dx must not alias x or dy");
    if (int rc = check_ws(h, ws, ws_bytes, dense_train_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_dense_train_backward(begin(h, stream), x, dy, B, T, F, dense_params(params), dx, dense_params(grads),
                                (float*)ws);
    return check_launch(h, "dense_train_backward");
}

// ---- DenseEncoder / MaskDecoder / ComplexDecoder -----------------------------------------------------------------
static bool encoder_params_ok(const cmgan_encoder_params* p) {
    return p && p->conv1_weight && p->conv1_bias && p->norm1_weight && p->norm1_bias && p->prelu1_weight &&
           dense_params_ok(&p->dense) && p->conv2_weight && p->conv2_bias && p->norm2_weight && p->norm2_bias &&
           p->prelu2_weight;
}
static EncoderTrainParams encoder_params(const cmgan_encoder_params* p) {
    EncoderTrainParams e;
    e.c1_w = p->conv1_weight; e.c1_b = p->conv1_bias; e.n1_w = p->norm1_weight; e.n1_b = p->norm1_bias; e.p1_w = p->prelu1_weight;
    e.dense = dense_params(&p->dense);
    e.c2_w = p->conv2_weight; e.c2_b = p->conv2_bias; e.n2_w = p->norm2_weight; e.n2_b = p->norm2_bias; e.p2_w = p->prelu2_weight;
    return e;
}

extern "C" size_t cmgan_encoder_train_workspace_bytes(const cmgan_handle* h, int B, int T, int F) {
    if (!h || B <= 0 || T <= 0 || F <= 0) return 0;
    return encoder_train_ws_floats(B, T, F) * sizeof(float);
}

extern "C" int cmgan_encoder_train_forward(cmgan_handle* h, const float* xin, int B, int T, int F,
                                           const cmgan_encoder_params* params, float* y, void* ws, size_t ws_bytes,
                                           void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!xin || !y || B <= 0 || T <= 0 || F <= 0 || !encoder_params_ok(params))
        return fail(h, CMGAN_E_BADARG, "cmgan_encoder_train_forward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, encoder_train_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_encoder_train_forward(begin(h, stream), xin, B, T, F, encoder_params(params), y, (float*)ws);
    return check_launch(h, "encoder_train_forward");
}

extern "C" int cmgan_encoder_train_backward(cmgan_handle* h, const float* xin, const float* dy, int B, int T, int F,
                                            const cmgan_encoder_params* params, const cmgan_encoder_params* grads,
                                            void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!xin || !dy || B <= 0 || T <= 0 || F <= 0 || !encoder_params_ok(params) || !encoder_params_ok(grads))
        return fail(h, CMGAN_E_BADARG, "cmgan_encoder_train_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, encoder_train_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_encoder_train_backward(begin(h, stream), xin, dy, B, T, F, encoder_params(params), encoder_params(grads),
                                  (float*)ws);
    return check_launch(h, "encoder_train_backward");
}

static bool decoder_params_ok(const cmgan_decoder_params* p, int kind) {
    if (!p || !dense_params_ok(&p->dense) || !p->sub_pixel_weight || !p->sub_pixel_bias || !p->conv_weight ||
        !p->conv_bias || !p->norm_weight || !p->norm_bias || !p->prelu_weight)
        return false;
    return kind == CMGAN_DECODER_COMPLEX || (p->final_weight && p->final_bias && p->prelu_out_weight);
}
static DecoderTrainParams decoder_params(const cmgan_decoder_params* p) {
    DecoderTrainParams d;
    d.dense = dense_params(&p->dense);
    d.sp_w = p->sub_pixel_weight; d.sp_b = p->sub_pixel_bias; d.c_w = p->conv_weight; d.c_b = p->conv_bias;
    d.n_w = p->norm_weight; d.n_b = p->norm_bias; d.p_w = p->prelu_weight;
    d.f_w = p->final_weight; d.f_b = p->final_bias; d.po_w = p->prelu_out_weight;
    return d;
}

extern "C" size_t cmgan_decoder_train_workspace_bytes(const cmgan_handle* h, int B, int T, int Fe) {
    if (!h || B <= 0 || T <= 0 || Fe <= 0) return 0;
    return decoder_train_ws_floats(B, T, Fe) * sizeof(float);
}

extern "C" int cmgan_decoder_train_forward(cmgan_handle* h, int kind, const float* x, int B, int T, int Fe,
                                           const cmgan_decoder_params* params, float* out, void* ws, size_t ws_bytes,
                                           void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if ((kind != CMGAN_DECODER_MASK && kind != CMGAN_DECODER_COMPLEX) || !x || !out || B <= 0 || T <= 0 || Fe <= 0 ||
        !decoder_params_ok(params, kind))
        return fail(h, CMGAN_E_BADARG, "cmgan_decoder_train_forward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, decoder_train_ws_floats(B, T, Fe) * sizeof(float))) return rc;
    launch_decoder_train_forward(begin(h, stream), kind, x, B, T, Fe, decoder_params(params), out, (float*)ws);
    return check_launch(h, "decoder_train_forward");
}

extern "C" int cmgan_decoder_train_backward(cmgan_handle* h, int kind, const float* x, const float* dout, int B, int T,
                                            int Fe, const cmgan_decoder_params* params, float* dx,
                                            const cmgan_decoder_params* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if ((kind != CMGAN_DECODER_MASK && kind != CMGAN_DECODER_COMPLEX) || !x || !dout || !dx || B <= 0 || T <= 0 ||
        Fe <= 0 || !decoder_params_ok(params, kind) || !decoder_params_ok(grads, kind))
        return fail(h, CMGAN_E_BADARG, "cmgan_decoder_train_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, decoder_train_ws_floats(B, T, Fe) * sizeof(float))) return rc;
    launch_decoder_train_backward(begin(h, stream), kind, x, dout, B, T, Fe, decoder_params(params), dx,
                                  decoder_params(grads), (float*)ws);
    return check_launch(h, "decoder_train_backward");
}

// ---- TSCNet.forward glue and the loss gradient ---------------------------------------------------------------------
extern "C" int cmgan_tscnet_prologue(cmgan_handle* h, const float* spec, int B, int T, float* xin, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!spec || !xin || B <= 0 || T <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_tscnet_prologue: bad argument");
    launch_tsc_prologue(begin(h, stream), spec, B, T, h->cfg.num_features, xin);
    return check_launch(h, "tscnet_prologue");
}

extern "C" int cmgan_tscnet_epilogue_forward(cmgan_handle* h, const float* spec, const float* mask, const float* cplx,
                                             int B, int T, float* est_real, float* est_imag, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!spec || !mask || !cplx || !est_real || !est_imag || B <= 0 || T <= 0)
        return fail(h, CMGAN_E_BADARG, "cmgan_tscnet_epilogue_forward: bad argument");
    launch_tsc_epilogue_forward(begin(h, stream), spec, mask, cplx, B, T, h->cfg.num_features, est_real, est_imag);
    return check_launch(h, "tscnet_epilogue_forward");
}

extern "C" int cmgan_tscnet_epilogue_backward(cmgan_handle* h, const float* spec, const float* d_real,
                                              const float* d_imag, int B, int T, float* dmask, float* dcplx,
                                              void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!spec || !d_real || !d_imag || !dmask || !dcplx || B <= 0 || T <= 0)
        return fail(h, CMGAN_E_BADARG, "cmgan_tscnet_epilogue_backward: bad argument");
    launch_tsc_epilogue_backward(begin(h, stream), spec, d_real, d_imag, B, T, h->cfg.num_features, dmask, dcplx);
    return check_launch(h, "tscnet_epilogue_backward");
}

extern "C" int cmgan_loss_backward(cmgan_handle* h, const float* est_real, const float* est_imag,
                                   const float* clean_spec, int B, int T, const float* est_audio,
                                   const float* clean_audio, float w_ri, float w_mag, float w_time, float* d_real,
                                   float* d_imag, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!est_real || !est_imag || !clean_spec || !d_real || !d_imag || B <= 0 || T <= 1)
        return fail(h, CMGAN_E_BADARG, "cmgan_loss_backward: bad argument");
    if ((est_audio == nullptr) != (clean_audio == nullptr))
        return fail(h, CMGAN_E_BADARG, "cmgan_loss_backward: the time term needs est_audio and clean_audio");
    if (w_time != 0.f && !est_audio)
        return fail(h, CMGAN_E_BADARG, "cmgan_loss_backward: w_time != 0 needs est_audio and clean_audio");
    launch_loss_backward(begin(h, stream), est_real, est_imag, clean_spec, est_audio, clean_audio, B, T,
                         h->cfg.num_features, h->cfg.n_fft, h->cfg.hop, w_ri, w_mag, w_time, d_real, d_imag);
    return check_launch(h, "loss_backward");
}

// ---- metric discriminator --------------------------------------------------------------------------------------------
static bool disc_params_ok(const cmgan_disc_params* p, bool need_uv) {
    if (!p) return false;
    for (int i = 0; i < 4; ++i) {
        if (!p->conv_weight_orig[i] || !p->norm_weight[i] || !p->norm_bias[i] || !p->prelu_weight[i]) return false;
        if (need_uv && (!p->conv_u[i] || !p->conv_v[i])) return false;
    }
    if (!p->fc1_weight_orig || !p->fc1_bias || !p->prelu5_weight || !p->fc2_weight_orig || !p->fc2_bias || !p->slope)
        return false;
    return !need_uv || (p->fc1_u && p->fc1_v && p->fc2_u && p->fc2_v);
}
static DiscParams disc_params(const cmgan_disc_params* p) {
    DiscParams d;
    for (int i = 0; i < 4; ++i) {
        d.conv_w[i] = p->conv_weight_orig[i]; d.conv_u[i] = p->conv_u[i]; d.conv_v[i] = p->conv_v[i];
        d.norm_w[i] = p->norm_weight[i]; d.norm_b[i] = p->norm_bias[i]; d.prelu_w[i] = p->prelu_weight[i];
    }
    d.fc1_w = p->fc1_weight_orig; d.fc1_b = p->fc1_bias; d.fc1_u = p->fc1_u; d.fc1_v = p->fc1_v;
    d.prelu5_w = p->prelu5_weight;
    d.fc2_w = p->fc2_weight_orig; d.fc2_b = p->fc2_bias; d.fc2_u = p->fc2_u; d.fc2_v = p->fc2_v;
    d.slope = p->slope;
    return d;
}

extern "C" size_t cmgan_disc_workspace_bytes(const cmgan_handle* h, int B, int T) {
    if (!h || B <= 0 || !disc_shape_ok(T, h->cfg.num_features)) return 0;
    return disc_ws_floats(B, T, h->cfg.num_features) * sizeof(float);
}

extern "C" int cmgan_mag_pair(cmgan_handle* h, const float* clean_spec, const float* est_real, const float* est_imag,
                              int B, int T, float* xy, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!clean_spec || !xy || B <= 0 || T <= 0 || ((est_real == nullptr) != (est_imag == nullptr)))
        return fail(h, CMGAN_E_BADARG, "cmgan_mag_pair: bad argument");
    launch_mag_pair(begin(h, stream), clean_spec, est_real, est_imag, B, T, h->cfg.num_features, xy);
    return check_launch(h, "mag_pair");
}

extern "C" int cmgan_mag_pair_backward(cmgan_handle* h, const float* est_real, const float* est_imag, const float* dxy,
                                       int B, int T, float scale, float* d_real, float* d_imag, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!est_real || !est_imag || !dxy || !d_real || !d_imag || B <= 0 || T <= 0)
        return fail(h, CMGAN_E_BADARG, "cmgan_mag_pair_backward: bad argument");
    launch_mag_pair_backward(begin(h, stream), est_real, est_imag, dxy, B, T, h->cfg.num_features, scale, d_real, d_imag);
    return check_launch(h, "mag_pair_backward");
}

extern "C" int cmgan_score_mse(cmgan_handle* h, const float* score, const float* target, int B, float scale, float* loss,
                               float* dscore, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!score || !loss || B <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_score_mse: bad argument");
    launch_score_mse(begin(h, stream), score, target, B, scale, loss, dscore);
    return check_launch(h, "score_mse");
}

extern "C" int cmgan_disc_forward(cmgan_handle* h, const float* xy, int B, int T, const cmgan_disc_params* params,
                                  const float* mask, int update_uv, float* score, void* ws, size_t ws_bytes,
                                  void* stream) {
    if (!h) return CMGAN_E_BADARG;
    const int F = h->cfg.num_features;
    if (!xy || !score || B <= 0 || !disc_shape_ok(T, F) || !disc_params_ok(params, true))
        return fail(h, CMGAN_E_BADARG, "cmgan_disc_forward: bad argument (T and F must be >= 16)");
    if (int rc = check_ws(h, ws, ws_bytes, disc_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_disc_forward(begin(h, stream), xy, B, T, F, disc_params(params), mask, update_uv, score, (float*)ws);
    return check_launch(h, "disc_forward");
}

extern "C" int cmgan_disc_backward(cmgan_handle* h, const float* xy, const float* dscore, int B, int T,
                                   const cmgan_disc_params* params, const float* mask, float* dxy,
                                   const cmgan_disc_params* grads, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    const int F = h->cfg.num_features;
    if (!xy || !dscore || B <= 0 || !disc_shape_ok(T, F) || !disc_params_ok(params, false) || !disc_params_ok(grads, false))
        return fail(h, CMGAN_E_BADARG, "cmgan_disc_backward: bad argument");
    if (int rc = check_ws(h, ws, ws_bytes, disc_ws_floats(B, T, F) * sizeof(float))) return rc;
    launch_disc_backward(begin(h, stream), xy, dscore, B, T, F, disc_params(params), mask, dxy, disc_params(grads),
                         (float*)ws);
    return check_launch(h, "disc_backward");
}

extern "C" int cmgan_adamw_step_dev(cmgan_handle* h, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                    long long n, float* state, float beta1, float beta2, float eps, float weight_decay,
                                    void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !state || n <= 0 || !(beta1 >= 0.f && beta1 < 1.f) ||
        !(beta2 >= 0.f && beta2 < 1.f))
        return fail(h, CMGAN_E_BADARG, "cmgan_adamw_step_dev: bad argument");
    launch_adamw_dev(begin(h, stream), params, grads, exp_avg, exp_avg_sq, (long)n, state, beta1, beta2, eps, weight_decay);
    return check_launch(h, "adamw_step_dev");
}

extern "C" int cmgan_adamw_step(cmgan_handle* h, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                long long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int step, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step < 1 || !(beta1 >= 0.f && beta1 < 1.f) ||
        !(beta2 >= 0.f && beta2 < 1.f))
        return fail(h, CMGAN_E_BADARG, "cmgan_adamw_step: bad argument");
    launch_adamw(begin(h, stream), params, grads, exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, step);
    return check_launch(h, "adamw_step");
}

