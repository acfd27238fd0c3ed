/*
 * cmgan_hip.h - C ABI of libcmgan_hip.so: the CMGAN generator forward path
 * (waveform -> STFT -> power-compress -> TSCNet -> power-uncompress -> ISTFT)
 * as hand-written HIP kernels for MI355X (gfx950 / CDNA4).
 *
 * The reference (ruizhecao96/CMGAN) is pure Python and has no operator / FFI
 * boundary; the seam this library sits behind is the Python module boundary
 * (SURVEY.md section 8b).  Each entry point cites the reference code it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - plain C types only; every pointer named *_dev is a device pointer owned by
 *    the caller (e.g. torch.Tensor.data_ptr()); fp32 everywhere.
 *  - every call returns 0 (CMGAN_OK) or a negative CMGAN_E_* code; the message is
 *    available from cmgan_last_error().  Nothing throws across the ABI.
 *  - all kernels are enqueued on the caller's `stream` (a hipStream_t passed as
 *    void*; NULL = the legacy default stream) and return without synchronising.
 *    No allocation, free or synchronisation happens inside forward calls, so they
 *    can be captured into a hipGraph.
 *  - a handle is bound to the device that was current at cmgan_create() and is
 *    not re-entrant; use one handle per GPU / per process.
 */
#ifndef CMGAN_HIP_H
#define CMGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMGAN_OK             0
#define CMGAN_E_BADARG      -1   /* null pointer / non-positive size               */
#define CMGAN_E_BADSHAPE    -2   /* shape incompatible with the handle's config    */
#define CMGAN_E_UNSUPPORTED -3   /* config outside what the kernels are built for  */
#define CMGAN_E_WEIGHTS     -4   /* blob malformed / tensor missing / wrong size   */
#define CMGAN_E_WORKSPACE   -5   /* workspace too small or misaligned              */
#define CMGAN_E_HIP         -6   /* a HIP runtime call failed (see last_error)     */

#define CMGAN_ABI_VERSION 6

typedef struct cmgan_handle cmgan_handle;

/* Hyper-parameters that are constructor arguments or literals in the reference:
 * TSCNet(num_channel=64, num_features=n_fft/2+1)  src/models/generator.py:160-172
 * n_fft=400, hop=100                               src/evaluation.py:62,78
 * heads=4, dim_head=16, conv kernel 31             src/models/generator.py:75-90
 * max_pos_emb=512                                  src/models/conformer.py:76      */
typedef struct cmgan_config {
    int32_t n_fft;         /* 400 (16 kHz) or 1200 (48 kHz); multiple of 16, even  */
    int32_t hop;           /* 100 / 300; must divide n_fft                          */
    int32_t num_features;  /* F = n_fft/2 + 1                                       */
    int32_t num_channel;   /* 64 (the only value the kernels are specialised for)   */
    int32_t num_tscb;      /* 4                                                     */
    int32_t heads;         /* 4                                                     */
    int32_t dim_head;      /* 16                                                    */
    int32_t conv_kernel;   /* 31                                                    */
    int32_t max_pos_emb;   /* 512                                                   */
    int32_t mfma_mode;     /* CMGAN_MFMA_F32, CMGAN_MFMA_F16X3 (default), _F16X1 or _F16MIX */
    int32_t single_mask;   /* CMGAN_MFMA_F16MIX only: CMGAN_MIX_* bits = the kernel families that run ONE fp16 product */
} cmgan_config;

/* How the dense contractions (convs, linears, attention) are evaluated.  Three modes; all keep fp32 storage and
 * fp32 accumulation.  The first two are fp32-class and meet the 1e-3 parity gate by >2 decades; the third (F16X1,
 * below) is an opt-in reduced-precision mode that sits inside the gate without margin:
 *   F32   : v_mfma_f32_16x16x4_f32, bit-exact fp32 products (157 TF peak)
 *   F16X3 : every operand split x = hi + lo in fp16 and the product evaluated as
 *           hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 (~2^-21 relative product error,
 *           ~5x the fp32 matrix rate; gfx950 has no TF32).  At n_fft 400 / hop 100 the STFT / ISTFT
 *           use folded split-f16 DFT kernels too; other sizes (48 kHz: 1200 / 300) run the
 *           dense fp32 DFT kernels in either mode.                                          */
#define CMGAN_MFMA_F32   0
#define CMGAN_MFMA_F16X3 1
/* F16X1 - REDUCED precision, opt-in, never the default: the TSCNet body (dense convs, conformers) on ONE fp16 product
 * per contraction (operands rounded to nearest fp16, fp32 accumulation): the "half-precision" throughput mode
 * BASELINE configs[1] names (fp16 carries 3 more mantissa bits than bf16).  ~1.3x the F16X3 rate; error vs the fp32
 * reference 6e-4 .. 9e-4 of the peak on synthetic clips and real recordings alike
 * (tests/test_gpu_parity.py::test_f16x1_mode_error_bands pins the measured bands): inside the 1e-3 gate without
 * margin, 200x the error of the other two modes - not fp32-class.  STFT / ISTFT run as in F16X3 (|X|^-0.7 on
 * near-silent bins is what a single-product front end gets wrong by several 1e-2). */
#define CMGAN_MFMA_F16X1 2
/* F16MIX - REDUCED precision, opt-in: the F16X3 library with the kernel families named in cmgan_config.single_mask
 * switched to their single-product (F16X1) build, every other family on three split products.  What each family
 * costs in accuracy when it alone is single is tabulated in DESIGN.md (tools/mix_ablation.py); the shipped "f16mix"
 * preset of the Python host (cmgan_amd.engine.F16MIX_PRESET) takes every family whose own contribution stays below
 * 1e-4 of the peak, for an end-to-end error <= 2e-4 (5 x margin to the 1e-3 gate, asserted two-sidedly by
 * tests/test_gpu_parity.py::test_f16mix_mode_error_band).  single_mask = 0 is F16X3, all bits = F16X1.            */
#define CMGAN_MFMA_F16MIX 3
#define CMGAN_MIX_CONV   1    /* dilated dense convs, conv_2, sub-pixel convs   src/models/generator.py:39-47,60,107-118 */
#define CMGAN_MIX_FF1    2    /* ff1 (LN -> W1 -> Swish -> W2)                  src/models/conformer.py:136-148, 216     */
#define CMGAN_MIX_FF2    4    /* ff2 + post norm                                src/models/conformer.py:220-221          */
#define CMGAN_MIX_QKV    8    /* LN -> to_q / to_kv                             src/models/conformer.py:100-101          */
#define CMGAN_MIX_ATTN  16    /* q k^T, q E^T, P V, to_out                      src/models/conformer.py:103-133          */
#define CMGAN_MIX_PW1   32    /* conv module: LN -> pointwise 64 -> 256 -> GLU  src/models/conformer.py:161-164          */
#define CMGAN_MIX_DWPW2 64    /* depthwise k = 31 -> Swish -> pointwise 128 -> 64   src/models/conformer.py:165-170      */
#define CMGAN_MIX_ALL  127

/* Fills *cfg with the reference's 16 kHz defaults (above). */
void cmgan_default_config(cmgan_config* cfg);

int  cmgan_abi_version(void);

/* Creates a handle on the current device; builds the window / DFT tables
 * (replaces the torch.hamming_window + cuFFT plan of src/evaluation.py:36-38,44-50). */
int  cmgan_create(cmgan_handle** out, const cmgan_config* cfg);
void cmgan_destroy(cmgan_handle* h);

/* Last error message for this handle (or for a failed cmgan_create when h == NULL). */
const char* cmgan_last_error(const cmgan_handle* h);

/* Uploads packed weights.  `blob` is the host buffer produced by
 * cmgan_amd.packer.pack_state_dict() from the reference generator state_dict
 * (the 359-entry dict loaded at src/evaluation.py:63-64).  Layout: see
 * cmgan_amd/csrc/weights.h.  May be called again to swap weights (synchronises the
 * device).  The swap is atomic: on any failure the previously loaded weights stay in
 * place.  Device buffers are re-allocated, so hipGraphs captured before the call hold
 * stale pointers: re-capture when cmgan_weights_generation() has changed.          */
int  cmgan_load_weights(cmgan_handle* h, const void* blob, size_t bytes);
/* Number of successful cmgan_load_weights calls on this handle (-1 for NULL).      */
int  cmgan_weights_generation(const cmgan_handle* h);

/* Bytes of scratch the forward calls need for a batch of B spectrograms of T
 * frames (0 on bad arguments).  The caller allocates it once (256-byte aligned). */
size_t cmgan_workspace_bytes(const cmgan_handle* h, int B, int T);

/* c[b] = sqrt(L / sum_l wav[b,l]^2)            src/evaluation.py:21, src/train.py:75-79 */
int cmgan_rms_scale(cmgan_handle* h, const float* wav_dev, int B, int L,
                    float* scale_dev, void* stream);

/* Frames of an L-sample row: L / hop + 1 (torch.stft, center=True). */
int cmgan_num_frames(const cmgan_handle* h, int L);

/* wav[B,L] (optionally scaled by scale_dev[B]; NULL = 1) -> power-compressed
 * spectrogram spec[B,2,T,F] = the tensor fed to TSCNet.forward.
 * Replaces torch.stft + utils.power_compress + permute
 * (src/evaluation.py:36-39, src/utils.py:20-29).  Any L > n_fft/2 (reflect padding):
 * T = L / hop + 1 frames, exactly torch.stft(center=True).                      */
int cmgan_stft_compress(cmgan_handle* h, const float* wav_dev, const float* scale_dev,
                        int B, int L, float* spec_dev, void* stream);

/* TSCNet.forward (src/models/generator.py:174-196):
 * spec[B,2,T,F] -> out_real[B,1,T,F], out_imag[B,1,T,F].                       */
int cmgan_tscnet_forward(cmgan_handle* h, const float* spec_dev, int B, int T,
                         float* out_real_dev, float* out_imag_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* est_real/est_imag [B,1,T,F] -> waveform wav_out[B, hop*(T-1)], divided by
 * scale_dev[B] when it is not NULL.  Replaces permute + utils.power_uncompress +
 * torch.istft + '/ c'  (src/evaluation.py:41-51, src/utils.py:32-39).          */
int cmgan_uncompress_istft(cmgan_handle* h, const float* real_dev, const float* imag_dev,
                           const float* scale_dev, int B, int T, float* wav_out_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* The whole device pipeline of enhance_one_track (src/evaluation.py:21-51) for a
 * batch of equal-length rows: per-row RMS scale, STFT, TSCNet, ISTFT, un-scale.
 * wav[B,L] -> wav_out[B,L];  L % hop == 0.                                      */
int cmgan_enhance(cmgan_handle* h, const float* wav_dev, int B, int L, float* wav_out_dev,
                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* cmgan_enhance as `branches` (1..8) part-batch branches: rows split evenly, branch 0 on `stream`,
 * the others on streams the handle owns, forked and joined with events so that a stream capture
 * of `stream` records them as parallel paths of one hipGraph.  Branch i + 1 starts once branch i
 * has issued `offset_launches` kernels (0 = all together).  Utterances are independent in eval
 * mode (src/models/generator.py:35, src/models/conformer.py:168; src/evaluation.py:30-34 batches
 * rows the same way) and each branch runs the unchanged per-row arithmetic: the result equals
 * cmgan_enhance bit for bit.  Workspace: cmgan_workspace_bytes_branched(B, T, branches).
 * The first call creates the side streams and events - make it outside a capture.            */
size_t cmgan_workspace_bytes_branched(const cmgan_handle* h, int B, int T, int branches);
int cmgan_enhance_branched(cmgan_handle* h, const float* wav_dev, int B, int L, float* wav_out_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream,
                           int branches, int offset_launches);

/* The non-adversarial terms of Trainer.calculate_generator_loss (src/train.py:124-151)
 * as deterministic device reductions, out4_dev = {loss_ri, loss_mag, time_loss, time_mse}:
 *   loss_ri   = mse(est_real, clean_real) + mse(est_imag, clean_imag)      train.py:135-137
 *   loss_mag  = mse(|est|, |clean|)                                         train.py:132-134
 *   time_loss = mean |est_audio - clean_audio|                              train.py:139-141
 *   time_mse  = mean (est_audio - clean_audio)^2   (logging only, not in the reference)
 * est_real/est_imag [B,1,T,F] are the TSCNet outputs, clean_spec [B,2,T,F] the
 * power-compressed clean spectrogram in the model layout (cmgan_stft_compress of the
 * clean rows; the reference's permutes do not change an elementwise mean), est/clean
 * audio [B,L_audio].  Either group may be all-NULL (its terms are then 0).  These are
 * the per-rank scalars the data-parallel step all-reduces (one RCCL call).          */
int cmgan_loss_terms(cmgan_handle* h, const float* est_real_dev, const float* est_imag_dev,
                     const float* clean_spec_dev, int B, int T, const float* est_audio_dev,
                     const float* clean_audio_dev, int L_audio, float* out4_dev, void* stream);

/* Training-mode FeedForward branch of a ConformerBlock with its backward - the first slice of the training
 * step:  y = Scale(0.5, PreNorm(64, FeedForward(64, mult=4, dropout)))(x)
 *                         = 0.5 * m2 * (W2 (m1 * Swish(W1 LayerNorm(x) + b1)) + b2)
 * (src/models/conformer.py:54-72, 136-148, 211-212).  The residual add of :216/:219 is fused on request:
 * residual_dev [M,64] (NULL = none) is added to y in the final store, dresidual_dev [M,64] (NULL = none; usually dy
 * itself) to dx, so that `x = ff(x) + x` and its backward cost no extra pass over the sequence.  The same pair of
 * optional pointers exists on the conv-module and attention entry points below.
 * Parameters are the RAW tensors of the reference state_dict for one ff{1,2} branch (row-major, device):
 *   ln_weight/ln_bias [64] = ff.fn.norm.{weight,bias};  w1 [256,64], b1 [256] = ff.fn.fn.net.0;
 *   w2 [64,256], b2 [64] = ff.fn.fn.net.3.
 * x, y, dy, dx are [M,64].  mask1 [M,256] / mask2 [M,64] are the keep-masks of the two nn.Dropout layers as BYTES
 * (non-zero = keep; a kept value is multiplied by mask_scale = 1/(1-p)); NULL = no dropout (p = 0 or eval) - bytes
 * because the masks are the largest per-token traffic of the step.  The backward recomputes the hidden activations from x,
 * writes dL/dx to dx and dL/dparam to the six tensors of *grads (overwritten, not accumulated), with
 * fixed-order reductions (bit-reproducible).  Workspace: cmgan_ffn_train_workspace_bytes(h, M).            */
typedef struct cmgan_ffn_params {
    float *ln_weight, *ln_bias, *w1, *b1, *w2, *b2;
} cmgan_ffn_params;
size_t cmgan_ffn_train_workspace_bytes(const cmgan_handle* h, long long M);
int cmgan_ffn_train_forward(cmgan_handle* h, const float* x_dev, long long M, const cmgan_ffn_params* params,
                            const unsigned char* mask1_dev, const unsigned char* mask2_dev, float mask_scale,
                            const float* residual_dev, float* y_dev, void* workspace_dev, size_t workspace_bytes,
                            void* stream);
int cmgan_ffn_train_backward(cmgan_handle* h, const float* x_dev, const float* dy_dev, long long M,
                             const cmgan_ffn_params* params, const unsigned char* mask1_dev,
                             const unsigned char* mask2_dev, float mask_scale, const float* dresidual_dev,
                             float* dx_dev, const cmgan_ffn_params* grads,
                             void* workspace_dev, size_t workspace_bytes, void* stream);

/* Training-mode ConformerConvModule with its backward - second slice of the training step:
 *   y = Conv1d(128,64,1)(Swish(BatchNorm1d(DepthWiseConv1d_31(GLU(Conv1d(64,256,1)(LayerNorm(x)))))))
 * (src/models/conformer.py:151-176; residual_dev / dresidual_dev fuse the residual add of :218; the module's Dropout has
 * p = conv_dropout = 0).  TRAIN semantics: BatchNorm1d normalises with the statistics of this batch (biased
 * variance over all N*L positions, eps 1e-5) and updates running_mean / running_var in place (momentum 0.1, unbiased
 * variance) when both pointers are non-NULL.  x, y, dy, dx: contiguous sequences [N, L, 64].  Parameters are the
 * RAW tensors of the reference state_dict (conv.net.{0,2,4.conv,5,7}.{weight,bias}; the [256,64,1], [128,1,31] and
 * [64,128,1] conv weights are used as [256,64], [128,31], [64,128]).  The forward keeps its GLU output, depthwise
 * output and batch statistics in the workspace; the backward must be given the SAME workspace, untouched in
 * between, and writes dL/dx and the ten parameter gradients (overwritten, fixed-order reductions).               */
typedef struct cmgan_convmod_params {
    float *ln_weight, *ln_bias, *pw1_weight, *pw1_bias, *dw_weight, *dw_bias, *bn_weight, *bn_bias,
          *pw2_weight, *pw2_bias;
} cmgan_convmod_params;
size_t cmgan_convmod_train_workspace_bytes(const cmgan_handle* h, int N, int L);
int cmgan_convmod_train_forward(cmgan_handle* h, const float* x_dev, int N, int L, const cmgan_convmod_params* params,
                                float* running_mean_dev, float* running_var_dev, const float* residual_dev,
                                float* y_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_convmod_train_backward(cmgan_handle* h, const float* x_dev, const float* dy_dev, int N, int L,
                                 const cmgan_convmod_params* params, const float* dresidual_dev, float* dx_dev,
                                 const cmgan_convmod_params* grads,
                                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* Training-mode PreNorm(Attention) with its backward - third slice of the training step:
 *   y = mask * to_out(softmax((q k^T + q E[clamp(i - j, +-max_pos)]^T) / 4) v),  q = to_q(LN(x)), k|v = to_kv(LN(x))
 * (src/models/conformer.py:54-72, 75-133; 4 heads of 16; residual_dev / dresidual_dev fuse the residual add of :217).  `mask`
 * [N,L,64] is the byte keep-mask (non-zero = keep, kept values x mask_scale) of the nn.Dropout on the to_out output
 * (conformer.py:133; NULL = none).  Parameters are
 * the RAW tensors attn.norm.{weight,bias}, attn.fn.to_q.weight [64,64], attn.fn.to_kv.weight [128,64],
 * attn.fn.to_out.{weight [64,64], bias}, attn.fn.rel_pos_emb.weight [2 max_pos + 1, 16].  L <= 4096
 * (CMGAN_E_UNSUPPORTED beyond; distances past +-max_pos share the table's end rows, as in the reference).  The forward keeps q|k|v, the attention output and the row log-sum-exp in the
 * workspace; the backward needs the SAME workspace untouched and writes dL/dx and the seven parameter gradients
 * (the embedding-table gradient is dense [2 max_pos + 1, 16], rows of unused distances are zero).                 */
typedef struct cmgan_attn_params {
    float *ln_weight, *ln_bias, *to_q_weight, *to_kv_weight, *to_out_weight, *to_out_bias, *rel_pos_emb;
} cmgan_attn_params;
size_t cmgan_attn_train_workspace_bytes(const cmgan_handle* h, int N, int L);
int cmgan_attn_train_forward(cmgan_handle* h, const float* x_dev, int N, int L, const cmgan_attn_params* params,
                             const unsigned char* mask_dev, float mask_scale, const float* residual_dev,
                             float* y_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_attn_train_backward(cmgan_handle* h, const float* x_dev, const float* dy_dev, int N, int L,
                              const cmgan_attn_params* params, const unsigned char* mask_dev, float mask_scale,
                              const float* dresidual_dev, float* dx_dev, const cmgan_attn_params* grads,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/* Glue of ConformerBlock.forward in train mode (src/models/conformer.py:216-222): out = a + b over n floats (the
 * residual adds; n % 4 == 0), and the closing post_norm = nn.LayerNorm(64) (eps 1e-5) on [M,64] rows with its backward
 * (dL/dx, dL/dweight, dL/dbias; fixed-order reductions).                                                          */
/* in [B, A, C, 64] -> out [B, C, A, 64] (out != in): the time-axis <-> frequency-axis layout flip of a TSCB on
 * channels-last activations (src/models/generator.py:94,96 permute + contiguous).  add_dev (optional, laid out like
 * in_dev; NULL = none) is summed in before the flip: out = flip(in + add), which carries the residual add of
 * generator.py:95 (and of the backward) without a pass of its own.  cmgan_layernorm_train_forward's residual_dev
 * [M,64] (NULL = none) does the same for generator.py:97: y = LayerNorm(x) + residual.                            */
int cmgan_swap_axes(cmgan_handle* h, const float* in_dev, const float* add_dev, float* out_dev, int B, int A, int C,
                    void* stream);
int cmgan_add(cmgan_handle* h, const float* a_dev, const float* b_dev, float* out_dev, long long n, void* stream);
/* Keep-masks of nn.Dropout(p) layers (src/models/conformer.py:64,70,98,172) as bytes, 1 = keep with probability
 * keep_prob = 1 - p (compared at 16-bit resolution), for the mask_dev arguments of the training functions above: the
 * reference draws them inside F.dropout with torch's Philox stream, here one launch fills ALL masks of a step.
 * state_dev: two 64-bit words in device memory {seed, offset}; Philox4x32-10 counters offset .. offset + nbytes / 16 are
 * consumed and the offset is advanced on the device, so the call is replayable inside a captured graph.  nbytes must be
 * a positive multiple of 16.                                                                                        */
int cmgan_dropout_masks(cmgan_handle* h, unsigned char* masks_dev, long long nbytes, float keep_prob,
                        unsigned long long* state_dev, void* stream);
size_t cmgan_layernorm_train_workspace_bytes(const cmgan_handle* h, long long M);
int cmgan_layernorm_train_forward(cmgan_handle* h, const float* x_dev, long long M, const float* weight_dev,
                                  const float* bias_dev, const float* residual_dev, float* y_dev, void* stream);
int cmgan_layernorm_train_backward(cmgan_handle* h, const float* x_dev, const float* dy_dev, long long M,
                                   const float* weight_dev, const float* bias_dev, float* dx_dev,
                                   float* dweight_dev, float* dbias_dev,
                                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* Training-mode DilatedDenseNet with its backward - fourth slice of the training step:
 * four layers of  pad(top = 2^i, l/r = 1) -> Conv2d(64 (i+1) -> 64, (2,3), dilation (2^i, 1)) -> InstanceNorm2d(affine)
 * -> PReLU(64) -> cat([out, skip]) (newest first), returning the last layer's output (src/models/generator.py:6-47;
 * used by the encoder and both decoders).  Activations are channels-last [B, T, F, 64] (the reference's NCHW tensor
 * is x.permute(0, 2, 3, 1)).  Parameters are the RAW tensors conv{i}.weight [64, 64 i, 2, 3], conv{i}.bias,
 * norm{i}.weight/bias, prelu{i}.weight for i = 1..4 (index i-1 below).  The forward keeps the layer outputs, the raw
 * conv outputs and the InstanceNorm statistics in the workspace; the backward needs the SAME workspace untouched and
 * writes dL/dx and the twenty parameter gradients (the conv biases sit in front of an InstanceNorm, so their
 * gradients are zero up to rounding).  dx_dev is accumulated in place while the backward runs: it must not alias
 * x_dev or dy_dev (dy_dev is read, never written).                                                               */
typedef struct cmgan_dense_params {
    float *conv_weight[4], *conv_bias[4], *norm_weight[4], *norm_bias[4], *prelu_weight[4];
} cmgan_dense_params;
size_t cmgan_dense_train_workspace_bytes(const cmgan_handle* h, int B, int T, int F);
int cmgan_dense_train_forward(cmgan_handle* h, const float* x_dev, int B, int T, int F,
                              const cmgan_dense_params* params, float* y_dev,
                              void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_dense_train_backward(cmgan_handle* h, const float* x_dev, const float* dy_dev, int B, int T, int F,
                               const cmgan_dense_params* params, float* dx_dev, const cmgan_dense_params* grads,
                               void* workspace_dev, size_t workspace_bytes, void* stream);

/* Training-mode DenseEncoder with its backward (src/models/generator.py:50-69): conv_1 (1x1, 3 -> 64) +
 * InstanceNorm2d(affine) + PReLU(64), the DilatedDenseNet above, conv_2 ((1,3), stride (1,2), padding (0,1)) +
 * InstanceNorm2d(affine) + PReLU(64).  xin [B,T,F,3] is the channels-last x_in of TSCNet.forward
 * (cmgan_tscnet_prologue); y [B,T,F',64] with F' = (F - 1) / 2 + 1.  x_in carries no gradient, so the backward only
 * writes the thirty parameter gradients.  Same workspace contract as the dense block.                            */
typedef struct cmgan_encoder_params {
    float *conv1_weight, *conv1_bias, *norm1_weight, *norm1_bias, *prelu1_weight;   /* conv_1.{0.weight [64,3,1,1], 0.bias, 1.weight, 1.bias, 2.weight} */
    cmgan_dense_params dense;                                                       /* dilated_dense.* */
    float *conv2_weight, *conv2_bias, *norm2_weight, *norm2_bias, *prelu2_weight;   /* conv_2.{0.weight [64,64,1,3], ...} */
} cmgan_encoder_params;
size_t cmgan_encoder_train_workspace_bytes(const cmgan_handle* h, int B, int T, int F);
int cmgan_encoder_train_forward(cmgan_handle* h, const float* xin_dev, int B, int T, int F,
                                const cmgan_encoder_params* params, float* y_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_encoder_train_backward(cmgan_handle* h, const float* xin_dev, const float* dy_dev, int B, int T, int F,
                                 const cmgan_encoder_params* params, const cmgan_encoder_params* grads,
                                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* Training-mode MaskDecoder (kind CMGAN_DECODER_MASK, generator.py:121-138) and ComplexDecoder (kind
 * CMGAN_DECODER_COMPLEX, generator.py:141-156) with their backward.  Both: DilatedDenseNet -> SPConvTranspose2d
 * (pad 1/1, Conv2d(64 -> 128, (1,3)), pixel shuffle x2 along frequency, generator.py:102-119) -> head.
 *   mask head:    conv_1 (1,2) 64 -> 1, InstanceNorm2d(1, affine), PReLU(1), final_conv 1x1, PReLU(num_features);
 *                 out [B,T,F] (= the reference's [B,1,T,F])
 *   complex head: InstanceNorm2d(64, affine), PReLU(64), conv (1,2) 64 -> 2;  out [B,T,F,2] (reference: [B,2,T,F])
 * x [B,T,Fe,64] channels-last, F = 2 Fe - 1.  final_* / prelu_out_weight are ignored for the complex head.        */
enum { CMGAN_DECODER_MASK = 0, CMGAN_DECODER_COMPLEX = 1 };
typedef struct cmgan_decoder_params {
    cmgan_dense_params dense;                         /* dense_block.* */
    float *sub_pixel_weight, *sub_pixel_bias;         /* sub_pixel.conv.{weight [128,64,1,3], bias [128]} */
    float *conv_weight, *conv_bias;                   /* conv_1 (mask) / conv (complex): [NO,64,1,2], [NO] */
    float *norm_weight, *norm_bias, *prelu_weight;    /* norm.{weight,bias}, prelu.weight: [1] (mask) or [64] */
    float *final_weight, *final_bias;                 /* mask only: final_conv.{weight [1,1,1,1], bias [1]} */
    float *prelu_out_weight;                          /* mask only: prelu_out.weight [num_features] */
} cmgan_decoder_params;
size_t cmgan_decoder_train_workspace_bytes(const cmgan_handle* h, int B, int T, int Fe);
int cmgan_decoder_train_forward(cmgan_handle* h, int kind, const float* x_dev, int B, int T, int Fe,
                                const cmgan_decoder_params* params, float* out_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_decoder_train_backward(cmgan_handle* h, int kind, const float* x_dev, const float* dout_dev, int B, int T,
                                 int Fe, const cmgan_decoder_params* params, float* dx_dev,
                                 const cmgan_decoder_params* grads,
                                 void* workspace_dev, size_t workspace_bytes, void* stream);

/* The glue of TSCNet.forward (generator.py:176-201) for the training step, F = num_features of the handle:
 *   prologue:  spec [B,2,T,F] -> xin [B,T,F,3] = (|spec|, re, im)                       generator.py:177-181
 *   epilogue:  est = mask * |spec| * (cos, sin)(angle spec) + complex_out = mask * spec + complex_out
 *              (mask [B,T,F], complex_out [B,T,F,2] -> est_real, est_imag [B,T,F])      generator.py:188-199
 *   backward:  dmask = d_real re + d_imag im,  dcomplex = (d_real, d_imag); the noisy spectrogram is data.       */
int cmgan_tscnet_prologue(cmgan_handle* h, const float* spec_dev, int B, int T, float* xin_dev, void* stream);
int cmgan_tscnet_epilogue_forward(cmgan_handle* h, const float* spec_dev, const float* mask_dev,
                                  const float* complex_dev, int B, int T, float* est_real_dev, float* est_imag_dev,
                                  void* stream);
int cmgan_tscnet_epilogue_backward(cmgan_handle* h, const float* spec_dev, const float* d_real_dev,
                                   const float* d_imag_dev, int B, int T, float* dmask_dev, float* dcomplex_dev,
                                   void* stream);

/* Gradient of  w_ri * loss_ri + w_mag * loss_mag + w_time * time_loss  (the non-adversarial part of
 * Trainer.calculate_generator_loss, src/train.py:133-148; same arguments as cmgan_loss_terms) with respect to
 * est_real / est_imag [B,T,F].  The time term runs the adjoint of torch.istft (window, overlap-add, envelope
 * division, centre trim; train.py:106-112) and of utils.power_uncompress (utils.py:32-39) in one kernel; est_audio /
 * clean_audio [B, hop (T-1)] may both be NULL when w_time = 0.  n_fft / hop / F are the handle's.                 */
int cmgan_loss_backward(cmgan_handle* h, const float* est_real_dev, const float* est_imag_dev,
                        const float* clean_spec_dev, int B, int T, const float* est_audio_dev,
                        const float* clean_audio_dev, float w_ri, float w_mag, float w_time,
                        float* d_real_dev, float* d_imag_dev, void* stream);

/* The metric discriminator Discriminator(ndf=16) of the reference trainer (src/models/discriminator.py:29-64) with its
 * backward: four spectral-norm Conv2d(4x4, stride 2, pad 1, no bias) + InstanceNorm2d(affine) + PReLU stages
 * (2 -> 16 -> 32 -> 64 -> 128), global max pool, spectral-norm Linear(128,64), Dropout(0.3), PReLU(64), spectral-norm
 * Linear(64,1), LearnableSigmoid(1).  Parameters are the RAW state_dict tensors (weight_orig + the power-iteration
 * buffers weight_u / weight_v of torch.nn.utils.spectral_norm).  xy [B,T,F,2] = (|clean|, |est|) channels-last is the
 * reference's cat([clean_mag, est_mag], 1) [B,2,F,T] (cmgan_mag_pair builds it from the model-layout tensors);
 * T, F >= 16.  update_uv != 0 is train mode: one power iteration per spectral norm before use, written back to the
 * u / v buffers; mask [B,64] is the Dropout keep-mask (0 or 1/0.7) or NULL.  The backward needs the forward's workspace
 * untouched (it holds the u, v, sigma that forward used), writes dL/dxy when dxy is not NULL and the 22 parameter
 * gradients (the u / v fields of `grads` are ignored).                                                            */
typedef struct cmgan_disc_params {
    float *conv_weight_orig[4], *conv_u[4], *conv_v[4];   /* layers.{0,3,6,9}.weight_orig / weight_u / weight_v */
    float *norm_weight[4], *norm_bias[4], *prelu_weight[4]; /* layers.{1,4,7,10}.{weight,bias}, layers.{2,5,8,11}.weight */
    float *fc1_weight_orig, *fc1_bias, *fc1_u, *fc1_v;    /* layers.14.* */
    float *prelu5_weight;                                 /* layers.16.weight */
    float *fc2_weight_orig, *fc2_bias, *fc2_u, *fc2_v;    /* layers.17.* */
    float *slope;                                         /* layers.18.slope */
} cmgan_disc_params;
size_t cmgan_disc_workspace_bytes(const cmgan_handle* h, int B, int T);
int cmgan_disc_forward(cmgan_handle* h, const float* xy_dev, int B, int T, const cmgan_disc_params* params,
                       const float* mask_dev, int update_uv, float* score_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_disc_backward(cmgan_handle* h, const float* xy_dev, const float* dscore_dev, int B, int T,
                        const cmgan_disc_params* params, const float* mask_dev, float* dxy_dev,
                        const cmgan_disc_params* grads, void* workspace_dev, size_t workspace_bytes, void* stream);
/* xy [B,T,F,2] = (|clean_spec|, |est|) (train.py:102-103); est_* NULL -> (|clean|, |clean|) (train.py:166).
 * Backward: d_real / d_imag += scale * dxy[...,1] * est / |est|  (the path of gen_loss_GAN into the generator).   */
int cmgan_mag_pair(cmgan_handle* h, const float* clean_spec_dev, const float* est_real_dev, const float* est_imag_dev,
                   int B, int T, float* xy_dev, void* stream);
int cmgan_mag_pair_backward(cmgan_handle* h, const float* est_real_dev, const float* est_imag_dev,
                            const float* dxy_dev, int B, int T, float scale, float* d_real_dev, float* d_imag_dev,
                            void* stream);
/* loss = mean((score - target)^2) with target = 1 when NULL (train.py:129-131, 168-170); dscore (may be NULL) =
 * scale * d loss / d score.                                                                                       */
int cmgan_score_mse(cmgan_handle* h, const float* score_dev, const float* target_dev, int B, float scale,
                    float* loss_dev, float* dscore_dev, void* stream);

/* One torch.optim.AdamW step (src/train.py:63-66, 192-193; defaults betas (0.9, 0.999), eps 1e-8, weight_decay
 * 0.01) over a FLAT fp32 bucket of n parameters: params, grads and the two moment buffers are parallel device
 * arrays (the bucket the gradient all-reduce runs over), `step` = 1, 2, ... is the update count for the bias
 * corrections, `lr` the scheduled learning rate (StepLR(30, 0.5) is host arithmetic, train.py:248-253).          */
int cmgan_adamw_step(cmgan_handle* h, float* params_dev, const float* grads_dev, float* exp_avg_dev,
                     float* exp_avg_sq_dev, long long n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, void* stream);

/* The same update with the optimiser's scalars in device memory: state_dev[0] = learning rate, state_dev[1] = update
 * count (a float, advanced by one by this call before it is used for the bias corrections).  Nothing in the launch
 * depends on host values that change from step to step, so it can be captured in a hipGraph and replayed; a learning
 * rate schedule writes state_dev[0] between replays.                                                               */
int cmgan_adamw_step_dev(cmgan_handle* h, float* params_dev, const float* grads_dev, float* exp_avg_dev,
                         float* exp_avg_sq_dev, long long n, float* state_dev, float beta1, float beta2, float eps,
                         float weight_decay, void* stream);

/* utils.power_compress (src/utils.py:20-29): x[B,F,T,2] -> y[B,2,F,T].          */
int cmgan_power_compress(cmgan_handle* h, const float* x_dev, int B, int F, int T,
                         float* y_dev, void* stream);
/* utils.power_uncompress (src/utils.py:32-39): real,imag[B,1,F,T] -> y[B,1,F,T,2]. */
int cmgan_power_uncompress(cmgan_handle* h, const float* real_dev, const float* imag_dev,
                           int B, int F, int T, float* y_dev, void* stream);

/* ConformerBlock.forward (src/models/conformer.py:216-222), eval mode, for the
 * conformer stored in slot `index` of the loaded weights (2*(k-1) = TSCB_k.time,
 * 2*(k-1)+1 = TSCB_k.freq; a standalone block is packed into slot 0).
 * x[N,L,64] contiguous -> y[N,L,64].  `taps_dev` (may be NULL) receives the
 * residual stream after ff1 / attn / conv / ff2 as 4 consecutive [N,L,64] tensors
 * (test hook).  Workspace: cmgan_conformer_workspace_bytes(h, N, L).            */
size_t cmgan_conformer_workspace_bytes(const cmgan_handle* h, int N, int L);
int cmgan_conformer_forward(cmgan_handle* h, int index, const float* x_dev, int N, int L,
                            float* y_dev, float* taps_dev,
                            void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same with the attention mask of ConformerBlock.forward(x, mask) (conformer.py:113-126, 217):
 * mask_dev [N, L] bytes, non-zero = keep.  A (query i, key j) pair keeps its score only when both are kept; every
 * other score is the reference's -finfo.max, i.e. kept queries ignore masked keys and a masked query attends
 * uniformly to all L keys.  Only the attention uses the mask (feed-forward, conv module and norms see every row),
 * exactly like the reference.  CMGAN itself never passes one (generator.py:95,97).                              */
int cmgan_conformer_forward_masked(cmgan_handle* h, int index, const float* x_dev, int N, int L,
                                   const unsigned char* mask_dev, float* y_dev, float* taps_dev,
                                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* Stage taps of TSCNet.forward for parity tests (NCHW like the reference so the
 * tests read like the reference's module outputs); any pointer may be NULL:
 *   encoder[B,64,T,F']  = dense_encoder(x_in)        generator.py:181
 *   tscb[4][B,64,T,F']  = TSCB_1..4 outputs          generator.py:182-185
 *   mask[B,1,T,F], complex_out[B,2,T,F]              generator.py:187,190
 * Runs the same kernels as cmgan_tscnet_forward.                                */
typedef struct cmgan_taps {
    float* encoder_dev;
    float* tscb_dev[4];
    float* mask_dev;
    float* complex_dev;
} cmgan_taps;
int cmgan_tscnet_forward_taps(cmgan_handle* h, const float* spec_dev, int B, int T,
                              float* out_real_dev, float* out_imag_dev, const cmgan_taps* taps,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- Streaming with carried state (BASELINE.json configs[4]) --------------------------
 * The reference has no streaming mode; the hooks it leaves are its already-causal dilated convs (generator.py:16-20,
 * 39-47) and the unused `causal` flag of the conv module (conformer.py:153,158,168).  What makes the encoder and the
 * decoders non-causal in the reference is ONLY InstanceNorm2d's statistics over all of T (generator.py:35,55,61,
 * 128,148).  With the statistics FROZEN (taken from a calibration pass and then held, like BatchNorm in eval mode) the
 * dense encoder and both decoders are exactly time-causal with a receptive field of 1 + 2 + 4 + 8 = 15 frames back:
 * frame t of their output is a function of frames t - 15 .. t of their input.  So their state CAN be carried across
 * windows exactly: 15 frames of input history in front of the new frames, first 15 output frames dropped.  The four
 * TSCBs attend over the whole window (bidirectional, no exact cache exists): they run on [context | window |
 * look-ahead] frames of cached / fresh encoder outputs, as before.
 *
 *   cmgan_stats_floats(h, B)        floats of one statistics blob: the (scale, shift) pairs of the 15 InstanceNorms
 *                                   per row + the mask head's one-channel norm.  Opaque; valid for that B.
 *   cmgan_tscnet_forward_stats      cmgan_tscnet_forward with `frozen_stats` (may be NULL = the tensor's own
 *                                   statistics, i.e. the reference's arithmetic) and `stats_out` (may be NULL): the blob
 *                                   of the statistics this call used.  A call with frozen_stats = its own stats_out
 *                                   reproduces the unfrozen call bit for bit.
 *   cmgan_stream_encoder            spec[B,2,T,F] -> x[B,T,F',64] (channels-last, the layout the TSCBs and decoders
 *                                   share) = dense_encoder (generator.py:65-69) under frozen statistics
 *   cmgan_stream_tscb               x[B,T,F',64] in place = TSCB_1..4 (generator.py:92-99, 182-185)
 *   cmgan_stream_decoder            x[B,T,F',64] + spec[B,2,T,F] -> est_real / est_imag [B,1,T,F] = mask + complex decoder
 *                                   + recombination (generator.py:187-194) under frozen statistics
 * Exactness (tests/test_gpu_parity.py::test_stream_*): frames t >= 15 of cmgan_stream_encoder on ANY slice of a clip's
 * spectrogram equal the same frames of the whole-clip call bit for bit; likewise the decoder.  Workspace:
 * cmgan_workspace_bytes(h, B, T) of the call's own T.                                                           */
size_t cmgan_stats_floats(const cmgan_handle* h, int B);
int cmgan_tscnet_forward_stats(cmgan_handle* h, const float* spec_dev, int B, int T, float* out_real_dev,
                               float* out_imag_dev, const float* frozen_stats_dev, float* stats_out_dev,
                               void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_stream_encoder(cmgan_handle* h, const float* spec_dev, int B, int T, const float* frozen_stats_dev,
                         float* x_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int cmgan_stream_tscb(cmgan_handle* h, float* x_dev, int B, int T, void* workspace_dev, size_t workspace_bytes,
                      void* stream);
int cmgan_stream_decoder(cmgan_handle* h, const float* x_dev, const float* spec_dev, int B, int T,
                         const float* frozen_stats_dev, float* out_real_dev, float* out_imag_dev,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* Self-test of the MFMA fragment conventions every kernel relies on: computes
 * D = A(16xK) * B(Kx16) with the f32 16x16x4 MFMA and the library's fragment
 * packing; returns max |D - reference| through *max_err_host (synchronises).   */
int cmgan_selftest_mfma(cmgan_handle* h, float* max_err_host);
/* Same for the f16 16x16x32 MFMA + split-product images of the F16X3 mode. */
int cmgan_selftest_mfma_x3(cmgan_handle* h, float* max_err_host);

/* Names + durations (ms) of the kernels of the most recent forward when
 * profiling is enabled with cmgan_set_profiling(h, 1): HIP events are recorded
 * around every launch on the caller's stream.  cmgan_profile_read synchronises
 * the events, writes up to `cap` entries and returns the count.  bench.py uses it
 * for the live per-kernel roofline figure.                                      */
typedef struct cmgan_kernel_time { const char* name; float ms; } cmgan_kernel_time;
int cmgan_set_profiling(cmgan_handle* h, int enabled);
int cmgan_profile_read(cmgan_handle* h, cmgan_kernel_time* out, int cap);

/* ---- Environment ---------------------------------------------------------------------------------------------------
 * Launch-shape / kernel-choice overrides, read ONCE per process on first use (same-session A/B sweeps,
 * tools/knob_sweep.sh).  None changes a result beyond the last bits: each selects between kernels / launch shapes that
 * are parity-tested on their own.  Values outside the range shown (or not a number) are ignored - the built-in default
 * applies, so no setting can produce an invalid launch (kernels.h, env_knob):
 *   CMGAN_FFN32=0|1            FeedForward on 32x32x16 MFMAs (1, default) or the 16x16x32 kernel (0)
 *   CMGAN_STFT_BSPLIT=1..13    bin blocks per thread block of the small-batch STFT
 *   CMGAN_ASP_TPB_LONG=1..64, CMGAN_ASP_GROUP_LONG=1..4096, CMGAN_ASP_GROUP_SHORT=1..4096, CMGAN_ASP_ALIGN_SHORT=0|1,
 *   CMGAN_ASP_SLOTS=8..65536, CMGAN_ASP_TAILK=0|1    tile order / block shape of the attention kernel
 *   CMGAN_STFT_FFT=0|1         front / back end as 16 x 25 real FFTs (1, default) or the folded DFT products (0)
 *   CMGAN_FFN_BWD_FUSED=0|1, CMGAN_CM_BWD1_FUSED=0|1, CMGAN_CM_BWD2_FUSED=0|1
 *                              training: FeedForward backward / conv-module backward parts 1 and 2 with their weight gradients
 *                              contracted on the chip (1, default) or the un-fused kernels + token-contraction launches (0)
 *   CMGAN_RC_FWD_X3=0|1, CMGAN_RC_WGRAD_X3=0|1, CMGAN_RC_DGRAD_X3=0|1
 *                              training: forward (through the inference kernels), weight gradient and data gradient of the
 *                              encoder's stride-2 conv and the decoders' sub-pixel conv on split-f16 products (1, default)
 *                              or the fp32-MFMA row-conv kernels (0)
 *   CMGAN_ATTN_BWD=cores       training: the three attention backward cores instead of the fused kernel (read per launch)
 *   CMGAN_ATF_SLOTS_SHORT=1|2  training: wrapped diagonals per wave of the fused attention backward at L <= 128 (2, default:
 *                              4-wave blocks, three per CU; 1: 7-wave blocks, one per CU)
 * (The Python host adds CMGAN_BRANCHES=1|2 and CMGAN_BRANCH_OFFSET=n for Engine.enhance_graphed; CMGAN_HIP_LIB selects
 * the library file.)                                                                                                  */

#ifdef __cplusplus
}
#endif
#endif /* CMGAN_HIP_H */
