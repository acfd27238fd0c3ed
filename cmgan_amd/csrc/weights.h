// weights.h - layout of the packed weight blob shared by the Python packer
// (cmgan_amd/packer.py) and the library (api.hip).  Host-only header.
//
// Blob = header | directory | fp32 payload
//   header    : u32 magic 'CMGB' (0x42474d43), u32 version, u32 n_entries, u32 payload_floats
//   directory : n_entries x { u32 id, u32 offset_floats, u32 count_floats, u32 reserved }
//   payload   : float32[payload_floats]; every entry offset is a multiple of 64 floats
//
// id = group * 64 + item.  "fm" = MFMA fragment-major: a matrix M[R][K] (R, K
// multiples of 16) is stored as [R/16][K/16][64 lanes][4] with
//   fm[rb][kb][lane][r] = M[16*rb + (lane & 15)][16*kb + 4*(lane >> 4) + r]
// which is exactly the per-lane float4 a wave feeds to v_mfma_f32_16x16x4_f32 for
// the four k-steps r = 0..3 (k-step r consumes k = 16*kb + 4*g + r on lane group g),
// so every operand fetch is one fully coalesced 1 KiB load.
#pragma once
#include <stdint.h>

#define CMGAN_BLOB_MAGIC   0x42474d43u
#define CMGAN_BLOB_VERSION 1u

enum {
    G_ENC   = 0,   // dense_encoder conv_1 / conv_2
    G_DB_E  = 1,   // dense_encoder.dilated_dense
    G_DB_M  = 2,   // mask_decoder.dense_block
    G_DB_C  = 3,   // complex_decoder.dense_block
    G_MASK  = 4,   // mask_decoder tail
    G_CPLX  = 5,   // complex_decoder tail
    G_CONF0 = 8,   // conformers 0..7: TSCB_k.time -> 2(k-1), TSCB_k.freq -> 2(k-1)+1
    G_COUNT = 16
};
#define WID(group, item) ((group) * 64 + (item))

// G_ENC items
enum {
    ENC_C1_W = 0,     // [4][64]: w_mag, w_re, w_im, bias           generator.py:54
    ENC_C1_GB = 1,    // [2][64]: InstanceNorm gamma, beta            generator.py:55
    ENC_C1_PRELU = 2, // [64]                                         generator.py:56
    ENC_C2_W = 3,     // conv (1,3) stride (1,2): [4 chunks][3 taps][4 cb][64][4] fm   generator.py:60
    ENC_C2_BIAS = 4,  // [64]
    ENC_C2_GB = 5,    // [2][64]                                      generator.py:61
    ENC_C2_PRELU = 6  // [64]                                         generator.py:62
};
// G_DB_* items: layer i = 0..3 (reference conv{i+1}), dilation 2^i    generator.py:14-37
//   DB_W   : [4(i+1) chunks][6 taps = kt*3+kf][4 cb][64][4] fm; input channels in SLOT order
//            (slot 0 = block input, slot s = output of layer s), i.e. the reverse of the
//            reference's newest-first concat (generator.py:46)
#define DB_W(i)     ((i) * 4 + 0)
#define DB_BIAS(i)  ((i) * 4 + 1)   // [64]
#define DB_GB(i)    ((i) * 4 + 2)   // [2][64]
#define DB_PRELU(i) ((i) * 4 + 3)   // [64]
// G_MASK items                                                          generator.py:122-139
enum {
    MK_SP_W = 0,      // sub_pixel conv (1,3) 64->128: [4 chunks][3 taps][8 cb][64][4] fm
    MK_SP_BIAS = 1,   // [128]
    MK_TAIL_W = 2,    // conv_1 (1,2) 64->1 as fm [1 rb][4 kb][64][4]; row 0 = kf0, row 1 = kf1
    MK_SCALARS = 3,   // [8]: conv_1.bias, norm.weight, norm.bias, prelu, final_conv.w, final_conv.b, 0, 0
    MK_PRELU_OUT = 4  // [F]  per-frequency slopes                        generator.py:131
};
// G_CPLX items                                                          generator.py:142-156
enum {
    CX_SP_W = 0, CX_SP_BIAS = 1,
    CX_GB = 2,        // [2][64]
    CX_PRELU = 3,     // [64]
    CX_TAIL_W = 4,    // conv (1,2) 64->2 as fm [1][4][64][4]; rows: (o0,kf0),(o0,kf1),(o1,kf0),(o1,kf1)
    CX_BIAS = 5       // [2] (padded to 64 in the payload)
};
// conformer items                                                        conformer.py:75-222
enum {
    CF_FF1_W1 = 0,    // fm [16][4]   -log2(e) * Linear(64,256) with the PreNorm LayerNorm affine folded in
    CF_FF1_B1 = 1,    // [256]        -log2(e) * (b1 + W1 @ beta)      (Swish evaluated on h' = -log2(e) h)
    CF_FF1_W2 = 2,    // fm [4][16]   -ln2 * 0.5 * Linear(256,64)   (Scale(0.5), conformer.py:211)
    CF_FF1_B2 = 3,    // [64]         0.5 * b2
    CF_QKV_W = 4,     // fm [12][4]   rows 0..63 = 0.25*log2(e)*to_q, 64..191 = to_kv; attn LayerNorm folded
    CF_QKV_B = 5,     // [192]        W @ beta (the reference has no bias; this is the folded LN shift)
    CF_WO = 6,        // fm [4][4]    to_out
    CF_BO = 7,        // [64]
    CF_REL = 8,       // [2*max_pos+1][16]  rel_pos_emb                     conformer.py:86
    CF_PW1_W = 9,     // fm [16][4]   Conv1d(64,256,1) with the conv-module LayerNorm folded
    CF_PW1_B = 10,    // [256]
    CF_DW_W = 11,     // [31][128]    depthwise taps with BatchNorm1d(eval) folded, tap-major
    CF_DW_B = 12,     // [128]
    CF_PW2_W = 13,    // fm [4][8]    Conv1d(128,64,1)
    CF_PW2_B = 14,    // [64]
    CF_FF2_W1 = 15, CF_FF2_B1 = 16, CF_FF2_W2 = 17, CF_FF2_B2 = 18,
    CF_POST_GB = 19   // [2][64]      post_norm gamma, beta                  conformer.py:214
};
