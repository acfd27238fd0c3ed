// api.hip - C ABI of libcmgan_hip.so (see include/cmgan_hip.h): handle, weight upload,
// workspace plan and the launch sequences of the generator forward path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "api_internal.h"
#include "weights.h"

extern "C" void cmgan_default_config(cmgan_config* c) {
    if (!c) return;
    c->n_fft = 400; c->hop = 100; c->num_features = 201; c->num_channel = 64; c->num_tscb = 4;
    c->heads = 4; c->dim_head = 16; c->conv_kernel = 31; c->max_pos_emb = 512;
    c->mfma_mode = CMGAN_MFMA_F16X3;
    c->single_mask = 0;
}

extern "C" int cmgan_abi_version(void) { return CMGAN_ABI_VERSION; }

extern "C" const char* cmgan_last_error(const cmgan_handle* h) {
    return h ? h->err.c_str() : g_create_error.c_str();
}

// fm[rb][kb][lane][r] = M[16*rb + (lane&15)][16*kb + 4*(lane>>4) + r]   (weights.h)
static void pack_fm(const std::vector<double>& M, int R, int K, float* out) {
    const int RB = R / 16, KB = K / 16;
    for (int rb = 0; rb < RB; ++rb)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rb + (lane & 15), col = 16 * kb + 4 * (lane >> 4) + r;
                    out[(((size_t)rb * KB + kb) * 64 + lane) * 4 + r] = (float)M[(size_t)row * K + col];
                }
}

static void split_h(float v, _Float16& hi, _Float16& lo);

// Folded forward DFT images for stft_fold_x3_kernel (see kernels.h).  k-slot of lane group g, slot e in
// k32 block m is sample n = 32 m + 8 g + e; columns are the 16 bins of block bb.
static void build_fold_fwd(int N, int F, const std::vector<double>& win, std::vector<_Float16>& img) {
    const int FB = (F + 15) / 16, H = N / 2, M32 = (H + 1 + 31) / 32;
    const double PI2 = 6.283185307179586476925286766559;
    img.assign((size_t)FB * 2 * M32 * 1024, (_Float16)0.f);
    for (int bb = 0; bb < FB; ++bb)
        for (int cs = 0; cs < 2; ++cs)
            for (int m = 0; m < M32; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int k = 16 * bb + (lane & 15), n = 32 * m + 8 * (lane >> 4) + e;
                        double v = 0.0;
                        if (k < F && n <= H) {
                            const long mm = ((long)k * n) % N;
                            if (cs == 0) v = win[n] * cos(PI2 * mm / N);
                            else if (n >= 1 && n < H) v = -win[n] * sin(PI2 * mm / N);
                        }
                        _Float16 hi, lo;
                        split_h((float)v, hi, lo);
                        const size_t o = (((size_t)bb * 2 + cs) * M32 + m) * 1024 + lane * 8 + e;
                        img[o] = hi;
                        img[o + 512] = lo;
                    }
}

// Folded inverse DFT images for irfft_fold_x3_kernel: [13 sample blocks][cos | sin][7 k32 blocks][hi | lo][64][8];
// lane (c = sample n within the block, g) slot e of block m <-> bin k = 32 m + 8 g + e.
static void build_fold_inv(int N, const std::vector<double>& /*win*/, std::vector<_Float16>& img) {
    const int H = N / 2, NB = (H + 1 + 15) / 16, M32 = (H + 1 + 31) / 32;
    const double PI2 = 6.283185307179586476925286766559;
    img.assign((size_t)NB * 2 * M32 * 1024, (_Float16)0.f);
    for (int nb = 0; nb < NB; ++nb)
        for (int cs = 0; cs < 2; ++cs)
            for (int m = 0; m < M32; ++m)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int n = 16 * nb + (lane & 15), k = 32 * m + 8 * (lane >> 4) + e;
                        double v = 0.0;
                        if (n <= H && k <= H) {
                            const long mm = ((long)k * n) % N;
                            const double wk = (k == 0 || k == H) ? 1.0 : 2.0;
                            if (cs == 0) v = wk * cos(PI2 * mm / N) / N;
                            else if (k >= 1 && k < H) v = wk * sin(PI2 * mm / N) / N;
                        }
                        _Float16 hi, lo;
                        split_h((float)v, hi, lo);
                        const size_t o = (((size_t)nb * 2 + cs) * M32 + m) * 1024 + lane * 8 + e;
                        img[o] = hi;
                        img[o + 512] = lo;
                    }
}

extern "C" int cmgan_create(cmgan_handle** out, const cmgan_config* cfg) {
    if (!out || !cfg) return fail(nullptr, CMGAN_E_BADARG, "cmgan_create: null argument");
    *out = nullptr;
    if (cfg->num_channel != 64 || cfg->heads != 4 || cfg->dim_head != 16 || cfg->conv_kernel != 31)
        return fail(nullptr, CMGAN_E_UNSUPPORTED,
                    "kernels are specialised for num_channel=64, heads=4, dim_head=16, conv_kernel=31");
    if (cfg->n_fft <= 0 || cfg->n_fft % 16 || cfg->hop <= 0 || cfg->hop % 4 || cfg->n_fft % cfg->hop ||
        cfg->num_features != cfg->n_fft / 2 + 1 || (cfg->num_features & 1) == 0)
        return fail(nullptr, CMGAN_E_UNSUPPORTED,
                    "need n_fft %% 16 == 0, hop %% 4 == 0, hop | n_fft, num_features == n_fft/2+1 (odd)");
    if (cfg->num_tscb < 1 || cfg->num_tscb > 4 || cfg->max_pos_emb < 1)
        return fail(nullptr, CMGAN_E_UNSUPPORTED, "num_tscb must be 1..4, max_pos_emb >= 1");
    if (cfg->mfma_mode < CMGAN_MFMA_F32 || cfg->mfma_mode > CMGAN_MFMA_F16MIX)
        return fail(nullptr, CMGAN_E_UNSUPPORTED,
                    "mfma_mode must be CMGAN_MFMA_F32 (0), CMGAN_MFMA_F16X3 (1), CMGAN_MFMA_F16X1 (2) or CMGAN_MFMA_F16MIX (3)");
    if (cfg->mfma_mode == CMGAN_MFMA_F16MIX && (cfg->single_mask & ~CMGAN_MIX_ALL))
        return fail(nullptr, CMGAN_E_UNSUPPORTED, "single_mask has bits outside CMGAN_MIX_ALL");
    cmgan_handle* h = new cmgan_handle();
    h->cfg = *cfg;
    // one representation inside: F16X3 = no family single, F16X1 = all of them (the pure modes keep their own entry points)
    if (cfg->mfma_mode != CMGAN_MFMA_F16MIX) h->cfg.single_mask = cfg->mfma_mode == CMGAN_MFMA_F16X1 ? CMGAN_MIX_ALL : 0;
    hipError_t e = hipGetDevice(&h->device);
    if (e != hipSuccess) {
        fail(nullptr, CMGAN_E_HIP, "hipGetDevice: %s", hipGetErrorString(e));
        delete h;
        return CMGAN_E_HIP;
    }
    // ---- window + forward / inverse DFT matrices (host fp64 -> fp32, fragment-major) ----
    const int N = cfg->n_fft, F = cfg->num_features, FB = (F + 15) / 16;
    const double PI2 = 6.283185307179586476925286766559;
    std::vector<double> win(N);
    for (int n = 0; n < N; ++n) win[n] = 0.54 - 0.46 * cos(PI2 * n / N);   // torch.hamming_window(periodic)
    const int RF = 2 * FB * 16;                                             // rows: re bins | im bins
    std::vector<double> fwd((size_t)RF * N, 0.0), inv((size_t)N * RF, 0.0);
    for (int k = 0; k < F; ++k)
        for (int n = 0; n < N; ++n) {
            const long m = ((long)k * n) % N;
            const double cs = cos(PI2 * m / N), sn = sin(PI2 * m / N);
            fwd[(size_t)k * N + n] = win[n] * cs;
            fwd[(size_t)(FB * 16 + k) * N + n] = -win[n] * sn;
            const double wk = (k == 0 || k == N / 2) ? 1.0 : 2.0;
            inv[(size_t)n * RF + k] = win[n] * wk * cs / N;
            inv[(size_t)n * RF + FB * 16 + k] = (k == 0 || k == N / 2) ? 0.0 : -win[n] * wk * sn / N;
        }
    const size_t n_fwd = (size_t)RF * N, n_inv = (size_t)N * RF;
    std::vector<float> host(n_fwd + n_inv + N);
    pack_fm(fwd, RF, N, host.data());
    pack_fm(inv, N, RF, host.data() + n_fwd);
    for (int n = 0; n < N; ++n) host[n_fwd + n_inv + n] = (float)win[n];
    e = hipMalloc(&h->d_tables, host.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_tables, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fail(nullptr, CMGAN_E_HIP, "table upload: %s", hipGetErrorString(e));
        if (h->d_tables) hipFree(h->d_tables);
        delete h;
        return CMGAN_E_HIP;
    }
    e = hipMalloc(&h->d_loss, (size_t)LOSS_BLOCKS * 4 * sizeof(double));
    if (e != hipSuccess) {
        fail(nullptr, CMGAN_E_HIP, "loss scratch: %s", hipGetErrorString(e));
        hipFree(h->d_tables);
        delete h;
        return CMGAN_E_HIP;
    }
    h->st.n_fft = N; h->st.hop = cfg->hop; h->st.F = F; h->st.FB = FB;
    h->st.fwd_fm = h->d_tables; h->st.inv_fm = h->d_tables + n_fwd; h->st.window = h->d_tables + n_fwd + n_inv;
    h->st.fold_fwd16 = nullptr; h->st.fold_inv16 = nullptr;
    if (cfg->mfma_mode != CMGAN_MFMA_F32 && N == 400 && cfg->hop == 100) {
        std::vector<_Float16> img, inv_img;
        build_fold_fwd(N, F, win, img);
        build_fold_inv(N, win, inv_img);
        const size_t fwd_halfs = img.size();
        img.insert(img.end(), inv_img.begin(), inv_img.end());
        e = hipMalloc(&h->d_fold, img.size() * sizeof(_Float16));
        if (e == hipSuccess) e = hipMemcpy(h->d_fold, img.data(), img.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            fail(nullptr, CMGAN_E_HIP, "folded DFT upload: %s", hipGetErrorString(e));
            hipFree(h->d_tables);
            hipFree(h->d_loss);
            if (h->d_fold) hipFree(h->d_fold);
            delete h;
            return CMGAN_E_HIP;
        }
        h->st.fold_fwd16 = h->d_fold;
        h->st.fold_inv16 = reinterpret_cast<const _Float16*>(h->d_fold) + fwd_halfs;
    }
    *out = h;
    return CMGAN_OK;
}

extern "C" void cmgan_destroy(cmgan_handle* h) {
    if (!h) return;
    if (h->d_tables) hipFree(h->d_tables);
    if (h->d_fold) hipFree(h->d_fold);
    if (h->d_loss) hipFree(h->d_loss);
    if (h->d_weights) hipFree(h->d_weights);
    if (h->d_w16) hipFree(h->d_w16);
    for (auto ev : h->prof.pool) hipEventDestroy(ev);
    for (auto ev : h->ev_fork) hipEventDestroy(ev);
    for (auto ev : h->ev_join) hipEventDestroy(ev);
    for (auto st : h->side) hipStreamDestroy(st);
    delete h;
}

// ------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------
static size_t expected_count(const cmgan_config& c, uint32_t id) {
    const uint32_t grp = id / 64, item = id % 64;
    const size_t F = c.num_features;
    if (grp == G_ENC) {
        switch (item) {
            case ENC_C1_W: return 256; case ENC_C1_GB: return 128; case ENC_C1_PRELU: return 64;
            case ENC_C2_W: return 4 * 3 * 4 * 256; case ENC_C2_BIAS: return 64; case ENC_C2_GB: return 128;
            case ENC_C2_PRELU: return 64;
        }
    } else if (grp == G_DB_E || grp == G_DB_M || grp == G_DB_C) {
        const uint32_t i = item / 4, k = item % 4;
        if (i < 4) {
            if (k == 0) return (size_t)4 * (i + 1) * 6 * 4 * 256;
            if (k == 1) return 64; if (k == 2) return 128; return 64;
        }
    } else if (grp == G_MASK) {
        switch (item) {
            case MK_SP_W: return 4 * 3 * 8 * 256; case MK_SP_BIAS: return 128; case MK_TAIL_W: return 4 * 256;
            case MK_SCALARS: return 8; case MK_PRELU_OUT: return F;
        }
    } else if (grp == G_CPLX) {
        switch (item) {
            case CX_SP_W: return 4 * 3 * 8 * 256; case CX_SP_BIAS: return 128; case CX_GB: return 128;
            case CX_PRELU: return 64; case CX_TAIL_W: return 4 * 256; case CX_BIAS: return 2;
        }
    } else if (grp >= G_CONF0 && grp < G_CONF0 + 8) {
        switch (item) {
            case CF_FF1_W1: case CF_FF2_W1: return 16 * 4 * 256;
            case CF_FF1_B1: case CF_FF2_B1: return 256;
            case CF_FF1_W2: case CF_FF2_W2: return 4 * 16 * 256;
            case CF_FF1_B2: case CF_FF2_B2: return 64;
            case CF_QKV_W: return 12 * 4 * 256; case CF_QKV_B: return 192;
            case CF_WO: return 4 * 4 * 256; case CF_BO: return 64;
            case CF_REL: return (size_t)(2 * c.max_pos_emb + 1) * 16;
            case CF_PW1_W: return 16 * 4 * 256; case CF_PW1_B: return 256;
            case CF_DW_W: return 31 * 128; case CF_DW_B: return 128;
            case CF_PW2_W: return 4 * 8 * 256; case CF_PW2_B: return 64;
            case CF_POST_GB: return 128;
        }
    }
    return 0;
}

// ---- x3 operand images (common.hip.h): pure re-indexing of the fp32 fragment-major data ----------
static void split_h(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// fm [RB][KB][64][4]  ->  [RB][KB/2][hi|lo][64][8]
static void x3_image(const float* fm, int RB, int KB, std::vector<_Float16>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)RB * (KB / 2) * 1024);
    for (int rb = 0; rb < RB; ++rb)
        for (int m = 0; m < KB / 2; ++m)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const float v = fm[(((size_t)rb * KB + 2 * m + (e >> 2)) * 64 + lane) * 4 + (e & 3)];
                    _Float16 hi, lo;
                    split_h(v, hi, lo);
                    const size_t o = base + ((size_t)rb * (KB / 2) + m) * 1024 + lane * 8 + e;
                    out[o] = hi;
                    out[o + 512] = lo;
                }
}
// FeedForward operand images of ffn32_x3_kernel (32x32x16 MFMAs; layouts: ffn32_x3.hip, tests/test_ffn32_tile_model.py).
// M[row][col] read back from the fragment-major fp32 array fm [RB][KB][64][4]:
static float fm_at(const float* fm, int KB, int row, int col) {
    return fm[(((size_t)(row >> 4) * KB + (col >> 4)) * 64 + (row & 15) + 16 * ((col & 15) >> 2)) * 4 + (col & 3)];
}
// W1 [256][64] -> [t 8][kk 4][hi|lo][64][8]: lane (row, hh) slot e = W1[32 t + row][16 kk + 8 (e >> 2) + 4 hh + (e & 3)]
static void x3_image_ffn32_w1(const float* fm, std::vector<_Float16>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)8 * 4 * 1024);
    for (int t = 0; t < 8; ++t)
        for (int kk = 0; kk < 4; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = 32 * t + (lane & 31), hh = lane >> 5;
                    const int col = 8 * (2 * kk + (e >> 2)) + 4 * hh + (e & 3);
                    _Float16 hi, lo;
                    split_h(fm_at(fm, 4, row, col), hi, lo);
                    const size_t o = base + ((size_t)t * 4 + kk) * 1024 + lane * 8 + e;
                    out[o] = hi;
                    out[o + 512] = lo;
                }
}
// W2 [64][256] -> [u 2][ks 16][hi|lo][64][8]: lane (row, hh) slot e = W2[32 u + row][32 (ks >> 1) + 16 (ks & 1) + 8 (e >> 2) + 4 hh + (e & 3)]
static void x3_image_ffn32_w2(const float* fm, std::vector<_Float16>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)2 * 16 * 1024);
    for (int u = 0; u < 2; ++u)
        for (int ks = 0; ks < 16; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = 32 * u + (lane & 31), hh = lane >> 5;
                    const int col = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (e >> 2) + 4 * hh + (e & 3);
                    _Float16 hi, lo;
                    split_h(fm_at(fm, 16, row, col), hi, lo);
                    const size_t o = base + ((size_t)u * 16 + ks) * 1024 + lane * 8 + e;
                    out[o] = hi;
                    out[o + 512] = lo;
                }
}
// conv fm [chunk16][taps][CB][64][4]  ->  [chunk32][taps][CB][hi|lo][64][8]
static void x3_conv_image(const float* fm, int nchunk16, int taps, int CB, std::vector<_Float16>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)(nchunk16 / 2) * taps * CB * 1024);
    for (int c32 = 0; c32 < nchunk16 / 2; ++c32)
        for (int tap = 0; tap < taps; ++tap)
            for (int cb = 0; cb < CB; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int c16 = 2 * c32 + (e >> 2);
                        const float v = fm[((((size_t)c16 * taps + tap) * CB + cb) * 64 + lane) * 4 + (e & 3)];
                        _Float16 hi, lo;
                        split_h(v, hi, lo);
                        const size_t o = base + (((size_t)c32 * taps + tap) * CB + cb) * 1024 + lane * 8 + e;
                        out[o] = hi;
                        out[o + 512] = lo;
                    }
}

// depthwise taps [31][128] (BatchNorm folded) -> A operands of dwpw2t_x3_kernel: [channel group 8][q 9][lane 64][hi 4 | lo 4],
// lane = 4 b + i (b = channel of the group, i = output row of the 4 x 4 block), element k: w_c[4 q + k - i - 2] or 0
static void dw_toeplitz_image(const float* w, std::vector<_Float16>& out) {
    const size_t base = out.size();
    out.resize(base + (size_t)8 * 9 * 64 * 8);
    for (int cg = 0; cg < 8; ++cg)
        for (int q = 0; q < 9; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int ch = 16 * cg + (lane >> 2), i = lane & 3;
                const size_t o = base + (((size_t)cg * 9 + q) * 64 + lane) * 8;
                for (int k = 0; k < 4; ++k) {
                    const int tau = 4 * q + k - i - 2;
                    _Float16 hi, lo;
                    split_h(tau >= 0 && tau < 31 ? w[tau * 128 + ch] : 0.f, hi, lo);
                    out[o + k] = hi;
                    out[o + 4 + k] = lo;
                }
            }
}

// host-side only: the caller uploads the image and swaps it in together with the fp32 payload
static int build_x3_images(const float* payload, const std::map<uint32_t, WEntry>& dir,
                           std::vector<_Float16>& host_img, std::map<uint32_t, size_t>& d16_out) {
    std::vector<_Float16> img;
    std::map<uint32_t, size_t> d16;
    auto pad = [&]() { while (img.size() % 64) img.push_back((_Float16)0.f); };
    for (auto& kv : dir) {
        const uint32_t id = kv.first, grp = id / 64, item = id % 64;
        const float* src = payload + kv.second.off;
        if (grp >= G_CONF0 && grp < G_CONF0 + 8) {
            int RB = 0, KB = 0;
            switch (item) {
                case CF_FF1_W1: case CF_FF2_W1: case CF_PW1_W: RB = 16; KB = 4; break;
                case CF_FF1_W2: case CF_FF2_W2: RB = 4; KB = 16; break;
                case CF_QKV_W: RB = 12; KB = 4; break;
                case CF_WO: RB = 4; KB = 4; break;
                case CF_PW2_W: RB = 4; KB = 8; break;
                default: break;
            }
            if (RB) { pad(); d16[id] = img.size(); x3_image(src, RB, KB, img); }
            if (item == CF_FF1_W1 || item == CF_FF2_W1) { pad(); d16[id | 0x2000] = img.size(); x3_image_ffn32_w1(src, img); }
            if (item == CF_FF1_W2 || item == CF_FF2_W2) { pad(); d16[id | 0x2000] = img.size(); x3_image_ffn32_w2(src, img); }
            if (item == CF_DW_W) { pad(); d16[id] = img.size(); dw_toeplitz_image(src, img); }
            if (item == CF_REL) {                               // rows of [hi 16 | lo 16] halfs
                const size_t rows = kv.second.count / 16;
                pad(); d16[id] = img.size();
                for (size_t r = 0; r < rows; ++r) {
                    _Float16 hi[16], lo[16];
                    for (int d = 0; d < 16; ++d) split_h(src[r * 16 + d], hi[d], lo[d]);
                    for (int d = 0; d < 16; ++d) img.push_back(hi[d]);
                    for (int d = 0; d < 16; ++d) img.push_back(lo[d]);
                }
                // attn32_x3.hip: four planes [hi d 0..7 | hi d 8..15 | lo d 0..7 | lo d 8..15] of [rows][8 halfs], rows in
                // REVERSED distance order (row r' = max_pos - distance), so that the 32 consecutive - descending -
                // distances an MFMA operand tile needs are 32 consecutive 16-byte units: a coalesced fetch
                pad(); d16[id | 0x4000] = img.size();
                for (int plane = 0; plane < 4; ++plane)
                    for (size_t r = 0; r < rows; ++r)
                        for (int e = 0; e < 8; ++e) {
                            _Float16 hi1, lo1;
                            split_h(src[(rows - 1 - r) * 16 + 8 * (plane & 1) + e], hi1, lo1);
                            img.push_back(plane < 2 ? hi1 : lo1);
                        }
            }
        } else if (grp == G_DB_E || grp == G_DB_M || grp == G_DB_C) {
            if (item % 4 == 0) { pad(); d16[id] = img.size(); x3_conv_image(src, 4 * (item / 4 + 1), 6, 4, img); }
        } else if (grp == G_ENC && item == ENC_C2_W) {
            pad(); d16[id] = img.size(); x3_conv_image(src, 4, 3, 4, img);
        } else if ((grp == G_MASK && item == MK_SP_W) || (grp == G_CPLX && item == CX_SP_W)) {
            pad(); d16[id] = img.size(); x3_conv_image(src, 4, 3, 8, img);
        }
    }
    pad();
    host_img.swap(img);
    d16_out.swap(d16);
    return CMGAN_OK;
}

extern "C" int cmgan_load_weights(cmgan_handle* h, const void* blob, size_t bytes) {
    if (!h) return CMGAN_E_BADARG;
    if (!blob || bytes < 16) return fail(h, CMGAN_E_BADARG, "cmgan_load_weights: null / short blob");
    const uint32_t* u = (const uint32_t*)blob;
    if (u[0] != CMGAN_BLOB_MAGIC || u[1] != CMGAN_BLOB_VERSION)
        return fail(h, CMGAN_E_WEIGHTS, "bad blob magic/version (%08x, %u)", u[0], u[1]);
    const uint32_t n = u[2], payload = u[3];
    const size_t head = 16 + (size_t)n * 16;
    if (bytes != head + (size_t)payload * 4) return fail(h, CMGAN_E_WEIGHTS, "blob size mismatch");
    std::map<uint32_t, WEntry> dir;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t* e = u + 4 + 4 * i;
        const uint32_t id = e[0], off = e[1], cnt = e[2];
        if ((size_t)off + cnt > payload || (off % 4) != 0)
            return fail(h, CMGAN_E_WEIGHTS, "entry %u (id %u) out of range / misaligned", i, id);
        const size_t want = expected_count(h->cfg, id);
        if (want == 0) return fail(h, CMGAN_E_WEIGHTS, "unknown weight id %u", id);
        if (want != cnt) return fail(h, CMGAN_E_WEIGHTS, "weight id %u has %u floats, expected %zu", id, cnt, want);
        dir[id] = {off, cnt};
    }
    // Atomic swap: the new fp32 payload and the new x3 image are built and uploaded into fresh buffers first; the
    // handle's pointers and both directories change only after every step has succeeded, so a failed load leaves
    // the previous weights fully usable.
    std::vector<_Float16> img;
    std::map<uint32_t, size_t> d16;
    if (int rc = build_x3_images((const float*)((const char*)blob + head), dir, img, d16)) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    float* nw = nullptr;
    _Float16* nw16 = nullptr;
    hipError_t e = hipMalloc(&nw, (size_t)payload * 4 + 256);
    if (e == hipSuccess) e = hipMalloc(&nw16, img.size() * sizeof(_Float16) + 256);
    if (e == hipSuccess) e = hipMemcpy(nw, (const char*)blob + head, (size_t)payload * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(nw16, img.data(), img.size() * sizeof(_Float16), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();          // no launch may still be reading the old buffers
    if (e != hipSuccess) {
        if (nw) hipFree(nw);
        if (nw16) hipFree(nw16);
        return fail(h, CMGAN_E_HIP, "cmgan_load_weights: %s (previous weights kept)", hipGetErrorString(e));
    }
    if (h->d_weights) hipFree(h->d_weights);
    if (h->d_w16) hipFree(h->d_w16);
    h->d_weights = nw;
    h->d_w16 = nw16;
    h->weight_floats = payload;
    h->dir.swap(dir);
    h->dir16.swap(d16);
    ++h->weights_generation;
    return CMGAN_OK;
}

extern "C" int cmgan_weights_generation(const cmgan_handle* h) { return h ? h->weights_generation : -1; }

static const float* W(cmgan_handle* h, uint32_t id, bool& okflag) {
    auto it = h->dir.find(id);
    if (it == h->dir.end()) {
        if (okflag) fail(h, CMGAN_E_WEIGHTS, "weight id %u (group %u item %u) not loaded", id, id / 64, id % 64);
        okflag = false;
        return nullptr;
    }
    return h->d_weights + it->second.off;
}

static const _Float16* W16(cmgan_handle* h, uint32_t id, bool& okflag) {
    auto it = h->dir16.find(id);
    if (it == h->dir16.end()) {
        if (okflag) fail(h, CMGAN_E_WEIGHTS, "x3 image of weight id %u not built", id);
        okflag = false;
        return nullptr;
    }
    return h->d_w16 + it->second;
}

static bool conf_weights_x3(cmgan_handle* h, int index, ConfWeightsX3& w) {
    bool ok = true;
    const int g = G_CONF0 + index;
    w.ff1_w1 = W16(h, WID(g, CF_FF1_W1), ok); w.ff1_w2 = W16(h, WID(g, CF_FF1_W2), ok);
    w.qkv_w = W16(h, WID(g, CF_QKV_W), ok);   w.wo = W16(h, WID(g, CF_WO), ok);
    w.pw1_w = W16(h, WID(g, CF_PW1_W), ok);   w.pw2_w = W16(h, WID(g, CF_PW2_W), ok);
    w.ff2_w1 = W16(h, WID(g, CF_FF2_W1), ok); w.ff2_w2 = W16(h, WID(g, CF_FF2_W2), ok);
    w.rel_img = W16(h, WID(g, CF_REL), ok);
    w.dw_img = W16(h, WID(g, CF_DW_W), ok);
    w.rel_planes = W16(h, WID(g, CF_REL) | 0x4000, ok);
    w.ff1_w1_32 = W16(h, WID(g, CF_FF1_W1) | 0x2000, ok); w.ff1_w2_32 = W16(h, WID(g, CF_FF1_W2) | 0x2000, ok);
    w.ff2_w1_32 = W16(h, WID(g, CF_FF2_W1) | 0x2000, ok); w.ff2_w2_32 = W16(h, WID(g, CF_FF2_W2) | 0x2000, ok);
    return ok;
}

static bool conf_weights(cmgan_handle* h, int index, ConfWeights& w) {
    bool ok = true;
    const int g = G_CONF0 + index;
    w.ff1_w1 = W(h, WID(g, CF_FF1_W1), ok); w.ff1_b1 = W(h, WID(g, CF_FF1_B1), ok);
    w.ff1_w2 = W(h, WID(g, CF_FF1_W2), ok); w.ff1_b2 = W(h, WID(g, CF_FF1_B2), ok);
    w.qkv_w = W(h, WID(g, CF_QKV_W), ok);   w.qkv_b = W(h, WID(g, CF_QKV_B), ok);
    w.wo = W(h, WID(g, CF_WO), ok);         w.bo = W(h, WID(g, CF_BO), ok);
    w.rel = W(h, WID(g, CF_REL), ok);
    w.pw1_w = W(h, WID(g, CF_PW1_W), ok);   w.pw1_b = W(h, WID(g, CF_PW1_B), ok);
    w.dw_w = W(h, WID(g, CF_DW_W), ok);     w.dw_b = W(h, WID(g, CF_DW_B), ok);
    w.pw2_w = W(h, WID(g, CF_PW2_W), ok);   w.pw2_b = W(h, WID(g, CF_PW2_B), ok);
    w.ff2_w1 = W(h, WID(g, CF_FF2_W1), ok); w.ff2_b1 = W(h, WID(g, CF_FF2_B1), ok);
    w.ff2_w2 = W(h, WID(g, CF_FF2_W2), ok); w.ff2_b2 = W(h, WID(g, CF_FF2_B2), ok);
    w.post_gb = W(h, WID(g, CF_POST_GB), ok);
    w.max_pos = h->cfg.max_pos_emb;
    return ok;
}

// ------------------------------------------------------------------------------------
// workspace plan (all offsets in floats, 64-float aligned)
// ------------------------------------------------------------------------------------
#ifndef CX_IMG
#define CX_IMG 1             // 0 = every layer re-normalises the raw slots (A/B builds)
#endif
#ifndef CX_NIMG
#define CX_NIMG 1            // images written per block: of the block input and of slots 1 .. CX_NIMG - 1.  1, 2 and 3 run at the
                             // same speed (5.96 / 5.92 / 5.91 ms, same-session A/B: what an image saves in staging VALU it costs as a
                             // 266 - 528 MB store); 1 moves the fewest bytes: conv_dense's HBM traffic 1.29 x -> 1.14 x algorithmic
                             // (PMC, profiles/r05_x3_hbm_traffic.json).  plan_ws reserves exactly CX_NIMG images
#endif
struct WsPlan {
    size_t total = 0;
    size_t e[5];          // encoder dense slots [B,P,64]; e[1..4] double as decoder slots, e[0] as SP
    size_t img[3];        // F16X3 dense blocks: (hi, lo) fp16 images of the block input and of slots 1, 2 (ConvArgs::img_out);
                          // only the first CX_NIMG are reserved (the others are never touched: run_dense_block)
    size_t xa, xb, q, k, v, o, u, w;
    size_t dm, dc;        // tail projections [B, T*W, 4]
    size_t partials;
    size_t ns;            // 16 x {scale[B][64], shift[B][64]}
    size_t mstat;         // [B][2]
    size_t scale;         // [B] rms scale (enhance)
    size_t spec;          // [B,2,T,F] (enhance)
    size_t est;           // 2 x [B,T,F] (enhance)
    size_t frames;        // [B,T,n_fft] (istft)
};

static size_t take(size_t& cur, size_t n) {
    const size_t o = cur;
    cur += (n + 63) & ~(size_t)63;
    return o;
}

static WsPlan plan_ws(const cmgan_config& c, int B, int T) {
    WsPlan p;
    const size_t F = c.num_features, F2 = (F + 1) / 2, W2 = 2 * F2;
    const size_t P = (size_t)T * F, P2 = (size_t)T * F2, M = (size_t)B * P2;
    size_t cur = 0;
    p.e[0] = take(cur, (size_t)B * T * std::max(F, W2) * 64);
    for (int i = 1; i < 5; ++i) p.e[i] = take(cur, (size_t)B * P * 64);
    for (int i = 0; i < 3; ++i)
        p.img[i] = (CX_IMG && c.mfma_mode != CMGAN_MFMA_F32 && i < CX_NIMG) ? take(cur, (size_t)B * P * 64) : 0;
    p.xa = take(cur, M * 64);
    p.xb = take(cur, M * 64);
    const size_t qf = std::max(conf_qkv_floats((int)(B * F2), T), conf_qkv_floats(B * T, (int)F2));
    p.q = take(cur, qf); p.k = take(cur, qf); p.v = take(cur, qf); p.o = take(cur, qf);
    p.u = take(cur, M * 128);
    p.w = take(cur, M * 128);
    p.dm = take(cur, (size_t)B * T * W2 * 4 + 64);
    p.dc = take(cur, (size_t)B * T * W2 * 4 + 64);
    const size_t nt = std::max({(size_t)conv3_ntiles(T, (int)F), (size_t)conv_in_ntiles((int)P),
                                (size_t)conv3x_ntiles(T, (int)F, 64), (size_t)conv3x_ntiles(T, (int)F, 128)});
    p.partials = take(cur, (size_t)B * nt * 128 * 2);
    p.ns = take(cur, (size_t)16 * 2 * B * 64);
    p.mstat = take(cur, (size_t)B * 2);
    p.scale = take(cur, (size_t)B);
    p.spec = take(cur, (size_t)B * 2 * P);
    p.est = take(cur, (size_t)B * 2 * P);
    p.frames = take(cur, (size_t)B * T * c.n_fft);
    p.total = cur;
    return p;
}

extern "C" size_t cmgan_workspace_bytes(const cmgan_handle* h, int B, int T) {
    if (!h || B <= 0 || T <= 0) return 0;
    return plan_ws(h->cfg, B, T).total * sizeof(float);
}

// ------------------------------------------------------------------------------------
// front / back end
// ------------------------------------------------------------------------------------
extern "C" int cmgan_num_frames(const cmgan_handle* h, int L) { return h ? L / h->cfg.hop + 1 : 0; }

// torch.stft(center=True) takes any L > n_fft/2 and yields 1 + L / hop frames; only the fused wav -> wav call
// (equal input and output length) needs a whole number of hops.
static int check_wave_len(cmgan_handle* h, int L, bool whole_hops) {
    if (L <= 0 || (whole_hops && L % h->cfg.hop != 0))
        return fail(h, CMGAN_E_BADSHAPE, "L=%d must be a positive multiple of hop=%d", L, h->cfg.hop);
    if (L <= h->cfg.n_fft / 2) return fail(h, CMGAN_E_BADSHAPE, "L=%d must exceed n_fft/2=%d (reflect padding)", L, h->cfg.n_fft / 2);
    return CMGAN_OK;
}

extern "C" int cmgan_rms_scale(cmgan_handle* h, const float* wav, int B, int L, float* scale, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!wav || !scale || B <= 0 || L <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_rms_scale: bad argument");
    launch_rms_scale(begin(h, stream), wav, B, L, scale);
    return check_launch(h, "rms_scale");
}

extern "C" int cmgan_stft_compress(cmgan_handle* h, const float* wav, const float* scale, int B, int L,
                                   float* spec, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!wav || !spec || B <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_stft_compress: bad argument");
    if (int rc = check_wave_len(h, L, false)) return rc;
    launch_stft_compress(begin(h, stream), h->st, wav, scale, B, L, L / h->cfg.hop + 1, spec);
    return check_launch(h, "stft_compress");
}

extern "C" int cmgan_uncompress_istft(cmgan_handle* h, const float* re, const float* im, const float* scale, int B,
                                      int T, float* wav_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!re || !im || !wav_out || B <= 0 || T < 2) return fail(h, CMGAN_E_BADARG, "cmgan_uncompress_istft: bad argument");
    const WsPlan p = plan_ws(h->cfg, B, T);
    if (int rc = check_ws(h, ws, ws_bytes, p.total * sizeof(float))) return rc;
    launch_uncompress_istft(begin(h, stream), h->st, re, im, scale, B, T, (float*)ws + p.frames, wav_out);
    return check_launch(h, "uncompress_istft");
}

extern "C" int cmgan_power_compress(cmgan_handle* h, const float* x, int B, int F, int T, float* y, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || B <= 0 || F <= 0 || T <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_power_compress: bad argument");
    launch_power_compress(begin(h, stream), x, B, F, T, y);
    return check_launch(h, "power_compress");
}

extern "C" int cmgan_power_uncompress(cmgan_handle* h, const float* re, const float* im, int B, int F, int T,
                                      float* y, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!re || !im || !y || B <= 0 || F <= 0 || T <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_power_uncompress: bad argument");
    launch_power_uncompress(begin(h, stream), re, im, B, F, T, y);
    return check_launch(h, "power_uncompress");
}

// ------------------------------------------------------------------------------------
// conformer (stand-alone entry)
// ------------------------------------------------------------------------------------
struct ConfPlan { size_t total, xa, xb, q, k, v, o, u, w; };
static ConfPlan plan_conf(int N, int L) {
    ConfPlan p;
    size_t cur = 0;
    const size_t M = (size_t)N * L, qf = conf_qkv_floats(N, L);
    p.xa = take(cur, M * 64); p.xb = take(cur, M * 64);
    p.q = take(cur, qf); p.k = take(cur, qf); p.v = take(cur, qf); p.o = take(cur, qf);
    p.u = take(cur, M * 128); p.w = take(cur, M * 128);
    p.total = cur;
    return p;
}

extern "C" size_t cmgan_conformer_workspace_bytes(const cmgan_handle* h, int N, int L) {
    if (!h || N <= 0 || L <= 0) return 0;
    return plan_conf(N, L).total * sizeof(float);
}

static int conformer_forward_impl(cmgan_handle* h, int index, const float* x, int N, int L, const unsigned char* mask,
                                  float* y, float* taps, void* ws, size_t ws_bytes, void* stream);

// One ConformerBlock in the handle's split-f16 mode: F16X3 / F16X1 through their own entry points, F16MIX through the
// stage tables of both builds (kernels.h, ConfStageTbl), family by family as cmgan_config.single_mask says.
static bool conformer_mix(cmgan_handle* h, LaunchCtx ctx, const ConfWeights& w, const ConfWeightsX3& w16, const ConfBuffers& b,
                          const TokMap& seq, long M, float* taps, bool outer_residual, const unsigned char* mask) {
    const int sm = h->cfg.single_mask;
    if (sm == 0) return conformer_forward_x3(ctx, w, w16, b, seq, M, taps, outer_residual, mask);
    if (sm == CMGAN_MIX_ALL || (sm | CMGAN_MIX_CONV) == CMGAN_MIX_ALL)
        return conformer_forward_x1(ctx, w, w16, b, seq, M, taps, outer_residual, mask);
    if (!conformer_x3_addressable(seq)) return false;
    const ConfStageTbl &t3 = conf_stages_x3(), &t1 = conf_stages_x1();
    auto pick = [&](int bit) -> const ConfStageTbl& { return (sm & bit) ? t1 : t3; };
    return conformer_forward_tbl(ctx, pick(CMGAN_MIX_FF1), pick(CMGAN_MIX_QKV), pick(CMGAN_MIX_ATTN), pick(CMGAN_MIX_PW1),
                                 pick(CMGAN_MIX_DWPW2), pick(CMGAN_MIX_FF2), w, w16, b, seq, M, taps, outer_residual, mask);
}

extern "C" int cmgan_conformer_forward(cmgan_handle* h, int index, const float* x, int N, int L, float* y,
                                       float* taps, void* ws, size_t ws_bytes, void* stream) {
    return conformer_forward_impl(h, index, x, N, L, nullptr, y, taps, ws, ws_bytes, stream);
}

extern "C" int cmgan_conformer_forward_masked(cmgan_handle* h, int index, const float* x, int N, int L,
                                              const unsigned char* mask, float* y, float* taps, void* ws,
                                              size_t ws_bytes, void* stream) {
    if (h && !mask) return fail(h, CMGAN_E_BADARG, "cmgan_conformer_forward_masked: mask is null");
    return conformer_forward_impl(h, index, x, N, L, mask, y, taps, ws, ws_bytes, stream);
}

static int conformer_forward_impl(cmgan_handle* h, int index, const float* x, int N, int L, const unsigned char* mask,
                                  float* y, float* taps, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x || !y || N <= 0 || L <= 0 || index < 0 || index >= 8) return fail(h, CMGAN_E_BADARG, "cmgan_conformer_forward: bad argument");
    if ((long)N * L >= (1l << 31)) return fail(h, CMGAN_E_BADARG, "cmgan_conformer_forward: N x L too large");
    const TokMap seq = make_seq_map(N, L, 1, L, 0, 1);
    // the split-f16 kernels' own addressing limit (conformer_x3.hip, next to the kernel that has it); none in F32 mode
    if (h->cfg.mfma_mode != CMGAN_MFMA_F32 && !conformer_x3_addressable(seq))
        return fail(h, CMGAN_E_BADARG, "cmgan_conformer_forward: N x L too large for the split-f16 kernels "
                                       "(L < 2^23, N * ceil(L / 32) < 2^31)");
    const ConfPlan p = plan_conf(N, L);
    if (int rc = check_ws(h, ws, ws_bytes, p.total * sizeof(float))) return rc;
    ConfWeights w;
    if (!conf_weights(h, index, w)) return CMGAN_E_WEIGHTS;
    float* f = (float*)ws;
    ConfBuffers b{f + p.xa, f + p.xb, f + p.q, f + p.k, f + p.v, f + p.o, f + p.u, f + p.w};
    hipStream_t s = (hipStream_t)stream;
    const size_t M = (size_t)N * L;
    HIPCHK(h, hipMemcpyAsync(b.xa, x, M * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (h->cfg.mfma_mode != CMGAN_MFMA_F32) {
        ConfWeightsX3 w16;
        if (!conf_weights_x3(h, index, w16)) return CMGAN_E_WEIGHTS;
        if (!conformer_mix(h, begin(h, stream), w, w16, b, seq, (long)M, taps, false, mask))
            return fail(h, CMGAN_E_BADARG, "cmgan_conformer_forward: N x L too large for the split-f16 kernels");
    } else {
        conformer_forward(begin(h, stream), w, b, seq, (long)M, taps, false, mask);
    }
    HIPCHK(h, hipMemcpyAsync(y, b.xa, M * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return check_launch(h, "conformer_forward");
}

// ------------------------------------------------------------------------------------
// TSCNet.forward
// ------------------------------------------------------------------------------------
struct DenseW { const float *w[4], *bias[4], *gb[4], *prelu[4]; const _Float16* w16[4]; };
static bool dense_weights(cmgan_handle* h, int grp, DenseW& d) {
    bool ok = true;
    for (int i = 0; i < 4; ++i) {
        d.w[i] = W(h, WID(grp, DB_W(i)), ok); d.bias[i] = W(h, WID(grp, DB_BIAS(i)), ok);
        d.gb[i] = W(h, WID(grp, DB_GB(i)), ok); d.prelu[i] = W(h, WID(grp, DB_PRELU(i)), ok);
        d.w16[i] = W16(h, WID(grp, DB_W(i)), ok);
    }
    return ok;
}

// DilatedDenseNet (generator.py:39-47): slot 0 = x0 (with optional norm-on-load), layer i writes slot i+1.
// ns(j) -> {scale, shift} storage for norm instance j; returns the instance index used by the last layer.
// F16X3 mode (imgs != null): layer i is the FIRST consumer of its newest input (x0 for i = 0, else slot i - 1): it
// normalises it on load as before and also stores its split-fp16 image (imgs[i], i < 3); layers i + 1 .. 3 read that
// image instead of the raw slot - same values to the bit, without the per-re-read normalise / PReLU / split.
typedef void (*Conv3xFn)(LaunchCtx, const ConvArgs&, const void*, int, int, int);
static void run_dense_block(LaunchCtx ctx, bool x3, const DenseW& d, const float* x0, const float* x0_scale,
                            const float* x0_shift, const float* x0_alpha, float* const slots[4], float* partials,
                            float* const nsc[4], float* const nsh[4], int B, int T, int F, float* const* imgs = nullptr,
                            Conv3xFn conv3x = launch_conv3_x3, bool frozen = false) {
    const int nt = x3 ? conv3x_ntiles(T, F, 64) : conv3_ntiles(T, F);
    for (int i = 0; i < 4; ++i) {
        ConvArgs a{};
        a.in[0] = x0; a.nscale[0] = x0_scale; a.nshift[0] = x0_shift; a.nalpha[0] = x0_alpha;
        for (int s = 1; s <= i; ++s) {
            a.in[s] = slots[s - 1]; a.nscale[s] = nsc[s - 1]; a.nshift[s] = nsh[s - 1]; a.nalpha[s] = d.prelu[s - 1];
        }
        if (CX_IMG && x3 && imgs) {
            const int ni = i < CX_NIMG ? i : CX_NIMG;            // slots 0 .. ni - 1 are read as images
            for (int s = 0; s < ni; ++s) a.in[s] = imgs[s];
            a.img_mask = (1u << ni) - 1u;
            a.img_out = i < CX_NIMG ? imgs[i] : nullptr;
        }
        a.nslots = i + 1;
        a.w = d.w[i]; a.bias = d.bias[i];
        a.out = slots[i]; a.partials = partials;
        a.T = T; a.F = F; a.dil = 1 << i; a.mode = 0; a.ntiles = nt;
        if (x3) conv3x(ctx, a, d.w16[i], B, 2, 64);
        else launch_conv3(ctx, a, B, 2, 64);
        if (!frozen) launch_in_finalize(ctx, partials, B, nt, 64, 0, (double)T * F, d.gb[i], nsc[i], nsh[i]);
    }
}

// Options of the sliced / frozen-statistics forms of the forward (streaming with carried state, cmgan_stream_*):
//   frozen    every InstanceNorm uses the scale / shift pairs of this statistics blob (cmgan_stats_floats) instead of
//             the statistics of the tensor in hand: the in_finalize / mask_stats launches are skipped.  With frozen
//             statistics the encoder and the decoders are exactly time-causal (receptive field 15 frames back,
//             generator.py:16-20, 39-47): frame t of their output depends on frames t - 15 .. t of their input only.
//   stats_out the blob of THIS call's own statistics is copied here after the run (calibration)
//   phases    which third(s) of TSCNet.forward run: encoder (spec -> x_io), TSCBs (x_io in place), decoders
//             (x_io + spec -> out_re / out_im); x_io = channels-last [B, T, F', 64], the layout the three share
enum { PH_ENC = 1, PH_TSCB = 2, PH_DEC = 4, PH_ALL = 7 };
struct TscnetOpts {
    const float* frozen = nullptr;
    float* stats_out = nullptr;
    int phases = PH_ALL;
    float* x_io = nullptr;
};
static size_t stats_floats(int B) { return (size_t)16 * 2 * B * 64 + (size_t)B * 2; }

static int tscnet_impl(cmgan_handle* h, const float* spec, int B, int T, float* out_re, float* out_im,
                       const cmgan_taps* taps, void* ws, size_t ws_bytes, void* stream, bool reset_prof,
                       const TscnetOpts& opt = TscnetOpts()) {
    const bool ph_enc = opt.phases & PH_ENC, ph_tscb = opt.phases & PH_TSCB, ph_dec = opt.phases & PH_DEC;
    if (B <= 0 || T <= 0 || ((ph_enc || ph_dec) && !spec) || (ph_dec && (!out_re || !out_im)) ||
        (opt.phases != PH_ALL && !opt.x_io))
        return fail(h, CMGAN_E_BADARG, "tscnet_forward: bad argument");
    const bool frozen = opt.frozen != nullptr;
    // a time-axis sequence is T rows F2 rows apart: the split-f16 conv-module kernel addresses them with 32-bit byte
    // offsets (conformer_x3_addressable, conformer_x3.hip: 83 k frames per clip at F = 201); the fp32 kernels have no limit
    if (h->cfg.mfma_mode != CMGAN_MFMA_F32 && (long)(T - 1) * ((h->cfg.num_features + 1) / 2) * 512 + 512 >= (1l << 32))
        return fail(h, CMGAN_E_BADARG, "tscnet_forward: T too large (T * ceil(F / 2) * 512 must stay below 2^32)");
    const WsPlan p = plan_ws(h->cfg, B, T);
    if (int rc = check_ws(h, ws, ws_bytes, p.total * sizeof(float))) return rc;
    const int F = h->cfg.num_features, F2 = (F + 1) / 2, W2 = 2 * F2;
    const long P = (long)T * F, P2 = (long)T * F2, M = (long)B * P2;
    float* f = (float*)ws;
    LaunchCtx ctx = begin(h, stream);
    if (h->prof.enabled && reset_prof) h->prof.reset();

    bool ok = true;
    const float* c1w = W(h, WID(G_ENC, ENC_C1_W), ok);
    const float* c1gb = W(h, WID(G_ENC, ENC_C1_GB), ok);
    const float* c1pr = W(h, WID(G_ENC, ENC_C1_PRELU), ok);
    const float* c2w = W(h, WID(G_ENC, ENC_C2_W), ok);
    const float* c2b = W(h, WID(G_ENC, ENC_C2_BIAS), ok);
    const float* c2gb = W(h, WID(G_ENC, ENC_C2_GB), ok);
    const float* c2pr = W(h, WID(G_ENC, ENC_C2_PRELU), ok);
    DenseW dbe, dbm, dbc;
    ok = dense_weights(h, G_DB_E, dbe) && ok;
    ok = dense_weights(h, G_DB_M, dbm) && ok;
    ok = dense_weights(h, G_DB_C, dbc) && ok;
    const float* mk_spw = W(h, WID(G_MASK, MK_SP_W), ok);
    const float* mk_spb = W(h, WID(G_MASK, MK_SP_BIAS), ok);
    const float* mk_tail = W(h, WID(G_MASK, MK_TAIL_W), ok);
    const float* mk_sca = W(h, WID(G_MASK, MK_SCALARS), ok);
    const float* mk_pout = W(h, WID(G_MASK, MK_PRELU_OUT), ok);
    const float* cx_spw = W(h, WID(G_CPLX, CX_SP_W), ok);
    const float* cx_spb = W(h, WID(G_CPLX, CX_SP_BIAS), ok);
    const float* cx_gb = W(h, WID(G_CPLX, CX_GB), ok);
    const float* cx_pr = W(h, WID(G_CPLX, CX_PRELU), ok);
    const float* cx_tail = W(h, WID(G_CPLX, CX_TAIL_W), ok);
    const float* cx_bias = W(h, WID(G_CPLX, CX_BIAS), ok);
    const bool x3 = h->cfg.mfma_mode != CMGAN_MFMA_F32;
    const Conv3xFn conv3x = (h->cfg.single_mask & CMGAN_MIX_CONV) ? launch_conv3_x1 : launch_conv3_x3;
    auto conformer_x = [&](LaunchCtx c, const ConfWeights& w_, const ConfWeightsX3& w16_, const ConfBuffers& b_, const TokMap& m_,
                           long M_, float* taps_, bool outer_, const unsigned char* mask_) {
        return conformer_mix(h, c, w_, w16_, b_, m_, M_, taps_, outer_, mask_);
    };
    ConfWeights cw[8];
    ConfWeightsX3 cw16[8];
    for (int i = 0; i < 2 * h->cfg.num_tscb; ++i) {
        ok = conf_weights(h, i, cw[i]) && ok;
        if (x3) ok = conf_weights_x3(h, i, cw16[i]) && ok;
    }
    const _Float16* c2w16 = x3 ? W16(h, WID(G_ENC, ENC_C2_W), ok) : nullptr;
    const _Float16* mk_spw16 = x3 ? W16(h, WID(G_MASK, MK_SP_W), ok) : nullptr;
    const _Float16* cx_spw16 = x3 ? W16(h, WID(G_CPLX, CX_SP_W), ok) : nullptr;
    if (!ok) return CMGAN_E_WEIGHTS;

    auto nsc = [&](int j) { return f + p.ns + (size_t)j * 2 * B * 64; };
    auto nsh = [&](int j) { return f + p.ns + (size_t)j * 2 * B * 64 + (size_t)B * 64; };
    float* partials = f + p.partials;
    hipStream_t hs = (hipStream_t)stream;
    const size_t ns_floats = (size_t)16 * 2 * B * 64;
    if (frozen) {                                         // the blob = [16 x {scale[B][64], shift[B][64]} | mask stats [B][2]]
        HIPCHK(h, hipMemcpyAsync(f + p.ns, opt.frozen, ns_floats * sizeof(float), hipMemcpyDeviceToDevice, hs));
        HIPCHK(h, hipMemcpyAsync(f + p.mstat, opt.frozen + ns_floats, (size_t)B * 2 * sizeof(float), hipMemcpyDeviceToDevice, hs));
    }
    // (frozen statistics: the convs still write their partial sums, nobody reduces them)
    auto finalize = [&](int ntiles, int cstride, int fold2, double count, const float* gb, float* sc_, float* sh_) {
        if (!frozen) launch_in_finalize(ctx, partials, B, ntiles, cstride, fold2, count, gb, sc_, sh_);
    };

    // ---- dense encoder (generator.py:65-69) --------------------------------------------
    if (ph_enc) {
    launch_conv_in(ctx, spec, c1w, f + p.e[0], partials, B, (int)P);
    finalize(conv_in_ntiles((int)P), 64, 0, (double)P, c1gb, nsc(0), nsh(0));
    {
        float* slots[4] = {f + p.e[1], f + p.e[2], f + p.e[3], f + p.e[4]};
        float* sc[4] = {nsc(1), nsc(2), nsc(3), nsc(4)};
        float* sh[4] = {nsh(1), nsh(2), nsh(3), nsh(4)};
        float* imgs[3] = {f + p.img[0], f + p.img[1], f + p.img[2]};
        run_dense_block(ctx, x3, dbe, f + p.e[0], nsc(0), nsh(0), c1pr, slots, partials, sc, sh, B, T, F, x3 ? imgs : nullptr, conv3x, frozen);
    }
    {   // conv_2: (1,3) stride (1,2) pad (0,1) == stride-1 conv keeping the even columns
        ConvArgs a{};
        a.in[0] = f + p.e[4]; a.nscale[0] = nsc(4); a.nshift[0] = nsh(4); a.nalpha[0] = dbe.prelu[3];
        a.nslots = 1; a.w = c2w; a.bias = c2b; a.out = f + p.xb; a.partials = partials;
        a.T = T; a.F = F; a.dil = 1; a.mode = 1; a.ntiles = x3 ? conv3x_ntiles(T, F, 64) : conv3_ntiles(T, F);
        if (x3) conv3x(ctx, a, c2w16, B, 1, 64);
        else launch_conv3(ctx, a, B, 1, 64);
        finalize(a.ntiles, 64, 0, (double)P2, c2gb, nsc(5), nsh(5));
        launch_in_apply(ctx, f + p.xb, nsc(5), nsh(5), c2pr, f + p.xa, B, P2);
    }
    if (taps && taps->encoder_dev) launch_cl_to_nchw(ctx, f + p.xa, taps->encoder_dev, B, P2);
    }   // ph_enc
    const size_t x_bytes = (size_t)M * 64 * sizeof(float);
    if (opt.x_io && !ph_enc) HIPCHK(h, hipMemcpyAsync(f + p.xa, opt.x_io, x_bytes, hipMemcpyDeviceToDevice, hs));

    // ---- TSCBs (generator.py:92-99) -----------------------------------------------------
    ConfBuffers cb{f + p.xa, f + p.xb, f + p.q, f + p.k, f + p.v, f + p.o, f + p.u, f + p.w};
    const TokMap tmap = make_seq_map(B * F2, T, F2, (long)T * F2, 1, F2);
    const TokMap fmap = make_seq_map(B * T, F2, 1, F2, 0, 1);
    for (int k = 0; ph_tscb && k < h->cfg.num_tscb; ++k) {
        if (x3) {
            if (!conformer_x(ctx, cw[2 * k], cw16[2 * k], cb, tmap, M, nullptr, true, nullptr) ||
                !conformer_x(ctx, cw[2 * k + 1], cw16[2 * k + 1], cb, fmap, M, nullptr, true, nullptr))
                return fail(h, CMGAN_E_BADARG, "tscnet_forward: B x T x F too large for the split-f16 conformer kernels "
                                               "(conformer_x3_addressable, conformer_x3.hip)");
        } else {
            conformer_forward(ctx, cw[2 * k], cb, tmap, M, nullptr, true);
            conformer_forward(ctx, cw[2 * k + 1], cb, fmap, M, nullptr, true);
        }
        if (taps && taps->tscb_dev[k]) launch_cl_to_nchw(ctx, f + p.xa, taps->tscb_dev[k], B, P2);
    }

    if (opt.x_io && !ph_dec) HIPCHK(h, hipMemcpyAsync(opt.x_io, f + p.xa, x_bytes, hipMemcpyDeviceToDevice, hs));
    if (!ph_dec) return check_launch(h, "tscnet_forward");

    // ---- decoders (generator.py:133-139, 151-156) ------------------------------------
    float* dslots[4] = {f + p.e[1], f + p.e[2], f + p.e[3], f + p.e[4]};
    float* sp = f + p.e[0];
    const int nt2 = x3 ? conv3x_ntiles(T, F2, 128) : conv3_ntiles(T, F2);
    for (int dec = 0; dec < 2; ++dec) {
        const DenseW& d = dec == 0 ? dbm : dbc;
        const int j0 = dec == 0 ? 6 : 10;
        float* sc[4] = {nsc(j0), nsc(j0 + 1), nsc(j0 + 2), nsc(j0 + 3)};
        float* sh[4] = {nsh(j0), nsh(j0 + 1), nsh(j0 + 2), nsh(j0 + 3)};
        float* imgs[3] = {f + p.img[0], f + p.img[1], f + p.img[2]};
        run_dense_block(ctx, x3, d, f + p.xa, nullptr, nullptr, nullptr, dslots, partials, sc, sh, B, T, F2, x3 ? imgs : nullptr, conv3x, frozen);
        ConvArgs a{};
        a.in[0] = dslots[3]; a.nscale[0] = sc[3]; a.nshift[0] = sh[3]; a.nalpha[0] = d.prelu[3];
        a.nslots = 1; a.w = dec == 0 ? mk_spw : cx_spw; a.bias = dec == 0 ? mk_spb : cx_spb;
        a.out = sp; a.partials = dec == 0 ? nullptr : partials;
        a.T = T; a.F = F2; a.dil = 1; a.mode = 2; a.ntiles = nt2;
        if (x3) conv3x(ctx, a, dec == 0 ? mk_spw16 : cx_spw16, B, 1, 128);
        else launch_conv3(ctx, a, B, 1, 128);
        if (dec == 0) {
            launch_tail_proj(ctx, sp, nullptr, nullptr, nullptr, mk_tail, f + p.dm, B, (long)T * W2);
        } else {
            finalize(nt2, 128, 1, (double)T * W2, cx_gb, nsc(14), nsh(14));
            launch_tail_proj(ctx, sp, nsc(14), nsh(14), cx_pr, cx_tail, f + p.dc, B, (long)T * W2);
        }
    }
    if (!frozen) launch_mask_stats(ctx, f + p.dm, mk_sca, B, T, F, f + p.mstat);
    launch_final_combine(ctx, spec, f + p.dm, f + p.dc, f + p.mstat, mk_sca, mk_pout, cx_bias, B, T, F, out_re,
                         out_im, taps ? taps->mask_dev : nullptr, taps ? taps->complex_dev : nullptr);
    if (opt.stats_out) {
        HIPCHK(h, hipMemcpyAsync(opt.stats_out, f + p.ns, ns_floats * sizeof(float), hipMemcpyDeviceToDevice, hs));
        HIPCHK(h, hipMemcpyAsync(opt.stats_out + ns_floats, f + p.mstat, (size_t)B * 2 * sizeof(float), hipMemcpyDeviceToDevice, hs));
    }
    return check_launch(h, "tscnet_forward");
}

extern "C" int cmgan_tscnet_forward(cmgan_handle* h, const float* spec, int B, int T, float* out_re, float* out_im,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    return tscnet_impl(h, spec, B, T, out_re, out_im, nullptr, ws, ws_bytes, stream, true);
}

extern "C" int cmgan_tscnet_forward_taps(cmgan_handle* h, const float* spec, int B, int T, float* out_re,
                                         float* out_im, const cmgan_taps* taps, void* ws, size_t ws_bytes,
                                         void* stream) {
    if (!h) return CMGAN_E_BADARG;
    return tscnet_impl(h, spec, B, T, out_re, out_im, taps, ws, ws_bytes, stream, true);
}

// ---- frozen-statistics / sliced forms: streaming with carried state (include/cmgan_hip.h, "streaming") ----
extern "C" size_t cmgan_stats_floats(const cmgan_handle* h, int B) { return h && B > 0 ? stats_floats(B) : 0; }

extern "C" int cmgan_tscnet_forward_stats(cmgan_handle* h, const float* spec, int B, int T, float* out_re, float* out_im,
                                          const float* frozen_stats, float* stats_out, void* ws, size_t ws_bytes,
                                          void* stream) {
    if (!h) return CMGAN_E_BADARG;
    TscnetOpts o;
    o.frozen = frozen_stats; o.stats_out = stats_out;
    return tscnet_impl(h, spec, B, T, out_re, out_im, nullptr, ws, ws_bytes, stream, true, o);
}

extern "C" int cmgan_stream_encoder(cmgan_handle* h, const float* spec, int B, int T, const float* frozen_stats,
                                    float* x_out, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!frozen_stats || !x_out) return fail(h, CMGAN_E_BADARG, "cmgan_stream_encoder: bad argument");
    TscnetOpts o;
    o.frozen = frozen_stats; o.phases = PH_ENC; o.x_io = x_out;
    return tscnet_impl(h, spec, B, T, nullptr, nullptr, nullptr, ws, ws_bytes, stream, false, o);
}

extern "C" int cmgan_stream_tscb(cmgan_handle* h, float* x, int B, int T, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!x) return fail(h, CMGAN_E_BADARG, "cmgan_stream_tscb: bad argument");
    TscnetOpts o;
    o.phases = PH_TSCB; o.x_io = x;
    return tscnet_impl(h, nullptr, B, T, nullptr, nullptr, nullptr, ws, ws_bytes, stream, false, o);
}

extern "C" int cmgan_stream_decoder(cmgan_handle* h, const float* x, const float* spec, int B, int T,
                                    const float* frozen_stats, float* out_re, float* out_im, void* ws, size_t ws_bytes,
                                    void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!frozen_stats || !x) return fail(h, CMGAN_E_BADARG, "cmgan_stream_decoder: bad argument");
    TscnetOpts o;
    o.frozen = frozen_stats; o.phases = PH_DEC; o.x_io = const_cast<float*>(x);
    return tscnet_impl(h, spec, B, T, out_re, out_im, nullptr, ws, ws_bytes, stream, false, o);
}

extern "C" int cmgan_enhance(cmgan_handle* h, const float* wav, int B, int L, float* wav_out, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!h) return CMGAN_E_BADARG;
    if (!wav || !wav_out || B <= 0) return fail(h, CMGAN_E_BADARG, "cmgan_enhance: bad argument");
    if (int rc = check_wave_len(h, L, true)) return rc;
    const int T = L / h->cfg.hop + 1;
    const WsPlan p = plan_ws(h->cfg, B, T);
    if (int rc = check_ws(h, ws, ws_bytes, p.total * sizeof(float))) return rc;
    float* f = (float*)ws;
    const long P = (long)T * h->cfg.num_features;
    if (h->prof.enabled) h->prof.reset();
    LaunchCtx ctx = begin(h, stream);
    launch_rms_scale(ctx, wav, B, L, f + p.scale);
    launch_stft_compress(ctx, h->st, wav, f + p.scale, B, L, T, f + p.spec);
    if (int rc = check_launch(h, "stft_compress")) return rc;
    if (int rc = tscnet_impl(h, f + p.spec, B, T, f + p.est, f + p.est + (size_t)B * P, nullptr, ws, ws_bytes, stream,
                             false))
        return rc;
    launch_uncompress_istft(ctx, h->st, f + p.est, f + p.est + (size_t)B * P, f + p.scale, B, T, f + p.frames, wav_out);
    return check_launch(h, "enhance");
}

// The same pipeline as `branches` part-batch branches on as many streams (fork / join by events, so a stream capture of
// `stream` records them as parallel paths of one hipGraph): branch 0 on `stream`, the others on streams the handle
// owns; branch i + 1 starts once branch i has issued `offset_launches` kernels (0 = all together).  The rows are
// independent and each branch runs the unchanged per-row arithmetic, so the result equals cmgan_enhance bit
// for bit.  Why it pays: a kernel's ramp-up and drain (and the dependent launch behind it) leave CUs idle ~240 times
// per forward; with a second branch in flight the other branch's workgroups fill them.
#define CMGAN_MAX_BRANCHES 8
static int branch_rows(int B, int n, int i) { return B / n + (i < B % n ? 1 : 0); }

extern "C" size_t cmgan_workspace_bytes_branched(const cmgan_handle* h, int B, int T, int branches) {
    if (!h || B <= 0 || T <= 0 || branches < 1 || branches > CMGAN_MAX_BRANCHES) return 0;
    const int n = branches < B ? branches : B;
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += plan_ws(h->cfg, branch_rows(B, n, i), T).total * sizeof(float);
    return total;
}

extern "C" int cmgan_enhance_branched(cmgan_handle* h, const float* wav, int B, int L, float* wav_out, void* ws,
                                      size_t ws_bytes, void* stream, int branches, int offset_launches) {
    if (!h) return CMGAN_E_BADARG;
    if (!wav || !wav_out || B <= 0 || offset_launches < 0 || branches < 1 || branches > CMGAN_MAX_BRANCHES)
        return fail(h, CMGAN_E_BADARG, "cmgan_enhance_branched: bad argument");
    const int n = branches < B ? branches : B;
    if (n == 1) return cmgan_enhance(h, wav, B, L, wav_out, ws, ws_bytes, stream);
    if (h->prof.enabled) return fail(h, CMGAN_E_BADARG, "cmgan_enhance_branched: per-launch profiling needs the one-stream form (cmgan_enhance)");
    if (int rc = check_wave_len(h, L, true)) return rc;
    const int T = L / h->cfg.hop + 1;
    if (int rc = check_ws(h, ws, ws_bytes, cmgan_workspace_bytes_branched(h, B, T, n))) return rc;
    while ((int)h->side.size() < n - 1) {                 // first call (the warm-up a caller runs before capturing)
        hipStream_t st;
        hipEvent_t ef, ej;
        HIPCHK(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&ef, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        h->side.push_back(st); h->ev_fork.push_back(ef); h->ev_join.push_back(ej);
    }
    int rc = CMGAN_OK, rows_done = 0;
    char* wsp = (char*)ws;
    for (int i = 0; i < n && rc == CMGAN_OK; ++i) {
        hipStream_t si = i == 0 ? (hipStream_t)stream : h->side[i - 1];
        const int Bi = branch_rows(B, n, i);
        const size_t wi = plan_ws(h->cfg, Bi, T).total * sizeof(float);
        Fork fork;                                        // releases branch i + 1 after `offset_launches` launches of this one
        if (i + 1 < n) {
            fork.at = offset_launches; fork.ev = h->ev_fork[i];
            if (offset_launches == 0) { hipEventRecord(fork.ev, si); fork.fired = true; }
            h->fork = &fork;
        }
        rc = cmgan_enhance(h, wav + (size_t)rows_done * L, Bi, L, wav_out + (size_t)rows_done * L, wsp, wi, (void*)si);
        h->fork = nullptr;
        if (i + 1 < n) {
            if (!fork.fired) hipEventRecord(fork.ev, si);  // fewer launches than the offset: plain sequence
            hipStreamWaitEvent(h->side[i], fork.ev, 0);   // (even after a failure: a capturing caller needs every stream joined)
        }
        rows_done += Bi;
        wsp += wi;
    }
    // join every side stream that was forked (a failed branch included: no unjoined stream may be left in a capture)
    for (int i = 1; i < n; ++i) {
        hipEventRecord(h->ev_join[i - 1], h->side[i - 1]);
        hipStreamWaitEvent((hipStream_t)stream, h->ev_join[i - 1], 0);
    }
    if (rc) return rc;
    return check_launch(h, "enhance_branched");
}

// ------------------------------------------------------------------------------------
// diagnostics
// ------------------------------------------------------------------------------------
extern "C" int cmgan_selftest_mfma(cmgan_handle* h, float* max_err_host) {
    if (!h || !max_err_host) return CMGAN_E_BADARG;
    const int KB = 3, K = 16 * KB;
    std::vector<double> A(16 * K), Bt(16 * K);     // A[i][k], Bt[j][k] = B[k][j]; asymmetric on purpose
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < K; ++k) {
            A[i * K + k] = 0.25 * ((i * 7 + k * 3) % 11) - 1.0;
            Bt[i * K + k] = 0.125 * ((i * 5 + k * 13) % 17) - 0.5 + 0.01 * i;
        }
    std::vector<float> af(16 * K), bf(16 * K), d(256);
    pack_fm(A, 16, K, af.data());
    pack_fm(Bt, 16, K, bf.data());
    float *da, *db, *dd;
    HIPCHK(h, hipMalloc(&da, af.size() * 4)); HIPCHK(h, hipMalloc(&db, bf.size() * 4)); HIPCHK(h, hipMalloc(&dd, 1024));
    HIPCHK(h, hipMemcpy(da, af.data(), af.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(db, bf.data(), bf.size() * 4, hipMemcpyHostToDevice));
    launch_selftest_mfma(nullptr, da, db, dd, KB);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dd);
    double worst = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)(float)A[i * K + k] * (double)(float)Bt[j * K + k];
            worst = std::max(worst, fabs(ref - (double)d[i * 16 + j]));
        }
    *max_err_host = (float)worst;
    return CMGAN_OK;
}

extern "C" int cmgan_selftest_mfma_x3(cmgan_handle* h, float* max_err_host) {
    if (!h || !max_err_host) return CMGAN_E_BADARG;
    const int KB = 4, K = 16 * KB;                   // two k32 blocks
    std::vector<double> A(16 * K), Bt(16 * K);
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < K; ++k) {
            A[i * K + k] = 0.173 * ((i * 7 + k * 3) % 11) - 0.83 + 1e-3 * k;
            Bt[i * K + k] = 0.0917 * ((i * 5 + k * 13) % 17) - 0.61 + 0.013 * i;
        }
    std::vector<float> af(16 * K), bf(16 * K), d(256);
    pack_fm(A, 16, K, af.data());
    pack_fm(Bt, 16, K, bf.data());
    std::vector<_Float16> aimg;
    x3_image(af.data(), 1, KB, aimg);
    _Float16* da; float *db, *dd;
    HIPCHK(h, hipMalloc(&da, aimg.size() * 2)); HIPCHK(h, hipMalloc(&db, bf.size() * 4)); HIPCHK(h, hipMalloc(&dd, 1024));
    HIPCHK(h, hipMemcpy(da, aimg.data(), aimg.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(db, bf.data(), bf.size() * 4, hipMemcpyHostToDevice));
    launch_selftest_x3(nullptr, da, db, dd, KB / 2);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dd);
    double worst = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0.0;
            for (int k = 0; k < K; ++k) ref += (double)(float)A[i * K + k] * (double)(float)Bt[j * K + k];
            worst = std::max(worst, fabs(ref - (double)d[i * 16 + j]));
        }
    *max_err_host = (float)worst;
    return CMGAN_OK;
}

extern "C" int cmgan_set_profiling(cmgan_handle* h, int enabled) {
    if (!h) return CMGAN_E_BADARG;
    h->prof.enabled = enabled != 0;
    h->prof.reset();
    return CMGAN_OK;
}

extern "C" int cmgan_profile_read(cmgan_handle* h, cmgan_kernel_time* out, int cap) {
    if (!h || !out || cap <= 0) return 0;
    int n = 0;
    for (auto& r : h->prof.recs) {
        if (n >= cap) break;
        hipEventSynchronize(r.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        out[n].name = r.name;
        out[n].ms = ms;
        ++n;
    }
    return n;
}
