// api_internal.h - the handle and the small helpers shared by the two C-ABI translation units (api.hip: inference
// path; api_train.hip: training-step slices).  Host-only.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/cmgan_hip.h"
#include "kernels.h"

#ifndef LOSS_BLOCKS
#define LOSS_BLOCKS 256                       // partial-sum slots of cmgan_loss_terms (scratch allocated at create)
#endif

inline thread_local std::string g_create_error;

struct WEntry { size_t off; size_t count; };

struct cmgan_handle {
    cmgan_config cfg;
    int device = 0;
    std::string err;
    // tables
    float* d_tables = nullptr;
    void* d_fold = nullptr;               // folded-DFT fp16 hi/lo images (x3 mode, n_fft 400)
    double* d_loss = nullptr;             // LOSS_BLOCKS x 4 partial sums of cmgan_loss_terms
    SpectralTables st{};
    // weights
    float* d_weights = nullptr;
    size_t weight_floats = 0;
    int weights_generation = 0;           // bumped by every successful cmgan_load_weights (stale-graph detection)
    std::map<uint32_t, WEntry> dir;
    // x3 (f16 split) operand images, built from the fp32 fragment-major weights at load time
    _Float16* d_w16 = nullptr;
    std::map<uint32_t, size_t> dir16;     // id -> offset in halfs (rel-pos lo plane at id | 0x8000)
    Profiler prof;
    std::vector<std::string> prof_names;
    // cmgan_enhance_branched: side streams + fork / join events (created by the first call, outside any capture)
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> ev_fork, ev_join;
    Fork* fork = nullptr;                 // non-null while the first branch of cmgan_enhance_branched is being issued
};

inline int fail(cmgan_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t _e = (call);                                                                  \
        if (_e != hipSuccess) return fail(h, CMGAN_E_HIP, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)

inline int check_launch(cmgan_handle* h, const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, CMGAN_E_HIP, "%s: %s", where, hipGetErrorString(e));
    return CMGAN_OK;
}


inline int check_ws(cmgan_handle* h, void* ws, size_t bytes, size_t need) {
    if (!ws) return fail(h, CMGAN_E_BADARG, "workspace is null");
    if (((uintptr_t)ws & 255) != 0) return fail(h, CMGAN_E_WORKSPACE, "workspace must be 256-byte aligned");
    if (bytes < need) return fail(h, CMGAN_E_WORKSPACE, "workspace too small: %zu < %zu bytes", bytes, need);
    return CMGAN_OK;
}

inline LaunchCtx begin(cmgan_handle* h, void* stream) {
    return LaunchCtx{(hipStream_t)stream, &h->prof, h->fork};
}

