// Microbenchmark: how much VALU / LDS work hides in the shadow of a wave's OWN MFMA stream on gfx950?
// One wave per SIMD (256-thread block, 1 block per CU), NIT iterations of [12 x v_mfma_f32_16x16x32_f16
// on 4 independent accumulators] + K filler instructions of one kind, timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o interleave interleave.hip && ./interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KV, int KL, int WAVES, int NACC>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, long long* cyc, int nit) {
    __shared__ float lds[25600];          // 100 KB: exactly one block per CU, so WAVES / 4 waves per SIMD
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * lane + i); b[i] = (_Float16)(0.002f * lane - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
    lds[threadIdx.x] = lane;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j % NACC], 0, 0, 0);
            // fillers, spread evenly between the MFMAs
#pragma unroll
            for (int q = 0; q < (KV + 11 - j) / 12; ++q) v[(j + q) & 7] = __builtin_fmaf(v[(j + q) & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = 0; q < (KL + 11 - j) / 12; ++q) v[(j + q + 4) & 7] += lds[(lane * 4 + 64 * ((j + q) & 15)) & 4095];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// same FLOPs per iteration with 6 x v_mfma_f32_32x32x16_f16 (8 passes each) on 2 accumulators
template <int KV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(float* out, long long* cyc, int nit) {
    __shared__ float lds[25600];
    const int lane = threadIdx.x & 63;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * lane + i); b[i] = (_Float16)(0.002f * lane - i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.5f + i;
    lds[threadIdx.x] = lane;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < (KV + 5 - j) / 6; ++q) v[(j + q) & 7] = __builtin_fmaf(v[(j + q) & 7], 1.0001f, 0.5f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KV, int WAVES>
void run32(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 64 * WAVES * 4);
    hipMalloc(&cyc, 8);
    const int nit = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k32<KV, WAVES><<<256, 64 * WAVES>>>(out, cyc, nit);
    hipEventRecord(e0);
    k32<KV, WAVES><<<256, 64 * WAVES>>>(out, cyc, nit);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * WAVES * nit * 6 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-34s 32x32x16 waves/SIMD %d : kernel %.3f ms = %.0f TFLOP/s\n", name, WAVES / 4, ms, tf);
    hipFree(out); hipFree(cyc);
}

template <int KV, int KL, int WAVES, int NACC = 4>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 64 * WAVES * 4);
    hipMalloc(&cyc, 8);
    const int nit = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KV, KL, WAVES, NACC><<<256, 64 * WAVES>>>(out, cyc, nit);
    hipEventRecord(e0);
    k<KV, KL, WAVES, NACC><<<256, 64 * WAVES>>>(out, cyc, nit);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * WAVES * nit * 12 * 16384.0 / (ms * 1e-3) / 1e12;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s acc %2d waves/SIMD %d : %7.1f cycles per 12-MFMA group (%.1f per MFMA)  kernel %.3f ms = %.0f TFLOP/s\n", name, NACC, WAVES / 4, (double)c / nit, (double)c / nit / 12, ms, tf);
    hipFree(out); hipFree(cyc);
}

int main() {
    run32<0, 8>("6 MFMA only");
    run32<24, 8>("6 MFMA + 24 v_fma");
    run32<36, 8>("6 MFMA + 36 v_fma");
    run32<48, 8>("6 MFMA + 48 v_fma");
    run32<0, 4>("6 MFMA only");
    run32<24, 4>("6 MFMA + 24 v_fma");
    run32<48, 4>("6 MFMA + 48 v_fma");
    run<0, 0, 4, 12>("12 MFMA only");
    run<0, 0, 8, 12>("12 MFMA only");
    run<0, 0, 4, 6>("12 MFMA only");
    run<24, 0, 4, 12>("12 MFMA + 24 v_fma");
    run<36, 0, 4, 12>("12 MFMA + 36 v_fma");
    run<24, 6, 4, 12>("12 MFMA + 24 v_fma + 6 ds_read");
    run<0, 0, 4>("12 MFMA only");
    run<12, 0, 4>("12 MFMA + 12 v_fma");
    run<24, 0, 4>("12 MFMA + 24 v_fma");
    run<36, 0, 4>("12 MFMA + 36 v_fma");
    run<48, 0, 4>("12 MFMA + 48 v_fma");
    run<0, 6, 4>("12 MFMA + 6 ds_read_b32");
    run<0, 12, 4>("12 MFMA + 12 ds_read_b32");
    run<24, 6, 4>("12 MFMA + 24 v_fma + 6 ds_read");
    run<0, 0, 8>("12 MFMA only");
    run<24, 0, 8>("12 MFMA + 24 v_fma");
    run<36, 0, 8>("12 MFMA + 36 v_fma");
    run<24, 6, 8>("12 MFMA + 24 v_fma + 6 ds_read");
    return 0;
}
