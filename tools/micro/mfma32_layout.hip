// Prints the operand / result layout of v_mfma_f32_32x32x16_f16 on the device it runs on:
// A[i][k] = i (row id) for one probe, B[k][j] = 1 -> D[i][j] = 16 * i tells which row a register holds;
// then A = 1, B[k][j] = j -> which column; then a k-probe.   hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    f32x16 c;
    // probe 1: rows.  lane (i = lane & 31) as A row -> value i; B = 1
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)(lane & 31); b[e] = (_Float16)1.f; }
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(0 * 64 + lane) * 16 + r] = c[r] / 16.f;
    // probe 2: columns.  A = 1, B column j = lane & 31 -> value j
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)1.f; b[e] = (_Float16)(float)(lane & 31); }
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(1 * 64 + lane) * 16 + r] = c[r] / 16.f;
    // probe 3: contraction pairing.  A slot (kh, e) -> 2^(kh*8+e) only on row 0 lanes; B slot (kh,e) = 1 if (kh,e)==(1,3)
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)(1 + (lane >> 5) * 8 + e); b[e] = (_Float16)(((lane >> 5) == 1 && e == 3) ? 1.f : 0.f); }
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) out[(2 * 64 + lane) * 16 + r] = c[r];
}
int main() {
    float* d; hipMalloc(&d, 3 * 64 * 16 * 4);
    probe<<<1, 64>>>(d);
    static float h[3 * 64 * 16];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) {
        printf("lane %2d rows:", l); for (int r = 0; r < 16; ++r) printf(" %2.0f", h[(0 * 64 + l) * 16 + r]);
        printf("   col: %2.0f   kprobe: %2.0f\n", h[(1 * 64 + l) * 16 + 0], h[(2 * 64 + l) * 16 + 0]);
    }
    return 0;
}
