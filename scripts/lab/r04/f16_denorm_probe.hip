// Lab probe (round 4): does v_mfma_f32_32x32x16_f16 keep fp16 subnormal A/B operands, and does the f32 -> f16
// conversion the compiler emits produce them?  Build: hipcc --offload-arch=gfx950 -O3 f16_denorm_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* a_in, const float* b_in, float* out, float* cvt) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)a_in[j];          // same value on every row / k
        b[j] = (_Float16)b_in[j];
    }
    if (lane == 0)
        for (int j = 0; j < 8; ++j) cvt[j] = (float)a[j];
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

int main() {
    float ha[8], hb[8], *da, *db, *dout, *dcvt;
    // a: fp16 subnormals (2^-20 .. ), b: 1024 -> products representable and exact in f32
    for (int j = 0; j < 8; ++j) { ha[j] = ldexpf(1.0f + j, -24); hb[j] = 1024.0f; }
    hipMalloc(&da, 32); hipMalloc(&db, 32); hipMalloc(&dout, 4); hipMalloc(&dcvt, 32);
    hipMemcpy(da, ha, 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout, dcvt);
    float out, cvt[8];
    hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost); hipMemcpy(cvt, dcvt, 32, hipMemcpyDeviceToHost);
    double expect = 0;
    for (int j = 0; j < 8; ++j) expect += 2.0 * ha[j] * 1024.0;     // k = 8h + j, both halves h = 0, 1 hold the same values
    printf("converted subnormals:");
    for (int j = 0; j < 8; ++j) printf(" %g (want %g)", cvt[j], ha[j]);
    printf("\nmfma sum %.9g, expected with subnormals kept %.9g, flushed 0\n", out, expect);
    // mixed: a = subnormal, b = subnormal product underflow irrelevant; a normal, b subnormal
    for (int j = 0; j < 8; ++j) { ha[j] = 1024.0f; hb[j] = ldexpf(1.0f + j, -24); }
    hipMemcpy(da, ha, 32, hipMemcpyHostToDevice); hipMemcpy(db, hb, 32, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dout, dcvt);
    hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost);
    printf("B subnormal: mfma sum %.9g, expected %.9g\n", out, expect);
    return 0;
}
