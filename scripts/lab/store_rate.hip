// Lab probe (not part of the product): sustained global_store_dwordx4 rate per CU for two address shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
// shape 0: one wave-instruction = 4 rows x 256 B (16 lanes per row)      -- the staged GEMM epilogue
// shape 1: one wave-instruction = 32 rows x 32 B (lane pair per row)      -- stores straight from a transposed accumulator
// shape 2: one wave-instruction = 1 KiB contiguous
template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(float* out, size_t ld, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
    // every workgroup owns a 128-row x 256-col f32 tile per iteration (ld floats per row), like the GEMM
    for (int it = 0; it < iters; ++it) {
        float* tile = out + ((size_t)(blockIdx.x * iters + it) * 128) * ld;
        float* sub = tile + (size_t)(wave >> 2) * 64 * ld + (wave & 3) * 64;      // 64 x 64 per wave
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float* p;
            if (SHAPE == 0) p = sub + (size_t)(j * 4 + (lane >> 4)) * ld + (lane & 15) * 4;
            else if (SHAPE == 1) p = sub + (size_t)((j & 1) * 32 + (lane & 31)) * ld + (j >> 1) * 8 + (lane >> 5) * 4;
            else p = tile + (size_t)(wave * 16 + j) * 256 + lane * 4;
            *reinterpret_cast<float4*>(p) = v;
        }
    }
}
template <int SHAPE>
static void run(const char* name, float* out, size_t ld, int iters, int wgs = 256) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(store_kernel<SHAPE>, dim3(wgs), dim3(512), 0, 0, out, ld, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)wgs * iters * 131072.0;
    printf("%-40s %8.3f ms  %6.2f TB/s  %5.1f B/clk/CU@2.4GHz  (%.0f cyc per 128 KiB tile)\n", name, ms, bytes / ms / 1e9,
           bytes / wgs / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / iters);
}
int main() {
    const int iters = 36;
    const size_t ld = 512;      // N = 512 output rows
    float* out;
    const size_t n = (size_t)256 * iters * 128 * ld;
    if (hipMalloc(&out, n * 4 + (1 << 20)) != hipSuccess) return 1;
    run<0>("4 rows x 256 B per instruction", out, ld, iters);
    run<1>("32 rows x 32 B per instruction", out, ld, iters);
    run<2>("1 KiB contiguous per instruction", out, ld, iters);
    run<0>("4 rows x 256 B, 64 workgroups", out, ld, iters, 64);
    run<0>("4 rows x 256 B, 32 workgroups", out, ld, iters, 32);
    run<0>("4 rows x 256 B, 8 workgroups", out, ld, iters, 8);
    run<1>("32 rows x 32 B, 32 workgroups", out, ld, iters, 32);
    return 0;
}
