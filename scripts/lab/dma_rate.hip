// Lab probe (not part of the product): sustained rate of global_load_lds_dwordx4 per CU for several
// source-address shapes, L2-resident or streaming.  hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// Each workgroup (8 waves) moves `steps` stages of 48 KiB (6 wave-instructions of 1 KiB per wave per stage)
// into a 3-stage ring; two stages stay in flight.  rows_per_instr x bytes_per_row = 1 KiB.
template <int BYTES_PER_ROW, bool BARRIER>
__global__ __launch_bounds__(512) void dma_kernel(const char* src, size_t row_stride, size_t wg_stride, size_t step_stride,
                                                 size_t wrap, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int CPR = BYTES_PER_ROW / 16, RPI = 1024 / BYTES_PER_ROW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rin = lane / CPR, pch = lane % CPR;
    const char* base[6];
    for (int i = 0; i < 6; ++i) {
        const int j = wave * 6 + i;    // 48 instructions per stage
        base[i] = src + (size_t)blockIdx.x * wg_stride + (size_t)(j * RPI + rin) * row_stride + pch * 16;
    }
    auto issue = [&](int t) {
        const size_t off = ((size_t)t * step_stride) % wrap;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds(base[i] + off, (lds_ptr_t)(lds + (t % 3) * 49152 + (wave * 6 + i) * 1024), 16, 0, 0);
    };
    issue(0);
    issue(1);
    for (int t = 0; t < steps; ++t) {
        if (t + 1 < steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        if (t + 2 < steps) issue(t + 2);
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(lds + 64);
}

template <int BPR, bool BAR>
static void run(const char* name, const char* src, size_t row_stride, size_t wg_stride, size_t step_stride, size_t wrap,
                int steps, unsigned* sink, int wgs) {
    auto k = dma_kernel<BPR, BAR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 147456, 0, src, row_stride, wg_stride, step_stride, wrap, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)wgs * steps * 49152.0;
    printf("%-44s %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU@2.4GHz\n", name, ms, bytes / ms / 1e9, bytes / wgs / (ms * 1e-3 * 2.4e9));
    fflush(stdout);
}

int main() {
    const size_t big = (size_t)6 << 30;
    char* buf;
    if (hipMalloc(&buf, big + (1 << 20)) != hipSuccess) return 1;
    hipMemset(buf, 1, big);
    unsigned* sink;
    hipMalloc(&sink, 4096 * 4);
    const int steps = 2000, wgs = 256;
    // ---- L2-resident: every workgroup reads the same 1-4 MiB (weights-like).  48 instr * rows each.
    // 64-B rows: 768 rows x row_stride 1 KiB = 768 KiB window, step advances 64 B inside the row, wraps at 1 KiB
    run<64, true>("L2  16 rows x  64 B (row stride 1 KiB)", buf, 1024, 0, 64, 1024, steps, sink, wgs);
    run<128, true>("L2   8 rows x 128 B (row stride 2 KiB)", buf, 2048, 0, 128, 2048, steps, sink, wgs);
    run<256, true>("L2   4 rows x 256 B (row stride 2 KiB)", buf, 2048, 0, 256, 2048, steps, sink, wgs);
    run<1024, true>("L2   1 KiB contiguous", buf, 1024, 0, 49152, 1 << 20, steps, sink, wgs);
    run<1024, false>("L2   1 KiB contiguous, no barrier", buf, 1024, 0, 49152, 1 << 20, steps, sink, wgs);
    run<128, false>("L2   8 rows x 128 B, no barrier", buf, 2048, 0, 128, 2048, steps, sink, wgs);
    // ---- streaming: every workgroup walks its own 24 MiB region (HBM)
    const size_t wg_stride = (size_t)24 << 20;
    run<1024, true>("HBM  1 KiB contiguous", buf, 1024, wg_stride, 49152, wg_stride, 400, sink, wgs);
    run<128, true>("HBM  8 rows x 128 B (row stride 48 KiB)", buf, 49152, wg_stride, 128, 49152, 380, sink, wgs);
    run<64, true>("HBM 16 rows x  64 B (row stride 24 KiB)", buf, 24576, wg_stride, 64, 24576, 380, sink, wgs);
    // ---- mixed like the GEMM: handled by the numbers above (2/3 L2 weights + 1/3 streamed rows)
    return 0;
}
