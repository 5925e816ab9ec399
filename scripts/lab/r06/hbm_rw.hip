// Lab, round 6 (not part of the product): what HBM delivers for reads, for writes and for mixes of the two, with the access
// pattern of the row kernels (a wave moves whole 2-KiB rows, 16-byte accesses per lane, consecutive rows per wave).
//   hipcc --offload-arch=gfx950 -O3 scripts/lab/r06/hbm_rw.hip -o scripts/lab/r06/bin/hbm_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// mode: reads per row R (0..3 tensors), writes per row W (0..3 tensors); each tensor `rows` x 512 floats
template <int R, int W, bool IL = false>
__global__ __launch_bounds__(256) void rw_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t rows, size_t tstride, float* sink, int rw) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t r0 = wave * rw;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t r = r0; r < r0 + rw && r < rows; ++r) {
        float4 v[2] = {make_float4(1.f, 2.f, 3.f, 4.f), make_float4(5.f, 6.f, 7.f, 8.f)};
#pragma unroll
        for (int t = 0; t < R; ++t) {
            const float4* p = src + t * tstride + r * 128 + lane * 2;
            const float4 a = p[0], b = p[1];
            v[0].x += a.x; v[0].y += a.y; v[0].z += a.z; v[0].w += a.w;
            v[1].x += b.x; v[1].y += b.y; v[1].z += b.z; v[1].w += b.w;
        }
        if (W == 0) { acc.x += v[0].x + v[1].x; acc.y += v[0].y + v[1].y; acc.z += v[0].z + v[1].z; acc.w += v[0].w + v[1].w; }
#pragma unroll
        for (int t = 0; t < W; ++t) {
            float4* p = IL ? dst + (r * W + t) * 128 + lane * 2 : dst + t * tstride + r * 128 + lane * 2;
            p[0] = v[0], p[1] = v[1];
        }
    }
    if (W == 0 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
template <int R, int W, bool IL = false>
static void run(const float4* src, float4* dst, size_t rows, size_t tstride, float* sink, int rw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)(((rows + rw - 1) / rw + 3) / 4);
    for (int i = 0; i < 3; ++i) rw_kernel<R, W, IL><<<grid, 256>>>(src, dst, rows, tstride, sink, rw);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) rw_kernel<R, W, IL><<<grid, 256>>>(src, dst, rows, tstride, sink, rw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double gb = (double)rows * 2048 * (R + W) / 1e9;
    printf("rows per wave %2d  reads %d writes %d%s per row: %7.3f ms  %6.2f GB  %6.2f TB/s\n", rw, R, W, IL ? " (one interleaved tensor)" : "", ms, gb, gb / ms);
}
int main() {
    const size_t rows = 589824;                      // 2048 pairs x 288 frames, 512 channels: 1.2 GB per tensor
    const size_t tstride = rows * 128;               // float4 per tensor
    float4 *src, *dst; float* sink;
    hipMalloc(&src, 3 * tstride * 16); hipMalloc(&dst, 3 * tstride * 16); hipMalloc(&sink, 4);
    hipMemset(src, 0x11, 3 * tstride * 16); hipMemset(dst, 0, 3 * tstride * 16);
    const int rws[] = {16, 1, 4, 8, 16, 32};
    for (int rw : rws) {
        run<1, 0>(src, dst, rows, tstride, sink, rw);
        run<0, 1>(src, dst, rows, tstride, sink, rw);
        run<0, 3>(src, dst, rows, tstride, sink, rw);
        run<1, 1>(src, dst, rows, tstride, sink, rw);
        run<1, 3>(src, dst, rows, tstride, sink, rw);
        run<1, 3, true>(src, dst, rows, tstride, sink, rw);
    }
    return 0;
}
