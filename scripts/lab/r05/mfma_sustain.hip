// Lab probe (round 5, not part of the product): what the chip SUSTAINS on the split-precision GEMM's MFMA stream.
// The 256 x 256 kernel's K step per wave is 48 v_mfma_f32_32x32x16_f16 (a_lo*w_hi, a_hi*w_lo, a_hi*w_hi on 4 x 2 accumulators,
// two k16 halves) next to 24 ds_read_b128 of fragments; the chip lowers its clock under that load (MI355X_MICROARCH.md,
// "DVFS give-back"), so the roofline fraction quoted against the nominal 2.5 PFLOP/s hides how much of the gap is the clock.
// This probe runs exactly that instruction mix on random pair-row data (hi = f16(16 x), lo = f16(16 x - hi), x ~ N(0, 1)),
// two waves per SIMD on every CU, back to back for ~2.5 s, and reports MFMA TFLOP/s by event time over the last second and the
// in-kernel clock (s_memtime / s_memrealtime), for
//   shape 32: v_mfma_f32_32x32x16_f16      shape 16: v_mfma_f32_16x16x32_f16 (same output tile per wave, same operand bytes)
//   lds 0: operands stay in registers       lds 1: every fragment re-read from LDS each K step (24 ds_read_b128 per wave)
// hipcc --offload-arch=gfx950 -O3 scripts/lab/r05/mfma_sustain.hip -o scripts/lab/r05/mfma_sustain
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

constexpr int ROWB = 128;      // one K step of a tile row: [32 hi | 32 lo] f16

// LDS image: 256 A rows + 256 W rows of one K step (64 KiB), filled once from `img`
template <int SHAPE, int LDS>
__global__ __launch_bounds__(512) void sustain(const uint4* __restrict__ img, float* __restrict__ sink, int iters,
                                               unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = img[i];
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5, l15 = lane & 15, l4 = lane >> 4;
    auto swz = [](int row) { return (row >> 1) & 7; };
    const int a_base = SHAPE == 16 ? (wm * 128 + l15) * ROWB + ((l4 ^ swz(l15)) * 16) : (wm * 128 + li) * ROWB + ((lh ^ swz(li)) * 16);
    const int w_base = 32768 + (SHAPE == 16 ? (wn * 64 + l15) * ROWB + ((l4 ^ swz(l15)) * 16) : (wn * 64 + li) * ROWB + ((lh ^ swz(li)) * 16));
    f32x16 acc[4][2];
    f32x4 acc16[8][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if (SHAPE == 32) {
        // fragments of one K step: A (s, mi) hi / lo, W (s, nj) hi / lo
        f16x8 ah[2][4], al[2][4], wh[2][2], wl[2][2];
        auto load_all = [&](int flip) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int off = ((a_base ^ (s * 32)) + mi * 32 * ROWB) ^ flip;
                    ah[s][mi] = *reinterpret_cast<const f16x8*>(lds + off);
                    al[s][mi] = *reinterpret_cast<const f16x8*>(lds + (off ^ 64));
                }
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    const int off = ((w_base ^ (s * 32)) + nj * 32 * ROWB) ^ flip;
                    wh[s][nj] = *reinterpret_cast<const f16x8*>(lds + off);
                    wl[s][nj] = *reinterpret_cast<const f16x8*>(lds + (off ^ 64));
                }
            }
        };
        load_all(0);
        t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            // (flip moves the reads to the other k16 half: an address the compiler cannot hoist; same bank pattern)
            if (LDS) load_all((it & 1) * 32);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj) {
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s][mi], wh[s][nj], acc[mi][nj], 0, 0, 0);
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], wl[s][nj], acc[mi][nj], 0, 0, 0);
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s][mi], wh[s][nj], acc[mi][nj], 0, 0, 0);
                    }
        }
    } else {
        // 16x16x32: lane holds row l & 15, k = 8 * (l >> 4) .. +7 of the whole 32-channel step
        f16x8 ah[8], al[8], wh[4], wl[4];
        auto load_all = [&](int flip) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int off = (a_base + mi * 16 * ROWB) ^ flip;
                ah[mi] = *reinterpret_cast<const f16x8*>(lds + off);
                al[mi] = *reinterpret_cast<const f16x8*>(lds + (off ^ 64));
            }
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
                const int off = (w_base + nj * 16 * ROWB) ^ flip;
                wh[nj] = *reinterpret_cast<const f16x8*>(lds + off);
                wl[nj] = *reinterpret_cast<const f16x8*>(lds + (off ^ 64));
            }
        };
        load_all(0);
        t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            if (LDS) load_all((it & 1) * 32);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int nj = 0; nj < 4; ++nj) {
                    acc16[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], wh[nj], acc16[mi][nj], 0, 0, 0);
                    acc16[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wl[nj], acc16[mi][nj], 0, 0, 0);
                    acc16[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wh[nj], acc16[mi][nj], 0, 0, 0);
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 4; ++e) s += acc16[i][j][e];
    if (s == 12345.678f) sink[0] = s;       // never true: keeps the accumulators alive
    if (tid == 0) {
        stamps[blockIdx.x * 2 + 0] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

int main(int argc, char** argv) {
    const int zero_data = argc > 1 ? atoi(argv[1]) : 0;
    const int grid = 256;
    const int iters = 4096;                  // 4096 K steps of 48 (or 96) MFMAs per wave per launch: ~8 ms
    std::vector<_Float16> h(65536 / 2);
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t row = 0; row < 512; ++row)
        for (int c = 0; c < 32; ++c) {
            const float y = zero_data ? 0.f : nd(rng) * 16.f;
            const _Float16 hi = (_Float16)y;
            h[row * 64 + c] = hi;
            h[row * 64 + 32 + c] = (_Float16)(y - (float)hi);
        }
    uint4* d_img;
    float* d_sink;
    unsigned long long* d_st;
    hipMalloc(&d_img, 65536);
    hipMalloc(&d_sink, 64);
    hipMalloc(&d_st, grid * 16);
    hipMemcpy(d_img, h.data(), 65536, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Var { int shape, lds; const char* name; };
    const Var vars[] = {{32, 0, "32x32x16, operands in registers"}, {16, 0, "16x16x32, operands in registers"},
                        {32, 1, "32x32x16, 24 ds_read_b128 per K step"}, {16, 1, "16x16x32, 24 ds_read_b128 per K step"}};
    printf("data: %s; grid %d x 512 threads (two waves per SIMD), %d K steps per launch\n", zero_data ? "zeros" : "random pair rows (f16 hi/lo of 16 x, x ~ N(0,1))", grid, iters);
    for (int rep = 0; rep < 2; ++rep)
    for (const Var& v : vars) {
        auto launch = [&] {
            if (v.shape == 32 && !v.lds) hipLaunchKernelGGL((sustain<32, 0>), dim3(grid), dim3(512), 65536, 0, d_img, d_sink, iters, d_st);
            if (v.shape == 16 && !v.lds) hipLaunchKernelGGL((sustain<16, 0>), dim3(grid), dim3(512), 65536, 0, d_img, d_sink, iters, d_st);
            if (v.shape == 32 && v.lds) hipLaunchKernelGGL((sustain<32, 1>), dim3(grid), dim3(512), 65536, 0, d_img, d_sink, iters, d_st);
            if (v.shape == 16 && v.lds) hipLaunchKernelGGL((sustain<16, 1>), dim3(grid), dim3(512), 65536, 0, d_img, d_sink, iters, d_st);
        };
        // ~1.5 s of warm-up at load, then ~1 s timed
        const int warm = 180, timed = 120;
        for (int i = 0; i < warm; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < timed; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st(grid * 2);
        hipMemcpy(st.data(), d_st, grid * 16, hipMemcpyDeviceToHost);
        std::vector<double> clk, cyc;
        for (int b = 0; b < grid; ++b) {
            clk.push_back((double)st[b * 2] / (double)st[b * 2 + 1] * 0.1);          // GHz (realtime = 100 MHz)
            cyc.push_back((double)st[b * 2] / iters);
        }
        std::sort(clk.begin(), clk.end());
        std::sort(cyc.begin(), cyc.end());
        // per K step and CU: 8 waves x 48 MFMAs x 32*32*16*2 FLOP
        const double flop = (double)timed * iters * grid * 8.0 * 48.0 * 32768.0;
        printf("%-40s %7.1f MFMA TFLOP/s (= %5.1f algorithmic, %.3f of 2,500)  %6.0f cycles per K step (3,072 = MFMA-bound)  clock %.3f GHz  [%.1f ms per launch]\n",
               v.name, flop / ms / 1e9, flop / ms / 1e9 / 3.0, flop / ms / 1e9 / 2500.0, cyc[grid / 2], clk[grid / 2], ms / timed);
        fflush(stdout);
    }
    return 0;
}
