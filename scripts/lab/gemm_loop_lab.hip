// The K loop of the 256 x 256 split-precision GEMM as a traffic pattern, to compare workgroup shapes before building one:
//   WPS = 2: 8 waves, wave tile 128 x 64 (4 x 2 accumulators): per K step and wave 48 MFMAs, 24 ds_read_b128, 8 LDS-DMA   (the product)
//   WPS = 1: 4 waves, wave tile 128 x 128 (4 x 4 accumulators in the accumulator half): 96 MFMAs, 32 ds_read_b128, 16 LDS-DMA
// Same ring (3 activation + 2 weight stages of 32 KiB), same barrier and counted waits, activations streamed from a buffer much
// larger than the caches, weights from a small (L2-resident) one.  The arithmetic is meaningless; the instruction and memory
// pattern is the kernel's.  Prints s_memtime ticks per K step (3,072 = MFMA-bound).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/gemm_loop_lab.hip -o scripts/lab/gemm_loop_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int STAGE = 32768;
template <int WPS, int READS, int DMA_ON>
__global__ __launch_bounds__(256 * WPS, 1) void loop_probe(unsigned long long* out, const char* act, const char* wgt, int ksteps, size_t act_stride) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NWAVE = 4 * WPS;
    constexpr int NACC = WPS == 1 ? 16 : 8;              // 32 x 32 accumulators per wave
    constexpr int NB = WPS == 1 ? 4 : 2;                 // B blocks per wave (A blocks: 4)
    constexpr int PER = 32 / NWAVE;                      // DMA instructions per wave, operand and K step (1 KiB each)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds);
    const char* a_src = act + (size_t)blockIdx.x * act_stride;          // this workgroup's activation rows (streamed)
    const unsigned voff = (unsigned)lane * 16;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto dma = [&](const char* src, unsigned dst) {
        if (DMA_ON) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(src) : "memory");
    };
    auto issue_a = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) dma(a_src + (size_t)t * STAGE + (wave * PER + i) * 1024, lds_base + buf * STAGE + (wave * PER + i) * 1024);
    };
    auto issue_w = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) dma(wgt + (size_t)(t & 15) * STAGE + (wave * PER + i) * 1024, lds_base + (3 + buf) * STAGE + (wave * PER + i) * 1024);
    };
    // fragment read: row (lane & 31) of a 32-row block, 16-byte chunk swizzled by the row (conflict-free like the product's)
    const int frow = lane & 31, fswz = (frow >> 1) & 7, fh = lane >> 5;
    auto foff = [&](int chunk) { return (unsigned)frow * 128 + (unsigned)((chunk ^ fswz) * 16); };      // chunks 0..3 hi, 4..7 lo of the row's line
    issue_w(0, 0);
    issue_a(0, 0);
    issue_a(1, 1);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ksteps; ++t) {
        if (DMA_ON) {
            if (t + 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");      // A(t+1) may stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + 1 < ksteps) issue_w(t + 1, (t + 1) & 1);
        if (t + 2 < ksteps) issue_a(t + 2, (t + 2) % 3);
        const char* sa = lds + (t % 3) * STAGE + (WPS == 2 ? (wave >> 2) * 16384 : (wave >> 1) * 16384);
        const char* sw = lds + (3 + (t & 1)) * STAGE + (WPS == 2 ? (wave & 3) * 8192 : (wave & 1) * 16384);
#pragma unroll
        for (int s = 0; s < 2; ++s) {                    // two k16 steps
            bf16x8 ah[4], al[4], bh[NB], bl[NB];
            if (READS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(sa + i * 4096 + foff(2 * s + fh));
                    al[i] = *reinterpret_cast<const bf16x8*>(sa + i * 4096 + foff(4 + 2 * s + fh));
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(sw + j * 4096 + foff(2 * s + fh));
                    bl[j] = *reinterpret_cast<const bf16x8*>(sw + j * 4096 + foff(4 + 2 * s + fh));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) ah[i] = al[i] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)t, 1u, 2u, 3u});
#pragma unroll
                for (int j = 0; j < NB; ++j) bh[j] = bl[j] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)t, 5u, 2u, 3u});
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i * NB + j], 0, 0, 0);
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i * NB + j], 0, 0, 0);
                    acc[i * NB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i * NB + j], 0, 0, 0);
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.678f) out[0] = 1;
}

static unsigned long long* d_out;
static char *d_act, *d_wgt;
template <int WPS, int READS, int DMA_ON>
void run(const char* name, int grid, int ksteps) {
    auto k = loop_probe<WPS, READS, DMA_ON>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 3; ++w) k<<<grid, 256 * WPS, 160 * 1024>>>(d_out, d_act, d_wgt, ksteps, (size_t)ksteps * STAGE);
    hipDeviceSynchronize();
    std::vector<unsigned long long> o(grid);
    hipMemcpy(o.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < grid; ++b) c.push_back((double)o[b] / ksteps);
    std::sort(c.begin(), c.end());
    printf("%-64s %7.0f ticks per K step (median of %d workgroups; 3072 = MFMA-bound)\n", name, c[grid / 2], grid);
    fflush(stdout);
}
int main() {
    const int grid = 256, ksteps = 64;
    hipMalloc(&d_out, grid * 8);
    const size_t act_bytes = (size_t)grid * ksteps * STAGE;          // 512 MiB: streamed
    hipMalloc(&d_act, act_bytes);
    hipMalloc(&d_wgt, 16 * STAGE);
    hipMemset(d_act, 0x3c, act_bytes);
    hipMemset(d_wgt, 0x3c, 16 * STAGE);
    run<2, 0, 0>("2 waves/SIMD (4x2): MFMAs only", grid, ksteps);
    run<2, 1, 0>("2 waves/SIMD (4x2): + fragment reads (24 per wave)", grid, ksteps);
    run<2, 1, 1>("2 waves/SIMD (4x2): + reads + LDS-DMA (8 per wave)", grid, ksteps);
    run<1, 0, 0>("1 wave/SIMD (4x4): MFMAs only", grid, ksteps);
    run<1, 1, 0>("1 wave/SIMD (4x4): + fragment reads (32 per wave)", grid, ksteps);
    run<1, 1, 1>("1 wave/SIMD (4x4): + reads + LDS-DMA (16 per wave)", grid, ksteps);
    return 0;
}
