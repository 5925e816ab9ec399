// The MFMA streams of the one-wave-per-SIMD attention kernel in isolation: 48 MFMAs per phase, fragments by ds_read one group
// ahead (S^T phase: 2 x ds_read_b128 per 6 MFMAs, O^T phase: 4 x ds_read_b64_tr_b16 per 6), optionally with LDS-DMA writes
// into the same LDS running alongside.   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/mfma_phase.hip -o scripts/lab/mfma_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// MODE 0: S^T pattern, no LDS reads; 1: S^T pattern with ds_read_b128; 2: O^T pattern (C/D agpr) no reads; 3: O^T with tr reads;
// bit 4 (+16): every wave also issues 8 LDS-DMA (global_load_lds_dwordx4) per phase
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, const unsigned* in, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool READS = (MODE & 1) != 0, OPH = (MODE & 2) != 0, DMA = (MODE & 16) != 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 a[2], b;
    for (int i = 0; i < 4; ++i) {
        a[0][i] = in[threadIdx.x * 4 + i];
        a[1][i] = in[512 + threadIdx.x * 4 + i];
        b[i] = in[1024 + threadIdx.x * 4 + i];
    }
    for (int i = threadIdx.x; i < 32768; i += 256) reinterpret_cast<unsigned*>(lds)[i] = in[i & 2047];
    __syncthreads();
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) c0[i] = c1[i] = 0.f;
    asm volatile("" ::: "a0", "a255");
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\ts_nop 4" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    const unsigned raddr = (unsigned)(lane & 31) * 256 + ((lane >> 5) ^ (lane & 15)) * 16;       // a K-fragment-like address (swizzled)
    const unsigned taddr = (unsigned)(((lane >> 2) & 3) + 4 * (lane >> 5)) * 256 + (lane & 3) * 8 + 32 * ((lane >> 4) & 1);
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds);
    const unsigned voff = lane * 16;
    const char* gsrc = reinterpret_cast<const char*>(in) + wave * 1024;
    asm volatile("s_nop 4");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 n0 = a[0], n1 = a[1];
#pragma unroll
        for (int g = 0; g < 8; ++g) {            // 8 groups of 6 MFMAs
            u32x4 f0 = n0, f1 = n1;
            if (READS && !OPH) {
                asm volatile("ds_read_b128 %0, %2 offset:%c3\n\tds_read_b128 %1, %2 offset:%c4" : "=v"(n0), "=v"(n1) : "v"(raddr), "n"(0), "n"(8192));
            }
            if (READS && OPH) {
                u32x2 t0_, t1_, t2_, t3_;
                asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c5\n\tds_read_b64_tr_b16 %1, %4 offset:%c6\n\tds_read_b64_tr_b16 %2, %4 offset:%c7\n\tds_read_b64_tr_b16 %3, %4 offset:%c8"
                             : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_) : "v"(taddr), "n"(0), "n"(2048), "n"(8192), "n"(10240));
                n0 = u32x4{t0_[0], t0_[1], t1_[0], t1_[1]};
                n1 = u32x4{t2_[0], t2_[1], t3_[0], t3_[1]};
            }
            if (DMA) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base + 65536 + wave * 8192 + g * 1024), "v"(voff), "s"(gsrc) : "memory");
            }
            if (!OPH) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c0) : "v"(f0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c1) : "v"(f0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c0) : "v"(f1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c1) : "v"(f1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c0) : "v"(f1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, a[0:3], %0" : "+v"(c1) : "v"(f1));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(f0), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(f0), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(f1), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(f1), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(f1), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(f1), "v"(b));
            }
        }
        a[0] = n0;
        a[1] = n1;
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = __builtin_bit_cast(float, a[0][0] ^ a[1][1]);
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.678f) out[3] = 1;
}

static unsigned long long* d_out;
static unsigned* d_in;
template <int MODE>
void run(const char* name, int grid) {
    const int iters = 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int w = 0; w < 3; ++w) probe<MODE><<<grid, 256, 160 * 1024>>>(d_out, d_in, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> o(grid);
    hipMemcpy(o.data(), d_out, grid * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc;
    for (int b = 0; b < grid; ++b) cyc.push_back((double)o[b] / (iters * 48.0));
    std::sort(cyc.begin(), cyc.end());
    printf("%-70s %6.1f ticks per MFMA\n", name, cyc[grid / 2]);
    fflush(stdout);
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    hipMalloc(&d_out, grid * 8);
    std::vector<unsigned> h(65536);
    for (auto& x : h) x = 0x3f803f80u ^ (rand() & 0x007f007f);
    hipMalloc(&d_in, 65536 * 4);
    hipMemcpy(d_in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    run<0>("S^T pattern (C/D vgpr, B agpr), no LDS reads", grid);
    run<1>("S^T pattern + 2 ds_read_b128 per 6 MFMAs", grid);
    run<2>("O^T pattern (C/D agpr), no LDS reads", grid);
    run<3>("O^T pattern + 4 ds_read_b64_tr_b16 per 6 MFMAs", grid);
    run<17>("S^T pattern + reads + 8 LDS-DMA per wave and 48 MFMAs", grid);
    run<19>("O^T pattern + reads + 8 LDS-DMA per wave and 48 MFMAs", grid);
    return 0;
}
