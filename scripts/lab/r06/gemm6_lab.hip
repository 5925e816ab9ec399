// Lab harness, round 6 (not part of the product; r05/gemm5_lab.hip + a sustained-clock mode: GEMM_LAB_SUSTAIN_MS of
// back-to-back launches before the timed ones, GEMM_LAB_SHAPES = bit mask of the shapes to run): the 256 x 256 split-precision GEMM alone, per-tile cycle stamps,
// on the path's shapes; GEMM_LAB_F16=1 runs the product's default format on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVRD_LAB_STAMP [-DVRD_BIG_BUFDMA=1] scripts/lab/r06/gemm6_lab.hip -o ...
#include "../../../vrdone_amd/csrc/vrd_runtime.hip"
namespace vrd { double take_f32_skipped_flops() { return 0.0; } }      // (vrd_gemm.hip is not part of this harness)
__device__ int g_lab_mode;
__device__ unsigned long long g_lab[8 * 65536];
__device__ unsigned long long g_lab_phase[16 * 4096];   // [tile][group][5 phase accumulators]
#define LAB_STAMP(slot)
#define LAB_REAL(slot)
#define LAB_PHASE_DECL unsigned long long lab_prev = __builtin_amdgcn_s_memtime(), lab_acc[5] = {0, 0, 0, 0, 0}
#define LAB_PHASE(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        lab_acc[i] += now_ - lab_prev;                                \
        lab_prev = now_;                                              \
    } while (0)
#define LAB_PHASE_FLUSH(grp)
#include "../../../vrdone_amd/csrc/vrd_gemm_x3_big.hip"
#include <algorithm>
#include <vector>
// random pair rows in the scaled-f16 format: hi = f16(16 x), lo = f16(16 x - hi), x ~ roughly N(0, 1) (sum of four uniforms)
__global__ void fill_pair_f16(uint32_t* dst, size_t n_pairs_of_channels, float scale, uint32_t seed) {
    // dst viewed as rows of 32-channel blocks [32 hi | 32 lo]; thread i fills 16-bit slots for channel pair (2i, 2i+1)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_pairs_of_channels; i += (size_t)gridDim.x * blockDim.x) {
        const size_t blk = i / 16, w = i % 16;          // block of 32 channels, dword inside its hi half
        _Float16 h[2], l[2];
        for (int j = 0; j < 2; ++j) {
            uint32_t z = (uint32_t)(i * 2 + j) * 2654435761u + seed;
            float u = 0.f;
            for (int r = 0; r < 4; ++r) {
                z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
                u += (float)(z >> 8) * (1.0f / 16777216.0f) - 0.5f;
            }
            const float y = u * 1.732f * scale * 16.f;
            h[j] = (_Float16)y;
            l[j] = (_Float16)(y - (float)h[j]);
        }
        uint32_t hw, lw;
        __builtin_memcpy(&hw, h, 4);
        __builtin_memcpy(&lw, l, 4);
        dst[blk * 32 + w] = hw;
        dst[blk * 32 + 16 + w] = lw;
    }
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : v[v.size() / 2];
}

int main(int argc, char** argv) {
    const int lab_mode = argc > 1 ? atoi(argv[1]) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_lab_mode), &lab_mode, sizeof lab_mode);
    printf("lab mode %d\n", lab_mode);
    struct Shape { int64_t M; int N, Cin, taps; const char* label; };
    const Shape shapes[] = {{147456, 512, 512, 1, "qkv/proj"}, {147456, 2048, 512, 1, "mlp up"}, {147456, 512, 2048, 1, "mlp down"},
                            {147456, 512, 1024, 3, "embd k3"}, {589824, 512, 512, 1, "qkv/proj chunk1024"}};
    const int T = 288;
    const int shape_mask = getenv("GEMM_LAB_SHAPES") ? atoi(getenv("GEMM_LAB_SHAPES")) : 31;
    const int sustain_ms = getenv("GEMM_LAB_SUSTAIN_MS") ? atoi(getenv("GEMM_LAB_SUSTAIN_MS")) : 0;
    int shape_i = -1;
    for (const Shape& sh : shapes) {
        if (!((shape_mask >> ++shape_i) & 1)) continue;
        const int K = sh.Cin * sh.taps;
        float *A, *C, *bias;
        void* W;
        hipMalloc(&A, (size_t)sh.M * sh.Cin * 4);
        hipMalloc(&C, (size_t)sh.M * sh.N * 4);
        hipMalloc(&W, (size_t)sh.N * K * 4);
        hipMalloc(&bias, sh.N * 4);
        hipMemset(A, 0x3c, (size_t)sh.M * sh.Cin * 4);
        hipMemset(W, 0x3c, (size_t)sh.N * K * 4);
        hipMemset(bias, 0, sh.N * 4);
        // GEMM_LAB_F16=1: the product's default format (scaled f16 planes) on RANDOM data -- constant bit patterns let the chip
        // hold a higher clock than real operands do (MI355X_MICROARCH.md, DVFS give-back)
        static const bool lab_f16 = getenv("GEMM_LAB_F16") && atoi(getenv("GEMM_LAB_F16"));
        float* w_scale = nullptr;
        if (lab_f16) {
            fill_pair_f16<<<4096, 256>>>((uint32_t*)A, (size_t)sh.M * sh.Cin / 2, 1.0f, 17u);
            fill_pair_f16<<<4096, 256>>>((uint32_t*)W, (size_t)sh.N * K / 2, 64.0f, 99u);       // weights: max |w| 2^e in [2^14, 2^15)
            hipMalloc(&w_scale, 4);
            const float al = 1.0f / (16.f * 1024.f);
            hipMemcpy(w_scale, &al, 4, hipMemcpyHostToDevice);
        }
        vrd_gemm_args a = {};
        a.A = A; a.lda = sh.Cin; a.W = nullptr; a.bias = bias; a.C = C; a.ldc = sh.N; a.M = sh.M; a.N = sh.N; a.Cin = sh.Cin;
        a.taps = sh.taps; a.T = T; a.act = 0; a.W_split = (const uint16_t*)W; a.a_pair_width = sh.Cin; a.c_pair = 0;
        if (lab_f16) { a.split_fmt = VRD_PAIR_F16; a.w_scale = w_scale; }
        // GEMM_LAB_VARS=11: only the product's 256 x 256 kernel (e.g. for the -DVRD_LAB_VALU=... synthetic-load builds)
        const char* vars_env = getenv("GEMM_LAB_VARS");
        std::vector<int> vars = {11};
        if (vars_env) vars = {atoi(vars_env)};
        for (int var : vars) {
            char env[8];
            snprintf(env, sizeof env, "%d", var);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float ms = 0;
            if (sustain_ms) {          // back-to-back launches until the clock has settled under load
                hipEventRecord(e0);
                for (;;) {
                    for (int i = 0; i < 50; ++i) vrd::launch_gemm_x3_big(a, 0);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms >= sustain_ms) break;
                }
            }
            const int reps = sustain_ms ? 20 : 3;
            float best = 0, sum = 0;
            for (int rep = 0; rep < reps; ++rep) {
                hipEventRecord(e0);
                int rc = vrd::launch_gemm_x3_big(a, 0);
                hipEventRecord(e1);
                if (rc) { printf("launch failed: %s\n", vrd_last_error()); return 1; }
                if (!sustain_ms || rep == reps - 1) hipEventSynchronize(e1);
                if (!sustain_ms) hipEventElapsedTime(&ms, e0, e1);
            }
            if (sustain_ms) {          // the average of 20 more launches, no host wait between them
                hipEventRecord(e0);
                for (int rep = 0; rep < 20; ++rep) vrd::launch_gemm_x3_big(a, 0);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                ms /= 20;
            }
            (void)best; (void)sum;
            const int tm_rows = var == 14 ? 128 : var >= 11 ? 256 : 128;
            const int tiles = (int)((sh.M + tm_rows - 1) / tm_rows) * ((sh.N + 255) / 256);

            const int n = std::min(tiles, 65536);
            std::vector<unsigned long long> st((size_t)n * 8);
            hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lab), st.size() * 8);
            std::vector<double> pro, loop, epi, tot, clk;
            for (int i = 0; i < n; ++i) {
                const unsigned long long* s = &st[(size_t)i * 8];
                pro.push_back((double)(s[1] - s[0]));
                loop.push_back((double)(s[2] - s[1]));
                epi.push_back((double)(s[3] - s[2]));
                tot.push_back((double)(s[3] - s[0]));
                clk.push_back((double)(s[3] - s[0]) / ((double)(s[5] - s[4]) * 10.0));   // cycles per ns (realtime = 100 MHz)
            }
            const double nkt = K / 32.0, ghz = median(clk);
            printf("%-20s K=%5d N=%4d var %d: %7.3f ms  %6.1f TF/s | per tile (wave 0): setup %6.0f  loop %7.0f (%5.0f/kstep)  epilogue %6.0f  total %7.0f cyc @ %.2f GHz | tiles/CU %.1f -> busy %.3f ms\n",
                   sh.label, K, sh.N, var, ms, 2.0 * sh.M * sh.N * K / ms / 1e9, median(pro), median(loop), median(loop) / nkt, median(epi),
                   median(tot), ghz, tiles / 256.0, tiles / 256.0 * median(tot) / ghz * 1e-6);
            {
                std::vector<double> t6, t7;
                for (int i = 0; i < n; ++i) {
                    const unsigned long long* s_ = &st[(size_t)i * 8];
                    t6.push_back((double)(s_[6] - s_[1]));
                    t7.push_back((double)(s_[7] - s_[6]));
                }
                printf("      tile start: first-stage wait (vmcnt) %6.0f, barrier %6.0f\n", median(t6), median(t7));
            }
            if (var == 12 || var == 14) {
                std::vector<double> t6, t7, tb;
                for (int i = 0; i < n; ++i) {
                    const unsigned long long* s_ = &st[(size_t)i * 8];
                    t6.push_back((double)(s_[6] - s_[0]));
                    t7.push_back((double)(s_[7] - s_[6]));
                    tb.push_back((double)(s_[1] - s_[7]));
                }
                printf("      row kernel setup: start -> loads issued %6.0f, loads -> A(0) W(0) landed %6.0f, barrier + first fragments %6.0f\n",
                       median(t6), median(t7), median(tb));
            }
            if (var == 0 || var == 11) {
                std::vector<unsigned long long> ph((size_t)4096 * 16);
                hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_lab_phase), ph.size() * 8);
                const char* names[4] = {"vmwait", "barrier", "dma issue", "reads+mfma"};
                for (int g = 0; g < 2; ++g) {
                    printf("      waves %d per kstep:", g * 4);
                    for (int i = 0; i < 4; ++i) {
                        std::vector<double> v;
                        for (int w = 0; w < std::min(tiles, 4096); ++w) v.push_back((double)ph[(size_t)w * 16 + g * 8 + i] / nkt);
                        printf("  %s %5.0f", names[i], median(v));
                    }
                    printf("\n");
                }
            }
            if (var >= 6) {
                std::vector<unsigned long long> ph((size_t)4096 * 16);
                hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_lab_phase), ph.size() * 8);
                const char* names[2][3] = {{"half0+wait", "barrier", "half1"}, {"vmwait", "barrier", "issue"}};
                for (int g = 0; g < 2; ++g) {
                    printf("      %s per kstep:", g ? "producer 0" : "consumer 0");
                    for (int i = 0; i < 3; ++i) {
                        std::vector<double> v;
                        for (int w = 0; w < std::min(tiles, 4096); ++w) v.push_back((double)ph[(size_t)w * 16 + g * 8 + i] / nkt);
                        printf("  %s %5.0f", names[g][i], median(v));
                    }
                    printf("\n");
                }
            }
            if (var == 3) {
                std::vector<unsigned long long> ph((size_t)4096 * 16);
                hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_lab_phase), ph.size() * 8);
                const char* names[5] = {"load", "bar1", "mfma", "vmwait", "bar2"};
                for (int g = 0; g < 2; ++g) {
                    printf("      group %d per kstep:", g);
                    for (int i = 0; i < 5; ++i) {
                        std::vector<double> v;
                        for (int w = 0; w < std::min(tiles, 4096); ++w) v.push_back((double)ph[(size_t)w * 16 + g * 8 + i] / nkt);
                        printf("  %s %5.0f", names[i], median(v));
                    }
                    printf("\n");
                }
            }
            fflush(stdout);
        }
        hipFree(A); hipFree(C); hipFree(W); hipFree(bias);
    }
    return 0;
}
