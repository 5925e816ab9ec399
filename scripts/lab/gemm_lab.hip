// Lab harness (not part of the product): runs the LDS-DMA split-precision GEMM on the path's shapes with
// per-workgroup cycle stamps.  Build:  hipcc --offload-arch=gfx950 -O3 -DVRD_LAB_STAMP gemm_lab.hip -o gemm_lab
// Timing only: operands are arbitrary bf16 bit patterns.
#include "../../vrdone_amd/csrc/vrd_runtime.hip"
#include "../../vrdone_amd/csrc/vrd_gemm_x3_dma.hip"
#include "../../vrdone_amd/csrc/vrd_gemm_x3_big.hip"
#include "vrd_gemm_x3_row.hip"
#include <algorithm>
#include <vector>

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
    for (const Shape& sh : shapes) {
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
        vrd_gemm_args a = {};
        a.A = A; a.lda = sh.Cin; a.W = nullptr; a.bias = bias; a.C = C; a.ldc = sh.N; a.M = sh.M; a.N = sh.N; a.Cin = sh.Cin;
        a.taps = sh.taps; a.T = T; a.act = 0; a.W_split = (const uint16_t*)W; a.a_pair_width = sh.Cin; a.c_pair = 0;
        // GEMM_LAB_VARS=11: only the product's 256 x 256 kernel (e.g. for the -DVRD_LAB_VALU=... synthetic-load builds)
        const char* vars_env = getenv("GEMM_LAB_VARS");
        std::vector<int> vars = {11, 12, 14};
        if (vars_env) vars = {atoi(vars_env)};
        for (int var : vars) {
            char env[8];
            snprintf(env, sizeof env, "%d", var);
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                int rc = var == 14 ? vrd::launch_gemm_x3_row_nw(&a, vrd::GemmBatch{}, 1, 4, 0) : var == 12 ? vrd::launch_gemm_x3_row_nw(&a, vrd::GemmBatch{}, 1, 8, 0) : var == 11 ? vrd::launch_gemm_x3_big(a, 0) : vrd::launch_gemm_x3_dma_variant(a, 0, var);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                if (rc) { printf("launch failed: %s\n", vrd_last_error()); return 1; }
                hipEventElapsedTime(&ms, e0, e1);
            }
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
