// Split-precision conv GEMM, LDS-DMA variant for pair-row activations (see vrd_gemm_x3.hip for the
// arithmetic).  Both operands are already [hi | lo] bf16 planes in HBM, so nothing passes through
// registers on the way in: every operand byte is moved by global_load_lds_dwordx4 (16 B per lane, 1 KiB
// per wave instruction) straight into a 3-stage LDS ring.
//
// Why this shape: the register-staged kernel sits at ~40 % MFMA utilisation because a CU can keep only
// ~64 KiB of operand bytes in flight while a 128 x 128 tile needs 42 B/clk at full MFMA rate.  Here
//   * the tile is 128 x 256 (25 % fewer operand bytes per FLOP),
//   * a stage (K step 32) is 48 KiB and two stages are always in flight behind the one being consumed,
//   * one s_barrier per K step: [wait until stage t landed (counted vmcnt, the newer stage stays in
//     flight)] -> barrier -> issue stage t+2 into the buffer everyone just finished reading -> MFMAs of t.
// 8 waves (2 x 4), each a 64 x 64 sub-tile = 2 x 2 accumulators of 32 x 32.
//
// LDS image: plane tiles are [row][64 B] with NO padding (the DMA writes 64 lanes x 16 B linearly), so the
// 16-byte chunk index is XOR-swizzled with (row >> 2) & 3 on the SOURCE address and again on the fragment
// read; 16 consecutive rows of one logical chunk then land in 16 distinct 16-byte LDS slots.
// Zero padding of the k=3 convolution at sequence ends is a per-lane source pointer to a zero block.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"

namespace {

using vrd::f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int DBN = 256;

// geometry of one K step of DBK bf16 elements for a DBM x 256 tile and an NSTG-stage ring
template <int DBK, int DBM, int NSTG>
struct Geo {
    static constexpr int ROWB = DBK * 2;                      // bytes per tile row
    static constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row (4 or 2)
    static constexpr int RPI = 1024 / ROWB;                   // rows covered by one wave DMA instruction
    static constexpr int RB = 16 / CPR;                       // rows per 256-byte LDS bank row
    static constexpr int A_PLANE = DBM * ROWB;
    static constexpr int W_PLANE = DBN * ROWB;
    static constexpr int STAGE = 2 * A_PLANE + 2 * W_PLANE;   // a_hi | a_lo | w_hi | w_lo
    static constexpr int A_INSTR = DBM / RPI, W_INSTR = DBN / RPI;
    static constexpr int DMA_PER_WAVE = (2 * A_INSTR + 2 * W_INSTR) / 8;
    static constexpr int WM = DBM / 64, WN = 8 / WM;          // 8 waves as WM x WN, each 64 rows x (256/WN) columns
    static constexpr int NJ = (DBN / WN) / 32;                // 32-wide accumulator columns per wave (2 or 4)
    static constexpr size_t LDS = (size_t)NSTG * STAGE;
    __device__ static constexpr int swz(int row) { return (row / RB) % CPR; }
};

__device__ uint4 g_zero_block[4];                     // 64 zero bytes: source of padded taps

template <int TAPS, int DBK, int DBM, int NSTG>
__global__ __launch_bounds__(512) void gemm_bf16x3_dma_kernel(vrd_gemm_args p, int tiles_m, int tiles_n) {
    using G = Geo<DBK, DBM, NSTG>;
    constexpr int NJ = G::NJ;
    constexpr int ROWB = G::ROWB, A_PLANE = G::A_PLANE, W_PLANE = G::W_PLANE, STAGE = G::STAGE;
    constexpr int DMA_PER_WAVE = G::DMA_PER_WAVE, KSUB = DBK / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * DBM;
    const int n0 = tn * DBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / G::WN, wn = wave % G::WN;
    const int li = lane & 31, lh = lane >> 5;
    const int K = p.Cin * TAPS;
    const int nkt = K / DBK;      // K is a multiple of 32 (host check)
    const int PW = p.a_pair_width;
    const char* Whi = reinterpret_cast<const char*>(p.W_split);
    const char* Wlo = Whi + (int64_t)p.N * K * 2;

    // ---- DMA assignment: instruction j = wave*DMA_PER_WAVE + i of the flat list [a_hi | a_lo | w_hi | w_lo]
    const char* gsrc[DMA_PER_WAVE];      // per-lane source row base (bytes), K offset added per stage
    int ldst[DMA_PER_WAVE];              // wave-uniform LDS offset inside a stage
    int tseq[DMA_PER_WAVE];              // A rows: position inside the sequence (k=3 padding)
    bool is_a[DMA_PER_WAVE];
    int lchunk;                          // this lane's logical chunk (bytes) after the source-side swizzle
    {
        const int rin = lane / G::CPR, pch = lane % G::CPR;
        // row blocks start at multiples of RPI (a multiple of RB*CPR), so swz(row) == swz(rin)
        lchunk = (pch ^ G::swz(rin)) * 16;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int j = wave * DMA_PER_WAVE + i;
            if (j < 2 * G::A_INSTR) {                      // activation planes
                const int lo = j / G::A_INSTR, rb = j % G::A_INSTR;
                int64_t r = m0 + rb * G::RPI + rin;
                if (r >= p.M) r = p.M - 1;                 // rows past M are computed on duplicates and dropped
                gsrc[i] = reinterpret_cast<const char*>(p.A + r * p.lda) + (lo ? PW * 2 : 0);
                tseq[i] = (TAPS == 3) ? (int)(r % p.T) : 0;
                ldst[i] = lo * A_PLANE + rb * 1024;
                is_a[i] = true;
            } else {                                       // weight planes
                const int lo = (j - 2 * G::A_INSTR) / G::W_INSTR, rb = (j - 2 * G::A_INSTR) % G::W_INSTR;
                int n = n0 + rb * G::RPI + rin;
                if (n >= p.N) n = p.N - 1;
                gsrc[i] = (lo ? Wlo : Whi) + (int64_t)n * K * 2;
                tseq[i] = 0;
                ldst[i] = 2 * A_PLANE + lo * W_PLANE + rb * 1024;
                is_a[i] = false;
            }
        }
    }
    const char* zero_src = reinterpret_cast<const char*>(g_zero_block);

    auto issue = [&](int kt) {
        const int buf = kt % NSTG;
        const int k0 = kt * DBK;
        int tap = 0, ci0 = k0;
        if (TAPS == 3) {
            tap = (k0 >= p.Cin) + (k0 >= 2 * p.Cin);
            ci0 = k0 - tap * p.Cin;
        }
        const int slab = ci0 / PW;
        // byte offset of this K step inside an activation row (pair rows: slab base + hi-plane offset)
        const int64_t a_off = (int64_t)(tap - (TAPS == 3 ? 1 : 0)) * p.lda * 4 + (int64_t)(2 * slab * PW + (ci0 - slab * PW)) * 2;
        const int64_t w_off = (int64_t)k0 * 2;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const char* src;
            if (is_a[i]) {
                src = gsrc[i] + a_off + lchunk;
                if (TAPS == 3) {
                    const int tt = tseq[i] + tap - 1;
                    if (tt < 0 || tt >= p.T) src = zero_src + lchunk;
                }
            } else {
                src = gsrc[i] + w_off + lchunk;
            }
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(lds + buf * STAGE + ldst[i]), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a stage) for the k16 sub-steps
    int a_rd[2][KSUB], w_rd[NJ][KSUB];   // [mi | nj][s]
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wm * 64 + t * 32 + li;
            a_rd[t][s] = ra * ROWB + (((2 * s + lh) ^ G::swz(ra)) * 16);
        }
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int rw = wn * (32 * NJ) + t * 32 + li;
            w_rd[t][s] = 2 * A_PLANE + rw * ROWB + (((2 * s + lh) ^ G::swz(rw)) * 16);
        }
    }

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // NSTG - 1 stages are kept in flight behind the one being consumed
#pragma unroll
    for (int t = 0; t < NSTG - 1; ++t)
        if (t < nkt) issue(t);
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed once at most the DMAs of the NSTG-2 newer stages are still outstanding
        if (NSTG > 2 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSTG - 1 < nkt) issue(kt + NSTG - 1);
        const char* st = lds + (kt % NSTG) * STAGE;
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            bf16x8 ah[2], al[2], wh[NJ], wl[NJ];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8*>(st + a_rd[t][s]);
                al[t] = *reinterpret_cast<const bf16x8*>(st + A_PLANE + a_rd[t][s]);
            }
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                wh[t] = *reinterpret_cast<const bf16x8*>(st + w_rd[t][s]);
                wl[t] = *reinterpret_cast<const bf16x8*>(st + W_PLANE + w_rd[t][s]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) {
                    acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], wh[nj], acc[mi][nj], 0, 0, 0);
                    acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], wl[nj], acc[mi][nj], 0, 0, 0);
                    acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], wh[nj], acc[mi][nj], 0, 0, 0);
                }
        }
    }
    // every wave must be done with the ring before it is reused as epilogue staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int hn = 0; hn < NJ / 2; ++hn) {      // the epilogue works on 64 x 64 halves of the wave's sub-tile
        f32x16 part[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) part[i][j] = acc[i][2 * hn + j];
        vrd::gemm_epilogue<true>(p, part, smem, m0 + wm * 64, n0 + wn * (32 * NJ) + hn * 64, wave, lane);
    }
}

}  // namespace

namespace vrd {

template <int TAPS, int DBK, int DBM, int NSTG>
static int launch_dma_one(const vrd_gemm_args& a, hipStream_t s) {
    auto kern = gemm_bf16x3_dma_kernel<TAPS, DBK, DBM, NSTG>;
    constexpr size_t lds = Geo<DBK, DBM, NSTG>::LDS;
    static_assert(lds >= 8 * 8192 && lds <= 160 * 1024, "ring must hold the epilogue slabs and fit the CU");
    static bool reserved = false;
    if (!reserved) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("vrd_gemm(bf16x3 dma): cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return -2;
        }
        reserved = true;
    }
    const int tiles_m = (int)((a.M + DBM - 1) / DBM), tiles_n = (a.N + DBN - 1) / DBN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, a, tiles_m, tiles_n);
    return 0;
}

// eligibility: pair-row A whose slab width and Cin are multiples of 32, 16-byte aligned output rows
bool gemm_bf16x3_dma_ok(const vrd_gemm_args& a, bool staged) {
    return staged && a.a_pair_width > 0 && a.a_pair_width % 32 == 0 && a.Cin % 32 == 0 && a.N >= 192;
}

int launch_gemm_bf16x3_dma(const vrd_gemm_args& a, hipStream_t s) {
    // variants (measured with scripts/gemm_bench.py --pair on the path's shapes):
    //   0: 128 x 256 tile, K step 32, 3-stage ring (144 KiB)          -- default for small row counts
    //   1: 128 x 256 tile, K step 16, 3-stage ring (72 KiB, two workgroups per CU)
    //   2: 256 x 256 tile, K step 32, 2-stage ring (128 KiB): a third fewer operand bytes per FLOP
    static const int var_env = [] { const char* e = getenv("VRD_X3_DMA_VARIANT"); return e ? atoi(e) : -1; }();
    int var = var_env;
    if (var < 0) var = 0;
    if (a.taps == 1) {
        if (var == 2) return launch_dma_one<1, 32, 256, 2>(a, s);
        if (var == 1) return launch_dma_one<1, 16, 128, 3>(a, s);
        return launch_dma_one<1, 32, 128, 3>(a, s);
    }
    if (var == 2) return launch_dma_one<3, 32, 256, 2>(a, s);
    if (var == 1) return launch_dma_one<3, 16, 128, 3>(a, s);
    return launch_dma_one<3, 32, 128, 3>(a, s);
}

}  // namespace vrd
