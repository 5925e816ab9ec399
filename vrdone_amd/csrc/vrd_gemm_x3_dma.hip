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
// 8 waves (2 x 4), each a 64 x 64 sub-tile = 4 x 4 accumulators of v_mfma_f32_16x16x32 (vrd_gemm_x3.hip: one K step of 32 per
// instruction, products in the order lo x hi, hi x lo, hi x hi -- the same bits as the other split-precision kernels).
//
// LDS image: a tile row is the 128-byte line [32 hi | 32 lo] of one K step with NO padding (the DMA writes 64 lanes x 16 B
// linearly), so the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 on the SOURCE address and again on the fragment
// read; 16 consecutive rows of one logical chunk then land in 16 distinct 16-byte LDS slots.
// Zero padding of the k=3 convolution at sequence ends is a per-lane source pointer to a zero block.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"
#include <type_traits>

namespace {

using vrd::acc32q;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int DBN = 256, DBM = 128, DBK = 32, NSTG = 3;
constexpr int ROWB = DBK * 4;                       // bytes per tile row: hi and lo of a K step side by side
constexpr int RPI = 1024 / ROWB;                    // rows covered by one wave DMA instruction (8)
constexpr int A_PLANE = DBM * ROWB, W_PLANE = DBN * ROWB, STAGE = A_PLANE + W_PLANE;     // 48 KiB
constexpr int A_INSTR = DBM / RPI, W_INSTR = DBN / RPI, DMA_PER_WAVE = (A_INSTR + W_INSTR) / 8;
constexpr size_t DMA_LDS = (size_t)NSTG * STAGE;
__device__ constexpr int swz(int row) { return (row >> 1) & 7; }

__device__ uint4 g_zero_block[8];                     // 128 zero bytes: source of padded taps

template <int TAPS, bool F16>
__global__ __launch_bounds__(512) void gemm_x3_dma_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, unsigned* rflag) {
    typedef typename vrd::SplitFmt<F16>::x8 e16x8;      // fragment of eight 16-bit elements: bf16, or f16 (VRD_PAIR_F16)
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
    const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves, 64 x 64 each
    const int l15 = lane & 15, l4 = lane >> 4;
    const int K = p.Cin * TAPS;
    const int nkt = K / DBK;      // K is a multiple of 32 (host check)
    const char* Wsp = reinterpret_cast<const char*>(p.W_split);

    // ---- DMA assignment: instruction j = wave*DMA_PER_WAVE + i of the flat list [a | w]
    const char* gsrc[DMA_PER_WAVE];      // per-lane source row base (bytes), K offset added per stage
    int ldst[DMA_PER_WAVE];              // wave-uniform LDS offset inside a stage
    int tseq[DMA_PER_WAVE];              // A rows: position inside the sequence (k=3 padding)
    bool is_a[DMA_PER_WAVE];
    int lchunk[DMA_PER_WAVE];            // this lane's logical chunk (bytes) after the source-side swizzle
    {
        const int rin = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int j = wave * DMA_PER_WAVE + i;
            if (j < A_INSTR) {                             // activation rows
                int64_t r = m0 + j * RPI + rin;
                if (r >= p.M) r = p.M - 1;                 // rows past M are computed on duplicates and dropped
                gsrc[i] = reinterpret_cast<const char*>(p.A + r * p.lda);
                tseq[i] = (TAPS == 3) ? (int)(r % p.T) : 0;
                ldst[i] = j * 1024;
                lchunk[i] = (pch ^ swz(j * RPI + rin)) * 16;
                is_a[i] = true;
            } else {                                       // weight rows
                const int rb = j - A_INSTR;
                int n = n0 + rb * RPI + rin;
                if (n >= p.N) n = p.N - 1;
                gsrc[i] = Wsp + (int64_t)n * K * 4;
                tseq[i] = 0;
                ldst[i] = A_PLANE + rb * 1024;
                lchunk[i] = (pch ^ swz(rb * RPI + rin)) * 16;
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
        // byte offset of this K step inside an activation row
        const int64_t a_off = (int64_t)(tap - (TAPS == 3 ? 1 : 0)) * p.lda * 4 + (int64_t)ci0 * 4;
        const int64_t w_off = (int64_t)k0 * 4;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const char* src;
            if (is_a[i]) {
                src = gsrc[i] + a_off + lchunk[i];
                if (TAPS == 3) {
                    const int tt = tseq[i] + tap - 1;
                    if (tt < 0 || tt >= p.T) src = zero_src + lchunk[i];
                }
            } else {
                src = gsrc[i] + w_off + lchunk[i];
            }
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(lds + buf * STAGE + ldst[i]), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a stage): lane (l15, l4) reads row l15 of a 16-row block, hi chunk l4 (k = 8 l4 ..
    // + 7 of the K step) and the lo chunk behind it (chunk index + 4: the swizzle XORs the low three bits, so XOR 64 bytes)
    const int ra0 = wm * 64 + l15, rw0 = wn * 64 + l15;
    const int a_rd = ra0 * ROWB + ((l4 ^ swz(ra0)) * 16);                 // 16-row block t: + t * 16 * ROWB (same swizzle)
    const int w_rd = A_PLANE + rw0 * ROWB + ((l4 ^ swz(rw0)) * 16);

    acc32q acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) vrd::acc_clear(acc[i][j]);

    // NSTG - 1 stages are kept in flight behind the one being consumed
#pragma unroll
    for (int t = 0; t < NSTG - 1; ++t)
        if (t < nkt) issue(t);
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed once at most the DMAs of the NSTG-2 newer stages are still outstanding
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NSTG - 1 < nkt) issue(kt + NSTG - 1);
        const char* st = lds + (kt % NSTG) * STAGE;
        e16x8 wh[4], wl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wh[t] = *reinterpret_cast<const e16x8*>(st + w_rd + t * 16 * ROWB);
            wl[t] = *reinterpret_cast<const e16x8*>(st + (w_rd ^ 64) + t * 16 * ROWB);
        }
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            const e16x8 ah = *reinterpret_cast<const e16x8*>(st + a_rd + bi * 16 * ROWB);
            const e16x8 al = *reinterpret_cast<const e16x8*>(st + (a_rd ^ 64) + bi * 16 * ROWB);
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int bj = 0; bj < 4; ++bj) {
                    vrd::f32x4_t& c = acc[bi >> 1][bj >> 1].b[bi & 1][bj & 1];
                    c = vrd::mfma16(pr == 0 ? al : ah, pr == 1 ? wl[bj] : wh[bj], c);
                }
        }
    }
    // every wave must be done with the ring before it is reused as epilogue staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vrd::gemm_epilogue<true, 64>(p, acc, smem, m0 + wm * 64, n0 + wn * 64, wave, lane, rflag);
}

}  // namespace

namespace vrd {

template <int TAPS, bool F16>
static int launch_dma_one(const vrd_gemm_args& a, hipStream_t s) {
    auto kern = gemm_x3_dma_kernel<TAPS, F16>;
    static_assert(DMA_LDS >= 8 * 16384 && DMA_LDS <= 160 * 1024, "ring must hold the epilogue slabs and fit the CU");
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), DMA_LDS, "vrd_gemm(bf16x3 dma)")) return rc;
    const int tiles_m = (int)((a.M + DBM - 1) / DBM), tiles_n = (a.N + DBN - 1) / DBN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), DMA_LDS, s, a, tiles_m, tiles_n, a.c_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    return 0;
}

// eligibility: pair-row A whose slab width and Cin are multiples of 32, 16-byte aligned output rows
bool gemm_x3_dma_ok(const vrd_gemm_args& a, bool staged) {
    return staged && a.a_pair_width > 0 && a.N >= 192;
}

// (other schedules of this tile -- ping-pong wave groups, dedicated producer waves -- were measured in rounds 1-2 and not kept:
// LABNOTES.md; their sources are in the history of this file)
int launch_gemm_x3_dma(const vrd_gemm_args& a, hipStream_t s) {
    if (a.split_fmt == VRD_PAIR_F16) return a.taps == 1 ? launch_dma_one<1, true>(a, s) : launch_dma_one<3, true>(a, s);
    return a.taps == 1 ? launch_dma_one<1, false>(a, s) : launch_dma_one<3, false>(a, s);
}

}  // namespace vrd
