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
#include <type_traits>

namespace {

using vrd::f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int DBN = 256;

// geometry of one K step of DBK bf16 elements for a DBM x 256 tile and an NSTG-stage ring
template <int DBK, int DBM, int NSTG, bool BLK = false>
struct Geo {
    static constexpr int ROWB = BLK ? DBK * 4 : DBK * 2;      // bytes per tile row (blocked: hi and lo of a K step side by side)
    static constexpr int CPR = ROWB / 16;                     // 16-byte chunks per row (4 or 2)
    static constexpr int RPI = 1024 / ROWB;                   // rows covered by one wave DMA instruction
    static constexpr int RB = 16 / CPR;                       // rows per 256-byte LDS bank row
    static constexpr int A_PLANE = DBM * ROWB;
    static constexpr int W_PLANE = DBN * ROWB;
    static constexpr int NPL = BLK ? 1 : 2;                   // LDS images per operand
    static constexpr int STAGE = NPL * (A_PLANE + W_PLANE);   // a_hi | a_lo | w_hi | w_lo   (blocked: a | w)
    static constexpr int A_INSTR = DBM / RPI, W_INSTR = DBN / RPI;
    static constexpr int DMA_PER_WAVE = NPL * (A_INSTR + W_INSTR) / 8;
    static constexpr int WM = DBM / 64, WN = 8 / WM;          // 8 waves as WM x WN, each 64 rows x (256/WN) columns
    static constexpr int NJ = (DBN / WN) / 32;                // 32-wide accumulator columns per wave (2 or 4)
    static constexpr size_t LDS = (size_t)NSTG * STAGE;
    __device__ static constexpr int swz(int row) { return (row / RB) % CPR; }
};

__device__ uint4 g_zero_block[8];                     // 128 zero bytes: source of padded taps

#ifdef VRD_LAB_STAMP   // scripts/lab/gemm_lab.hip only: per-workgroup cycle stamps (never compiled into the library)
__device__ int g_lab_mode;      // bit 0: no MFMAs, bit 1: no A DMAs, bit 2: no W DMAs (timing experiments)
#define LAB_MODE_DECL const int lab_mode_ = __builtin_amdgcn_readfirstlane(g_lab_mode)
#define LAB_MODE(bit) (lab_mode_ & (bit))
__device__ unsigned long long g_lab[8 * 65536];
__device__ unsigned long long g_lab_phase[16 * 4096];   // [wg][group][5 phase accumulators]
#define LAB_STAMP(slot)                                                                            \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 65536) g_lab[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define LAB_REAL(slot)                                                                             \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 65536) g_lab[blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define LAB_PHASE_DECL unsigned long long lab_prev = __builtin_amdgcn_s_memtime(), lab_acc[5] = {0, 0, 0, 0, 0}
#define LAB_PHASE(i)                                                  \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        lab_acc[i] += now_ - lab_prev;                                \
        lab_prev = now_;                                              \
    } while (0)
#define LAB_PHASE_FLUSH(grp)                                                                  \
    do {                                                                                      \
        if ((threadIdx.x & 255) == 0 && blockIdx.x < 4096)                                    \
            for (int i_ = 0; i_ < 5; ++i_) g_lab_phase[blockIdx.x * 16 + (grp) * 8 + i_] = lab_acc[i_]; \
    } while (0)
#define LAB_PHASE_FLUSH2(grp)                                                                 \
    do {                                                                                      \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                     \
            for (int i_ = 0; i_ < 5; ++i_) g_lab_phase[blockIdx.x * 16 + (grp) * 8 + i_] = lab_acc[i_]; \
    } while (0)
#else
#define LAB_PHASE_FLUSH2(grp)
#define LAB_MODE_DECL
#define LAB_MODE(bit) 0
#define LAB_STAMP(slot)
#define LAB_REAL(slot)
#define LAB_PHASE_DECL
#define LAB_PHASE(i)
#define LAB_PHASE_FLUSH(grp)
#endif

template <int TAPS, int DBK, int DBM, int NSTG, bool PP, bool BLK, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_x3_dma_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, unsigned* rflag) {
    typedef typename vrd::SplitFmt<F16>::x8 bf16x8;      // fragment of eight 16-bit elements: bf16, or f16 (VRD_PAIR_F16)
    using G = Geo<DBK, DBM, NSTG, BLK>;
    constexpr int NJ = G::NJ;
    constexpr int ROWB = G::ROWB, A_PLANE = G::A_PLANE, W_PLANE = G::W_PLANE, STAGE = G::STAGE;
    constexpr int DMA_PER_WAVE = G::DMA_PER_WAVE, KSUB = DBK / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    LAB_STAMP(0);
    LAB_REAL(4);

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
    int lchunk[DMA_PER_WAVE];            // this lane's logical chunk (bytes) after the source-side swizzle
    {
        const int rin = lane / G::CPR, pch = lane % G::CPR;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int j = wave * DMA_PER_WAVE + i;
            if (j < G::NPL * G::A_INSTR) {                 // activation planes
                const int lo = j / G::A_INSTR, rb = j % G::A_INSTR;
                int64_t r = m0 + rb * G::RPI + rin;
                if (r >= p.M) r = p.M - 1;                 // rows past M are computed on duplicates and dropped
                gsrc[i] = reinterpret_cast<const char*>(p.A + r * p.lda) + (lo ? PW * 2 : 0);   // (blocked: lo == 0)
                tseq[i] = (TAPS == 3) ? (int)(r % p.T) : 0;
                ldst[i] = lo * A_PLANE + rb * 1024;
                lchunk[i] = (pch ^ G::swz(rb * G::RPI + rin)) * 16;
                is_a[i] = true;
            } else {                                       // weight planes
                const int jw = j - G::NPL * G::A_INSTR;
                const int lo = jw / G::W_INSTR, rb = jw % G::W_INSTR;
                int n = n0 + rb * G::RPI + rin;
                if (n >= p.N) n = p.N - 1;
                gsrc[i] = BLK ? Whi + (int64_t)n * K * 4 : (lo ? Wlo : Whi) + (int64_t)n * K * 2;
                tseq[i] = 0;
                ldst[i] = G::NPL * A_PLANE + lo * W_PLANE + rb * 1024;
                lchunk[i] = (pch ^ G::swz(rb * G::RPI + rin)) * 16;
                is_a[i] = false;
            }
        }
    }
    const char* zero_src = reinterpret_cast<const char*>(g_zero_block);

    // DMA instructions [i0, i1) of this wave's share of stage kt
    auto issue_part = [&](int kt, int i0, int i1) {
        const int buf = kt % NSTG;
        const int k0 = kt * DBK;
        int tap = 0, ci0 = k0;
        if (TAPS == 3) {
            tap = (k0 >= p.Cin) + (k0 >= 2 * p.Cin);
            ci0 = k0 - tap * p.Cin;
        }
        const int slab = ci0 / PW;
        // byte offset of this K step inside an activation row (pair rows: slab base + hi-plane offset)
        const int64_t a_off = (int64_t)(tap - (TAPS == 3 ? 1 : 0)) * p.lda * 4 +
                              (BLK ? (int64_t)ci0 * 4 : (int64_t)(2 * slab * PW + (ci0 - slab * PW)) * 2);
        const int64_t w_off = BLK ? (int64_t)k0 * 4 : (int64_t)k0 * 2;
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            if (i < i0 || i >= i1) continue;
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
    auto issue = [&](int kt) { issue_part(kt, 0, DMA_PER_WAVE); };

    // ---- fragment read offsets (bytes inside a stage) for the k16 sub-steps
    // blocked rows hold the hi half of the K step in chunks 0 .. DBK/8-1 and the lo half behind it
    constexpr int LO_A = BLK ? 0 : A_PLANE, LO_W = BLK ? 0 : W_PLANE, LO_CH = BLK ? DBK / 8 : 0;
    int a_rd[2][KSUB], w_rd[NJ][KSUB];   // [mi | nj][s]
    int a_rl[2][KSUB], w_rl[NJ][KSUB];   // the lo halves
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wm * 64 + t * 32 + li;
            a_rd[t][s] = ra * ROWB + (((2 * s + lh) ^ G::swz(ra)) * 16);
            a_rl[t][s] = LO_A + ra * ROWB + (((LO_CH + 2 * s + lh) ^ G::swz(ra)) * 16);
        }
#pragma unroll
        for (int t = 0; t < NJ; ++t) {
            const int rw = wn * (32 * NJ) + t * 32 + li;
            w_rd[t][s] = G::NPL * A_PLANE + rw * ROWB + (((2 * s + lh) ^ G::swz(rw)) * 16);
            w_rl[t][s] = G::NPL * A_PLANE + LO_W + rw * ROWB + (((LO_CH + 2 * s + lh) ^ G::swz(rw)) * 16);
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
    LAB_STAMP(1);
    if constexpr (PP) {
        // Ping-pong schedule (NSTG == 3).  Waves w and w+4 share a SIMD; group g = wave >> 2 runs one barrier
        // phase behind group 0, so in every phase one wave of each SIMD issues MFMAs at priority while its
        // partner refills its fragment registers from LDS:
        //     phase 2t   : g0 LOAD(t)     | g1 MFMA(t-1)
        //     phase 2t+1 : g0 MFMA(t)     | g1 LOAD(t)
        // The buffer of stage t-1 is free once both groups' reads of it are retired (lgkmcnt(0)) ahead of the
        // barrier that opens phase 2t, and stage t+2 must have landed by the barrier that opens phase 2t+4.
        // Every wave issues the first half of its DMA share of stage t+2 in phase 2t and the second half in
        // phase 2t+1 (g0: LOAD(t) / MFMA(t); g1: MFMA(t-1) / LOAD(t)), so each phase carries half a stage of
        // DMA issue, split between a loading and a computing wave of every SIMD; inside an MFMA phase the
        // DMA instructions sit between MFMAs, whose execution hides their issue.
        // Every wave waits (counted vmcnt: the six newer DMAs stay in flight) for its share of stage t+1
        // before the barrier that opens phase 2t+2; the first reader starts after that barrier.
        // Both groups execute 2*nkt + 1 barriers.
        static_assert(!PP || (NSTG == 3 && DMA_PER_WAVE % 2 == 0), "ping-pong schedule is written for a 3-stage ring");
        constexpr int H = DMA_PER_WAVE / 2;
        constexpr int NT = KSUB * 2 * NJ;           // MFMA triples per K step
        const int grp = wave >> 2;
        if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp) {
            if (2 < nkt) issue_part(2, 0, H);
            __builtin_amdgcn_s_barrier();
        }
        LAB_PHASE_DECL;
        for (int kt = 0; kt < nkt; ++kt) {
            // ---- LOAD(kt)
            if (kt + 2 < nkt) {
                if (grp) issue_part(kt + 2, H, DMA_PER_WAVE);
                else issue_part(kt + 2, 0, H);
            }
            const char* st = lds + (kt % NSTG) * STAGE;
            bf16x8 ah[KSUB][2], al[KSUB][2], wh[KSUB][NJ], wl[KSUB][NJ];
#pragma unroll
            for (int s = 0; s < KSUB; ++s) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[s][t] = *reinterpret_cast<const bf16x8*>(st + a_rd[t][s]);
                    al[s][t] = *reinterpret_cast<const bf16x8*>(st + a_rl[t][s]);
                }
#pragma unroll
                for (int t = 0; t < NJ; ++t) {
                    wh[s][t] = *reinterpret_cast<const bf16x8*>(st + w_rd[t][s]);
                    wl[s][t] = *reinterpret_cast<const bf16x8*>(st + w_rl[t][s]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            LAB_PHASE(0);
            if (grp) {
                if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LAB_PHASE(3);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            LAB_PHASE(1);
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA(kt), with this phase's half stage of DMA between the MFMAs
            const int dkt = grp ? kt + 3 : kt + 2;
            const bool dma = dkt < nkt;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int s = u / (2 * NJ), mi = (u / NJ) % 2, nj = u % NJ;
                acc[mi][nj] = vrd::mfma32(al[s][mi], wh[s][nj], acc[mi][nj]);
                acc[mi][nj] = vrd::mfma32(ah[s][mi], wl[s][nj], acc[mi][nj]);
                acc[mi][nj] = vrd::mfma32(ah[s][mi], wh[s][nj], acc[mi][nj]);
                if (u % 2 == 0 && u / 2 < H) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma) {
                        if (grp) issue_part(dkt, u / 2, u / 2 + 1);
                        else issue_part(dkt, H + u / 2, H + u / 2 + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            LAB_PHASE(2);
            if (!grp) {
                if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LAB_PHASE(3);
            }
            __builtin_amdgcn_sched_barrier(0);
            // group 1 skips its last barrier (group 0 ran one fewer up front): group 0's epilogue then overlaps
            // group 1's last MFMA phase.  The staging slabs are wave-private and every LDS read and DMA of
            // the ring was retired before the barrier group 0 passed last.
            if (!(grp && kt + 1 == nkt)) __builtin_amdgcn_s_barrier();
            LAB_PHASE(4);
            __builtin_amdgcn_sched_barrier(0);
        }
        LAB_PHASE_FLUSH(grp);
    } else {
    LAB_PHASE_DECL;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed once at most the DMAs of the NSTG-2 newer stages are still outstanding
        if (NSTG > 2 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LAB_PHASE(0);
        __builtin_amdgcn_s_barrier();
        LAB_PHASE(1);
        if (kt + NSTG - 1 < nkt) issue(kt + NSTG - 1);
        LAB_PHASE(2);
        const char* st = lds + (kt % NSTG) * STAGE;
#pragma unroll
        for (int s = 0; s < KSUB; ++s) {
            bf16x8 ah[2], al[2], wh[NJ], wl[NJ];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const bf16x8*>(st + a_rd[t][s]);
                al[t] = *reinterpret_cast<const bf16x8*>(st + a_rl[t][s]);
            }
#pragma unroll
            for (int t = 0; t < NJ; ++t) {
                wh[t] = *reinterpret_cast<const bf16x8*>(st + w_rd[t][s]);
                wl[t] = *reinterpret_cast<const bf16x8*>(st + w_rl[t][s]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) {
                    acc[mi][nj] = vrd::mfma32(al[mi], wh[nj], acc[mi][nj]);
                    acc[mi][nj] = vrd::mfma32(ah[mi], wl[nj], acc[mi][nj]);
                    acc[mi][nj] = vrd::mfma32(ah[mi], wh[nj], acc[mi][nj]);
                }
        }
#ifdef VRD_LAB_STAMP
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[1][1][15]));
#endif
        LAB_PHASE(3);
    }
    LAB_PHASE_FLUSH(wave >> 2);
    // every wave must be done with the ring before it is reused as epilogue staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    }
    LAB_STAMP(2);
#pragma unroll
    for (int hn = 0; hn < NJ / 2; ++hn) {      // the epilogue works on 64 x 64 halves of the wave's sub-tile
        f32x16 part[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) part[i][j] = acc[i][2 * hn + j];
        vrd::gemm_epilogue<true, 64>(p, part, smem, m0 + wm * 64, n0 + wn * (32 * NJ) + hn * 64, wave, lane, rflag);
    }
    LAB_STAMP(3);
    LAB_REAL(5);
}


#ifdef VRD_LAB_STAMP
// (lab builds only: the wave-specialised schedule of the study in LABNOTES.md, measured and not kept)
#include "../../scripts/lab/vrd_gemm_x3_ws.inc"
#endif
}  // namespace

namespace vrd {

template <int TAPS, int DBK, int DBM, int NSTG, bool PP = false, bool BLK = false, bool F16 = false>
static int launch_dma_one(const vrd_gemm_args& a, hipStream_t s) {
    auto kern = gemm_x3_dma_kernel<TAPS, DBK, DBM, NSTG, PP, BLK, F16>;
    constexpr size_t lds = Geo<DBK, DBM, NSTG, BLK>::LDS;
    static_assert(lds >= 8 * 16384 && lds <= 160 * 1024, "ring must hold the epilogue slabs and fit the CU");
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "vrd_gemm(bf16x3 dma)")) return rc;
    const int tiles_m = (int)((a.M + DBM - 1) / DBM), tiles_n = (a.N + DBN - 1) / DBN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, s, a, tiles_m, tiles_n, a.c_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    return 0;
}

#ifdef VRD_LAB_STAMP
template <int TAPS, bool BLK, int NPROD, int NCONS = 8>
static int launch_ws_one(const vrd_gemm_args& a, hipStream_t s) {
    auto kern = gemm_x3_ws_kernel<TAPS, BLK, NPROD, NCONS>;
    constexpr size_t lds = Geo<32, 128, 3, BLK>::LDS;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "vrd_gemm(bf16x3 dma)")) return rc;
    const int tiles_m = (int)((a.M + 127) / 128), tiles_n = (a.N + DBN - 1) / DBN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(64 * (NCONS + NPROD)), lds, s, a, tiles_m, tiles_n);
    return 0;
}

#endif

// eligibility: pair-row A whose slab width and Cin are multiples of 32, 16-byte aligned output rows
bool gemm_x3_dma_ok(const vrd_gemm_args& a, bool staged) {
    return staged && a.a_pair_width > 0 && a.N >= 192;
}

int launch_gemm_x3_dma_variant(const vrd_gemm_args& a, hipStream_t s, int var);

// The library builds schedule 0 only; the study variants are instantiated by the lab harness
// (scripts/lab/gemm_lab.hip defines VRD_LAB_STAMP), which is where they were measured.
int launch_gemm_x3_dma(const vrd_gemm_args& a, hipStream_t s) { return launch_gemm_x3_dma_variant(a, s, 0); }

int launch_gemm_x3_dma_variant(const vrd_gemm_args& a, hipStream_t s, int var) {
    // schedules of the 128 x 256 x 32 tile (all measured at 2,700-3,000 cycles per K step: DMA-issue-bound):
    //   0: every MFMA wave issues its share of a stage's DMA (8 waves)
    //   3: the same with the ping-pong schedule (two wave groups one barrier phase apart)
    //   6 / 7 / 8: wave-specialised, 1 / 2 / 4 producer waves beside 8 MFMA waves of 64 x 64
    //   9 / 10: wave-specialised, 4 / 2 producer waves beside 4 MFMA waves of 64 x 128 (spills)
#ifdef VRD_LAB_STAMP
    if (var == 6) return a.taps == 1 ? launch_ws_one<1, true, 1>(a, s) : launch_ws_one<3, true, 1>(a, s);
    if (var == 7) return a.taps == 1 ? launch_ws_one<1, true, 2>(a, s) : launch_ws_one<3, true, 2>(a, s);
    if (var == 8) return a.taps == 1 ? launch_ws_one<1, true, 4>(a, s) : launch_ws_one<3, true, 4>(a, s);
    if (var == 9) return a.taps == 1 ? launch_ws_one<1, true, 4, 4>(a, s) : launch_ws_one<3, true, 4, 4>(a, s);
    if (var == 10) return a.taps == 1 ? launch_ws_one<1, true, 2, 4>(a, s) : launch_ws_one<3, true, 2, 4>(a, s);
    if (var == 3) return a.taps == 1 ? launch_dma_one<1, 32, 128, 3, true, true>(a, s) : launch_dma_one<3, 32, 128, 3, true, true>(a, s);
#endif
    (void)var;
    if (a.split_fmt == VRD_PAIR_F16)
        return a.taps == 1 ? launch_dma_one<1, 32, 128, 3, false, true, true>(a, s) : launch_dma_one<3, 32, 128, 3, false, true, true>(a, s);
    return a.taps == 1 ? launch_dma_one<1, 32, 128, 3, false, true>(a, s) : launch_dma_one<3, 32, 128, 3, false, true>(a, s);
}

}  // namespace vrd
