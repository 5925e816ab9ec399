// Split-precision conv GEMM, 256 x 256 tile (see vrd_gemm_x3.hip for the arithmetic and vrd_gemm_x3_dma.hip for
// the LDS-DMA staging it shares).
//
// Why a second DMA kernel: measured on the 128 x 256 kernel (scripts/lab/gemm_lab.hip), a 1-KiB LDS-DMA
// instruction costs the CU ~45 cycles while MFMAs are running -- whoever issues it and at whatever priority --
// so a 48-KiB K step costs ~2,100 cycles of DMA issue against 1,536 cycles of MFMA: the step is DMA-issue-bound.
// A 256 x 256 tile moves 64 KiB per K step for twice the MFMAs (3,072 cycles), which puts the MFMAs back in
// front.  The activations stream from HBM (latency of microseconds), the weights come from L2, so the ring is
// split: three activation stages (two in flight, ~6k cycles ahead) and two weight stages:
//     A ring 3 x 32 KiB | W ring 2 x 32 KiB = 160 KiB, all of the CU's LDS (one workgroup per CU).
// Per K step t every wave:  wait (counted vmcnt) until its pieces of W(t) and A(t) landed -> barrier ->
// issue W(t+1) then A(t+2) (4 + 4 DMAs; W first so that the counted wait can leave A(t+2) in flight) ->
// fragment reads and MFMAs of step t.  The barrier also tells everyone that stage t-1 is no longer being read,
// which frees A buffer (t+2) % 3 and W buffer (t+1) % 2.
// 8 waves as 2 (M) x 4 (N), each 128 x 64 = 4 x 2 accumulators of 32 x 32.
// Operands are pair rows in 32-channel blocks [32 hi | 32 lo] (vrd_common.h): a tile row of one K step is one
// 128-byte line; its 16-byte chunks are XOR-swizzled with (row >> 1) & 7 on the DMA source and on the reads.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"
#include <cstdlib>
#include <type_traits>

namespace {

using vrd::f32x16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int TM = 256, TN = 256;
constexpr int ROWB = 128;                      // bytes of a tile row per K step (32 hi + 32 lo bf16)
constexpr int A_STAGE = TM * ROWB, W_STAGE = TN * ROWB;
constexpr int NA_STG = 3, NW_STG = 2;
constexpr int W_RING = NA_STG * A_STAGE;       // byte offset of the W ring
constexpr size_t BIG_LDS = (size_t)NA_STG * A_STAGE + (size_t)NW_STG * W_STAGE;      // 163,840
constexpr int PER = 4;                         // DMA instructions per wave, operand and K step (8 rows x 128 B each)
#ifndef VRD_BIG_BUFDMA
#define VRD_BIG_BUFDMA 0                       // 1: LDS-DMA by buffer_load ... lds (scalar base + offsets), 0: global_load_lds (per-lane pointers)
#endif
#if VRD_BIG_BUFDMA && defined(__HIP_DEVICE_COMPILE__)      // (the descriptor type exists in the device pass only)
#define VRD_BUFDMA_DEV 1
#else
#define VRD_BUFDMA_DEV 0
#endif

__device__ __attribute__((aligned(128))) uint4 g_big_zero[8];      // 128 zero bytes: source of padded taps and outside pieces
// sum over the tiles that skipped their contraction (padding map) of K * tile columns: 2 * 256 * this = FLOPs that
// were launched but not executed (vrd_prof_read_skipped; the profile keeps executed and launched work apart)
__device__ unsigned long long g_big_skipped_kn;

#ifndef VRD_LAB_STAMP      // the lab harness (scripts/lab/gemm_lab.hip) defines these before including this file
#define LAB_STAMP(slot)
#define LAB_REAL(slot)
#define LAB_PHASE_DECL
#define LAB_PHASE(i)
#define LAB_PHASE_FLUSH(grp)
#else                      // per TILE (virtual block id vb) instead of per workgroup
#undef LAB_STAMP
#undef LAB_REAL
#undef LAB_PHASE_FLUSH
#define LAB_STAMP(slot)                                                                        \
    do {                                                                                       \
        if (threadIdx.x == 0 && vb < 65536) g_lab[vb * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define LAB_REAL(slot)                                                                         \
    do {                                                                                       \
        if (threadIdx.x == 0 && vb < 65536) g_lab[vb * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define LAB_PHASE_FLUSH(grp)                                                                   \
    do {                                                                                       \
        if ((threadIdx.x & 255) == 0 && vb < 4096)                                             \
            for (int i_ = 0; i_ < 5; ++i_) g_lab_phase[vb * 16 + (grp) * 8 + i_] = lab_acc[i_]; \
    } while (0)
#endif

__device__ constexpr int swz(int row) { return (row >> 1) & 7; }

// operands of problems 1 .. 3 of a batched launch (blockIdx.y; problem 0 is the argument struct itself): GEMMs that
// differ only in A, the weights, the bias and C -- the q / k / v projections of one attention block -- run as one
// grid, so a block's three projections have one ragged last round of tiles instead of three
struct BigBatch {
    const float* A[3];
    const uint16_t* W_split[3];
    const float* bias[3];
    float* C[3];
    const float* w_scale[3];
};

// F16: operands in the scaled-f16 format (VRD_PAIR_F16) on v_mfma_f32_32x32x16_f16 -- the same bytes, instruction count and
// cycles; the epilogue multiplies the accumulators by *w_scale
template <int TAPS, bool M16, bool PERSIST, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_x3_big_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, BigBatch bb, int stagger) {
    typedef typename vrd::SplitFmt<F16>::x8 e16x8;      // fragment of eight 16-bit elements (bf16 or f16)
    // Phase stagger.  Every tile of a launch takes the same time, so without it all CUs reach their epilogues together and
    // 256 x 256 KiB of stores meet an HBM that was idle a moment before.  The first workgroup of every CU (the first 256 of
    // the grid: one per CU) starts `slot * stagger` x 1,024 cycles late, slot = its place among the 32 CUs of its XCD; the
    // offsets then persist through the launch (a CU's next workgroup starts when its last one ends).
    if (stagger && blockIdx.y == 0 && blockIdx.x < 256) {
        const int n = ((blockIdx.x >> 3) & 31) * stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
    if (blockIdx.y) {                                    // uniform selects, no indexed access to the arguments
        const int z = blockIdx.y;
        p.A = z == 1 ? bb.A[0] : z == 2 ? bb.A[1] : bb.A[2];
        p.w_scale = z == 1 ? bb.w_scale[0] : z == 2 ? bb.w_scale[1] : bb.w_scale[2];
        p.W_split = z == 1 ? bb.W_split[0] : z == 2 ? bb.W_split[1] : bb.W_split[2];
        p.bias = z == 1 ? bb.bias[0] : z == 2 ? bb.bias[1] : bb.bias[2];
        p.C = z == 1 ? bb.C[0] : z == 2 ? bb.C[1] : bb.C[2];
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int K = p.Cin * TAPS;
    const int nkt = K / 32;
    const int nwg = tiles_m * tiles_n;
    // ---- tiles.  Virtual block id -> tile through the XCD-aware renumbering; the tile's rows are eight 32-row blocks,
    // slots tm*8 .. tm*8+7 of the block list (identity without one).  With a padding map (vrd_row_blocks) the list is
    // cut into segments -- one per XCD's contiguous share of the tiles when there are eight -- and inside a segment
    // the blocks holding valid frames come first, so a tile is either a contraction tile or, behind those, a tile of
    // fully padded blocks that only runs the epilogue on a zero accumulator (the reference's value wherever row_mask
    // zeroes the row).
    const int nblk = (int)(p.M >> 5);
    const int32_t* const rb = p.row_blocks;
    auto blk_of = [&](int slot) { return slot < nblk ? (rb ? rb[slot] : slot) : -1; };
    struct Tile {
        int tm, n0;
        bool contract;
    };
    auto tile_of = [&](int vb_) {
        const int xcd = vb_ & 7, q = nwg >> 3, rem = nwg & 7;
        const int lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (vb_ >> 3);
        Tile t;
        t.tm = lid / tiles_n;
        t.n0 = (lid - t.tm * tiles_n) * TN;
        t.contract = true;
        if (rb) {
            const int seg_len = p.row_block_seg_len;                          // a multiple of 8 (host-checked)
            const int seg = (t.tm * 8) / seg_len;
            t.contract = t.tm * 8 < nblk && t.tm * 8 - seg * seg_len < p.row_blocks_active[seg];
        }
        return t;
    };
    // ---- DMA sources of a tile: this wave moves row blocks wave*4 .. wave*4+3 (8 rows each) of both operands.  Only
    // piece 0's per-lane pointers are kept: piece i is a wave-uniform stride further, its source-side swizzle differs
    // from piece 0's by bit 6 for odd i, and (M, N multiples of 64) a piece lies entirely inside or outside the matrix,
    // which is a scalar test; pieces outside read the zero block.  (The bit-6 flip is applied to the pointer, which is
    // why the host sends only 128-byte aligned A and W_split here: address bits 4..6 are then the chunk index.)
    const int rin = lane >> 3, pch = lane & 7;
    const int row0 = wave * PER * 8 + rin;                              // row inside the tile, for both operands
    const int chunk0 = (pch ^ swz(row0)) * 16;
    const char* const zero_src = reinterpret_cast<const char*>(g_big_zero);
    const int64_t a_pstride = p.lda * 32, w_pstride = (int64_t)K * 32;  // bytes between pieces (8 rows)
    struct Src {
        const char *a0, *w0;
        int tseq0, a_in, w_in, w_last;
#if VRD_BUFDMA_DEV
        __amdgpu_buffer_rsrc_t ra, rw;          // wave-uniform descriptors: the wave's A block (one row back for k = 3), its W rows
        unsigned va[2], vw[2];                  // per-lane byte offsets inside them, even / odd pieces (source-side swizzle)
#endif
    };
    auto src_of = [&](const Tile& t) {
        const int my_blk = blk_of(t.tm * 8 + wave);          // the block whose A rows this wave stages
        const int64_t a_row = (int64_t)(my_blk < 0 ? 0 : my_blk) * 32 + rin;
        Src r;
        r.a0 = reinterpret_cast<const char*>(p.A + a_row * p.lda) + chunk0;
        r.tseq0 = (TAPS == 3) ? (int)(a_row % p.T) : 0;
        r.a_in = my_blk < 0 ? 0 : PER;                                   // pieces inside (M % 32 == 0)
        r.w_in = (p.N - t.n0 - wave * PER * 8 + 7) / 8;
        // weight pieces beyond N re-read the last piece inside (a wave entirely beyond N: the tile's first rows): columns
        // >= N are never stored, and a uniform minimum costs the loop less than a per-lane select of the zero block
        r.w_last = (r.w_in > 0 ? (r.w_in < PER ? r.w_in : PER) : 1) - 1;
        r.w0 = reinterpret_cast<const char*>(p.W_split) + (int64_t)(t.n0 + (r.w_in > 0 ? row0 : rin)) * K * 4 + chunk0;
#if VRD_BUFDMA_DEV
        {
            // buffer form of the same requests: base (scalar) + per-lane offset (constant for the tile) + scalar offset (piece,
            // K step, tap).  Offsets >= 2^31 are out of range: such a lane's 16 bytes arrive in LDS as zeros, which is how
            // padded taps and pieces outside the matrix are produced here (no zero block, no per-lane pointer select).
            const int64_t a_row_u = (int64_t)(my_blk < 0 ? 0 : my_blk) * 32 - (TAPS == 3 ? 1 : 0);
            r.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + a_row_u * p.lda), 0, 0x80000000u, 0x00020000);
            r.rw = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(p.W_split) + (int64_t)(t.n0 + (r.w_in > 0 ? wave * PER * 8 : 0)) * K * 4), 0,
                0x80000000u, 0x00020000);
            const unsigned a_lane = (unsigned)rin * (unsigned)(p.lda * 4), w_lane = (unsigned)rin * (unsigned)(K * 4);
            r.va[0] = a_lane + chunk0, r.va[1] = a_lane + (chunk0 ^ 64);
            r.vw[0] = w_lane + chunk0, r.vw[1] = w_lane + (chunk0 ^ 64);
        }
#endif
        return r;
    };
    // piece i of W(kt) / A(kt): one DMA instruction each
    auto issue_w1 = [&](const Src& c, int kt, int i) {
        char* const dst = lds + W_RING + (kt % NW_STG) * W_STAGE + wave * PER * 1024;
#if VRD_BUFDMA_DEV
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rw, (lds_ptr_t)(dst + i * 1024), 16, c.vw[i & 1],
                                                 (i < c.w_last ? i : c.w_last) * (int)w_pstride + kt * 128, 0, 0);
        return;
#endif
        const char* src = c.w0 + (i < c.w_last ? i : c.w_last) * w_pstride + (int64_t)kt * 128;
        if (i & 1) src = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(src) ^ 64);
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
    };
    auto issue_a1 = [&](const Src& c, int kt, int i) {
        char* const dst = lds + (kt % NA_STG) * A_STAGE + wave * PER * 1024;
        const int k0 = kt * 32;
        int tap = 0, ci0 = k0;
        if (TAPS == 3) {
            tap = (k0 >= p.Cin) + (k0 >= 2 * p.Cin);
            ci0 = k0 - tap * p.Cin;
        }
#if VRD_BUFDMA_DEV
        {
            unsigned vo = c.va[i & 1];
            if (TAPS == 3) {
                int tt = c.tseq0 + 8 * i;
                if (tt >= p.T) tt -= p.T;
                tt += tap - 1;
                if (tt < 0 || tt >= p.T || i >= c.a_in) vo = 0x80000000u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_ptr_t)(dst + i * 1024), 16, vo,
                                                     i * (int)a_pstride + tap * (int)(p.lda * 4) + ci0 * 4, 0, 0);
            return;
        }
#endif
        const int64_t off = (int64_t)(tap - (TAPS == 3 ? 1 : 0)) * p.lda * 4 + (int64_t)ci0 * 4;
        // (k = 1: a wave whose block lies outside the matrix reads block 0 -- its accumulator rows are never stored)
        const char* src = TAPS == 1 || i < c.a_in ? c.a0 + i * a_pstride + off : zero_src + chunk0;
        if (TAPS == 3) {
            int tt = c.tseq0 + 8 * i;                   // position of this piece's row in its sequence (T >= 32)
            if (tt >= p.T) tt -= p.T;
            tt += tap - 1;
            if (tt < 0 || tt >= p.T) src = zero_src + chunk0;
        }
        if (i & 1) src = reinterpret_cast<const char*>(reinterpret_cast<uintptr_t>(src) ^ 64);
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
    };
    auto issue_w = [&](const Src& c, int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_w1(c, kt, i);
    };
    auto issue_a = [&](const Src& c, int kt) {
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_a1(c, kt, i);
    };

    // PERSIST (VRD_BIG_PERSIST=1, off by default): one workgroup per CU walks tiles vb, vb + gridDim.x, ...; behind a
    // tile's main loop, before its epilogue, stage 0 of the NEXT tile's operands is requested, so that tile does not
    // start with an HBM round trip (the epilogue then stages through 32-row slabs inside activation stages 1-2, which
    // stage 0 of either ring leaves free).  Measured: correct, but 3 % SLOWER on the whole step (98.8 vs 96.1 ms of
    // this kernel): with the tile loop around the body everything derived from the arguments stays live across it --
    // 100 SGPRs spilled to VGPR lanes and 39 VGPRs to scratch, inside the K loop -- and re-reading the arguments per
    // tile through a laundered kernarg pointer moved them into VGPRs instead (99 spills).  The ~4 k cycles of tile
    // setup it would hide are 5 % of a tile; not worth a second, argument-light kernel this round.
    bool staged0 = false;
    int vb = blockIdx.x;
    do {
    LAB_STAMP(0);
    LAB_REAL(4);
    const Tile tile = tile_of(vb);
    const int tm = tile.tm, n0 = tile.n0;
    const bool contract = tile.contract;
    const Src cur = src_of(tile);
    if (!contract && tid == 0 && tm * 8 < nblk)
        atomicAdd(&g_big_skipped_kn, (unsigned long long)K * (unsigned)(p.N - n0 < TN ? p.N - n0 : TN));

    // ---- fragment read offsets (bytes inside a stage).  A tile row is 128 bytes: hi chunks 0..3, lo chunks 4..7,
    // chunk index XOR-swizzled with (row >> 1) & 7 = (li >> 1) & 7 for every 32-row block.  Hence all fragment
    // addresses of an operand derive from ONE per-lane base: k16 half s flips bit 5 (chunk ^ 2), lo flips bit 6
    // (chunk ^ 4), and the 32-row blocks are constant offsets (kept out of registers: two VGPRs instead of 24).
    // (M16: v_mfma_f32_16x16x32 -- lane l holds row l & 15, k = 8 * (l >> 4) .. +7 of the whole K step, i.e. hi chunk
    //  l >> 4; the swizzle of rows (16-row block) + (l & 15) is again that of l & 15)
    const int l15 = lane & 15, l4 = lane >> 4;
    const int a_base = M16 ? (wm * 128 + l15) * ROWB + ((l4 ^ swz(l15)) * 16) : (wm * 128 + li) * ROWB + ((lh ^ swz(li)) * 16);
    const int w_base = W_RING + (M16 ? (wn * 64 + l15) * ROWB + ((l4 ^ swz(l15)) * 16) : (wn * 64 + li) * ROWB + ((lh ^ swz(li)) * 16));

    // bias / scale of this lane's columns: requested now, used by the epilogue
    const vrd::EpiCols cols = vrd::load_epi_cols(p, n0 + wn * 64, lane);
    f32x16 acc[M16 ? 1 : 4][M16 ? 1 : 2];
    vrd::f32x4_t acc16[M16 ? 8 : 1][M16 ? 4 : 1];        // M16: 8 x 4 tiles of 16 x 16
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : 4); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 1 : 2); ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? 8 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 4 : 1); ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;

    // ---- main loop.  A K step is eight groups g = (s, mi) of six MFMAs (k16 half s, 32-row block mi, both
    // column blocks).  Fragment reads run one group ahead of the MFMAs (A fragments of group g+1, and the W
    // fragments of the next half two groups ahead), so no LDS latency is exposed; for that to hold across K
    // steps the step's barrier sits BEFORE its last group: by then every fragment of the step is in registers
    // (lgkmcnt(0)), so the barrier both publishes stage kt+1 (every wave waited for its own pieces first) and
    // frees the buffers of stage kt, which the DMAs issued after it refill: W(kt+2), then A(kt+3), one per
    // group over the next eight groups (a DMA issue stalls its wave for 100-200 cycles while MFMAs run; the
    // two waves of a SIMD place theirs half a group apart).
    constexpr int NWF = M16 ? 4 : 2;                 // W fragments (column blocks) held at a time
    struct AF { e16x8 hi, lo; };
    struct WF { e16x8 hi[NWF], lo[NWF]; };
    // 32x32x16: (s2, mi) = k16 half, 32-row block.  M16: s2 unused, mi = 16-row block 0..7 (g of the group)
    auto load_a = [&](const char* sa, int s2, int mi) {
        AF f;
        const int off = M16 ? a_base + mi * 16 * ROWB : (a_base ^ (s2 * 32)) + mi * 32 * ROWB;
        f.hi = *reinterpret_cast<const e16x8*>(sa + off);
        f.lo = *reinterpret_cast<const e16x8*>(sa + (off ^ 64));
        return f;
    };
    auto load_w = [&](const char* sw, int s2) {
        WF f;
#pragma unroll
        for (int t = 0; t < NWF; ++t) {
            const int off = M16 ? w_base + t * 16 * ROWB : (w_base ^ (s2 * 32)) + t * 32 * ROWB;
            f.hi[t] = *reinterpret_cast<const e16x8*>(sw + off);
            f.lo[t] = *reinterpret_cast<const e16x8*>(sw + (off ^ 64));
        }
        return f;
    };
    if (contract) {
    if (!(PERSIST && staged0)) {
        issue_a(cur, 0);
        issue_w(cur, 0);
    }
    if (nkt > 1) {
        issue_a(cur, 1);
        issue_w(cur, 1);
    }
    LAB_STAMP(1);
    // stage 0: what was issued after A(0), W(0) may stay in flight (A(2) follows inside step 0, see below)
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    WF w_cur = load_w(lds, 0), w_nxt = w_cur;
    AF a_cur = load_a(lds, 0, 0), a_nxt = a_cur;
    LAB_PHASE_DECL;
    // DMA slot d (0..7) of the batch opened by the barrier inside step kt: W(kt+2) pieces 0..3, A(kt+3) pieces 0..3
    // Which of these requests exist depends only on where the step stands in the K loop, so the loop body exists in five
    // copies with the tests resolved at compile time: first step, steady steps (1 .. nkt-3), the step before the last, the
    // last, and a generic one with run-time tests for K loops of fewer than three steps.  (The per-slot scalar branches of the
    // generic body cost the steady loop 7 %: 4,300 -> 3,730-4,060 cycles per K step.)
    enum { POS_GENERIC, POS_FIRST, POS_STEADY, POS_PEN, POS_LAST };
    auto kstep = [&](int kt, auto pos_c) __attribute__((always_inline)) {
        constexpr int POS = decltype(pos_c)::value;
        // W(kt_open+2) pieces / A(kt_open+3) pieces exist?
        auto has_w = [&](int kt_open) { return POS == POS_GENERIC ? kt_open + 2 < nkt : true; };
        auto dma_slot = [&](int kt_open, int d, bool w_ok, bool a_ok) {
            if (d < PER) {
                if (w_ok) issue_w1(cur, kt_open + 2, d);
            } else {
                if (a_ok) issue_a1(cur, kt_open + 3, d - PER);
            }
        };
        (void)has_w;
        const char* sa = lds + (kt % NA_STG) * A_STAGE;
        const char* sw = lds + (kt % NW_STG) * W_STAGE;
        const char* sa1 = lds + ((kt + 1) % NA_STG) * A_STAGE;
        const char* sw1 = lds + ((kt + 1) % NW_STG) * W_STAGE;
        const bool last = POS == POS_GENERIC ? kt + 1 == nkt : POS == POS_LAST;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int mi = g & 3;
            // ---- reads for what comes next
            if (g < 7) a_nxt = M16 ? load_a(sa, 0, g + 1) : load_a(sa, (g + 1) >> 2, (g + 1) & 3);
            if (!M16 && g == 2) w_nxt = load_w(sw, 1);
            if (g == 7 && !last) {
                // (the barrier was passed at the end of group 6)
                a_nxt = load_a(sa1, 0, 0);
                w_nxt = load_w(sw1, 0);
            }
            // ---- the group's MFMAs (six 32x32x16 or twelve 16x16x32), this wave's DMA of the group in the middle or
            // at the end
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                if (M16) {
#pragma unroll
                    for (int t = 2 * nj; t < 2 * nj + 2; ++t) {
                        acc16[g][t] = vrd::mfma16(a_cur.lo, w_cur.hi[t], acc16[g][t]);
                        acc16[g][t] = vrd::mfma16(a_cur.hi, w_cur.lo[t], acc16[g][t]);
                        acc16[g][t] = vrd::mfma16(a_cur.hi, w_cur.hi[t], acc16[g][t]);
                    }
                } else {
                    acc[mi][nj] = vrd::mfma32(a_cur.lo, w_cur.hi[nj], acc[mi][nj]);
                    acc[mi][nj] = vrd::mfma32(a_cur.hi, w_cur.lo[nj], acc[mi][nj]);
                    acc[mi][nj] = vrd::mfma32(a_cur.hi, w_cur.hi[nj], acc[mi][nj]);
                }
                if ((wave >> 2) == nj) {
                    __builtin_amdgcn_sched_barrier(0);
                    // groups 0..6 carry slots 1..7 of the batch opened in step kt-1, group 7 slot 0 of this step's
                    if (g < 7) {
                        // the batch opened in step kt-1: W(kt+1) pieces 1..3, A(kt+2) pieces 0..3
                        if (POS == POS_GENERIC) {
                            if (kt > 0) dma_slot(kt - 1, g + 1, kt + 1 < nkt, kt + 2 < nkt);
                            else if (g < PER && nkt > 2) issue_a1(cur, 2, g);      // step 0 has no batch of its own yet
                        } else if (POS == POS_FIRST) {
                            if (g < PER) issue_a1(cur, 2, g);
                        } else if (POS == POS_STEADY) {
                            dma_slot(kt - 1, g + 1, true, true);
                        } else if (POS == POS_PEN) {
                            dma_slot(kt - 1, g + 1, true, false);
                        }
                    } else if (!last) {
                        // slot 0 of this step's own batch: W(kt+2) piece 0
                        dma_slot(kt, 0, POS == POS_GENERIC ? kt + 2 < nkt : POS != POS_PEN, false);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if defined(VRD_LAB_STAMP) && defined(VRD_LAB_VALU)
            // lab only (round 3, LABNOTES.md "producing the q / k / v operands inside the projection GEMM"): what the K loop
            // pays for VRD_LAB_VALU extra vector instructions (and VRD_LAB_LDSR extra 16-byte LDS reads) per wave and K step,
            // an eighth of them behind each MFMA group -- the in-loop LayerNorm -> depthwise conv -> LayerNorm -> hi / lo split
            // of a fused attention-input stage would need ~210 + ~36 per K step
            {
                static_assert(VRD_LAB_VALU % 8 == 0 && VRD_LAB_LDSR % 8 == 0, "per group");
                float d0 = acc[0][0][0] * 0.f + 1.f, d1 = 2.f, d2 = 3.f, d3 = 4.f;
#pragma unroll
                for (int i = 0; i < VRD_LAB_VALU / 8 / 4; ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1"
                                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#pragma unroll
                for (int i = 0; i < VRD_LAB_LDSR / 8; ++i) {
                    vrd::f32x4_t t = *reinterpret_cast<const vrd::f32x4_t*>(sa + ((a_base + i * 2048) & (A_STAGE - 16)));
                    asm volatile("" ::"v"(t));
                }
                asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3));
            }
#endif
            a_cur = a_nxt;
            if ((!M16 && g == 3) || g == 7) w_cur = w_nxt;
            if (g == 6 && !last) {
                // every fragment of stage kt is in registers or landed; stage kt+1 must be visible before group 7
                // starts reading it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                LAB_PHASE(3);
                if (POS == POS_GENERIC ? kt + 2 < nkt : POS != POS_PEN) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LAB_PHASE(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                LAB_PHASE(1);
            }
        }
    };
    if (nkt >= 3) {
        kstep(0, std::integral_constant<int, POS_FIRST>{});
        for (int kt = 1; kt + 2 < nkt; ++kt) kstep(kt, std::integral_constant<int, POS_STEADY>{});
        kstep(nkt - 2, std::integral_constant<int, POS_PEN>{});
        kstep(nkt - 1, std::integral_constant<int, POS_LAST>{});
    } else {
        for (int kt = 0; kt < nkt; ++kt) kstep(kt, std::integral_constant<int, POS_GENERIC>{});
    }
    LAB_PHASE_FLUSH(wave >> 2);
    }       // contract
#ifdef VRD_LAB_STAMP
    asm volatile("" ::"v"(acc[0][0][0]), "v"(acc16[0][0][0]));
#endif
    // every wave must be done with the rings before they are reused as epilogue staging
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    LAB_STAMP(2);
    if (PERSIST) {
        // stage 0 of the next tile (activation stage 0, weight stage 0): in flight during this tile's epilogue
        const int vn = vb + (int)gridDim.x;
        staged0 = false;
        if (vn < nwg) {
            const Tile nt = tile_of(vn);
            if (nt.contract) {
                const Src nx = src_of(nt);
                issue_a(nx, 0);
                issue_w(nx, 0);
                staged0 = true;
            }
        }
    }
    // staging slab of this wave: 64 rows at the front of LDS, or (PERSIST) 32 rows inside activation stages 1-2
    float* const stg = PERSIST ? smem + (A_STAGE + wave * (32 * vrd::STG_PITCH * 4)) / 4 : smem + wave * (64 * vrd::STG_PITCH);
    constexpr int SLAB = PERSIST ? 32 : 64;
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {          // the epilogue works on 64 x 64 halves of the wave's 128 x 64
        // (M % 64 == 0 and N % 64 == 0, checked on the host: the sub-tile is inside C or entirely outside)
        const int slot = tm * 8 + wm * 4 + hm * 2;
        const int blk_a = blk_of(slot), blk_b = blk_of(slot + 1);
        const int nw = n0 + wn * 64;
        if (blk_a < 0 || nw >= p.N) continue;
        const int64_t mw = (int64_t)blk_a * 32, mw1 = (int64_t)blk_b * 32;      // rows of passes 0-1 / 2-3
        const bool rowin = p.row_mask || p.scale || p.res || p.res2;
        if (M16) {
            vrd::f32x4_t part[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[i][j] = acc16[M16 ? 4 * hm + i : 0][M16 ? j : 0];
            if (rowin) vrd::gemm_epilogue_lean16<true, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
            else if (p.act == VRD_ACT_GELU) vrd::gemm_epilogue_lean16<false, VRD_ACT_GELU, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
            else vrd::gemm_epilogue_lean16<false, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
        } else {
            f32x16 part[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) part[i][j] = acc[M16 ? 0 : 2 * hm + i][M16 ? 0 : j];
            if (rowin) vrd::gemm_epilogue_lean<true, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
            else if (p.act == VRD_ACT_GELU) vrd::gemm_epilogue_lean<false, VRD_ACT_GELU, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
            else vrd::gemm_epilogue_lean<false, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane, cols);
        }
    }
    LAB_STAMP(3);
    LAB_REAL(5);
    if (PERSIST) {
        // the slabs lie where the next tile's stage-1 / stage-2 DMAs land: everybody is done reading theirs first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    vb += (int)gridDim.x;
    } while (PERSIST && vb < nwg);       // tiles of this workgroup
}

}  // namespace

namespace vrd {

template <int TAPS, bool M16, bool PERSIST, bool F16 = false>
static int launch_big_one(const vrd_gemm_args& a, hipStream_t s, const BigBatch& bb = BigBatch{}, int count = 1) {
    auto kern = gemm_x3_big_kernel<TAPS, M16, PERSIST, F16>;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), BIG_LDS, "vrd_gemm(bf16x3 256x256)")) return rc;
    const int tiles_m = (int)((a.M + TM - 1) / TM), tiles_n = (a.N + TN - 1) / TN;
    const int nwg = tiles_m * tiles_n;
    // (measured, scripts/dev/stagger_sweep.sh, profiles/r04_lab_gemm_stagger.txt: 0 / 1 / 2 / 4 / 8 units -> 98.1-98.2 / 97.4 / 97.1 /
    // 97.8 / 99.4 ms of this kernel per step: about 1 %, so the epilogues were not waiting for each other's stores much)
    static const int stagger = [] { const char* e = getenv("VRD_BIG_STAGGER"); return e ? atoi(e) : 2; }();
    hipLaunchKernelGGL(kern, dim3(PERSIST ? (nwg < 256 ? nwg : 256) : nwg, count), dim3(512), BIG_LDS, s, a, tiles_m, tiles_n, bb,
                       (PERSIST || nwg < 512) ? 0 : stagger);
    return 0;
}

// `count` (2 .. 4) problems that differ only in A, W_split, bias and C, as one launch of the default kernel
int launch_gemm_x3_big_batch(const vrd_gemm_args* a, int count, hipStream_t s) {
    BigBatch bb{};
    for (int i = 1; i < count; ++i) {
        bb.A[i - 1] = a[i].A;
        bb.W_split[i - 1] = a[i].W_split;
        bb.bias[i - 1] = a[i].bias;
        bb.C[i - 1] = a[i].C;
        bb.w_scale[i - 1] = a[i].w_scale;
    }
    if (a[0].split_fmt == VRD_PAIR_F16)
        return a[0].taps == 1 ? launch_big_one<1, false, false, true>(a[0], s, bb, count) : launch_big_one<3, false, false, true>(a[0], s, bb, count);
    return a[0].taps == 1 ? launch_big_one<1, false, false>(a[0], s, bb, count) : launch_big_one<3, false, false>(a[0], s, bb, count);
}

// FLOPs of contractions skipped through padding maps since the last call (reads and clears the device counter)
double take_big_skipped_flops() {
    unsigned long long v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_big_skipped_kn), sizeof(v)) != hipSuccess) return 0.0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_big_skipped_kn), &zero, sizeof(zero));
    return 2.0 * TM * (double)v;
}

// same eligibility as the 128 x 256 DMA kernel (pair-row A, staged epilogue); the caller picks by tile count
int launch_gemm_x3_big(const vrd_gemm_args& a, hipStream_t s) {
    // MFMA shape: 32x32x16 (default) or 16x16x32 (VRD_BIG_M16=1).  Same fragments, LDS traffic and MFMA cycles per K
    // step; interleaved A/B runs in one process put 16x16x32 0.5-1 % ahead on the whole step, but it sums the K
    // dimension in a different order than the 32x32x16 kernels that serve small batches, and the path keeps its
    // results independent of the batch composition to the last bit (tests/test_gpu_model.py), so it stays opt-in.
    static const int m16 = [] { const char* e = getenv("VRD_BIG_M16"); return e ? atoi(e) : 0; }();
    // VRD_BIG_PERSIST=1: one workgroup per CU walking its tiles, the next tile's first stage requested under the epilogue
    static const int persist = [] { const char* e = getenv("VRD_BIG_PERSIST"); return e ? atoi(e) : 0; }();
    if (a.split_fmt == VRD_PAIR_F16) {      // (the persistent variant exists for the bf16 format only)
        if (m16) return a.taps == 1 ? launch_big_one<1, true, false, true>(a, s) : launch_big_one<3, true, false, true>(a, s);
        return a.taps == 1 ? launch_big_one<1, false, false, true>(a, s) : launch_big_one<3, false, false, true>(a, s);
    }
    if (m16) return a.taps == 1 ? launch_big_one<1, true, false>(a, s) : launch_big_one<3, true, false>(a, s);
    if (persist) return a.taps == 1 ? launch_big_one<1, false, true>(a, s) : launch_big_one<3, false, true>(a, s);
    return a.taps == 1 ? launch_big_one<1, false, false>(a, s) : launch_big_one<3, false, false>(a, s);
}

}  // namespace vrd
