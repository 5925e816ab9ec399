// Split-precision conv GEMM, 256 x 256 tile (see vrd_gemm_x3.hip for the arithmetic and vrd_gemm_x3_dma.hip for
// the LDS-DMA staging it shares).
//
// Why a second DMA kernel: measured on the 128 x 256 kernel (round 2's lab harness; LABNOTES.md), a 1-KiB LDS-DMA
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
// 8 waves as 2 (M) x 4 (N), each 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32 (round 6: on random operands the chip
// holds a 10-17 % higher clock on this shape than on 32x32x16 at the same cycles per FLOP -- MI355X_MICROARCH.md, DVFS item 7;
// profiles/r06_lab_gemm_m16.txt).
// Operands are pair rows in 32-channel blocks [32 hi | 32 lo] (vrd_common.h): a tile row of one K step is one
// 128-byte line; its 16-byte chunks are XOR-swizzled with (row >> 1) & 7 on the DMA source and on the reads.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int TM = 256, TN = 256;
constexpr int ROWB = 128;                      // bytes of a tile row per K step (32 hi + 32 lo bf16)
constexpr int A_STAGE = TM * ROWB, W_STAGE = TN * ROWB;
constexpr int NA_STG = 3, NW_STG = 2;
__attribute__((unused)) constexpr int W_RING = NA_STG * A_STAGE;       // byte offset of the W ring
constexpr size_t BIG_LDS = (size_t)NA_STG * A_STAGE + (size_t)NW_STG * W_STAGE;      // 163,840
__attribute__((unused)) constexpr int PER = 4;                         // DMA instructions per wave, operand and K step (8 rows x 128 B each)
// LDS-DMA requests are buffer loads (buffer_load_dwordx4 ... lds): a wave-uniform descriptor per operand and tile, ONE per-lane
// byte offset per operand that never changes (row inside the piece, swizzled chunk), and a scalar offset for piece, K step and
// tap -- no vector arithmetic per request (the global_load_lds form needed a 64-bit per-lane pointer each: -4.5 % cycles per
// K step, profiles/r05_lab_gemm_tile_stamps.txt), and a lane whose offset is out of range (>= 2^31 here) receives zeros, which
// is how padded taps and pieces outside the matrix are produced: no zero block, no per-lane pointer select.
// sum over the tiles that skipped their contraction (padding map) of K * tile columns: 2 * 256 * this = FLOPs that
// were launched but not executed (vrd_prof_read_skipped; the profile keeps executed and launched work apart)
__device__ unsigned long long g_big_skipped_kn;

#ifndef VRD_LAB_STAMP      // the lab harness (scripts/lab/r06/gemm6_lab.hip) defines these before including this file
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

// The kernel's one argument.  It is read through the kernarg segment pointer (scalar loads from the constant address space),
// and the persistent form reads it AGAIN for every tile, and once more between a tile's main loop and its epilogue: with
// the ~40 argument words loaded once in front of a tile loop, everything derived from them stayed live across the loop and
// hipcc spilled 100+ scalar registers into the K loop (round 3: 3 % slower than one workgroup per tile).  Re-reading costs
// a few scalar loads per tile and leaves the K loop the register budget of the one-tile kernel.
struct BigKArgs {
    vrd_gemm_args p;
    int tiles_m, tiles_n;
    BigBatch bb;
    int stagger;
    int count;                // problems of the launch (1 + the BigBatch entries in use)
    unsigned* rflag;          // the device's f16 operand-range flag (vrd_common.h, RangeTrack) when C is written as f16 pair rows
};
typedef const __attribute__((address_space(4))) unsigned* kargs_ptr_t;
template <typename T>
__device__ __forceinline__ T load_karg(kargs_ptr_t base, int byte_off) {
    constexpr int N = (sizeof(T) + 3) / 4;
    unsigned tmp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i] = base[byte_off / 4 + i];
    T out;
    __builtin_memcpy(&out, tmp, sizeof(T));
    return out;
}

// F16: operands in the scaled-f16 format (VRD_PAIR_F16) on v_mfma_f32_16x16x32_f16 -- the same bytes, instruction count and
// cycles; the epilogue multiplies the accumulators by *w_scale
// PERSIST: one workgroup per CU walks tiles vb, vb + gridDim.x, ...; behind a tile's main loop, before its epilogue, the first
// stages of the NEXT tile are requested (A(0), A(1): the operand that comes from HBM, microseconds away; W(0)), so that tile
// starts on landed data: the one-tile form spends ~9 k of a K = 512 tile's ~66 k cycles on its set-up and on waiting for its
// first stage (profiles/r05_lab_gemm_tile_stamps.txt).  The epilogue then stages through 32-row slabs in the two ring
// buffers the prefetch leaves free (activation stage 2, weight stage 1).
// WH = wave >> 2 as a compile-time constant: the two halves of the workgroup run their own copy of the whole body (the place
// of a wave's LDS-DMA request inside an MFMA group is then no branch: 16 per K step before), and the copies never join, so the
// register assignment of one does not constrain the other (joined behind the K loop, one copy spilled accumulators)
template <int TAPS, bool PERSIST, bool F16, int WH>
__device__ __forceinline__ void gemm_x3_big_body() {
#if defined(__HIP_DEVICE_COMPILE__)       // (the buffer descriptor type exists in the device pass only)
    typedef typename vrd::SplitFmt<F16>::x8 e16x8;      // fragment of eight 16-bit elements (bf16 or f16)
    kargs_ptr_t kp = (kargs_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    vrd_gemm_args p;
    int tiles_m, tiles_n, K, nkt, nwg, nblk;
    const int32_t* rb;
    int64_t a_pstride, w_pstride;                        // bytes between DMA pieces (8 rows)
    // problem of a batched launch (BigBatch): blockIdx.y, or (PERSIST: one grid row walks the tiles of all problems) the
    // problem of the tile at hand
    int zsel = PERSIST ? 0 : (int)blockIdx.y;
    auto reload = [&]() __attribute__((always_inline)) {
        if (PERSIST) asm volatile("" : "+s"(kp));        // (a new value as far as the compiler can tell: nothing loaded before survives)
        p = load_karg<vrd_gemm_args>(kp, offsetof(BigKArgs, p));
        tiles_m = load_karg<int>(kp, offsetof(BigKArgs, tiles_m));
        tiles_n = load_karg<int>(kp, offsetof(BigKArgs, tiles_n));
        if (zsel) {                                      // uniform selects, no indexed access to the arguments
            const BigBatch bb = load_karg<BigBatch>(kp, offsetof(BigKArgs, bb));
            const int z = zsel;
            p.A = z == 1 ? bb.A[0] : z == 2 ? bb.A[1] : bb.A[2];
            p.w_scale = z == 1 ? bb.w_scale[0] : z == 2 ? bb.w_scale[1] : bb.w_scale[2];
            p.W_split = z == 1 ? bb.W_split[0] : z == 2 ? bb.W_split[1] : bb.W_split[2];
            p.bias = z == 1 ? bb.bias[0] : z == 2 ? bb.bias[1] : bb.bias[2];
            p.C = z == 1 ? bb.C[0] : z == 2 ? bb.C[1] : bb.C[2];
        }
        K = p.Cin * TAPS;
        nkt = K / 32;
        nwg = tiles_m * tiles_n;
        nblk = (int)(p.M >> 5);
        rb = p.row_blocks;
        a_pstride = p.lda * 32, w_pstride = (int64_t)K * 32;
    };
    reload();
    const int stagger = PERSIST ? 0 : load_karg<int>(kp, offsetof(BigKArgs, stagger));
    unsigned* rflag = nullptr;
    // Phase stagger.  Every tile of a launch takes the same time, so without it all CUs reach their epilogues together and
    // 256 x 256 KiB of stores meet an HBM that was idle a moment before.  The first workgroup of every CU (the first 256 of
    // the grid: one per CU) starts `slot * stagger` x 1,024 cycles late, slot = its place among the 32 CUs of its XCD; the
    // offsets then persist through the launch (a CU's next workgroup starts when its last one ends).
    if (stagger && blockIdx.y == 0 && blockIdx.x < 256) {
        const int n = ((blockIdx.x >> 3) & 31) * stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = WH * 4 + (__builtin_amdgcn_readfirstlane(tid >> 6) & 3);
    const int wm = WH, wn = wave & 3;
    // ---- tiles.  Virtual block id -> tile through the XCD-aware renumbering; the tile's rows are eight 32-row blocks,
    // slots tm*8 .. tm*8+7 of the block list (identity without one).  With a padding map (vrd_row_blocks) the list is
    // cut into segments -- one per XCD's contiguous share of the tiles when there are eight -- and inside a segment
    // the blocks holding valid frames come first, so a tile is either a contraction tile or, behind those, a tile of
    // fully padded blocks that only runs the epilogue on a zero accumulator (the reference's value wherever row_mask
    // zeroes the row).
    auto blk_of = [&](int slot) { return slot < nblk ? (rb ? vrd::uniform_load(rb + slot) : slot) : -1; };
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
            t.contract = t.tm * 8 < nblk && t.tm * 8 - seg * seg_len < vrd::uniform_load(p.row_blocks_active + seg);
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
    // per-lane byte offsets inside a descriptor, even / odd pieces (source-side swizzle); the same for every tile
    unsigned va[2], vw[2];
    auto lane_offsets = [&]() {
        const unsigned a_lane = (unsigned)rin * (unsigned)(p.lda * 4), w_lane = (unsigned)rin * (unsigned)(K * 4);
        va[0] = a_lane + chunk0, va[1] = a_lane + (chunk0 ^ 64);
        vw[0] = w_lane + chunk0, vw[1] = w_lane + (chunk0 ^ 64);
    };
    lane_offsets();
    struct Src {
        int tseq0, a_in, w_last;
        __amdgpu_buffer_rsrc_t ra, rw;          // wave-uniform descriptors: the wave's A block (one row back for k = 3), its W rows
    };
    auto src_of = [&](const Tile& t) {
        const int my_blk = blk_of(t.tm * 8 + wave);          // the block whose A rows this wave stages
        const int64_t a_row = (int64_t)(my_blk < 0 ? 0 : my_blk) * 32 + rin;
        Src r;
        r.tseq0 = (TAPS == 3) ? (int)(a_row % p.T) : 0;
        r.a_in = my_blk < 0 ? 0 : PER;                                   // pieces inside (M % 32 == 0)
        const int w_in = (p.N - t.n0 - wave * PER * 8 + 7) / 8;
        // weight pieces beyond N re-read the last piece inside (a wave entirely beyond N: the tile's first rows): columns
        // >= N are never stored, and a uniform minimum costs the loop less than a per-lane out-of-range offset
        r.w_last = (w_in > 0 ? (w_in < PER ? w_in : PER) : 1) - 1;
        // (k = 1: a wave whose block lies outside the matrix reads block 0 -- its accumulator rows are never stored)
        const int64_t a_row_u = (int64_t)(my_blk < 0 ? 0 : my_blk) * 32 - (TAPS == 3 ? 1 : 0);
        r.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + a_row_u * p.lda), 0, 0x80000000u, 0x00020000);
        r.rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(p.W_split) + (int64_t)(t.n0 + (w_in > 0 ? wave * PER * 8 : 0)) * K * 4), 0,
            0x80000000u, 0x00020000);
        return r;
    };
    // piece i of W / A of K step `ks` of the tile `c` describes, into the ring slot of GLOBAL step `gk` (the ring runs on
    // across the tiles of a persistent workgroup: gk = steps of the earlier tiles + ks): one DMA instruction each
    auto issue_w1 = [&](const Src& c, int gk, int ks, int i) {
        char* const dst = lds + W_RING + (gk % NW_STG) * W_STAGE + wave * PER * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rw, (lds_ptr_t)(dst + i * 1024), 16, vw[i & 1],
                                                 (i < c.w_last ? i : c.w_last) * (int)w_pstride + ks * 128, 0, 0);
    };
    auto issue_a1 = [&](const Src& c, int gk, int ks, int i) {
        char* const dst = lds + (gk % NA_STG) * A_STAGE + wave * PER * 1024;
        const int k0 = ks * 32;
        int tap = 0, ci0 = k0;
        unsigned vo = va[i & 1];
        if (TAPS == 3) {
            tap = (k0 >= p.Cin) + (k0 >= 2 * p.Cin);
            ci0 = k0 - tap * p.Cin;
            int tt = c.tseq0 + 8 * i;                   // position of this piece's row in its sequence (T >= 32)
            if (tt >= p.T) tt -= p.T;
            tt += tap - 1;
            if (tt < 0 || tt >= p.T || i >= c.a_in) vo = 0x80000000u;       // zero padding of the taps / a block outside the matrix
        }
        // (k = 3: the descriptor starts one row early, so tap t is t rows further)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.ra, (lds_ptr_t)(dst + i * 1024), 16, vo,
                                                 i * (int)a_pstride + tap * (int)(p.lda * 4) + ci0 * 4, 0, 0);
    };
    auto issue_w = [&](const Src& c, int gk, int ks) {
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_w1(c, gk, ks, i);
    };
    auto issue_a = [&](const Src& c, int gk, int ks) {
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_a1(c, gk, ks, i);
    };

    bool staged = false;          // PERSIST: A(0), A(1), W(0) of this tile were requested inside the previous tile's last two K steps
    bool stored_all = false;      // PERSIST: this wave stored both 64-row halves in the previous tile's epilogue (>= 32 stores)
    int g0 = 0;                   // PERSIST: K steps of this workgroup's earlier tiles, modulo 6 (the rings run on across tiles)
    // PERSIST: vq walks the tiles of all `count` problems of the launch, problem after problem; vb = the tile inside its problem
    const int count = PERSIST ? load_karg<int>(kp, offsetof(BigKArgs, count)) : 1;
    int vq = blockIdx.x, vb = blockIdx.x;
#ifdef VRD_BIG_NOCARRY
    constexpr bool CARRY = false;
#else
    constexpr bool CARRY = true;
#endif
    Tile nt_keep = Tile{0, 0, false};
    Src nx_keep = src_of(nt_keep);
    do {
    if (PERSIST) {
        zsel = (vq >= nwg) + (vq >= 2 * nwg) + (vq >= 3 * nwg);      // vq / nwg for at most four problems (nwg is the same for all)
        vb = vq - zsel * nwg;
        if (vq != (int)blockIdx.x || zsel) reload();      // (the per-lane request offsets do not depend on the tile: not recomputed)
    }
    LAB_STAMP(0);
    LAB_REAL(4);
    // (PERSIST: a tile whose first stages were requested by the previous one was described then: tile and sources are carried over
    // instead of being worked out again -- the set-up is bound by its scalar instructions, 8 waves on the CU's one scalar unit)
    const Tile tile = (PERSIST && staged && CARRY) ? nt_keep : tile_of(vb);
    const int tm = tile.tm, n0 = tile.n0;
    const bool contract = tile.contract;
    const Src cur = (PERSIST && staged && CARRY) ? nx_keep : src_of(tile);
    // PERSIST: the tile this workgroup computes next, if it is a contraction tile too (its first stages are requested inside
    // this tile's last two K steps); otherwise those requests re-read this tile's own first stages into the free ring slots
    // (harmless, keeps the loop free of tests) and the next tile starts like a first one
    bool next_staged = false;
    Src nx = cur;
    if (PERSIST && contract && vb + (int)gridDim.x < nwg) {      // (inside the same problem)
        const Tile nt = tile_of(vb + (int)gridDim.x);
        if (nt.contract) {
            nx = src_of(nt);
            next_staged = true;
            nt_keep = nt;
        }
    }
    nx_keep = nx;
    if (!contract && tid == 0 && tm * 8 < nblk)
        atomicAdd(&g_big_skipped_kn, (unsigned long long)K * (unsigned)(p.N - n0 < TN ? p.N - n0 : TN));

    // ---- fragment read offsets (bytes inside a stage).  A tile row is 128 bytes: hi chunks 0..3, lo chunks 4..7, chunk index
    // XOR-swizzled with (row >> 1) & 7.  v_mfma_f32_16x16x32: lane l holds row (or column) l & 15, k = 8 * (l >> 4) .. + 7 of the
    // K step, i.e. hi chunk l >> 4 and the lo chunk 64 bytes (XOR) away; the swizzle of row (16-row block) + (l & 15) is that of
    // l & 15, so an operand needs ONE per-lane base per plane and its blocks are immediate offsets of the reads (the bases are
    // laundered: left to itself hipcc keeps a register per block).
    const int l15 = lane & 15, l4 = lane >> 4;
    int a_hi_off = (wm * 128 + l15) * ROWB + ((l4 ^ swz(l15)) * 16), a_lo_off = a_hi_off ^ 64;
    int w_hi_off = W_RING + (wn * 64 + l15) * ROWB + ((l4 ^ swz(l15)) * 16), w_lo_off = w_hi_off ^ 64;
    asm volatile("" : "+v"(a_hi_off), "+v"(a_lo_off), "+v"(w_hi_off), "+v"(w_lo_off));

    // bias of this lane's columns: requested now, used by the epilogue
    // (through loads the compiler does not track: see load_epi_cols_async; they are older than every K-loop request, so the
    // first counted wait of the loop covers them)
    const vrd::EpiCols cols = vrd::load_epi_cols_async(p, n0 + wn * 64, lane);
    // 8 x 4 accumulators of 16 x 16.  A contraction tile's first K step starts every accumulator from the constant 0 (the MFMA's
    // C operand): zeroing 128 registers per tile was ~1 k cycles of vector issue in the set-up of every tile
    vrd::f32x4_t acc[8][4];
    if (!contract) {
        float z;          // (a zero the compiler cannot hoist: it moves plain constant initialisation in front of the branch)
        asm volatile("v_mov_b32 %0, 0" : "=v"(z));
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = z;
    }

    // ---- main loop.  A K step is eight groups g of twelve MFMAs: 16-row block g against the four column blocks, the three
    // products (lo x hi, hi x lo, hi x hi) product-major, so that consecutive MFMAs write different accumulators.  The A
    // fragments of group g+1 are read during group g; the W fragments of the NEXT step go into the registers of the current
    // ones as those are used for the last time (the step's last group runs block-major for that: block t's three products,
    // then its reload -- a second set of weight fragments would be 32 registers the loop does not have); for the reads to
    // cross K steps the step's barrier sits BEFORE its last group: by then every fragment of the step is in registers
    // (lgkmcnt(0)), so the barrier both publishes stage kt+1 (every wave waited for its own pieces first) and
    // frees the buffers of stage kt, which the DMAs issued after it refill: W(kt+2), then A(kt+3), one per
    // group over the next eight groups (a DMA issue stalls its wave for 100-200 cycles while MFMAs run; the
    // two waves of a SIMD place theirs half a group apart).
    struct AF { e16x8 hi, lo; };
    struct WF { e16x8 hi[4], lo[4]; };
    auto load_a = [&](const char* sa, int mi) {
        AF f;
        f.hi = *reinterpret_cast<const e16x8*>(sa + a_hi_off + mi * 16 * ROWB);
        f.lo = *reinterpret_cast<const e16x8*>(sa + a_lo_off + mi * 16 * ROWB);
        return f;
    };
    auto load_w1 = [&](const char* sw, int t, WF& f) {       // column block t
        f.hi[t] = *reinterpret_cast<const e16x8*>(sw + w_hi_off + t * 16 * ROWB);
        f.lo[t] = *reinterpret_cast<const e16x8*>(sw + w_lo_off + t * 16 * ROWB);
    };
    if (contract) {
    if (!(PERSIST && staged)) {
        issue_a(cur, g0, 0);
        issue_w(cur, g0, 0);
        issue_a(cur, g0 + 1, 1);
        issue_w(cur, g0 + 1, 1);
        LAB_STAMP(1);
        // stage 0: what was issued after A(0), W(0) may stay in flight (A(2) follows inside step 0, see below)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    } else {
        // In this wave's queue, oldest first: A(0), W(0), A(1) (requested inside the previous tile's last two K steps), that
        // tile's epilogue loads and stores, and now W(1).  The counter retires in order, so the wait for W(0) must not ask
        // for more than it needs: a wave that stored its whole sub-tile has >= 32 epilogue operations behind A(1) and leaves
        // the youngest 40 (W(1), 32 stores, A(1)) alone -- it does not wait for its stores to be acknowledged.
        issue_w(cur, g0 + 1, 1);
        LAB_STAMP(1);
        if (stored_all) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER + 32) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    }
    LAB_STAMP(6);
    __builtin_amdgcn_s_barrier();
    LAB_STAMP(7);
    // The two waves of a SIMD share its matrix pipe, and between equals the older one (waves 0-3) wins every arbitration: it
    // runs its MFMAs of a K step nearly alone, waits ~1,100-1,300 cycles at the step's barrier, and the younger one then
    // issues its requests with nobody to cover their stalls (profiles/r05_lab_gemm_tile_stamps.txt).  Issue priority BY
    // PROGRESS evens them out: a wave runs the quarters of its K step at priorities 3, 2, 1, 0, so whichever of the two is
    // behind wins (round 5: the barrier wait of the older half dropped to ~650 cycles, the K step from ~3,700 to ~3,550,
    // the whole step by 1.5 ms; a static priority for the younger half: no gain; two levels per step: half the gain).
    WF w_cur;
#pragma unroll
    for (int t = 0; t < 4; ++t) load_w1(lds + (g0 % NW_STG) * W_STAGE, t, w_cur);
    AF a_cur = load_a(lds + (g0 % NA_STG) * A_STAGE, 0), a_nxt = a_cur;
    LAB_PHASE_DECL;
    // DMA slot d (0..7) of the batch opened by the barrier inside step kt: W(kt+2) pieces 0..3, A(kt+3) pieces 0..3
    // Which of these requests exist depends only on where the step stands in the K loop, so the loop body exists in four
    // copies with the tests resolved at compile time: first step, steady steps (1 .. nkt-3), the step before the last and the
    // last (the host sends K loops of three steps or more: vrd_gemm.hip, choose_x3).  (Per-slot scalar branches in one
    // generic body cost the steady loop 7 %: 4,300 -> 3,730-4,060 cycles per K step.)
    enum { POS_FIRST, POS_STEADY, POS_PEN, POS_LAST };
    auto kstep = [&](int kt, auto pos_c) __attribute__((always_inline)) {
        constexpr int POS = decltype(pos_c)::value;
        // W(kt_open+2) pieces / A(kt_open+3) pieces exist?
        auto dma_slot = [&](int kt_open, int d, bool w_ok, bool a_ok) {
            if (d < PER) {
                if (w_ok) issue_w1(cur, g0 + kt_open + 2, kt_open + 2, d);
            } else {
                if (a_ok) issue_a1(cur, g0 + kt_open + 3, kt_open + 3, d - PER);
            }
        };
        // PERSIST: the same slots of the tile's last two steps, where this tile has nothing left to request, carry the NEXT
        // tile's first stages -- the rings run on: its step j is global step g0 + nkt + j
        auto dma_slot_next = [&](int j_w, int j_a, int d) {
            if (d < PER) issue_w1(nx, g0 + nkt + j_w, j_w, d);
            else issue_a1(nx, g0 + nkt + j_a, j_a, d - PER);
        };
        const char* sa = lds + ((g0 + kt) % NA_STG) * A_STAGE;
        const char* sa1 = lds + ((g0 + kt + 1) % NA_STG) * A_STAGE;
        const char* sw1 = lds + ((g0 + kt + 1) % NW_STG) * W_STAGE;
        constexpr bool last = POS == POS_LAST;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // four priority levels, a quarter of the step each
            if (g == 0) __builtin_amdgcn_s_setprio(3);
            if (g == 2) __builtin_amdgcn_s_setprio(2);
            if (g == 4) __builtin_amdgcn_s_setprio(1);
            if (g == 6) __builtin_amdgcn_s_setprio(0);
            // ---- reads for what comes next
            if (g < 7) a_nxt = load_a(sa, g + 1);
            if (g == 7 && !last) a_nxt = load_a(sa1, 0);      // (the barrier was passed at the end of group 6)
            // ---- the group's twelve MFMAs, this wave's DMA of the group in the middle (waves 0-3)
            // or at the end (waves 4-7): WH = wave >> 2 is a compile-time constant, the loop exists once per half
            auto dma_of_group = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
#ifdef VRD_LAB_NODMA
                if (POS == POS_STEADY) return;
#endif
                // groups 0..6 carry slots 1..7 of the batch opened in step kt-1, group 7 slot 0 of this step's
                if (g < 7) {
                    // the batch opened in step kt-1: W(kt+1) pieces 1..3, A(kt+2) pieces 0..3
                    if (POS == POS_FIRST) {
                        if (g < PER) issue_a1(cur, g0 + 2, 2, g);                       // step 0 has no batch of its own yet
                    } else if (POS == POS_STEADY) {
                        dma_slot(kt - 1, g + 1, true, true);
                    } else if (POS == POS_PEN) {
                        // W(nkt-1) pieces 1..3; then (PERSIST) the next tile's A(0)
                        if (g + 1 < PER) dma_slot(kt - 1, g + 1, true, false);
                        else if (PERSIST) dma_slot_next(0, 0, g + 1);
                    } else if (POS == POS_LAST && PERSIST) {
                        dma_slot_next(0, 1, g + 1);        // the next tile's W(0) pieces 1..3, A(1) pieces 0..3
                    }
                } else if (!last) {
                    // slot 0 of this step's own batch: W(kt+2) piece 0 -- in the step before the last (PERSIST) the next
                    // tile's W(0) piece 0
                    if (POS == POS_PEN && PERSIST) dma_slot_next(0, 0, 0);
                    else dma_slot(kt, 0, POS != POS_PEN, false);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // (block-major in a step's last group: a column block's weight fragments are reloaded right behind its third product)
            constexpr bool blockmajor = !last;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                const int pr = (blockmajor && g == 7) ? q % 3 : q >> 2, t = (blockmajor && g == 7) ? q / 3 : q & 3;
                const vrd::f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
                acc[g][t] = vrd::mfma16(pr == 0 ? a_cur.lo : a_cur.hi, pr == 1 ? w_cur.lo[t] : w_cur.hi[t], (POS == POS_FIRST && pr == 0) ? zero4 : acc[g][t]);
                if (blockmajor && g == 7 && pr == 2) load_w1(sw1, t, w_cur);
                if ((q == 5 && WH == 0) || (q == 11 && WH == 1)) dma_of_group();
            }
            a_cur = a_nxt;
            if (g == 6 && !last) {
                // every fragment of stage kt is in registers or landed; stage kt+1 must be visible before group 7
                // starts reading it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                LAB_PHASE(3);
                if (POS != POS_PEN || PERSIST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                LAB_PHASE(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                LAB_PHASE(1);
            }
        }
    };
    kstep(0, std::integral_constant<int, POS_FIRST>{});
    for (int kt = 1; kt + 2 < nkt; ++kt) kstep(kt, std::integral_constant<int, POS_STEADY>{});
    kstep(nkt - 2, std::integral_constant<int, POS_PEN>{});
    kstep(nkt - 1, std::integral_constant<int, POS_LAST>{});
    LAB_PHASE_FLUSH(WH);
    __builtin_amdgcn_s_setprio(0);
    }       // contract
#ifdef VRD_LAB_STAMP
    asm volatile("" ::"v"(acc[0][0][0]));
#endif
    // every wave must be done with the rings before they are reused as epilogue staging
    if (!contract) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the bias loads; otherwise waited for in step 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    LAB_STAMP(2);
    rflag = load_karg<unsigned*>(kp, offsetof(BigKArgs, rflag));
    if (PERSIST) {
        reload();          // (the epilogue's arguments are read here, not carried across the main loop)
        staged = next_staged;
    }
    // staging slab of this wave: 64 rows at the front of LDS, or (PERSIST) 32 rows inside the two ring slots the next tile's
    // requested stages do not occupy: the slots its A(2) (waves 0-3) and its W(1) (waves 4-7) will take
    const int slab_a = ((g0 + nkt + 2) % NA_STG) * A_STAGE, slab_w = W_RING + ((g0 + nkt + 1) % NW_STG) * W_STAGE;
    float* const stg = PERSIST ? smem + ((wave < 4 ? slab_a : slab_w - 4 * (32 * vrd::STG_PITCH * 4)) + wave * (32 * vrd::STG_PITCH * 4)) / 4
                               : smem + wave * (64 * vrd::STG_PITCH);
    bool all_stored = true;
    constexpr int SLAB = PERSIST ? 32 : 64;
    // (a fresh copy of the lane index per tile: what the epilogue derives from it is computed here, behind the main loop --
    // hoisted in front of the tile loop, those values held ~40 registers across a K loop that has none to spare)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#pragma unroll
    for (int hm = 0; hm < 2; ++hm) {          // the epilogue works on 64 x 64 halves of the wave's 128 x 64
        // (M % 64 == 0 and N % 64 == 0, checked on the host: the sub-tile is inside C or entirely outside)
        const int slot = tm * 8 + wm * 4 + hm * 2;
        const int blk_a = blk_of(slot), blk_b = blk_of(slot + 1);
        const int nw = n0 + wn * 64;
        if (blk_a < 0 || nw >= p.N) {
            all_stored = false;
            continue;
        }
        const int64_t mw = (int64_t)blk_a * 32, mw1 = (int64_t)blk_b * 32;      // rows of passes 0-1 / 2-3
        // (PERSIST: the host sends no GEMM with per-row epilogue inputs here.  Their loads are the only vector loads of the tile
        // loop the compiler tracks, and with them in the loop it opens every tile with `s_waitcnt vmcnt(0)` -- a wait for the
        // previous tile's stores to be acknowledged; their waits also queue behind the look-ahead requests.)
        const bool rowin = !PERSIST && (p.row_mask || p.scale || p.res || p.res2);
        vrd::f32x4_t part[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) part[i][j] = acc[4 * hm + i][j];
        if (rowin) vrd::gemm_epilogue_lean16<true, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane_e, cols, rflag);
        else if (p.act == VRD_ACT_GELU) vrd::gemm_epilogue_lean16<false, VRD_ACT_GELU, SLAB>(p, part, stg, mw, mw1, nw, lane_e, cols, rflag);
        else vrd::gemm_epilogue_lean16<false, VRD_ACT_NONE, SLAB>(p, part, stg, mw, mw1, nw, lane_e, cols, rflag);
    }
    LAB_STAMP(3);
    LAB_REAL(5);
    if (PERSIST) {
        // the slabs lie where the next tile's W(1) and A(2) land: everybody is done reading theirs first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    stored_all = all_stored;
    g0 = (g0 + nkt) % 6;
    vq += (int)gridDim.x;
    } while (PERSIST && vq < nwg * count);       // tiles of this workgroup
    // (the last tile's look-ahead requests wrote into this workgroup's LDS: nothing may be in flight when it is handed on)
    if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <int TAPS, bool PERSIST, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_x3_big_kernel(BigKArgs ka_unused_) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) gemm_x3_big_body<TAPS, PERSIST, F16, 1>();
    else gemm_x3_big_body<TAPS, PERSIST, F16, 0>();
}

}  // namespace

namespace vrd {

template <int TAPS, bool PERSIST, bool F16 = false>
static int launch_big_one(const vrd_gemm_args& a, hipStream_t s, const BigBatch& bb = BigBatch{}, int count = 1) {
    auto kern = gemm_x3_big_kernel<TAPS, PERSIST, F16>;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), BIG_LDS, "vrd_gemm(bf16x3 256x256)")) return rc;
    const int tiles_m = (int)((a.M + TM - 1) / TM), tiles_n = (a.N + TN - 1) / TN;
    const int nwg = tiles_m * tiles_n;
    // (measured, scripts/dev/stagger_sweep.sh, profiles/r04_lab_gemm_stagger.txt: 0 / 1 / 2 / 4 / 8 units -> 98.1-98.2 / 97.4 / 97.1 /
    // 97.8 / 99.4 ms of this kernel per step: about 1 %, so the epilogues were not waiting for each other's stores much)
    static const int stagger = [] { const char* e = getenv("VRD_BIG_STAGGER"); return e ? atoi(e) : 2; }();
    const int n_cu = PERSIST ? device_cu_count() : 0;
    BigKArgs ka;
    ka.p = a, ka.tiles_m = tiles_m, ka.tiles_n = tiles_n, ka.bb = bb, ka.stagger = (PERSIST || nwg < 512) ? 0 : stagger;
    ka.rflag = a.c_pair == VRD_PAIR_F16 ? range_flag() : nullptr;
    ka.count = count;
    if (PERSIST) hipLaunchKernelGGL(kern, dim3(nwg * count < n_cu ? nwg * count : n_cu), dim3(512), BIG_LDS, s, ka);
    else hipLaunchKernelGGL(kern, dim3(nwg, count), dim3(512), BIG_LDS, s, ka);
    return 0;
}

// The persistent form (one workgroup per CU walking its tiles, the next tile's first stages requested inside the current tile's
// last K steps) takes k = 1 GEMMs without per-row epilogue inputs and at least three K steps; VRD_BIG_PERSIST=0 switches it off.
static bool big_persist(const vrd_gemm_args& a) {
    static const int persist = [] { const char* e = getenv("VRD_BIG_PERSIST"); return e ? atoi(e) : 1; }();
    return persist && a.taps == 1 && a.Cin >= 96 && !(a.row_mask || a.scale || a.res || a.res2);
}

static int launch_big_any(const vrd_gemm_args& a, hipStream_t s, const BigBatch& bb, int count) {
    const bool f16 = a.split_fmt == VRD_PAIR_F16;
    if (big_persist(a)) return f16 ? launch_big_one<1, true, true>(a, s, bb, count) : launch_big_one<1, true, false>(a, s, bb, count);
    if (f16) return a.taps == 1 ? launch_big_one<1, false, true>(a, s, bb, count) : launch_big_one<3, false, true>(a, s, bb, count);
    return a.taps == 1 ? launch_big_one<1, false, false>(a, s, bb, count) : launch_big_one<3, false, false>(a, s, bb, count);
}

// `count` (2 .. 4) problems that differ only in A, W_split, bias and C, as one launch
int launch_gemm_x3_big_batch(const vrd_gemm_args* a, int count, hipStream_t s) {
    BigBatch bb{};
    for (int i = 1; i < count; ++i) {
        bb.A[i - 1] = a[i].A;
        bb.W_split[i - 1] = a[i].W_split;
        bb.bias[i - 1] = a[i].bias;
        bb.C[i - 1] = a[i].C;
        bb.w_scale[i - 1] = a[i].w_scale;
    }
    return launch_big_any(a[0], s, bb, count);
}

// FLOPs of contractions skipped through padding maps since the last call (reads and clears the device counter)
double take_big_skipped_flops() {
    unsigned long long v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_big_skipped_kn), sizeof(v)) != hipSuccess) return 0.0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_big_skipped_kn), &zero, sizeof(zero));
    return 2.0 * TM * (double)v;
}

// same eligibility as the 128 x 256 DMA kernel (pair-row A, staged epilogue); the caller picks by tile count
int launch_gemm_x3_big(const vrd_gemm_args& a, hipStream_t s) { return launch_big_any(a, s, BigBatch{}, 1); }

}  // namespace vrd
