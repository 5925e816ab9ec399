// Split-precision conv GEMM, 256 x 256 tile, ACTIVATIONS STRAIGHT INTO REGISTERS (see vrd_gemm_x3.hip for the arithmetic
// and vrd_gemm_x3_big.hip for the kernel this one grew out of).  LAB ONLY since round 3 (scripts/lab/gemm_lab.hip; it used to be an opt-in of the library, VRD_BIG_ROW=1), k = 1 shapes with Cin % 128 == 0:
// a study of what the 256 x 256 kernel gains when half of its LDS-DMA traffic disappears; measured result at the end.
//
// What limits the 256 x 256 LDS-DMA kernel is the issue of its LDS-DMA instructions: a 1-KiB global_load_lds costs the CU
// ~45 cycles while MFMAs run, a K step of 32 moves 32 KiB of activations + 32 KiB of weights = 64 of them, ~2,900 cycles
// beside 3,072 cycles of MFMA, and the step takes 4,170-4,300.  The activations do not need LDS at all if a wave owns its
// rows: here the 8 waves split the tile by ROWS -- wave w computes rows 32w .. 32w+31 (one 32-row block of the padding
// map) times all 256 columns, 1 x 8 accumulators of 32 x 32 -- so an activation row is read by exactly one wave, as
// ordinary 16-byte global loads that land in MFMA operand layout: a pair row's 128-byte K-step line [32 hi | 32 lo] holds,
// for lane (row li, k half lh), the four fragments hi/lo x k16 half at bytes lh*16 + {0, 32, 64, 96}.  Only the weights go
// through LDS (every wave needs all of them): half the DMA instructions per step, the other half replaced by 4 vector
// loads per lane.  The activations stream from HBM with a latency of several thousand cycles, so their loads run THREE K
// steps ahead into four rotating register buffers (64 VGPRs; the K loop is unrolled by four, hence K % 128 == 0); weight
// fragments are read from LDS three MFMA groups ahead into a ring of four.
//   LDS: W ring 2 x 32 KiB = 64 KiB | epilogue slabs 8 x 8 KiB = 64 KiB (not aliased with the ring)
//   per K step t and wave: MFMA groups q = 0..15 (k16 half q >> 3, column block q & 7, three MFMAs each);
//   groups 0..3 issue the wave's four loads of A(t+3); at group 12 every LDS read of the step has been issued: wait for
//   them and for the own pieces of W(t+1) (counted vmcnt: A(t+3) stays in flight), barrier; groups 12..15 then issue the four
//   DMAs of W(t+2) into the stage just freed and groups 13..15 already read the first fragments of step t+1.
// Every memory operation of the loop is inline assembly with hand-counted s_waitcnt: with compiler-visible loads the
// wait-count pass drains all loads at the loop head (the buffers cross the back edge), and next to ANY inline-asm statement
// it stops counting LDS operations and waits lgkmcnt(0) before every fragment use.
// Same products in the same order as the other split-precision kernels (per K step and k16 half: a_lo*w_hi, a_hi*w_lo,
// a_hi*w_hi), so results do not depend on which kernel a batch size selects (the GPU suite passes with VRD_BIG_ROW=1).
//
// Measured (scripts/lab/gemm_lab.hip, per 256 x 256 tile, cycles at ~2.1 GHz; LDS-DMA kernel -> this one):
//   K step      4,100-4,330 -> 3,600-3,800      (3,500 with either operand's loads removed: barrier and issue overhead)
//   tile start  4,500-5,500 -> 7,300-11,000     (see below)
//   epilogue    8,400-9,500 -> 9,500-10,800     (four 32 x 64 pieces instead of two 64 x 64)
//   GEMM alone  M 147k-590k, N 512 / 2048:  K = 512: +1 .. +4 %,  K = 1024: +7 %,  K = 2048: +8 .. +15 %
//   whole step  96.3 -> 95.3 ms of this kernel family, 145.0 -> 144.0 ms per step  (+0.7 %)
// The faster loop needs two activation steps in flight from HBM before it runs at its rate, and a CU's share of the HBM
// stream is ~15 B/clk: whichever way the first requests are ordered (all three steps up front; first stage alone behind a
// barrier so that everybody's first stage is served first; first two steps by LDS-DMA into the idle epilogue slab, which is
// what the code does: gathers touch 32 rows per instruction and are ~3 x slower on first touch), the time until the loop
// reaches its rate stays ~10 k cycles -- 6 k more than the LDS-DMA kernel's, i.e. what 16 faster K steps save.  A persistent
// variant (next tile's first requests issued under the epilogue, which works in its own LDS here) was built and measured:
// tile start 2.2-3 k, but the epilogue then runs with the prefetch registers live on top of 128 accumulators: 52-170
// spilled VGPRs, epilogue 15-30 k cycles, slower overall; it is not kept.
//
// Two workgroups per CU (VRD_BIG_ROW=2, NWV = 4 below): the other way to put one tile's start and epilogue under another
// tile's K loop -- 128 x 256 tiles of 4 waves x (32 rows x 256 columns), 64 KiB of LDS each (the epilogue slabs alias the
// ring), 246 VGPRs, no spills; the two workgroups of a CU run out of phase by themselves.  Measured (same harness, K = 512,
// M = 590k): K step 3,840 cycles PER WORKGROUP -- the two waves a SIMD now holds from independent workgroups share the
// MFMA pipe exactly as the two of one workgroup did, 3,072 cycles of MFMA per 256 x 256 of work in ~3,850 -- but tile start
// 14.7 k and epilogue 13.1 k cycles while the neighbour computes (8.5 k / 10.4 k alone), and a workgroup that has the CU's
// MFMA pipes to itself during those phases does not run its loop faster: 89 k cycles per 128 x 256 tile, two at a time, =
// 0.798 ms against 0.727 ms of the LDS-DMA kernel and 0.771 ms of the 8-wave form; K = 2048: 0.630 / 0.650 / 0.644 ms.
// Whole step 144.0 ms (LDS-DMA kernel 141.2, 8-wave form 141.9).  Bit-identical results (GPU suite green); stays opt-in.
#include "../../vrdone_amd/csrc/vrd_common.h"
#include "../../vrdone_amd/csrc/vrd_gemm_epilogue.h"
#include <cstdlib>
#include <type_traits>

// (lab) operands of problems 1 .. 3 of a batched launch; the product's batched kernel carries its own BigBatch
namespace vrd {
struct GemmBatch {
    const float* A[3];
    const uint16_t* W_split[3];
    const float* bias[3];
    float* C[3];
};
}  // namespace vrd

#ifndef LAB_STAMP           // the lab harness (scripts/lab/gemm_lab.hip) defines these through vrd_gemm_x3_big.hip
#define LAB_STAMP(slot)
#define LAB_REAL(slot)
#endif
#ifdef VRD_LAB_STAMP        // lab ablations (timing only): 1 = no activation loads in the loop, 2 = no weight DMAs in the loop
#define LAB_MODE lab_mode_v
#else
#define LAB_MODE 0
#endif

namespace {
namespace rowk {       // (own namespace: the lab harness compiles this file and vrd_gemm_x3_big.hip as one unit)

using vrd::f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int TN = 256;
constexpr int ROWB = 128;                      // bytes of a tile row per K step (32 hi + 32 lo bf16)
constexpr int W_STAGE = TN * ROWB;
constexpr int NW_STG = 2;
constexpr int NA = 4;                          // activation loads per lane and K step
// NWV waves, each 32 rows x 256 columns.  8: one 256 x 256 workgroup per CU, epilogue slabs behind the ring.
// 4: 128 x 256 tiles, TWO workgroups per CU (64 KiB of LDS and 4 x 256 VGPRs each): the two run out of phase, so one's
// tile start and epilogue overlap the other's K loop; the epilogue slabs then alias the ring (a barrier after the last
// step), and only A(0) comes by DMA -- into the wave's 4 KiB of ring stage 1, which W(1) overwrites behind the barrier.
template <int NWV>
struct RowGeo {
    static constexpr int TM = 32 * NWV;
    static constexpr int PER = 32 / NWV;       // W DMA instructions per wave and K step (8 rows x 128 B each)
    static constexpr bool ALIAS = NWV == 4;
    static constexpr int SLAB_OFF = ALIAS ? 0 : NW_STG * W_STAGE;     // byte offset of the epilogue slabs
    static constexpr size_t LDS = ALIAS ? (size_t)NW_STG * W_STAGE : (size_t)SLAB_OFF + 8 * 32 * vrd::STG_PITCH * sizeof(float);
};

__device__ unsigned long long g_row_skipped_kn;                    // as g_big_skipped_kn (vrd_gemm_x3_big.hip)

__device__ constexpr int swz(int row) { return (row >> 1) & 7; }

template <int TAPS, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void gemm_x3_row_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, vrd::GemmBatch bb) {
    using G = RowGeo<NWV>;
    constexpr int PER = G::PER, SLAB_OFF = G::SLAB_OFF;
    constexpr bool ALIAS = G::ALIAS;
    if (blockIdx.y) {                                    // uniform selects, no indexed access to the arguments
        const int z = blockIdx.y;
        p.A = z == 1 ? bb.A[0] : z == 2 ? bb.A[1] : bb.A[2];
        p.W_split = z == 1 ? bb.W_split[0] : z == 2 ? bb.W_split[1] : bb.W_split[2];
        p.bias = z == 1 ? bb.bias[0] : z == 2 ? bb.bias[1] : bb.bias[2];
        p.C = z == 1 ? bb.C[0] : z == 2 ? bb.C[1] : bb.C[2];
    }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int K = p.Cin * TAPS;
    const int nkt = K / 32;                              // a multiple of 4 (host-checked)
    const int nwg = tiles_m * tiles_n;

    // ---- tile of this workgroup (XCD-aware renumbering and padding-map block list as in the LDS-DMA 256 x 256 kernel)
    const int nblk = (int)(p.M >> 5);
    const int32_t* const rb = p.row_blocks;
    const int vb = blockIdx.x;
    const int xcd = vb & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (vb >> 3);
    const int tm = lid / tiles_n;
    const int n0 = (lid - tm * tiles_n) * TN;
    bool contract = true;
    if (rb) {
        const int seg_len = p.row_block_seg_len;                          // a multiple of 8 (host-checked)
        const int seg = (tm * NWV) / seg_len;
        contract = tm * NWV < nblk && tm * NWV - seg * seg_len < p.row_blocks_active[seg];
    }
    const int slot = tm * NWV + wave;                    // this wave's 32-row block
    const int my_blk = slot < nblk ? (rb ? rb[slot] : slot) : -1;
    if (!contract && tid == 0 && tm * NWV < nblk)
        atomicAdd(&g_row_skipped_kn, (unsigned long long)K * (unsigned)(p.N - n0 < TN ? p.N - n0 : TN) * NWV);

    LAB_STAMP(0);
    LAB_REAL(4);
#ifdef VRD_LAB_STAMP
    const int lab_mode_v = __builtin_amdgcn_readfirstlane(g_lab_mode);
#endif
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    if (contract) {
        // ---- activation rows: lane (li, lh) owns row my_blk*32 + li, bytes lh*16 + {0, 32, 64, 96} of each K-step line:
        // a scalar base (the block's first row, advanced by 512 bytes per group of four K steps) + one 32-bit lane offset.
        // (a wave whose block lies outside the matrix reads block 0: its accumulators are never stored)
        const char* a_base = reinterpret_cast<const char*>(p.A + (int64_t)(my_blk < 0 ? 0 : my_blk) * 32 * p.lda);
        const unsigned a_off = (unsigned)(li * (int)p.lda * 4 + lh * 16);
        // fragment i of a K step: 0 = hi of k16 half 0, 1 = hi of half 1, 2 = lo of half 0, 3 = lo of half 1.
        // The loads are inline assembly on purpose: the compiler's wait-count insertion drains every load in flight at the
        // head of a loop whose body consumes loads of the previous iteration, which would stop the three-steps-ahead
        // stream every four steps.  An asm load is invisible to it; the counted waits before the barriers (below) are what
        // guarantees that a buffer has landed before its step.  "+v": a buffer keeps its registers across the load.
#define VRD_ROW_LOAD_A(dst, base, u, i) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(a_off), "s"(base), "n"(128 * (u) + 32 * (i)))

        // ---- weight DMA: this wave moves rows wave*32 .. wave*32+31 of the stage, 8 rows x 128 B per instruction; the
        // 16-byte chunks of a row are XOR-swizzled with (row >> 1) & 7 on the source address (and on the reads): scalar
        // base of the wave's rows + the lane's offset in an 8-row piece (bit 6 flipped for the odd pieces: rows +8, +24).
        // (rows beyond N re-read the tile's first rows: those columns are never stored)
        const int rin = lane >> 3, pch = lane & 7;
        const int chunk0 = (pch ^ swz(wave * PER * 8 + rin)) * 16;
        const int w_in = (p.N - n0 - wave * PER * 8 + 7) / 8;              // pieces inside the matrix (N % 64 == 0)
        const char* const w_base = reinterpret_cast<const char*>(p.W_split) + (int64_t)(n0 + (w_in > 0 ? wave * PER * 8 : 0)) * K * 4;
        const unsigned w_off[2] = {(unsigned)(rin * K * 4 + chunk0), (unsigned)(rin * K * 4 + chunk0) ^ 64u};
        const int64_t w_pstride = (int64_t)K * 32;                         // bytes between pieces (8 rows)
        // (inline assembly too: through the builtin the compiler keeps one 64-bit per-lane pointer per piece, eight VGPRs the
        // loop does not have; here the piece is a scalar base + the lane offset.  M0 = LDS address of the piece; it is a
        // reserved register, nothing of the compiler's lives in it across statements.)
        const unsigned w_dst = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)lds) + wave * PER * 1024;
        auto issue_w1 = [&](int kt, int stage, int i) {
            const char* const src = w_base + (i < w_in ? i * w_pstride : 0) + (int64_t)kt * 128;
            const unsigned dst = w_dst + stage * W_STAGE + i * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(w_off[i & 1]), "s"(src) : "memory");
        };
        // ---- weight fragments: column block j, k16 half s: row j*32 + li, chunk (2s + lh) hi, +4 lo -- four per-lane
        // bases (half and hi/lo flip bits 5 and 6 of the swizzled chunk), stage and column block are immediates
        // The reads are inline assembly with counted waits as well: next to an inline-asm statement the compiler's wait-count
        // insertion stops counting LDS operations and waits for ALL of them before every use (lgkmcnt(0)), which would expose
        // the latency of the reads just issued.  A group's fragments were requested three groups earlier; the six reads
        // issued since may stay in flight: s_waitcnt lgkmcnt(6).
        const unsigned lds_addr = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)lds);
        const unsigned w_lane = lds_addr + li * ROWB + ((lh ^ swz(li)) * 16);
        const unsigned w_hi[2] = {w_lane, w_lane ^ 32};
        const unsigned w_lo[2] = {w_lane ^ 64, w_lane ^ 96};
        struct WF { bf16x8 hi, lo; };
#define VRD_ROW_LOAD_WF(f, stage, q)                                                                                             \
    do {                                                                                                                         \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"((f).hi) : "v"(w_hi[(q) >> 3]), "n"((stage) * W_STAGE + ((q) & 7) * 32 * ROWB)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"((f).lo) : "v"(w_lo[(q) >> 3]), "n"((stage) * W_STAGE + ((q) & 7) * 32 * ROWB)); \
    } while (0)

        bf16x8 abuf[4][4];
        WF wf[4];
        // ---- prologue.  The gathers above are slow on first touch (a wave instruction touches 32 rows, 32 bytes of each: ~8 k
        // cycles until the first stage is there, against ~2.5 k for whole lines by LDS-DMA), so the first TWO activation
        // steps come by DMA into the wave's epilogue slab (8 KiB, idle until the epilogue) and are read into the register
        // buffers from there; A(2) is gathered behind them and every step t then issues A(t+3) (groups 0..3) and, behind
        // its barrier, W(t+2) (groups 12..15).
        const unsigned a_dma_off = (unsigned)(rin * (int)p.lda * 4 + pch * 16);       // 8 rows x 128 B per instruction
        // (ALIAS: the wave's 4 KiB of ring stage 1 -- free until W(1) is issued behind the first barrier)
        const unsigned a_stg = ALIAS ? lds_addr + W_STAGE + wave * 4096 : lds_addr + SLAB_OFF + wave * (32 * vrd::STG_PITCH * 4);
        auto issue_a_dma = [&](int u, int i) {
            const char* const src = a_base + (int64_t)i * 8 * p.lda * 4 + u * 128;
            const unsigned dst = a_stg + u * 4096 + i * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(a_dma_off), "s"(src) : "memory");
        };
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) abuf[u][i] = bf16x8{};       // (defined before the asm's read-write operands)
        LAB_STAMP(6);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_a_dma(0, i);
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_w1(0, 0, i);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // A(0), W(0) landed
        LAB_STAMP(7);
        // the wave's own rows: slab row li, bytes lh*16 + {0, 32, 64, 96} (no barrier needed for these); step 1's follow
        // behind the barrier of step 0
        const unsigned a_rd = a_stg + li * ROWB + lh * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(abuf[0][i]) : "v"(a_rd), "n"(32 * i));
        if (ALIAS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // out of stage 1 before anybody's W(1) goes there
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int q = 0; q < 3; ++q) VRD_ROW_LOAD_WF(wf[q], 0, q);
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                 // the activation reads (the fragments may stay in flight)
        // the second stage and A(2) go out only now: whatever is requested before the first stage has arrived delays the
        // first MFMA (a CU's share of the HBM stream is ~15 B/clk)
        if (!ALIAS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_a_dma(1, i);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) issue_w1(1, 1, i);
        __builtin_amdgcn_sched_barrier(0);
        if (ALIAS) {                  // no slab outside the ring: A(1) is gathered like A(2)
#pragma unroll
            for (int i = 0; i < 4; ++i) VRD_ROW_LOAD_A(abuf[1][i], a_base, 1, i);
        }
        if (LAB_MODE != 3)
#pragma unroll
        for (int i = 0; i < 4; ++i) VRD_ROW_LOAD_A(abuf[2][i], a_base, 2, i);
        __builtin_amdgcn_sched_barrier(0);
        LAB_STAMP(1);

        // one K step; u = kt & 3 (register buffer, weight stage u & 1); TAIL: the last four steps (kt = nkt - 4 + u)
        auto step = [&](int kt, auto u_c, auto tail_c, const char* grp_cur) __attribute__((always_inline)) {
            constexpr int u = decltype(u_c)::value;
            constexpr bool TAIL = decltype(tail_c)::value;
            constexpr int stage = u & 1;
            constexpr bool last = TAIL && u == 3;
            constexpr bool has_a3 = !TAIL || u < 1;      // A(kt+3) exists
            constexpr bool has_w2 = !TAIL || u < 2;      // W(kt+2) exists
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                // fragment reads three groups ahead; groups 13..15 read the next step's first three (behind the barrier)
                if (q + 3 < 16) VRD_ROW_LOAD_WF(wf[(q + 3) & 3], stage, q + 3);
                else if (!last) VRD_ROW_LOAD_WF(wf[(q + 3) & 3], stage ^ 1, q + 3 - 16);
                // this group's fragments (requested at group q - 3) have landed once at most the reads of the three groups
                // behind them are in flight
                if (!last || q < 13) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (15 - q)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                const int s2 = q >> 3, j = q & 7;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(abuf[u][2 + s2], wf[q & 3].hi, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(abuf[u][s2], wf[q & 3].lo, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(abuf[u][s2], wf[q & 3].hi, acc[j], 0, 0, 0);
                if (q < NA && has_a3) {
                    // A(kt+3) into the buffer step kt-1 used (u + 3 = u - 1 mod 4): step (u + 3) & 3 of its group of four
                    __builtin_amdgcn_sched_barrier(0);
                    // (steps 1..3 load steps 0..2 of the NEXT group of four: 512 bytes on)
                    if (LAB_MODE != 1) VRD_ROW_LOAD_A(abuf[(u + 3) & 3][q], grp_cur, (u == 0 ? 0 : 4) + ((u + 3) & 3), q);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q == 12 && !last) {
                    // every fragment of stage kt is requested: once they are here the stage is free for W(kt+2).  W(kt+1)
                    // (own pieces) and with it the older A(kt+1), A(kt+2) must have landed; A(kt+3) stays in flight
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // (step 0 of the tile: A(2) of the prologue is younger than W(1) as well)
                    const bool tile_start = has_a3 && u == 0 && kt == 0;
                    if (tile_start) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NA) : "memory");
                    else if (has_a3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (tile_start && !ALIAS) {
                        // step 1's activations came by DMA into the slab (prologue) and have landed with W(1): into registers.
                        // (four more reads in flight than the counted fragment waits assume: those only wait a little longer)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(abuf[1][i]) : "v"(a_rd), "n"(4096 + 32 * i));
                    }
                }
                if (q >= 12 && has_w2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (LAB_MODE != 2) {
#pragma unroll
                        for (int i = 0; i < PER / 4; ++i) issue_w1(kt + 2, stage, (q - 12) * (PER / 4) + i);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the groups apart: the scheduler would pull reads and loads far ahead
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        // groups of four steps: step kt0 loads A(kt0+3) of its own group, steps kt0+1 .. kt0+3 load A(kt0+4 .. kt0+6) of the
        // next one; the last group runs with its conditions resolved at compile time
        auto group = [&](int kt0, auto tail_c, const char* cur) __attribute__((always_inline)) {
            step(kt0 + 0, integral_constant<int, 0>{}, tail_c, cur);
            step(kt0 + 1, integral_constant<int, 1>{}, tail_c, cur);
            step(kt0 + 2, integral_constant<int, 2>{}, tail_c, cur);
            step(kt0 + 3, integral_constant<int, 3>{}, tail_c, cur);
        };
        int kt0 = 0;
        for (; kt0 + 4 < nkt; kt0 += 4) {
            group(kt0, false_type{}, a_base);
            a_base += 512;
        }
        group(kt0, true_type{}, a_base);
    }       // contract
    LAB_STAMP(2);
    if (ALIAS) {                      // the slabs lie in the ring: every wave's last fragment reads first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: the wave's 32 rows x 256 columns as four 32 x 64 pieces through its private slab (own LDS region: no
    // barrier, the ring is not touched)
    if (my_blk < 0) return;
    float* const stg = smem + SLAB_OFF / 4 + wave * (32 * vrd::STG_PITCH);       // (8 KiB per wave)
    const int64_t mw = (int64_t)my_blk * 32;
    const bool rowin = p.row_mask || p.scale || p.res || p.res2;
    vrd::EpiCols cols[4];                                // bias / scale of the four pieces, requested together
#pragma unroll
    for (int g = 0; g < 4; ++g) cols[g] = vrd::load_epi_cols(p, n0 + g * 64, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int nw = n0 + g * 64;
        if (nw >= p.N) continue;                         // (N % 64 == 0: a piece is inside or entirely outside)
        if (rowin) vrd::gemm_epilogue_lean_rows32<true, VRD_ACT_NONE>(p, acc[2 * g], acc[2 * g + 1], stg, mw, nw, lane, cols[g]);
        else if (p.act == VRD_ACT_GELU) vrd::gemm_epilogue_lean_rows32<false, VRD_ACT_GELU>(p, acc[2 * g], acc[2 * g + 1], stg, mw, nw, lane, cols[g]);
        else vrd::gemm_epilogue_lean_rows32<false, VRD_ACT_NONE>(p, acc[2 * g], acc[2 * g + 1], stg, mw, nw, lane, cols[g]);
    }
    LAB_STAMP(3);
    LAB_REAL(5);
}

}  // namespace rowk
}  // namespace

namespace vrd {

// the LDS-DMA 256 x 256 kernel's eligibility (checked by the caller), k = 1, and K % 128 == 0 (K loop unrolled by four)
bool gemm_x3_row_ok(const vrd_gemm_args& a) { return a.taps == 1 && a.Cin % 128 == 0; }

int launch_gemm_x3_row_nw(const vrd_gemm_args* a, const GemmBatch& bb, int count, int nwv, hipStream_t s);

// `count` (1 .. 4) problems that differ only in A, W_split, bias and C, as one launch
int launch_gemm_x3_row(const vrd_gemm_args* a, int count, hipStream_t s) {
    GemmBatch bb{};
    for (int i = 1; i < count; ++i) {
        bb.A[i - 1] = a[i].A;
        bb.W_split[i - 1] = a[i].W_split;
        bb.bias[i - 1] = a[i].bias;
        bb.C[i - 1] = a[i].C;
    }
    // VRD_BIG_ROW=1: 8 waves, one 256 x 256 workgroup per CU; =2: 4 waves, two 128 x 256 workgroups per CU
    static const int nwv = [] { const char* e = getenv("VRD_BIG_ROW"); return e && atoi(e) == 2 ? 4 : 8; }();
    return launch_gemm_x3_row_nw(a, bb, count, nwv, s);
}

int launch_gemm_x3_row_nw(const vrd_gemm_args* a, const GemmBatch& bb, int count, int nwv, hipStream_t s) {
    const int tm_rows = 32 * nwv;
    const int tiles_m = (int)((a[0].M + tm_rows - 1) / tm_rows), tiles_n = (a[0].N + rowk::TN - 1) / rowk::TN;
    const dim3 grid(tiles_m * tiles_n, count);
    // k = 1 only (a k = 3 conv keeps the LDS-DMA kernel)
    if (nwv == 4) {
        auto kern = rowk::gemm_x3_row_kernel<1, 4>;
        if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), rowk::RowGeo<4>::LDS, "vrd_gemm(bf16x3 128x256 row)")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), rowk::RowGeo<4>::LDS, s, a[0], tiles_m, tiles_n, bb);
    } else {
        auto kern = rowk::gemm_x3_row_kernel<1, 8>;
        if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), rowk::RowGeo<8>::LDS, "vrd_gemm(bf16x3 256x256 row)")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(512), rowk::RowGeo<8>::LDS, s, a[0], tiles_m, tiles_n, bb);
    }
    return 0;
}

double take_row_skipped_flops() {
    unsigned long long v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(rowk::g_row_skipped_kn), sizeof(v)) != hipSuccess) return 0.0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rowk::g_row_skipped_kn), &zero, sizeof(zero));
    return 2.0 * 32 * (double)v;
}

}  // namespace vrd
