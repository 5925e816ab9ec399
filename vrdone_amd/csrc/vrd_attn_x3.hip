// Global masked attention (the SOS self / cross attention) in split precision: both contractions of flash
// attention as three 16-bit MFMA products each (x = x_hi + x_lo, see vrd_gemm_x3.hip), f32 accumulate,
// f32 softmax.  Inputs q, k, v are pair rows ([hi | lo] 16-bit planes) written by the projection GEMMs.
// Both kernels exist per element format (template parameter F16, vrd_common.h): bf16 planes, or f16 planes of the values
// times 2^VRD_F16_ACT_EXP -- then the scores come out times 2^(2 * VRD_F16_ACT_EXP), which the host folds into the softmax
// scale, the probabilities are split as P * 2^VRD_F16_ACT_EXP as well (P <= 1, or <= 2^8 with the deferred rescaling of the
// second kernel: far inside the f16 range), and the powers of two cancel in O / l up to one exact factor at the end.
//
// One workgroup = NW waves = NW*32 query rows of one (b, head); KV tiles of 32 keys.
//   S^T = K . Q^T   A = K fragments from LDS (ds_read_b128), B = Q^T fragments held in registers for the whole
//                   kernel.  The result puts each query on a lane and 16 of its 32 keys in that lane's registers:
//                   softmax statistics are register-local plus one exchange with lane^32.
//   O^T += V^T . P^T  B = the probability registers themselves (registers 8s..8s+7 are the k-slots of k16 step s,
//                   in the instruction's permuted order key = 16s + 8(j>>2) + 4*half + (j&3)), split into hi/lo on
//                   the fly; A = V^T fragments fetched with ds_read_b64_tr_b16 (hardware transpose of a 4-key x
//                   16-column block), two per fragment, in exactly that key order.
// K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a 2-stage ring: tile t+1 is in flight while tile t
// is consumed; one barrier per tile.  LDS rows are unpadded (the DMA writes linearly), so 16-byte chunks are
// XOR-swizzled on the source address and on every read: K by (key & 15) for the row reads, V by (key & 3) << 2
// for the transposed reads; both patterns are conflict free for head_dim 128.
#include "vrd_common.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ uint4 g_attn_zero[16];            // 256 zero bytes: source of keys past Tk

template <int HD>
struct AG {
    static constexpr int ROWB = HD * 2;                 // bytes per key row of one plane
    static constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (16 or 8)
    static constexpr int RPI = 1024 / ROWB;             // key rows per wave DMA instruction (4 or 8)
    static constexpr int PLANE = 32 * ROWB;             // 8 KiB (hd 128)
    static constexpr int STAGE = 4 * PLANE;             // k_hi | k_lo | v_hi | v_lo
    static constexpr int N_DMA = 4 * 32 / RPI;          // wave instructions per tile (32 or 16)
    __device__ static constexpr int kswz(int key) { return key % CPR; }
    __device__ static constexpr int vswz(int key) { return ((key & 3) << 2) % CPR ^ (CPR == 8 ? ((key >> 1) & 1) << 2 : 0); }
};

template <int HD, int NW, bool F16>
__global__ __launch_bounds__(NW * 64, 2) void attn_flash_x3_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ k, const float* __restrict__ v,
                                                                int64_t ldkv, const uint8_t* __restrict__ kv_mask,
                                                                const uint8_t* __restrict__ q_mask, int Tq,
                                                                int Tk, int width, float scale, float* __restrict__ out,
                                                                int64_t ldo, int pair_out, int q_blocks, int n_head_) {
    using G = AG<HD>;
    typedef typename vrd::SplitFmt<F16>::x8 bf16x8;      // eight 16-bit elements of this instantiation's format (bf16 or f16)
    constexpr int KS = HD / 16;                   // k16 steps of the S^T contraction
    constexpr int DT = HD / 32;                   // 32-row d tiles of O^T
    constexpr int PER_WAVE = (G::N_DMA + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const kbias = reinterpret_cast<float*>(lds + 2 * G::STAGE);      // [32 * nkt]: 0 or -inf per key

    // 1-D grid with the XCD-aware renumbering of the GEMM kernels: the workgroups that share one (b, h)'s K and V
    // (consecutive logical ids) get the same XCD label, i.e. the same L2, and are dispatched close together.  With a
    // (q-block, h, b) grid they went round-robin to different XCDs and K/V were fetched from HBM once per q-block
    // (4.2 GB fetched per launch against 1.8 GB of q, k, v).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    const int qblk = lid % q_blocks, h = (lid / q_blocks) % n_head_, b = lid / (q_blocks * n_head_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = (qblk * NW + wave) * 32;
    // pair rows are blocks of [32 hi | 32 lo] bf16 (vrd_common.h), so a head's HD channels are HD*4 contiguous
    // bytes; 16-byte chunk lc of the head's logical hi (lo) plane is at block lc/4, +64 bytes for lo
    const char* kb = reinterpret_cast<const char*>(k + (int64_t)b * Tk * ldkv) + h * HD * 4;
    const char* vb = reinterpret_cast<const char*>(v + (int64_t)b * Tk * ldkv) + h * HD * 4;
    const char* zero_src = reinterpret_cast<const char*>(g_attn_zero);

    // DMA of tile kt: instruction j covers plane j / (32/RPI) rows (j % (32/RPI))*RPI ...; planes k_hi, k_lo, v_hi, v_lo
    const int rin = lane / G::CPR, pch = lane % G::CPR;
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int j = wave + NW * i;
            if (j < G::N_DMA) {
                const int plane = j / (32 / G::RPI), rb = j % (32 / G::RPI);
                const int row = rb * G::RPI + rin;                 // key inside the tile
                const int key = kt * 32 + row;
                const int sw = plane < 2 ? G::kswz(row) : G::vswz(row);
                const int lc = pch ^ sw;                           // logical 16-byte chunk of the plane row
                const char* base = (plane < 2 ? kb : vb) + (int64_t)key * ldkv * 4 + ((plane & 1) ? 64 : 0);
                // branch-free select (as a ternary on pointers hipcc emits an exec-masked branch per DMA)
                const uintptr_t pa = reinterpret_cast<uintptr_t>(base + (lc >> 2) * 128 + (lc & 3) * 16);
                const uintptr_t pz = reinterpret_cast<uintptr_t>(zero_src + lc * 16);
                const char* src = reinterpret_cast<const char*>(pz ^ ((pa ^ pz) & (uintptr_t)0 - (uintptr_t)(key < Tk)));
                // as an asm statement: the compiler put an s_waitcnt vmcnt(0) in front of the tile's V^T reads for the builtin (it
                // cannot tell the stage being filled from the one being read), i.e. the NEXT tile's bytes were waited for in the
                // middle of the current tile.  The loop's own wait at its top is the only one needed.
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)(lds + buf * G::STAGE + plane * G::PLANE + rb * 1024)));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
            }
        }
    };
    // Everything the workgroup reads before its first tile goes out together -- the first key tile (requested before its mask
    // is known: a tile without a valid key adds exact zeros to every sum and leaves the running maxima alone, so visiting it
    // changes no bit), the query-mask byte, the rows of Q, the key-mask bytes -- one round trip where the mask bytes, Q and the
    // first tile used to be three in series (a third of a three-tile sequence's time).
    issue(0, 0);
    const int tq = q0 + li;
    const bool q_row_live = tq < Tq && (!q_mask || q_mask[(int64_t)b * Tq + (tq < Tq ? tq : Tq - 1)] != 0);
    // Q^T fragments: lane (query li, half lh) holds d = 16s + 8*lh + 0..7 of its query, hi and lo
    bf16x8 qh[KS], ql[KS];
    {
        const char* qr = reinterpret_cast<const char*>(q + ((int64_t)b * Tq + (tq < Tq ? tq : Tq - 1)) * ldq) + h * HD * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int off = vrd::pair_index(16 * s + 8 * lh) * 2;
            qh[s] = *reinterpret_cast<const bf16x8*>(qr + off);
            ql[s] = *reinterpret_cast<const bf16x8*>(qr + off + 64);
        }
    }
    const int nkt = (Tk + 31) / 32;
    // key bias of the whole row of tiles, once: an ordinary global load inside the loop would make the compiler
    // drain the LDS-DMA queue (vmcnt(0)) at its first use, every tile
    // ... and, per tile of 32 keys, whether any key is valid: a tile of masked keys only adds exp(-inf) = 0 to every
    // sum and leaves the running maxima alone, so it is not loaded or multiplied at all (padding behind the pair's
    // frames: 32 of 288 keys at the benchmark shape, most of a max_seq_len batch of short pairs)
    int* const tile_on = reinterpret_cast<int*>(kbias + nkt * 32);
    for (int key = tid; key < nkt * 32; key += NW * 64) {
        const bool ok = key < Tk && (!kv_mask || kv_mask[(int64_t)b * Tk + key]);
        kbias[key] = ok ? 0.f : -INFINITY;
        const unsigned long long bal = __ballot(ok);
        if ((lane & 31) == 0) tile_on[key >> 5] = ((bal >> (lane & 32)) & 0xffffffffull) != 0ull;
    }
    // a wave whose 32 queries are all padding (q_mask) takes part in the staging and the barriers only; its rows are
    // written as zeros (the caller masks them)
    const bool q_live = __any(q_row_live);
    // a workgroup without a single live query (the padded tail of the sequence, or tiles past Tq) does not stream K / V
    // (the barrier also publishes the key-bias row and the tile flags)
    if (!__syncthreads_or(q_live)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tile requested above: not into LDS that is no longer ours
        if (tq < Tq) {
            float* orow = out + ((int64_t)b * Tq + tq) * ldo + h * HD;
            for (int c = lh * 4; c < HD; c += 8) *reinterpret_cast<float4*>(orow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    // the fragments are in their registers before the tile loop: the compiler does not see the asm requests, and its own
    // counted waits for these loads, placed at their first use inside the loop, would wait for every tile's successor there
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(qh[s]), "+v"(ql[s]));

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    // per-lane LDS offsets: K fragment row = key li; V^T transposed reads: lane 4q+p of a 16-lane group addresses
    // row r0 + q, columns c0 + 4p .. 4p+3, where c0 = 32*dt + 16*((lane >> 4) & 1) and r0 = 16s + 8*part + 4*lh
    const int krow = li * G::ROWB;
    const int vq = (lane >> 2) & 3, vp = lane & 3;
    const int vcol0 = 16 * ((lane >> 4) & 1) + 4 * vp;          // column inside a 32-wide d tile

    unsigned long long act = ~0ull;                      // rows of more than 64 tiles: every tile is visited
    if (nkt <= 64) act = __ballot(lane < nkt && tile_on[lane < nkt ? lane : 0] != 0);
    auto next_on = [&](int from) {                       // first tile >= from that has a valid key, or nkt
        if (from >= nkt) return nkt;
        if (nkt > 64) return from;
        const unsigned long long m = act & (~0ull << from);
        return m ? (int)__builtin_ctzll(m) : nkt;
    };
    int kt = 0;                                          // tile 0 is on its way (requested above) and is visited whatever its mask
    for (int it = 0; kt < nkt; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                           // ... and everybody else's; the previous tile is fully consumed
        const int kt_next = next_on(kt + 1);
        if (kt_next < nkt) issue(kt_next, (it + 1) & 1);
        const char* st = lds + (it & 1) * G::STAGE;
        const float* kbs = kbias + kt * 32;
        if (!q_live) {
            kt = kt_next;
            continue;
        }

        // ---- S^T = K . Q^T (three products per k16 step)
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        // the K fragments of step s+1 are requested before the three MFMAs of step s (one chain of dependent
        // MFMAs: nothing else would hide the LDS latency)
        bf16x8 kh = *reinterpret_cast<const bf16x8*>(st + krow + ((lh ^ G::kswz(li)) * 16));
        bf16x8 kl = *reinterpret_cast<const bf16x8*>(st + G::PLANE + krow + ((lh ^ G::kswz(li)) * 16));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 nh = kh, nl = kl;
            if (s + 1 < KS) {
                const int off = krow + (((2 * (s + 1) + lh) ^ G::kswz(li)) * 16);
                nh = *reinterpret_cast<const bf16x8*>(st + off);
                nl = *reinterpret_cast<const bf16x8*>(st + G::PLANE + off);
            }
            sacc = vrd::mfma32(kl, qh[s], sacc);
            sacc = vrd::mfma32(kh, ql[s], sacc);
            sacc = vrd::mfma32(kh, qh[s], sacc);
            kh = nh;
            kl = nl;
        }

        // ---- online softmax for query column li; this lane holds keys (e&3) + 8*(e>>2) + 4*lh
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = sacc[e] * scale + kbs[(e & 3) + 8 * (e >> 2) + 4 * lh];
            mx = fmaxf(mx, sacc[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = __expf(sacc[e] - m_use);
            psum += sacc[e];
        }
        l_part = l_part * alpha + psum;
        m_run = m_new;
        // rescale the running output only if some query's maximum moved (the kernel is instruction-bound: ~550
        // instructions per tile and wave, 16 * DT of them this rescale; after the first tiles the maxima rarely move)
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = sacc[8 * s + j];
            bf16x8 ph, pl;
            vrd::split_n<F16>(pf, ph, pl);          // (f16: P * 2^VRD_F16_ACT_EXP)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x8 vh, vl;
                s16x8 rh, rl;
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int row = 16 * s + 8 * part + 4 * lh + vq;                  // key row this lane addresses
                    const int col = 32 * d + vcol0;                                   // first of its 4 columns
                    const int off = row * G::ROWB + ((((col * 2) >> 4) ^ G::vswz(row)) * 16) + ((col * 2) & 15);
                    const s16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 2 * G::PLANE + off));
                    const s16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 3 * G::PLANE + off));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        rh[4 * part + j] = th[j];
                        rl[4 * part + j] = tl[j];
                    }
                }
                vh = __builtin_bit_cast(bf16x8, rh);
                vl = __builtin_bit_cast(bf16x8, rl);
                oacc[d] = vrd::mfma32(vl, ph, oacc[d]);
                oacc[d] = vrd::mfma32(vh, pl, oacc[d]);
                oacc[d] = vrd::mfma32(vh, ph, oacc[d]);
            }
        }
        kt = kt_next;
    }

    const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
    // f16: the accumulators hold sum (P 2^e)(V 2^e); l is the sum of the unscaled P
    const float inv = q_live ? (F16 ? vrd::F16_ACT_INV * vrd::F16_ACT_INV : 1.0f) / l_tot : 0.f;
    if (tq < Tq) {
        float* orow = out + ((int64_t)b * Tq + tq) * ldo;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = h * HD + 32 * d + 8 * g + 4 * lh;           // registers 4g..4g+3 are d = 32d + 8g + 4lh + 0..3
                const float4 val = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv,
                                               oacc[d][4 * g + 3] * inv);
                if (pair_out) vrd::store_pair4(orow, c, width, val, vrd::SplitFmt<F16>::fmt);
                else *reinterpret_cast<float4*>(orow + c) = val;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: one wave per SIMD, 64 queries per wave (attn_flash_x3_w64_kernel).
//
// The kernel above is latency-bound: at 247 registers and two waves per SIMD nothing can be read ahead, a wave's tile is a
// serial chain S^T (24 dependent MFMAs) -> softmax (~190 VALU) -> O^T (24 MFMAs behind 32 transposed reads), and each of
// the two workgroups of a CU streams its own copy of K / V (profiles/r02_sq_counters.json: MFMA pipe busy 0.29).
// Here a workgroup is 4 waves = 256 queries of one (b, h) -- at the benchmark shape all valid queries, so K / V are
// streamed ONCE -- and a wave owns the whole register file of its SIMD (launch bounds 256 x 1 -> 512 registers):
//   * two 32-query blocks per wave: every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) feeds six MFMAs
//     instead of three, and the two blocks' accumulation chains are independent;
//   * software pipeline across key tiles: the S^T MFMAs of tile t+1 are issued between the softmax instructions of tile t
//     (a wave issues in order: vector instructions only overlap ITS OWN MFMAs if they sit between them in program order,
//     ~19 cycles of them per 32-cycle MFMA: scripts/lab/mfma_gap.hip), then the O^T MFMAs of tile t between the second half
//     of the softmax and the next requests; every softmax piece is one volatile asm statement, so it stays in its gap;
//   * register file by hand: the Q^T fragments (a[0:127] at head_dim 128) and the O^T accumulators (a[128:255]) live in the
//     accumulator half (an MFMA takes A / B / C / D from either half), scores, probabilities and K / V fragments in the
//     architectural half.  The compiler cannot be talked into that split -- with "a" operand constraints it keeps the values
//     in the architectural half and copies them over before every MFMA (+680 v_accvgpr_write, 366 spills) -- so the
//     accumulator half is addressed by literal register names inside the asm statements and the compiler owns none of it
//     (checked after every edit: no v_accvgpr_* outside ASMSTART / ASMEND, no scratch);
//   * a 4-stage LDS ring (32 KiB per 32-key tile at head_dim 128): two tiles in flight behind the two being read, one
//     barrier per tile, counted vmcnt waits, 8 LDS-DMA instructions per wave and tile (half the other kernel's), each a
//     scalar base + a per-lane offset that never changes (no vector address arithmetic in the loop).
// Arithmetic per element is the other kernel's (same products, same accumulation order per accumulator), except that the
// exponentials are taken in base 2 with log2(e) folded into the score scale (one FMA per score instead of FMA + multiply).
//
// What the compiler does not do for the asm MFMAs (cdna_hip_programming.md 5.7): wait states between a vector instruction
// that writes an operand and the MFMA reading it (the first MFMA behind a split carries an s_nop 1), and between an MFMA
// and a vector instruction reading its result (scores are read one pipeline stage later; the accumulators behind an
// explicit s_nop block).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#define VRD_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define VRD_SB() __builtin_amdgcn_sched_barrier(0)

template <int I>
using ic = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(ic<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The value of lane l ^ 32 combined with this lane's.  v_permlane32_swap exchanges the upper 32 lanes of its first register with
// the lower 32 of its second; as inline assembly, because hipcc (ROCm 7.2) returns the FIRST result for both elements of
// __builtin_amdgcn_permlane32_swap's result pair (measured: the row maximum / sum came out as that of one half-row).
// The s_nop 1 in front: a vector write of the register needs two wait states before a permlane reads it.
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xchg32_sum(float v) {
    float a = v, b = v;
    swap32(a, b);
    return a + b;
}
// (p0, p1) -> packed hi pair and lo pair (lo = round(p - hi)) in bf16, or (F16) in f16 -- of the values as they are: the
// caller has the f16 format's scale in them already
template <bool F16>
__device__ __forceinline__ void split_pair(float p0, float p1, unsigned& hi, unsigned& lo) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        const f16x2 h = {(_Float16)p0, (_Float16)p1};
        hi = __builtin_bit_cast(unsigned, h);
        const f16x2 l = {(_Float16)(p0 - (float)h[0]), (_Float16)(p1 - (float)h[1])};
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        const bf16x2 h = {(__bf16)p0, (__bf16)p1};
        hi = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
        const bf16x2 l = {(__bf16)(p0 - h0), (__bf16)(p1 - h1)};
        lo = __builtin_bit_cast(unsigned, l);
    }
}

// one MFMA of the element format as an asm statement: VRD_MFMA(prefix, "operands", constraints...)
#define VRD_MFMA(pre, ops, ...)                                                   \
    do {                                                                          \
        if constexpr (F16) asm volatile(pre "v_mfma_f32_32x32x16_f16 " ops __VA_ARGS__);  \
        else asm volatile(pre "v_mfma_f32_32x32x16_bf16 " ops __VA_ARGS__);       \
    } while (0)

// a[R0 .. R0+N-1] *= alpha
template <int R0, int N>
__device__ __forceinline__ void agpr_scale(float alpha) {
    if constexpr (N > 0) {
        float t;
        asm volatile("v_accvgpr_read_b32 %0, a%c2\n\tv_mul_f32 %0, %0, %1\n\tv_accvgpr_write_b32 a%c2, %0"
                     : "=&v"(t) : "v"(alpha), "n"(R0));
        agpr_scale<R0 + 1, N - 1>(alpha);
    }
}

// softmax state of one 32-query block while a tile is in the pipeline
struct SmBlock {
    float x[16];                // scaled scores, then probabilities
    float mx, mu, ae, alpha, sum;
    float mr, lp;               // reference point of the exponentials so far (log2 units), running sum
    u32x4 ph[2], pl[2];         // P^T fragments: k16 step s -> hi / lo (bf16x8 as four dwords)
};

// lab builds (-DVRD_ATTN_STAMP): s_memtime stamps of workgroup 0's items, summed per stamp index (scripts/dev/flash_stamps.py)
#ifdef VRD_ATTN_STAMP
__device__ unsigned long long g_attn_stamp[65];
#define VRD_STAMP(i)                                                                      \
    do {                                                                                  \
        if (wave == 0) {                                                                  \
            unsigned long long t_;                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            if (lane == 0) stamp_lds[i] = t_;                                             \
        }                                                                                 \
    } while (0)
#else
#define VRD_STAMP(i) do { } while (0)
#endif

template <int HD, bool F16>
__global__ __launch_bounds__(256, 1) void attn_flash_x3_w64_kernel(const float* __restrict__ q, int64_t ldq,
                                                                  const float* __restrict__ k, const float* __restrict__ v,
                                                                  int64_t ldkv, const uint8_t* __restrict__ kv_mask,
                                                                  const uint8_t* __restrict__ q_mask, int Tq, int Tk, int width,
                                                                  float scale_log2e, float* __restrict__ out, int64_t ldo,
                                                                  int pair_out, int q_blocks, int n_head_, int n_batch, int prefetch, int item_step) {
    using G = AG<HD>;
    constexpr int KS = HD / 16;                   // k16 steps of the S^T contraction
    constexpr int DT = HD / 32;                   // 32-row d tiles of O^T
    constexpr int NS = 4;                         // ring stages
    constexpr int PER_WAVE = G::N_DMA / 4;        // LDS-DMA instructions per wave and tile (8 or 4)
    constexpr int PPP = 32 / G::RPI;              // pieces per plane (8 or 4)
    constexpr int NG_S = 6 * KS;                  // MFMAs (= gaps) of the S^T phase
    constexpr int NG_O = 12 * DT;                 // MFMAs of the O^T phase
    constexpr int HPG_S = 48 / NG_S, HPG_O = 48 / NG_O;       // softmax half-pieces per gap (1 or 2)
    constexpr int LEAD = 4;                       // pieces of the S^T phase that run in front of its first MFMA
    // accumulator-half map: Q^T fragment (block qb, k16 step s): hi a[AQ .. AQ+3], lo a[AQ+4 .. AQ+7]; O^T accumulator
    // (block qb, d tile d): a[AO .. AO+15]
#define VRD_AQ(qb, s) (8 * ((qb) * KS + (s)))
#define VRD_AO(qb, d) (128 + 16 * ((qb) * DT + (d)))
    static_assert(G::N_DMA % 4 == 0 && NG_S >= 24 && NG_O >= 24 && VRD_AQ(1, KS - 1) + 8 <= 128, "layout assumptions");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    // behind the ring: key bias of the item's row of tiles, per tile whether none / some / all of its keys are valid, per group
    // of 32 queries whether any is live; then (prefetch mode) the NEXT item's mask bytes as LDS-DMA leaves them, a dword each
    const int nkt = (Tk + 31) / 32, nqg = (Tq + 31) / 32;
    float* const kbias = reinterpret_cast<float*>(lds + NS * G::STAGE);      // [32 * nkt]: 0 or -inf per key
    int* const tile_on = reinterpret_cast<int*>(kbias + nkt * 32);           // [nkt]
    int* const qgrp = tile_on + nkt;                                         // [nqg]
    const int nk64 = (nkt * 32 + 63) & ~63, nq64 = (nqg * 32 + 63) & ~63;
    int* const raw_k = qgrp + nqg;                                           // [nk64]  (prefetch mode only)
    int* const raw_q = raw_k + nk64;                                         // [nq64]
#ifdef VRD_ATTN_STAMP
    unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + 163840 - 64 * 8);
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr_t)lds);
    // Persistent workgroups (one per CU: the grid is min(items, CUs)) walk the (b, h) items blockIdx.x, + gridDim.x, ...: a
    // workgroup of this size has its CU to itself, so every dispatch is a drained CU (measured: ~9 us per workgroup between
    // the end of one and the first instruction of the next).
    //
    // The masks of an item (Tk + Tq bytes) decide which tiles are requested, so they sit in front of everything else: read
    // with ordinary loads at the start of the item they were two round trips to HBM in series with the first tile's (mask
    // bytes -> barrier -> requests).  With `prefetch` (the host sets it when the staging area fits in LDS) the NEXT item's
    // bytes are requested by LDS-DMA (global_load_lds_ubyte: a dword per byte, no registers held) before the item's first
    // tile, land under its tile loop -- the vector-memory counter is in order: every counted wait for a tile covers them --
    // and are turned into the three tables between two barriers at the start of their item.
    const int n_items = n_head_ * n_batch;
    bool masks_pending = true;                    // mask bytes requested that no wait of this wave has covered yet
    // (per-lane values of the mask staging are formed from a lane id read afresh: as loop invariants the compiler computed
    // them once, spilled them -- every register is spoken for in the tile loop -- and reloaded them at the start of each
    // item behind an s_waitcnt vmcnt(0), i.e. behind the previous item's output stores)
    auto fresh_lane = [&]() {
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        return ln;
    };
    auto dma_masks = [&](int it2) {
        const int b2 = it2 / n_head_;
        const int ln = fresh_lane();
        if (kv_mask) {
            const uint8_t* base = kv_mask + (int64_t)b2 * Tk;
            for (int j = wave; j < nk64 / 64; j += 4) {
                const int key = j * 64 + ln;
                const unsigned voff = (unsigned)(key < Tk ? key : Tk - 1);
                const unsigned dst = lds_base + (unsigned)(reinterpret_cast<char*>(raw_k) - lds) + j * 256;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2" ::"s"(dst), "v"(voff), "s"(base) : "memory");
            }
        }
        if (q_mask) {
            const uint8_t* base = q_mask + (int64_t)b2 * Tq;
            for (int j = wave; j < nq64 / 64; j += 4) {
                const int r = j * 64 + ln;
                const unsigned voff = (unsigned)(r < Tq ? r : Tq - 1);
                const unsigned dst = lds_base + (unsigned)(reinterpret_cast<char*>(raw_q) - lds) + j * 256;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2" ::"s"(dst), "v"(voff), "s"(base) : "memory");
            }
        }
    };
    // item walk: strided (item_step = grid) or, item_step = 1, a contiguous run of items per workgroup
    const int item_first = item_step == 1 ? (int)(((long long)n_items * blockIdx.x) / gridDim.x) : (int)blockIdx.x;
    const int item_end = item_step == 1 ? (int)(((long long)n_items * (blockIdx.x + 1)) / gridDim.x) : n_items;
    for (int item = item_first; item < item_end; item += item_step) {
    VRD_STAMP(0);
    const int h = item % n_head_, b = item / n_head_;
    if (prefetch && item == item_first) dma_masks(item);
    // The ring, the tables and the output slabs of the previous item are free; (prefetch) this item's mask bytes have landed:
    // they are older than the rows of Q every live block waits for, so only an item without one has to wait here.  The
    // barriers are for LDS only (not __syncthreads, which would also drain the previous item's output stores: 3-4 k cycles).
    if (prefetch && masks_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const int ln = fresh_lane(), tn = wave * 64 + ln;
        for (int key = tn; key < nkt * 32; key += 256) {
            const bool ok = key < Tk && (!kv_mask || (prefetch ? raw_k[key] != 0 : kv_mask[(int64_t)b * Tk + (key < Tk ? key : Tk - 1)] != 0));
            kbias[key] = ok ? 0.f : -INFINITY;
            const unsigned long long bal = __ballot(ok);
            const unsigned bits = (unsigned)((bal >> (ln & 32)) & 0xffffffffull);
            if ((ln & 31) == 0) tile_on[key >> 5] = bits == 0u ? 0 : (bits == 0xffffffffu ? 2 : 1);      // keys of the tile: none / some / all valid
        }
        for (int r = tn; r < nqg * 32; r += 256) {
            const bool ok = r < Tq && (!q_mask || (prefetch ? raw_q[r] != 0 : q_mask[(int64_t)b * Tq + (r < Tq ? r : Tq - 1)] != 0));
            const unsigned long long bal = __ballot(ok);
            if ((ln & 31) == 0) qgrp[r >> 5] = ((bal >> (ln & 32)) & 0xffffffffull) != 0ull;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (prefetch && item + item_step < item_end) {
        dma_masks(item + item_step);
        masks_pending = true;
    }
    VRD_STAMP(1);
    // One workgroup per (b, h) walks the 256-query blocks of the sequence.  (A grid with one workgroup per block made the
    // blocks of pure padding -- rows 256 .. 287 at the benchmark shape -- cost 22 us each: a workgroup of this size has a CU
    // to itself, so even one that leaves at once waits for the CU to drain.  Measured: 1.92 ms per launch against 1.2.)
    for (int qblk = 0; qblk < q_blocks; ++qblk) {
    if (qblk) {                                   // the ring and the output slabs of the previous block are no longer in use
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    const int q0 = qblk * 256 + wave * 64;        // the wave's two blocks: q0 .. q0+31, q0+32 .. q0+63
    const char* kb = reinterpret_cast<const char*>(k + (int64_t)b * Tk * ldkv) + h * HD * 4;
    const char* vb = reinterpret_cast<const char*>(v + (int64_t)b * Tk * ldkv) + h * HD * 4;

    // which of the block's eight groups of 32 queries have a live one: one LDS read, the same answer in every wave
    const int gq = qblk * 8 + (lane & 7);
    const unsigned blk_bits = (unsigned)(__ballot(gq < nqg && qgrp[gq < nqg ? gq : 0] != 0) & 0xffull);
    const bool live0 = (blk_bits >> (2 * wave)) & 1u, live1 = (blk_bits >> (2 * wave + 1)) & 1u;
    const bool q_live = live0 || live1;
    if (blk_bits == 0u) {                         // no live query in these 256 rows: zeros, K / V are not streamed
#ifndef VRD_ATTN_LAB_NOZERO
        {   // whole rows of the head's channels, 16 bytes per lane; the block's rows go round the four waves
            constexpr int LPRZ = HD / 4, RPIZ = 64 / LPRZ;
            const int row_end = Tq < qblk * 256 + 256 ? Tq : qblk * 256 + 256;
            const int ln = fresh_lane();
            int row = qblk * 256 + wave * RPIZ + ln / LPRZ;
            char* zp = reinterpret_cast<char*>(out + ((int64_t)b * Tq + row) * ldo) + h * HD * 4 + (ln % LPRZ) * 16;
            const int64_t zstep = ldo * 4 * (4 * RPIZ);
            const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll 1
            for (; row < row_end; row += 4 * RPIZ, zp += zstep) *reinterpret_cast<u32x4*>(zp) = zero4;
        }
#endif
        continue;
    }
    VRD_STAMP(55);
    // every accumulator-half register is named in a clobber list once: that is what makes the kernel descriptor allocate them
    asm volatile("" ::: VRD_ALL_AGPRS);
    // Q^T fragments of both blocks: lane (query li, half lh) holds d = 16s + 8*lh + 0..7 of its query, hi and lo -- the shape of
    // a K fragment.  Read straight into registers that is 32 loads per lane that touch 32 rows each, 32 bytes of a row: the
    // address unit took ~12 k cycles over them, a sixth of the item (stamps; nothing else of the workgroup can run meanwhile).
    // So the wave's 64 rows come by LDS-DMA, whole rows per instruction, into the ring, laid out as K tiles are (planes q_hi, q_lo
    // of 32 rows; the K swizzle) and read back as K fragments are: the first 32 rows of waves 0, 1 / 2, 3 into the two halves
    // of stage 0 / 1, the other 32 rows into stages 2 / 3 -- once every wave has moved its first 32 rows, stages 0 and 1 take
    // the first two tiles, which then travel while the other rows move (the ring is free from the barrier at the start of the
    // item, or of the block, to the barriers below).
    {
        const char* qbase = reinterpret_cast<const char*>(q + (int64_t)b * Tq * ldq) + h * HD * 4;
        const int rin_q = lane / G::CPR, pch_q = lane % G::CPR;
        const unsigned qrowbytes = (unsigned)(ldq * 4);                 // (host-checked: a (b) slab of Q is below 2 GiB)
        static_for<2 * PPP>([&](auto u_c) {
            constexpr int qb = decltype(u_c)::value / PPP, rb = decltype(u_c)::value % PPP;
            const int row = rb * G::RPI + rin_q;
            const int tq = q0 + 32 * qb + row;
            const int lc = pch_q ^ G::kswz(row);
            const unsigned voff = (unsigned)(tq < Tq ? tq : Tq - 1) * qrowbytes + (unsigned)((lc >> 2) * 128 + (lc & 3) * 16);
            const unsigned dst = lds_base + (2 * qb + (wave >> 1)) * G::STAGE + (wave & 1) * 2 * G::PLANE + rb * 1024;
            const char* const src_hi = qbase + 0;          // (named inside the lambda: an asm operand alone does not capture)
            const char* const src_lo = src_hi + 64;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(src_hi) : "memory");
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + G::PLANE), "v"(voff), "s"(src_lo) : "memory");
        });
    }
    VRD_STAMP(56);
    // O^T accumulators = 0
    static_for<2 * DT * 16>([&](auto r_c) {
        asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "n"(128 + decltype(r_c)::value));
    });
    VRD_STAMP(57);

    // ---- LDS-DMA.  Piece i (0 .. PER_WAVE-1) of a tile: plane (wave + 4i) / PPP (k_hi, k_lo, v_hi, v_lo), row block
    // rb = (wave + 4i) % PPP, key row rb * RPI + rin.  Row blocks differ by multiples of 4 rows (head_dim 128: by 16), which
    // leaves both swizzles unchanged, so a lane has ONE source offset for the K planes and one for the V planes; plane,
    // row block and tile are added to the scalar base.
    const int rin = lane / G::CPR, pch = lane % G::CPR;
    const int row0 = (wave % PPP) * G::RPI + rin;                      // key row of this lane in piece 0 (and in every even piece)
    const int rowbytes = (int)(ldkv * 4);                              // (host-checked: a (b) slab of K / V is below 2 GiB)
    auto chunk_of = [&](int row, bool is_v) {
        const int lc = pch ^ (is_v ? G::vswz(row) : G::kswz(row));     // logical 16-byte chunk of the plane row
        return (lc >> 2) * 128 + (lc & 3) * 16;
    };
    constexpr int HALF_ROWS = HD == 128 ? 16 : 0;                      // rows between an even piece and the odd piece behind it
    // what the pieces of one tile share: scalar bases of the K and V slabs at the tile, per-lane offsets for even / odd pieces
    struct Req {
        const char *kbase, *vbase;
        unsigned vk[2], vv[2];
        int half;                                                      // bytes from an even piece's rows to the odd piece's
    };
    auto req_of = [&](int kt) {
        Req r;
        if (kt * 32 + 32 <= Tk) {
            r.kbase = kb + (int64_t)kt * 32 * rowbytes;
            r.vbase = vb + (int64_t)kt * 32 * rowbytes;
            r.vk[0] = r.vk[1] = (unsigned)(row0 * rowbytes) + chunk_of(row0, false);
            r.vv[0] = r.vv[1] = (unsigned)(row0 * rowbytes) + chunk_of(row0, true);
            r.half = HALF_ROWS * rowbytes;
        } else {        // the last, partial tile: keys past Tk re-read key Tk-1 (finite values; their scores get -inf, their P is 0)
            r.kbase = kb;
            r.vbase = vb;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int key = kt * 32 + row0 + HALF_ROWS * u;
                key = key < Tk ? key : Tk - 1;
                r.vk[u] = (unsigned)(key * rowbytes) + chunk_of(row0, false);
                r.vv[u] = (unsigned)(key * rowbytes) + chunk_of(row0, true);
            }
            r.half = 0;
        }
        return r;
    };
    auto issue1 = [&](const Req& r, int buf, auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        constexpr int u = HD == 128 ? (i & 1) : 0;
        const int j = wave + 4 * i;
        const int plane = j / PPP, rb = j % PPP;                        // plane is the same for every wave: i / 2 or i
        const unsigned dst = lds_base + buf * G::STAGE + plane * G::PLANE + rb * 1024;
        const char* sbase = (plane < 2 ? r.kbase : r.vbase) + ((plane & 1) ? 64 : 0) + (u ? r.half : 0);
        const unsigned voff = plane < 2 ? r.vk[u] : r.vv[u];
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory");
    };

    unsigned long long act = ~0ull;                      // rows of more than 64 tiles: every tile is visited
    if (nkt <= 64) act = __ballot(lane < nkt && tile_on[lane < nkt ? lane : 0] != 0);
    auto next_on = [&](int from) {                       // first tile >= from that has a valid key, or nkt
        if (from >= nkt) return nkt;
        if (nkt > 64) return from;
        const unsigned long long m = act & (~0ull << from);
        return m ? (int)__builtin_ctzll(m) : nkt;
    };
    const int n_act = nkt <= 64 ? (int)__builtin_popcountll(act) : nkt;      // tiles that are visited
    unsigned long long cln = 0ull;                       // tiles whose 32 keys are all valid (no key bias needed)
    if (nkt <= 64) cln = __ballot(lane < nkt && tile_on[lane < nkt ? lane : 0] == 2);
    const int krow = li * G::ROWB;
    const int vq = (lane >> 2) & 3, vp = lane & 3;
    const int vcol0 = 16 * ((lane >> 4) & 1) + 4 * vp;          // column inside a 32-wide d tile

    struct KF { bf16x8 h, l; };
    auto load_k = [&](const char* st, int s) __attribute__((always_inline)) {
        KF f;
        const int off = krow + (((2 * s + lh) ^ G::kswz(li)) * 16);
        f.h = *reinterpret_cast<const bf16x8*>(st + off);
        f.l = *reinterpret_cast<const bf16x8*>(st + G::PLANE + off);
        return f;
    };
    // V^T fragment (s, d) of the tile in stage `st`: two transposed reads per plane
    auto load_v = [&](const char* st, int s, int d) __attribute__((always_inline)) {
        KF f;
        s16x8 rh, rl;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int row = 16 * s + 8 * part + 4 * lh + vq;                  // key row this lane addresses
            const int col = 32 * d + vcol0;                                   // first of its 4 columns
            const int off = row * G::ROWB + ((((col * 2) >> 4) ^ G::vswz(row)) * 16) + ((col * 2) & 15);
            const s16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 2 * G::PLANE + off));
            const s16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 3 * G::PLANE + off));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rh[4 * part + j] = th[j];
                rl[4 * part + j] = tl[j];
            }
        }
        f.h = __builtin_bit_cast(bf16x8, rh);
        f.l = __builtin_bit_cast(bf16x8, rl);
        return f;
    };

    // ---- the softmax of one tile as 73 pieces, run in the gaps between the MFMAs (one per gap at head_dim 128, two at 64).
    // What fits in a gap was measured (scripts/lab/mfma_gap.hip, one wave per SIMD, cycles per MFMA + k instructions):
    //   independent v_fma_f32: 33 up to k = 3, 36 at 4, 38-47 at 5, then +5..8 each -- an MFMA costs its wave ~13 cycles of
    //   issue, which leaves ~19 per 32-cycle MFMA;  a DEPENDENT chain costs 8 per instruction beyond three;  v_exp_f32
    //   12 cycles, v_cvt_pk_bf16_f32 8, v_max3_f32 4;  v_pk_fma_f32 and v_dot2_f32_bf16 stall the MFMA pipe (51 / 54 cycles
    //   with ONE of them in the gap): no packed f32 arithmetic here;  a ds_read next to an MFMA ~6.
    // So a tile's softmax (~1.8 k cycles of vector issue for 32 scores per lane) fits under its 96 MFMAs only if every gap
    // carries <= ~19 cycles of independent instructions.  Each piece is ONE volatile asm statement (volatile statements keep
    // their order: as plain C++ the compiler moved the tails of pieces into lumps of 10-20 instructions and left other gaps
    // empty), and the two blocks A / B alternate inside a piece so that no instruction reads its predecessor's result.
    //   S^T phase (pieces 0..47):  0-2 row maxima of the RAW scores (v_max3);  3 exchange with the other half-row;  4-7
    //     new reference mu, alpha;  8-23 elements 0..7 of A, B: y = s * scale - mu (one FMA: the key bias of a tile with
    //     masked keys is added to its raw scores before the step), exp2, running sum, software-pipelined one element apart;
    //     24-39 hi / lo split of P^T's k16 step 0 (four pieces per pair);  40-47 elements 8..11.
    //   O^T phase (pieces 48..):   48-55 elements 12..15;  56-71 split of k16 step 1 (needed from the phase's second half);
    //     72 the running sums.
    // Hazards the compiler cannot see inside: a transcendental's result needs one wait state before a vector instruction
    // reads it (read a piece later at the earliest), a vector write two before a permlane reads it.
    SmBlock A, B;
    A.mr = B.mr = -INFINITY;
    A.lp = B.lp = 0.f;
    float tsp[4];                                // temporaries of the split in flight
    unsigned long long cmask[2] = {0ull, 0ull};
    // how far (log2 units) a row maximum may run ahead of the reference point of its exponentials before the running output
    // is rescaled: probabilities reach 2^thr.  f16: P * 2^VRD_F16_ACT_EXP is what gets split, so 2^(8 + 4) at most
    const float thr = F16 ? 8.0f : 16.0f;
    auto fea_piece = [&](SmBlock& X, const f32x16& sx, auto e_c) __attribute__((always_inline)) {
        constexpr int e = decltype(e_c)::value;  // exp of element e, FMA of element e+1, element e-1 joins the sum
        if constexpr (e == 0) {
            asm volatile("v_fma_f32 %0, %2, %3, -%4\n\tv_exp_f32 %1, %1"
                         : "=&v"(X.x[1]), "+v"(X.x[0]) : "v"(sx[1]), "s"(scale_log2e), "v"(X.mu));
        } else if constexpr (e == 1) {           // sum = x[0]: the first term
            asm volatile("v_fma_f32 %0, %3, %4, -%5\n\tv_exp_f32 %1, %1\n\tv_mov_b32 %2, %6"
                         : "=&v"(X.x[2]), "+v"(X.x[1]), "=&v"(X.sum) : "v"(sx[2]), "s"(scale_log2e), "v"(X.mu), "v"(X.x[0]));
        } else if constexpr (e < 15) {
            asm volatile("v_fma_f32 %0, %3, %4, -%5\n\tv_exp_f32 %1, %1\n\tv_add_f32 %2, %2, %6"
                         : "=&v"(X.x[e + 1]), "+v"(X.x[e]), "+v"(X.sum) : "v"(sx[e + 1]), "s"(scale_log2e), "v"(X.mu), "v"(X.x[e - 1]));
        } else {
            asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %2" : "+v"(X.x[15]), "+v"(X.sum) : "v"(X.x[14]));
        }
    };
    auto split_piece = [&](auto s_c, auto q_c) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value, q = decltype(q_c)::value, j = q >> 2, sub = q & 3, e0 = 8 * s + 2 * j;
        if constexpr (sub == 0) {
            if constexpr (F16)
                asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                             : "=&v"(A.ph[s][j]), "=&v"(B.ph[s][j]) : "v"(A.x[e0]), "v"(A.x[e0 + 1]), "v"(B.x[e0]), "v"(B.x[e0 + 1]));
            else
                asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5"
                             : "=&v"(A.ph[s][j]), "=&v"(B.ph[s][j]) : "v"(A.x[e0]), "v"(A.x[e0 + 1]), "v"(B.x[e0]), "v"(B.x[e0 + 1]));
        } else if constexpr (sub == 1) {        // the hi halves back as f32 (bf16: bit moves; f16: conversions, the upper half by SDWA select)
            if constexpr (F16)
                asm volatile("v_cvt_f32_f16 %0, %4\n\tv_cvt_f32_f16 %2, %5\n\t"
                             "v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                             "v_cvt_f32_f16_sdwa %3, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                             : "=&v"(tsp[0]), "=&v"(tsp[1]), "=&v"(tsp[2]), "=&v"(tsp[3]) : "v"(A.ph[s][j]), "v"(B.ph[s][j]));
            else
                asm volatile("v_lshlrev_b32 %0, 16, %4\n\tv_lshlrev_b32 %2, 16, %5\n\tv_and_b32 %1, 0xffff0000, %4\n\tv_and_b32 %3, 0xffff0000, %5"
                             : "=&v"(tsp[0]), "=&v"(tsp[1]), "=&v"(tsp[2]), "=&v"(tsp[3]) : "v"(A.ph[s][j]), "v"(B.ph[s][j]));
        } else if constexpr (sub == 2) {
            asm volatile("v_sub_f32 %0, %4, %0\n\tv_sub_f32 %2, %6, %2\n\tv_sub_f32 %1, %5, %1\n\tv_sub_f32 %3, %7, %3"
                         : "+v"(tsp[0]), "+v"(tsp[1]), "+v"(tsp[2]), "+v"(tsp[3])
                         : "v"(A.x[e0]), "v"(A.x[e0 + 1]), "v"(B.x[e0]), "v"(B.x[e0 + 1]));
        } else {
            if constexpr (F16)
                asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                             : "=&v"(A.pl[s][j]), "=&v"(B.pl[s][j]) : "v"(tsp[0]), "v"(tsp[1]), "v"(tsp[2]), "v"(tsp[3]));
            else
                asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5"
                             : "=&v"(A.pl[s][j]), "=&v"(B.pl[s][j]) : "v"(tsp[0]), "v"(tsp[1]), "v"(tsp[2]), "v"(tsp[3]));
        }
    };
    auto sm_piece = [&](auto i_c, const f32x16& sa, const f32x16& sb) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        if constexpr (i == 0) {
            asm volatile("v_max_f32 %0, %2, %3\n\tv_max_f32 %1, %4, %5\n\tv_max3_f32 %0, %6, %7, %0\n\tv_max3_f32 %1, %8, %9, %1\n\t"
                         "v_max3_f32 %0, %10, %11, %0\n\tv_max3_f32 %1, %12, %13, %1"
                         : "=&v"(A.mx), "=&v"(B.mx)
                         : "v"(sa[0]), "v"(sa[1]), "v"(sb[0]), "v"(sb[1]), "v"(sa[2]), "v"(sa[3]), "v"(sb[2]), "v"(sb[3]),
                           "v"(sa[4]), "v"(sa[5]), "v"(sb[4]), "v"(sb[5]));
        } else if constexpr (i == 1) {
            asm volatile("v_max3_f32 %0, %2, %3, %0\n\tv_max3_f32 %1, %4, %5, %1\n\tv_max3_f32 %0, %6, %7, %0\n\tv_max3_f32 %1, %8, %9, %1\n\t"
                         "v_max3_f32 %0, %10, %11, %0\n\tv_max3_f32 %1, %12, %13, %1"
                         : "+v"(A.mx), "+v"(B.mx)
                         : "v"(sa[6]), "v"(sa[7]), "v"(sb[6]), "v"(sb[7]), "v"(sa[8]), "v"(sa[9]), "v"(sb[8]), "v"(sb[9]),
                           "v"(sa[10]), "v"(sa[11]), "v"(sb[10]), "v"(sb[11]));
        } else if constexpr (i == 2) {
            asm volatile("v_max3_f32 %0, %2, %3, %0\n\tv_max3_f32 %1, %4, %5, %1\n\tv_max3_f32 %0, %6, %7, %0\n\tv_max3_f32 %1, %8, %9, %1"
                         : "+v"(A.mx), "+v"(B.mx)
                         : "v"(sa[12]), "v"(sa[13]), "v"(sb[12]), "v"(sb[13]), "v"(sa[14]), "v"(sa[15]), "v"(sb[14]), "v"(sb[15]));
        } else if constexpr (i == 3) {           // the other half-row's maximum
            float ta, tb;
            asm volatile("v_mov_b32 %2, %0\n\tv_mov_b32 %3, %1\n\ts_nop 0\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\t"
                         "v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %3"
                         : "+v"(A.mx), "+v"(B.mx), "=&v"(ta), "=&v"(tb));
        } else if constexpr (i == 4) {
            // The reference point of the exponentials moves only when the row maximum has grown by more than `thr`
            // (log2 units) since it was set: probabilities then reach 2^THR instead of 1, which costs nothing here (they are
            // split into hi + lo relative to their own magnitude and accumulated in f32), and the running output -- 64
            // accumulator-half registers per block, three instructions each -- is rescaled in the first tile or two only
            // instead of in nearly every tile (some query of the 32 almost always finds a slightly larger score).
            // mx (log2 units) = raw maximum * scale (scale > 0: the same value as the maximum of the scaled scores);
            // ae = mx - m_old for now
            // f16: the whole frame of reference sits VRD_F16_ACT_EXP (= 4) lower -- mx' = mx - 4, hence mu' and the stored
            // reference too -- so that exp2(s * scale - mu') = P * 2^4, the value that is split; differences (ae, alpha) are
            // what they were
            static_assert(vrd::F16_ACT_EXP == 4, "the inline constant below");
            if constexpr (F16)
                asm volatile("v_fma_f32 %0, %0, %4, -4.0\n\tv_fma_f32 %1, %1, %4, -4.0\n\tv_sub_f32 %2, %0, %5\n\tv_sub_f32 %3, %1, %6"
                             : "+v"(A.mx), "+v"(B.mx), "=&v"(A.ae), "=&v"(B.ae) : "s"(scale_log2e), "v"(A.mr), "v"(B.mr));
            else
                asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_sub_f32 %2, %0, %5\n\tv_sub_f32 %3, %1, %6"
                             : "+v"(A.mx), "+v"(B.mx), "=&v"(A.ae), "=&v"(B.ae) : "s"(scale_log2e), "v"(A.mr), "v"(B.mr));
        } else if constexpr (i == 5) {           // moved = mx - m_old > THR (m_old = -inf: always);  mu = moved ? mx : m_old
            asm volatile("v_cmp_lt_f32 %0, %4, %5\n\tv_cmp_lt_f32 %1, %4, %6\n\tv_cndmask_b32 %2, %7, %9, %0\n\tv_cndmask_b32 %3, %8, %10, %1"
                         : "=&s"(cmask[0]), "=&s"(cmask[1]), "=&v"(A.mu), "=&v"(B.mu)
                         : "v"(thr), "v"(A.ae), "v"(B.ae), "v"(A.mr), "v"(B.mr), "v"(A.mx), "v"(B.mx));
        } else if constexpr (i == 6) {
            // exponent of alpha = moved ? m_old - mx : 0 (alpha exactly 1 where the reference stays);  mu >= -FLT_MAX: a row
            // that has seen masked keys only so far (rows of more than 64 tiles visit every tile) keeps exp2(-inf - mu) = 0
            asm volatile("v_cndmask_b32 %0, 0, -%0, %4\n\tv_cndmask_b32 %1, 0, -%1, %5\n\tv_max_f32 %2, 0xff7fffff, %2\n\tv_max_f32 %3, 0xff7fffff, %3"
                         : "+v"(A.ae), "+v"(B.ae), "+v"(A.mu), "+v"(B.mu) : "s"(cmask[0]), "s"(cmask[1]));
            A.mr = A.mu;
            B.mr = B.mu;
        } else if constexpr (i == 7) {
            asm volatile("v_fma_f32 %0, %2, %4, -%5\n\tv_fma_f32 %1, %3, %4, -%6"
                         : "=&v"(A.x[0]), "=&v"(B.x[0]) : "v"(sa[0]), "v"(sb[0]), "s"(scale_log2e), "v"(A.mu), "v"(B.mu));
        } else if constexpr (i < 24) {
            if constexpr (i & 1) fea_piece(B, sb, ic<(i - 8) / 2>{});
            else fea_piece(A, sa, ic<(i - 8) / 2>{});
        } else if constexpr (i < 40) {
            split_piece(ic<0>{}, ic<i - 24>{});
        } else if constexpr (i < 56) {
            if constexpr (i & 1) fea_piece(B, sb, ic<8 + (i - 40) / 2>{});
            else fea_piece(A, sa, ic<8 + (i - 40) / 2>{});
        } else if constexpr (i < 72) {
            split_piece(ic<1>{}, ic<i - 56>{});
        } else if constexpr (i == 72) {          // alpha = exp2(ae);  l = l * alpha + (sum + the last exponential)
            float ta, tb;
            asm volatile("v_exp_f32 %2, %6\n\tv_exp_f32 %3, %7\n\tv_add_f32 %4, %8, %10\n\tv_add_f32 %5, %9, %11\n\t"
                         "v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5"
                         : "+v"(A.lp), "+v"(B.lp), "=&v"(A.alpha), "=&v"(B.alpha), "=&v"(ta), "=&v"(tb)
                         : "v"(A.ae), "v"(B.ae), "v"(A.sum), "v"(B.sum), "v"(A.x[15]), "v"(B.x[15]));
        }
    };

    // S^T MFMA j (0..5) of k16 step s: both blocks take the three products in the order (k_lo q_hi), (k_hi q_lo), (k_hi q_hi);
    // the first one of a chain starts from 0
    auto mfma_s = [&](auto s_c, auto j_c, const KF& kf, f32x16& sa, f32x16& sb) __attribute__((always_inline)) {
        constexpr int s = decltype(s_c)::value, j = decltype(j_c)::value, qb = j & 1;
        f32x16& acc = qb ? sb : sa;
        constexpr int aq = VRD_AQ(qb, s) + (j >= 2 && j < 4 ? 4 : 0);          // q_lo for the middle product
        if constexpr (s == 0 && j < 2) {
            VRD_MFMA("", "%0, %1, a[%c2:%c3], 0", : "=&v"(acc) : "v"(kf.l), "n"(aq), "n"(aq + 3));
        } else if constexpr (j < 2) {
            VRD_MFMA("", "%0, %1, a[%c2:%c3], %0", : "+v"(acc) : "v"(kf.l), "n"(aq), "n"(aq + 3));
        } else {
            VRD_MFMA("", "%0, %1, a[%c2:%c3], %0", : "+v"(acc) : "v"(kf.h), "n"(aq), "n"(aq + 3));
        }
    };
    // O^T MFMA j (0..5) of group g = (k16 step s, d tile d): (v_lo p_hi), (v_hi p_lo), (v_hi p_hi) for both blocks
    auto mfma_o = [&](auto g_c, auto j_c, const KF& vf) __attribute__((always_inline)) {
        constexpr int g = decltype(g_c)::value, j = decltype(j_c)::value, qb = j & 1, s = g / DT, d = g % DT;
        constexpr int ao = VRD_AO(qb, d);
        const SmBlock& X = qb ? B : A;
        if constexpr (j < 2) {
            if constexpr (d == 0 && j == 0)      // the fragments of this k16 step were written by vector instructions just now
                VRD_MFMA("s_nop 1\n\t", "a[%c2:%c3], %0, %1, a[%c2:%c3]", :: "v"(vf.l), "v"(X.ph[s]), "n"(ao), "n"(ao + 15));
            else
                VRD_MFMA("", "a[%c2:%c3], %0, %1, a[%c2:%c3]", :: "v"(vf.l), "v"(X.ph[s]), "n"(ao), "n"(ao + 15));
        } else if constexpr (j < 4) {
            VRD_MFMA("", "a[%c2:%c3], %0, %1, a[%c2:%c3]", :: "v"(vf.h), "v"(X.pl[s]), "n"(ao), "n"(ao + 15));
        } else {
            VRD_MFMA("", "a[%c2:%c3], %0, %1, a[%c2:%c3]", :: "v"(vf.h), "v"(X.ph[s]), "n"(ao), "n"(ao + 15));
        }
    };
    // running output of block qb *= alpha (rare: only when some query's maximum moved)
    auto rescale = [&](auto qb_c, float alpha) __attribute__((always_inline)) {
        agpr_scale<VRD_AO(decltype(qb_c)::value, 0), 16 * DT>(alpha);
    };

    // One pipeline step.  In: the raw scores of tile `it` (sa_c, sb_c; stage st), out: the raw scores of tile it+1 (WITH_S;
    // stage st1) and tile it's contribution to the outputs.  do_req: the pieces of tile kt_req are requested into stage
    // buf_req during the second half of the O^T phase.
    auto tile = [&](auto with_s_c, f32x16& sa_c, f32x16& sb_c, f32x16& sa_n, f32x16& sb_n, const char* st,
                    const char* st1, const float* kbs, bool dirty, bool do_req, const Req& req, int buf_req, bool first) __attribute__((always_inline)) {
        constexpr bool WITH_S = decltype(with_s_c)::value;
        if (dirty) {                                 // masked keys in this tile: their raw scores become -inf
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(kbs + 8 * g + 4 * lh);
                sa_c[4 * g + 0] += bv.x;
                sa_c[4 * g + 1] += bv.y;
                sa_c[4 * g + 2] += bv.z;
                sa_c[4 * g + 3] += bv.w;
                sb_c[4 * g + 0] += bv.x;
                sb_c[4 * g + 1] += bv.y;
                sb_c[4 * g + 2] += bv.z;
                sb_c[4 * g + 3] += bv.w;
            }
        }
        KF kf{}, kn{}, vf{}, vn{};
        if (WITH_S) kf = load_k(st1, 0);
        VRD_SB();
        // the first K fragment is on its way (it could not be asked for before the barrier): the row maxima run meanwhile
        static_for<LEAD>([&](auto u_c) { sm_piece(u_c, sa_c, sb_c); });
        VRD_SB();
        static_for<KS>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            static_for<6>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value, gap = 6 * s + j;
                if constexpr (WITH_S) mfma_s(s_c, j_c, kf, sa_n, sb_n);
                if constexpr (j == 0 && WITH_S && s + 1 < KS) kn = load_k(st1, s + 1);
                if constexpr (gap == NG_S - 4) vf = load_v(st, 0, 0);
                static_for<HPG_S>([&](auto u_c) {
                    constexpr int piece = LEAD + gap * HPG_S + decltype(u_c)::value;
                    if constexpr (piece < 48) sm_piece(ic<piece>{}, sa_c, sb_c);
                });
                VRD_SB();
            });
            kf = kn;
        });
        if (!first && cmask[0] != 0ull) rescale(ic<0>{}, __builtin_amdgcn_exp2f(A.ae));
        if (!first && cmask[1] != 0ull) rescale(ic<1>{}, __builtin_amdgcn_exp2f(B.ae));
        VRD_SB();
        // ---- O^T += V^T . P^T for both blocks: each V^T fragment pair feeds six MFMAs, fragments one group ahead
        static_for<2 * DT>([&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            static_for<6>([&](auto j_c) {
                constexpr int j = decltype(j_c)::value, gap = 6 * g + j;
                mfma_o(g_c, j_c, vf);
                if constexpr (j == 0 && g + 1 < 2 * DT) vn = load_v(st, (g + 1) / DT, (g + 1) % DT);
                static_for<HPG_O>([&](auto u_c) { sm_piece(ic<48 + gap * HPG_O + decltype(u_c)::value>{}, sa_c, sb_c); });
                constexpr int rq = gap - NG_O / 2 - 1;
                if constexpr (rq >= 0 && rq % 3 == 0 && rq / 3 < PER_WAVE) {
                    if (do_req) issue1(req, buf_req, ic<rq / 3>{});
                }
                VRD_SB();
            });
            vf = vn;
        });
    };

    // ---- prologue: the first NS-1 visited tiles are requested, S^T of the first one is formed
    int kt_iss = -1;                                     // last tile requested
    int issued = 0;                                      // tiles requested so far (sequence numbers 0 .. issued-1)
    auto request_next = [&](int buf) {
        if (issued < n_act) {
            kt_iss = next_on(kt_iss + 1);
            const Req r = req_of(kt_iss);
            static_for<PER_WAVE>([&](auto i_c) { issue1(r, buf, i_c); });
            ++issued;
        }
    };
    VRD_STAMP(51);
    // the Q^T fragments move to their accumulator-half registers; then the wave's quarter of the ring belongs to the tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the wave's own rows (and, older, the next item's mask bytes) have landed
    masks_pending = false;
    VRD_STAMP(52);
    static_for<2>([&](auto qb_c) {
        constexpr int qb = decltype(qb_c)::value;
        static_for<KS>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            const char* stq = lds + (2 * qb + (wave >> 1)) * G::STAGE + (wave & 1) * 2 * G::PLANE;
            const int off = li * G::ROWB + (((2 * s + lh) ^ G::kswz(li)) * 16);
            const u32x4 th = *reinterpret_cast<const u32x4*>(stq + off), tl = *reinterpret_cast<const u32x4*>(stq + G::PLANE + off);
            constexpr int r = VRD_AQ(qb, s);
            asm volatile("v_accvgpr_write_b32 a%c8, %0\n\tv_accvgpr_write_b32 a%c9, %1\n\tv_accvgpr_write_b32 a%c10, %2\n\t"
                         "v_accvgpr_write_b32 a%c11, %3\n\tv_accvgpr_write_b32 a%c12, %4\n\tv_accvgpr_write_b32 a%c13, %5\n\t"
                         "v_accvgpr_write_b32 a%c14, %6\n\tv_accvgpr_write_b32 a%c15, %7"
                         :: "v"(th[0]), "v"(th[1]), "v"(th[2]), "v"(th[3]), "v"(tl[0]), "v"(tl[1]), "v"(tl[2]), "v"(tl[3]), "n"(r), "n"(r + 1),
                            "n"(r + 2), "n"(r + 3), "n"(r + 4), "n"(r + 5), "n"(r + 6), "n"(r + 7));
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (qb == 0) {                 // stages 0 and 1 are free
            request_next(0);
            request_next(1);
        } else {
            request_next(2);
        }
    });
    static_assert(NS == 4, "the prologue requests three tiles");
    VRD_STAMP(2);
    int kt_cur = next_on(0);                             // tile whose raw scores the next step consumes
    f32x16 s0a, s0b, s1a, s1b;
#pragma unroll
    for (int e = 0; e < 16; ++e) s0a[e] = s0b[e] = s1a[e] = s1b[e] = 0.f;
    if (n_act > 0) {
        // tile 0 has landed when at most the requests of the tiles behind it are outstanding
        if (n_act >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
        else if (n_act == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        VRD_STAMP(54);
        if (q_live) {
            KF kf = load_k(lds, 0), kn = kf;
            static_for<KS>([&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                if constexpr (s + 1 < KS) kn = load_k(lds, s + 1);
                static_for<6>([&](auto j_c) { mfma_s(s_c, j_c, kf, s0a, s0b); });
                kf = kn;
            });
            asm volatile("s_nop 15" : "+v"(s0a), "+v"(s0b));       // MFMA result -> vector read: wait states by hand
        }
    }
    VRD_STAMP(3);
    // a step: wait for tile it+1, barrier, pipeline step (scores of tile `it` in (ca, cb), of tile it+1 into (na, nb))
    int it = 0;
    auto step = [&](f32x16& ca, f32x16& cb, f32x16& na, f32x16& nb) __attribute__((always_inline)) {
        const bool has_next = it + 1 < n_act;
        const int kt_next = has_next ? next_on(kt_cur + 1) : nkt;
        // tile it+1 landed (mine: counted wait; everybody's: the barrier); the barrier also says that nobody reads the
        // stage of tile it-1 any more, which the request for tile it+3 refills
        if (has_next) {
            if (it + 2 < issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        VRD_STAMP(4 + 3 * it);
        __builtin_amdgcn_s_barrier();
        VRD_STAMP(5 + 3 * it);
        const bool do_req = issued < n_act;
        if (do_req) {
            kt_iss = next_on(kt_iss + 1);
            ++issued;
        }
        const Req req = req_of(do_req ? kt_iss : 0);
        const char* st = lds + (it & (NS - 1)) * G::STAGE;
        const char* st1 = lds + ((it + 1) & (NS - 1)) * G::STAGE;
        const float* kbs = kbias + kt_cur * 32;
        const int buf_req = (it + NS - 1) & (NS - 1);
        if (q_live) {
            const bool dirty = nkt <= 64 ? ((cln >> kt_cur) & 1ull) == 0ull : __builtin_amdgcn_readfirstlane(tile_on[kt_cur]) != 2;
            if (has_next) tile(std::true_type{}, ca, cb, na, nb, st, st1, kbs, dirty, do_req, req, buf_req, it == 0);
            else tile(std::false_type{}, ca, cb, na, nb, st, st1, kbs, dirty, do_req, req, buf_req, it == 0);
        } else if (do_req) {
            static_for<PER_WAVE>([&](auto i_c) { issue1(req, buf_req, i_c); });
        }
        kt_cur = kt_next;
        VRD_STAMP(6 + 3 * it);
        ++it;
    };
    while (it < n_act) {
        step(s0a, s0b, s1a, s1b);
        if (it >= n_act) break;
        step(s1a, s1b, s0a, s0b);
    }

    // ---- epilogue.  MFMA result -> v_accvgpr_read: wait states by hand (the compiler does not know the asm statements are
    // MFMAs).  The outputs go through LDS: a lane holds 4 channels of ONE query per register quad, so stores straight from
    // the accumulators touch 32 rows per instruction, 32 bytes of each (8 k cycles per workgroup, measured); staged, an
    // instruction writes whole 512-byte rows.  The ring is free once every wave has passed the barrier (the last tile's V):
    // wave w's 64 x HD f32 slab is the ring's bytes [w * 64 * HD * 4, ...).  16-byte chunk c of row r is stored at chunk
    // c ^ (r % chunks) (conflict-free writes -- lanes are rows -- and reads -- lanes are chunks).
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_s_barrier();
    VRD_STAMP(40);
    constexpr int CH = HD / 4;                                       // 16-byte chunks per output row of this head
    char* const slab = lds + wave * (64 * HD * 4);
    static_for<2>([&](auto qb_c) {
        constexpr int qb = decltype(qb_c)::value;
        const float l_tot = xchg32_sum(qb ? B.lp : A.lp);
        // f16: accumulators = sum (P 2^e)(V 2^e), l = sum P 2^e: acc / l is the output times 2^e -- what a pair row stores
        const float inv = (qb ? live1 : live0) ? ((F16 && !pair_out) ? vrd::F16_ACT_INV : 1.0f) / l_tot : 0.f;
        static_for<DT>([&](auto d_c) {
            constexpr int d = decltype(d_c)::value;
            static_for<4>([&](auto g_c) {
                constexpr int g = decltype(g_c)::value, r = VRD_AO(qb, d) + 4 * g;
                float4 val;                                        // registers 4g..4g+3 are d = 32d + 8g + 4lh + 0..3
                asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\t"
                             "v_accvgpr_read_b32 %3, a%c7"
                             : "=v"(val.x), "=v"(val.y), "=v"(val.z), "=v"(val.w) : "n"(r), "n"(r + 1), "n"(r + 2), "n"(r + 3));
                val.x *= inv;
                val.y *= inv;
                val.z *= inv;
                val.w *= inv;
                const int c16 = 8 * d + 2 * g + lh;
                *reinterpret_cast<float4*>(slab + (32 * qb + li) * (HD * 4) + ((c16 ^ (li & (CH - 1))) * 16)) = val;
            });
        });
    });
    VRD_STAMP(41);
    // (wave-private slab: the wave's own LDS writes are ordered before its reads by the compiler's lgkmcnt wait)
    // Rows leave in batches of eight wave instructions: the reads of a batch together, then its stores; the row pointer
    // advances by a constant (no 64-bit multiply per row), and the rows-inside-the-sequence test is per wave unless the
    // wave's 64 rows straddle Tq.  f32 output: a lane stores one 16-byte chunk (4 channels); pair-row output: a lane takes
    // two chunks (8 channels) and stores 16 bytes of hi and 16 bytes of lo (8-byte stores run at ~0.6 of the 16-byte rate).
    const bool all_rows = q0 + 64 <= Tq;                             // wave-uniform
    if (pair_out) {
        constexpr int LPR = CH / 2;                                  // lanes per row (16 or 8)
        constexpr int ROWS_PI = 64 / LPR;                            // rows per wave instruction (4 or 8)
        constexpr int NT = 64 / ROWS_PI, BATCH = 4;
        const int rr = lane / LPR, cj = lane % LPR;                  // chunks 2 cj, 2 cj + 1 = channels 8 cj .. 8 cj + 7
        char* gp = reinterpret_cast<char*>(out + ((int64_t)b * Tq + (q0 + rr < Tq ? q0 + rr : Tq - 1)) * ldo) +
                   vrd::pair_index(h * HD + 8 * cj) * 2;
        const int64_t gstep = ldo * 4 * ROWS_PI;
#pragma unroll 1
        for (int t0 = 0; t0 < NT; t0 += BATCH) {
            float4 va[BATCH], vb[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = (t0 + u) * ROWS_PI + rr;
                va[u] = *reinterpret_cast<const float4*>(slab + r * (HD * 4) + (((2 * cj) ^ (r & (CH - 1))) * 16));
                vb[u] = *reinterpret_cast<const float4*>(slab + r * (HD * 4) + (((2 * cj + 1) ^ (r & (CH - 1))) * 16));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = (t0 + u) * ROWS_PI + rr;
                if (all_rows || q0 + r < Tq) {
                    uint4 hi, lo;
                    split_pair<F16>(va[u].x, va[u].y, hi.x, lo.x);
                    split_pair<F16>(va[u].z, va[u].w, hi.y, lo.y);
                    split_pair<F16>(vb[u].x, vb[u].y, hi.z, lo.z);
                    split_pair<F16>(vb[u].z, vb[u].w, hi.w, lo.w);
#ifndef VRD_ATTN_LAB_NOSTORE
                    *reinterpret_cast<uint4*>(gp) = hi;
                    *reinterpret_cast<uint4*>(gp + 64) = lo;
#else
                    asm volatile("" ::"v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w), "v"(lo.x), "v"(lo.y), "v"(lo.z), "v"(lo.w));
#endif
                }
                gp += gstep;
            }
        }
    } else {
        constexpr int ROWS_PI = 64 / CH;                             // rows per wave instruction (2 or 4)
        constexpr int NT = 64 / ROWS_PI, BATCH = 8;
        const int rr = lane / CH, cj = lane % CH;
        char* gp = reinterpret_cast<char*>(out + ((int64_t)b * Tq + (q0 + rr < Tq ? q0 + rr : Tq - 1)) * ldo) + (h * HD + 4 * cj) * 4;
        const int64_t gstep = ldo * 4 * ROWS_PI;
#pragma unroll 1
        for (int t0 = 0; t0 < NT; t0 += BATCH) {
            float4 val[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = (t0 + u) * ROWS_PI + rr;               // row of the wave's 64
                val[u] = *reinterpret_cast<const float4*>(slab + r * (HD * 4) + ((cj ^ (r & (CH - 1))) * 16));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = (t0 + u) * ROWS_PI + rr;
                if (all_rows || q0 + r < Tq) *reinterpret_cast<float4*>(gp) = val[u];
                gp += gstep;
            }
        }
    }
    VRD_STAMP(42 + qblk);
    }       // 256-query blocks
#ifdef VRD_ATTN_STAMP
    VRD_STAMP(50);
    if (blockIdx.x == 0 && wave == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        g_attn_stamp[lane] += stamp_lds[lane] - stamp_lds[0];
        if (lane == 0) g_attn_stamp[64] += 1;
    }
#endif
    }       // items
#undef VRD_AQ
#undef VRD_AO
#undef VRD_MFMA
}

template <int HD, bool F16>
int launch_w64(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask, const uint8_t* q_mask, int B,
               int Tq, int Tk, int n_head, float scale, float* out, int64_t ldo, int pair_out, hipStream_t s) {
    auto kern = attn_flash_x3_w64_kernel<HD, F16>;
    // behind the ring: key bias + tile flags for Tk <= 4096, a flag per 32 queries (Tq <= 65536: the dispatch keeps longer ones
    // away), and -- if it fits in what is left of the 160 KiB -- the staging area of the next item's mask bytes, a dword each
    constexpr size_t lds_all = 160 * 1024;
    constexpr size_t lds_max = 4 * AG<HD>::STAGE + (4096 + 128 + 2048) * sizeof(float);
    const size_t nkt = (size_t)(Tk + 31) / 32, nqg = (size_t)(Tq + 31) / 32;
    const size_t lds_tables = 4 * AG<HD>::STAGE + (nkt * 33 + nqg) * sizeof(float);
    const size_t lds_staged = lds_tables + (((nkt * 32 + 63) & ~(size_t)63) + ((nqg * 32 + 63) & ~(size_t)63)) * sizeof(float);
    if (lds_tables > lds_max) {
        vrd::set_error("vrd_attention_pair: Tk = %d exceeds the 4096 keys the key-bias row is sized for", Tk);
        return -1;
    }
    const char* const pf_e = getenv("VRD_FLASH_PREFETCH");                  // (read per call, as VRD_FLASH_W64)
    const int prefetch = (!pf_e || atoi(pf_e) != 0) && lds_staged <= lds_all;
    if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(kern), lds_all, "vrd_attention_pair(w64)")) return rc;
    const int q_blocks = (Tq + 255) / 256;
#ifdef VRD_ATTN_STAMP
    const size_t lds_launch = lds_all;
#else
    const size_t lds_launch = prefetch ? lds_staged : lds_tables;
#endif
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int n_items = n_head * B;
    // VRD_FLASH_GRID caps the number of workgroups (read per call: the tests walk several items per workgroup on small inputs)
    const char* const grid_e = getenv("VRD_FLASH_GRID");
    const int grid_cap = grid_e && atoi(grid_e) > 0 ? atoi(grid_e) : n_cu;
    const int grid = n_items < grid_cap ? n_items : grid_cap;
    // a workgroup walks a contiguous run of items (the heads of a sequence, then the next sequence: measured 2 % faster than
    // items blockIdx.x, + grid, ... -- 37 MB apart in each tensor at the benchmark shape); VRD_FLASH_WALK=0: the strided walk
    const char* const walk_e = getenv("VRD_FLASH_WALK");
    const int item_step = walk_e && atoi(walk_e) == 0 ? grid : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_launch, s, q, ldq, k, v, ldkv, kv_mask, q_mask,
                       Tq, Tk, n_head * HD, scale * 1.44269504088896340736f, out, ldo, pair_out, q_blocks, n_head, B, prefetch, item_step);      // (scale: see vrd_attention_pair)
    return 0;
}

template <int HD, int NW, bool F16>
int launch(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask, const uint8_t* q_mask, int B, int Tq,
           int Tk, int n_head, float scale, float* out, int64_t ldo, int pair_out, hipStream_t s) {
    auto kern = attn_flash_x3_kernel<HD, NW, F16>;
    constexpr size_t lds_max = 2 * AG<HD>::STAGE + (4096 + 128) * sizeof(float);       // key bias + tile flags for Tk <= 4096
    const size_t lds = 2 * AG<HD>::STAGE + (size_t)((Tk + 31) / 32) * 33 * sizeof(float);
    if (lds > lds_max) {
        vrd::set_error("vrd_attention_pair: Tk = %d exceeds the 4096 keys the key-bias row is sized for", Tk);
        return -1;
    }
    if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(kern), lds_max, "vrd_attention_pair")) return rc;
    const int tiles = (Tq + 31) / 32;
    const int q_blocks = (tiles + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3((unsigned)q_blocks * n_head * B), dim3(NW * 64), lds, s, q, ldq, k, v, ldkv, kv_mask, q_mask, Tq, Tk,
                       n_head * HD, scale, out, ldo, pair_out, q_blocks, n_head);
    return 0;
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

}  // namespace

#ifdef VRD_ATTN_STAMP
extern "C" int vrd_lab_attn_stamps(unsigned long long* dst65, int reset) {
    if (hipMemcpyFromSymbol(dst65, HIP_SYMBOL(g_attn_stamp), 65 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[65] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int vrd_attention_pair(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                  const uint8_t* kv_mask, const uint8_t* q_mask, int B, int Tq, int Tk, int n_head, int head_dim,
                                  float* out, int64_t ldo, int out_pair, int pair_fmt, void* stream) {
    VRD_CHECK_ARG(q && k && v && out, "vrd_attention_pair: null pointer");
    VRD_CHECK_ARG(head_dim == 64 || head_dim == 128, "vrd_attention_pair: head_dim must be 64 or 128 (got %d)", head_dim);
    VRD_CHECK_ARG(B > 0 && B <= 65535 && Tq > 0 && Tk > 0 && n_head > 0 && n_head <= 65535, "vrd_attention_pair: bad sizes");
    const int width = n_head * head_dim;
    VRD_CHECK_ARG(ldq >= width && ldkv >= width && ldo >= width && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 &&
                      aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out),
                  "vrd_attention_pair: rows must be 16-byte aligned pair rows of width n_head*head_dim");
    VRD_CHECK_ARG(pair_fmt == VRD_PAIR_BF16 || pair_fmt == VRD_PAIR_F16, "vrd_attention_pair: pair_fmt must be VRD_PAIR_BF16 or VRD_PAIR_F16");
    VRD_CHECK_ARG(out_pair == VRD_PAIR_NONE || out_pair == pair_fmt, "vrd_attention_pair: pair output comes in the operands' format");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool f16 = pair_fmt == VRD_PAIR_F16;
    // f16 operands hold q and k times 2^VRD_F16_ACT_EXP each: the raw scores are 2^(2 e) too large, undone in the softmax scale
    const float scale = (f16 ? vrd::F16_ACT_INV * vrd::F16_ACT_INV : 1.0f) / sqrtf((float)head_dim);
    vrd::ProfScope prof(VRD_K_ATTN_FLASH, s, 4.0 * B * (double)n_head * Tq * Tk * head_dim,
                        4.0 * B * (double)width * (2.0 * Tq + 2.0 * Tk));
    // waves (32-query tiles) per workgroup: every workgroup streams the whole K / V row of its (b, h), and the kernel is
    // bound by that stream (LDS-DMA issue), so fewer, fuller workgroups win even when the last one is mostly empty:
    // 288 queries = 9 tiles run as 4 + 4 + 1 (the lone tile is the padded tail at the benchmark shape and exits at
    // once) 17 % faster than as 3 + 3 + 3.  Two workgroups of 4 waves fill a CU's registers (247 VGPRs per wave); ONE
    // workgroup of 8 waves (a single stream per (b, h)) was measured 70 % slower: nothing runs while it waits at its
    // barrier for a tile.
    static const int nw_env = [] { const char* e = getenv("VRD_FLASH_NW"); return e ? atoi(e) : 0; }();
    const int nw = nw_env == 3 || nw_env == 4 ? nw_env : 4;
    // 64 queries per wave, one wave per SIMD (attn_flash_x3_w64_kernel): head_dim 128 from 160 query rows.  Measured per launch
    // (scripts/flash_bench.py --pair, MI355X, round 6: its item loop rebuilt), 4 heads x 128, 2048 sequences of T rows, this kernel
    // against the two-waves-per-SIMD one: T = 96: 0.51 / 0.36 ms, 128: 0.60 / 0.48, 160: 0.68 / 0.78, 192: 0.78 / 0.91, 224: 0.95 / 1.08,
    // 288 (256 valid): 1.07 / 1.25-1.33.  At head_dim 64 it has half the MFMAs per softmax instruction and no more than ties (8 heads x
    // 64, 1024 sequences: T = 128: 0.37 / 0.25, 256: 0.63 / 0.63, 288: 1.13 / 0.88 -- a second block of 32 rows --, 512: 2.00 / 2.01).
    // VRD_FLASH_W64=0 / 1 forces the choice (read per call: tests compare the two kernels in one process).
    const char* const w64_e = getenv("VRD_FLASH_W64");
    const int w64_env = w64_e ? atoi(w64_e) : -1;
    // (its LDS-DMA offsets are 32-bit: a batch element's Q and K / V slabs have to stay below 2 GiB)
    const bool w64 = (w64_env >= 0 ? w64_env != 0 : (head_dim == 128 && Tq >= 160)) && (int64_t)Tk * ldkv * 4 < (int64_t(1) << 31) &&
                     (int64_t)Tq * ldq * 4 < (int64_t(1) << 31) && Tq <= 65536;
    int rc;
#define VRD_ATTN_ARGS q, ldq, k, v, ldkv, kv_mask, q_mask, B, Tq, Tk, n_head, scale, out, ldo, out_pair, s
    if (w64) rc = head_dim == 128 ? (f16 ? launch_w64<128, true>(VRD_ATTN_ARGS) : launch_w64<128, false>(VRD_ATTN_ARGS))
                                  : (f16 ? launch_w64<64, true>(VRD_ATTN_ARGS) : launch_w64<64, false>(VRD_ATTN_ARGS));
    else if (head_dim == 128) rc = nw == 3 ? (f16 ? launch<128, 3, true>(VRD_ATTN_ARGS) : launch<128, 3, false>(VRD_ATTN_ARGS))
                                           : (f16 ? launch<128, 4, true>(VRD_ATTN_ARGS) : launch<128, 4, false>(VRD_ATTN_ARGS));
    else rc = nw == 3 ? (f16 ? launch<64, 3, true>(VRD_ATTN_ARGS) : launch<64, 3, false>(VRD_ATTN_ARGS))
                      : (f16 ? launch<64, 4, true>(VRD_ATTN_ARGS) : launch<64, 4, false>(VRD_ATTN_ARGS));
#undef VRD_ATTN_ARGS
    if (rc) return rc;
    VRD_LAUNCH_CHECK();
    return 0;
}
