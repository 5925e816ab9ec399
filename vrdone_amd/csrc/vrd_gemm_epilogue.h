// Epilogue shared by the f32 and the split-precision GEMM kernels: both leave a 64 x 64 sub-tile per
// wave as 2 x 2 blocks of 32 x 32 (one 32x32 MFMA accumulator each, or 2 x 2 16x16 ones: forall32 below).
#pragma once
#include "vrd_common.h"
#include <type_traits>

namespace vrd {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int STG_PITCH = 64;   // staging slab: 32 rows x 64 floats per wave = 8 KiB

// A 32 x 32 block of C in registers, two forms:
//   f32x16 -- one accumulator of a 32x32 MFMA (the exact-f32 kernel, v_mfma_f32_32x32x2_f32): element e of lane (li, lh) is
//             C[(e & 3) + 8 * (e >> 2) + 4 * lh][li];
//   acc32q -- 2 x 2 accumulators of v_mfma_f32_16x16x32 (every split-precision GEMM kernel: the shape the chip holds its
//             highest clock on, MI355X_MICROARCH.md DVFS item 7): element j of sub-block (bi, bj) in lane l is
//             C[16 bi + 4 (l >> 4) + j][16 bj + (l & 15)].
// forall32(acc, lane, f) calls f(row, column, value) for the lane's 16 elements (unrolled; row / column inside the block).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct acc32q {
    f32x4_t b[2][2];
};
__device__ __forceinline__ void acc_clear(f32x16& a) {
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
}
__device__ __forceinline__ void acc_clear(acc32q& a) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) a.b[i][j][e] = 0.f;
}
template <typename Fn>
__device__ __forceinline__ void forall32(const f32x16& a, int lane, Fn&& f) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) f((e & 3) + 8 * (e >> 2) + 4 * lh, li, a[e]);
}
template <typename Fn>
__device__ __forceinline__ void forall32(const acc32q& a, int lane, Fn&& f) {
    const int lc = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int j = 0; j < 4; ++j) f(16 * bi + 4 * lq + j, 16 * bj + lc, a.b[bi][bj][j]);
}

__device__ __forceinline__ float epilogue_value(const vrd_gemm_args& p, float v, float mk, float scale, float r1, float rmk,
                                                float r2) {
    if (p.act == VRD_ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == VRD_ACT_GELU) v = gelu_erf(v);
    return v * mk * scale + r1 * rmk + r2;
}

// v = acc + bias; v = act(v); v *= row_mask; v *= scale; v += res * (res_masked ? row_mask : 1); v += res2
// Per-column epilogue inputs of a lane's float4 column group (columns nw + 4*(lane & 15) ...), loaded by the kernel
// before its main loop so that their latency is not paid, exposed, by the tile's epilogue.
struct EpiCols {
    float bias[4], scale[4];
    float alpha;        // accumulator factor: 1, or (VRD_PAIR_F16) the power of two that undoes the operand scaling
};
// the factor on the accumulators of a split-precision GEMM (wave-uniform; a scalar load)
__device__ __forceinline__ float acc_alpha(const vrd_gemm_args& p) {
    if (!(p.split_fmt == VRD_PAIR_F16 && p.w_scale)) return 1.0f;
    const float alpha = uniform_load(p.w_scale);          // 2^-(e_w + VRD_F16_ACT_EXP)
    // (rows split with the caller's factor 2^e instead of 2^VRD_F16_ACT_EXP: a_scale[1] = 2^-e)
    return (p.a_scale && p.a_pair_width == 0) ? alpha * F16_ACT_SCALE * uniform_load(p.a_scale + 1) : alpha;
}
__device__ __forceinline__ EpiCols load_epi_cols(const vrd_gemm_args& p, int nw, int lane) {
    EpiCols c;
    c.alpha = acc_alpha(p);
    const int n = nw + (lane & 15) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) c.bias[j] = 0.f, c.scale[j] = 1.f;
    if (n + 3 < p.N) {                 // (the caller guarantees 16-byte aligned bias / scale for the lean epilogue)
        if (p.bias) {
            const float4 b = gload4(p.bias + n);
            c.bias[0] = b.x, c.bias[1] = b.y, c.bias[2] = b.z, c.bias[3] = b.w;
        }
        if (p.scale) {
            const float4 b = gload4(p.scale + n);
            c.scale[0] = b.x, c.scale[1] = b.y, c.scale[2] = b.z, c.scale[3] = b.w;
        }
    }
    return c;
}

// The same through loads the compiler does not see (inline asm): the caller waits for them itself (`s_waitcnt vmcnt`) before
// the epilogue.  For the persistent 256 x 256 kernel: hipcc follows a loop-carried register that a tracked load once wrote
// with `s_waitcnt vmcnt(0)` at the top of the tile loop, i.e. with a wait for the previous tile's stores.
__device__ __forceinline__ EpiCols load_epi_cols_async(const vrd_gemm_args& p, int nw, int lane) {
    EpiCols c;
    c.alpha = acc_alpha(p);
    const int n = nw + (lane & 15) * 4;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (n + 3 < p.N) {
        if (p.bias) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(b) : "v"(p.bias + n) : "memory");
        if (p.scale) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sc) : "v"(p.scale + n) : "memory");
    }
    c.bias[0] = b.x, c.bias[1] = b.y, c.bias[2] = b.z, c.bias[3] = b.w;
    c.scale[0] = sc.x, c.scale[1] = sc.y, c.scale[2] = sc.z, c.scale[3] = sc.w;
    return c;
}

// Lean epilogue of the 256 x 256 kernel: a 64 x 64 wave sub-tile that lies fully inside C (the host sends only
// GEMMs with M % 64 == 0 and N % 64 == 0 to that kernel, so a sub-tile is inside or entirely outside), 16-byte
// aligned rows, bias and scale, 64-row private slab.  Which terms exist is a template parameter, so a pass is
// straight-line code without per-row predicates or per-element selects: the epilogue is instruction-bound (two
// waves per SIMD, ~8 cycles per instruction and wave; the general version below runs ~240 instructions per
// 16-row pass).   ROWIN: any of row_mask / scale / res / res2 may be set;  ACT: VRD_ACT_NONE or VRD_ACT_GELU.
// SLAB: rows of the wave's private staging slab `stg`: 64 (both 32-row halves transposed up front) or 32 (the second
// half is transposed after the first one's passes, into the same slab: DS operations of a wave execute in order).
// ROWS: 64 (four passes), or 32: one 32-row block, two passes, mw1 unused.
template <bool ROWIN, int ACT, int SLAB, int ROWS, typename Transposer>
__device__ __forceinline__ void gemm_epilogue_lean_tr(const vrd_gemm_args& p, Transposer&& transpose_into, float* stg, int64_t mw,
                                                      int64_t mw1, int nw, int lane, const EpiCols& cols, unsigned* rflag = nullptr) {
    RangeTrack rt;            // largest scaled magnitude written as an f16 pair (reported once, behind the last pass)
    // rows: passes 0-1 are mw .. mw+31, passes 2-3 are mw1 .. mw1+31 (mw1 = mw + 32 unless the tile's 32-row blocks
    // come from a block list); hop = wave-uniform distance the second block is away from its contiguous place
    const int64_t hop = mw1 - (mw + 32);
    const int c4 = (lane & 15) * 4, rb0 = lane >> 4;
    const int n = nw + c4;
    // lane base pointers (row mw + rb0, column n); the rows of the passes are wave-uniform offsets from them
    const int64_t m_lane = mw + rb0;
    const bool pair = p.c_pair != 0;
    char* const c_lane = reinterpret_cast<char*>(p.C + m_lane * p.ldc) + (pair ? pair_index(n) * 2 : n * 4);
    const float* const r1_lane = (ROWIN && p.res) ? p.res + m_lane * p.ldres + n : nullptr;
    const float* const r2_lane = (ROWIN && p.res2) ? p.res2 + m_lane * p.ldres2 + n : nullptr;
    const unsigned char* const mk_lane = (ROWIN && p.row_mask) ? p.row_mask + m_lane : nullptr;
    const int64_t c_step = p.ldc * 16, r1_step = p.ldres * 4, r2_step = p.ldres2 * 4;     // bytes / floats per 4 rows
    const int64_t c_hop = hop * p.ldc * 4;
    struct RowIn {
        unsigned char mb[4];
        float4 r1[4], r2[4];
    };
    auto fetch = [&](int pass) {
        RowIn in;
        if (ROWIN) {
            if (p.row_mask) {
#pragma unroll
                for (int j = 0; j < 4; ++j) in.mb[j] = *as_global(mk_lane + pass * 16 + 4 * j + (pass >= 2 ? hop : 0));
            }
            if (p.res) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    in.r1[j] = gload4(r1_lane + (pass * 4 + j) * r1_step + (pass >= 2 ? hop * p.ldres : 0));
            }
            if (p.res2) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    in.r2[j] = gload4(r2_lane + (pass * 4 + j) * r2_step + (pass >= 2 ? hop * p.ldres2 : 0));
            }
        }
        return in;
    };
    // row inputs run two passes ahead (HBM latency under load is several thousand cycles, a pass ~1.5k); the first
    // two requests go out before the transposition
    constexpr int NPASS = ROWS / 16;
    RowIn q0 = fetch(0), q1 = fetch(1);
    unsigned rd_addr[4];          // LDS byte addresses of rows rb0 + 4 j, this lane's (swizzled) float4 column group
    {
        typedef __attribute__((address_space(3))) const float* lds_cf;
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_addr[j] = (unsigned)reinterpret_cast<uintptr_t>((lds_cf)(stg + (rb0 + 4 * j) * STG_PITCH + (c4 ^ (16 * j))));
    }
    transpose_into(stg, SLAB == 64 ? -1 : 0);            // the wave's 64 x 64 sub-tile (or its upper half), accumulator layout -> slab rows
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (SLAB == 32 && pass == 2) transpose_into(stg, 1);
        const RowIn cur = q0;
        q0 = q1;
        if (pass + 2 < NPASS) q1 = fetch(pass + 2);
        float v[4][4];
        // The slab is read back by instructions the compiler does not see: next to LDS-DMA requests in flight (the persistent
        // kernel's look-ahead for its next tile, which lands in OTHER ring slots) hipcc puts `s_waitcnt vmcnt(0)` in front of
        // every LDS read it cannot tell apart from them.  DS operations of a wave execute in order, so the reads see the
        // transposition's writes.
        // (slab columns are swizzled in blocks of 16 with (row >> 2) & 3 -- see gemm_epilogue_lean16 --, which for the rows
        // srow + 4 j of this lane is j: one address per j, the pass is an immediate offset)
        float4 tr[4];
        {
            constexpr int PASS_OFF = (SLAB == 64 ? 1 : 0);      // (SLAB 64: pass p reads rows 16 p ..; SLAB 32: rows 16 (p & 1) ..)
            const int poff = (PASS_OFF ? pass : (pass & 1)) * 16 * STG_PITCH * 4;
            asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(tr[0]), "=&v"(tr[1]), "=&v"(tr[2]), "=&v"(tr[3])
                         : "v"(rd_addr[0]), "v"(rd_addr[1]), "v"(rd_addr[2]), "v"(rd_addr[3]), "n"(poff)
                         : "memory");
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = tr[j];
            // (alpha = 1 outside the f16 format: fmaf(t, 1, b) is t + b to the last bit)
            v[j][0] = fmaf(t.x, cols.alpha, cols.bias[0]), v[j][1] = fmaf(t.y, cols.alpha, cols.bias[1]);
            v[j][2] = fmaf(t.z, cols.alpha, cols.bias[2]), v[j][3] = fmaf(t.w, cols.alpha, cols.bias[3]);
        }
        if (ACT == VRD_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    f32x2_t t2;
                    t2.x = v[j][c], t2.y = v[j][c + 1];
                    t2 = gelu_erf2(t2);
                    v[j][c] = t2.x, v[j][c + 1] = t2.y;
                }
        }
        if (ROWIN) {
            float mk[4] = {1.f, 1.f, 1.f, 1.f};
            if (p.row_mask) {
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[j] = (float)cur.mb[j];
            }
            if (p.row_mask || p.scale) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[j][c] *= mk[j] * cols.scale[c];
            }
            if (p.res) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rmk = p.res_masked ? mk[j] : 1.f;
                    v[j][0] += cur.r1[j].x * rmk, v[j][1] += cur.r1[j].y * rmk;
                    v[j][2] += cur.r1[j].z * rmk, v[j][3] += cur.r1[j].w * rmk;
                }
            }
            if (p.res2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j][0] += cur.r2[j].x, v[j][1] += cur.r2[j].y;
                    v[j][2] += cur.r2[j].z, v[j][3] += cur.r2[j].w;
                }
            }
        }
        if (pair) {
            if (p.c_pair == VRD_PAIR_F16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    char* const rowp = c_lane + (pass * 4 + j) * c_step + (pass >= 2 ? c_hop : 0);
                    f16x4_t h, l;
                    split_n<true>(v[j], h, l, &rt);
                    *as_global(reinterpret_cast<f16x4_t*>(rowp)) = h;
                    *as_global(reinterpret_cast<f16x4_t*>(rowp + 64)) = l;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    char* const rowp = c_lane + (pass * 4 + j) * c_step + (pass >= 2 ? c_hop : 0);
                    bf16x4_t h, l;
                    split_n<false>(v[j], h, l);
                    *as_global(reinterpret_cast<bf16x4_t*>(rowp)) = h;
                    *as_global(reinterpret_cast<bf16x4_t*>(rowp + 64)) = l;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gstore4(c_lane + (pass * 4 + j) * c_step + (pass >= 2 ? c_hop : 0), make_float4(v[j][0], v[j][1], v[j][2], v[j][3]));
        }
    }
    if (p.c_pair == VRD_PAIR_F16) rt.report(rflag, RANGE_GEMM_OUT);
}

// 16 x 16 accumulators (v_mfma_f32_16x16x32): element j of lane l is C[4 * (l >> 4) + j][l & 15].  The four row groups of a
// wave (l >> 4) write the same 16 columns of rows 4 apart: with a 64-float slab pitch that is one bank for all four, so the slab's
// 16-column blocks are XOR-swizzled with (row >> 2) & 3 = l >> 4 (the 128 transposition writes per wave and tile ran 4-way conflicted).
template <bool ROWIN, int ACT, int SLAB = 64>
__device__ __forceinline__ void gemm_epilogue_lean16(const vrd_gemm_args& p, const f32x4_t (&acc)[4][4], float* stg, int64_t mw,
                                                     int64_t mw1, int nw, int lane, const EpiCols& cols, unsigned* rflag = nullptr) {
    const int lc = lane & 15, lq = lane >> 4;
    gemm_epilogue_lean_tr<ROWIN, ACT, SLAB, 64>(
        p,
        [&](float* slab, int half) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                if (half >= 0 && (ti >> 1) != half) continue;
                const int r0 = half < 0 ? ti * 16 : (ti & 1) * 16;
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int j = 0; j < 4; ++j) slab[(r0 + 4 * lq + j) * STG_PITCH + (tj ^ lq) * 16 + lc] = acc[ti][tj][j];
            }
        },
        stg, mw, mw1, nw, lane, cols, rflag);
}

// true when vrd_gemm arguments fit the lean epilogue (checked on the host before the 256 x 256 kernel is chosen)
inline bool gemm_epilogue_lean_ok(const vrd_gemm_args& a) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return a.M % 64 == 0 && a.N % 64 == 0 && (a.act == VRD_ACT_NONE || a.act == VRD_ACT_GELU) &&
           !(a.act == VRD_ACT_GELU && (a.row_mask || a.scale || a.res || a.res2)) && al16(a.bias) && al16(a.scale);
}

// SLAB_ROWS: rows of the wave's private staging slab.  64 (16 KiB per wave) lets both accumulator halves be
// transposed up front so their registers are dead for the rest of the epilogue; 32 for kernels with less LDS.
// mw / mw1: first row of the sub-tile's two 32-row blocks (mw1 < 0: the block behind the first; apart when the tile's blocks come
// from a block list: vrd_gemm_args.row_blocks)
template <bool STAGED, int SLAB_ROWS = 32, typename Acc>
__device__ __forceinline__ void gemm_epilogue(const vrd_gemm_args& p, const Acc (&acc)[2][2], float* smem, int64_t mw,
                                              int nw, int wave, int lane, unsigned* rflag = nullptr, int64_t mw1 = -1) {
    if (mw1 < 0) mw1 = mw + 32;
    auto rowbase = [&](int mi) { return mi ? mw1 : mw; };
    RangeTrack rt;
    // 16 x 16 accumulators: the four row groups of a wave write the same 16 columns of rows 4 apart -- one bank for all four at a
    // 64-float slab pitch --, so the slab's 16-column blocks are XOR-swizzled with (row >> 2) & 3 (as in gemm_epilogue_lean16);
    // the 32 x 32 layout's two half-rows stay as they are (two addresses per element instead of an immediate offset cost the
    // exact-f32 kernel more than the 2-way conflict)
    constexpr bool SWZ = std::is_same<Acc, acc32q>::value;
    if (STAGED) {
        // Through LDS: each wave transposes its sub-tile, 32 rows at a time, through a private 32 x 64 slab
        // so that global traffic is whole 256-B row segments as float4 (the raw accumulator layout would
        // give 64 scalar stores per lane).  The caller's main loop ended on a barrier, so the operand tiles
        // are dead; slabs are wave-private and the DS operations of one wave execute in order.
        // Each round is written as whole-array passes (row inputs, transpose, read-back, arithmetic, stores)
        // with the activation chosen once per pass, so the 8 rows of a lane are in flight together instead
        // of one load -> wait -> compute -> store chain per row.
        float* stg = smem + wave * (SLAB_ROWS * STG_PITCH);
        const int c4 = (lane & 15) * 4, rb0 = lane >> 4;
        const int n = nw + c4;
        const bool nfull = n + 3 < p.N;
        const float alpha = acc_alpha(p);
        float bias[4] = {0.f, 0.f, 0.f, 0.f}, scale[4] = {1.f, 1.f, 1.f, 1.f};
        if (nfull && ((reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.scale)) & 15) == 0) {
            if (p.bias) {
                const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                bias[0] = b.x, bias[1] = b.y, bias[2] = b.z, bias[3] = b.w;
            }
            if (p.scale) {
                const float4 b = *reinterpret_cast<const float4*>(p.scale + n);
                scale[0] = b.x, scale[1] = b.y, scale[2] = b.z, scale[3] = b.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) {
                    if (p.bias) bias[j] = p.bias[n + j];
                    if (p.scale) scale[j] = p.scale[n + j];
                }
        }
        const bool row_inputs = p.row_mask || p.res || p.res2;
        const bool ragged = __any(n < p.N && !nfull);          // wave-uniform
        // A 64 x 64 sub-tile is written in four passes of 16 rows (accumulator half mi, row group h): the row
        // inputs of the next pass are requested before the current one is worked on.
        struct RowIn {
            float mk[4];
            float4 r1[4], r2[4];
        };
        auto fetch = [&](int pass) {
            RowIn in;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                in.mk[j] = 1.f;
                in.r1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                in.r2[j] = in.r1[j];
            }
            if (!row_inputs) return in;
            const int64_t mr = rowbase(pass >> 1) + rb0 + (pass & 1) * 16;     // rows mr + 4*j
            if (!ragged) {
                // unpredicated: rows past M and column groups past N read a clamped, valid address and are
                // dropped at the store
                const int nc = n < p.N ? n : 0;
                unsigned char mb[4];
                int64_t mc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) mc[j] = (mr + 4 * j < p.M) ? mr + 4 * j : p.M - 1;
                if (p.row_mask) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mb[j] = p.row_mask[mc[j]];
                }
                if (p.res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) in.r1[j] = *reinterpret_cast<const float4*>(p.res + mc[j] * p.ldres + nc);
                }
                if (p.res2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) in.r2[j] = *reinterpret_cast<const float4*>(p.res2 + mc[j] * p.ldres2 + nc);
                }
                if (p.row_mask) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) in.mk[j] = (float)mb[j];
                }
                return in;
            }
            // ragged column group in this wave (N % 4 != 0): guarded element loads
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t m = mr + 4 * j;
                const bool ok = m < p.M && n < p.N;
                if (p.row_mask && ok) in.mk[j] = (float)p.row_mask[m];
                if (ok && nfull) {
                    if (p.res) in.r1[j] = *reinterpret_cast<const float4*>(p.res + m * p.ldres + n);
                    if (p.res2) in.r2[j] = *reinterpret_cast<const float4*>(p.res2 + m * p.ldres2 + n);
                } else if (ok) {
                    float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (n + c < p.N) {
                            if (p.res) a1[c] = p.res[m * p.ldres + n + c];
                            if (p.res2) a2[c] = p.res2[m * p.ldres2 + n + c];
                        }
                    in.r1[j] = make_float4(a1[0], a1[1], a1[2], a1[3]);
                    in.r2[j] = make_float4(a2[0], a2[1], a2[2], a2[3]);
                }
            }
            return in;
        };
        RowIn cur = fetch(0);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int mi = pass >> 1, h = pass & 1;
            const int slab_row0 = SLAB_ROWS == 64 ? mi * 32 : 0;
            if (SLAB_ROWS == 64 ? pass == 0 : h == 0) {
                // transpose: accumulator layout -> slab (32 rows x 64 columns per accumulator half)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    if (SLAB_ROWS == 64 ? false : mt != mi) continue;
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
                        forall32(acc[mt][nj], lane, [&](int r, int c, float x) {
                            stg[((SLAB_ROWS == 64 ? mt * 32 : 0) + r) * STG_PITCH + ((nj * 32 + c) ^ (SWZ ? ((r >> 2) & 3) << 4 : 0))] = x;
                        });
                }
            }
            RowIn nxt = cur;
            if (pass + 1 < 4) nxt = fetch(pass + 1);
            float v[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 t = *reinterpret_cast<const float4*>(stg + (slab_row0 + rb0 + h * 16 + 4 * j) * STG_PITCH + (c4 ^ (SWZ ? 16 * j : 0)));
                v[j][0] = fmaf(t.x, alpha, bias[0]), v[j][1] = fmaf(t.y, alpha, bias[1]);
                v[j][2] = fmaf(t.z, alpha, bias[2]), v[j][3] = fmaf(t.w, alpha, bias[3]);
            }
            // activation (one wave-uniform choice per pass), then mask * scale + residuals
            if (p.act == VRD_ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[j][c] = gelu_erf(v[j][c]);
            } else if (p.act == VRD_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[j][c] = fmaxf(v[j][c], 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rmk = p.res_masked ? cur.mk[j] : 1.f;
                const float a1[4] = {cur.r1[j].x, cur.r1[j].y, cur.r1[j].z, cur.r1[j].w};
                const float a2[4] = {cur.r2[j].x, cur.r2[j].y, cur.r2[j].z, cur.r2[j].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) v[j][c] = v[j][c] * cur.mk[j] * scale[c] + a1[c] * rmk + a2[c];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t m = rowbase(mi) + rb0 + h * 16 + 4 * j;
                if (m >= p.M || n >= p.N) continue;
                float* crow = p.C + m * p.ldc + n;
                if (p.c_pair) {          // pair rows of width N (host checks N % 8 == 0, so a float4 group is whole)
                    store_pair4(p.C + m * p.ldc, n, p.N, make_float4(v[j][0], v[j][1], v[j][2], v[j][3]), p.c_pair, &rt);
                } else if (nfull) {
                    *reinterpret_cast<float4*>(crow) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (n + c < p.N) crow[c] = v[j][c];
                }
            }
            cur = nxt;
        }
        if (p.c_pair == VRD_PAIR_F16) rt.report(rflag, RANGE_GEMM_OUT);
        return;
    }
    // fallback straight from the accumulator layout (rows that are not 16-byte aligned, e.g. ldc = 133)
    const float alpha = acc_alpha(p);
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            forall32(acc[mi][nj], lane, [&](int r, int c, float x) {
                const int n = nw + nj * 32 + c;
                const int64_t m = rowbase(mi) + r;
                if (n >= p.N || m >= p.M) return;
                const float bias = p.bias ? p.bias[n] : 0.f;
                const float scale = p.scale ? p.scale[n] : 1.f;
                const float mk = p.row_mask ? (float)p.row_mask[m] : 1.f;
                const float r1 = p.res ? p.res[m * p.ldres + n] : 0.f;
                const float r2 = p.res2 ? p.res2[m * p.ldres2 + n] : 0.f;
                const float v = epilogue_value(p, fmaf(x, alpha, bias), mk, scale, r1, p.res_masked ? mk : 1.f, r2);
                if (p.c_pair) store_pair1(p.C + m * p.ldc, n, p.N, v, p.c_pair, &rt);
                else p.C[m * p.ldc + n] = v;
            });
    if (p.c_pair == VRD_PAIR_F16) rt.report(rflag, RANGE_GEMM_OUT);
}

// One 32 x 32 block straight from its register layout: the epilogue of the 64 x 64-tile kernels, whose problems are small
// enough for its stores of short row pieces not to matter.
template <typename Acc>
__device__ __forceinline__ void gemm_epilogue_tile32(const vrd_gemm_args& p, const Acc& acc, int64_t mw, int nw, int lane,
                                                     unsigned* rflag = nullptr) {
    RangeTrack rt;
    const float alpha = acc_alpha(p);
    forall32(acc, lane, [&](int r, int c, float x) {
        const int n = nw + c;
        const int64_t m = mw + r;
        if (n >= p.N || m >= p.M) return;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float scale = p.scale ? p.scale[n] : 1.f;
        const float mk = p.row_mask ? (float)p.row_mask[m] : 1.f;
        const float r1 = p.res ? p.res[m * p.ldres + n] : 0.f;
        const float r2 = p.res2 ? p.res2[m * p.ldres2 + n] : 0.f;
        const float v = epilogue_value(p, fmaf(x, alpha, bias), mk, scale, r1, p.res_masked ? mk : 1.f, r2);
        if (p.c_pair) store_pair1(p.C + m * p.ldc, n, p.N, v, p.c_pair, &rt);
        else p.C[m * p.ldc + n] = v;
    });
    if (p.c_pair == VRD_PAIR_F16) rt.report(rflag, RANGE_GEMM_OUT);
}

}  // namespace vrd
