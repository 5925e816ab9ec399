// Split-precision variant of the conv GEMM: the same C = epilogue(A' . W^T) as vrd_gemm.hip, with
// every f32 product replaced by three 16-bit MFMA products (v_mfma_f32_16x16x32_bf16 / _f16, f32 accumulate; every
// split-precision GEMM kernel sums a K step of 32 per instruction and the three products of a step in the order lo x hi,
// hi x lo, hi x hi, so which kernel serves a shape does not change a bit of the result):
//
//     x = x_hi + x_lo,  x_hi = bf16(x),  x_lo = bf16(x - x_hi)            (|x - x_hi - x_lo| <= 2^-17 |x|)
//     a * w  ~=  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi                          (drops a_lo*w_lo ~ 2^-18 |a w|)
//
// Each bf16 x bf16 product is exact in the f32 accumulator, so the result carries ~17 significand bits
// per product (relative error ~1e-5 per GEMM) at 3/16 of the f32-MFMA instruction time.  Weights are
// split once on the host side of the ABI (W_split = [hi | lo], each N x K bf16).  Activations come either
// as f32 rows, split while being staged into LDS, or (APAIR) as "pair" rows already holding [hi | lo]
// 16-bit planes written by the producing kernel, staged as plain 16-byte copies.
// F16 (VRD_PAIR_F16, the f16x3 mode): the same with f16 planes of power-of-two scaled operands (vrd_common.h): 2^-22 per
// plane pair, ~2^-22 |a w| dropped; the epilogue multiplies the accumulator by the power of two that undoes the scaling.
//
// Tiling: 128 x 128 x 32 per 256-thread workgroup, four waves x (2 x 2) blocks of 32 x 32 (each 2 x 2 accumulators of 16 x 16), operand
// tiles [row][k] in bf16 with an 80-byte row pitch (16 consecutive rows hit 16 distinct 16-byte LDS slots,
// so the ds_read_b128 fragment reads are conflict free).  Two LDS buffers (80 KiB, two workgroups per CU)
// and a register prefetch two K steps ahead: while the MFMAs of step t run, the registers holding step
// t+1 are split and written to the other buffer and the global loads of step t+2 are issued; one barrier
// per step.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"
#include <cstdlib>

namespace {

using vrd::acc32q;
constexpr int BK = 32;
constexpr int XP = 40;                                   // row pitch in bf16 elements (80 B)
// SMALL: 64 x 64 tiles, one 32 x 32 accumulator per wave -- for problems whose 128 x 128 tiles would leave most of the
// chip idle (a training batch, the predictor's 9 queries per pair: 2.3 k rows x 512 columns are 72 tiles for 256 CUs, each a
// serial 16-step K loop of 24 MFMAs per wave): four times the workgroups, a quarter of the MFMA time per wave.  Same
// products in the same order per output element, so the same bits.
template <bool SMALL>
struct X3Geo {
    static constexpr int BM = SMALL ? 64 : 128, BN = BM;
    static constexpr int WT = BM / 2;                                    // rows / columns of a wave's sub-tile
    static constexpr int NT = WT / 32;                                   // 32 x 32 blocks per wave and dimension
    static constexpr int TILE = BM * XP;                                 // elements per operand tile
    static constexpr size_t LDS = 2 * 4 * TILE * 2;         // 2 buffers x (a_hi, a_lo, w_hi, w_lo): 80 / 40 KiB
    static constexpr int NPA = BM / 32, NPAP = BM / 64, NPW = BN / 64;   // staging pieces per thread: f32 A, pair A, W
};

template <int TAPS, bool STAGED, bool APAIR, bool SMALL = false, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_x3_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, unsigned* rflag) {
    using G = X3Geo<SMALL>;
    typedef typename vrd::SplitFmt<F16>::elem e16;          // the 16-bit element of this instantiation (bf16 or f16)
    typedef typename vrd::SplitFmt<F16>::x8 e16x8;
    typedef typename vrd::SplitFmt<F16>::x4 e16x4;
    constexpr int BM = G::BM, BN = G::BN, TILE = G::TILE, WT = G::WT, NT = G::NT;
    constexpr int NPA = G::NPA, NPAP = G::NPAP, NPW = G::NPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    e16* const lds = reinterpret_cast<e16*>(smem);      // buffer b: lds + b*4*TILE; tiles a_hi, a_lo, w_hi, w_lo
    vrd::RangeTrack rt_in;                              // f32 activation rows split while staged (vrd_common.h)
    // the factor the f32 rows are multiplied by before the f16 split: 2^VRD_F16_ACT_EXP, or the caller's (vrd_gemm_args.a_scale:
    // operands without a fixed range, i.e. gradients)
    const float amul = (F16 && !APAIR && p.a_scale) ? vrd::uniform_load(p.a_scale) : vrd::F16_ACT_SCALE;

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;      // v_mfma_f32_16x16x32 operand: row / column l15, k = 8 * l4 .. + 7
    const int K = p.Cin * TAPS;          // multiple of 32 (checked on the host)
    const int nkt = K / BK;
    const e16* Wsp = reinterpret_cast<const e16*>(p.W_split);      // [N][K/32][32 hi | 32 lo]

    // A staging: 4 float4 pieces per thread, piece i = (row = (tid + 256 i) / 8, k = 4 * ((tid + 256 i) % 8))
    // W staging: 2 x (hi, lo) 16-byte pieces per thread, piece i = (row = (tid + 256 i) / 4, k = 8 * (f % 4))
    int st[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int64_t r = m0 + ((tid + 256 * i) >> 3);
        st[i] = (TAPS == 3 && r < p.M) ? (int)(r % p.T) : 0;
    }
    float4 ra[4];                   // f32 A pieces (APAIR: reinterpreted as hi[2] | lo[2] 16-byte pieces)
    uint4 rwh[NPW], rwl[NPW];

    auto fetch = [&](int kt) {
        if (APAIR) {
            // piece i = (row = (tid + 256 i) / 4, 8 consecutive k): one 16-byte load from each plane
#pragma unroll
            for (int i = 0; i < NPAP; ++i) {
                const int f = tid + 256 * i;
                const int64_t r = m0 + (f >> 2);
                const int k = kt * BK + (f & 3) * 8;
                uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
                if (r < p.M) {
                    int tap = 0, ci = k;
                    bool ok = true;
                    if (TAPS == 3) {
                        tap = (k >= p.Cin) + (k >= 2 * p.Cin);
                        ci = k - tap * p.Cin;
                        const int tt = (int)(r % p.T) + tap - 1;
                        ok = tt >= 0 && tt < p.T;
                    }
                    if (ok) {
                        const e16* row = reinterpret_cast<const e16*>(p.A + (r + tap - (TAPS == 3 ? 1 : 0)) * p.lda) +
                                            vrd::pair_index(ci);
                        h = *reinterpret_cast<const uint4*>(row);
                        l = *reinterpret_cast<const uint4*>(row + 32);
                    }
                }
                ra[i] = *reinterpret_cast<float4*>(&h);
                ra[2 + i] = *reinterpret_cast<float4*>(&l);
            }
        } else
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int f = tid + 256 * i;
            const int64_t r = m0 + (f >> 3);
            const int k = kt * BK + (f & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < p.M) {
                int tap = 0, ci = k;
                bool ok = true;
                if (TAPS == 3) {
                    tap = (k >= p.Cin) + (k >= 2 * p.Cin);
                    ci = k - tap * p.Cin;
                    const int tt = st[i] + tap - 1;
                    ok = tt >= 0 && tt < p.T;
                }
                if (ok) v = *reinterpret_cast<const float4*>(p.A + (r + tap - (TAPS == 3 ? 1 : 0)) * p.lda + ci);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int f = tid + 256 * i;
            const int n = n0 + (f >> 2);
            const int k = kt * BK + (f & 3) * 8;
            uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
            if (n < p.N) {
                const e16* wrow = Wsp + (int64_t)n * K * 2 + vrd::pair_index(k);
                h = *reinterpret_cast<const uint4*>(wrow);
                l = *reinterpret_cast<const uint4*>(wrow + 32);
            }
            rwh[i] = h;
            rwl[i] = l;
        }
    };
    auto stage = [&](int buf) {
        e16* a_hi = lds + buf * 4 * TILE;
        e16* a_lo = a_hi + TILE;
        e16* w_hi = a_lo + TILE;
        e16* w_lo = w_hi + TILE;
        if (APAIR) {
#pragma unroll
            for (int i = 0; i < NPAP; ++i) {
                const int f = tid + 256 * i;
                const int off = (f >> 2) * XP + (f & 3) * 8;
                *reinterpret_cast<float4*>(a_hi + off) = ra[i];
                *reinterpret_cast<float4*>(a_lo + off) = ra[2 + i];
            }
        } else
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int f = tid + 256 * i;
            const int off = (f >> 3) * XP + (f & 7) * 4;
            const float x[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
            e16x4 h, l;
            vrd::split_n_scaled<F16>(x, amul, h, l, &rt_in);
            *reinterpret_cast<e16x4*>(a_hi + off) = h;
            *reinterpret_cast<e16x4*>(a_lo + off) = l;
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int f = tid + 256 * i;
            const int off = (f >> 2) * XP + (f & 3) * 8;
            *reinterpret_cast<uint4*>(w_hi + off) = rwh[i];
            *reinterpret_cast<uint4*>(w_lo + off) = rwl[i];
        }
    };

    acc32q acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) vrd::acc_clear(acc[i][j]);

    fetch(0);
    stage(0);
    if (nkt > 1) fetch(1);
    __syncthreads();
    const int arow = (wm * WT + l15) * XP + 8 * l4, wrow = (wn * WT + l15) * XP + 8 * l4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const e16* a_hi = lds + cur * 4 * TILE;
        const e16* a_lo = a_hi + TILE;
        const e16* w_hi = a_lo + TILE;
        const e16* w_lo = w_hi + TILE;
        // fragments of the whole K step: 16-row / 16-column blocks t of the wave's sub-tile
        e16x8 ah[2 * NT], al[2 * NT], wh[2 * NT], wl[2 * NT];
#pragma unroll
        for (int t = 0; t < 2 * NT; ++t) {
            ah[t] = *reinterpret_cast<const e16x8*>(a_hi + arow + t * 16 * XP);
            al[t] = *reinterpret_cast<const e16x8*>(a_lo + arow + t * 16 * XP);
            wh[t] = *reinterpret_cast<const e16x8*>(w_hi + wrow + t * 16 * XP);
            wl[t] = *reinterpret_cast<const e16x8*>(w_lo + wrow + t * 16 * XP);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // product-major inside a row block: consecutive MFMAs write different accumulators
#pragma unroll
            for (int bi = half * NT; bi < (half + 1) * NT; ++bi)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int bj = 0; bj < 2 * NT; ++bj) {
                        vrd::f32x4_t& c = acc[bi >> 1][bj >> 1].b[bi & 1][bj & 1];
                        c = vrd::mfma16(pr == 0 ? al[bi] : ah[bi], pr == 1 ? wl[bj] : wh[bj], c);
                    }
            if (half == 0 && kt + 1 < nkt) {
                // the registers hold step kt+1 (loaded one iteration ago): split + write them into the
                // other buffer under this step's MFMAs, then reuse them for the loads of step kt+2
                stage(cur ^ 1);
                if (kt + 2 < nkt) fetch(kt + 2);
            }
        }
        __syncthreads();
    }
    if (F16 && !APAIR) rt_in.report(rflag, vrd::RANGE_GEMM_IN);
    if constexpr (SMALL) vrd::gemm_epilogue_tile32(p, acc[0][0], m0 + wm * 32, n0 + wn * 32, lane, rflag);
    else vrd::gemm_epilogue<STAGED>(p, acc, smem, m0 + wm * 64, n0 + wn * 64, wave, lane, rflag);
}

}  // namespace

namespace vrd {

// called by vrd_gemm() after argument validation when W_split is given and the shape qualifies
template <int TAPS, bool STAGED, bool APAIR, bool SMALL = false>
static int launch_one(const vrd_gemm_args& a, int tiles_m, int tiles_n, hipStream_t s) {
    auto kern = a.split_fmt == VRD_PAIR_F16 ? gemm_x3_kernel<TAPS, STAGED, APAIR, SMALL, true> : gemm_x3_kernel<TAPS, STAGED, APAIR, SMALL, false>;
    constexpr size_t lds = X3Geo<SMALL>::LDS;
    if (int rc = reserve_lds(reinterpret_cast<const void*>(kern), lds, "vrd_gemm(bf16x3)")) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m, tiles_n, a.split_fmt == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    return 0;
}

template <int TAPS>
static int launch_taps(const vrd_gemm_args& a, bool staged, int tiles_m, int tiles_n, hipStream_t s, bool small) {
    const bool apair = a.a_pair_width > 0;
    if (small) return apair ? launch_one<TAPS, false, true, true>(a, tiles_m, tiles_n, s) : launch_one<TAPS, false, false, true>(a, tiles_m, tiles_n, s);
    if (staged) return apair ? launch_one<TAPS, true, true>(a, tiles_m, tiles_n, s) : launch_one<TAPS, true, false>(a, tiles_m, tiles_n, s);
    return apair ? launch_one<TAPS, false, true>(a, tiles_m, tiles_n, s) : launch_one<TAPS, false, false>(a, tiles_m, tiles_n, s);
}

// called by vrd_gemm() after argument validation when W_split is given and the shape qualifies
int launch_gemm_x3(const vrd_gemm_args& a, bool staged, hipStream_t s) {
    // 64 x 64 tiles while the 128 x 128 ones would not give every CU a workgroup (VRD_X3_SMALL=0: never)
    static const int small_env = [] { const char* e = getenv("VRD_X3_SMALL"); return e ? atoi(e) : 1; }();
    const int64_t tiles128 = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    const bool small = small_env && tiles128 < 256;
    const int bm = small ? 64 : 128;
    const int tiles_m = (int)((a.M + bm - 1) / bm), tiles_n = (a.N + bm - 1) / bm;
    return a.taps == 1 ? launch_taps<1>(a, staged, tiles_m, tiles_n, s, small) : launch_taps<3>(a, staged, tiles_m, tiles_n, s, small);
}

}  // namespace vrd
