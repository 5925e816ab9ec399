// Backward kernels of the relation-encoding path (training step, BASELINE config 3).
//
// The reference differentiates its ATen graph with autograd (train.py:186); here every forward kernel family gets
// the hand-written kernel(s) that compute its input / parameter gradients, bound through the same C ABI and wrapped
// as torch.autograd.Function in vrdone_amd/autograd.py.  Training batches are small (24 pairs x 96 frames = 2,304
// rows for vidvrd.yaml; 48 x 512 for vidor), so these kernels are written for correctness and sane memory access
// (coalesced rows, one wave per row, f32 MFMA for the one real contraction), not tuned like the forward path.
// Everything is f32; parameter gradients are ACCUMULATED (+=) into caller-zeroed buffers with float atomics where
// several workgroups contribute to the same element (summation order, hence the last bits, vary from run to run).
//
//  vrd_gemm_wgrad      dW of a dense conv (k = 1 / 3):  dW[n, tap*Cin+ci] += sum_r G[r,n] X[r+tap-1, ci]   (f32 MFMA)
//  vrd_colsum          out[c] += sum_r a[r,c] * b[s*r+shift, c*bc+bo] * mask[r] * rscale[r]: bias, drop-path-scale,
//                      depthwise-conv-weight gradients
//  vrd_rowcol_scale    out = v * colscale[c] * rowscale[r] * mask[r] + res * (mask) + res2: the affine drop-path
//                      residual of blocks.py:1074-1076,1148 in training form, and its input gradient
//  vrd_act / _bwd      GELU (erf) / ReLU and their derivative
//  vrd_layernorm_bwd   dx, dgamma, dbeta of the channel LayerNorm (+ReLU)
//  vrd_dwconv_bwd      input gradient of the depthwise conv (k 1/3, stride 1/2, 1 or 2 inputs per group, up to 3 sets,
//                      optional nearest-x2-upsample-add input)
//  vrd_local_attn_bwd  banded attention: dq, dk, dv
//  vrd_attn_bwd_probs  global attention: probabilities P and score gradients dS, row by row
//  vrd_bmm             strided batched matmul (dq = dS K, dk = dS^T Q, dv = P^T dO; mask head)
//  vrd_maxpool_bwd     MaxPool1d(3,2,1) * mask
#include "vrd_common.h"
#include <cstdlib>
#include <cmath>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

// ------------------------------------------------------------------------------------------------------------------
// dW[n, j] += sum_{r in chunk} G[r, n] * X[r + tap(j) - taps/2, ci(j)],  j = tap*Cin + ci, rows outside the length-T
// sequence of r contribute 0.  One wave per 64 x 64 tile of dW and per chunk of rows: v_mfma_f32_32x32x2_f32 takes
// the two operands of two rows straight from global memory (a lane holds G[r0 + (lane >> 5)][n0 + (lane & 31)] and
// X[r0 + (lane >> 5) + shift][ci]: 128-byte row segments), accumulates in f32 and adds its partial tile atomically.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_CHUNK = 128;      // rows per wave

__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X,
                                                    int64_t ldx, const uint8_t* __restrict__ row_mask, int64_t M, int N,
                                                    int Cin, int taps, int T, int tiles_k, float* __restrict__ dW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = Cin * taps;
    const int tile = blockIdx.x;
    // a wave owns a 64 x 64 tile of dW as 2 x 2 accumulators: two G values and two X values per row feed four MFMAs (with
    // one 32 x 32 tile per wave every MFMA needed its own two loads and the kernel ran at the rate of its L2 traffic)
    const int n0 = (tile / tiles_k) * 64, j0 = (tile % tiles_k) * 64;
    const int li = lane & 31, lh = lane >> 5;
    int shift[2];
    const float* xp[2];
    const float* gp[2];
    bool a_col[2], b_col[2];
    const int64_t r_begin = ((int64_t)blockIdx.y * 4 + wave) * WG_CHUNK;
    const int64_t r_end = r_begin + WG_CHUNK < M ? r_begin + WG_CHUNK : M;        // (empty for a wave beyond the last row)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + 32 * h + li, j = j0 + 32 * h + li;
        const int tap = j < K ? j / Cin : 0;
        const int ci = j < K ? j - tap * Cin : 0;
        shift[h] = tap - taps / 2;
        a_col[h] = n < N, b_col[h] = j < K;
        gp[h] = G + (r_begin + lh) * ldg + n;
        xp[h] = X + (r_begin + lh + shift[h]) * ldx + ci;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int hj = 0; hj < 2; ++hj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[hn][hj][e] = 0.f;
    // eight rows (four row pairs) per iteration, all sixteen loads requested before the first MFMA: with one row pair per
    // iteration the loop ran at the latency of its loads (~900 cycles per MFMA on a 4.6 k-row training batch).
    // The position t of a row in its sequence (k = 3: zero padding at the sequence ends) is carried along instead of a
    // 64-bit modulo per row.
    int t = taps == 3 ? (int)((r_begin + lh) % T) : 0;          // position of row r0 + lh
    const uint8_t* mp = row_mask ? row_mask + r_begin + lh : nullptr;
    // (the loads of iteration i + 1 are requested before the MFMAs of iteration i: a small batch leaves ~1 wave per SIMD,
    // nothing else would cover their latency)
    auto fetch = [&](int64_t r0, float (&a)[4][2], float (&b)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = r0 + 2 * u + lh < r_end;
            const bool live = in && (!mp || mp[2 * u]);
            int tt = 0;
            if (taps == 3) {
                tt = t + 2 * u;
                while (tt >= T) tt -= T;                        // (at most a few wraps: 8 rows, T >= 1)
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[u][h] = live && a_col[h] ? gp[h][(int64_t)(2 * u) * ldg] : 0.f;
                const bool ok = in && b_col[h] && (taps == 1 || (tt + shift[h] >= 0 && tt + shift[h] < T));
                b[u][h] = ok ? xp[h][(int64_t)(2 * u) * ldx] : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) gp[h] += 8 * ldg, xp[h] += 8 * ldx;
        if (mp) mp += 8;
        if (taps == 3) {
            t += 8;
            while (t >= T) t -= T;
        }
    };
    float a0[4][2], b0[4][2], a1[4][2], b1[4][2];
    if (r_begin < r_end) fetch(r_begin, a0, b0);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 16) {
        fetch(r0 + 8, a1, b1);                                  // (past r_end: zeros, no loads)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj) acc[hn][hj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][hn], b0[u][hj], acc[hn][hj], 0, 0, 0);
        fetch(r0 + 16, a0, b0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int hj = 0; hj < 2; ++hj) acc[hn][hj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][hn], b1[u][hj], acc[hn][hj], 0, 0, 0);
    }
    // the four waves of the block (four row chunks of the same tile) add up in LDS: one atomic per element and BLOCK, a
    // plain update when the block covers all rows
    __shared__ f32x16 red[3][4][64];
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = acc[q >> 1][q & 1];
    }
    __syncthreads();
    if (wave > 0) return;
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x16 v = acc[q >> 1][q & 1];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const f32x16 part = red[o][q][lane];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += part[e];
        }
        // element e of lane (li, lh): row (e & 3) + 8 * (e >> 2) + 4 * lh (n index), column li (j index)
        const int j = j0 + 32 * (q & 1) + li;
        if (j >= K) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int nn = n0 + 32 * (q >> 1) + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (nn < N) {
                if (single) dW[(int64_t)nn * K + j] += v[e];         // (dW holds the caller's initial value: zeros)
                else atomicAdd(dW + (int64_t)nn * K + j, v[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same contraction in the forward path's split precision (bf16x3): every product as g_lo x_hi + g_hi x_lo + g_hi x_hi on
// v_mfma_f32_32x32x16_bf16 (16 rows per MFMA instead of 2, at 8x the f32 MFMA rate: ~5x for three products), f32
// accumulate.  Same decomposition: a wave owns a 64 x 64 tile of dW for a chunk of rows and takes its operands straight
// from global memory in MFMA layout -- lane (m = lane & 31, half = lane >> 5) holds rows r0 + 8 half .. + 7 of column m:
// eight dword loads, each a 128-byte row segment per half-wave -- splits them into bf16 hi / lo in registers (24 vector
// instructions per eight values; four fragments feed twelve MFMAs) and accumulates.  The four waves of a block reduce in
// LDS as above.  Used in the bf16x3 mode only (vrd_gemm_wgrad_x3); gradients then carry ~2^-17 relative product error like
// the forward pass, the f32 mode keeps exact products.
// ------------------------------------------------------------------------------------------------------------------
// F16 (the f16x3 mode's backward): both operands as f16 planes of power-of-two multiples -- the gradient rows times gscale[0]
// (vrd_absmax_scale: max |g| lands in [2^13, 2^14)), the activation rows times 2^VRD_F16_ACT_EXP as in the forward pass (they went
// through a forward GEMM's range check) -- and the accumulators times gscale[1] * 2^-VRD_F16_ACT_EXP on the way out: ~22-bit
// products at the cost of the bf16 split's ~17.
using bf16x8 = vrd::bf16x8_t;
template <bool F16>
struct WFragT { typename vrd::SplitFmt<F16>::x8 h, l; };
template <bool F16>
__device__ __forceinline__ WFragT<F16> wsplit8(const float (&v)[8], float mul) {
    typedef typename vrd::SplitFmt<F16>::elem E;
    WFragT<F16> f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float y = F16 ? v[i] * mul : v[i];
        const E h = (E)y;
        f.h[i] = h;
        f.l[i] = (E)(y - (float)h);
    }
    return f;
}

template <bool F16>
__global__ __launch_bounds__(256) void wgrad_x3_kernel(const float* __restrict__ G, int64_t ldg, const float* __restrict__ X,
                                                       int64_t ldx, const uint8_t* __restrict__ row_mask, int64_t M, int N,
                                                       int Cin, int taps, int T, int tiles_k, int chunk, float* __restrict__ dW,
                                                       const float* __restrict__ gscale) {
    const float gmul = F16 ? vrd::uniform_load(gscale) : 1.f;
    const float unscale = F16 ? vrd::uniform_load(gscale + 1) * vrd::F16_ACT_INV : 1.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = Cin * taps;
    const int tile = blockIdx.x;
    const int n0 = (tile / tiles_k) * 64, j0 = (tile % tiles_k) * 64;
    const int li = lane & 31, lh = lane >> 5;
    int shift[2];
    const float* xp[2];
    const float* gp[2];
    bool a_col[2], b_col[2];
    const int64_t r_begin = ((int64_t)blockIdx.y * 4 + wave) * chunk;
    const int64_t r_end = r_begin + chunk < M ? r_begin + chunk : M;              // (empty for a wave beyond the last row)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + 32 * h + li, j = j0 + 32 * h + li;
        const int tap = j < K ? j / Cin : 0;
        const int ci = j < K ? j - tap * Cin : 0;
        shift[h] = tap - taps / 2;
        a_col[h] = n < N, b_col[h] = j < K;
        gp[h] = G + (r_begin + 8 * lh) * ldg + n;
        xp[h] = X + (r_begin + 8 * lh + shift[h]) * ldx + ci;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int hj = 0; hj < 2; ++hj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[hn][hj][e] = 0.f;
    int t = taps == 3 ? (int)((r_begin + 8 * lh) % T) : 0;      // position of this lane's first row in its sequence
    const uint8_t* mp = row_mask ? row_mask + r_begin + 8 * lh : nullptr;
    // sixteen rows per step; the 32 loads of step i + 1 are requested before the MFMAs of step i
    auto fetch = [&](int64_t r0, float (&a)[2][8], float (&b)[2][8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool in = r0 + 8 * lh + i < r_end;
            const bool live = in && (!mp || mp[i]);
            int tt = 0;
            if (taps == 3) {
                tt = t + i;
                while (tt >= T) tt -= T;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[h][i] = live && a_col[h] ? gp[h][(int64_t)i * ldg] : 0.f;
                const bool ok = in && b_col[h] && (taps == 1 || (tt + shift[h] >= 0 && tt + shift[h] < T));
                b[h][i] = ok ? xp[h][(int64_t)i * ldx] : 0.f;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) gp[h] += 16 * ldg, xp[h] += 16 * ldx;
        if (mp) mp += 16;
        if (taps == 3) {
            t += 16;
            while (t >= T) t -= T;
        }
    };
    auto mac = [&](const float (&a)[2][8], const float (&b)[2][8]) {
        const WFragT<F16> fa0 = wsplit8<F16>(a[0], gmul), fa1 = wsplit8<F16>(a[1], gmul);
        const WFragT<F16> fb0 = wsplit8<F16>(b[0], vrd::F16_ACT_SCALE), fb1 = wsplit8<F16>(b[1], vrd::F16_ACT_SCALE);
#pragma unroll
        for (int hn = 0; hn < 2; ++hn)
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
                const WFragT<F16>& fa = hn ? fa1 : fa0;
                const WFragT<F16>& fb = hj ? fb1 : fb0;
                acc[hn][hj] = vrd::mfma32(fa.l, fb.h, acc[hn][hj]);
                acc[hn][hj] = vrd::mfma32(fa.h, fb.l, acc[hn][hj]);
                acc[hn][hj] = vrd::mfma32(fa.h, fb.h, acc[hn][hj]);
            }
    };
    float a0[2][8], b0[2][8], a1[2][8], b1[2][8];
    if (r_begin < r_end) fetch(r_begin, a0, b0);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += 32) {
        fetch(r0 + 16, a1, b1);                                 // (past r_end: zeros, no loads)
        mac(a0, b0);
        fetch(r0 + 32, a0, b0);
        mac(a1, b1);
    }
    __shared__ f32x16 red[3][4][64];
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = acc[q >> 1][q & 1];
    }
    __syncthreads();
    if (wave > 0) return;
    const bool single = gridDim.y == 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x16 v = acc[q >> 1][q & 1];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const f32x16 part = red[o][q][lane];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] += part[e];
        }
        const int j = j0 + 32 * (q & 1) + li;
        if (j >= K) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int nn = n0 + 32 * (q >> 1) + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (nn < N) {
                if (single) dW[(int64_t)nn * K + j] += v[e] * unscale;
                else atomicAdd(dW + (int64_t)nn * K + j, v[e] * unscale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The split-precision weight gradient with operand reuse: the wave-per-tile kernel above moves 8 KiB from L2 per twelve
// MFMAs (16 FLOP per byte: measured 150 TFLOP/s, the rate of its L2 traffic).  Here a block owns a TILE x TILE tile of dW for
// a chunk of rows; per step of 32 rows its threads load the 32 x TILE slabs of G and X once (four float4 each), split them
// into bf16 hi / lo and write four row-major planes into LDS (two stages, one barrier per step); each wave then forms its
// 64 x SJ sub-tile, taking the transposed fragments the MFMA wants -- eight consecutive ROWS of one column per lane --
// with ds_read_b64_tr_b16 (the read the attention kernels use for V^T; 16-byte chunks swizzled by row & 3).
//   BIG = 0: TILE 128, 4 waves x (64 x 64), two blocks per CU -- every input of >= 256 rows
//   BIG = 1: TILE 256, 8 waves x (64 x 128), one block per CU -- inputs with enough rows per chunk.  The kernel is bound by
//            its vector instructions, not by its MFMAs: the split costs ~3 instructions per element and wave, ~150 per wave
//            and step whatever the tile, against 24 MFMAs per wave and step at TILE 128 and 48 at TILE 256.
// float4 loads when N, Cin, ldg, ldx are multiples of 4 and the operands 16-byte aligned (a group of four columns then never
// straddles a tap), scalar loads otherwise; the wave kernel above serves inputs of fewer than 256 rows.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WL_ROWS = 32;                 // rows per step

template <int BIG>
struct WlGeo {
    static constexpr int TILE = BIG ? 256 : 128;
    static constexpr int NTHR = BIG ? 512 : 256;
    static constexpr int CGS = TILE / 4;                // column groups (four floats) per slab row; NTHR / CGS = 8 rows per pass
    static constexpr int ROWB = TILE * 2;               // bytes per plane row
    static constexpr int PLANE = WL_ROWS * ROWB;
    static constexpr int STAGE = 4 * PLANE;             // g_hi | g_lo | x_hi | x_lo: 32 / 64 KiB
    static constexpr int SJ = BIG ? 128 : 64;           // a wave's sub-tile: 64 (n) x SJ (j)
};

typedef __attribute__((ext_vector_type(4))) short wl_s16x4;
typedef __attribute__((address_space(3))) wl_s16x4* wl_lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) short wl_s16x8;

template <int BIG, bool VEC, int TAPS, bool F16>
__global__ __launch_bounds__(WlGeo<BIG>::NTHR, BIG ? 1 : 2) void wgrad_x3_lds_kernel(
    const float* __restrict__ G, int64_t ldg, const float* __restrict__ X, int64_t ldx, const uint8_t* __restrict__ row_mask, int64_t M,
    int N, int Cin, int T, int tiles_k, int chunk, float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ partial,
    const float* __restrict__ gscale) {
    using Geo = WlGeo<BIG>;
    typedef typename vrd::SplitFmt<F16>::x4 e16x4;
    typedef typename vrd::SplitFmt<F16>::elem e16;
    const float gmul = F16 ? vrd::uniform_load(gscale) : 1.f;                                        // (see wgrad_x3_kernel)
    const float unscale = F16 ? vrd::uniform_load(gscale + 1) * vrd::F16_ACT_INV : 1.f;
    constexpr int TILE = Geo::TILE, ROWB = Geo::ROWB, PLANE = Geo::PLANE, STAGE = Geo::STAGE, SJ = Geo::SJ, NJ = SJ / 32;
    extern __shared__ __attribute__((aligned(16))) char lds[];          // 2 * STAGE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = Cin * TAPS;
    // XCD-aware renumbering (as in the GEMM kernels): workgroups are dealt round-robin to the eight XCDs; a contiguous range of
    // (chunk, tile) ids per XCD keeps all tiles of a chunk -- which share its G and X slabs -- on one L2
    const int tiles = (int)gridDim.x, nwg = tiles * (int)gridDim.y, bid = (int)blockIdx.x + tiles * (int)blockIdx.y;
    const int xcd = bid & 7, xq = nwg >> 3, xrem = nwg & 7;
    const int lid = (xcd < xrem ? xcd * (xq + 1) : xrem * (xq + 1) + (xcd - xrem) * xq) + (bid >> 3);
    const int tile = lid % tiles, chunk_id = lid / tiles;
    const int n0 = (tile / tiles_k) * TILE, j0 = (tile % tiles_k) * TILE;
    const int64_t r_begin = (int64_t)chunk_id * chunk;
    const int64_t r_end = r_begin + chunk < M ? r_begin + chunk : M;
    // ---- loader: thread -> column group cg (4 columns) of both slabs, rows lr, lr + 8, lr + 16, lr + 24 of the step (BIG: a
    // wave is one slab row, so the row arithmetic is scalar).
    // VEC: float4 loads; otherwise four scalar loads per group, each column with its own tap.
    // Every fetch issues the same loads whatever the step looks like: rows past the end of the input and taps outside the
    // row's sequence read a valid address and are zeroed when the slab is written to LDS, and the row mask bytes come with the
    // slab instead of deciding its loads.  (The compiler counts vmcnt per path: with a path without loads, or a mask byte that
    // had to arrive before the row's load was issued, every wait was a vmcnt(0).)
    const int cg = tid % Geo::CGS;
    const int lr = BIG ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid / Geo::CGS;
    const int gn = n0 + 4 * cg, xj = j0 + 4 * cg;
    int g_ok[4], x_ok[4], shift[4], gcol[4];
    int64_t xoff[4];                             // element offset of column e's source inside row r of X: shift * ldx + ci
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        g_ok[e] = gn + e < N;
        x_ok[e] = xj + e < K;
        gcol[e] = g_ok[e] ? gn + e : 0;
        const int tap = x_ok[e] ? (xj + e) / Cin : 0;
        const int ci = x_ok[e] ? (xj + e) - tap * Cin : 0;
        shift[e] = tap - TAPS / 2;
        xoff[e] = (int64_t)shift[e] * ldx + ci;
    }
    const int step_t = WL_ROWS % T;              // what 32 rows add to a row's position in its sequence (k = 3)
    int tt[4] = {0, 0, 0, 0};                    // positions of the thread's four rows of the next fetch
    if (TAPS == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tt[i] = (int)((r_begin + lr + 8 * i) % T);
    }
    struct Slab {
        float4 g[4], x[4];
        int mk[4];               // row mask bytes
        int sq;                  // k = 3, bit 4 i + e: column e's tap of row i lies inside the row's sequence
    };
    const uint8_t* mbytes = row_mask ? row_mask : reinterpret_cast<const uint8_t*>(G);      // (any M readable bytes)
    const int64_t last = M - 1;
    int64_t frow = r_begin + lr;                 // the thread's first row of the next fetch
    // bias gradient on the way (the blocks of the first tile column only): column sums of the masked G slab this thread loads
    const bool do_bias = dbias != nullptr && tile % tiles_k == 0;                     // (block-uniform)
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](Slab& sl) {
        sl.sq = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t rr = frow + 8 * i;
            const int64_t rc = rr < last ? rr : last;
            sl.mk[i] = mbytes[rc];
            const float* gr = G + rc * ldg;
            const float* xr = X + rc * ldx;
            int sq[4] = {1, 1, 1, 1};
            if (TAPS == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sq[e] = (unsigned)(tt[i] + shift[e]) < (unsigned)T && rr <= last;      // (a row past the input has no taps:
                                                                                           // row `last` + 1 is not memory)
                    sl.sq |= sq[e] << (4 * i + e);
                }
                const int nt = tt[i] + step_t;
                tt[i] = nt >= T ? nt - T : nt;
            }
            if (VEC) {
                sl.g[i] = ld4(gr + gcol[0]);
                sl.x[i] = ld4(xr + (x_ok[0] & sq[0] ? xoff[0] : 0));
            } else {
                sl.g[i].x = gr[gcol[0]];
                sl.g[i].y = gr[gcol[1]];
                sl.g[i].z = gr[gcol[2]];
                sl.g[i].w = gr[gcol[3]];
                sl.x[i].x = xr[x_ok[0] & sq[0] ? xoff[0] : 0];
                sl.x[i].y = xr[x_ok[1] & sq[1] ? xoff[1] : 0];
                sl.x[i].z = xr[x_ok[2] & sq[2] ? xoff[2] : 0];
                sl.x[i].w = xr[x_ok[3] & sq[3] ? xoff[3] : 0];
            }
        }
        frow += WL_ROWS;
    };
    auto put = [&](char* plane_hi, int row, float4 v, float mul) {
        if (F16) v.x *= mul, v.y *= mul, v.z *= mul, v.w *= mul;
        const e16x4 h = {(e16)v.x, (e16)v.y, (e16)v.z, (e16)v.w};
        const e16x4 l = {(e16)(v.x - (float)h[0]), (e16)(v.y - (float)h[1]), (e16)(v.z - (float)h[2]), (e16)(v.w - (float)h[3])};
        const int off = row * ROWB + (((cg >> 1) ^ ((row & 3) << 2)) * 16) + (cg & 1) * 8;
        *reinterpret_cast<e16x4*>(plane_hi + off) = h;
        *reinterpret_cast<e16x4*>(plane_hi + PLANE + off) = l;
    };
    // (a row contributes G[r, n] X[r', j]: zeroing G's row takes care of masked rows and of rows past the chunk -- their X values
    // are real rows of the input --, and columns of X at or beyond K only reach entries of dW that are never stored; what X
    // needs is the zero of a tap outside the row's sequence, and without float4 groups the zero of a column beyond K next to
    // valid ones is cheap enough to keep)
    auto store = [&](char* st, const Slab& sl, int64_t r0) {       // r0: the slab's first row
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int in = r0 + lr + 8 * i < r_end;
            const int live = in & (row_mask ? sl.mk[i] != 0 : 1);
            float4 g = sl.g[i], x = sl.x[i];
            g.x = live & g_ok[0] ? g.x : 0.f;
            g.y = live & g_ok[VEC ? 0 : 1] ? g.y : 0.f;
            g.z = live & g_ok[VEC ? 0 : 2] ? g.z : 0.f;
            g.w = live & g_ok[VEC ? 0 : 3] ? g.w : 0.f;
            if (TAPS == 3 || !VEC) {
                const int sq = TAPS == 3 ? sl.sq >> (4 * i) : 15;
                x.x = x_ok[0] & sq ? x.x : 0.f;
                x.y = x_ok[VEC ? 0 : 1] & (sq >> (VEC ? 0 : 1)) ? x.y : 0.f;
                x.z = x_ok[VEC ? 0 : 2] & (sq >> (VEC ? 0 : 2)) ? x.z : 0.f;
                x.w = x_ok[VEC ? 0 : 3] & (sq >> (VEC ? 0 : 3)) ? x.w : 0.f;
            }
            put(st, lr + 8 * i, g, gmul);
            put(st + 2 * PLANE, lr + 8 * i, x, vrd::F16_ACT_SCALE);
            if (do_bias) bsum.x += g.x, bsum.y += g.y, bsum.z += g.z, bsum.w += g.w;
        }
    };
    // ---- compute: wave (wn, wj) owns the 64 x SJ sub-tile at (64 wn, SJ wj) as 2 x NJ accumulators
    const int wn = wave >> 1, wj = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int vq = (lane >> 2) & 3, vp = lane & 3, vcol0 = 16 * ((lane >> 4) & 1) + 4 * vp;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int hn = 0; hn < 2; ++hn)
#pragma unroll
        for (int hj = 0; hj < NJ; ++hj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[hn][hj][e] = 0.f;
    // fragment (k16 step s, 32 columns from colbase) of the plane pair at `pl`: lane (column lane & 31, half) <- rows 16 s + 8 half .. + 7
    auto frag = [&](const char* pl, int s, int colbase) {
        WFragT<F16> f;
        wl_s16x8 rh, rl;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const int row = 16 * s + 8 * part + 4 * lh + vq;
            const int col = colbase + vcol0;
            const int off = row * ROWB + ((((col * 2) >> 4) ^ ((row & 3) << 2)) * 16) + ((col * 2) & 15);
            const wl_s16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wl_lds_s16x4_ptr)(pl + off));
            const wl_s16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wl_lds_s16x4_ptr)(pl + PLANE + off));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rh[4 * part + q] = th[q];
                rl[4 * part + q] = tl[q];
            }
        }
        f.h = __builtin_bit_cast(typename vrd::SplitFmt<F16>::x8, rh);
        f.l = __builtin_bit_cast(typename vrd::SplitFmt<F16>::x8, rl);
        return f;
    };
    auto compute = [&](const char* st) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            WFragT<F16> fa[2], fb[NJ];
#pragma unroll
            for (int hn = 0; hn < 2; ++hn) fa[hn] = frag(st, s, 64 * wn + 32 * hn);
#pragma unroll
            for (int hj = 0; hj < NJ; ++hj) fb[hj] = frag(st + 2 * PLANE, s, SJ * wj + 32 * hj);
#pragma unroll
            for (int hn = 0; hn < 2; ++hn)
#pragma unroll
                for (int hj = 0; hj < NJ; ++hj) {
                    acc[hn][hj] = vrd::mfma32(fa[hn].l, fb[hj].h, acc[hn][hj]);
                    acc[hn][hj] = vrd::mfma32(fa[hn].h, fb[hj].l, acc[hn][hj]);
                    acc[hn][hj] = vrd::mfma32(fa[hn].h, fb[hj].h, acc[hn][hj]);
                }
        }
    };
    // ---- rows: the loads of the next step(s) are in flight under the MFMAs of step i; one barrier per step (a stage is rewritten
    // only after every wave has passed the barrier behind its last read)
    // (two steps in flight where the registers allow it: 64 accumulators, float4 loads, no tap bookkeeping)
    constexpr int DEPTH = (!BIG && VEC && TAPS == 1) ? 2 : 1;
    if (DEPTH == 2) {
        // (even steps live in stage 0 and slab `sa`, odd ones in stage 1 and `sb`; a fetch past the chunk's end loads rows of the
        // input again and is never stored)
        Slab sa, sb;
        fetch(sa);
        fetch(sb);
        store(lds, sa, r_begin);
        __syncthreads();
        for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * WL_ROWS) {
            fetch(sa);                                   // step r0 + 64
            compute(lds);
            if (r0 + WL_ROWS < r_end) store(lds + STAGE, sb, r0 + WL_ROWS);
            __syncthreads();
            if (r0 + WL_ROWS >= r_end) break;
            fetch(sb);                                   // step r0 + 96
            compute(lds + STAGE);
            if (r0 + 2 * WL_ROWS < r_end) store(lds, sa, r0 + 2 * WL_ROWS);
            __syncthreads();
        }
    } else {
        // One slab: written to the other stage during the step before its own, and fetched again right behind that.  The MFMAs of
        // a step and the split of the next step's slab do not depend on each other, and measured alone each is about as long as
        // the other (and as the loads): run one after the other by every wave -- the barrier puts all waves of a block into the
        // same phase -- the step cost their sum.  BIG: waves w and w + 4 share a SIMD; the first four take compute -> split,
        // the other four split -> compute, so that a SIMD has one wave on the MFMA pipe and one on the vector ALU all along.
        Slab sa;
        fetch(sa);
        store(lds, sa, r_begin);
        fetch(sa);                                       // step 1
        __syncthreads();
        const bool split_first = BIG && wave >= 4;
        int cur = 0;
        for (int64_t r0 = r_begin; r0 < r_end; r0 += WL_ROWS) {
            const bool more = r0 + WL_ROWS < r_end;
            if (split_first) {
                if (more) {
                    store(lds + (cur ^ 1) * STAGE, sa, r0 + WL_ROWS);
                    fetch(sa);                           // step r0 + 64
                }
                compute(lds + cur * STAGE);
            } else {
                compute(lds + cur * STAGE);
                if (more) {
                    store(lds + (cur ^ 1) * STAGE, sa, r0 + WL_ROWS);
                    fetch(sa);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    if (do_bias) {       // the eight row threads of a column add up in LDS (free behind the loop's last barrier): one atomic per
                         // column and block -- same-address atomics queue up one behind the other in L2
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<float4*>(red + lr * TILE + 4 * cg) = bsum;
        __syncthreads();
        if (tid < TILE && n0 + tid < N) {     // (columns beyond N carry zeros)
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * TILE + tid];
            atomicAdd(dbias + n0 + tid, t);
        }
    }
    // the chunk's share of the tile: straight into dW when there is one chunk; otherwise as plain stores into the chunk's slice of
    // `partial` (summed by wgrad_reduce_kernel: every (chunk, n, j) is written by exactly one block), or -- without a scratch
    // buffer -- as atomics.  L2 works float atomics off at about one element per clock and channel: the 16 k atomics of each of
    // ~512 blocks were 12-17 us of a 70 us launch at M = 24,576, N = K = 512, whatever the number of rows.
    const bool single = gridDim.y == 1;
    float* dst = single ? dW : partial ? partial + (int64_t)chunk_id * N * K : dW;
#pragma unroll
    for (int q = 0; q < 2 * NJ; ++q) {
        const int hn = q / NJ, hj = q % NJ;
        const int j = j0 + SJ * wj + 32 * hj + li;
        if (j >= K) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int nn = n0 + 64 * wn + 32 * hn + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (nn < N) {
                if (single) dst[(int64_t)nn * K + j] += acc[hn][hj][e] * unscale;
                else if (partial) dst[(int64_t)nn * K + j] = acc[hn][hj][e] * unscale;
                else atomicAdd(dst + (int64_t)nn * K + j, acc[hn][hj][e] * unscale);
            }
        }
    }
}

// dW[i] += sum_c partial[c * NK + i]: the row chunks' partial tiles of wgrad_x3_lds_kernel, in chunk order (so the sum does not
// depend on the order the blocks ran in, as the atomics' did).  float4 per thread when NK % 4 == 0.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, int64_t NK,
                                                           float* __restrict__ dW) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= NK) return;
    if ((NK & 3) == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int c = 0;
        for (; c + 4 <= chunks; c += 4) {       // four loads in flight
            const float4 a = ld4(partial + (int64_t)c * NK + i), b = ld4(partial + (int64_t)(c + 1) * NK + i);
            const float4 d = ld4(partial + (int64_t)(c + 2) * NK + i), e = ld4(partial + (int64_t)(c + 3) * NK + i);
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
            s.x += d.x, s.y += d.y, s.z += d.z, s.w += d.w;
            s.x += e.x, s.y += e.y, s.z += e.z, s.w += e.w;
        }
        for (; c < chunks; ++c) {
            const float4 a = ld4(partial + (int64_t)c * NK + i);
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
        float4 o = ld4(dW + i);
        o.x += s.x, o.y += s.y, o.z += s.z, o.w += s.w;
        *reinterpret_cast<float4*>(dW + i) = o;
    } else {
        for (int64_t k = i; k < i + 4 && k < NK; ++k) {
            float s = 0.f;
            for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * NK + k];
            dW[k] += s;
        }
    }
}

constexpr int WL_BIG_ROWS = 256;     // rows per block from which the 256 x 256 tiles pay

// one launch of wgrad_x3_lds_kernel<BIG, ...> (+ the reduction of its partial tiles): as few row chunks as still give
// 2 (BIG: 1) blocks per CU
template <int BIG>
int launch_wgrad_lds(const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* row_mask, int64_t M, int N, int Cin,
                     int taps, int T, float* dW, float* dbias, float* scratch, int64_t scratch_floats, bool vec, int n_cu, hipStream_t s,
                     const float* gscale) {
    using Geo = WlGeo<BIG>;
    const int K = Cin * taps;
    const int tiles_n = (N + Geo::TILE - 1) / Geo::TILE, tiles_k = (K + Geo::TILE - 1) / Geo::TILE;
    const int64_t tiles = (int64_t)tiles_n * tiles_k;
    int64_t want = ((BIG ? 1 : 2) * (int64_t)n_cu + tiles - 1) / tiles;
    if (want < 1) want = 1;
    int64_t chunk = (M + want - 1) / want;
    chunk = (chunk + WL_ROWS - 1) / WL_ROWS * WL_ROWS;
    if (chunk < 4 * WL_ROWS) chunk = 4 * WL_ROWS;
    const int64_t chunks = (M + chunk - 1) / chunk;
    VRD_CHECK_ARG(chunks <= 65535 && chunk < (1ll << 30) && tiles < (1 << 20), "vrd_gemm_wgrad_x3: too many rows (%lld)", (long long)M);
    const dim3 grid((unsigned)tiles, (unsigned)chunks);
    // the chunks' partial tiles go through `scratch` when it holds them (chunks x N x K floats, 16-byte aligned); else atomics
    const int64_t NK = (int64_t)N * K;
    float* partial = chunks > 1 && scratch && aligned16(scratch) && aligned16(dW) && scratch_floats >= chunks * NK ? scratch : nullptr;
    constexpr size_t lds = 2 * Geo::STAGE;
#define VRD_WGRAD_LAUNCH(VEC_, TAPS_, F16_)                                                                                        \
    do {                                                                                                                           \
        auto kern = wgrad_x3_lds_kernel<BIG, VEC_, TAPS_, F16_>;                                                                   \
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(kern), lds, "vrd_gemm_wgrad_x3")) return rc;                   \
        hipLaunchKernelGGL(kern, grid, dim3(Geo::NTHR), lds, s, G, ldg, X, ldx, row_mask, M, N, Cin, T, tiles_k, (int)chunk, dW,   \
                           dbias, partial, gscale);                                                                                \
    } while (0)
    if (gscale) {
        if (vec && taps == 1) VRD_WGRAD_LAUNCH(true, 1, true);
        else if (vec) VRD_WGRAD_LAUNCH(true, 3, true);
        else if (taps == 1) VRD_WGRAD_LAUNCH(false, 1, true);
        else VRD_WGRAD_LAUNCH(false, 3, true);
    } else if (vec && taps == 1) VRD_WGRAD_LAUNCH(true, 1, false);
    else if (vec) VRD_WGRAD_LAUNCH(true, 3, false);
    else if (taps == 1) VRD_WGRAD_LAUNCH(false, 1, false);
    else VRD_WGRAD_LAUNCH(false, 3, false);
#undef VRD_WGRAD_LAUNCH
    VRD_LAUNCH_CHECK();
    if (partial) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((NK + 1023) / 1024)), dim3(256), 0, s, partial, (int)chunks, NK, dW);
        VRD_LAUNCH_CHECK();
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// out[c] += sum_r a[r, c] * (b ? b[brow(r), c * bc + bo] : 1) * (mask ? mask[r] : 1) * (rscale ? rscale[r] : 1)
// brow(r): r = s * T + t  ->  s * (bs * T) + bs * t + shift, contributing only if 0 <= bs * t + shift < bs * T.
// block = rpb rows x 64 columns, a quarter of the rows per wave, one atomic per column and block: atomics on one address queue up
// behind each other in L2 (~50 ns each), and with 32 rows per atomic a 49 k-row input put 1,536 of them on every output element.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CS_ROWS = 32;      // smallest rows-per-block
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                     int64_t ldb, int bc, int bo, int bs, int shift, int T,
                                                     const uint8_t* __restrict__ mask, const float* __restrict__ rscale,
                                                     int64_t rows, int C, int rpb, float* __restrict__ out) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool col_ok = c < C;
    const int wrows = rpb / 4;                   // (rpb is a multiple of 32)
    const int64_t r0 = (int64_t)blockIdx.x * rpb + (int64_t)wave * wrows;
    const int64_t r1 = r0 + wrows < rows ? r0 + wrows : rows;
    // eight rows per iteration with their loads requested together (one row per iteration ran at the latency of its loads);
    // the row's (sequence, position) pair is carried along instead of a 64-bit division per row
    int64_t seq = b ? r0 / T : 0;
    int tpos = b ? (int)(r0 - seq * T) : 0;
    float s = 0.f;
    for (int64_t rb = r0; rb < r1; rb += 8) {
        float av[8], bv[8], fv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t r = rb + u;
            bool live = col_ok && r < r1 && (!mask || mask[r]);
            av[u] = 0.f, bv[u] = 1.f, fv[u] = 1.f;
            if (b) {
                int tq = tpos + u;
                int64_t sq = seq;
                while (tq >= T) tq -= T, ++sq;
                const int tb = bs * tq + shift;
                live = live && tb >= 0 && tb < bs * T;
                if (live) bv[u] = b[(sq * (int64_t)bs * T + tb) * ldb + (int64_t)c * bc + bo];
            }
            if (live) {
                av[u] = a[r * lda + c];
                if (rscale) fv[u] = rscale[r];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(av[u] * fv[u], bv[u], s);
        if (b) {
            tpos += 8;
            while (tpos >= T) tpos -= T, ++seq;
        }
    }
    // the four waves (four row ranges of the same 64 columns) add up in LDS: one atomic per column and block
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && col_ok) atomicAdd(out + c, s + red[0][lane] + red[1][lane] + red[2][lane]);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight and bias gradient of a depthwise conv (k = 1 / 3, stride bs, gin inputs per group) in ONE pass over dD:
//   dw[c, g, kk] += sum_r dD[r, c] * m[r] * x[brow(r, kk), c * gin + g],   dbias[c] += sum_r dD[r, c] * m[r]
// (dw in the parameter's own (C, gin, k) layout)
// (vrd_colsum computes one (g, kk) per launch: twelve launches and twelve passes over dD per q / k / v convolution triple).
// block = rpb rows x 64 columns, a quarter of the rows per wave (as vrd_colsum); one atomic per column, output and block.
// ------------------------------------------------------------------------------------------------------------------
template <int KS, int GIN>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ x,
                                                           int64_t ldx, int bs, int T, const uint8_t* __restrict__ mask,
                                                           int64_t rows, int C, int rpb, float* __restrict__ dw,
                                                           float* __restrict__ dbias) {
    __shared__ float red[3][GIN * KS + 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool col_ok = c < C;
    const int wrows = rpb / 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb + (int64_t)wave * wrows;
    const int64_t r1 = r0 + wrows < rows ? r0 + wrows : rows;
    int64_t seq = r0 / T;
    int tpos = (int)(r0 - seq * T);
    float sw[GIN][KS], sb = 0.f;
#pragma unroll
    for (int g = 0; g < GIN; ++g)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) sw[g][kk] = 0.f;
    for (int64_t rb = r0; rb < r1; rb += 4) {
        float av[4], xv[4][GIN][KS];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + u;
            const bool live = col_ok && r < r1 && (!mask || mask[r]);
            int tq = tpos + u;
            int64_t sq = seq;
            while (tq >= T) tq -= T, ++sq;
            av[u] = live ? a[r * lda + c] : 0.f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int tb = bs * tq + kk - KS / 2;
                const bool ok = live && tb >= 0 && tb < bs * T;
#pragma unroll
                for (int g = 0; g < GIN; ++g) xv[u][g][kk] = ok ? x[(sq * (int64_t)bs * T + tb) * ldx + (int64_t)c * GIN + g] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sb += av[u];
#pragma unroll
            for (int g = 0; g < GIN; ++g)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) sw[g][kk] = fmaf(av[u], xv[u][g][kk], sw[g][kk]);
        }
        tpos += 4;
        while (tpos >= T) tpos -= T, ++seq;
    }
    if (wave > 0) {
#pragma unroll
        for (int g = 0; g < GIN; ++g)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) red[wave - 1][g * KS + kk][lane] = sw[g][kk];
        red[wave - 1][GIN * KS][lane] = sb;
    }
    __syncthreads();
    if (wave > 0 || !col_ok) return;
#pragma unroll
    for (int g = 0; g < GIN; ++g)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            atomicAdd(dw + (int64_t)c * (GIN * KS) + g * KS + kk, sw[g][kk] + red[0][g * KS + kk][lane] + red[1][g * KS + kk][lane] + red[2][g * KS + kk][lane]);
    if (dbias) atomicAdd(dbias + c, sb + red[0][GIN * KS][lane] + red[1][GIN * KS][lane] + red[2][GIN * KS][lane]);
}

// ------------------------------------------------------------------------------------------------------------------
// The two column-sum kernels above with four channels per lane (float4 rows: 1 KiB per wave and row instead of 256 B) for their
// common cases, and the row blocks' sums as rows of `partial` for colpartial_reduce_kernel (nullable: then one atomic per column
// and block as above).  Block = rpb rows x 256 columns, a quarter of the rows per wave, four rows' loads in flight.
//   colsum_vec_kernel:        out[c] += sum_r a[r, c] * (b ? b[r, c] : 1) * m[r] * rs[r]        (b on the rows of a)
//   dwconv_wgrad_vec_kernel:  k = 3, one input per group: dw[c, kk] += sum_r dD[r, c] m[r] x[in_row(r, kk), c]; dbias[c] += ...
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fma4(float4& s, const float4& a, const float4& b) {
    s.x = fmaf(a.x, b.x, s.x), s.y = fmaf(a.y, b.y, s.y), s.z = fmaf(a.z, b.z, s.z), s.w = fmaf(a.w, b.w, s.w);
}
__device__ __forceinline__ void add4(float4& s, const float4& a) { s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w; }

__global__ __launch_bounds__(256) void colsum_vec_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                         const uint8_t* __restrict__ mask, const float* __restrict__ rscale,
                                                         int64_t rows, int C, int rpb, float* __restrict__ out,
                                                         float* __restrict__ partial) {
    __shared__ float4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    const bool col_ok = c < C;
    const int wrows = rpb / 4;                   // (rpb is a multiple of 16)
    const int64_t r0 = (int64_t)blockIdx.x * rpb + (int64_t)wave * wrows;
    const int64_t r1 = r0 + wrows < rows ? r0 + wrows : rows;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f), one = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 s = zero;
    for (int64_t rb = r0; rb < r1; rb += 4) {
        float4 av[4], bv[4];
        float fv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + u;
            const bool live = col_ok && r < r1 && (!mask || mask[r]);
            av[u] = live ? ld4(a + r * lda + c) : zero;
            bv[u] = live && b ? ld4(b + r * ldb + c) : one;
            fv[u] = live && rscale ? rscale[r] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            av[u].x *= fv[u], av[u].y *= fv[u], av[u].z *= fv[u], av[u].w *= fv[u];
            fma4(s, av[u], bv[u]);
        }
    }
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && col_ok) {
        add4(s, red[0][lane]), add4(s, red[1][lane]), add4(s, red[2][lane]);
        if (partial) st4(partial + (int64_t)blockIdx.x * C + c, s);
        else atomicAdd(out + c, s.x), atomicAdd(out + c + 1, s.y), atomicAdd(out + c + 2, s.z), atomicAdd(out + c + 3, s.w);
    }
}

__global__ __launch_bounds__(256) void dwconv_wgrad_vec_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ x,
                                                               int64_t ldx, int bs, int T, const uint8_t* __restrict__ mask,
                                                               int64_t rows, int C, int rpb, float* __restrict__ dw,
                                                               float* __restrict__ dbias, float* __restrict__ partial) {
    __shared__ float4 red[3][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 256 + lane * 4;
    const bool col_ok = c < C;
    const int wrows = rpb / 4;
    const int64_t r0 = (int64_t)blockIdx.x * rpb + (int64_t)wave * wrows;
    const int64_t r1 = r0 + wrows < rows ? r0 + wrows : rows;
    int64_t seq = r0 / T;
    int tpos = (int)(r0 - seq * T);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sw[3] = {zero, zero, zero}, sb = zero;         // sw[kk] = the four channels' sums of tap kk
    for (int64_t rb = r0; rb < r1; rb += 4) {
        float4 av[4], xv[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t r = rb + u;
            const bool live = col_ok && r < r1 && (!mask || mask[r]);
            int tq = tpos + u;
            int64_t sq = seq;
            while (tq >= T) tq -= T, ++sq;
            av[u] = live ? ld4(a + r * lda + c) : zero;
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int tb = bs * tq + kk - 1;
                const bool ok = live && tb >= 0 && tb < bs * T;
                xv[u][kk] = ok ? ld4(x + (sq * (int64_t)bs * T + tb) * ldx + c) : zero;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            add4(sb, av[u]);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) fma4(sw[kk], av[u], xv[u][kk]);
        }
        tpos += 4;
        while (tpos >= T) tpos -= T, ++seq;
    }
    if (wave > 0) {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) red[wave - 1][kk][lane] = sw[kk];
        red[wave - 1][3][lane] = sb;
    }
    __syncthreads();
    if (wave > 0 || !col_ok) return;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) add4(sw[kk], red[o][kk][lane]);
        add4(sb, red[o][3][lane]);
    }
    // dw is (C, 1, 3): the lane's four channels are twelve consecutive floats, channel-major
    const float o12[12] = {sw[0].x, sw[1].x, sw[2].x, sw[0].y, sw[1].y, sw[2].y, sw[0].z, sw[1].z, sw[2].z, sw[0].w, sw[1].w, sw[2].w};
    if (partial) {
        float* pw = partial + (int64_t)blockIdx.x * ((int64_t)C * 3 + (dbias ? C : 0));
#pragma unroll
        for (int q = 0; q < 3; ++q) st4(pw + (int64_t)c * 3 + 4 * q, make_float4(o12[4 * q], o12[4 * q + 1], o12[4 * q + 2], o12[4 * q + 3]));
        if (dbias) st4(pw + (int64_t)C * 3 + c, sb);
    } else {
#pragma unroll
        for (int q = 0; q < 12; ++q) atomicAdd(dw + (int64_t)c * 3 + q, o12[q]);
        if (dbias) atomicAdd(dbias + c, sb.x), atomicAdd(dbias + c + 1, sb.y), atomicAdd(dbias + c + 2, sb.z), atomicAdd(dbias + c + 3, sb.w);
    }
}

// out[r,c] = v[r,c] * cs[c] * rs[r] * m[r] + res[r,c] * (res_masked ? m[r] : 1) + res2[r,c]
__global__ __launch_bounds__(256) void rowcol_scale_kernel(const float* __restrict__ v, int64_t ldv, int64_t rows, int C4,
                                                           const float* __restrict__ cs, const float* __restrict__ rs,
                                                           const uint8_t* __restrict__ mask, const float* __restrict__ res,
                                                           int64_t ldres, int res_masked, const float* __restrict__ res2,
                                                           int64_t ldres2, float* __restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const int64_t r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const float m = mask ? (float)mask[r] : 1.f;
    const float f = (rs ? rs[r] : 1.f) * m;
    float4 x = ld4(v + r * ldv + c);
    float4 k = cs ? ld4(cs + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    x.x *= k.x * f; x.y *= k.y * f; x.z *= k.z * f; x.w *= k.w * f;
    if (res) {
        const float4 q = ld4(res + r * ldres + c);
        const float rm = res_masked ? m : 1.f;
        x.x += q.x * rm; x.y += q.y * rm; x.z += q.z * rm; x.w += q.w * rm;
    }
    if (res2) {
        const float4 q = ld4(res2 + r * ldres2 + c);
        x.x += q.x; x.y += q.y; x.z += q.z; x.w += q.w;
    }
    st4(out + r * ldo + c, x);
}

// y = act(x), or dx = dy * act'(x)  (act: 1 ReLU, 2 GELU(erf)); n4 float4 groups of a dense (rows x C) matrix with
// leading dimensions
__device__ __forceinline__ float act_fwd(float x, int act) { return act == VRD_ACT_RELU ? fmaxf(x, 0.f) : vrd::gelu_erf(x); }
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == VRD_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
    return 0.5f * (1.f + vrd::erf_f32(x * 0.70710678118654752440f)) + x * phi;
}
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                  int64_t lddy, int64_t rows, int C4, int act, float* __restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C4) return;
    const int64_t r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const float4 a = ld4(x + r * ldx + c);
    float4 o;
    if (dy) {
        const float4 g = ld4(dy + r * lddy + c);
        o = make_float4(g.x * act_grad(a.x, act), g.y * act_grad(a.y, act), g.z * act_grad(a.z, act), g.w * act_grad(a.w, act));
    } else {
        o = make_float4(act_fwd(a.x, act), act_fwd(a.y, act), act_fwd(a.z, act), act_fwd(a.w, act));
    }
    st4(out + r * ldo + c, o);
}

// ------------------------------------------------------------------------------------------------------------------
// LayerNorm backward.  One wave per row, several rows per wave; y = xhat * gamma + beta (ReLU optional):
//   g = dy * (relu ? y > 0 : 1) * gamma;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
//   dgamma += sum_r dy' * xhat,  dbeta += sum_r dy'     (per-lane partial sums, one atomic per channel and wave)
// ------------------------------------------------------------------------------------------------------------------
constexpr int LNB_ROWS = 8;        // rows per wave at least (the host asks for more on long inputs: every block ends in 2 C
                                   // atomics on the same 2 C addresses, and atomics on one address queue up in L2)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                            int64_t lddy, int64_t rows, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int relu, float* __restrict__ dx,
                                                            int64_t lddx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rpw, float* __restrict__ partial) {
    constexpr float inv_c = 1.0f / (256.0f * NV);
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r0 = w * rpw;             // (a wave beyond the last row adds zeros)
    float4 g4[NV], b4[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g4[i] = ld4(gamma + i * 256 + lane * 4);
        b4[i] = ld4(beta + i * 256 + lane * 4);
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the rows of this wave in groups of four, a group's loads requested together (one row at a time ran at the latency of
    // its loads)
    for (int h = 0; h < rpw / 4; ++h) {
    const int64_t rh = r0 + 4 * h;
    if (rh >= rows) break;
    float4 vr[4][NV], dr[4][NV];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = rh + q < rows ? rh + q : rows - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            vr[q][i] = ld4(x + r * ldx + i * 256 + lane * 4);
            dr[q][i] = ld4(dy + r * lddy + i * 256 + lane * 4);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t r = rh + q;
        if (r >= rows) break;
        float4 v[NV], d[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = vr[q][i];
            d[i] = dr[q][i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = vrd::wave_sum(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = 1.0f / sqrtf(vrd::wave_sum(ss) * inv_c + LN_EPS);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float* xh = reinterpret_cast<float*>(&v[i]);
            float* dd = reinterpret_cast<float*>(&d[i]);
            const float* gg = reinterpret_cast<const float*>(&g4[i]);
            const float* bb = reinterpret_cast<const float*>(&b4[i]);
            float* pdg = reinterpret_cast<float*>(&dg[i]);
            float* pdb = reinterpret_cast<float*>(&db[i]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xh[c] *= rstd;
                if (relu && fmaf(xh[c], gg[c], bb[c]) <= 0.f) dd[c] = 0.f;
                pdg[c] = fmaf(dd[c], xh[c], pdg[c]);
                pdb[c] += dd[c];
                dd[c] *= gg[c];
                sg += dd[c];
                sgx = fmaf(dd[c], xh[c], sgx);
            }
        }
        const float mg = vrd::wave_sum(sg) * inv_c, mgx = vrd::wave_sum(sgx) * inv_c;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 o;
            o.x = rstd * (d[i].x - mg - v[i].x * mgx);
            o.y = rstd * (d[i].y - mg - v[i].y * mgx);
            o.z = rstd * (d[i].z - mg - v[i].z * mgx);
            o.w = rstd * (d[i].w - mg - v[i].w * mgx);
            st4(dx + r * lddx + i * 256 + lane * 4, o);
        }
    }
    }       // groups of four rows
    // the four waves of the block add their partial sums in LDS; one atomic per channel and BLOCK (the atomics on the 2 C
    // addresses were what the kernel spent its time on)
    __shared__ float4 red[2][3][NV][64];
    const int wv = threadIdx.x >> 6;
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[0][wv - 1][i][lane] = dg[i], red[1][wv - 1][i][lane] = db[i];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 g = dg[i], b = db[i];
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float4 pg = red[0][o][i][lane], pb = red[1][o][i][lane];
                g.x += pg.x, g.y += pg.y, g.z += pg.z, g.w += pg.w;
                b.x += pb.x, b.y += pb.y, b.z += pb.z, b.w += pb.w;
            }
            if (partial) {       // the block's row of the (blocks, 2 C) partial sums: colpartial_reduce_kernel adds them up
                float* pp = partial + (int64_t)blockIdx.x * (2 * 256 * NV) + i * 256 + lane * 4;
                st4(pp, g);
                st4(pp + 256 * NV, b);
                continue;
            }
            float* dgp = dgamma + i * 256 + lane * 4;
            float* dbp = dbeta + i * 256 + lane * 4;
            atomicAdd(dgp + 0, g.x), atomicAdd(dgp + 1, g.y), atomicAdd(dgp + 2, g.z), atomicAdd(dgp + 3, g.w);
            atomicAdd(dbp + 0, b.x), atomicAdd(dbp + 1, b.y), atomicAdd(dbp + 2, b.z), atomicAdd(dbp + 3, b.w);
        }
    }
}

// out[c] += sum_p partial[p * cols + c]: the per-block column sums of a kernel whose blocks would otherwise each end in one
// atomic per column (atomics on one address are worked off one after the other, ~50 ns each: 512 blocks = 25 us).  Block =
// 64 columns x 32 partial rows (a wave takes eight of them), one atomic per column and block: parts / 32 per address.
__global__ __launch_bounds__(256) void colpartial_reduce_kernel(const float* __restrict__ partial, int parts, int cols,
                                                                float* __restrict__ out0, float* __restrict__ out1, int split) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int p0 = blockIdx.y * 32 + wave * 8;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (c < cols && p0 + q < parts) ? partial[(int64_t)(p0 + q) * cols + c] : 0.f;
    float s = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave == 0 && c < cols) {
        s += red[0][lane] + red[1][lane] + red[2][lane];
        atomicAdd(c < split ? out0 + c : out1 + (c - split), s);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// depthwise conv, input gradient.  Forward: D_o[b, to, c] = mask[b, to] * (bias + sum_{g,k} w_o[c,g,k] *
// xin[b, s*to + k - ks/2, gin*c + g]),  xin = x (+ x_up[b, t/2]).  Hence
//   dxin[b, ti, gin*c+g] = sum_o sum_k [to = (ti - k + ks/2) / s integral, in range] mask[b,to] dD_o[b,to,c] w_o[c,g,k]
// thread = one input element (b, ti, cin); with dx_up the thread owns the input pair (2 tu, 2 tu + 1) and also writes
// their sum (gradient of the nearest x2 upsample).
// ------------------------------------------------------------------------------------------------------------------
struct DwBwdArgs {
    const float* dD[3];
    int64_t lddd[3];
    const float* w[3];
    int n_out, B, Tin, C, ksize, stride, gin;
    const uint8_t* mask_out;
    float* dx;
    int64_t lddx;
    float* dx_up;
    int64_t lddx_up;
};
__device__ __forceinline__ float dw_bwd_elem(const DwBwdArgs& p, int b, int ti, int cin) {
    const int c = cin / p.gin, g = cin - c * p.gin;
    const int Tout = p.Tin / p.stride;
    float s = 0.f;
    for (int k = 0; k < p.ksize; ++k) {
        const int tn = ti - k + p.ksize / 2;
        if (tn < 0 || tn % p.stride) continue;
        const int to = tn / p.stride;
        if (to >= Tout) continue;
        const int64_t row = (int64_t)b * Tout + to;
        if (p.mask_out && !p.mask_out[row]) continue;
        for (int o = 0; o < p.n_out; ++o) s = fmaf(p.dD[o][row * p.lddd[o] + c], p.w[o][(c * p.gin + g) * p.ksize + k], s);
    }
    return s;
}
// The common case -- k = 3, stride 1, one input per group, no upsample branch, C % 4 == 0 -- with four channels per thread:
// dx[b, ti, c] = sum_o sum_k m[b, ti + 1 - k] dD_o[b, ti + 1 - k, c] w_o[c, k]: three rows x n_out float4 of dD and 3 n_out float4
// of weights (a channel's taps are contiguous: 12 floats for 4 channels) per 4 outputs.  The element-per-thread form below spends
// ~100 instructions per element on index arithmetic and scalar loads and ran at 3.2 TB/s of its traffic.
template <int NOUT>
__global__ __launch_bounds__(256) void dwconv_bwd_vec_kernel(DwBwdArgs p) {
    const int C4 = p.C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.B * p.Tin * C4) return;
    const int64_t r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const int ti = (int)(r % p.Tin);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w[NOUT][3];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int q = 0; q < 3; ++q) w[o][q] = ld4(p.w[o] + (int64_t)c * 3 + 4 * q);       // w[c .. c+3][0 .. 2], row-major
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int to = ti + 1 - k;
        if (to < 0 || to >= p.Tin) continue;
        const int64_t row = r + 1 - k;
        if (p.mask_out && !p.mask_out[row]) continue;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float4 d = ld4(p.dD[o] + row * p.lddd[o] + c);
            const float* wf = reinterpret_cast<const float*>(&w[o][0]);          // wf[3 * ch + k]
            acc.x = fmaf(d.x, wf[k], acc.x);
            acc.y = fmaf(d.y, wf[3 + k], acc.y);
            acc.z = fmaf(d.z, wf[6 + k], acc.z);
            acc.w = fmaf(d.w, wf[9 + k], acc.w);
        }
    }
    st4(p.dx + r * p.lddx + c, acc);
}

__global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwBwdArgs p) {
    const int Cin = p.C * p.gin;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p.dx_up) {
        const int64_t total = (int64_t)p.B * (p.Tin / 2) * Cin;
        if (idx >= total) return;
        const int cin = (int)(idx % Cin);
        const int64_t ru = idx / Cin;
        const int b = (int)(ru / (p.Tin / 2)), tu = (int)(ru - (int64_t)b * (p.Tin / 2));
        const float g0 = dw_bwd_elem(p, b, 2 * tu, cin), g1 = dw_bwd_elem(p, b, 2 * tu + 1, cin);
        p.dx[((int64_t)b * p.Tin + 2 * tu) * p.lddx + cin] = g0;
        p.dx[((int64_t)b * p.Tin + 2 * tu + 1) * p.lddx + cin] = g1;
        p.dx_up[ru * p.lddx_up + cin] = g0 + g1;
        return;
    }
    const int64_t total = (int64_t)p.B * p.Tin * Cin;
    if (idx >= total) return;
    const int cin = (int)(idx % Cin);
    const int64_t r = idx / Cin;
    const int b = (int)(r / p.Tin), ti = (int)(r - (int64_t)b * p.Tin);
    p.dx[r * p.lddx + cin] = dw_bwd_elem(p, b, ti, cin);
}

// ------------------------------------------------------------------------------------------------------------------
// banded attention backward (C = 512; lane l owns channels [8l, 8l+8); GROUP = head_dim / 8 lanes per head).
// Pass 1, one wave per query row t: recompute the window probabilities P[t, j] (models/blocks.py:950-986: masked keys
// -1e4, out-of-range -inf, masked query rows all zero), dP = dO . v, dS = P * (dP - sum_j P dP); dq = scale * sum_j dS k;
// P and dS of the row go to scratch (rows x heads x W).  Pass 2, one wave per key row j: dk_j = scale * sum_t dS[t, j] q_t,
// dv_j = sum_t P[t, j] dO_t over the (at most W) queries whose window holds j.
// ------------------------------------------------------------------------------------------------------------------
template <int GROUP>
__device__ __forceinline__ float head_sum(float d) { return vrd::group_sum<GROUP>(d); }
__device__ __forceinline__ float dot8(const float4& a0, const float4& a1, const float4& b0, const float4& b1) {
    return (a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w) + (a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w);
}
constexpr int LA_WMAX = 9;
template <int GROUP>
__global__ __launch_bounds__(256) void local_attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, int64_t ld,
                                                               const float* __restrict__ dO, int64_t lddo,
                                                               const uint8_t* __restrict__ mask, const float* __restrict__ rel,
                                                               int B, int T, int W, float scale,
                                                               float* __restrict__ dq, int64_t lddq, float* __restrict__ P,
                                                               float* __restrict__ dS) {
    const int HW = W / 2;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * T) return;
    const int t = (int)(row % T);
    constexpr int H = 64 / GROUP;
    const int head = lane / GROUP;
    float* const Pr = P + (row * H + head) * W;
    float* const dSr = dS + (row * H + head) * W;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    if (!mask[row]) {
        if (lane % GROUP == 0)
            for (int j = 0; j < W; ++j) Pr[j] = 0.f, dSr[j] = 0.f;
        st4(dq + row * lddq + lane * 8, g0);
        st4(dq + row * lddq + lane * 8 + 4, g0);
        return;
    }
    const float4 q0 = ld4(q + row * ld + lane * 8), q1 = ld4(q + row * ld + lane * 8 + 4);
    const float4 o0 = ld4(dO + row * lddo + lane * 8), o1 = ld4(dO + row * lddo + lane * 8 + 4);
    float s[LA_WMAX], dp[LA_WMAX];
    float m = -INFINITY;
    for (int j = 0; j < W; ++j) {
        const int tj = t + j - HW;
        s[j] = -INFINITY;
        dp[j] = 0.f;
        if (tj < 0 || tj >= T) continue;
        const float* kr = k + (row + j - HW) * ld + lane * 8;
        const float* vr = v + (row + j - HW) * ld + lane * 8;
        const float d = head_sum<GROUP>(dot8(q0, q1, ld4(kr), ld4(kr + 4))) * scale;
        dp[j] = head_sum<GROUP>(dot8(o0, o1, ld4(vr), ld4(vr + 4)));
        s[j] = (rel ? d + rel[head * W + j] : d) + (mask[row + j - HW] ? 0.f : -1e4f);
        m = fmaxf(m, s[j]);
    }
    float den = 0.f;
    for (int j = 0; j < W; ++j) { s[j] = __expf(s[j] - m); den += s[j]; }
    const float inv = 1.0f / den;
    float dsum = 0.f;
    for (int j = 0; j < W; ++j) { s[j] *= inv; dsum = fmaf(s[j], dp[j], dsum); }
    for (int j = 0; j < W; ++j) {
        const float ds = s[j] * (dp[j] - dsum);
        if (lane % GROUP == 0) Pr[j] = s[j], dSr[j] = ds;
        const int tj = t + j - HW;
        if (tj < 0 || tj >= T) continue;
        const float* kr = k + (row + j - HW) * ld + lane * 8;
        const float4 k0 = ld4(kr), k1 = ld4(kr + 4);
        const float f = ds * scale;
        g0.x = fmaf(f, k0.x, g0.x); g0.y = fmaf(f, k0.y, g0.y); g0.z = fmaf(f, k0.z, g0.z); g0.w = fmaf(f, k0.w, g0.w);
        g1.x = fmaf(f, k1.x, g1.x); g1.y = fmaf(f, k1.y, g1.y); g1.z = fmaf(f, k1.z, g1.z); g1.w = fmaf(f, k1.w, g1.w);
    }
    st4(dq + row * lddq + lane * 8, g0);
    st4(dq + row * lddq + lane * 8 + 4, g1);
}
template <int GROUP>
__global__ __launch_bounds__(256) void local_attn_bwd_kv_kernel(const float* __restrict__ q, int64_t ld,
                                                                const float* __restrict__ dO, int64_t lddo, int B, int T, int W,
                                                                float scale, const float* __restrict__ P,
                                                                const float* __restrict__ dS, float* __restrict__ dk,
                                                                float* __restrict__ dv, int64_t lddkv) {
    const int HW = W / 2;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // key row j
    if (row >= (int64_t)B * T) return;
    const int tj = (int)(row % T);
    constexpr int H = 64 / GROUP;
    const int head = lane / GROUP;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, c0 = a0, c1 = a0;
    for (int i = 0; i < W; ++i) {           // query t = tj + HW - i sees key tj at window position i
        const int t = tj + HW - i;
        if (t < 0 || t >= T) continue;
        const int64_t qr = row + HW - i;
        const float p = P[(qr * H + head) * W + i], ds = dS[(qr * H + head) * W + i] * scale;
        const float4 q0 = ld4(q + qr * ld + lane * 8), q1 = ld4(q + qr * ld + lane * 8 + 4);
        const float4 o0 = ld4(dO + qr * lddo + lane * 8), o1 = ld4(dO + qr * lddo + lane * 8 + 4);
        a0.x = fmaf(ds, q0.x, a0.x); a0.y = fmaf(ds, q0.y, a0.y); a0.z = fmaf(ds, q0.z, a0.z); a0.w = fmaf(ds, q0.w, a0.w);
        a1.x = fmaf(ds, q1.x, a1.x); a1.y = fmaf(ds, q1.y, a1.y); a1.z = fmaf(ds, q1.z, a1.z); a1.w = fmaf(ds, q1.w, a1.w);
        c0.x = fmaf(p, o0.x, c0.x); c0.y = fmaf(p, o0.y, c0.y); c0.z = fmaf(p, o0.z, c0.z); c0.w = fmaf(p, o0.w, c0.w);
        c1.x = fmaf(p, o1.x, c1.x); c1.y = fmaf(p, o1.y, c1.y); c1.z = fmaf(p, o1.z, c1.z); c1.w = fmaf(p, o1.w, c1.w);
    }
    st4(dk + row * lddkv + lane * 8, a0);
    st4(dk + row * lddkv + lane * 8 + 4, a1);
    st4(dv + row * lddkv + lane * 8, c0);
    st4(dv + row * lddkv + lane * 8 + 4, c1);
}

// ------------------------------------------------------------------------------------------------------------------
// global attention backward, scores: one wave per (b, h, tq).  P = softmax_j(scale q.k_j | kv_mask (masked: -inf)),
// dP_j = dO . v_j, dS = P * (dP - sum P dP).  Lane j handles keys j, j + 64, ...; q and dO rows sit in LDS.
// (models/local_transformer.py:163-183; the predictor's 9-query attention :44-63 runs through the same kernel)
// ------------------------------------------------------------------------------------------------------------------
constexpr int AB_HD_MAX = 128, AB_TK_MAX = 1024;
__global__ __launch_bounds__(256) void attn_bwd_probs_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                             const float* __restrict__ v, int64_t ldkv,
                                                             const float* __restrict__ dO, int64_t lddo,
                                                             const uint8_t* __restrict__ kv_mask, int B, int Tq, int Tk, int H,
                                                             int hd, float scale, float* __restrict__ P, float* __restrict__ dS) {
    __shared__ float qs[4][AB_HD_MAX], os[4][AB_HD_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;          // (b, h, tq)
    const bool live = w < (int64_t)B * H * Tq;
    const int tq = live ? (int)(w % Tq) : 0;
    const int h = live ? (int)((w / Tq) % H) : 0;
    const int b = live ? (int)(w / ((int64_t)Tq * H)) : 0;
    if (live)
        for (int d = lane; d < hd; d += 64) {
            qs[wave][d] = q[((int64_t)b * Tq + tq) * ldq + h * hd + d] * scale;
            os[wave][d] = dO[((int64_t)b * Tq + tq) * lddo + h * hd + d];
        }
    __syncthreads();
    if (!live) return;
    constexpr int NJ = AB_TK_MAX / 64;
    float s[NJ], dp[NJ];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        s[i] = -INFINITY;
        dp[i] = 0.f;
        if (j >= Tk) continue;
        if (kv_mask && !kv_mask[(int64_t)b * Tk + j]) continue;
        const float* kr = k + ((int64_t)b * Tk + j) * ldkv + h * hd;
        const float* vr = v + ((int64_t)b * Tk + j) * ldkv + h * hd;
        float a = 0.f, c = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 kk = ld4(kr + d), vv = ld4(vr + d);
            a += qs[wave][d] * kk.x + qs[wave][d + 1] * kk.y + qs[wave][d + 2] * kk.z + qs[wave][d + 3] * kk.w;
            c += os[wave][d] * vv.x + os[wave][d + 1] * vv.y + os[wave][d + 2] * vv.z + os[wave][d + 3] * vv.w;
        }
        s[i] = a;
        dp[i] = c;
        m = fmaxf(m, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        s[i] = (m == -INFINITY || s[i] == -INFINITY) ? 0.f : __expf(s[i] - m);
        den += s[i];
    }
    den = vrd::wave_sum(den);
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) { s[i] *= inv; dsum = fmaf(s[i], dp[i], dsum); }
    dsum = vrd::wave_sum(dsum);
    float* const Pr = P + w * Tk;
    float* const dSr = dS + w * Tk;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        if (j < Tk) { Pr[j] = s[i]; dSr[j] = s[i] * (dp[i] - dsum); }
    }
}

// C[z][i][n] (= or +=) alpha * sum_k A[z][i][k] * B[z][k][n];  z = (z0, z1) with z1 < Z1; every operand addressed by
// (offset of z0, offset of z1, stride of the row index, stride of the column index), in floats.  thread = (i, n), n
// fastest: choose the operand roles so that B and C are contiguous along n.  K loop in registers, no staging.
struct BmmArgs {
    const float *A, *B;
    float* C;
    int64_t a0, a1, ai, ak;
    int64_t b0, b1, bk, bn;
    int64_t c0, c1, ci, cn;
    int Z0, Z1, M, N, K;
    float alpha;
    int accumulate;
};
__global__ __launch_bounds__(256) void bmm_kernel(BmmArgs p) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int z = blockIdx.z;
    if (n >= p.N || i >= p.M) return;
    const int z0 = z / p.Z1, z1 = z - z0 * p.Z1;
    const float* a = p.A + z0 * p.a0 + z1 * p.a1 + i * p.ai;
    const float* b = p.B + z0 * p.b0 + z1 * p.b1 + n * p.bn;
    float s = 0.f;
    // eight k at a time with their loads requested together (same summation order; one k per iteration ran at the latency
    // of its two loads)
    int k = 0;
    for (; k + 8 <= p.K; k += 8) {
        float av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            av[u] = a[(k + u) * p.ak];
            bv[u] = b[(k + u) * p.bk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(av[u], bv[u], s);
    }
    for (; k < p.K; ++k) s = fmaf(a[k * p.ak], b[k * p.bk], s);
    float* c = p.C + z0 * p.c0 + z1 * p.c1 + i * p.ci + n * p.cn;
    *c = p.accumulate ? *c + p.alpha * s : p.alpha * s;
}

// The same product on the matrix cores, exact f32 (v_mfma_f32_32x32x2_f32: an fmaf chain per output, k ascending) -- round 3:
// at vidor.yaml's training sizes (48 pairs x 512 frames x 8 heads) the three products of the attention backward were
// 40 ms of a 150 ms step on the kernel above (one thread per output, every operand element re-read per thread).
// Workgroup = 4 waves = a 64 x 64 tile of C, one 32 x 32 accumulator per wave; K steps of 16 through k-major LDS tiles
// (a lane's operand of one MFMA is A[k][row] / B[k][col]: 32 consecutive floats per half-wave, conflict free).  An operand is
// read along whichever of its two indices has stride 1 (four elements per thread and K step, 16-byte loads when the launch's
// pointers and strides allow); anything goes through the scalar gather.  The next K step's elements are in registers while
// the current one multiplies.
typedef __attribute__((ext_vector_type(16))) float bmm_f32x16;
constexpr int BM_T = 64, BM_K = 16, BM_LD = BM_T + 4;
template <bool VEC>
__global__ __launch_bounds__(256) void bmm_mfma_kernel(BmmArgs p) {
    __shared__ float As[BM_K][BM_LD], Bs[BM_K][BM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int z = blockIdx.z, z0 = z / p.Z1, z1 = z - z0 * p.Z1;
    const int i0 = blockIdx.y * BM_T, n0 = blockIdx.x * BM_T;
    const float* A = p.A + z0 * p.a0 + z1 * p.a1;
    const float* Bm = p.B + z0 * p.b0 + z1 * p.b1;
    // staging roles: "along k" = thread (row t / 4, four consecutive k), "along the row" = thread (k t / 16, four consecutive rows)
    const bool a_k = p.ak == 1, b_n = p.bn == 1;
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        if (a_k) {                    // A[i][k], k contiguous
            const int r = tid >> 2, kq = (tid & 3) * 4;
            const float* src = A + (int64_t)(i0 + r) * p.ai + (k0 + kq);
            if (VEC && i0 + r < p.M && k0 + kq + 3 < p.K) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                ra[0] = t.x, ra[1] = t.y, ra[2] = t.z, ra[3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ra[u] = (i0 + r < p.M && k0 + kq + u < p.K) ? src[u] : 0.f;
            }
        } else {                      // rows contiguous (or a general stride): thread (k, four rows)
            const int k = tid >> 4, rq = (tid & 15) * 4;
            const float* src = A + (int64_t)(k0 + k) * p.ak + (int64_t)(i0 + rq) * p.ai;
            if (VEC && p.ai == 1 && k0 + k < p.K && i0 + rq + 3 < p.M) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                ra[0] = t.x, ra[1] = t.y, ra[2] = t.z, ra[3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) ra[u] = (k0 + k < p.K && i0 + rq + u < p.M) ? src[u * p.ai] : 0.f;
            }
        }
        if (b_n) {                    // B[k][n], n contiguous: thread (k, four columns)
            const int k = tid >> 4, cq = (tid & 15) * 4;
            const float* src = Bm + (int64_t)(k0 + k) * p.bk + (n0 + cq);
            if (VEC && k0 + k < p.K && n0 + cq + 3 < p.N) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                rb[0] = t.x, rb[1] = t.y, rb[2] = t.z, rb[3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) rb[u] = (k0 + k < p.K && n0 + cq + u < p.N) ? src[u] : 0.f;
            }
        } else {                      // k contiguous (B given transposed) or general: thread (column, four k)
            const int c = tid >> 2, kq = (tid & 3) * 4;
            const float* src = Bm + (int64_t)(n0 + c) * p.bn + (int64_t)(k0 + kq) * p.bk;
            if (VEC && p.bk == 1 && n0 + c < p.N && k0 + kq + 3 < p.K) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                rb[0] = t.x, rb[1] = t.y, rb[2] = t.z, rb[3] = t.w;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) rb[u] = (n0 + c < p.N && k0 + kq + u < p.K) ? src[u * p.bk] : 0.f;
            }
        }
    };
    auto stage = [&]() {
        if (a_k) {
            const int r = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) As[kq + u][r] = ra[u];
        } else {
            const int k = tid >> 4, rq = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&As[k][rq]) = make_float4(ra[0], ra[1], ra[2], ra[3]);
        }
        if (b_n) {
            const int k = tid >> 4, cq = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&Bs[k][cq]) = make_float4(rb[0], rb[1], rb[2], rb[3]);
        } else {
            const int c = tid >> 2, kq = (tid & 3) * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) Bs[kq + u][c] = rb[u];
        }
    };
    bmm_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += BM_K) {
        __syncthreads();                              // everybody is done reading the previous tiles
        stage();
        __syncthreads();
        if (k0 + BM_K < p.K) fetch(k0 + BM_K);
#pragma unroll
        for (int kk = 0; kk < BM_K / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + lh][wm * 32 + li], Bs[2 * kk + lh][wn * 32 + li], acc, 0, 0, 0);
    }
    // accumulator register e of lane (li, lh): row (e & 3) + 8 * (e >> 2) + 4 * lh, column li
    float* C = p.C + z0 * p.c0 + z1 * p.c1;
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = i0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (i < p.M && n < p.N) {
            float* c = C + (int64_t)i * p.ci + (int64_t)n * p.cn;
            *c = p.accumulate ? *c + p.alpha * acc[e] : p.alpha * acc[e];
        }
    }
}

// Global attention backward on matrices (round 3): S = scale * Q K^T and dP = dO V^T come from two vrd_bmm products; this
// kernel turns a row of each into P = softmax_j(S | kv_mask (masked: 0)) and dS = P * (dP - sum_j P dP), in place.
// One wave per (b, h, tq) row, lanes over the keys (Tk <= 1024: 16 per lane in registers).
__global__ __launch_bounds__(256) void attn_bwd_softmax_kernel(float* __restrict__ P, float* __restrict__ dS, const uint8_t* __restrict__ kv_mask,
                                                               int64_t rows, int Tq, int Tk, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= rows) return;
    const int b = (int)(w / ((int64_t)Tq * H));
    constexpr int NJ = AB_TK_MAX / 64;
    float s[NJ], dp[NJ];
    float* const Pr = P + w * Tk;
    float* const dSr = dS + w * Tk;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        s[i] = -INFINITY;
        dp[i] = 0.f;
        if (j >= Tk || (kv_mask && !kv_mask[(int64_t)b * Tk + j])) continue;
        s[i] = Pr[j];
        dp[i] = dSr[j];
        m = fmaxf(m, s[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        s[i] = (m == -INFINITY || s[i] == -INFINITY) ? 0.f : __expf(s[i] - m);
        den += s[i];
    }
    den = vrd::wave_sum(den);
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    float dsum = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) { s[i] *= inv; dsum = fmaf(s[i], dp[i], dsum); }
    dsum = vrd::wave_sum(dsum);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        const int j = lane + 64 * i;
        if (j < Tk) { Pr[j] = s[i]; dSr[j] = s[i] * (dp[i] - dsum); }
    }
}

// MaxPool1d(3, 2, 1)(x) * mask[::2] backward: dx[b, ti, c] = sum over the (1 or 2) windows holding ti in which ti is the
// FIRST maximum (ATen's tie rule) of mask[2 to] * dy[b, to, c]
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                          int64_t lddy, int B, int Tin, int C, const uint8_t* __restrict__ mask_in,
                                                          float* __restrict__ dx, int64_t lddx) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * Tin * C) return;
    const int c = (int)(idx % C);
    const int64_t r = idx / C;
    const int b = (int)(r / Tin), ti = (int)(r - (int64_t)b * Tin);
    const int Tout = Tin / 2;
    const float xv = x[r * ldx + c];
    float g = 0.f;
    // windows: to with 2 to - 1 <= ti <= 2 to + 1
    for (int to = (ti + 1) / 2 - ((ti & 1) ? 1 : 0); to <= (ti + 1) / 2; ++to) {
        if (to < 0 || to >= Tout) continue;
        if (!mask_in[(int64_t)b * Tin + 2 * to]) continue;
        bool first_max = true;
        for (int tt = 2 * to - 1; tt <= 2 * to + 1; ++tt) {
            if (tt < 0 || tt >= Tin || tt == ti) continue;
            const float o = x[((int64_t)b * Tin + tt) * ldx + c];
            if (o > xv || (o == xv && tt < ti)) first_max = false;
        }
        if (first_max) g += dy[((int64_t)b * Tout + to) * lddy + c];
    }
    dx[r * lddx + c] = g;
}

}  // namespace

// rows per block of the column-sum kernels: ~1024 blocks on long inputs, a multiple of 32 (a quarter per wave, eight rows per
// iteration), 32 .. 2048
static int64_t colsum_rows_per_block(int64_t rows, int col_blocks) {
    int64_t rpb = rows * col_blocks / 1024;
    rpb = (rpb + 31) / 32 * 32;
    if (rpb < CS_ROWS) rpb = CS_ROWS;
    if (rpb > 2048) rpb = 2048;
    return rpb;
}

// rows per block of the four-channels-per-lane column-sum kernels: ~4 blocks per CU (16 waves: the loop is bound by the latency of
// its loads; 2 per CU: 1.67 ms per training step in dwconv_wgrad_vec_kernel, 4: 1.11, 8: 1.24), a multiple of 16 rows
static int64_t colsum_vec_rows_per_block(int64_t rows, int col_blocks) {
    int64_t row_blocks = 1024 / col_blocks;
    if (row_blocks < 1) row_blocks = 1;
    int64_t rpb = (rows + row_blocks - 1) / row_blocks;
    rpb = (rpb + 15) / 16 * 16;
    return rpb < 16 ? 16 : rpb;
}

extern "C" {

int vrd_gemm_wgrad(const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* row_mask, int64_t M, int N, int Cin,
                   int taps, int T, float* dW, void* stream) {
    VRD_CHECK_ARG(G && X && dW, "vrd_gemm_wgrad: null pointer");
    VRD_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && (taps == 1 || taps == 3), "vrd_gemm_wgrad: bad sizes M=%lld N=%d Cin=%d taps=%d", (long long)M, N, Cin, taps);
    VRD_CHECK_ARG(ldg >= N && ldx >= Cin, "vrd_gemm_wgrad: leading dimension too small");
    VRD_CHECK_ARG(T > 0 && M % T == 0, "vrd_gemm_wgrad: M (%lld) must be a multiple of T (%d)", (long long)M, T);
    const int K = Cin * taps;
    const int tiles_n = (N + 63) / 64, tiles_k = (K + 63) / 64;
    const int64_t chunks = (M + 4 * WG_CHUNK - 1) / (4 * WG_CHUNK);
    VRD_CHECK_ARG(chunks <= 65535, "vrd_gemm_wgrad: too many rows (%lld)", (long long)M);
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 2.0 * (double)M * N * K, 4.0 * ((double)M * (N + Cin) + (double)N * K));
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles_n * tiles_k, (unsigned)chunks), dim3(256), 0, s, G, ldg, X, ldx, row_mask, M, N, Cin, taps, T,
                       tiles_k, dW);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_gemm_wgrad_x3(const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* row_mask, int64_t M, int N, int Cin,
                      int taps, int T, float* dW, float* dbias, float* scratch, int64_t scratch_floats, const float* g_scale,
                      void* stream) {
    VRD_CHECK_ARG(G && X && dW, "vrd_gemm_wgrad_x3: null pointer");
    VRD_CHECK_ARG(M > 0 && N > 0 && Cin > 0 && (taps == 1 || taps == 3), "vrd_gemm_wgrad_x3: bad sizes M=%lld N=%d Cin=%d taps=%d", (long long)M, N, Cin, taps);
    VRD_CHECK_ARG(ldg >= N && ldx >= Cin, "vrd_gemm_wgrad_x3: leading dimension too small");
    VRD_CHECK_ARG(T > 0 && M % T == 0, "vrd_gemm_wgrad_x3: M (%lld) must be a multiple of T (%d)", (long long)M, T);
    const int K = Cin * taps;
    const int n_cu = vrd::device_cu_count();
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 2.0 * (double)M * N * K, 4.0 * ((double)M * (N + Cin) + (double)N * K));
    static const bool use_lds = [] { const char* e = getenv("VRD_WGRAD_LDS"); return !(e && e[0] == '0'); }();
    // VRD_WGRAD_BIG: 0 = 128 x 128 tiles only, 2 = 256 x 256 tiles whenever the shape allows (lab); default: by rows per chunk
    static const int big_mode = [] { const char* e = getenv("VRD_WGRAD_BIG"); return e ? atoi(e) : 1; }();
    if (use_lds && M >= 256) {
        const bool vec = N % 4 == 0 && Cin % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(G) && aligned16(X);
        // 256 x 256 tiles, one block per CU, when a block then still walks >= WL_BIG_ROWS rows (its start, its 64 k partial sums
        // and their share of the reduction are paid per block); 128 x 128 tiles, two blocks per CU, otherwise
        const int64_t big_tiles = (int64_t)((N + 255) / 256) * ((K + 255) / 256);
        const bool big = big_mode != 0 && vec && N >= 256 && K >= 256 &&
                         (big_mode == 2 || M * big_tiles >= (int64_t)WL_BIG_ROWS * n_cu);
        return big ? launch_wgrad_lds<1>(G, ldg, X, ldx, row_mask, M, N, Cin, taps, T, dW, dbias, scratch, scratch_floats, vec, n_cu, s, g_scale)
                   : launch_wgrad_lds<0>(G, ldg, X, ldx, row_mask, M, N, Cin, taps, T, dW, dbias, scratch, scratch_floats, vec, n_cu, s, g_scale);
    }
    if (dbias) {                                 // the wave kernel has no bias path: a column-sum launch of its own
        const int col_blocks = (N + 63) / 64;
        const int64_t rpb = colsum_rows_per_block(M, col_blocks);
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((M + rpb - 1) / rpb), col_blocks), dim3(256), 0, s, G, ldg, (const float*)nullptr,
                           (int64_t)0, 1, 0, 1, 0, 1, row_mask, (const float*)nullptr, M, N, (int)rpb, dbias);
        VRD_LAUNCH_CHECK();
    }
    const int tiles_n = (N + 63) / 64, tiles_k = (K + 63) / 64;
    // rows per wave: as few row chunks as still fill the chip (every block ends in 4096 atomics), multiples of 32 rows
    const int64_t tiles = (int64_t)tiles_n * tiles_k;
    int64_t want_blocks = (2 * (int64_t)n_cu + tiles - 1) / tiles;                  // row chunks for ~2 blocks per CU
    if (want_blocks < 1) want_blocks = 1;
    int64_t chunk = (M + 4 * want_blocks - 1) / (4 * want_blocks);
    chunk = (chunk + 31) / 32 * 32;
    if (chunk < 64) chunk = 64;
    if (chunk > 1024) chunk = 1024;
    const int64_t chunks = (M + 4 * chunk - 1) / (4 * chunk);
    VRD_CHECK_ARG(chunks <= 65535, "vrd_gemm_wgrad_x3: too many rows (%lld)", (long long)M);
    if (g_scale)
        hipLaunchKernelGGL(wgrad_x3_kernel<true>, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, s, G, ldg, X, ldx, row_mask, M, N, Cin,
                           taps, T, tiles_k, (int)chunk, dW, g_scale);
    else
        hipLaunchKernelGGL(wgrad_x3_kernel<false>, dim3((unsigned)tiles, (unsigned)chunks), dim3(256), 0, s, G, ldg, X, ldx, row_mask, M, N, Cin,
                           taps, T, tiles_k, (int)chunk, dW, (const float*)nullptr);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_dwconv_wgrad(const float* dD, int64_t lddd, const float* x, int64_t ldx, int ksize, int stride, int group_in, int T,
                     const uint8_t* row_mask, int64_t rows, int C, float* dw, float* dbias, float* scratch, int64_t scratch_floats,
                     void* stream) {
    VRD_CHECK_ARG(dD && x && dw && rows > 0 && C > 0 && lddd >= C, "vrd_dwconv_wgrad: bad arguments");
    VRD_CHECK_ARG((ksize == 1 || ksize == 3) && (group_in == 1 || group_in == 2) && stride >= 1 && T > 0 && rows % T == 0,
                  "vrd_dwconv_wgrad: unsupported k=%d group_in=%d stride=%d T=%d rows=%lld", ksize, group_in, stride, T, (long long)rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)rows * C * (1 + group_in * stride));
    if (ksize == 3 && group_in == 1 && C % 4 == 0 && lddd % 4 == 0 && ldx % 4 == 0 && aligned16(dD) && aligned16(x)) {
        const int col_blocks = (C + 255) / 256;
        const int64_t rpb = colsum_vec_rows_per_block(rows, col_blocks);
        const dim3 grid((unsigned)((rows + rpb - 1) / rpb), col_blocks);
        const int64_t pcols = (int64_t)C * 3 + (dbias ? C : 0);
        float* partial = grid.x > 8 && scratch && aligned16(scratch) && scratch_floats >= (int64_t)grid.x * pcols ? scratch : nullptr;
        hipLaunchKernelGGL(dwconv_wgrad_vec_kernel, grid, dim3(256), 0, s, dD, lddd, x, ldx, stride, T, row_mask, rows, C, (int)rpb, dw, dbias,
                           partial);
        VRD_LAUNCH_CHECK();
        if (partial) {
            hipLaunchKernelGGL(colpartial_reduce_kernel, dim3((unsigned)((pcols + 63) / 64), (grid.x + 31) / 32), dim3(256), 0, s, partial,
                               (int)grid.x, (int)pcols, dw, dbias, C * 3);
            VRD_LAUNCH_CHECK();
        }
        return 0;
    }
    const int col_blocks = (C + 63) / 64;
    const int64_t rpb = colsum_rows_per_block(rows, col_blocks);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb), col_blocks);
#define VRD_DWW(KS, GIN) hipLaunchKernelGGL((dwconv_wgrad_kernel<KS, GIN>), grid, dim3(256), 0, s, dD, lddd, x, ldx, stride, T, row_mask, rows, C, (int)rpb, dw, dbias)
    if (ksize == 3 && group_in == 1) VRD_DWW(3, 1);
    else if (ksize == 3) VRD_DWW(3, 2);
    else if (group_in == 1) VRD_DWW(1, 1);
    else VRD_DWW(1, 2);
#undef VRD_DWW
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_colsum(const float* a, int64_t lda, const float* b, int64_t ldb, int b_cstride, int b_coffset, int b_rstride, int shift,
               int T, const uint8_t* row_mask, const float* row_scale, int64_t rows, int C, float* out, float* scratch,
               int64_t scratch_floats, void* stream) {
    VRD_CHECK_ARG(a && out && rows > 0 && C > 0 && lda >= C, "vrd_colsum: bad arguments");
    VRD_CHECK_ARG(!b || (T > 0 && rows % T == 0 && b_cstride >= 1 && b_rstride >= 1 && b_coffset >= 0 && b_coffset < b_cstride),
                  "vrd_colsum: bad second operand (T=%d rows=%lld)", T, (long long)rows);
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)rows * C * (b ? 2 : 1));
    const bool same_rows = !b || (b_cstride == 1 && b_coffset == 0 && b_rstride == 1 && shift == 0);
    if (same_rows && C % 4 == 0 && lda % 4 == 0 && aligned16(a) && (!b || (ldb % 4 == 0 && ldb >= C && aligned16(b)))) {
        const int col_blocks = (C + 255) / 256;
        const int64_t rpb = colsum_vec_rows_per_block(rows, col_blocks);
        const dim3 grid((unsigned)((rows + rpb - 1) / rpb), col_blocks);
        float* partial = grid.x > 8 && scratch && aligned16(scratch) && scratch_floats >= (int64_t)grid.x * C ? scratch : nullptr;
        hipLaunchKernelGGL(colsum_vec_kernel, grid, dim3(256), 0, s, a, lda, b, ldb, row_mask, row_scale, rows, C, (int)rpb, out, partial);
        VRD_LAUNCH_CHECK();
        if (partial) {
            hipLaunchKernelGGL(colpartial_reduce_kernel, dim3((unsigned)((C + 63) / 64), (grid.x + 31) / 32), dim3(256), 0, s, partial, (int)grid.x,
                               C, out, (float*)nullptr, C);
            VRD_LAUNCH_CHECK();
        }
        return 0;
    }
    const int col_blocks = (C + 63) / 64;
    const int64_t rpb = colsum_rows_per_block(rows, col_blocks);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((rows + rpb - 1) / rpb), col_blocks), dim3(256), 0, s, a, lda, b, ldb,
                       b ? b_cstride : 1, b ? b_coffset : 0, b ? b_rstride : 1, shift, b ? T : 1, row_mask, row_scale, rows, C, (int)rpb, out);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_rowcol_scale(const float* v, int64_t ldv, int64_t rows, int C, const float* col_scale, const float* row_scale,
                     const uint8_t* row_mask, const float* res, int64_t ldres, int res_masked, const float* res2, int64_t ldres2,
                     float* out, int64_t ldo, void* stream) {
    VRD_CHECK_ARG(v && out && rows > 0 && C > 0 && C % 4 == 0, "vrd_rowcol_scale: bad arguments (C %% 4 == 0 required)");
    VRD_CHECK_ARG(ldv >= C && ldo >= C && ldv % 4 == 0 && ldo % 4 == 0 && aligned16(v) && aligned16(out) && aligned16(col_scale),
                  "vrd_rowcol_scale: rows must be 16-byte aligned");
    VRD_CHECK_ARG(!res || (ldres >= C && ldres % 4 == 0 && aligned16(res)), "vrd_rowcol_scale: bad res layout");
    VRD_CHECK_ARG(!res2 || (ldres2 >= C && ldres2 % 4 == 0 && aligned16(res2)), "vrd_rowcol_scale: bad res2 layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)rows * C * (2 + (res ? 1 : 0) + (res2 ? 1 : 0)));
    const int64_t n = rows * (C / 4);
    hipLaunchKernelGGL(rowcol_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, ldv, rows, C / 4, col_scale, row_scale,
                       row_mask, res, ldres, res_masked, res2, ldres2, out, ldo);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_activation(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int C, int act, float* out, int64_t ldo,
            void* stream) {
    VRD_CHECK_ARG(x && out && rows > 0 && C > 0 && C % 4 == 0, "vrd_activation: bad arguments (C %% 4 == 0 required)");
    VRD_CHECK_ARG(act == VRD_ACT_RELU || act == VRD_ACT_GELU, "vrd_activation: activation must be ReLU or GELU");
    VRD_CHECK_ARG(ldx >= C && ldo >= C && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(out), "vrd_activation: bad layout");
    VRD_CHECK_ARG(!dy || (lddy >= C && lddy % 4 == 0 && aligned16(dy)), "vrd_activation: bad dy layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)rows * C * (dy ? 3 : 2));
    const int64_t n = rows * (C / 4);
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, dy, lddy, rows, C / 4, act, out, ldo);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_layernorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int C, const float* gamma,
                      const float* beta, int relu, float* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch,
                      int64_t scratch_floats, void* stream) {
    VRD_CHECK_ARG(x && dy && gamma && beta && dx && dgamma && dbeta, "vrd_layernorm_bwd: null pointer");
    VRD_CHECK_ARG(C == 256 || C == 512, "vrd_layernorm_bwd: C must be 256 or 512 (got %d)", C);
    VRD_CHECK_ARG(ldx >= C && lddy >= C && lddx >= C && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && aligned16(x) && aligned16(dy) &&
                      aligned16(dx) && aligned16(gamma) && aligned16(beta),
                  "vrd_layernorm_bwd: rows must be 16-byte aligned");
    if (rows <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 12.0 * (double)rows * C);
    int64_t rpw = (rows / (4 * 512) + 3) / 4 * 4;            // ~512 blocks on long inputs; a multiple of 4 rows, 8 .. 64
    if (rpw < LNB_ROWS) rpw = LNB_ROWS;
    if (rpw > 64) rpw = 64;
    dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
    // with a scratch buffer the blocks' column sums take two steps (rows of partial sums, then colpartial_reduce_kernel)
    float* partial = grid.x > 32 && scratch && aligned16(scratch) && scratch_floats >= (int64_t)grid.x * 2 * C ? scratch : nullptr;
    if (C == 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(256), 0, s, x, ldx, dy, lddy, rows, gamma, beta, relu, dx, lddx, dgamma, dbeta, (int)rpw, partial);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(256), 0, s, x, ldx, dy, lddy, rows, gamma, beta, relu, dx, lddx, dgamma, dbeta, (int)rpw, partial);
    VRD_LAUNCH_CHECK();
    if (partial) {
        hipLaunchKernelGGL(colpartial_reduce_kernel, dim3((unsigned)(2 * C / 64), (grid.x + 31) / 32), dim3(256), 0, s, partial, (int)grid.x, 2 * C,
                           dgamma, dbeta, C);
        VRD_LAUNCH_CHECK();
    }
    return 0;
}

int vrd_dwconv_bwd(const vrd_dwconv_bwd_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->dx, "vrd_dwconv_bwd: null args");
    VRD_CHECK_ARG(a->n_out >= 1 && a->n_out <= 3 && a->B > 0 && a->Tin > 0 && a->C > 0, "vrd_dwconv_bwd: bad sizes");
    VRD_CHECK_ARG((a->ksize == 1 || a->ksize == 3) && (a->stride == 1 || a->stride == 2) && (a->group_in == 1 || a->group_in == 2) &&
                      a->Tin % a->stride == 0,
                  "vrd_dwconv_bwd: ksize 1/3, stride 1/2, group_in 1/2, Tin %% stride == 0");
    VRD_CHECK_ARG(!a->dx_up || (a->Tin % 2 == 0 && a->lddx_up >= (int64_t)a->C * a->group_in), "vrd_dwconv_bwd: bad dx_up");
    VRD_CHECK_ARG(a->lddx >= (int64_t)a->C * a->group_in, "vrd_dwconv_bwd: lddx too small");
    DwBwdArgs p{};
    for (int o = 0; o < a->n_out; ++o) {
        VRD_CHECK_ARG(a->dD[o] && a->w[o] && a->lddd[o] >= a->C, "vrd_dwconv_bwd: bad set %d", o);
        p.dD[o] = a->dD[o], p.lddd[o] = a->lddd[o], p.w[o] = a->w[o];
    }
    p.n_out = a->n_out, p.B = a->B, p.Tin = a->Tin, p.C = a->C, p.ksize = a->ksize, p.stride = a->stride, p.gin = a->group_in;
    p.mask_out = a->mask_out, p.dx = a->dx, p.lddx = a->lddx, p.dx_up = a->dx_up, p.lddx_up = a->lddx_up;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)a->B * (a->dx_up ? a->Tin / 2 : a->Tin) * a->C * a->group_in;
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)a->B * a->Tin * a->C * (a->group_in + a->n_out));
    bool vec = a->ksize == 3 && a->stride == 1 && a->group_in == 1 && !a->dx_up && a->C % 4 == 0 && a->lddx % 4 == 0 && aligned16(a->dx);
    for (int o = 0; o < a->n_out; ++o) vec = vec && a->lddd[o] % 4 == 0 && aligned16(a->dD[o]) && aligned16(a->w[o]);
    if (vec) {
        const dim3 grid((unsigned)((n / 4 + 255) / 256));
        if (a->n_out == 3) hipLaunchKernelGGL(dwconv_bwd_vec_kernel<3>, grid, dim3(256), 0, s, p);
        else if (a->n_out == 2) hipLaunchKernelGGL(dwconv_bwd_vec_kernel<2>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(dwconv_bwd_vec_kernel<1>, grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(dwconv_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_local_attn_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* dO, int64_t lddo,
                       const uint8_t* mask, const float* rel_pe, int B, int T, int C, int n_head, int half_win,
                       float* dq, float* dk, float* dv, int64_t ldd, float* scratch, void* stream) {
    VRD_CHECK_ARG(q && k && v && dO && mask && dq && dk && dv && scratch, "vrd_local_attn_bwd: null pointer");
    VRD_CHECK_ARG(C == 512 && (n_head == 4 || n_head == 8), "vrd_local_attn_bwd: built for C = 512 with 4 or 8 heads");
    const int W = 2 * half_win + 1;
    VRD_CHECK_ARG(half_win >= 1 && W <= LA_WMAX, "vrd_local_attn_bwd: window %d not supported (max %d)", W, LA_WMAX);
    VRD_CHECK_ARG(ld % 4 == 0 && lddo % 4 == 0 && ldd % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(dO) &&
                      aligned16(dq) && aligned16(dk) && aligned16(dv),
                  "vrd_local_attn_bwd: rows must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)B * T;
    const float scale = 1.0f / sqrtf((float)(C / n_head));
    float* P = scratch;
    float* dS = scratch + rows * n_head * W;
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)rows * C * 8);
    dim3 grid((unsigned)((rows + 3) / 4));
    if (n_head == 4) {
        hipLaunchKernelGGL(local_attn_bwd_q_kernel<16>, grid, dim3(256), 0, s, q, k, v, ld, dO, lddo, mask, rel_pe, B, T, W, scale, dq, ldd, P, dS);
        hipLaunchKernelGGL(local_attn_bwd_kv_kernel<16>, grid, dim3(256), 0, s, q, ld, dO, lddo, B, T, W, scale, P, dS, dk, dv, ldd);
    } else {
        hipLaunchKernelGGL(local_attn_bwd_q_kernel<8>, grid, dim3(256), 0, s, q, k, v, ld, dO, lddo, mask, rel_pe, B, T, W, scale, dq, ldd, P, dS);
        hipLaunchKernelGGL(local_attn_bwd_kv_kernel<8>, grid, dim3(256), 0, s, q, ld, dO, lddo, B, T, W, scale, P, dS, dk, dv, ldd);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_attn_bwd_probs(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dO, int64_t lddo,
                       const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, int head_dim, float* P, float* dS, void* stream) {
    VRD_CHECK_ARG(q && k && v && dO && P && dS, "vrd_attn_bwd_probs: null pointer");
    VRD_CHECK_ARG(B > 0 && Tq > 0 && Tk > 0 && Tk <= AB_TK_MAX && n_head > 0 && head_dim > 0 && head_dim <= AB_HD_MAX && head_dim % 4 == 0,
                  "vrd_attn_bwd_probs: Tk <= %d, head_dim <= %d and %% 4 == 0 (got Tk %d, head_dim %d)", AB_TK_MAX, AB_HD_MAX, Tk, head_dim);
    VRD_CHECK_ARG(ldkv % 4 == 0 && aligned16(k) && aligned16(v), "vrd_attn_bwd_probs: k / v rows must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * n_head * Tq;
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 4.0 * (double)n * Tk * head_dim, 8.0 * (double)n * Tk);
    hipLaunchKernelGGL(attn_bwd_probs_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, q, ldq, k, v, ldkv, dO, lddo, kv_mask, B, Tq, Tk,
                       n_head, head_dim, 1.0f / sqrtf((float)head_dim), P, dS);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_bmm(const vrd_bmm_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->A && a->B && a->C, "vrd_bmm: null pointer");
    VRD_CHECK_ARG(a->Z0 > 0 && a->Z1 > 0 && a->M > 0 && a->N > 0 && a->K > 0 && (int64_t)a->Z0 * a->Z1 <= 65535 && (a->M + 3) / 4 <= 65535,
                  "vrd_bmm: bad sizes");
    BmmArgs p;
    p.A = a->A, p.B = a->B, p.C = a->C;
    p.a0 = a->a_z0, p.a1 = a->a_z1, p.ai = a->a_row, p.ak = a->a_col;
    p.b0 = a->b_z0, p.b1 = a->b_z1, p.bk = a->b_row, p.bn = a->b_col;
    p.c0 = a->c_z0, p.c1 = a->c_z1, p.ci = a->c_row, p.cn = a->c_col;
    p.Z0 = a->Z0, p.Z1 = a->Z1, p.M = a->M, p.N = a->N, p.K = a->K, p.alpha = a->alpha, p.accumulate = a->accumulate;
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 2.0 * a->Z0 * a->Z1 * (double)a->M * a->N * a->K, 0.0);
    // matrix-core tiles once a 64 x 64 tile is at least a quarter full and K fills an LDS step (the predictor's 9-query
    // attention and the mask head's Q-row products stay on the one-thread-per-output kernel: a tile would be mostly padding)
    static const int mfma_env = [] { const char* e = getenv("VRD_BMM_MFMA"); return e ? atoi(e) : 1; }();
    if (mfma_env && a->M >= 32 && a->N >= 32 && a->K >= 16 && (a->M + 63) / 64 <= 65535) {
        auto al = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; };
        auto m4 = [](int64_t v) { return v % 4 == 0; };
        // 16-byte loads: bases aligned and every stride that is not the unit one a multiple of 4 floats
        const bool vec = al(a->A) && al(a->B) && m4(a->a_z0) && m4(a->a_z1) && m4(a->b_z0) && m4(a->b_z1) &&
                         (a->a_col == 1 ? m4(a->a_row) : a->a_row == 1 && m4(a->a_col)) &&
                         (a->b_col == 1 ? m4(a->b_row) : a->b_row == 1 && m4(a->b_col));
        const dim3 grid((a->N + 63) / 64, (a->M + 63) / 64, a->Z0 * a->Z1);
        if (vec) hipLaunchKernelGGL(bmm_mfma_kernel<true>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(bmm_mfma_kernel<false>, grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL(bmm_kernel, dim3((a->N + 63) / 64, (a->M + 3) / 4, a->Z0 * a->Z1), dim3(256), 0, s, p);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_attn_bwd_softmax(float* P, float* dS, const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, void* stream) {
    VRD_CHECK_ARG(P && dS && B > 0 && Tq > 0 && Tk > 0 && Tk <= AB_TK_MAX && n_head > 0, "vrd_attn_bwd_softmax: bad arguments (Tk <= %d)", AB_TK_MAX);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * n_head * Tq;
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 16.0 * (double)n * Tk);
    hipLaunchKernelGGL(attn_bwd_softmax_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, P, dS, kv_mask, n, Tq, Tk, n_head);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_maxpool_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int B, int Tin, int C, const uint8_t* mask_in, float* dx,
                    int64_t lddx, void* stream) {
    VRD_CHECK_ARG(x && dy && mask_in && dx && B > 0 && Tin > 0 && Tin % 2 == 0 && C > 0, "vrd_maxpool_bwd: bad arguments");
    VRD_CHECK_ARG(ldx >= C && lddy >= C && lddx >= C, "vrd_maxpool_bwd: leading dimension too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * Tin * C;
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 4.0 * (double)n * 2.5);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, dy, lddy, B, Tin, C, mask_in, dx, lddx);
    VRD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
