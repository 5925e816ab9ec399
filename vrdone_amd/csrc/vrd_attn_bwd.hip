// Global masked attention, backward, without the (B, H, Tq, Tk) matrices (round 4).
//
// Autograd of models/local_transformer.py:163-183 (same math as :44-63).  The five-product form in vrd_backward.hip
// (S = Q K^T and dP = dO V^T by vrd_bmm, softmax / dS row by row, then dQ, dK, dV by three more vrd_bmm) writes and reads two
// (B, H, Tq, Tk) matrices -- 805 MB each at vidor.yaml's 48 pairs x 8 heads x 512 x 512 -- and is bound by them.  Here the
// scores are recomputed tile by tile, flash style, in the split precision of the other backward GEMMs (x = bf16 hi + bf16 lo,
// three MFMA products per product, f32 accumulate: vrd_gemm_x3.hip), in two kernels that mirror the two products of the
// forward kernel (vrd_attn_x3.hip):
//
//   attn_bwd_dq_kernel    a wave owns 32 queries (query = lane).  Pass 1 over the key tiles: S^T = K Q^T -> log-sum-exp of
//                         every query (the forward does not keep it).  Pass 2: S^T again, P^T = exp(S^T - lse),
//                         dP^T = V dO^T, dS^T = P^T o (dP^T - delta) * scale, dQ^T += K^T dS^T (K^T fragments by transposed
//                         LDS reads, dS^T straight from the accumulator registers).  delta_i = sum_d dO_id O_id.
//                         Writes dq, and lse / delta for the second kernel.
//   attn_bwd_dkv_kernel   a wave owns 32 keys (key = lane).  Over the query tiles: S = Q K^T, P = exp(S - lse), dP = dO V^T,
//                         dS = P o (dP - delta) * scale, dV^T += dO^T P, dK^T += Q^T dS.
//
// Operands arrive as f32 rows (the training path keeps every activation f32) and are split into bf16 hi / lo planes while a
// tile is staged into LDS; a tile that is read both as MFMA rows and through transposed reads is stored twice, once per
// swizzle.  head_dim 64 (vidor.yaml, vidor_x.yaml: 8 heads of 64); other shapes keep the five-product form.
#include "vrd_common.h"
#include <cmath>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef vrd::bf16x8_t bf16x8;
typedef vrd::bf16x4_t bf16x4;
template <bool F16> using e16x8 = typename vrd::SplitFmt<F16>::x8;
template <bool F16> using e16x4 = typename vrd::SplitFmt<F16>::x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr int HD = 64;
constexpr int KS = HD / 16;                  // k16 steps over head_dim
constexpr int DT = HD / 32;                  // 32-row d tiles
constexpr int ROWB = HD * 2;                 // bytes per tile row of one bf16 plane
constexpr int CPR = ROWB / 16;               // 16-byte chunks per row
constexpr int PLANE = 32 * ROWB;             // 4 KiB
constexpr int NW = 4;                        // waves per workgroup
constexpr size_t DQ_LDS = 2 * 6 * PLANE + 2 * 32 * 4, DKV_LDS = 2 * 8 * PLANE + 2 * 64 * 4, FWD_LDS = 2 * 4 * PLANE + 2 * 32 * 4;

// chunk swizzles: rswz for planes read row-wise (one 16-byte chunk per lane, lanes = rows), tswz for planes read through
// ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group); both conflict free for 128-byte rows
__device__ constexpr int rswz(int row) { return row % CPR; }
__device__ constexpr int tswz(int row) { return (((row & 3) << 2) % CPR) ^ (((row >> 1) & 1) << 2); }

// 32 rows x 64 channels of head h, rows t0 .. t0+31 of a (B, T, ld) f32 matrix -> bf16 hi / lo planes.  ROW / TR: which
// swizzled copies are written (bytes: hi at `row_hi`, lo PLANE behind it; likewise tr).  Rows >= T: zeros.  256 threads.
// The two halves of staging a tile: the global loads (issued a tile ahead, so that they fly under the current tile's MFMAs) ...
struct TileRegs {
    float4 v[2];
};
__device__ __forceinline__ TileRegs fetch_tile(const float* __restrict__ src, int64_t ld, int t0, int T) {
    TileRegs t;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = threadIdx.x + 256 * i;
        const int r = p >> 4, c = (p & 15) * 4;
        t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + r < T) t.v[i] = *reinterpret_cast<const float4*>(src + (int64_t)(t0 + r) * ld + c);
    }
    return t;
}
// ... and the hi / lo split into the swizzled planes
// (mul: the factor of the f16 format's planes when it is not the activations' fixed 2^VRD_F16_ACT_EXP: gradient operands)
template <bool ROW, bool TR, bool F16 = false>
__device__ __forceinline__ void commit_tile(const TileRegs& t, char* row_hi, char* tr_hi, vrd::RangeTrack* rt = nullptr,
                                            float mul = vrd::F16_ACT_SCALE) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = threadIdx.x + 256 * i;
        const int r = p >> 4, c = (p & 15) * 4;
        const float4 v = t.v[i];
        const float x[4] = {v.x, v.y, v.z, v.w};
        e16x4<F16> h, l;
        vrd::split_n_scaled<F16>(x, mul, h, l, rt);
        const int within = (c & 7) * 2;
        if (ROW) {
            const int off = r * ROWB + (((c >> 3) ^ rswz(r)) * 16) + within;
            *reinterpret_cast<e16x4<F16>*>(row_hi + off) = h;
            *reinterpret_cast<e16x4<F16>*>(row_hi + PLANE + off) = l;
        }
        if (TR) {
            const int off = r * ROWB + (((c >> 3) ^ tswz(r)) * 16) + within;
            *reinterpret_cast<e16x4<F16>*>(tr_hi + off) = h;
            *reinterpret_cast<e16x4<F16>*>(tr_hi + PLANE + off) = l;
        }
    }
}

// fragment (rows = lanes li, k = 16 s + 8 lh + 0..7) of a row plane
template <bool F16 = false>
__device__ __forceinline__ e16x8<F16> row_frag(const char* plane, int li, int lh, int s) {
    return *reinterpret_cast<const e16x8<F16>*>(plane + li * ROWB + (((2 * s + lh) ^ rswz(li)) * 16));
}

// transposed fragment of a tr plane: A[row = column 32 d + li of the tile][k = tile rows in the accumulator order
// 16 s + 8 (j >> 2) + 4 lh + (j & 3)] -- the order in which registers 8s .. 8s+7 of a 32 x 32 accumulator hold its rows
template <bool F16 = false>
__device__ __forceinline__ e16x8<F16> tr_frag(const char* plane, int lane, int s, int d) {
    const int lh = lane >> 5, vq = (lane >> 2) & 3, vp = lane & 3;
    const int col = 32 * d + 16 * ((lane >> 4) & 1) + 4 * vp;
    s16x8 r;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int row = 16 * s + 8 * part + 4 * lh + vq;
        const int off = row * ROWB + ((((col * 2) >> 4) ^ tswz(row)) * 16) + ((col * 2) & 15);
        const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(plane + off));
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * part + j] = t[j];
    }
    return __builtin_bit_cast(e16x8<F16>, r);
}

// lane (li, lh): elements d = 16 s + 8 lh + 0..7 of row `row` of head h (f32) as hi / lo fragments (the B operand of a product
// whose columns are this wave's 32 rows)
template <bool F16 = false>
__device__ __forceinline__ void load_col_frags(const float* __restrict__ rowp, int lh, bool ok, e16x8<F16> (&hi)[KS], e16x8<F16> (&lo)[KS],
                                               vrd::RangeTrack* rt = nullptr, float mul = vrd::F16_ACT_SCALE) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const float4 a = *reinterpret_cast<const float4*>(rowp + 16 * s + 8 * lh);
            const float4 b = *reinterpret_cast<const float4*>(rowp + 16 * s + 8 * lh + 4);
            x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
        }
        vrd::split_n_scaled<F16>(x, mul, hi[s], lo[s], rt);
    }
}

// acc += A . B in split precision
template <typename V>
__device__ __forceinline__ f32x16 mfma3(V ah, V al, V bh, V bl, f32x16 acc) {
    acc = vrd::mfma32(al, bh, acc);
    acc = vrd::mfma32(ah, bl, acc);
    return vrd::mfma32(ah, bh, acc);
}

// registers 8 s .. 8 s + 7 of a 32 x 32 accumulator as the hi / lo B fragment of k16 step s
template <bool F16 = false>
__device__ __forceinline__ void split_acc(const f32x16& a, int s, e16x8<F16>& hi, e16x8<F16>& lo, float mul = vrd::F16_ACT_SCALE) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = a[8 * s + j];
    vrd::split_n_scaled<F16>(x, mul, hi, lo);
}

__device__ __forceinline__ int acc_row(int e, int lh) { return (e & 3) + 8 * (e >> 2) + 4 * lh; }

// ---------------------------------------------------------------------------------------------------------------------
// Forward of a training step on the same building blocks (f32 rows in, split products, f32 rows + log-sum-exp out): what
// attn_flash_x3_kernel does on pair rows.  F16: the f16x3 mode's planes (values times 2^VRD_F16_ACT_EXP: the scores come out
// 2^(2e) too large, P is split as P 2^e, and O / l carries 2^(2e) once).
template <bool F16>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_rows_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                                  const float* __restrict__ v, int64_t ldkv,
                                                                  const uint8_t* __restrict__ kv_mask, int Tq, int Tk, int n_head,
                                                                  float scale, float* __restrict__ out, int64_t ldo,
                                                                  float* __restrict__ lse_out, unsigned* rflag) {
    extern __shared__ __attribute__((aligned(16))) char lds[];        // 2 stages x (k row h/l, v tr h/l) + key bias: FWD_LDS
    vrd::RangeTrack rt;                                               // q / k / v are f32 rows, split here (vrd_common.h)
    float* const kbias = reinterpret_cast<float*>(lds + 2 * 4 * PLANE);
    const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tq = (qblk * NW + wave) * 32 + li;
    const bool q_ok = tq < Tq;
    const float* const kb = k + (int64_t)b * Tk * ldkv + h * HD;
    const float* const vb = v + (int64_t)b * Tk * ldkv + h * HD;
    const uint8_t* const mk = kv_mask ? kv_mask + (int64_t)b * Tk : nullptr;
    constexpr float s_in = F16 ? vrd::F16_ACT_INV * vrd::F16_ACT_INV : 1.0f;          // undoes the operand scaling of a product
    e16x8<F16> qh[KS], ql[KS];
    load_col_frags<F16>(q + ((int64_t)b * Tq + (q_ok ? tq : 0)) * ldq + h * HD, lh, q_ok, qh, ql, &rt);
    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    TileRegs rk, rv;
    float rbias = 0.f;
    const int nkt = (Tk + 31) / 32;
    auto fetch = [&](int kt) {
        rk = fetch_tile(kb, ldkv, kt * 32, Tk);
        rv = fetch_tile(vb, ldkv, kt * 32, Tk);
        if (threadIdx.x < 32) {
            const int key = kt * 32 + threadIdx.x;
            rbias = (key < Tk && (!mk || mk[key])) ? 0.f : -INFINITY;
        }
    };
    auto commit = [&](int buf) {
        char* st = lds + buf * 4 * PLANE;
        commit_tile<true, false, F16>(rk, st, nullptr, &rt);
        commit_tile<false, true, F16>(rv, nullptr, st + 2 * PLANE, &rt);
        if (threadIdx.x < 32) kbias[buf * 32 + threadIdx.x] = rbias;
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1);
        const char* st = lds + (kt & 1) * 4 * PLANE;
        const float* kbs = kbias + (kt & 1) * 32;
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) s = mfma3(row_frag<F16>(st, li, lh, t), row_frag<F16>(st + PLANE, li, lh, t), qh[t], ql[t], s);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = s[e] * (scale * s_in) + kbs[acc_row(e, lh)];
            mx = fmaxf(mx, s[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = __expf(s[e] - m_use);
            ps += s[e];
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            e16x8<F16> ph, pl;
            split_acc<F16>(s, s2, ph, pl);
#pragma unroll
            for (int d = 0; d < DT; ++d)
                oacc[d] = mfma3(tr_frag<F16>(st + 2 * PLANE, lane, s2, d), tr_frag<F16>(st + 3 * PLANE, lane, s2, d), ph, pl, oacc[d]);
        }
        if (kt + 1 < nkt) commit((kt + 1) & 1);
        __syncthreads();
    }
    if (q_ok) {
        const float inv = s_in / l_run;
        float* const orow = out + ((int64_t)b * Tq + tq) * ldo + h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(orow + 32 * d + 8 * g + 4 * lh) =
                    make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
        if (lh == 0) lse_out[((int64_t)b * n_head + h) * Tq + tq] = l_run > 0.f ? m_run + __logf(l_run) : 0.f;
    }
    if (F16) rt.report(rflag, vrd::RANGE_GEMM_IN);
}

// ---------------------------------------------------------------------------------------------------------------------
// lse_in: the forward's log-sum-exp (attn_fwd_rows_kernel), or NULL: pass 1 recomputes it
// F16 (the f16x3 mode): every operand as f16 planes of a power-of-two multiple -- q, k, v, P at the activations' 2^VRD_F16_ACT_EXP
// (they went through the forward's range check), dO at s_o = o_scale[0] (vrd_absmax_scale of dO), dS at s_s = s_o s_v 2^-17 with
// s_v = v_scale[0] (vrd_absmax_scale of V): |dS| <= (|dP| + |delta|) / 8 <= 16 max|dO| max|V| < 2^32 / (s_o s_v), so dS s_s stays
// below 2^15; what the bound gives away against the values that actually occur costs nothing but range, the planes keep 2^-25
// absolute precision down to their subnormals.  The accumulators are rescaled where they leave the kernel.
template <bool F16>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                                const float* __restrict__ v, int64_t ldkv, const float* __restrict__ o,
                                                                const float* __restrict__ dO, int64_t ldo,
                                                                const uint8_t* __restrict__ kv_mask, int Tq, int Tk, int n_head,
                                                                float scale, float* __restrict__ dq, const float* __restrict__ lse_in,
                                                                float* __restrict__ lse_out, float* __restrict__ delta_out,
                                                                const float* __restrict__ o_scale, const float* __restrict__ v_scale) {
    typedef e16x8<F16> bf16x8;                                         // (this instantiation's eight 16-bit elements)
    constexpr float a_inv = F16 ? vrd::F16_ACT_INV : 1.0f;
    const float s_o = F16 ? vrd::uniform_load(o_scale) : 1.f, s_o_inv = F16 ? vrd::uniform_load(o_scale + 1) : 1.f;
    const float s_s = F16 ? s_o * vrd::uniform_load(v_scale) * 0x1p-17f : 1.f;
    const float s_s_inv = F16 ? s_o_inv * vrd::uniform_load(v_scale + 1) * 0x1p17f : 1.f;
    extern __shared__ __attribute__((aligned(16))) char lds[];        // 2 stages x (k row h/l, k tr h/l, v row h/l) + key bias: DQ_LDS
    float* const kbias = reinterpret_cast<float*>(lds + 2 * 6 * PLANE);               // [2][32]
    const int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tq = (qblk * NW + wave) * 32 + li;
    const bool q_ok = tq < Tq;
    const float* const qrow = q + ((int64_t)b * Tq + (q_ok ? tq : 0)) * ldq + h * HD;
    const float* const orow = o + ((int64_t)b * Tq + (q_ok ? tq : 0)) * ldo + h * HD;
    const float* const drow = dO + ((int64_t)b * Tq + (q_ok ? tq : 0)) * ldo + h * HD;
    const float* const kb = k + (int64_t)b * Tk * ldkv + h * HD;
    const float* const vb = v + (int64_t)b * Tk * ldkv + h * HD;
    const uint8_t* const mk = kv_mask ? kv_mask + (int64_t)b * Tk : nullptr;

    bf16x8 qh[KS], ql[KS], gh[KS], gl[KS];
    load_col_frags<F16>(qrow, lh, q_ok, qh, ql);
    load_col_frags<F16>(drow, lh, q_ok, gh, gl, nullptr, s_o);
    // delta = sum_d dO O over the query's row: this lane holds half of the d's
    float delta = 0.f;
    if (q_ok) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += drow[16 * s + 8 * lh + j] * orow[16 * s + 8 * lh + j];
    }
    delta += __shfl_xor(delta, 32, 64);

    const int nkt = (Tk + 31) / 32;
    TileRegs rk, rv;
    float rbias = 0.f;
    auto fetch = [&](int kt, bool second) {
        rk = fetch_tile(kb, ldkv, kt * 32, Tk);
        if (second) rv = fetch_tile(vb, ldkv, kt * 32, Tk);
        if (threadIdx.x < 32) {
            const int key = kt * 32 + threadIdx.x;
            rbias = (key < Tk && (!mk || mk[key])) ? 0.f : -INFINITY;
        }
    };
    auto commit = [&](int buf, bool second) {
        char* st = lds + buf * 6 * PLANE;
        if (second) {
            commit_tile<true, true, F16>(rk, st, st + 2 * PLANE);
            commit_tile<true, false, F16>(rv, st + 4 * PLANE, nullptr);
        } else {
            commit_tile<true, false, F16>(rk, st, nullptr);
        }
        if (threadIdx.x < 32) kbias[buf * 32 + threadIdx.x] = rbias;
    };
    auto scores = [&](const char* st, const float* kbs) {
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) s = mfma3(row_frag<F16>(st, li, lh, t), row_frag<F16>(st + PLANE, li, lh, t), qh[t], ql[t], s);
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = s[e] * (scale * a_inv * a_inv) + kbs[acc_row(e, lh)];
        return s;
    };

    // ---- pass 1: log-sum-exp of every query over the valid keys (unless the forward kept it)
    float m_run = -INFINITY, l_run = 0.f;
    if (!lse_in) {
    fetch(0, false);
    commit(0, false);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1, false);
        const f32x16 s = scores(lds + (kt & 1) * 6 * PLANE, kbias + (kt & 1) * 32);
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        float ps = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) ps += __expf(s[e] - m_use);
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * __expf(m_run - m_use) + ps;
        m_run = m_new;
        if (kt + 1 < nkt) commit((kt + 1) & 1, false);          // (the other buffer: last read before the previous barrier)
        __syncthreads();
    }
    }
    // (a query that sees no valid key: every p below is exp(-inf - 0) = 0)
    const float lse = lse_in ? (q_ok ? lse_in[((int64_t)b * n_head + h) * Tq + tq] : 0.f) : (l_run > 0.f ? m_run + __logf(l_run) : 0.f);

    // ---- pass 2: dQ^T += K^T . dS^T
    f32x16 dqa[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dqa[d][e] = 0.f;
    fetch(0, true);
    commit(0, true);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1, true);
        const char* st = lds + (kt & 1) * 6 * PLANE;
        f32x16 s = scores(st, kbias + (kt & 1) * 32);
        f32x16 dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t)
            dp = mfma3(row_frag<F16>(st + 4 * PLANE, li, lh, t), row_frag<F16>(st + 5 * PLANE, li, lh, t), gh[t], gl[t], dp);
        const float dp_unscale = a_inv * s_o_inv;              // (dP came out of V 2^a times dO s_o)
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = __expf(s[e] - lse) * (dp[e] * dp_unscale - delta) * scale;            // dS^T (scaled)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 dh, dl;
            split_acc<F16>(s, s2, dh, dl, s_s);
#pragma unroll
            for (int d = 0; d < DT; ++d)
                dqa[d] = mfma3(tr_frag<F16>(st + 2 * PLANE, lane, s2, d), tr_frag<F16>(st + 3 * PLANE, lane, s2, d), dh, dl, dqa[d]);
        }
        if (kt + 1 < nkt) commit((kt + 1) & 1, true);
        __syncthreads();
    }
    if (q_ok) {
        float* const out = dq + ((int64_t)b * Tq + tq) * ldq + h * HD;
        const float u = a_inv * s_s_inv;         // (K 2^a times dS s_s)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g)          // registers 4g .. 4g+3 are d = 32 d + 8 g + 4 lh + 0..3
                *reinterpret_cast<float4*>(out + 32 * d + 8 * g + 4 * lh) =
                    make_float4(dqa[d][4 * g] * u, dqa[d][4 * g + 1] * u, dqa[d][4 * g + 2] * u, dqa[d][4 * g + 3] * u);
        if (lh == 0) {
            lse_out[((int64_t)b * n_head + h) * Tq + tq] = lse;
            delta_out[((int64_t)b * n_head + h) * Tq + tq] = delta;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkv_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                                 const float* __restrict__ v, int64_t ldkv, const float* __restrict__ dO,
                                                                 int64_t ldo, const uint8_t* __restrict__ kv_mask, int Tq, int Tk,
                                                                 int n_head, float scale, const float* __restrict__ lse_in,
                                                                 const float* __restrict__ delta_in, float* __restrict__ dk,
                                                                 float* __restrict__ dv, const float* __restrict__ o_scale,
                                                                 const float* __restrict__ v_scale) {
    typedef e16x8<F16> bf16x8;
    constexpr float a_inv = F16 ? vrd::F16_ACT_INV : 1.0f;                       // (scales: see attn_bwd_dq_kernel)
    const float s_o = F16 ? vrd::uniform_load(o_scale) : 1.f, s_o_inv = F16 ? vrd::uniform_load(o_scale + 1) : 1.f;
    const float s_s = F16 ? s_o * vrd::uniform_load(v_scale) * 0x1p-17f : 1.f;
    const float s_s_inv = F16 ? s_o_inv * vrd::uniform_load(v_scale + 1) * 0x1p17f : 1.f;
    // 2 stages x (q row h/l, q tr h/l, dO row h/l, dO tr h/l) + lse / delta of the tile's 32 queries: DKV_LDS
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* const stat = reinterpret_cast<float*>(lds + 2 * 8 * PLANE);                // [2][lse 32 | delta 32]
    const int kblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int tk = (kblk * NW + wave) * 32 + li;
    const bool k_in = tk < Tk;
    const bool k_ok = k_in && (!kv_mask || kv_mask[(int64_t)b * Tk + tk]);
    const float* const krow = k + ((int64_t)b * Tk + (k_in ? tk : 0)) * ldkv + h * HD;
    const float* const vrow = v + ((int64_t)b * Tk + (k_in ? tk : 0)) * ldkv + h * HD;
    const float* const qb = q + (int64_t)b * Tq * ldq + h * HD;
    const float* const gb = dO + (int64_t)b * Tq * ldo + h * HD;
    const float* const lse_b = lse_in + ((int64_t)b * n_head + h) * Tq;
    const float* const del_b = delta_in + ((int64_t)b * n_head + h) * Tq;

    bf16x8 kh[KS], kl[KS], vh[KS], vl[KS];
    load_col_frags<F16>(krow, lh, k_in, kh, kl);
    load_col_frags<F16>(vrow, lh, k_in, vh, vl);
    f32x16 dka[DT], dva[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dka[d][e] = dva[d][e] = 0.f;

    const int nqt = (Tq + 31) / 32;
    TileRegs rq, rg;
    float rl = 0.f, rd = 0.f;
    auto fetch = [&](int qt) {
        rq = fetch_tile(qb, ldq, qt * 32, Tq);
        rg = fetch_tile(gb, ldo, qt * 32, Tq);
        if (threadIdx.x < 32) {
            const int t = qt * 32 + threadIdx.x;
            rl = t < Tq ? lse_b[t] : INFINITY;              // a query past Tq: p = exp(-inf) = 0
            rd = t < Tq ? del_b[t] : 0.f;
        }
    };
    auto commit = [&](int buf) {
        char* st = lds + buf * 8 * PLANE;
        commit_tile<true, true, F16>(rq, st, st + 2 * PLANE);
        commit_tile<true, true, F16>(rg, st + 4 * PLANE, st + 6 * PLANE, nullptr, s_o);
        if (threadIdx.x < 32) {
            stat[buf * 64 + threadIdx.x] = rl;
            stat[buf * 64 + 32 + threadIdx.x] = rd;
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
        if (qt + 1 < nqt) fetch(qt + 1);
        const char* st = lds + (qt & 1) * 8 * PLANE;
        const float* sl = stat + (qt & 1) * 64;
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = dp[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            s = mfma3(row_frag<F16>(st, li, lh, t), row_frag<F16>(st + PLANE, li, lh, t), kh[t], kl[t], s);                          // S = Q K^T
            dp = mfma3(row_frag<F16>(st + 4 * PLANE, li, lh, t), row_frag<F16>(st + 5 * PLANE, li, lh, t), vh[t], vl[t], dp);        // dP = dO V^T
        }
        f32x16 p;
        const float s_unscale = scale * a_inv * a_inv, dp_unscale = a_inv * s_o_inv;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = acc_row(e, lh);
            p[e] = k_ok ? __expf(s[e] * s_unscale - sl[r]) : 0.f;
            s[e] = p[e] * (dp[e] * dp_unscale - sl[32 + r]) * scale;                                                  // dS (scaled)
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl, dh, dl;
            split_acc<F16>(p, s2, ph, pl);
            split_acc<F16>(s, s2, dh, dl, s_s);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dva[d] = mfma3(tr_frag<F16>(st + 6 * PLANE, lane, s2, d), tr_frag<F16>(st + 7 * PLANE, lane, s2, d), ph, pl, dva[d]);      // dV^T += dO^T P
                dka[d] = mfma3(tr_frag<F16>(st + 2 * PLANE, lane, s2, d), tr_frag<F16>(st + 3 * PLANE, lane, s2, d), dh, dl, dka[d]);      // dK^T += Q^T dS
            }
        }
        if (qt + 1 < nqt) commit((qt + 1) & 1);
        __syncthreads();
    }
    if (k_in) {
        float* const ok_ = dk + ((int64_t)b * Tk + tk) * ldkv + h * HD;
        float* const ov = dv + ((int64_t)b * Tk + tk) * ldkv + h * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * d + 8 * g + 4 * lh;
                const float uk = a_inv * s_s_inv, uv = a_inv * s_o_inv;       // (Q 2^a times dS s_s; P 2^a times dO s_o)
                *reinterpret_cast<float4*>(ok_ + c) = make_float4(dka[d][4 * g] * uk, dka[d][4 * g + 1] * uk, dka[d][4 * g + 2] * uk, dka[d][4 * g + 3] * uk);
                *reinterpret_cast<float4*>(ov + c) = make_float4(dva[d][4 * g] * uv, dva[d][4 * g + 1] * uv, dva[d][4 * g + 2] * uv, dva[d][4 * g + 3] * uv);
            }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int vrd_attention_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* out,
                                 const float* dO, int64_t ldo, const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, int head_dim,
                                 float* dq, float* dk, float* dv, const float* lse, float* scratch, const float* o_scale,
                                 const float* v_scale, void* stream) {
    VRD_CHECK_ARG(q && k && v && out && dO && dq && dk && dv && scratch, "vrd_attention_bwd: null pointer");
    VRD_CHECK_ARG(head_dim == HD, "vrd_attention_bwd: built for head_dim %d (got %d)", HD, head_dim);
    VRD_CHECK_ARG((o_scale == nullptr) == (v_scale == nullptr), "vrd_attention_bwd: o_scale and v_scale go together");
    VRD_CHECK_ARG(B > 0 && B <= 65535 && n_head > 0 && n_head <= 65535 && Tq > 0 && Tk > 0, "vrd_attention_bwd: bad sizes");
    const int width = n_head * head_dim;
    VRD_CHECK_ARG(ldq >= width && ldkv >= width && ldo >= width && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && aligned16(q) &&
                      aligned16(k) && aligned16(v) && aligned16(out) && aligned16(dO) && aligned16(dq) && aligned16(dk) && aligned16(dv),
                  "vrd_attention_bwd: rows must be 16-byte aligned with leading dimensions >= n_head * head_dim");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    // (dq: two passes over the keys, three + two products; dk / dv: four products)
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 2.0 * 9.0 * B * (double)n_head * Tq * Tk * head_dim, 0.0);
    float* lse_own = scratch;
    float* delta = scratch + (int64_t)B * n_head * Tq;
    const dim3 gq((Tq + NW * 32 - 1) / (NW * 32), n_head, B), gk((Tk + NW * 32 - 1) / (NW * 32), n_head, B);
    if (o_scale) {
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_bwd_dq_kernel<true>), DQ_LDS, "vrd_attention_bwd(dq)")) return rc;
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<true>), DKV_LDS, "vrd_attention_bwd(dk, dv)")) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, gq, dim3(NW * 64), DQ_LDS, s, q, ldq, k, v, ldkv, out, dO, ldo, kv_mask, Tq, Tk, n_head, scale, dq,
                           lse, lse_own, delta, o_scale, v_scale);
        VRD_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, gk, dim3(NW * 64), DKV_LDS, s, q, ldq, k, v, ldkv, dO, ldo, kv_mask, Tq, Tk, n_head, scale,
                           lse ? lse : lse_own, delta, dk, dv, o_scale, v_scale);
    } else {
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_bwd_dq_kernel<false>), DQ_LDS, "vrd_attention_bwd(dq)")) return rc;
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<false>), DKV_LDS, "vrd_attention_bwd(dk, dv)")) return rc;
        hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, gq, dim3(NW * 64), DQ_LDS, s, q, ldq, k, v, ldkv, out, dO, ldo, kv_mask, Tq, Tk, n_head, scale, dq,
                           lse, lse_own, delta, (const float*)nullptr, (const float*)nullptr);
        VRD_LAUNCH_CHECK();
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, gk, dim3(NW * 64), DKV_LDS, s, q, ldq, k, v, ldkv, dO, ldo, kv_mask, Tq, Tk, n_head, scale,
                           lse ? lse : lse_own, delta, dk, dv, (const float*)nullptr, (const float*)nullptr);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vrd_attention_rows(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask, int B,
                                  int Tq, int Tk, int n_head, int head_dim, int fmt, float* out, int64_t ldo, float* lse, void* stream) {
    VRD_CHECK_ARG(q && k && v && out && lse, "vrd_attention_rows: null pointer");
    VRD_CHECK_ARG(head_dim == HD, "vrd_attention_rows: built for head_dim %d (got %d)", HD, head_dim);
    VRD_CHECK_ARG(fmt == VRD_PAIR_BF16 || fmt == VRD_PAIR_F16, "vrd_attention_rows: fmt must be VRD_PAIR_BF16 or VRD_PAIR_F16");
    VRD_CHECK_ARG(B > 0 && B <= 65535 && n_head > 0 && n_head <= 65535 && Tq > 0 && Tk > 0, "vrd_attention_rows: bad sizes");
    const int width = n_head * head_dim;
    VRD_CHECK_ARG(ldq >= width && ldkv >= width && ldo >= width && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && aligned16(q) &&
                      aligned16(k) && aligned16(v) && aligned16(out),
                  "vrd_attention_rows: rows must be 16-byte aligned with leading dimensions >= n_head * head_dim");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    vrd::ProfScope prof(VRD_K_ATTN_FLASH, s, 4.0 * B * (double)n_head * Tq * Tk * head_dim, 4.0 * B * (double)width * (2.0 * Tq + 2.0 * Tk));
    const dim3 grid((Tq + NW * 32 - 1) / (NW * 32), n_head, B);
    if (fmt == VRD_PAIR_F16) {
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_fwd_rows_kernel<true>), FWD_LDS, "vrd_attention_rows")) return rc;
        hipLaunchKernelGGL(attn_fwd_rows_kernel<true>, grid, dim3(NW * 64), FWD_LDS, s, q, ldq, k, v, ldkv, kv_mask, Tq, Tk, n_head, scale, out,
                           ldo, lse, vrd::range_flag());
    } else {
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(attn_fwd_rows_kernel<false>), FWD_LDS, "vrd_attention_rows")) return rc;
        hipLaunchKernelGGL(attn_fwd_rows_kernel<false>, grid, dim3(NW * 64), FWD_LDS, s, q, ldq, k, v, ldkv, kv_mask, Tq, Tk, n_head, scale, out,
                           ldo, lse, nullptr);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}
