// Attention kernels of the relation-encoding path.
//
//  * local_attn_strip_kernel / local_attn_kernel
//                        banded window attention of the stem / branch blocks.  O(T*w) work and
//                        HBM-bound: one wavefront per strip of 16 query rows (K / V window rows in a
//                        register ring) or per query row, covering all heads; window scores in
//                        registers, head-wise dot products reduced over the 8 or 16 lanes of a head.
//  * attn_small_kernel   generic masked attention on the VALU (any Tq/Tk/head_dim <= 128);
//                        used for the predictor's 9-query decoder.
//  * attn_flash_kernel   global masked attention (the SOS self/cross attention) on the f32
//                        MFMA, flash style: never materialises the Tq x Tk scores.
#include "vrd_common.h"
#include <cmath>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ------------------------------------------------------------------------------------------
// banded attention.  C = 512: lane l owns channels [8l, 8l+8); GROUP = head_dim / 8 lanes per head
// ------------------------------------------------------------------------------------------
// REL: the learnable per-(head, window slot) bias `rel_pe` (n_head x W) is added to the scaled scores before the key mask
// (blocks.py:957-958); a separate instantiation, so that the default path carries no extra instruction
template <int W, int GROUP, bool REL>
__global__ __launch_bounds__(256) void local_attn_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int64_t ld,
                                                         const uint8_t* __restrict__ mask, const float* __restrict__ rel,
                                                         int B, int T, float scale,
                                                         float* __restrict__ out, int64_t ldo, int pair, unsigned* rflag) {
    constexpr int HW = W / 2;
    const int lane = threadIdx.x & 63;
    // XCD-aware renumbering (as in the GEMM / flash kernels): consecutive workgroups are dealt round-robin to the
    // eight XCDs, and a row's K / V neighbours are re-read by the workgroups next to it -- a contiguous range of rows
    // per XCD keeps those re-reads in one L2 instead of fetching them from HBM once per XCD
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    const int64_t row = (int64_t)lid * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: addresses on the scalar unit
    if (row >= (int64_t)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (int64_t)b * T);
    float* o = out + row * ldo + lane * 8;
    if (!mask[row]) {      // masked query rows are zeroed after the softmax (blocks.py:977-978); 0 is 0 in pair rows too
        st4(o, make_float4(0.f, 0.f, 0.f, 0.f));
        st4(o + 4, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    float4 q0 = ld4(q + row * ld + lane * 8), q1 = ld4(q + row * ld + lane * 8 + 4);
    q0.x *= scale; q0.y *= scale; q0.z *= scale; q0.w *= scale;
    q1.x *= scale; q1.y *= scale; q1.z *= scale; q1.w *= scale;
    float s[W];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int tj = t + j - HW;
        if (tj < 0 || tj >= T) { s[j] = -INFINITY; continue; }
        const float* kr = k + (row + j - HW) * ld + lane * 8;
        float d = dot4(q0, ld4(kr)) + dot4(q1, ld4(kr + 4));
        d = vrd::group_sum<GROUP>(d);
        if (REL) d += rel[(lane / GROUP) * W + j];
        s[j] = d + (mask[row + j - HW] ? 0.f : -1e4f);
        m = fmaxf(m, s[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < W; ++j) { s[j] = __expf(s[j] - m); den += s[j]; }
    const float inv = 1.0f / den;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int tj = t + j - HW;
        if (tj < 0 || tj >= T) continue;
        const float* vr = v + (row + j - HW) * ld + lane * 8;
        const float4 v0 = ld4(vr), v1 = ld4(vr + 4);
        const float pj = s[j] * inv;
        a0.x += pj * v0.x; a0.y += pj * v0.y; a0.z += pj * v0.z; a0.w += pj * v0.w;
        a1.x += pj * v1.x; a1.y += pj * v1.y; a1.z += pj * v1.z; a1.w += pj * v1.w;
    }
    if (pair) {
        vrd::RangeTrack rt;           // (q, k, v are f32 rows here: nothing upstream has checked their range)
        vrd::store_pair4(out + row * ldo, lane * 8, 512, a0, pair, &rt);
        vrd::store_pair4(out + row * ldo, lane * 8 + 4, 512, a1, pair, &rt);
        rt.report(rflag, vrd::RANGE_ATTN_OUT);
    } else {
        st4(o, a0);
        st4(o + 4, a1);
    }
}

// Strip variant of the banded attention: one wave walks RW consecutive query rows of one sequence and keeps the K and V
// rows of the moving window in registers (a ring of W + 1 slots, the row entering the window next already requested),
// so every K / V row is read W + 1 -> (RW + W - 1) / RW times from the L1 instead of W times.  The ring is indexed with
// compile-time slots: the row loop runs in rounds of R = W + 1 unrolled phases.
template <int W, int GROUP, int RW, bool REL>
__global__ __launch_bounds__(256) void local_attn_strip_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, int64_t ld,
                                                               const uint8_t* __restrict__ mask, const float* __restrict__ rel,
                                                               int B, int T_u, int strips_per_seq,
                                                               float scale, float* __restrict__ out, int64_t ldo, int pair,
                                                               unsigned* rflag, vrd::SegTable sg) {
    constexpr int HW = W / 2, R = W + 1;
    vrd::RangeTrack rt;
    const int lane = threadIdx.x & 63;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    const int64_t ws = (int64_t)lid * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int T, t0;
    int64_t row_b;
    if (sg.count) {                                  // ragged row space: groups of sequences of different lengths
        int g, b, strip;
        if (!vrd::seg_find(sg, ws, g, b, strip)) return;
        T = sg.T[g], t0 = strip * RW, row_b = sg.row[g] + (int64_t)b * T;
    } else {
        const int b = (int)(ws / strips_per_seq);
        if (b >= B) return;
        T = T_u, t0 = (int)(ws - (int64_t)b * strips_per_seq) * RW, row_b = (int64_t)b * T;
    }
    const int t1 = min(t0 + RW, T);
    // validity of rows t0 - HW .. t0 + RW + HW - 1 as one bit each (bit i = row t0 - HW + i; outside [0, T) = 0)
    const int tm = t0 - HW + lane;
    const unsigned long long live = __ballot(lane < RW + 2 * HW && tm >= 0 && tm < T && mask[row_b + (tm >= 0 && tm < T ? tm : 0)] != 0);
    if (((live >> HW) & ((1ull << (t1 - t0)) - 1ull)) == 0ull) {      // every query row of the strip is padding
        for (int t = t0; t < t1; ++t) {
            float* o = out + (row_b + t) * ldo + lane * 8;
            st4(o, make_float4(0.f, 0.f, 0.f, 0.f));
            st4(o + 4, make_float4(0.f, 0.f, 0.f, 0.f));
        }
        return;
    }
    struct Row {
        float4 a, b;
    };
    auto load_row = [&](const float* base, int t) {
        Row r;
        r.a = r.b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T) {
            const float* p = base + (row_b + t) * ld + lane * 8;
            r.a = ld4(p), r.b = ld4(p + 4);
        }
        return r;
    };
    float rb[W];
#pragma unroll
    for (int j = 0; j < W; ++j) rb[j] = REL ? rel[(lane / GROUP) * W + j] : 0.f;
    Row kr[R], vr[R];
#pragma unroll
    for (int i = 0; i < W; ++i) {          // window of the first query row: rows t0 - HW .. t0 + HW in slots 0 .. W-1
        kr[i] = load_row(k, t0 - HW + i);
        vr[i] = load_row(v, t0 - HW + i);
    }
    for (int base = 0; base < RW; base += R) {
#pragma unroll
        for (int ph = 0; ph < R; ++ph) {
            const int t = t0 + base + ph;
            if (t >= t1) break;
            // the row that enters the window with the next query row goes into the slot the window does not cover
            kr[(ph + W) % R] = load_row(k, t + HW + 1);
            vr[(ph + W) % R] = load_row(v, t + HW + 1);
            float* o = out + (row_b + t) * ldo + lane * 8;
            const int bit0 = base + ph;                       // bit of window row j is bit0 + j (row t - HW + j)
            if (!((live >> (bit0 + HW)) & 1ull)) {            // masked query rows are zeroed after the softmax
                st4(o, make_float4(0.f, 0.f, 0.f, 0.f));
                st4(o + 4, make_float4(0.f, 0.f, 0.f, 0.f));
                continue;
            }
            float4 q0 = ld4(q + (row_b + t) * ld + lane * 8), q1 = ld4(q + (row_b + t) * ld + lane * 8 + 4);
            q0.x *= scale; q0.y *= scale; q0.z *= scale; q0.w *= scale;
            q1.x *= scale; q1.y *= scale; q1.z *= scale; q1.w *= scale;
            float sc[W];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int tj = t + j - HW;
                if (tj < 0 || tj >= T) { sc[j] = -INFINITY; continue; }
                const Row& kk = kr[(ph + j) % R];
                float d = dot4(q0, kk.a) + dot4(q1, kk.b);
                d = vrd::group_sum<GROUP>(d);
                if (REL) d += rb[j];
                sc[j] = d + (((live >> (bit0 + j)) & 1ull) ? 0.f : -1e4f);
                m = fmaxf(m, sc[j]);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < W; ++j) { sc[j] = __expf(sc[j] - m); den += sc[j]; }
            const float inv = 1.0f / den;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const int tj = t + j - HW;
                if (tj < 0 || tj >= T) continue;
                const Row& vv = vr[(ph + j) % R];
                const float pj = sc[j] * inv;
                a0.x += pj * vv.a.x; a0.y += pj * vv.a.y; a0.z += pj * vv.a.z; a0.w += pj * vv.a.w;
                a1.x += pj * vv.b.x; a1.y += pj * vv.b.y; a1.z += pj * vv.b.z; a1.w += pj * vv.b.w;
            }
            if (pair) {
                vrd::store_pair4(out + (row_b + t) * ldo, lane * 8, 512, a0, pair, &rt);
                vrd::store_pair4(out + (row_b + t) * ldo, lane * 8 + 4, 512, a1, pair, &rt);
            } else {
                st4(o, a0);
                st4(o + 4, a1);
            }
        }
    }
    rt.report(rflag, vrd::RANGE_ATTN_OUT);
}

// ------------------------------------------------------------------------------------------
// generic masked attention on the VALU.  One workgroup per (b, head); each wave walks queries.
// ------------------------------------------------------------------------------------------
constexpr int SM_MAX_TK = 2048, SM_MAX_HD = 128;

__global__ __launch_bounds__(256) void attn_small_kernel(const float* __restrict__ q, int64_t ldq,
                                                         const float* __restrict__ k, const float* __restrict__ v,
                                                         int64_t ldkv, const uint8_t* __restrict__ kv_mask, int Tq, int Tk,
                                                         int hd, float scale, float* __restrict__ out, int64_t ldo) {
    __shared__ float qs[4][SM_MAX_HD];
    __shared__ float ps[4][SM_MAX_TK];
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* kb = k + (int64_t)b * Tk * ldkv + h * hd;
    const float* vb = v + (int64_t)b * Tk * ldkv + h * hd;
    for (int tq = blockIdx.x * 4 + wave; tq < Tq; tq += gridDim.x * 4) {
        const float* qr = q + ((int64_t)b * Tq + tq) * ldq + h * hd;
        for (int d = lane; d < hd; d += 64) qs[wave][d] = qr[d] * scale;
        __builtin_amdgcn_wave_barrier();
        float m = -INFINITY;
        for (int j = lane; j < Tk; j += 64) {
            float s = -INFINITY;
            if (!kv_mask || kv_mask[(int64_t)b * Tk + j]) {
                const float* kr = kb + (int64_t)j * ldkv;
                s = 0.f;
                for (int d = 0; d < hd; d += 4) {
                    const float4 kk = ld4(kr + d);
                    s += qs[wave][d] * kk.x + qs[wave][d + 1] * kk.y + qs[wave][d + 2] * kk.z + qs[wave][d + 3] * kk.w;
                }
            }
            ps[wave][j] = s;
            m = fmaxf(m, s);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float den = 0.f;
        for (int j = lane; j < Tk; j += 64) {
            const float e = __expf(ps[wave][j] - m);
            ps[wave][j] = e;
            den += e;
        }
        den = vrd::wave_sum(den);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / den;
        for (int d = lane; d < hd; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < Tk; ++j) acc += ps[wave][j] * vb[(int64_t)j * ldkv + d];
            out[((int64_t)b * Tq + tq) * ldo + h * hd + d] = acc * inv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// The same for a handful of queries over a short key row (the predictor's decoder under vidor*.yaml: 10 queries, 8 heads of 32,
// 64 keys; round 4): ONE workgroup per (b, head) stages K and V of the head in LDS once and its four waves take the queries in
// turn -- the kernel above re-read K and V from global memory for every query (ceil(Tq / 4) workgroups per head, a dependent chain
// of Tk strided loads per output element): 1.4 ms per call at 4096 pairs against ~0.2 ms.  Tk * (2 hd + 1) floats of LDS.
constexpr int SL_MAX_FLOATS = 12 * 1024;       // K (pitch hd + 1) | V | 4 query rows | 4 score rows within 48 KiB

__global__ __launch_bounds__(256) void attn_small_lds_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                             const float* __restrict__ v, int64_t ldkv,
                                                             const uint8_t* __restrict__ kv_mask, int Tq, int Tk, int hd, float scale,
                                                             float* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int kp = hd + 1;                                   // K row pitch: lanes read different rows at the same d
    float* const ks = sl;
    float* const vs = ks + ((Tk * kp + 3) & ~3);             // V rows are written as float4: start on a 16-byte boundary for any Tk
    float* const qs = vs + Tk * hd;                          // [4][hd]
    float* const ps = qs + 4 * SM_MAX_HD;                    // [4][Tk]
    const int b = blockIdx.y, h = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* kb = k + (int64_t)b * Tk * ldkv + h * hd;
    const float* vb = v + (int64_t)b * Tk * ldkv + h * hd;
    const int hq = hd >> 2;
    for (int i = threadIdx.x; i < Tk * hq; i += 256) {
        const int j = i / hq, d = (i - j * hq) * 4;
        const bool ok = !kv_mask || kv_mask[(int64_t)b * Tk + j];
        const float4 kk = ok ? ld4(kb + (int64_t)j * ldkv + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vv = ok ? ld4(vb + (int64_t)j * ldkv + d) : make_float4(0.f, 0.f, 0.f, 0.f);
        ks[j * kp + d] = kk.x, ks[j * kp + d + 1] = kk.y, ks[j * kp + d + 2] = kk.z, ks[j * kp + d + 3] = kk.w;
        st4(vs + j * hd + d, vv);
    }
    __syncthreads();
    for (int tq = wave; tq < Tq; tq += 4) {
        const float* qr = q + ((int64_t)b * Tq + tq) * ldq + h * hd;
        for (int d = lane; d < hd; d += 64) qs[wave * SM_MAX_HD + d] = qr[d] * scale;
        __builtin_amdgcn_wave_barrier();
        float m = -INFINITY;
        for (int j = lane; j < Tk; j += 64) {
            float sc = -INFINITY;
            if (!kv_mask || kv_mask[(int64_t)b * Tk + j]) {
                sc = 0.f;
                for (int d = 0; d < hd; ++d) sc += qs[wave * SM_MAX_HD + d] * ks[j * kp + d];      // same order of terms as the kernel above
            }
            ps[wave * Tk + j] = sc;
            m = fmaxf(m, sc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        float den = 0.f;
        for (int j = lane; j < Tk; j += 64) {
            const float e = __expf(ps[wave * Tk + j] - m);
            ps[wave * Tk + j] = e;
            den += e;
        }
        den = vrd::wave_sum(den);
        __builtin_amdgcn_wave_barrier();
        const float inv = 1.0f / den;
        for (int d = lane; d < hd; d += 64) {
            float acc = 0.f;
            for (int j = 0; j < Tk; ++j) acc += ps[wave * Tk + j] * vs[j * hd + d];
            out[((int64_t)b * Tq + tq) * ldo + h * hd + d] = acc * inv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// flash attention on v_mfma_f32_32x32x2_f32.
//
// One workgroup = NW waves = NW*32 query rows of one (b, head).  KV tiles of 32 keys are
// staged once per workgroup in LDS and shared by its waves.  Per wave and KV tile:
//   S^T = K . Q^T   (32 keys x 32 queries, HD/2 MFMAs): Q^T lives in registers for the whole
//                   kernel (pre-scaled by hd^-0.5); K fragments come from LDS.
//   the result puts each QUERY on a lane (column) and 16 of its 32 keys in that lane's
//   registers, so the online-softmax row statistics are register-local plus one exchange with
//   lane^32 -- no 32-lane reductions.
//   O^T += V^T . P^T (HD x 32 queries, 16*HD/32 MFMAs): the probability registers are used
//                   directly as the B operand (register e of lane-half h is key row
//                   (e&3) + 8*(e>>2) + 4h, so the A operand reads that V row), no LDS round trip.
// The k order inside every contraction is a fixed permutation of the natural order; f32
// sums are order-dependent only in the last bits.
// ------------------------------------------------------------------------------------------
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_flash_kernel(const float* __restrict__ q, int64_t ldq,
                                                             const float* __restrict__ k, const float* __restrict__ v,
                                                             int64_t ldkv, const uint8_t* __restrict__ kv_mask, int Tq,
                                                             int Tk, float scale, float* __restrict__ out, int64_t ldo,
                                                             int pair, unsigned* rflag) {
    constexpr int NT = NW * 64;
    constexpr int KP = HD + 1;                    // K tile pitch: conflict-free b32 fragment reads
    constexpr int PIECES = 32 * HD / 4;           // float4 pieces per 32-key tile
    constexpr int PER_T = (PIECES + NT - 1) / NT;
    constexpr int DT = HD / 32;                   // 32-wide d tiles of the output
    __shared__ float ks[32 * KP];
    __shared__ float vs[32 * HD];
    __shared__ float kbias[32];

    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const float* kb = k + (int64_t)b * Tk * ldkv + h * HD;
    const float* vb = v + (int64_t)b * Tk * ldkv + h * HD;

    // Q^T fragment: lane (query li, half lh) holds d = lh*HD/2 + s for step s
    float qf[HD / 2];
    {
        const int tq = q0 + li;
        if (tq < Tq) {
            const float* qr = q + ((int64_t)b * Tq + tq) * ldq + h * HD + lh * (HD / 2);
#pragma unroll
            for (int s4 = 0; s4 < HD / 8; ++s4) {
                const float4 t = ld4(qr + 4 * s4);
                qf[4 * s4] = t.x * scale; qf[4 * s4 + 1] = t.y * scale;
                qf[4 * s4 + 2] = t.z * scale; qf[4 * s4 + 3] = t.w * scale;
            }
        } else {
#pragma unroll
            for (int s = 0; s < HD / 2; ++s) qf[s] = 0.f;
        }
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    float4 kreg[PER_T], vreg[PER_T];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / (HD / 4), c = (idx - r * (HD / 4)) * 4;
            const int key = kt * 32 + r;
            if (idx < PIECES && key < Tk) {
                kreg[i] = ld4(kb + (int64_t)key * ldkv + c);
                vreg[i] = ld4(vb + (int64_t)key * ldkv + c);
            } else {
                kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                vreg[i] = kreg[i];
            }
        }
    };
    auto stage = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * NT;
            if (idx < PIECES) {
                const int r = idx / (HD / 4), c = (idx - r * (HD / 4)) * 4;
                float* kd = ks + r * KP + c;
                kd[0] = kreg[i].x; kd[1] = kreg[i].y; kd[2] = kreg[i].z; kd[3] = kreg[i].w;
                st4(vs + r * HD + c, vreg[i]);
            }
        }
        if (tid < 32) {
            const int key = kt * 32 + tid;
            const bool ok = key < Tk && (!kv_mask || kv_mask[(int64_t)b * Tk + key]);
            kbias[tid] = ok ? 0.f : -INFINITY;
        }
    };

    const int nkt = (Tk + 31) / 32;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1);

        // S^T = K . Q^T
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        const float* kfrag = ks + li * KP + lh * (HD / 2);
#pragma unroll
        for (int s = 0; s < HD / 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kfrag[s], qf[s], sacc, 0, 0, 0);

        // online softmax for query column li; this lane holds keys (e&3) + 8*(e>>2) + 4*lh
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] += kbias[(e & 3) + 8 * (e >> 2) + 4 * lh];
            mx = fmaxf(mx, sacc[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);      // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = __expf(sacc[e] - m_use);
            psum += sacc[e];
        }
        l_part = l_part * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;

        // O^T += V^T . P^T
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float* vrow = vs + ((e & 3) + 8 * (e >> 2) + 4 * lh) * HD + li;
#pragma unroll
            for (int d = 0; d < DT; ++d)
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * d], sacc[e], oacc[d], 0, 0, 0);
        }

        __syncthreads();
        if (kt + 1 < nkt) stage(kt + 1);
        __syncthreads();
    }

    const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
    const float inv = 1.0f / l_tot;
    const int tq = q0 + li;
    if (tq < Tq) {
        vrd::RangeTrack rt;
        float* orow = out + ((int64_t)b * Tq + tq) * ldo;
        const int width = gridDim.y * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // registers 4g..4g+3 are d = 32*d + 8g + 4*lh + 0..3
                const int c = h * HD + 32 * d + 8 * g + 4 * lh;
                const float4 v = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv,
                                             oacc[d][4 * g + 3] * inv);
                if (pair) vrd::store_pair4(orow, c, width, v, pair, &rt);
                else st4(orow + c, v);
            }
        rt.report(rflag, vrd::RANGE_ATTN_OUT);
    }
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

}  // namespace

extern "C" {

static int local_attn_launch(const float* q, const float* k, const float* v, int64_t ld, const uint8_t* mask, const float* rel_pe,
                             int B, int T, const vrd_row_segs* segs, int C, int n_head, int half_win, float* out, int64_t ldo,
                             int out_pair, void* stream);

int vrd_local_attn(const float* q, const float* k, const float* v, int64_t ld, const uint8_t* mask, const float* rel_pe,
                   int B, int T, int C, int n_head, int half_win, float* out, int64_t ldo, int out_pair, void* stream) {
    VRD_CHECK_ARG(B > 0 && T > 0, "vrd_local_attn: bad B / T");
    return local_attn_launch(q, k, v, ld, mask, rel_pe, B, T, nullptr, C, n_head, half_win, out, ldo, out_pair, stream);
}

int vrd_local_attn_segs(const float* q, const float* k, const float* v, int64_t ld, const uint8_t* mask, const float* rel_pe,
                        const vrd_row_segs* segs, int C, int n_head, int half_win, float* out, int64_t ldo, int out_pair,
                        void* stream) {
    VRD_CHECK_ARG(segs, "vrd_local_attn_segs: null row groups");
    return local_attn_launch(q, k, v, ld, mask, rel_pe, 0, 0, segs, C, n_head, half_win, out, ldo, out_pair, stream);
}

static int local_attn_launch(const float* q, const float* k, const float* v, int64_t ld, const uint8_t* mask, const float* rel_pe,
                             int B, int T, const vrd_row_segs* segs, int C, int n_head, int half_win, float* out, int64_t ldo,
                             int out_pair, void* stream) {
    VRD_CHECK_ARG(q && k && v && mask && out, "vrd_local_attn: null pointer");
    VRD_CHECK_ARG(C == 512, "vrd_local_attn: built for C = 512 (got %d)", C);
    VRD_CHECK_ARG(n_head == 4 || n_head == 8, "vrd_local_attn: n_head must be 4 or 8 (got %d)", n_head);
    VRD_CHECK_ARG(half_win == 3 || half_win == 4, "vrd_local_attn: window must be 7 or 9 (half %d)", half_win);
    VRD_CHECK_ARG(ld >= C && ldo >= C && ld % 4 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) &&
                      aligned16(v) && aligned16(out), "vrd_local_attn: bad layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::SegTable sg;
    sg.count = 0;
    int64_t rows = (int64_t)B * T, seg_strips = 0;
    if (segs) {
        seg_strips = vrd::seg_table(sg, segs, 1, 16);
        VRD_CHECK_ARG(seg_strips >= 0, "vrd_local_attn_segs: bad row groups (1..%d groups)", VRD_MAX_SEGS);
        rows = 0;
        for (int g = 0; g < sg.count; ++g) rows += (int64_t)sg.n[g] * sg.T[g];
    }
    const int W = 2 * half_win + 1;
    vrd::ProfScope prof(VRD_K_LOCAL_ATTN, s, 4.0 * (double)rows * W * C, 16.0 * (double)rows * C);
    const float scale = 1.0f / sqrtf((float)(C / n_head));
    unsigned* const rflag = out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr;
    // strips of 16 query rows per wave (default; 32 measured the same) or, VRD_LOCAL_STRIP=0, one wave per query row
    // (6.1 vs 5.6 ms per step at the benchmark shape)
    static const int strip_env = [] { const char* e = getenv("VRD_LOCAL_STRIP"); return e ? atoi(e) : 1; }();
    if (strip_env || segs) {
        constexpr int RW = 16;
        const int strips = (T + RW - 1) / RW;
        dim3 grid((unsigned)(((segs ? seg_strips : (int64_t)B * strips) + 3) / 4)), block(256);
#define VRD_LS(Wn, G)                                                                                                     \
    do {                                                                                                                  \
        if (rel_pe)                                                                                                       \
            hipLaunchKernelGGL((local_attn_strip_kernel<Wn, G, RW, true>), grid, block, 0, s, q, k, v, ld, mask, rel_pe,   \
                               B, T, strips, scale, out, ldo, out_pair, rflag, sg);                                              \
        else                                                                                                              \
            hipLaunchKernelGGL((local_attn_strip_kernel<Wn, G, RW, false>), grid, block, 0, s, q, k, v, ld, mask, rel_pe,  \
                               B, T, strips, scale, out, ldo, out_pair, rflag, sg);                                              \
    } while (0)
        if (half_win == 3 && n_head == 4) VRD_LS(7, 16);
        else if (half_win == 3) VRD_LS(7, 8);
        else if (n_head == 4) VRD_LS(9, 16);
        else VRD_LS(9, 8);
#undef VRD_LS
        VRD_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define VRD_LA(Wn, G)                                                                                                     \
    do {                                                                                                                  \
        if (rel_pe)                                                                                                       \
            hipLaunchKernelGGL((local_attn_kernel<Wn, G, true>), grid, block, 0, s, q, k, v, ld, mask, rel_pe, B, T,      \
                               scale, out, ldo, out_pair, rflag);                                                                \
        else                                                                                                              \
            hipLaunchKernelGGL((local_attn_kernel<Wn, G, false>), grid, block, 0, s, q, k, v, ld, mask, rel_pe, B, T,     \
                               scale, out, ldo, out_pair, rflag);                                                                \
    } while (0)
    if (half_win == 3 && n_head == 4) VRD_LA(7, 16);
    else if (half_win == 3) VRD_LA(7, 8);
    else if (n_head == 4) VRD_LA(9, 16);
    else VRD_LA(9, 8);
#undef VRD_LA
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_attention(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask,
                  int B, int Tq, int Tk, int n_head, int head_dim, float* out, int64_t ldo, int algo, int out_pair,
                  void* stream) {
    VRD_CHECK_ARG(q && k && v && out, "vrd_attention: null pointer");
    VRD_CHECK_ARG(B > 0 && B <= 65535 && Tq > 0 && Tk > 0 && n_head > 0 && n_head <= 65535, "vrd_attention: bad sizes");
    VRD_CHECK_ARG(head_dim % 4 == 0 && head_dim <= SM_MAX_HD, "vrd_attention: head_dim must be a multiple of 4, <= %d", SM_MAX_HD);
    VRD_CHECK_ARG(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out),
                  "vrd_attention: rows must be 16-byte aligned");
    VRD_CHECK_ARG(ldq >= n_head * head_dim && ldkv >= n_head * head_dim && ldo >= n_head * head_dim, "vrd_attention: leading dimension too small");
    VRD_CHECK_ARG(algo >= 0 && algo <= 2, "vrd_attention: bad algo %d", algo);
    const bool flash_ok = head_dim == 64 || head_dim == 128;
    // the MFMA kernel also wins for a handful of queries (the predictor's 9: a 32-query tile is 72 % padding, but
    // the VALU kernel re-reads K and V once per query: 0.17 vs 0.38 ms at B = 2048, Tk = 36)
    if (algo == 0) algo = flash_ok ? 2 : 1;
    VRD_CHECK_ARG(algo != 2 || flash_ok, "vrd_attention: flash kernel needs head_dim 64 or 128");
    VRD_CHECK_ARG(!out_pair || algo == 2, "vrd_attention: pair output is only built for the flash kernel");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    const double flops = 4.0 * B * (double)n_head * Tq * Tk * head_dim;
    const double bytes = 4.0 * B * (double)n_head * head_dim * (2.0 * Tq + 2.0 * Tk);
    if (algo == 1) {
        VRD_CHECK_ARG(Tk <= SM_MAX_TK, "vrd_attention: generic kernel supports Tk <= %d", SM_MAX_TK);
        vrd::ProfScope prof(VRD_K_ATTN_SMALL, s, flops, bytes);
        const int lds_floats = ((Tk * (head_dim + 1) + 3) & ~3) + Tk * head_dim + 4 * SM_MAX_HD + 4 * Tk;      // (K region rounded up to 16 bytes)
        if (Tq <= 16 && lds_floats <= SL_MAX_FLOATS) {
            // a few queries over a short key row: K / V of a head staged in LDS once per (b, head)
            hipLaunchKernelGGL(attn_small_lds_kernel, dim3(n_head, B), dim3(256), (size_t)lds_floats * sizeof(float), s, q, ldq, k, v, ldkv,
                               kv_mask, Tq, Tk, head_dim, scale, out, ldo);
        } else {
            int gx = (Tq + 3) / 4;
            if (gx > 64) gx = 64;
            hipLaunchKernelGGL(attn_small_kernel, dim3(gx, n_head, B), dim3(256), 0, s, q, ldq, k, v, ldkv, kv_mask, Tq, Tk,
                               head_dim, scale, out, ldo);
        }
    } else {
        vrd::ProfScope prof(VRD_K_ATTN_FLASH, s, flops, bytes);
        const int tiles = (Tq + 31) / 32;
        // 3 or 4 query tiles per workgroup, whichever wastes fewer wave slots
        const int waste3 = ((tiles + 2) / 3) * 3 - tiles, waste4 = ((tiles + 3) / 4) * 4 - tiles;
        const int nw = (waste3 < waste4) ? 3 : 4;
        dim3 grid((tiles + nw - 1) / nw, n_head, B);
#define VRD_FA(HD, NW) hipLaunchKernelGGL((attn_flash_kernel<HD, NW>), grid, dim3(NW * 64), 0, s, q, ldq, k, v, ldkv, kv_mask, Tq, Tk, scale, out, ldo, out_pair, out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr)
        if (head_dim == 128 && nw == 3) VRD_FA(128, 3);
        else if (head_dim == 128) VRD_FA(128, 4);
        else if (nw == 3) VRD_FA(64, 3);
        else VRD_FA(64, 4);
#undef VRD_FA
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
