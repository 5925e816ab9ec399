// HBM-bound row kernels of the relation-encoding path: layout change at the boundary,
// channel LayerNorm, fused depthwise-conv * mask -> LayerNorm, max-pool skip, mask head.
//
// All of them keep one activation row (C = 256 or 512 contiguous floats) per wavefront:
// lane l owns channels [256*i + 4*l, 256*i + 4*l + 4) for i < C/256, so every global
// access is a full-wave 1 KiB float4 transaction and the per-row statistics are one
// 64-lane butterfly.
#include "vrd_common.h"
#include <cmath>

namespace {

constexpr float LN_EPS = 1e-5f;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ------------------------------------------------------------------------------------------
// (B, C_total, T) slab -> channels-last rows, 64 x 64 tiles through LDS
// ------------------------------------------------------------------------------------------
// Source sequence of output sequence b: src_batch[b] (a gather over the batch) or b; a source channel row holds T_src >= T
// frames, of which the first T are converted (the tail is padding the caller leaves out: MaskVRD's tight padding).
__global__ __launch_bounds__(256) void bct_to_btc_kernel(const float* __restrict__ src, int C_total, int T, int c0,
                                                         int count, float* __restrict__ dst, int64_t ld_dst, int pair, int T_src,
                                                         const int32_t* __restrict__ src_batch, unsigned* rflag) {
    __shared__ float tile[64][65];
    vrd::RangeTrack rt;
    const int b = blockIdx.z, ct = blockIdx.y * 64, tt = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float* s = src + ((int64_t)(src_batch ? src_batch[b] : b) * C_total + c0) * T_src;
    for (int c = ty; c < 64; c += 4)
        tile[c][tx] = (ct + c < count && tt + tx < T) ? s[(int64_t)(ct + c) * T_src + tt + tx] : 0.f;
    __syncthreads();
    for (int t = ty; t < 64; t += 4)
        if (tt + t < T && ct + tx < count) {
            float* row = dst + ((int64_t)b * T + tt + t) * ld_dst;
            if (pair) vrd::store_pair1(row, ct + tx, count, tile[tx][t], pair, &rt);
            else row[ct + tx] = tile[tx][t];
        }
    rt.report(rflag, vrd::RANGE_INPUT);
}

// same, 16-byte accesses on both sides (T % 4 == 0, count % 4 == 0, 16-byte aligned rows): float4 reads along t,
// float4 / pair-row writes of four channels
__global__ __launch_bounds__(256) void bct_to_btc_vec_kernel(const float* __restrict__ src, int C_total, int T, int c0,
                                                             int count, float* __restrict__ dst, int64_t ld_dst, int pair, int T_src,
                                                             const int32_t* __restrict__ src_batch, unsigned* rflag) {
    __shared__ float tile[64][65];
    vrd::RangeTrack rt;
    const int b = blockIdx.z, ct = blockIdx.y * 64, tt = blockIdx.x * 64;
    const float* s = src + ((int64_t)(src_batch ? src_batch[b] : b) * C_total + c0) * T_src;
    {
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // 16 float4 along t, 16 channels per sweep
#pragma unroll
        for (int c = ty; c < 64; c += 16) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ct + c < count && tt + 4 * tx < T) v = ld4(s + (int64_t)(ct + c) * T_src + tt + 4 * tx);
            tile[c][4 * tx] = v.x, tile[c][4 * tx + 1] = v.y, tile[c][4 * tx + 2] = v.z, tile[c][4 * tx + 3] = v.w;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int id = threadIdx.x + 256 * k;
        const int cg = id & 15, t = id >> 4;
        if (tt + t >= T || ct + 4 * cg >= count) continue;
        const float4 v = make_float4(tile[4 * cg][t], tile[4 * cg + 1][t], tile[4 * cg + 2][t], tile[4 * cg + 3][t]);
        float* row = dst + ((int64_t)b * T + tt + t) * ld_dst;
        if (pair) vrd::store_pair4(row, ct + 4 * cg, count, v, pair, &rt);
        else st4(row + ct + 4 * cg, v);
    }
    rt.report(rflag, vrd::RANGE_INPUT);
}

__global__ __launch_bounds__(256) void btc_to_bct_kernel(const float* __restrict__ src, int64_t ld_src, int C, int T,
                                                         float* __restrict__ dst) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, ct = blockIdx.y * 64, tt = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int t = ty; t < 64; t += 4)
        tile[t][tx] = (tt + t < T && ct + tx < C) ? src[((int64_t)b * T + tt + t) * ld_src + ct + tx] : 0.f;
    __syncthreads();
    for (int c = ty; c < 64; c += 4)
        if (ct + c < C && tt + tx < T) dst[((int64_t)b * C + ct + c) * T + tt + tx] = tile[tx][c];
}

// ------------------------------------------------------------------------------------------
// eval batching straight from the dataloader's per-pair matrices: pair p is an (L_p, C_in) frame-major
// matrix [s_vis | o_vis | (s_clip | o_clip) | so_box | s_box | o_box]; one wave copies one (pair, frame) row
// into the channels-last operand buffers of the backbone (zero rows past L_p), in pair-row format for the
// wide visual / clip slabs when requested.  Replaces the zero-padded (B, C_in, T) batch of
// models/maskvrd.py:382-385 and the channel slicing of models/backbones.py:161-166.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_segment(const float* src, bool live, int lane, int width, float* dst, int pair,
                                             vrd::RangeTrack* rt = nullptr) {
    for (int c = lane; c < width; c += 64) {
        const float v = live ? src[c] : 0.f;
        if (pair) vrd::store_pair1(dst, c, width, v, pair, rt);
        else dst[c] = v;
    }
}

__global__ __launch_bounds__(256) void pack_pairs_kernel(vrd_pack_args a, unsigned* rflag) {
    vrd::RangeTrack rt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: addresses on the scalar unit
    if (row >= (int64_t)a.P * a.T) return;
    const int p = (int)(row / a.T), t = (int)(row - (int64_t)p * a.T);
    const bool live = t < a.lens[p];
    const float* src = a.src[p] + (int64_t)t * a.C_in;
    const int64_t half = (int64_t)a.P * a.T;             // rows per (subject | object) half of the stacked buffers
    int c0 = 0;
    pack_segment(src + c0, live, lane, a.V, a.vis + row * a.V, a.pair_wide, &rt);              c0 += a.V;
    pack_segment(src + c0, live, lane, a.V, a.vis + (half + row) * a.V, a.pair_wide, &rt);     c0 += a.V;
    if (a.Cc) {
        pack_segment(src + c0, live, lane, a.Cc, a.clip + row * a.Cc, a.pair_wide, &rt);          c0 += a.Cc;
        pack_segment(src + c0, live, lane, a.Cc, a.clip + (half + row) * a.Cc, a.pair_wide, &rt); c0 += a.Cc;
    }
    pack_segment(src + c0, live, lane, a.S, a.so_box + row * a.S, 0);                     c0 += a.S;
    pack_segment(src + c0, live, lane, a.E, a.ent + row * a.E, 0);                        c0 += a.E;
    pack_segment(src + c0, live, lane, a.E, a.ent + (half + row) * a.E, 0);
    rt.report(rflag, vrd::RANGE_INPUT);
}

// ------------------------------------------------------------------------------------------
// eval batching straight from PER-TRACKLET features: the reference's dataloader builds one (L, C_in) matrix
// per ordered pair by slicing and concatenating tracklet features on the host (dataloaders/vidvrd.py:652-693: every
// tracklet is copied into 2 (N - 1) pairs) and computes the pairs' box features there (utils/misc.py:158-217).  Here
// the tracklets' rows live on the device once -- vis (sum L, V), clip, boxes (sum L, 4, already clamped) -- and one
// wave per (pair, frame) gathers the subject and object rows and computes the 5 + 8 + 8 box features, writing the
// backbone's channels-last operand buffers exactly like pack_pairs_kernel.  Frame t of pair p is row
// s_row[p] + t * stride (o_row[p] + t * stride) of the concatenated arrays.
// Box arithmetic is written operation by operation with the round-to-nearest intrinsics (no FMA contraction), in the
// reference's order, so everything but the three logarithms is the reference's f32 value bit for bit.
// ------------------------------------------------------------------------------------------
struct Box4 {
    float x0, y0, x1, y1;
};
__device__ __forceinline__ Box4 load_box(const float* boxes, int64_t row) {
    const float4 b = *reinterpret_cast<const float4*>(boxes + row * 4);
    return Box4{b.x, b.y, b.z, b.w};
}
// normalised (cx, cy, w, h) of utils/misc.py:184-192
__device__ __forceinline__ void entity_geom(const Box4& b, float w, float h, float (&g)[4]) {
    const float x0 = __fdiv_rn(b.x0, w), x1 = __fdiv_rn(b.x1, w), y0 = __fdiv_rn(b.y0, h), y1 = __fdiv_rn(b.y1, h);
    g[0] = __fdiv_rn(__fadd_rn(x1, x0), 2.0f);
    g[1] = __fdiv_rn(__fadd_rn(y1, y0), 2.0f);
    g[2] = __fsub_rn(x1, x0);
    g[3] = __fsub_rn(y1, y0);
}
// [cx, dcx, cy, dcy, w, dw, h, dh] of frame t of an n-frame strided box sequence starting at row0 (utils/misc.py:194-217)
__device__ __forceinline__ void entity_feats(const float* boxes, int64_t row0, int stride, int t, int n, float w, float h,
                                             float (&f)[8]) {
    float g[4], a[4], b[4];
    entity_geom(load_box(boxes, row0 + (int64_t)t * stride), w, h, g);
    float d[4];
    if (t > 0) {
        entity_geom(load_box(boxes, row0 + (int64_t)(t - 1) * stride), w, h, a);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __fsub_rn(g[i], a[i]);
    } else if (n < 2) {     // a one-frame pair has no difference (the dataloader never emits one; never read past the pair)
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = 0.f;
    } else {            // first frame: d0 - (d1 - d0) with d0 = v1 - v0, d1 = v2 - v1; just d0 when there are two frames
        entity_geom(load_box(boxes, row0 + stride), w, h, a);
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __fsub_rn(a[i], g[i]);
        if (n > 2) {
            entity_geom(load_box(boxes, row0 + 2 * (int64_t)stride), w, h, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = __fsub_rn(d[i], __fsub_rn(__fsub_rn(b[i], a[i]), d[i]));
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) f[2 * i] = g[i], f[2 * i + 1] = d[i];
}

__global__ __launch_bounds__(256) void gather_pairs_kernel(vrd_gather_args a, unsigned* rflag) {
    vrd::RangeTrack rt;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= (int64_t)a.P * a.T) return;
    const int p = (int)(row / a.T), t = (int)(row - (int64_t)p * a.T);
    const int n = a.lens[p];
    const bool live = t < n;
    const int64_t rs = a.s_row[p] + (int64_t)t * a.stride, ro = a.o_row[p] + (int64_t)t * a.stride;
    const int64_t half = (int64_t)a.P * a.T;
    if (a.out_vis) {        // NULL: box features only (the wide rows come from the per-tracklet streams, vrd_assemble_pairs)
        pack_segment(a.vis + (live ? rs : 0) * a.V, live, lane, a.V, a.out_vis + row * a.V, a.pair_wide, &rt);
        pack_segment(a.vis + (live ? ro : 0) * a.V, live, lane, a.V, a.out_vis + (half + row) * a.V, a.pair_wide, &rt);
        if (a.Cc) {
            pack_segment(a.clip + (live ? rs : 0) * a.Cc, live, lane, a.Cc, a.out_clip + row * a.Cc, a.pair_wide, &rt);
            pack_segment(a.clip + (live ? ro : 0) * a.Cc, live, lane, a.Cc, a.out_clip + (half + row) * a.Cc, a.pair_wide, &rt);
        }
        rt.report(rflag, vrd::RANGE_INPUT);
    }
    // box features: lanes 0 (subject-object), 1 (subject), 2 (object) compute, everybody stores zeros for padded frames
    float* const so = a.out_so_box + row * 5;
    float* const es = a.out_ent + row * 8;
    float* const eo = a.out_ent + (half + row) * 8;
    if (!live) {
        if (lane < 5) so[lane] = 0.f;
        if (lane < 8) es[lane] = 0.f, eo[lane] = 0.f;
        return;
    }
    if (lane == 0) {        // utils/misc.py:158-178
        const Box4 s = load_box(a.boxes, rs), o = load_box(a.boxes, ro);
        const float s_cx = __fdiv_rn(__fadd_rn(s.x1, s.x0), 2.0f), s_cy = __fdiv_rn(__fadd_rn(s.y1, s.y0), 2.0f);
        const float o_cx = __fdiv_rn(__fadd_rn(o.x1, o.x0), 2.0f), o_cy = __fdiv_rn(__fadd_rn(o.y1, o.y0), 2.0f);
        const float s_w = __fsub_rn(s.x1, s.x0), s_h = __fsub_rn(s.y1, s.y0), o_w = __fsub_rn(o.x1, o.x0), o_h = __fsub_rn(o.y1, o.y0);
        so[0] = __fdiv_rn(__fsub_rn(s_cx, o_cx), o_cx);
        so[1] = __fdiv_rn(__fsub_rn(s_cy, o_cy), o_cy);
        so[2] = logf(__fdiv_rn(s_w, o_w));
        so[3] = logf(__fdiv_rn(s_h, o_h));
        so[4] = logf(__fdiv_rn(__fmul_rn(s_w, s_h), __fmul_rn(o_w, o_h)));
    } else if (lane == 1 || lane == 2) {
        float f[8];
        entity_feats(a.boxes, lane == 1 ? a.s_row[p] : a.o_row[p], a.stride, t, n, a.w, a.h, f);
        float* const dst = lane == 1 ? es : eo;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = f[i];
    }
}

// ------------------------------------------------------------------------------------------
// entity-stage rows of a pair batch from per-tracklet rows + window-edge snippets (see vrd_assemble_args); one wave per row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_pairs_kernel(vrd_assemble_args a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= 2 * (int64_t)a.P * a.T) return;
    const int e = (int)(row / a.T), t = (int)(row - (int64_t)e * a.T);
    const int side = e >= a.P, p = e - side * a.P;
    const int n = a.lens[p];
    const float* src = nullptr;
    if (t < n) {
        const int64_t piece = ((int64_t)side * 2 * a.P + p) * a.L;            // this entity's start piece; its end piece is P pieces on
        const int end_len = n == a.T ? a.L : a.piece;                          // see vrd_assemble_args
        if (n <= a.piece || t < a.reach) src = a.snippets + (piece + t) * a.D;
        else if (t >= n - a.reach) src = a.snippets + (piece + (int64_t)a.P * a.L + (t - (n - end_len))) * a.D;
        else src = a.streams + (a.stream_row[e] + t) * a.D;
    }
    float4* dst = reinterpret_cast<float4*>(a.out + row * a.D);
    for (int c = lane; c < a.D / 4; c += 64) dst[c] = src ? reinterpret_cast<const float4*>(src)[c] : float4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------
// LayerNorm over channels; NV = C / 256
// ------------------------------------------------------------------------------------------
// lane's channels: WIDE (NV == 2): 8 consecutive channels lane*8 .. lane*8+7 (v[0], v[1]), so that pair rows are
// written with 16-byte stores; otherwise 4 channels in each 256-channel half
template <int NV, bool WIDE>
__device__ __forceinline__ int lane_chan(int i, int lane) { return WIDE ? lane * 8 + i * 4 : i * 256 + lane * 4; }

template <int NV, bool WIDE = false>
__device__ __forceinline__ void ln_rows(float4 (&v)[NV], const float* gamma, const float* beta, int lane, bool relu) {
    constexpr float inv_c = 1.0f / (256.0f * NV);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = vrd::wave_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    // one division per row (the reference divides every element by sqrt(var + eps); x * (1 / d) differs from x / d by
    // at most an ulp, and an f32 division is ~10 instructions per element in kernels that are instruction-bound)
    const float inv = 1.0f / sqrtf(vrd::wave_sum(ss) * inv_c + LN_EPS);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 g = ld4(gamma + lane_chan<NV, WIDE>(i, lane)), bb = ld4(beta + lane_chan<NV, WIDE>(i, lane));
        v[i].x = fmaf(v[i].x * inv, g.x, bb.x);
        v[i].y = fmaf(v[i].y * inv, g.y, bb.y);
        v[i].z = fmaf(v[i].z * inv, g.z, bb.z);
        v[i].w = fmaf(v[i].w * inv, g.w, bb.w);
        if (relu) {
            v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f);
            v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f);
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                        int64_t ldy, int64_t rows, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int relu,
                                                        const float* __restrict__ post_add, int64_t ld_add, int period,
                                                        int pair, unsigned* rflag) {
    vrd::RangeTrack rt;
    const int lane = threadIdx.x & 63;
    // Waves walk the rows with the stride of the grid (row = wave, wave + waves of the grid, ...) and request their next row before
    // working on the current one.  All resident waves then read and write one compact, moving window of memory, which is what HBM
    // serves fastest when reads and writes mix (profiles/r06_lab_hbm_rw.txt: a wave per row 5.8 TB/s, a strip of >= 12 rows per
    // wave 3.3 TB/s for one read : one write), and a wave's load latency hides behind its own previous row.
    int64_t row = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: addresses on the scalar unit
    const int64_t stride = (int64_t)gridDim.x * 4;
    if (row >= rows) return;
    float4 v[NV], vn[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ld4(x + row * ldx + i * 256 + lane * 4);
    for (;;) {
        const int64_t nxt = row + stride;
        if (nxt < rows) {
#pragma unroll
            for (int i = 0; i < NV; ++i) vn[i] = ld4(x + nxt * ldx + i * 256 + lane * 4);
        }
        ln_rows<NV>(v, gamma, beta, lane, relu != 0);
        if (post_add) {
            const float* a = post_add + (row % period) * ld_add;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = f4add(v[i], ld4(a + i * 256 + lane * 4));
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (pair) vrd::store_pair4(y + row * ldy, i * 256 + lane * 4, 256 * NV, v[i], pair, &rt);
            else st4(y + row * ldy + i * 256 + lane * 4, v[i]);
        }
        if (nxt >= rows) break;
        row = nxt;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vn[i];
    }
    rt.report(rflag, vrd::RANGE_LAYERNORM);
}

// ------------------------------------------------------------------------------------------
// dense conv with few input channels (taps * Cin <= 32) * mask -> [LayerNorm -> ReLU]: the box-feature embeddings
// (include/vrdone_hip.h, vrd_conv_ln).  A wave per row, rows walked with the grid's stride (layernorm_kernel's pattern: the
// kernel writes N floats per row and reads next to nothing); a row's taps * Cin inputs are wave-uniform and come through the
// scalar cache, the weights sit in LDS k-major ([taps * Cin][N]: a lane's channels of one k are one or two float4).
// ------------------------------------------------------------------------------------------
constexpr int CL_MAXK = 32;
template <int NV>
__global__ __launch_bounds__(256) void conv_ln_kernel(vrd_conv_ln_args p, unsigned* rflag) {
    vrd::RangeTrack rt;
    constexpr int N = 256 * NV;
    constexpr bool WIDE = NV == 2;
    extern __shared__ __attribute__((aligned(16))) float cl_lds[];      // [K][N] weights | bias [N]
    const int K = p.Cin * p.taps;
    // W (N, Cin, taps) -> k = tap * Cin + ci major
    for (int idx = threadIdx.x; idx < N * K; idx += 256) {
        const int n = idx / K, rem = idx - n * K, ci = rem / p.taps, tap = rem - ci * p.taps;
        cl_lds[(tap * p.Cin + ci) * N + n] = p.w[idx];
    }
    for (int n = threadIdx.x; n < N; n += 256) cl_lds[K * N + n] = p.bias ? p.bias[n] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // a wave takes RG consecutive rows per trip (groups walked with the grid's stride): a weight fragment read from LDS serves RG
    // rows -- with one row per trip the kernel was bound by its 48 LDS reads per row
    constexpr int RG = 4;
    int64_t grp = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    const int half = p.taps / 2;
    // the inputs of a trip -- rows row0 - half .. row0 + RG - 1 + half, Cin floats each, at most 64 values -- are ONE load: lane l
    // holds value l % Cin of row l / Cin (0 outside the matrix), the contraction reads them back with v_readlane.  (As scalar loads
    // they were 96 dependent round trips per trip.)  The next trip's are requested before this trip's arithmetic.
    const int lrow = lane / p.Cin, lci = lane - lrow * p.Cin;
    auto load_in = [&](int64_t g) {
        const int64_t r = g * RG - half + lrow;
        float val = 0.f;
        if (lrow < RG + 2 * half && r >= 0 && r < p.rows) val = p.x[r * p.ldx + lci];
        return val;
    };
    float xin = grp * RG < p.rows ? load_in(grp) : 0.f;
    for (; grp * RG < p.rows; grp += stride) {
        const int64_t row0 = grp * RG;
        const float xcur = xin;
        if ((grp + stride) * RG < p.rows) xin = load_in(grp + stride);
        float4 v[RG][NV];
        int t[RG];
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            t[j] = (int)((row0 + j) % p.T);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[j][i] = *reinterpret_cast<const float4*>(cl_lds + K * N + lane_chan<NV, WIDE>(i, lane));
        }
        for (int tap = 0; tap < p.taps; ++tap) {
            for (int ci = 0; ci < p.Cin; ++ci) {
                const float* wk = cl_lds + (tap * p.Cin + ci) * N;
                float4 w[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) w[i] = *reinterpret_cast<const float4*>(wk + lane_chan<NV, WIDE>(i, lane));
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    const int tt = t[j] + tap - half;
                    // (Conv1d zero padding at the sequence's ends, rows past the end of the matrix; wave-uniform: a scalar load)
                    const bool ok = tt >= 0 && tt < p.T && row0 + j < p.rows;
                    const float xl = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xcur), (j + tap) * p.Cin + ci));
                    const float xv = ok ? xl : 0.f;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        v[j][i].x = fmaf(w[i].x, xv, v[j][i].x); v[j][i].y = fmaf(w[i].y, xv, v[j][i].y);
                        v[j][i].z = fmaf(w[i].z, xv, v[j][i].z); v[j][i].w = fmaf(w[i].w, xv, v[j][i].w);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RG; ++j) {
            const int64_t row = row0 + j;
            if (row >= p.rows) break;
            if (p.row_mask) {
                const float mk = (float)vrd::uniform_load(p.row_mask + row);
#pragma unroll
                for (int i = 0; i < NV; ++i) v[j][i].x *= mk, v[j][i].y *= mk, v[j][i].z *= mk, v[j][i].w *= mk;
            }
            if (p.gamma) {
                ln_rows<NV, WIDE>(v[j], p.gamma, p.beta, lane, p.relu != 0);
            } else if (p.relu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    v[j][i].x = fmaxf(v[j][i].x, 0.f); v[j][i].y = fmaxf(v[j][i].y, 0.f);
                    v[j][i].z = fmaxf(v[j][i].z, 0.f); v[j][i].w = fmaxf(v[j][i].w, 0.f);
                }
            }
            float* yr = p.y + row * p.ldy;
            if (WIDE && p.out_pair) {
                vrd::store_pair8(yr, lane * 8, v[j][0], v[j][NV - 1], p.out_pair, &rt);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane_chan<NV, WIDE>(i, lane);
                    if (p.out_pair) vrd::store_pair4(yr, c, N, v[j][i], p.out_pair, &rt);
                    else st4(yr + c, v[j][i]);
                }
            }
        }
    }
    rt.report(rflag, vrd::RANGE_LAYERNORM);
}

// ------------------------------------------------------------------------------------------
// depthwise (or 2-in-per-group) conv along t, * mask, -> LayerNorm, up to three weight sets
// sharing the input rows (the q/k/v branches of the conv-attention modules)
// ------------------------------------------------------------------------------------------
// One wave walks a strip of RW consecutive output rows of one sequence, so the input rows it shares with its
// neighbours stay in registers (a 3-row window, the next row(s) already requested), and the per-channel
// parameters of all output sets (taps, bias, gamma, beta: 12 KiB per 512-channel set) sit in LDS, filled once per
// workgroup: with one row per wave they were re-read through the vector L1 for every row, 36 KiB per 8 KiB of
// row data, and the kernel ran at the L1's rate (2.7 TB/s of HBM) instead of HBM's.
// LDS per set o (floats): taps [GIN*KS][C] | bias [C] | gamma [C] | beta [C]; behind the sets the optional input
// LayerNorm's gamma [C] | beta [C].
constexpr int DW_RW = 16;                      // output rows per wave (default; the launcher may pick 8..32, see vrd_dwconv_ln)

template <int NV, int KS, int GIN>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(vrd_dwconv_ln_args p, int Tout_u, int strips_per_seq, int strips_per_wave, int rw, unsigned* rflag,
                                                        vrd::SegTable sg) {
    vrd::RangeTrack rt;
    constexpr int C = 256 * NV, NT = GIN * KS, SETF = (NT + 3) * C;
    constexpr bool WIDE = NV == 2 && GIN == 1;
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // ---- parameters -> LDS (taps transposed to tap-major so a lane's four channels are one float4)
    for (int o = 0; o < p.n_out; ++o) {
        float* const ls = dw_lds + o * SETF;
        if (p.packed[o]) {               // the caller's image of this block: a straight copy
            for (int idx = threadIdx.x * 4; idx < SETF; idx += 1024) st4(ls + idx, ld4(p.packed[o] + idx));
            continue;
        }
        for (int idx = threadIdx.x; idx < C * NT; idx += 256) ls[(idx % NT) * C + idx / NT] = p.w[o][idx];
        for (int c = threadIdx.x; c < C; c += 256) {
            ls[NT * C + c] = p.bias[o] ? p.bias[o][c] : 0.f;
            ls[(NT + 1) * C + c] = p.gamma[o] ? p.gamma[o][c] : 1.f;
            ls[(NT + 2) * C + c] = p.gamma[o] ? p.beta[o][c] : 0.f;
        }
    }
    const float* const pre = dw_lds + p.n_out * SETF;
    if (p.pre_gamma)
        for (int c = threadIdx.x; c < C; c += 256) {
            dw_lds[p.n_out * SETF + c] = p.pre_gamma[c];
            dw_lds[p.n_out * SETF + C + c] = p.pre_beta[c];
        }
    __syncthreads();
    // XCD-aware renumbering: a contiguous range of strips per XCD, so the halo rows two neighbouring strips share are
    // fetched into one L2 only
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    // the parameter blocks are filled once per workgroup; each wave then walks strips_per_wave consecutive strips
    for (int si = 0; si < strips_per_wave; ++si) {
    const int64_t ws = ((int64_t)lid * 4 + wave) * strips_per_wave + si;
    // the strip's sequence: its frame counts and the first row of its input / coarser-level / output rows
    int Tin, Tout, to0;
    int64_t xrow0, urow0, orow0;
    if (sg.count) {                                  // ragged row space: groups of sequences of different lengths
        int g, b, strip;
        if (!vrd::seg_find(sg, ws, g, b, strip)) break;
        Tin = sg.T[g], Tout = Tin / p.stride, to0 = strip * rw;
        xrow0 = sg.row[g] + (int64_t)b * Tin;
        urow0 = sg.row[g] / 2 + (int64_t)b * (Tin / 2);
        orow0 = sg.row[g] / p.stride + (int64_t)b * Tout;
    } else {
        const int b = (int)(ws / strips_per_seq);
        if (b >= p.B) break;
        Tin = p.Tin, Tout = Tout_u, to0 = (int)(ws - (int64_t)b * strips_per_seq) * rw;
        xrow0 = (int64_t)b * Tin, urow0 = (int64_t)b * (Tin / 2), orow0 = (int64_t)b * Tout;
    }
    const int to1 = min(to0 + rw, Tout);

    struct Row {
        float4 v[NV][GIN];
    };
    auto load_row = [&](int ti) {
        Row r;
        const bool ok = ti >= 0 && ti < Tin;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int g = 0; g < GIN; ++g) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) {
                    const int64_t coff = (int64_t)GIN * lane_chan<NV, WIDE>(i, lane) + 4 * g;
                    v = ld4(p.x + (xrow0 + ti) * p.ldx + coff);
                    if (p.x_up) v = f4add(v, ld4(p.x_up + (urow0 + (ti >> 1)) * p.ldx_up + coff));
                }
                r.v[i][g] = v;
            }
        if (GIN == 1 && p.pre_gamma && ok) {       // the block's ln1, applied to a real row once, as it enters the window
            float4 t[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) t[i] = r.v[i][0];
            ln_rows<NV, WIDE>(t, pre, pre + C, lane, false);
#pragma unroll
            for (int i = 0; i < NV; ++i) r.v[i][0] = t[i];
        }
        return r;
    };
    // ---- a strip of padded frames only (every mask_out byte 0): conv * 0 = 0, so each row is LayerNorm(0) = beta
    // (0 without LayerNorm), ReLU applied -- written without reading the input.  Same bits as the path below.
    if (p.mask_out) {
        const int64_t r0 = orow0 + to0;
        const bool mine = lane < to1 - to0 && p.mask_out[r0 + lane] != 0;
        if (!__any(mine)) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                if (o >= p.n_out) break;
                const float* const lb = dw_lds + o * SETF + (NT + 2) * C;          // beta (zeros without LayerNorm)
                float4 val[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    val[i] = *reinterpret_cast<const float4*>(lb + lane_chan<NV, WIDE>(i, lane));
                    if (p.relu[o]) {
                        val[i].x = fmaxf(val[i].x, 0.f); val[i].y = fmaxf(val[i].y, 0.f);
                        val[i].z = fmaxf(val[i].z, 0.f); val[i].w = fmaxf(val[i].w, 0.f);
                    }
                }
                for (int to = to0; to < to1; ++to) {
                    const int64_t row = orow0 + to;
                    if (WIDE && p.out_pair[o]) {
                        vrd::store_pair8(p.y[o] + row * p.ldy[o], lane * 8, val[0], val[NV - 1], p.out_pair[o], &rt);
                    } else {
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            const int c = lane_chan<NV, WIDE>(i, lane);
                            if (p.out_pair[o]) vrd::store_pair4(p.y[o] + row * p.ldy[o], c, 256 * NV, val[i], p.out_pair[o], &rt);
                            else st4(p.y[o] + row * p.ldy[o] + c, val[i]);
                        }
                    }
                }
            }
            continue;
        }
    }
    // window win[k] = input row stride*to + k - KS/2; between consecutive output rows it moves by `stride`
    Row win[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) win[k] = load_row(p.stride * to0 + k - KS / 2);
    for (int to = to0; to < to1; ++to) {
        // request what the next output row adds to the window before working on this one
        Row nxt[2];
        const int tn = p.stride * (to + 1) - KS / 2;            // first input row of the next window
        const bool more = to + 1 < to1;
        if (KS == 1) {
            if (more) nxt[0] = load_row(tn);
        } else if (p.stride == 1) {
            if (more) nxt[0] = load_row(tn + 2);
        } else {
            if (more) {
                nxt[0] = load_row(tn + 1);
                nxt[1] = load_row(tn + 2);
            }
        }
        const int64_t row = orow0 + to;
        const float mk = p.mask_out ? (float)p.mask_out[row] : 1.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            if (o >= p.n_out) break;
            const float* const ls = dw_lds + o * SETF;
            float4 acc[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = lane_chan<NV, WIDE>(i, lane);
                float4 r = *reinterpret_cast<const float4*>(ls + NT * C + c);          // bias
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    if (GIN == 1) {
                        const float4 w = *reinterpret_cast<const float4*>(ls + k * C + c);
                        const float4 v = win[k].v[i][0];
                        r.x += w.x * v.x; r.y += w.y * v.y; r.z += w.z * v.z; r.w += w.w * v.w;
                    } else {
                        // out channel c reads in channels 2c, 2c+1 (taps [g][k] per channel): lane's 8 inputs
                        const float4 w0 = *reinterpret_cast<const float4*>(ls + (0 * KS + k) * C + c);
                        const float4 w1 = *reinterpret_cast<const float4*>(ls + (1 * KS + k) * C + c);
                        const float4 v0 = win[k].v[i][0], v1 = win[k].v[i][GIN - 1];
                        r.x += w0.x * v0.x + w1.x * v0.y;
                        r.y += w0.y * v0.z + w1.y * v0.w;
                        r.z += w0.z * v1.x + w1.z * v1.y;
                        r.w += w0.w * v1.z + w1.w * v1.w;
                    }
                }
                acc[i] = make_float4(r.x * mk, r.y * mk, r.z * mk, r.w * mk);
            }
            if (p.gamma[o]) {
                ln_rows<NV, WIDE>(acc, ls + (NT + 1) * C, ls + (NT + 2) * C, lane, p.relu[o] != 0);
            } else if (p.relu[o]) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    acc[i].x = fmaxf(acc[i].x, 0.f); acc[i].y = fmaxf(acc[i].y, 0.f);
                    acc[i].z = fmaxf(acc[i].z, 0.f); acc[i].w = fmaxf(acc[i].w, 0.f);
                }
            }
            if (WIDE && p.out_pair[o]) {
                vrd::store_pair8(p.y[o] + row * p.ldy[o], lane * 8, acc[0], acc[NV - 1], p.out_pair[o], &rt);
            } else {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int c = lane_chan<NV, WIDE>(i, lane);
                    if (p.out_pair[o]) vrd::store_pair4(p.y[o] + row * p.ldy[o], c, 256 * NV, acc[i], p.out_pair[o], &rt);
                    else st4(p.y[o] + row * p.ldy[o] + c, acc[i]);
                }
            }
        }
        if (more) {
            if (KS == 1) {
                win[0] = nxt[0];
            } else if (p.stride == 1) {
                win[0] = win[1];
                win[KS > 1 ? 1 : 0] = win[KS - 1];
                win[KS - 1] = nxt[0];
            } else {
                win[0] = win[KS - 1];
                win[KS > 1 ? 1 : 0] = nxt[0];
                win[KS - 1] = nxt[1];
            }
        }
    }
    }       // strips of this wave
    rt.report(rflag, vrd::RANGE_DWCONV_LN);
}

// ------------------------------------------------------------------------------------------
// MaxPool1d(kernel 3, stride 2, padding 1 with -inf) * downsampled mask
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_mask_kernel(const float* __restrict__ x, int64_t ldx, int B, int Tin, int C,
                                                           const uint8_t* __restrict__ mask_in, float* __restrict__ y,
                                                           int64_t ldy, uint8_t* __restrict__ mask_out) {
    const int c4 = C / 4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int Tout = Tin / 2;
    if (idx >= (int64_t)B * Tout * c4) return;
    const int c = (int)(idx % c4) * 4;
    const int64_t row = idx / c4;
    const int b = (int)(row / Tout), to = (int)(row - (int64_t)b * Tout);
    const float* base = x + ((int64_t)b * Tin + 2 * to) * ldx + c;
    float4 m = ld4(base);
    const float4 r = ld4(base + ldx);          // 2*to + 1 < Tin always (Tin even)
    m.x = fmaxf(m.x, r.x); m.y = fmaxf(m.y, r.y); m.z = fmaxf(m.z, r.z); m.w = fmaxf(m.w, r.w);
    if (to > 0) {
        const float4 l = ld4(base - ldx);
        m.x = fmaxf(m.x, l.x); m.y = fmaxf(m.y, l.y); m.z = fmaxf(m.z, l.z); m.w = fmaxf(m.w, l.w);
    }
    const uint8_t mk = mask_in[(int64_t)b * Tin + 2 * to];
    const float f = (float)mk;
    st4(y + row * ldy + c, make_float4(m.x * f, m.y * f, m.z * f, m.w * f));
    if (c == 0 && mask_out) mask_out[row] = mk;
}

// ------------------------------------------------------------------------------------------
// mask head: seg[b,q,t] = <emb[b,q,:], feat[b,t,:]>, fill where the output mask is 0
// ------------------------------------------------------------------------------------------
constexpr int MH_D = 256, MH_TT = 32, MH_QMAX = 16;

__global__ __launch_bounds__(256) void mask_head_kernel(const float* __restrict__ emb, int64_t ld_emb,
                                                        const float* __restrict__ feat, int64_t ld_feat,
                                                        const uint8_t* __restrict__ out_mask, int Q, int T, float fill,
                                                        float* __restrict__ seg) {
    __shared__ float fs[MH_TT][MH_D + 1];
    __shared__ float es[MH_QMAX][MH_D];
    const int b = blockIdx.y, t0 = blockIdx.x * MH_TT, tid = threadIdx.x;
    for (int i = tid; i < MH_TT * (MH_D / 4); i += 256) {
        const int r = i / (MH_D / 4), c = (i % (MH_D / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + r < T) v = ld4(feat + ((int64_t)b * T + t0 + r) * ld_feat + c);
        fs[r][c] = v.x; fs[r][c + 1] = v.y; fs[r][c + 2] = v.z; fs[r][c + 3] = v.w;
    }
    for (int i = tid; i < Q * (MH_D / 4); i += 256) {
        const int r = i / (MH_D / 4), c = (i % (MH_D / 4)) * 4;
        st4(&es[r][c], ld4(emb + ((int64_t)b * Q + r) * ld_emb + c));
    }
    __syncthreads();
    const int tt = tid & 31, qg = tid >> 5;
    if (t0 + tt >= T) return;
    const bool on = out_mask[(int64_t)b * T + t0 + tt] != 0;
    for (int q = qg; q < Q; q += 8) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < MH_D; ++c) s = fmaf(es[q][c], fs[tt][c], s);
        seg[((int64_t)b * Q + q) * T + t0 + tt] = on ? s : fill;
    }
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

// ---- padding map: 32-row blocks dealt into segments of seg_len list entries (the last one may be shorter), inside a
// segment the blocks with a valid row first (one workgroup; rows/32 is a few ten thousand at most).  Thread t owns a
// contiguous run of blocks, so every part comes out ascending.  Segment s starts at entry start(s) = min(s * seg_len,
// nblocks) and gets the actives of rank first(s) .. first(s+1)-1, first(s) = floor(total * start(s) / nblocks) -- its
// proportional share, never more than it has entries; the padded blocks fill the rest of each segment, in order.
constexpr int RB_THREADS = 1024;
__global__ __launch_bounds__(RB_THREADS) void row_blocks_kernel(const uint8_t* __restrict__ mask, int nblocks, int seg_len, int S,
                                                                int32_t* __restrict__ order, int32_t* __restrict__ n_active) {
    __shared__ int counts[RB_THREADS];
    const int tid = threadIdx.x;
    const int per = (nblocks + RB_THREADS - 1) / RB_THREADS;
    const int b0 = min(tid * per, nblocks), b1 = min(b0 + per, nblocks);
    auto active = [&](int b) {
        const uint4* q = reinterpret_cast<const uint4*>(mask + (int64_t)b * 32);
        const uint4 lo = q[0], hi = q[1];
        return ((lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) != 0u) ? 1 : 0;
    };
    int mine = 0;
    for (int b = b0; b < b1; ++b) mine += active(b);
    counts[tid] = mine;
    __syncthreads();
    // inclusive scan (Hillis-Steele over 1024 entries)
    for (int d = 1; d < RB_THREADS; d <<= 1) {
        const int v = tid >= d ? counts[tid - d] : 0;
        __syncthreads();
        counts[tid] += v;
        __syncthreads();
    }
    const int total = counts[RB_THREADS - 1];
    auto start = [&](int s) { return min(s * seg_len, nblocks); };
    auto first = [&](int s) { return (int)(((int64_t)total * start(s)) / nblocks); };     // actives before segment s
    auto pads_before = [&](int s) { return start(s) - first(s); };
    int a = counts[tid] - mine;                       // rank of this run's first active / padded block
    int i = b0 - a;
    int sa = 0, sp = 0;
    for (int b = b0; b < b1; ++b) {
        if (active(b)) {
            while (sa + 1 < S && first(sa + 1) <= a) ++sa;
            order[start(sa) + (a - first(sa))] = b;
            ++a;
        } else {
            while (sp + 1 < S && pads_before(sp + 1) <= i) ++sp;
            order[start(sp) + (first(sp + 1) - first(sp)) + (i - pads_before(sp))] = b;
            ++i;
        }
    }
    if (tid < S) n_active[tid] = first(tid + 1) - first(tid);
}

}  // namespace

extern "C" {

int vrd_bct_to_btc(const float* src, int B, int C_total, int T, int c0, int count, float* dst, int64_t ld_dst,
                   int out_pair, int T_src, const int32_t* src_batch, void* stream) {
    VRD_CHECK_ARG(src && dst, "vrd_bct_to_btc: null pointer");
    if (T_src == 0) T_src = T;
    VRD_CHECK_ARG(T_src >= T, "vrd_bct_to_btc: T_src (%d) < T (%d)", T_src, T);
    VRD_CHECK_ARG(B > 0 && T > 0 && count > 0 && c0 >= 0 && c0 + count <= C_total && ld_dst >= count,
                  "vrd_bct_to_btc: bad slab c0=%d count=%d C=%d ld=%lld", c0, count, C_total, (long long)ld_dst);
    VRD_CHECK_ARG(B <= 65535, "vrd_bct_to_btc: B too large for one launch (%d)", B);
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, 8.0 * B * (double)count * T);
    dim3 grid((T + 63) / 64, (count + 63) / 64, B);
    VRD_CHECK_ARG(!out_pair || (count % 32 == 0 && ld_dst % 4 == 0 && aligned16(dst)), "vrd_bct_to_btc: pair rows need count %% 32 == 0");
    const bool vec = T % 4 == 0 && T_src % 4 == 0 && count % 4 == 0 && ld_dst % 4 == 0 && aligned16(dst) && aligned16(src);
    if (vec) hipLaunchKernelGGL(bct_to_btc_vec_kernel, grid, dim3(256), 0, s, src, C_total, T, c0, count, dst, ld_dst, out_pair, T_src, src_batch, out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    else hipLaunchKernelGGL(bct_to_btc_kernel, grid, dim3(256), 0, s, src, C_total, T, c0, count, dst, ld_dst, out_pair, T_src, src_batch, out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_pack_pairs(const vrd_pack_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->src && a->lens && a->vis && a->so_box && a->ent, "vrd_pack_pairs: null pointer");
    VRD_CHECK_ARG(a->P > 0 && a->T > 0 && a->V > 0 && a->S > 0 && a->E > 0 && a->Cc >= 0, "vrd_pack_pairs: bad sizes");
    VRD_CHECK_ARG(a->C_in == 2 * a->V + 2 * a->Cc + a->S + 2 * a->E, "vrd_pack_pairs: C_in %d does not match the slab widths", a->C_in);
    VRD_CHECK_ARG(a->Cc == 0 || a->clip, "vrd_pack_pairs: clip buffer missing");
    VRD_CHECK_ARG(!a->pair_wide || (a->V % 32 == 0 && a->Cc % 32 == 0), "vrd_pack_pairs: pair rows need widths %% 32 == 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)a->P * a->T;
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, 8.0 * (double)rows * a->C_in);
    hipLaunchKernelGGL(pack_pairs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *a, a->pair_wide == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_gather_pairs(const vrd_gather_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->vis && a->boxes && a->s_row && a->o_row && a->lens && a->out_so_box && a->out_ent,
                  "vrd_gather_pairs: null pointer");
    VRD_CHECK_ARG(a->P > 0 && a->T > 0 && a->V > 0 && a->Cc >= 0 && a->stride >= 1 && a->w > 0.f && a->h > 0.f, "vrd_gather_pairs: bad sizes");
    VRD_CHECK_ARG(a->Cc == 0 || !a->out_vis || (a->clip && a->out_clip), "vrd_gather_pairs: clip buffers missing");
    VRD_CHECK_ARG(!a->pair_wide || (a->V % 32 == 0 && a->Cc % 32 == 0), "vrd_gather_pairs: pair rows need widths %% 32 == 0");
    VRD_CHECK_ARG(aligned16(a->boxes), "vrd_gather_pairs: boxes must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = (int64_t)a->P * a->T;
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, 8.0 * (double)rows * ((a->out_vis ? 2 * a->V + 2 * a->Cc : 0) + 21));
    hipLaunchKernelGGL(gather_pairs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *a, a->pair_wide == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_assemble_pairs(const vrd_assemble_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->streams && a->snippets && a->stream_row && a->lens && a->out, "vrd_assemble_pairs: null pointer");
    VRD_CHECK_ARG(a->P > 0 && a->T > 0 && a->D > 0 && a->D % 4 == 0, "vrd_assemble_pairs: bad sizes");
    VRD_CHECK_ARG(a->reach >= 0 && a->piece >= 2 * a->reach && a->piece >= 1 && a->L > a->piece && a->T >= a->L,
                  "vrd_assemble_pairs: pieces of %d frames in buffers of %d (T = %d) cannot cover a reach of %d", a->piece, a->L, a->T, a->reach);
    VRD_CHECK_ARG(aligned16(a->streams) && aligned16(a->snippets) && aligned16(a->out), "vrd_assemble_pairs: rows must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t rows = 2 * (int64_t)a->P * a->T;
    VRD_CHECK_ARG((rows + 3) / 4 < ((int64_t)1 << 31), "vrd_assemble_pairs: grid too large");
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, 8.0 * (double)rows * a->D);
    hipLaunchKernelGGL(assemble_pairs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, *a);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_row_blocks(const uint8_t* mask, int64_t rows, int seg_len, int32_t* order, int32_t* n_active, void* stream) {
    VRD_CHECK_ARG(mask && order && n_active, "vrd_row_blocks: null pointer");
    VRD_CHECK_ARG(rows > 0 && rows % 32 == 0 && rows / 32 < (1 << 30), "vrd_row_blocks: rows %lld must be a positive multiple of 32", (long long)rows);
    const int nblocks = (int)(rows / 32);
    VRD_CHECK_ARG(seg_len >= 1 && (nblocks + seg_len - 1) / seg_len <= 64, "vrd_row_blocks: segment length %d gives more than 64 segments", seg_len);
    VRD_CHECK_ARG(aligned16(mask), "vrd_row_blocks: mask must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, (double)rows + 4.0 * nblocks);
    hipLaunchKernelGGL(row_blocks_kernel, dim3(1), dim3(RB_THREADS), 0, s, mask, nblocks, seg_len, (nblocks + seg_len - 1) / seg_len, order, n_active);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_btc_to_bct(const float* src, int64_t ld_src, int B, int C, int T, float* dst, void* stream) {
    VRD_CHECK_ARG(src && dst, "vrd_btc_to_bct: null pointer");
    VRD_CHECK_ARG(B > 0 && B <= 65535 && T > 0 && C > 0 && ld_src >= C, "vrd_btc_to_bct: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_TRANSPOSE, s, 0.0, 8.0 * B * (double)C * T);
    dim3 grid((T + 63) / 64, (C + 63) / 64, B);
    hipLaunchKernelGGL(btc_to_bct_kernel, grid, dim3(256), 0, s, src, ld_src, C, T, dst);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, const float* gamma,
                  const float* beta, int relu, const float* post_add, int64_t ld_add, int add_period, int out_pair,
                  void* stream) {
    VRD_CHECK_ARG(x && y && gamma && beta, "vrd_layernorm: null pointer");
    VRD_CHECK_ARG(C == 256 || C == 512, "vrd_layernorm: C must be 256 or 512 (got %d)", C);
    VRD_CHECK_ARG(ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(y) &&
                      aligned16(gamma) && aligned16(beta),
                  "vrd_layernorm: rows must be 16-byte aligned");
    VRD_CHECK_ARG(!post_add || (add_period > 0 && ld_add % 4 == 0 && aligned16(post_add)), "vrd_layernorm: bad post_add");
    if (rows <= 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_LAYERNORM, s, 0.0, 8.0 * (double)rows * C);
    unsigned* const rflag = out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr;
    // workgroups: at most VRD_LN_BLOCKS (default 32 per CU: the waves then walk the rows with the grid's stride; 7.4 -> 7.1 ms per step), one row per wave below that
    static const int64_t max_blocks = [] { const char* e = getenv("VRD_LN_BLOCKS"); return e ? atoll(e) : 8192; }();
    const int64_t want = (rows + 3) / 4;
    dim3 grid((unsigned)(max_blocks > 0 && want > max_blocks ? max_blocks : want));
    if (C == 256)
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, x, ldx, y, ldy, rows, gamma, beta, relu, post_add, ld_add, add_period, out_pair, rflag);
    else
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, ldx, y, ldy, rows, gamma, beta, relu, post_add, ld_add, add_period, out_pair, rflag);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_conv_ln(const vrd_conv_ln_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->x && a->w && a->y, "vrd_conv_ln: null pointer");
    VRD_CHECK_ARG(a->N == 256 || a->N == 512, "vrd_conv_ln: N must be 256 or 512 (got %d)", a->N);
    VRD_CHECK_ARG((a->taps == 1 || a->taps == 3) && a->Cin > 0 && a->Cin * a->taps <= CL_MAXK && (4 + a->taps - 1) * a->Cin <= 64,
                  "vrd_conv_ln: taps in {1, 3}, taps * Cin <= %d and (3 + taps) * Cin <= 64 (got %d x %d)", CL_MAXK, a->taps, a->Cin);
    VRD_CHECK_ARG(a->rows >= 0 && a->T > 0 && a->rows % a->T == 0 && a->ldx >= a->Cin, "vrd_conv_ln: bad rows / T / ldx");
    VRD_CHECK_ARG((a->gamma == nullptr) == (a->beta == nullptr), "vrd_conv_ln: gamma and beta go together");
    VRD_CHECK_ARG(a->ldy >= a->N && a->ldy % 4 == 0 && aligned16(a->y) && aligned16(a->gamma) && aligned16(a->beta),
                  "vrd_conv_ln: output rows and LayerNorm parameters must be 16-byte aligned");
    VRD_CHECK_ARG(!a->out_pair || (a->ldy % 32 == 0 && (a->out_pair == VRD_PAIR_BF16 || a->out_pair == VRD_PAIR_F16)),
                  "vrd_conv_ln: pair output needs rows that start on a 128-byte block");
    if (a->rows == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int K = a->Cin * a->taps;
    vrd::ProfScope prof(VRD_K_LAYERNORM, s, 2.0 * (double)a->rows * a->N * K, 4.0 * (double)a->rows * (a->N + a->Cin));
    unsigned* const rflag = a->out_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr;
    const size_t lds = (size_t)(K + 1) * a->N * sizeof(float);
    const int64_t want = (a->rows + 15) / 16;                   // (workgroup = 4 waves x 4 rows per trip)
    dim3 grid((unsigned)(want > 4096 ? 4096 : want));          // (the weights are staged per workgroup: a bounded grid, row groups by its stride)
    if (a->N == 256) {
        if (lds > 48 * 1024)
            if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(conv_ln_kernel<1>), lds, "vrd_conv_ln")) return rc;
        hipLaunchKernelGGL(conv_ln_kernel<1>, grid, dim3(256), lds, s, *a, rflag);
    } else {
        if (lds > 48 * 1024)
            if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(conv_ln_kernel<2>), lds, "vrd_conv_ln")) return rc;
        hipLaunchKernelGGL(conv_ln_kernel<2>, grid, dim3(256), lds, s, *a, rflag);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_dwconv_ln(const vrd_dwconv_ln_args* a, void* stream) {
    VRD_CHECK_ARG(a && a->x, "vrd_dwconv_ln: null args");
    VRD_CHECK_ARG(a->C == 256 || a->C == 512, "vrd_dwconv_ln: C must be 256 or 512 (got %d)", a->C);
    VRD_CHECK_ARG(a->ksize == 1 || a->ksize == 3, "vrd_dwconv_ln: ksize must be 1 or 3");
    VRD_CHECK_ARG(a->stride == 1 || a->stride == 2, "vrd_dwconv_ln: stride must be 1 or 2");
    VRD_CHECK_ARG(a->group_in == 1 || (a->group_in == 2 && a->C == 256 && a->ksize == 3),
                  "vrd_dwconv_ln: group_in=2 is only built for C=256, ksize=3");
    VRD_CHECK_ARG(a->segs || (a->B > 0 && a->Tin > 0 && a->Tin % a->stride == 0), "vrd_dwconv_ln: Tin %% stride != 0");
    VRD_CHECK_ARG(a->n_out >= 1 && a->n_out <= 3, "vrd_dwconv_ln: n_out must be 1..3");
    VRD_CHECK_ARG((a->pre_gamma == nullptr) == (a->pre_beta == nullptr) && (!a->pre_gamma || (a->group_in == 1 && !a->x_up)),
                  "vrd_dwconv_ln: input LayerNorm needs gamma and beta, group_in == 1 and no x_up");
    VRD_CHECK_ARG(a->ldx >= (int64_t)a->C * a->group_in && a->ldx % 4 == 0 && aligned16(a->x), "vrd_dwconv_ln: bad x layout");
    VRD_CHECK_ARG(!a->x_up || (a->Tin % 2 == 0 && a->ldx_up % 4 == 0 && aligned16(a->x_up)), "vrd_dwconv_ln: bad x_up layout");
    for (int o = 0; o < a->n_out; ++o) {
        VRD_CHECK_ARG((a->w[o] || a->packed[o]) && a->y[o] && aligned16(a->w[o]) && aligned16(a->packed[o]) && aligned16(a->y[o]) &&
                          a->ldy[o] % 4 == 0 && a->ldy[o] >= a->C,
                      "vrd_dwconv_ln: bad output set %d", o);
        VRD_CHECK_ARG((a->gamma[o] == nullptr) == (a->beta[o] == nullptr), "vrd_dwconv_ln: gamma/beta mismatch");
    }
    vrd::SegTable sg;
    sg.count = 0;
    int64_t rows_in = (int64_t)a->B * a->Tin;
    int Tmax = a->Tin;
    if (a->segs) {
        const int even = (a->x_up || a->stride == 2) ? 2 : 1;
        VRD_CHECK_ARG(vrd::seg_table(sg, a->segs, even, 16) >= 0, "vrd_dwconv_ln: bad row groups (1..%d groups, T %% %d == 0)", VRD_MAX_SEGS, even);
        rows_in = 0, Tmax = 0;
        for (int g = 0; g < sg.count; ++g) {
            VRD_CHECK_ARG(sg.row[g] % even == 0, "vrd_dwconv_ln: a group's first row must be even with stride 2 / x_up");
            rows_in += (int64_t)sg.n[g] * sg.T[g];
            Tmax = sg.T[g] > Tmax ? sg.T[g] : Tmax;
        }
    }
    const int Tout = Tmax / a->stride;
    const int64_t rows = rows_in / a->stride;
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_DWCONV_LN, s, 0.0,
                        4.0 * ((double)rows_in * a->C * a->group_in * (a->x_up ? 1.5 : 1.0) + (double)rows * a->C * a->n_out));
    const size_t lds = ((size_t)a->n_out * (a->group_in * a->ksize + 3) + (a->pre_gamma ? 2 : 0)) * a->C * sizeof(float);
    // rows per strip: 16 when the launch is many rounds of resident waves; for a small launch (e.g. one GPU's
    // 256-pair shard: 4608 strips of 16 against 4096 resident waves = two rounds for 1.1 rounds of work) the strip
    // length 8..32 that minimises rounds x (rows per strip + ~3 rows' worth of per-strip setup)
    static const int rw_env = [] { const char* e = getenv("VRD_DW_RW"); return e ? atoi(e) : 0; }();
    int rw = DW_RW;
    {
        const int64_t wgs_per_cu = lds ? (int64_t)(160 * 1024 / lds) : 8;
        // (the 512-channel variants use ~110 VGPRs: four waves per SIMD, i.e. four workgroups per CU at most)
        const int64_t resident = 256 * (wgs_per_cu < 1 ? 1 : wgs_per_cu > 4 ? 4 : wgs_per_cu) * 4;      // waves
        auto strips_at = [&](int r) -> int64_t {
            return a->segs ? vrd::seg_table(sg, a->segs, a->stride, r) : (int64_t)a->B * ((Tout + r - 1) / r);
        };
        if (strips_at(DW_RW) < 6 * resident) {
            double best = 1e30;
            for (int r = 8; r <= 32; ++r) {
                const int64_t n = strips_at(r);
                const double cost = (double)((n + resident - 1) / resident) * (r + 3);
                if (cost < best - 1e-9) best = cost, rw = r;
            }
        }
        if (rw_env >= 1 && rw_env <= 64) rw = rw_env;
    }
    const int strips = (Tout + rw - 1) / rw;
    // strips per wave (VRD_DW_SPW): the parameter fill (12 KiB per set, a straight copy of the caller's image) is per
    // workgroup, but measured at the benchmark shape 2 or 3 strips per wave are 5 % SLOWER than 1 (16.9 / 17.1 vs
    // 16.1 ms per step): the kernel runs at ~4.8 TB/s, three quarters of it writes, and more, shorter workgroups
    // keep more of them in flight
    static const int spw_env = [] { const char* e = getenv("VRD_DW_SPW"); return e ? atoi(e) : 0; }();
    const int64_t total_strips = a->segs ? vrd::seg_table(sg, a->segs, a->stride, rw) : (int64_t)a->B * strips;
    const int spw = spw_env > 0 ? spw_env : 1;
    dim3 grid((unsigned)((total_strips + 4 * spw - 1) / (4 * spw))), block(256);
    bool any_f16 = false;
    for (int o = 0; o < a->n_out; ++o) any_f16 |= a->out_pair[o] == VRD_PAIR_F16;
    unsigned* const rflag = any_f16 ? vrd::range_flag() : nullptr;
#define VRD_DW(NV, KS, GIN) hipLaunchKernelGGL((dwconv_ln_kernel<NV, KS, GIN>), grid, block, lds, s, *a, Tout, strips, spw, rw, rflag, sg)
    if (a->group_in == 2) VRD_DW(1, 3, 2);
    else if (a->C == 256 && a->ksize == 3) VRD_DW(1, 3, 1);
    else if (a->C == 256) VRD_DW(1, 1, 1);
    else if (a->ksize == 3) VRD_DW(2, 3, 1);
    else VRD_DW(2, 1, 1);
#undef VRD_DW
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_maxpool_mask(const float* x, int64_t ldx, int B, int Tin, int C, const uint8_t* mask_in, float* y, int64_t ldy,
                     uint8_t* mask_out, void* stream) {
    VRD_CHECK_ARG(x && y && mask_in, "vrd_maxpool_mask: null pointer");
    VRD_CHECK_ARG(B > 0 && Tin > 0 && Tin % 2 == 0 && C % 4 == 0, "vrd_maxpool_mask: Tin must be even, C %% 4 == 0");
    VRD_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && aligned16(x) && aligned16(y), "vrd_maxpool_mask: bad layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)B * (Tin / 2) * (C / 4);
    vrd::ProfScope prof(VRD_K_POOL, s, 0.0, 4.0 * (double)B * Tin * C * 1.5);
    hipLaunchKernelGGL(maxpool_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, B, Tin, C, mask_in, y, ldy, mask_out);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_mask_head(const float* emb, int64_t ld_emb, const float* feat, int64_t ld_feat, const uint8_t* out_mask, int B,
                  int Q, int T, int Dp, float fill, float* seg, void* stream) {
    VRD_CHECK_ARG(emb && feat && out_mask && seg, "vrd_mask_head: null pointer");
    VRD_CHECK_ARG(Dp == MH_D, "vrd_mask_head: mask dim must be %d (got %d)", MH_D, Dp);
    VRD_CHECK_ARG(Q >= 1 && Q <= MH_QMAX, "vrd_mask_head: num queries must be 1..%d (got %d)", MH_QMAX, Q);
    VRD_CHECK_ARG(B > 0 && B <= 65535 && T > 0, "vrd_mask_head: bad B/T");
    VRD_CHECK_ARG(ld_emb % 4 == 0 && ld_feat % 4 == 0 && aligned16(emb) && aligned16(feat), "vrd_mask_head: bad layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_MASK_HEAD, s, 2.0 * B * (double)Q * T * Dp, 4.0 * B * ((double)T * Dp + (double)Q * Dp + (double)Q * T));
    hipLaunchKernelGGL(mask_head_kernel, dim3((T + MH_TT - 1) / MH_TT, B), dim3(256), 0, s, emb, ld_emb, feat, ld_feat, out_mask, Q, T, fill, seg);
    VRD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
