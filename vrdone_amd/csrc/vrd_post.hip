// Eval post-processing on the device: what MaskVRD.forward_test does per candidate in a
// Python loop with one device sync each (reference models/maskvrd.py:247-309), done here
// once per (pair, query) by one wavefront:
//   softmax over the K1 class logits, top-k over classes 1..K1-1 (class 0 = background),
//   and the first / last frame whose sigmoid(mask logit) > 0.5 inside the valid length.
// Every (query, class) candidate of a query shares that query's segment (maskvrd.py:252).
#include "vrd_common.h"
#include <cmath>

namespace {

constexpr int PP_MAX_K1 = 256;   // classes incl. background
constexpr int PP_MAX_TOPK = 16;

__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ logits, const float* __restrict__ masks,
                                                          const int32_t* __restrict__ valid_len, int PQ, int Q, int K1, int T,
                                                          int topk, float* __restrict__ top_score, int32_t* __restrict__ top_cat,
                                                          int32_t* __restrict__ seg_first, int32_t* __restrict__ seg_last) {
    const int lane = threadIdx.x & 63;
    const int pq = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (pq >= PQ) return;
    const int p = pq / Q;

    // ---- class softmax + top-k ----
    const float* lg = logits + (int64_t)pq * K1;
    float val[PP_MAX_K1 / 64];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PP_MAX_K1 / 64; ++i) {
        const int c = lane + 64 * i;
        val[i] = c < K1 ? lg[c] : -INFINITY;
        mx = fmaxf(mx, val[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < PP_MAX_K1 / 64; ++i) {
        val[i] = (lane + 64 * i < K1) ? expf(val[i] - mx) : 0.f;
        den += val[i];
    }
    den = vrd::wave_sum(den);
    if (lane == 0) val[0] = -1.f;            // background never competes (probs[..., 1:])
#pragma unroll
    for (int i = 0; i < PP_MAX_K1 / 64; ++i)
        if (lane + 64 * i >= K1) val[i] = -1.f;
    for (int r = 0; r < topk; ++r) {
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < PP_MAX_K1 / 64; ++i)
            if (val[i] > best) { best = val[i]; bi = lane + 64 * i; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            top_score[(int64_t)pq * topk + r] = best / den;
            top_cat[(int64_t)pq * topk + r] = bi;
        }
        if ((bi & 63) == lane) {
#pragma unroll
            for (int i = 0; i < PP_MAX_K1 / 64; ++i)
                if (i == (bi >> 6)) val[i] = -1.f;
        }
    }

    // ---- mask -> [first, last] active frame ----
    const float* mr = masks + (int64_t)pq * T;
    const int n = min(valid_len[p], T);
    int first = 0x7fffffff, last = -1;
    for (int t = lane; t < n; t += 64) {
        const float sg = 1.0f / (1.0f + expf(-mr[t]));
        if (sg > 0.5f) {
            first = min(first, t);
            last = max(last, t);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        first = min(first, __shfl_xor(first, off, 64));
        last = max(last, __shfl_xor(last, off, 64));
    }
    if (lane == 0) {
        seg_first[pq] = last < 0 ? -1 : first;
        seg_last[pq] = last;
    }
}

}  // namespace

extern "C" int vrd_postprocess(const float* logits, const float* masks, const int32_t* valid_len, int P, int Q, int K1,
                               int T, int topk, float* top_score, int32_t* top_cat, int32_t* seg_first,
                               int32_t* seg_last, void* stream) {
    VRD_CHECK_ARG(logits && masks && valid_len && top_score && top_cat && seg_first && seg_last, "vrd_postprocess: null pointer");
    VRD_CHECK_ARG(P > 0 && Q > 0 && T > 0, "vrd_postprocess: bad sizes");
    VRD_CHECK_ARG(K1 >= 2 && K1 <= PP_MAX_K1, "vrd_postprocess: K1 must be 2..%d (got %d)", PP_MAX_K1, K1);
    VRD_CHECK_ARG(topk >= 1 && topk <= PP_MAX_TOPK && topk <= K1 - 1, "vrd_postprocess: bad topk %d", topk);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int PQ = P * Q;
    vrd::ProfScope prof(VRD_K_POSTPROC, s, 0.0, 4.0 * PQ * ((double)K1 + T));
    hipLaunchKernelGGL(postprocess_kernel, dim3((PQ + 3) / 4), dim3(256), 0, s, logits, masks, valid_len, PQ, Q, K1, T, topk,
                       top_score, top_cat, seg_first, seg_last);
    VRD_LAUNCH_CHECK();
    return 0;
}
