// The training criterion on the device: matching costs, losses and their gradients as three kernels
// for ALL decoder layers of a step, instead of ~70 tensor operations per layer.
//
// Reference: models/maskvrd.py:417-496 (bipartite_match: cost_class + cost_mask + cost_dice per pair), :498-588 (loss_labels,
// loss_masks over the final and the three auxiliary heads), models/losses.py:4-354 (focal / dice, plain and "fuzzy" targets).
//
//   vrd_criterion_costs     cost[l][g][q] = w_class * (-log softmax(logits_l[b, q])[id_g]) + w_mask * focal cost + w_dice * dice
//                           cost of giving relation g (of pair b = owner[g]) to query q, for every layer l: one wave per
//                           (relation, query, layer).  Entry [g, q] is entry [b*Q + q, g] of the reference's matrices.
//   (vrd_assign, vrd_train_tail.hip, turns the costs into query_of[l][g].)
//   vrd_criterion_losses    per layer: class-weighted cross-entropy over all (pair, query) rows, masked focal loss and dice
//                           loss of the matched rows; one workgroup per layer, sums in a fixed order (no atomics).
//   vrd_criterion_backward  d(sum_l sum_i gout[l][i] * loss[l][i]) / d logits_l, d masks_l: one wave per (pair, query, layer) row.
//
// All arithmetic is f32 like the reference's; softplus follows torch's (linear above 20).  gamma = 2 is evaluated as a square.
#include "vrd_common.h"
#include <cmath>

namespace {

__device__ __forceinline__ bool class_ok(const vrd_criterion_args& a, int g) {
    return (unsigned long long)a.tgt_ids[g] < (unsigned long long)a.K1;
}

constexpr float PI_F = 3.14159265358979323846f;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float softplusf_(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float powg(float v, float gamma) { return gamma == 2.0f ? v * v : powf(v, gamma); }
// d/dv v^gamma
__device__ __forceinline__ float dpowg(float v, float gamma) { return gamma == 2.0f ? 2.0f * v : gamma * powf(v, gamma - 1.0f); }

// soft relation mask of losses.py:214-227 at frame t: 1 inside the segment shrunk by scale_range, a sqrt(cos) ramp out to
// the segment widened by 1 / scale_range (valid frames only), 0 outside
__device__ __forceinline__ float target_at(const vrd_criterion_args& a, int g, int t, bool valid) {
    const float hard = a.tgt_masks[(int64_t)g * a.T + t];
    if (!a.segs) return hard;
    const int lo = a.segs[2 * g], hi = a.segs[2 * g + 1];
    const float centre = (float)(hi - 1 + lo) / 2.0f;
    const float off = (float)t - centre;
    const float length = (float)(hi - lo);
    const bool core = fabsf(off) < (length / 2.0f * a.scale_range);
    const bool wide = (fabsf(off) < (length / 2.0f / a.scale_range)) && valid;
    const bool ramp = (wide != core) && valid;
    float w = cosf(PI_F / (length / a.scale_range) * off);
    w = sqrtf(w > 0.0f ? w : 0.0f);
    return (ramp ? w : 0.0f) + (core ? hard : 0.0f);
}

// log-sum-exp of a row of K values, by one wave
__device__ __forceinline__ float wave_lse(const float* __restrict__ row, int K, int lane) {
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(row[k] - mx);
    s = vrd::wave_sum(s);
    return mx + logf(s);
}

__global__ void criterion_costs_kernel(vrd_criterion_args a, float* __restrict__ cost) {
    const int g = blockIdx.x, l = blockIdx.y;
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = a.owner[g];
    const float* x = a.masks[l] + ((int64_t)b * a.Q + q) * a.T;
    const uint8_t* valid = a.out_valid + (int64_t)b * a.T;
    float s_pos = 0.f, s_neg = 0.f, s_sig = 0.f, s_st = 0.f, s_t = 0.f, n_valid = 0.f;
    for (int t = lane; t < a.T; t += 64) {
        if (!valid[t]) continue;
        const float xt = x[t], p = sigmoidf_(xt);
        float pos = powg(1.0f - p, a.gamma) * softplusf_(-xt), neg = powg(p, a.gamma) * softplusf_(xt);
        if (a.alpha >= 0.f) pos *= a.alpha, neg *= 1.0f - a.alpha;
        const float tp = target_at(a, g, t, true);
        s_pos += pos * tp;
        s_neg += neg * (1.0f - tp);
        s_sig += p;
        s_st += p * tp;
        s_t += tp;
        n_valid += 1.0f;
    }
    s_pos = vrd::wave_sum(s_pos);
    s_neg = vrd::wave_sum(s_neg);
    s_sig = vrd::wave_sum(s_sig);
    s_st = vrd::wave_sum(s_st);
    s_t = vrd::wave_sum(s_t);
    n_valid = vrd::wave_sum(n_valid);
    const float* lr = a.logits[l] + ((int64_t)b * a.Q + q) * a.K1;
    const float lse = wave_lse(lr, a.K1, lane);
    if (lane == 0) {
        // a class id outside [0, K1) (F.cross_entropy raises on one): NaN cost -> the pair comes back unassigned -> poisoned losses
        const float c_class = class_ok(a, g) ? lse - lr[a.tgt_ids[g]] : __builtin_nanf("");
        const float c_mask = (s_pos + s_neg) / n_valid;
        const float c_dice = 1.0f - (2.0f * s_st + 1.0f) / (s_sig + s_t + 1.0f);
        cost[((int64_t)l * a.G + g) * a.Q + q] = a.w_class * c_class + a.w_mask * c_mask + a.w_dice * c_dice;
    }
}

// focal term and its pieces at one valid frame of a matched row
struct FocalAt {
    float p, tp, ce_t, ce, one_m_pt, A;
};
__device__ __forceinline__ FocalAt focal_at(const vrd_criterion_args& a, int g, int t, float xt) {
    FocalAt f;
    f.p = sigmoidf_(xt);
    f.tp = target_at(a, g, t, true);
    f.ce_t = f.tp;                                                       // (valid frame: tp * loss_mask = tp)
    f.ce = fmaxf(xt, 0.f) - xt * f.ce_t + log1pf(expf(-fabsf(xt)));      // binary_cross_entropy_with_logits
    f.one_m_pt = 1.0f - (f.p * f.tp + (1.0f - f.p) * (1.0f - f.tp));
    f.A = a.alpha >= 0.f ? a.alpha * f.tp + (1.0f - a.alpha) * (1.0f - f.tp) : 1.0f;
    return f;
}

constexpr int LOSS_WAVES = 16;
__global__ __launch_bounds__(LOSS_WAVES * 64) void criterion_losses_kernel(vrd_criterion_args a, const int32_t* __restrict__ query_of,
                                                                         const float* __restrict__ class_weight, float num_masks,
                                                                         float* __restrict__ out) {
    extern __shared__ int sm_i[];
    const int l = blockIdx.x, rows = a.B * a.Q;
    int* const tgt_cls = sm_i;                                            // class of every (pair, query) row: 0 = no relation
    float* const part = reinterpret_cast<float*>(sm_i + rows);            // [LOSS_WAVES][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t* qo = query_of + (int64_t)l * a.G;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) tgt_cls[r] = 0;
    __syncthreads();
    for (int g = threadIdx.x; g < a.G; g += blockDim.x) {
        const int q = qo[g] < 0 ? 0 : qo[g];
        tgt_cls[a.owner[g] * a.Q + q] = class_ok(a, g) ? (int)a.tgt_ids[g] : 0;       // (never an index out of range)
    }
    __syncthreads();
    // ---- class term: sum_r w[t_r] * (lse_r - x_r[t_r]) / sum_r w[t_r]
    float num = 0.f, den = 0.f;
    for (int r = wave; r < rows; r += LOSS_WAVES) {
        const float* lr = a.logits[l] + (int64_t)r * a.K1;
        const float lse = wave_lse(lr, a.K1, lane);
        const int t = tgt_cls[r];
        const float w = class_weight[t];
        num += w * (lse - lr[t]);
        den += w;
    }
    // ---- matched rows: focal (mean over ALL T frames of the masked term) and dice
    float focal = 0.f, dice = 0.f;
    for (int g = wave; g < a.G; g += LOSS_WAVES) {
        const int b = a.owner[g], q = qo[g] < 0 ? 0 : qo[g];
        const float* x = a.masks[l] + ((int64_t)b * a.Q + q) * a.T;
        const uint8_t* valid = a.out_valid + (int64_t)b * a.T;
        float s_f = 0.f, s_pt = 0.f, s_p = 0.f, s_t = 0.f;
        for (int t = lane; t < a.T; t += 64) {
            if (!valid[t]) continue;
            const FocalAt f = focal_at(a, g, t, x[t]);
            s_f += f.A * f.ce * powg(f.one_m_pt, a.gamma);
            s_pt += f.p * f.tp;
            s_p += f.p;
            s_t += f.tp;
        }
        s_f = vrd::wave_sum(s_f);
        s_pt = vrd::wave_sum(s_pt);
        s_p = vrd::wave_sum(s_p);
        s_t = vrd::wave_sum(s_t);
        focal += s_f / (float)a.T;
        dice += 1.0f - (2.0f * s_pt + 1.0f) / (s_p + s_t + 1.0f);
    }
    if (lane == 0) {
        part[wave * 4 + 0] = num;
        part[wave * 4 + 1] = den;
        part[wave * 4 + 2] = focal;
        part[wave * 4 + 3] = dice;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < LOSS_WAVES; ++w)
            for (int i = 0; i < 4; ++i) s[i] += part[w * 4 + i];
        out[l * 4 + 0] = s[0] / s[1];
        out[l * 4 + 1] = s[2] / num_masks;
        out[l * 4 + 2] = s[3] / num_masks;
        out[l * 4 + 3] = s[1];                       // the weight sum: the backward's normaliser of the class term
    }
}

__global__ __launch_bounds__(64) void criterion_backward_kernel(vrd_criterion_args a, const int32_t* __restrict__ query_of,
                                                               const float* __restrict__ class_weight, float num_masks,
                                                               const float* __restrict__ fwd_out, const float* __restrict__ gout,
                                                               vrd_criterion_grads gr) {
    const int r = blockIdx.x, l = blockIdx.y, lane = threadIdx.x;
    const int b = r / a.Q, q = r - b * a.Q;
    const int32_t* qo = query_of + (int64_t)l * a.G;
    // the relation this row was matched to, if any (at most one)
    int g_match = -1;
    for (int g0 = 0; g0 < a.G; g0 += 64) {
        const int g = g0 + lane;
        const bool hit = g < a.G && a.owner[g] == b && (qo[g] < 0 ? 0 : qo[g]) == q;
        const unsigned long long bal = __ballot(hit);
        if (bal) g_match = g0 + (int)__builtin_ctzll(bal);       // (several hits only after a failed assignment: first wins)
    }
    const float g_class = gout[l * 3 + 0], g_focal = gout[l * 3 + 1], g_dice = gout[l * 3 + 2];
    // ---- class term: g_class * w[t] / W * (softmax - onehot)
    {
        const float* lr = a.logits[l] + (int64_t)r * a.K1;
        float* go = gr.logits[l] + (int64_t)r * a.K1;
        const int t = (g_match >= 0 && class_ok(a, g_match)) ? (int)a.tgt_ids[g_match] : 0;
        const float lse = wave_lse(lr, a.K1, lane);
        const float coef = g_class * class_weight[t] / fwd_out[l * 4 + 3];
        for (int k = lane; k < a.K1; k += 64) go[k] = coef * (expf(lr[k] - lse) - (k == t ? 1.0f : 0.0f));
    }
    // ---- mask terms: only matched rows carry a gradient
    float* gm = gr.masks[l] + (int64_t)r * a.T;
    if (g_match < 0) {
        for (int t = lane; t < a.T; t += 64) gm[t] = 0.f;
        return;
    }
    const float* x = a.masks[l] + (int64_t)r * a.T;
    const uint8_t* valid = a.out_valid + (int64_t)b * a.T;
    float s_pt = 0.f, s_p = 0.f, s_t = 0.f;
    for (int t = lane; t < a.T; t += 64) {
        if (!valid[t]) continue;
        const float p = sigmoidf_(x[t]), tp = target_at(a, g_match, t, true);
        s_pt += p * tp;
        s_p += p;
        s_t += tp;
    }
    s_pt = vrd::wave_sum(s_pt);
    s_p = vrd::wave_sum(s_p);
    s_t = vrd::wave_sum(s_t);
    const float N = 2.0f * s_pt + 1.0f, D = s_p + s_t + 1.0f;
    const float cf = g_focal / ((float)a.T * num_masks), cd = g_dice / num_masks;
    for (int t = lane; t < a.T; t += 64) {
        if (!valid[t]) {
            gm[t] = 0.f;
            continue;
        }
        const float xt = x[t];
        const FocalAt f = focal_at(a, g_match, t, xt);
        const float dp = f.p * (1.0f - f.p);
        // focal = A * ce * (1 - p_t)^gamma;  d ce / dx = p - ce_t;  d (1 - p_t) / dx = -dp * (2 tp - 1)
        const float dfocal = f.A * ((f.p - f.ce_t) * powg(f.one_m_pt, a.gamma) - f.ce * dpowg(f.one_m_pt, a.gamma) * dp * (2.0f * f.tp - 1.0f));
        // dice = 1 - N / D, N = 2 sum p tp + 1, D = sum p + sum tp + 1
        const float ddice = -dp * (2.0f * f.tp * D - N) / (D * D);
        gm[t] = cf * dfocal + cd * ddice;
    }
}

int check_args(const vrd_criterion_args* a, const char* what) {
    VRD_CHECK_ARG(a, "%s: null arguments", what);
    VRD_CHECK_ARG(a->n_layers >= 1 && a->n_layers <= 4, "%s: 1 .. 4 layers (got %d)", what, a->n_layers);
    VRD_CHECK_ARG(a->B > 0 && a->Q >= 1 && a->Q <= 16 && a->K1 >= 2 && a->T >= 1 && a->G >= 1, "%s: bad sizes", what);
    for (int l = 0; l < a->n_layers; ++l) VRD_CHECK_ARG(a->logits[l] && a->masks[l], "%s: null prediction pointer (layer %d)", what, l);
    VRD_CHECK_ARG(a->out_valid && a->tgt_ids && a->tgt_masks && a->owner, "%s: null ground-truth pointer", what);
    VRD_CHECK_ARG(!a->segs || (a->scale_range > 0.f && a->scale_range <= 1.0f), "%s: scale_range must be in (0, 1]", what);
    return 0;
}

}  // namespace

extern "C" {

int vrd_criterion_costs(const vrd_criterion_args* a, float* cost, void* stream) {
    if (int rc = check_args(a, "vrd_criterion_costs")) return rc;
    VRD_CHECK_ARG(cost, "vrd_criterion_costs: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 0.0);
    hipLaunchKernelGGL(criterion_costs_kernel, dim3(a->G, a->n_layers), dim3(64 * a->Q), 0, s, *a, cost);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_criterion_losses(const vrd_criterion_args* a, const int32_t* query_of, const float* class_weight, float num_masks, float* out,
                         void* stream) {
    if (int rc = check_args(a, "vrd_criterion_losses")) return rc;
    VRD_CHECK_ARG(query_of && class_weight && out && num_masks > 0.f, "vrd_criterion_losses: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 0.0);
    const size_t lds = (size_t)a->B * a->Q * sizeof(int) + LOSS_WAVES * 4 * sizeof(float);
    VRD_CHECK_ARG(lds <= 48 * 1024, "vrd_criterion_losses: B * Q = %d rows do not fit the workgroup's table", a->B * a->Q);
    hipLaunchKernelGGL(criterion_losses_kernel, dim3(a->n_layers), dim3(LOSS_WAVES * 64), lds, s, *a, query_of, class_weight, num_masks, out);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_criterion_backward(const vrd_criterion_args* a, const int32_t* query_of, const float* class_weight, float num_masks,
                           const float* fwd_out, const float* gout, const vrd_criterion_grads* grads, void* stream) {
    if (int rc = check_args(a, "vrd_criterion_backward")) return rc;
    VRD_CHECK_ARG(query_of && class_weight && fwd_out && gout && grads && num_masks > 0.f, "vrd_criterion_backward: bad arguments");
    for (int l = 0; l < a->n_layers; ++l) VRD_CHECK_ARG(grads->logits[l] && grads->masks[l], "vrd_criterion_backward: null gradient pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 0.0);
    hipLaunchKernelGGL(criterion_backward_kernel, dim3(a->B * a->Q, a->n_layers), dim3(64), 0, s, *a, query_of, class_weight, num_masks,
                       fwd_out, gout, *grads);
    VRD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
