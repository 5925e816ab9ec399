// Internal helpers shared by the kernel translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include "../../include/vrdone_hip.h"

namespace vrd {

void set_error(const char* fmt, ...);

// Opt `kernel` in to `bytes` of dynamic LDS on the CURRENT device (the attribute belongs to the function on a device).
// Remembered per (function, device) under the library mutex; 0 on success, -2 (error string set) otherwise.
int reserve_lds(const void* kernel, size_t bytes, const char* what);

// CUs of the CURRENT device (cached per device ordinal: a process may drive devices of different sizes / partition modes)
int device_cu_count();

// The device's image of a vrd_row_segs ragged row space for a kernel whose waves walk strips of `rw` rows of one sequence: group g
// = n[g] sequences of T[g] frames from row row[g] on, sps[g] strips per sequence, the group's first strip strip0[g].  Passed by
// value as a kernel argument (scalar loads); count == 0: the caller's uniform (B, T).
struct SegTable {
    int count;
    int n[VRD_MAX_SEGS], T[VRD_MAX_SEGS], sps[VRD_MAX_SEGS];
    int64_t row[VRD_MAX_SEGS], strip0[VRD_MAX_SEGS + 1];
};
// host: fill `t` for strips of rw rows over T[g] / t_div frames per sequence; returns the number of strips (or -1: bad table)
inline int64_t seg_table(SegTable& t, const vrd_row_segs* s, int t_div, int rw) {
    t.count = 0;
    if (!s || s->count < 1 || s->count > VRD_MAX_SEGS) return -1;
    int64_t at = 0;
    for (int g = 0; g < s->count; ++g) {
        if (s->n[g] <= 0 || s->T[g] <= 0 || s->T[g] % t_div || s->row[g] < 0) return -1;
        t.n[g] = s->n[g], t.T[g] = s->T[g], t.row[g] = s->row[g];
        t.sps[g] = (s->T[g] / t_div + rw - 1) / rw;
        t.strip0[g] = at;
        at += (int64_t)t.n[g] * t.sps[g];
    }
    t.count = s->count;
    t.strip0[s->count] = at;
    return at;
}
// device: strip ws -> (group g, sequence b of the group, strip of the sequence); false behind the last strip
__device__ __forceinline__ bool seg_find(const SegTable& t, int64_t ws, int& g, int& b, int& strip) {
    if (ws >= t.strip0[t.count]) return false;
    g = 0;
    while (g + 1 < t.count && ws >= t.strip0[g + 1]) ++g;
    const int64_t r = ws - t.strip0[g];
    b = (int)(r / t.sps[g]);
    strip = (int)(r - (int64_t)b * t.sps[g]);
    return true;
}

// A wave-uniform read of kernel INPUT data (written before the launch, never by it) through the scalar cache: s_load instead
// of a flat_load, which hipcc would follow with `s_waitcnt vmcnt(0)` -- a wait for every vector memory operation the wave has
// in flight (its LDS-DMA requests, its stores), not just for this value.
template <typename T>
__device__ __forceinline__ T uniform_load(const T* ptr) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(ptr));
}

// A pointer out of an argument STRUCT is a generic pointer to hipcc (only direct kernel arguments are inferred to be global
// memory), and it reads / writes through flat_ instructions -- which might address LDS, so next to LDS-DMA requests in flight
// every one of them is preceded by `s_waitcnt vmcnt(0)`.  as_global() states what the ABI guarantees (device global memory).
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T* as_global(T* ptr) {
    return reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(ptr));
}

typedef float f32x4_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float* ptr) {          // 16-byte global load
    const f32x4_native v = *as_global(reinterpret_cast<const f32x4_native*>(ptr));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gstore4(void* ptr, float4 v) {        // 16-byte global store
    const f32x4_native t = {v.x, v.y, v.z, v.w};
    *as_global(reinterpret_cast<f32x4_native*>(ptr)) = t;
}

// Device address of the current device's f16 operand-range flag word (vrd_f16_range_flag of the ABI), or nullptr
unsigned* range_flag();

// RAII profiling scope: when profiling is on, records a HIP event pair on `stream`
// around the launch(es) issued inside the scope.
struct ProfScope {
    int id;
    hipStream_t stream;
    void* slot;
    ProfScope(int kernel_id, hipStream_t s, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
};

#define VRD_CHECK_ARG(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            vrd::set_error(__VA_ARGS__);              \
            return -1;                                \
        }                                             \
    } while (0)

#define VRD_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            vrd::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                           hipGetErrorString(e_));                                \
            return -2;                                                            \
        }                                                                         \
    } while (0)

// Sum over the 64 lanes, returned wave-uniform.  Data-parallel-primitive adds instead of six ds_bpermute shuffles
// (each of those is an address computation, an LDS-crossbar instruction and a wait): quad swaps, half-row and row
// mirrors give every lane its 16-lane row sum, two row broadcasts carry the running sum into lane 63, one readlane
// returns it as a scalar.  The row kernels are instruction-bound; this is 7 instructions against ~20.
__device__ __forceinline__ float wave_sum(float v) {
    const int iv0 = __builtin_bit_cast(int, v);
    float t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv0, 0xB1, 0xF, 0xF, true));                              // quad_perm [1,0,3,2]
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));     // row_mirror
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));    // row_bcast:15 -> rows 1, 3
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));    // row_bcast:31 -> rows 2, 3
    v += t;
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over aligned groups of G = 8 or 16 lanes, every lane of a group receiving the group's sum: the first steps of wave_sum
// (three or four v_add_f32 with a DPP operand).  __shfl_xor compiles to ds_bpermute_b32 plus its address arithmetic, seven
// instructions and an LDS round trip per step; the banded attention spends 28 of those per query row on its dot products.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G == 8 || G == 16, "lanes per group");
    float t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += t;
    t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += t;
    if (G == 16) {
        t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
        v += t;
    }
    return v;
}

// "Pair" rows (GEMM-operand format of the split-precision modes): a row of W logical channels (W % 32 == 0) stored in the
// 4*W bytes an f32 row would occupy, as blocks of 32 channels: [32 x 16-bit hi | 32 x 16-bit lo] per block.  One 128-byte
// line therefore holds everything a K step of 32 needs from a row, which is what the GEMM's LDS-DMA wants (whole lines per
// request).  Producers whose output is consumed only as a GEMM / attention operand write this directly.  Rows concatenated
// from several producers need no bookkeeping: the block structure does not depend on where a slab starts (slab widths are
// multiples of 32).  Two element formats (enum vrd_pair_format of the ABI; every producer's `out_pair` / `c_pair` /
// `pair_wide` argument is one of them, 0 = plain f32 rows):
//   VRD_PAIR_BF16 (bf16x3 mode): hi = bf16(x), lo = bf16(x - hi): |x - hi - lo| <= 2^-17 |x|.
//   VRD_PAIR_F16 (f16x3 mode): y = x * 2^VRD_F16_ACT_EXP (exact), hi = f16(y), lo = f16(y - hi), round to nearest both:
//       |y - hi - lo| <= 2^-22 |y| while lo is a normal f16 (|y| >= 2^-2), 2^-25 absolute below (f16 subnormals are kept by
//       the conversions and by the MFMA: scripts/lab/r04/f16_denorm_probe.hip).  |y| must stay below 65504, i.e.
//       |x| < 4094: beyond it hi = inf, lo = -inf and every product that touches the element is NaN -- the result is
//       poisoned, not silently wrong (the host mirror checks its outputs and repeats the batch in the f32 mode).
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

constexpr int F16_ACT_EXP = VRD_F16_ACT_EXP;
constexpr float F16_ACT_SCALE = (float)(1 << VRD_F16_ACT_EXP), F16_ACT_INV = 1.0f / (float)(1 << VRD_F16_ACT_EXP);

// element types and the matrix instruction of a split format (F16 = false: bf16, true: scaled f16)
template <bool F16>
struct SplitFmt;
template <>
struct SplitFmt<false> {
    typedef __bf16 elem;
    typedef bf16x4_t x4;
    typedef bf16x8_t x8;
    static constexpr float act_scale = 1.0f, act_inv = 1.0f;
    static constexpr int fmt = VRD_PAIR_BF16;
};
template <>
struct SplitFmt<true> {
    typedef _Float16 elem;
    typedef f16x4_t x4;
    typedef f16x8_t x8;
    static constexpr float act_scale = F16_ACT_SCALE, act_inv = F16_ACT_INV;
    static constexpr int fmt = VRD_PAIR_F16;
};
typedef __attribute__((ext_vector_type(16))) float mfma_acc32_t;
typedef __attribute__((ext_vector_type(4))) float mfma_acc16_t;
__device__ __forceinline__ mfma_acc32_t mfma32(bf16x8_t a, bf16x8_t b, mfma_acc32_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ mfma_acc32_t mfma32(f16x8_t a, f16x8_t b, mfma_acc32_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ mfma_acc16_t mfma16(bf16x8_t a, bf16x8_t b, mfma_acc16_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ mfma_acc16_t mfma16(f16x8_t a, f16x8_t b, mfma_acc16_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Range tracking of the f16 format.  A value whose scaled magnitude reaches 65,520 becomes hi = inf, lo = -inf, and every
// product that touches it NaN -- but a NaN does not reach the outputs reliably (a ReLU or a max-pool downstream drops it:
// fmax(NaN, x) = x), so producers REPORT instead: each thread keeps the largest scaled magnitude it converted (one v_max3
// per two elements) and ORs the producer's tag into the device's flag word (vrd_f16_range_flag) if that does not fit.  The
// host mirror reads the word with a call's results and repeats the call in the f32 mode.
enum RangeTag : unsigned {
    RANGE_INPUT = 1u,        // boundary tensors: vrd_bct_to_btc, vrd_pack_pairs, vrd_gather_pairs / vrd_assemble_pairs
    RANGE_LAYERNORM = 2u,    // vrd_layernorm
    RANGE_DWCONV_LN = 4u,    // vrd_dwconv_ln
    RANGE_GEMM_OUT = 8u,     // pair-row outputs of vrd_gemm (c_pair)
    RANGE_GEMM_IN = 16u,     // f32 rows split inside a GEMM / attention kernel while they are staged
    RANGE_ATTN_OUT = 32u,    // pair-row outputs of the attention kernels
    RANGE_OTHER = 64u,
};
struct RangeTrack {
    float amax = 0.f;
    __device__ __forceinline__ void see(float y) { amax = fmaxf(amax, fabsf(y)); }
    __device__ __forceinline__ void see(float y0, float y1) { amax = fmaxf(amax, fmaxf(fabsf(y0), fabsf(y1))); }
    // (NaN inputs: fmax keeps the other operand -- a NaN here means an overflow upstream, which was reported there)
    __device__ __forceinline__ void report(unsigned* flag, unsigned tag) const {
        if (flag && amax >= 65520.f) __hip_atomic_fetch_or(as_global(flag), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// hi / lo planes of N values in format F16 (the f16 format scales by 2^VRD_F16_ACT_EXP first); rt: see RangeTrack
template <bool F16, int N, typename V>
__device__ __forceinline__ void split_n(const float (&x)[N], V& h, V& l, RangeTrack* rt = nullptr) {
    typedef typename SplitFmt<F16>::elem E;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float y = F16 ? x[j] * F16_ACT_SCALE : x[j];
        h[j] = (E)y;
        l[j] = (E)(y - (float)h[j]);
    }
    if (F16 && rt) {
        if (N == 1) rt->see(x[0] * F16_ACT_SCALE);
#pragma unroll
        for (int j = 0; j + 1 < N; j += 2) rt->see(x[j] * F16_ACT_SCALE, x[j + 1] * F16_ACT_SCALE);
    }
}

// the same with a run-time factor in place of 2^VRD_F16_ACT_EXP (F16 only; bf16 planes take the values as they are)
template <bool F16, int N, typename V>
__device__ __forceinline__ void split_n_scaled(const float (&x)[N], float mul, V& h, V& l, RangeTrack* rt = nullptr) {
    typedef typename SplitFmt<F16>::elem E;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float y = F16 ? x[j] * mul : x[j];
        h[j] = (E)y;
        l[j] = (E)(y - (float)h[j]);
    }
    if (F16 && rt) {
        if (N == 1) rt->see(x[0] * mul);
#pragma unroll
        for (int j = 0; j + 1 < N; j += 2) rt->see(x[j] * mul, x[j + 1] * mul);
    }
}

// 16-bit index of the hi half of channel c inside a pair row (the lo half is 32 further)
__device__ __forceinline__ int pair_index(int c) { return ((c >> 5) << 6) + (c & 31); }

// fmt: VRD_PAIR_BF16 or VRD_PAIR_F16 (wave-uniform)
__device__ __forceinline__ void store_pair4(float* row, int c, int /*W*/, float4 v, int fmt, RangeTrack* rt = nullptr) {      // c % 4 == 0
    const float x[4] = {v.x, v.y, v.z, v.w};
    char* r = reinterpret_cast<char*>(row) + pair_index(c) * 2;
    if (fmt == VRD_PAIR_F16) {
        f16x4_t h, l;
        split_n<true>(x, h, l, rt);
        *reinterpret_cast<f16x4_t*>(r) = h;
        *reinterpret_cast<f16x4_t*>(r + 64) = l;
    } else {
        bf16x4_t h, l;
        split_n<false>(x, h, l);
        *reinterpret_cast<bf16x4_t*>(r) = h;
        *reinterpret_cast<bf16x4_t*>(r + 64) = l;
    }
}

// eight consecutive channels c .. c+7 (c % 8 == 0): one 16-byte store per half
__device__ __forceinline__ void store_pair8(float* row, int c, float4 v0, float4 v1, int fmt, RangeTrack* rt = nullptr) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    char* r = reinterpret_cast<char*>(row) + pair_index(c) * 2;
    if (fmt == VRD_PAIR_F16) {
        f16x8_t h, l;
        split_n<true>(x, h, l, rt);
        *reinterpret_cast<f16x8_t*>(r) = h;
        *reinterpret_cast<f16x8_t*>(r + 64) = l;
    } else {
        bf16x8_t h, l;
        split_n<false>(x, h, l);
        *reinterpret_cast<bf16x8_t*>(r) = h;
        *reinterpret_cast<bf16x8_t*>(r + 64) = l;
    }
}

__device__ __forceinline__ void store_pair1(float* row, int c, int /*W*/, float x, int fmt, RangeTrack* rt = nullptr) {
    char* r = reinterpret_cast<char*>(row) + pair_index(c) * 2;
    if (fmt == VRD_PAIR_F16) {
        const float y = x * F16_ACT_SCALE;
        if (rt) rt->see(y);
        const _Float16 h = (_Float16)y;
        *reinterpret_cast<_Float16*>(r) = h;
        *reinterpret_cast<_Float16*>(r + 64) = (_Float16)(y - (float)h);
    } else {
        const __bf16 h = (__bf16)x;
        *reinterpret_cast<__bf16*>(r) = h;
        *reinterpret_cast<__bf16*>(r + 64) = (__bf16)(x - (float)h);
    }
}

// Branch-free, single-path erf (the library erff branches on |z| < 1 per lane, and both sides run under exec masks in
// any wave that straddles it):  erf(z) = sign(z) * (1 - 2^-G(min(|z|, 4))),  G = -log2(erfc) as a degree-11 polynomial.
// Near 0 the form keeps the ABSOLUTE error at one rounding of 1 (what GELU needs), not the relative one.
// Coefficients and error (max 9.8e-8 absolute; GELU 8.3e-8) from scripts/fit_erf.py.
#define VRD_ERF_G(FMA, u, T)                                   \
    T g = T(-1.133601083e-07f);                                \
    g = FMA(g, u, T(2.984484956e-06f));                        \
    g = FMA(g, u, T(-3.493388466e-05f));                       \
    g = FMA(g, u, T(2.373710733e-04f));                        \
    g = FMA(g, u, T(-1.004358838e-03f));                       \
    g = FMA(g, u, T(2.405994177e-03f));                        \
    g = FMA(g, u, T(-7.954030742e-05f));                       \
    g = FMA(g, u, T(-2.782256332e-02f));                       \
    g = FMA(g, u, T(1.483803001e-01f));                        \
    g = FMA(g, u, T(9.184222287e-01f));                        \
    g = FMA(g, u, T(1.627909324e+00f));                        \
    g = FMA(g, u, T(-3.292586530e-08f));

__device__ __forceinline__ float erf_f32(float z) {
    const float u = fminf(fabsf(z), 4.0f);
    VRD_ERF_G(fmaf, u, float)
    return copysignf(1.0f - __builtin_amdgcn_exp2f(-g), z);
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_f32(x * 0.70710678118654752440f));
}

// two elements at a time on packed f32 math (v_pk_fma_f32 / v_pk_mul_f32: half the instructions; the GEMM epilogue is
// instruction-bound and GELU is most of the mlp-up epilogue).  Same arithmetic per element as gelu_erf.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    const f32x2_t z = x * 0.70710678118654752440f;
    f32x2_t u;
    u.x = fminf(fabsf(z.x), 4.0f);
    u.y = fminf(fabsf(z.y), 4.0f);
    VRD_ERF_G(__builtin_elementwise_fma, u, f32x2_t)
    f32x2_t e;
    e.x = copysignf(1.0f - __builtin_amdgcn_exp2f(-g.x), z.x);
    e.y = copysignf(1.0f - __builtin_amdgcn_exp2f(-g.y), z.y);
    return (x * 0.5f) * (e + 1.0f);
}

}  // namespace vrd
