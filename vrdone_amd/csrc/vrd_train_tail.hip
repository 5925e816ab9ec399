// Training tail on the device: the Hungarian assignment of the matcher and the EMA weight update.
//
//  vrd_assign       models/maskvrd.py:484-492 hands the (B*Q, sum N) cost matrix to the host and runs scipy's
//                   linear_sum_assignment once per pair.  Here every pair's (N_i relations x Q queries) block is solved by
//                   one thread with the O(n^2 m) shortest-augmenting-path form of the Hungarian algorithm (potentials
//                   u, v; N_i <= Q <= 16), in double precision; only the assignment (one int per relation) leaves the
//                   device.  The optimum is the one scipy finds whenever it is unique.
//  vrd_ema_update   utils/train_utils.py:21-29 walks the ~520 state-dict tensors with three elementwise launches
//                   each; here ONE launch updates all of them through a chunk table:
//                   ema = decay * ema + (1 - decay) * model, each product and the sum rounded to f32 like the
//                   reference's tensor expression (no FMA contraction), so the result is bit-identical.
#include "vrd_common.h"

namespace {

constexpr int AS_MAX = 16;

__global__ __launch_bounds__(64) void assign_kernel(const float* __restrict__ cost, int64_t ld, const int32_t* __restrict__ first,
                                                    const int32_t* __restrict__ count, int P, int Q, int32_t* __restrict__ query_of) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    const int n = count[p], r0 = first[p];
    if (n <= 0) return;
    if (n > Q) {                  // more relations than queries: no complete assignment exists (the augmenting search would
        for (int i = 0; i < n; ++i) query_of[r0 + i] = -1;      // never end); the host routes such pairs to scipy
        return;
    }
    // scipy's linear_sum_assignment refuses a matrix with a NaN or a -inf anywhere ("matrix contains invalid numeric entries",
    // which the reference lets propagate: models/maskvrd.py:492): such a pair comes back unassigned, whichever entries the
    // search itself would have visited
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < Q; ++j) {
            const float c = cost[(int64_t)(r0 + i) * ld + j];
            if (!(c == c) || c == -INFINITY) {
                for (int r = 0; r < n; ++r) query_of[r0 + r] = -1;
                return;
            }
        }
    // rows = relations 1..n, columns = queries 1..Q (1-based like the textbook form); way / minv per column
    double u[AS_MAX + 1], v[AS_MAX + 1], minv[AS_MAX + 1];
    int match[AS_MAX + 1], way[AS_MAX + 1];
    bool used[AS_MAX + 1];
    for (int j = 0; j <= Q; ++j) v[j] = 0.0, match[j] = 0;
    for (int i = 0; i <= n; ++i) u[i] = 0.0;
    for (int i = 1; i <= n; ++i) {
        match[0] = i;
        int j0 = 0;
        for (int j = 0; j <= Q; ++j) minv[j] = 1e300, used[j] = false;
        do {
            used[j0] = true;
            const int i0 = match[j0];
            double delta = 1e300;
            int j1 = 0;
            for (int j = 1; j <= Q; ++j) {
                if (used[j]) continue;
                const double cur = (double)cost[(int64_t)(r0 + i0 - 1) * ld + (j - 1)] - u[i0] - v[j];
                if (cur < minv[j]) minv[j] = cur, way[j] = j0;
                if (minv[j] < delta) delta = minv[j], j1 = j;
            }
            // No usable column: every remaining cost of the row is NaN or infinite (a diverged step).  `cur < minv[j]` is then
            // never true, j1 stays 0 and -- column 0 being the row under assignment -- the search would spin for ever.  The
            // pair's relations are marked unassigned (-1) instead; the host raises like scipy's linear_sum_assignment does on
            // such a matrix (ops.assign / MaskVRD.bipartite_match).
            if (j1 == 0 || !(delta < 1e300)) {
                for (int r = 0; r < n; ++r) query_of[r0 + r] = -1;
                return;
            }
            for (int j = 0; j <= Q; ++j) {
                if (used[j]) u[match[j]] += delta, v[j] -= delta;
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (match[j0] != 0);
        do {
            const int j1 = way[j0];
            match[j0] = match[j1];
            j0 = j1;
        } while (j0);
    }
    for (int j = 1; j <= Q; ++j)
        if (match[j]) query_of[r0 + match[j] - 1] = j - 1;
}

constexpr int EMA_CHUNK = 4096;
__global__ __launch_bounds__(256) void ema_kernel(float* const* __restrict__ ema, const float* const* __restrict__ model,
                                                  const int64_t* __restrict__ numel, const int32_t* __restrict__ chunk_tensor,
                                                  const int32_t* __restrict__ chunk_index, float decay, float one_minus) {
    const int t = chunk_tensor[blockIdx.x];
    const int64_t base = (int64_t)chunk_index[blockIdx.x] * EMA_CHUNK;
    const int64_t n = numel[t];
    float* const e = ema[t];
    const float* const m = model[t];
    for (int64_t i = base + threadIdx.x; i < base + EMA_CHUNK && i < n; i += 256)
        e[i] = __fadd_rn(__fmul_rn(decay, e[i]), __fmul_rn(one_minus, m[i]));
}

}  // namespace

extern "C" {

int vrd_assign(const float* cost, int64_t ld, const int32_t* first, const int32_t* count, int P, int Q, int32_t* query_of, void* stream) {
    VRD_CHECK_ARG(cost && first && count && query_of, "vrd_assign: null pointer");
    VRD_CHECK_ARG(P > 0 && Q >= 1 && Q <= AS_MAX && ld >= Q, "vrd_assign: Q must be 1..%d (got %d)", AS_MAX, Q);
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 0.0);
    hipLaunchKernelGGL(assign_kernel, dim3((P + 63) / 64), dim3(64), 0, s, cost, ld, first, count, P, Q, query_of);
    VRD_LAUNCH_CHECK();
    return 0;
}

int vrd_ema_update(float* const* ema, const float* const* model, const int64_t* numel, const int32_t* chunk_tensor,
                   const int32_t* chunk_index, int n_chunks, float decay, float one_minus_decay, void* stream) {
    VRD_CHECK_ARG(ema && model && numel && chunk_tensor && chunk_index && n_chunks > 0, "vrd_ema_update: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    vrd::ProfScope prof(VRD_K_BACKWARD, s, 0.0, 12.0 * (double)n_chunks * EMA_CHUNK);
    hipLaunchKernelGGL(ema_kernel, dim3(n_chunks), dim3(256), 0, s, ema, model, numel, chunk_tensor, chunk_index, decay, one_minus_decay);
    VRD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// f32 weight (any 3-stride view) -> blocked [hi | lo] 16-bit operand of the split-precision GEMMs; thread = one element
// VRD_PAIR_F16: the weight is multiplied by a per-tensor power of two first (vrd_split_weight in the header): a pass for
// max |w| (atomic maximum of the f32 bit patterns: non-negative floats order like their bits), then the split, which reads
// the exponent from it.  scale[2] holds the maximum's bits between the two.
// ------------------------------------------------------------------------------------------------------------------
namespace {
// e_w from the bits of max |w|: max * 2^e_w in [2^14, 2^15); clamped so that 2^e_w and 2^-(e_w + VRD_F16_ACT_EXP) are normal floats
__device__ __forceinline__ int weight_exp(unsigned max_bits) {
    const int E = (int)((max_bits >> 23) & 0xffu);          // max in [2^(E-127), 2^(E-126))
    if (E == 0 || E == 255) return 0;                      // zero / subnormal / non-finite maximum: leave the weight alone
    const int e = 141 - E;
    return e > 100 ? 100 : e;
}
__device__ __forceinline__ float pow2i(int e) { return __builtin_bit_cast(float, (unsigned)(e + 127) << 23); }

template <bool F16>
__device__ __forceinline__ void put_split(void* out, int64_t at, float w, float wmul) {
    typedef typename vrd::SplitFmt<F16>::elem E;
    E* o = reinterpret_cast<E*>(out) + at;
    const float y = F16 ? w * wmul : w;
    const E hi = (E)y;
    o[0] = hi;
    o[32] = (E)(y - (float)hi);
}

__global__ __launch_bounds__(256) void split_clear_kernel(const vrd_split_job* __restrict__ jobs, float* one, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float* sc = jobs ? jobs[i].scale : one;
    if (sc) reinterpret_cast<unsigned*>(sc)[2] = 0u;
}

__global__ __launch_bounds__(256) void weight_absmax_kernel(const float* __restrict__ src, int R, int Q, int taps, int64_t sr, int64_t st,
                                                            int64_t sq, float* __restrict__ scale) {
    const int K = taps * Q;
    const int64_t n = (int64_t)R * K;
    unsigned m = 0u;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        const int r = (int)(idx / K), c = (int)(idx - (int64_t)r * K);
        const int tap = c / Q, q = c - tap * Q;
        const unsigned b = __builtin_bit_cast(unsigned, src[r * sr + tap * st + q * sq]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned*>(scale) + 2, m);
}

template <bool F16>
__global__ __launch_bounds__(256) void split_weight_kernel(const float* __restrict__ src, int R, int Q, int taps, int64_t sr, int64_t st,
                                                           int64_t sq, void* __restrict__ out, float* __restrict__ scale) {
    const int K = taps * Q;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float wmul = 1.f;
    if (F16) {
        const int ew = weight_exp(reinterpret_cast<const unsigned*>(scale)[2]);
        wmul = pow2i(ew);
        if (idx == 0) {
            scale[0] = pow2i(-(ew + vrd::F16_ACT_EXP));
            scale[1] = wmul;
        }
    }
    if (idx >= (int64_t)R * K) return;
    const int r = (int)(idx / K), c = (int)(idx - (int64_t)r * K);
    const int tap = c / Q, q = c - tap * Q;
    put_split<F16>(out, ((int64_t)r * K + (c >> 5) * 32) * 2 + (c & 31), src[r * sr + tap * st + q * sq], wmul);   // block (r, c / 32): [32 hi | 32 lo]
}
}  // namespace

// The same for MANY weights in one launch (a training step re-splits every conv weight -- forward form and input-gradient form
// -- after each optimiser update: 242 launches of ~4 us on the 24-pair batch).  Chunk c of the launch is one 32 x 32 tile of job
// chunk_job[c]: rows 32 * (chunk_index[c] / (K / 32)) .., K block chunk_index[c] % (K / 32).  The source is read along
// whichever of its two indices is closer to contiguous (the input-gradient form is a transpose: consecutive output columns
// are Cin * k floats apart in the parameter, consecutive rows k apart) and turned in LDS, so reads and writes both coalesce
// (thread-per-element reads of the transposed form ran at 0.9 TB/s: 0.43 ms per step).
namespace {
// MODE 0: split (either format, per job); MODE 1: max |w| of the VRD_PAIR_F16 jobs' tiles
template <int MODE>
__global__ __launch_bounds__(256) void split_weights_kernel(const vrd_split_job* __restrict__ jobs, const int32_t* __restrict__ chunk_job,
                                                            const int32_t* __restrict__ chunk_index) {
    __shared__ float tile[32][33];
    const vrd_split_job j = jobs[chunk_job[blockIdx.x]];
    const bool f16 = j.fmt == VRD_PAIR_F16;            // block-uniform
    if (MODE == 1 && !f16) return;
    const int K = j.taps * j.Q, kb = K >> 5;
    const int r0 = (chunk_index[blockIdx.x] / kb) * 32, c0 = (chunk_index[blockIdx.x] % kb) * 32;
    const int64_t asr = j.sr < 0 ? -j.sr : j.sr, asq = j.sq < 0 ? -j.sq : j.sq;
    const bool rows_fastest = asr < asq;                 // block-uniform
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int rr = rows_fastest ? (e & 31) : (e >> 5), cc = rows_fastest ? (e >> 5) : (e & 31);
        const int r = r0 + rr, c = c0 + cc;
        const int tap = c / j.Q, q = c - tap * j.Q;
        const float w = r < j.R ? j.src[r * j.sr + tap * j.st + q * j.sq] : 0.f;
        if (MODE == 1) {
            const unsigned b = __builtin_bit_cast(unsigned, w) & 0x7fffffffu;
            m = b > m ? b : m;
        } else {
            tile[rr][cc] = w;
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
            m = t > m ? t : m;
        }
        if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned*>(j.scale) + 2, m);
        return;
    }
    __syncthreads();
    float wmul = 1.f;
    if (f16) {
        const int ew = weight_exp(reinterpret_cast<const unsigned*>(j.scale)[2]);
        wmul = pow2i(ew);
        if (chunk_index[blockIdx.x] == 0 && threadIdx.x == 0) {
            j.scale[0] = pow2i(-(ew + vrd::F16_ACT_EXP));
            j.scale[1] = wmul;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int rr = e >> 5, cc = e & 31;
        if (r0 + rr >= j.R) continue;
        const int64_t at = ((int64_t)(r0 + rr) * K + c0) * 2 + cc;        // block (r, c0 / 32): [32 hi | 32 lo]
        if (f16) put_split<true>(j.out, at, tile[rr][cc], wmul);
        else put_split<false>(j.out, at, tile[rr][cc], 1.f);
    }
}
}  // namespace

extern "C" int vrd_split_weights(const vrd_split_job* jobs, int n_jobs, const int32_t* chunk_job, const int32_t* chunk_index, int n_chunks,
                                 void* stream) {
    VRD_CHECK_ARG(jobs && n_jobs > 0 && chunk_job && chunk_index && n_chunks > 0, "vrd_split_weights: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (jobs is a device table: whether any job is VRD_PAIR_F16 is not known here; the two extra launches return at once otherwise)
    hipLaunchKernelGGL(split_clear_kernel, dim3((n_jobs + 255) / 256), dim3(256), 0, s, jobs, (float*)nullptr, n_jobs);
    hipLaunchKernelGGL(split_weights_kernel<1>, dim3(n_chunks), dim3(256), 0, s, jobs, chunk_job, chunk_index);
    hipLaunchKernelGGL(split_weights_kernel<0>, dim3(n_chunks), dim3(256), 0, s, jobs, chunk_job, chunk_index);
    VRD_LAUNCH_CHECK();
    return 0;
}

namespace {
// max |x| over a matrix -> {2^e, 2^-e}, e = 140 - (biased exponent of the maximum), in one launch.  The tensors are read once
// (a gradient that its two consumers read right after), so the launch is only as good as its loads in flight and its
// serial tail: at most one 1,024-thread workgroup per CU, four independent float4 loads per lane and trip (16 MB in flight on
// the chip), every workgroup's maximum as a plain store into scale[4 + block] and ONE atomic per workgroup (a ticket in
// scale[3]); the last one in reduces the partial maxima and writes the factors.  (History: one atomicMax per wave of a 2,048
// workgroup grid took 109 us on a 50 MB tensor that streams in 12 -- same-address atomics are worked off at 12-50 ns each;
// one atomicMax + one ticket per workgroup of a 512-workgroup grid, one load in flight per lane: 13-35 us per call, ~3 ms of a
// vidor-size training step.)
constexpr int ABSMAX_THREADS = 1024, ABSMAX_UNROLL = 4;

__device__ __forceinline__ unsigned absmax4(float4 v, unsigned m) {
    const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
    const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
    return q > m ? q : m;
}

template <bool FLAT>
__global__ __launch_bounds__(ABSMAX_THREADS) void absmax_scale_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols4,
                                                                      float* __restrict__ scale) {
    unsigned m = 0u;
    const int64_t n = rows * cols4;                                       // float4 pieces
    const int64_t stride = (int64_t)gridDim.x * ABSMAX_THREADS;
    const int64_t first = (int64_t)blockIdx.x * ABSMAX_THREADS + threadIdx.x;
    if (FLAT) {                                                           // ldx == cols: one run of float4
        const float4* const x4 = reinterpret_cast<const float4*>(x);
        int64_t idx = first;
        for (; idx + (ABSMAX_UNROLL - 1) * stride < n; idx += ABSMAX_UNROLL * stride) {
            float4 v[ABSMAX_UNROLL];
#pragma unroll
            for (int u = 0; u < ABSMAX_UNROLL; ++u) v[u] = x4[idx + u * stride];
#pragma unroll
            for (int u = 0; u < ABSMAX_UNROLL; ++u) m = absmax4(v[u], m);
        }
        for (; idx < n; idx += stride) m = absmax4(x4[idx], m);
    } else {                                                              // padded rows: (row, piece) stepped without a division per trip
        int64_t r = first / cols4;
        int c = (int)(first - r * cols4);
        const int64_t dr = stride / cols4;
        const int dc = (int)(stride - dr * cols4);
        for (int64_t idx = first; idx < n; idx += stride) {
            m = absmax4(*reinterpret_cast<const float4*>(x + r * ldx + 4 * c), m);
            r += dr;
            c += dc;
            if (c >= cols4) { c -= cols4; ++r; }
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
        m = t > m ? t : m;
    }
    unsigned* const words = reinterpret_cast<unsigned*>(scale);
    __shared__ unsigned wave_max[ABSMAX_THREADS / 64];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned bm = wave_max[0];
#pragma unroll
        for (int w = 1; w < ABSMAX_THREADS / 64; ++w) bm = wave_max[w] > bm ? wave_max[w] : bm;
        __hip_atomic_store(words + 4 + blockIdx.x, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(words + 3, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    unsigned t = threadIdx.x < gridDim.x ? __hip_atomic_load(words + 4 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const unsigned u = (unsigned)__shfl_xor((int)t, o, 64);
        t = u > t ? u : t;
    }
    __syncthreads();                                  // (wave_max is read above by thread 0 only, before the first barrier's twin)
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mx = wave_max[0];
#pragma unroll
        for (int w = 1; w < ABSMAX_THREADS / 64; ++w) mx = wave_max[w] > mx ? wave_max[w] : mx;
        const int E = (int)((mx >> 23) & 0xffu);
        int e = (E == 0 || E == 255) ? 0 : 140 - E;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        scale[0] = pow2i(e);
        scale[1] = pow2i(-e);
        __hip_atomic_store(words + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the ticket counter, for the next launch
    }
}
}  // namespace

extern "C" int vrd_absmax_scale(const float* x, int64_t ldx, int64_t rows, int cols, float* scale, void* stream) {
    VRD_CHECK_ARG(x && scale && rows > 0 && cols > 0 && ldx >= cols, "vrd_absmax_scale: bad arguments");
    VRD_CHECK_ARG(cols % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0,
                  "vrd_absmax_scale: rows must be float4-aligned (cols %d, ldx %lld)", cols, (long long)ldx);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n = rows * (cols / 4);
    const int64_t per_block = (int64_t)ABSMAX_THREADS * ABSMAX_UNROLL;
    const int64_t want = (n + per_block - 1) / per_block;
    int64_t cap = vrd::device_cu_count();                                  // one workgroup per CU, and never more than the partial slots
    if (cap > VRD_ABSMAX_SCALE_FLOATS - 4) cap = VRD_ABSMAX_SCALE_FLOATS - 4;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (ldx == cols) hipLaunchKernelGGL(absmax_scale_kernel<true>, dim3(blocks), dim3(ABSMAX_THREADS), 0, s, x, ldx, rows, cols / 4, scale);
    else hipLaunchKernelGGL(absmax_scale_kernel<false>, dim3(blocks), dim3(ABSMAX_THREADS), 0, s, x, ldx, rows, cols / 4, scale);
    VRD_LAUNCH_CHECK();
    return 0;
}

extern "C" int vrd_split_weight(const float* src, int R, int Q, int taps, int64_t sr, int64_t st, int64_t sq, uint16_t* out, int fmt,
                                float* scale, void* stream) {
    VRD_CHECK_ARG(src && out, "vrd_split_weight: null pointer");
    VRD_CHECK_ARG(R > 0 && Q > 0 && taps > 0 && ((int64_t)taps * Q) % 32 == 0, "vrd_split_weight: K = taps * Q must be a positive multiple of 32");
    VRD_CHECK_ARG(fmt == VRD_PAIR_BF16 || (fmt == VRD_PAIR_F16 && scale), "vrd_split_weight: fmt must be VRD_PAIR_BF16 or VRD_PAIR_F16 (with scale)");
    const int64_t n = (int64_t)R * taps * Q;
    VRD_CHECK_ARG((n + 255) / 256 < ((int64_t)1 << 31), "vrd_split_weight: too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (fmt == VRD_PAIR_F16) {
        hipLaunchKernelGGL(split_clear_kernel, dim3(1), dim3(256), 0, s, (const vrd_split_job*)nullptr, scale, 1);
        hipLaunchKernelGGL(weight_absmax_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, src, R, Q, taps, sr, st, sq, scale);
        hipLaunchKernelGGL(split_weight_kernel<true>, dim3(blocks), dim3(256), 0, s, src, R, Q, taps, sr, st, sq, (void*)out, scale);
    } else {
        hipLaunchKernelGGL(split_weight_kernel<false>, dim3(blocks), dim3(256), 0, s, src, R, Q, taps, sr, st, sq, (void*)out, (float*)nullptr);
    }
    VRD_LAUNCH_CHECK();
    return 0;
}
