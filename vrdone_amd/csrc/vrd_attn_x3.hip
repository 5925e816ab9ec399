// Global masked attention (the SOS self / cross attention) in split precision: both contractions of flash
// attention as three bf16 MFMA products each (x = x_hi + x_lo, see vrd_gemm_x3.hip), f32 accumulate,
// f32 softmax.  Inputs q, k, v are pair rows ([hi | lo] bf16 planes) written by the projection GEMMs.
//
// One workgroup = NW waves = NW*32 query rows of one (b, head); KV tiles of 32 keys.
//   S^T = K . Q^T   A = K fragments from LDS (ds_read_b128), B = Q^T fragments held in registers for the whole
//                   kernel.  The result puts each query on a lane and 16 of its 32 keys in that lane's registers:
//                   softmax statistics are register-local plus one exchange with lane^32.
//   O^T += V^T . P^T  B = the probability registers themselves (registers 8s..8s+7 are the k-slots of k16 step s,
//                   in the instruction's permuted order key = 16s + 8(j>>2) + 4*half + (j&3)), split into hi/lo on
//                   the fly; A = V^T fragments fetched with ds_read_b64_tr_b16 (hardware transpose of a 4-key x
//                   16-column block), two per fragment, in exactly that key order.
// K/V tiles arrive by LDS-DMA (global_load_lds_dwordx4) into a 2-stage ring: tile t+1 is in flight while tile t
// is consumed; one barrier per tile.  LDS rows are unpadded (the DMA writes linearly), so 16-byte chunks are
// XOR-swizzled on the source address and on every read: K by (key & 15) for the row reads, V by (key & 3) << 2
// for the transposed reads; both patterns are conflict free for head_dim 128.
#include "vrd_common.h"
#include <cmath>
#include <cstdlib>

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ uint4 g_attn_zero[16];            // 256 zero bytes: source of keys past Tk

template <int HD>
struct AG {
    static constexpr int ROWB = HD * 2;                 // bytes per key row of one plane
    static constexpr int CPR = ROWB / 16;               // 16-byte chunks per row (16 or 8)
    static constexpr int RPI = 1024 / ROWB;             // key rows per wave DMA instruction (4 or 8)
    static constexpr int PLANE = 32 * ROWB;             // 8 KiB (hd 128)
    static constexpr int STAGE = 4 * PLANE;             // k_hi | k_lo | v_hi | v_lo
    static constexpr int N_DMA = 4 * 32 / RPI;          // wave instructions per tile (32 or 16)
    __device__ static constexpr int kswz(int key) { return key % CPR; }
    __device__ static constexpr int vswz(int key) { return ((key & 3) << 2) % CPR ^ (CPR == 8 ? ((key >> 1) & 1) << 2 : 0); }
};

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)x[j];
        lo[j] = (__bf16)(x[j] - (float)hi[j]);
    }
}

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_flash_x3_kernel(const float* __restrict__ q, int64_t ldq,
                                                                const float* __restrict__ k, const float* __restrict__ v,
                                                                int64_t ldkv, const uint8_t* __restrict__ kv_mask,
                                                                const uint8_t* __restrict__ q_mask, int Tq,
                                                                int Tk, int width, float scale, float* __restrict__ out,
                                                                int64_t ldo, int pair_out, int q_blocks, int n_head_) {
    using G = AG<HD>;
    constexpr int KS = HD / 16;                   // k16 steps of the S^T contraction
    constexpr int DT = HD / 32;                   // 32-row d tiles of O^T
    constexpr int PER_WAVE = (G::N_DMA + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    float* const kbias = reinterpret_cast<float*>(lds + 2 * G::STAGE);      // [32 * nkt]: 0 or -inf per key

    // 1-D grid with the XCD-aware renumbering of the GEMM kernels: the workgroups that share one (b, h)'s K and V
    // (consecutive logical ids) get the same XCD label, i.e. the same L2, and are dispatched close together.  With a
    // (q-block, h, b) grid they went round-robin to different XCDs and K/V were fetched from HBM once per q-block
    // (4.2 GB fetched per launch against 1.8 GB of q, k, v).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qq = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (qq + 1) : rem * (qq + 1) + (xcd - rem) * qq) + (bid >> 3);
    const int qblk = lid % q_blocks, h = (lid / q_blocks) % n_head_, b = lid / (q_blocks * n_head_);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q0 = (qblk * NW + wave) * 32;
    // pair rows are blocks of [32 hi | 32 lo] bf16 (vrd_common.h), so a head's HD channels are HD*4 contiguous
    // bytes; 16-byte chunk lc of the head's logical hi (lo) plane is at block lc/4, +64 bytes for lo
    const char* kb = reinterpret_cast<const char*>(k + (int64_t)b * Tk * ldkv) + h * HD * 4;
    const char* vb = reinterpret_cast<const char*>(v + (int64_t)b * Tk * ldkv) + h * HD * 4;
    const char* zero_src = reinterpret_cast<const char*>(g_attn_zero);

    // a wave whose 32 queries are all padding (q_mask) takes part in the staging and the barriers only; its rows are
    // written as zeros (the caller masks them)
    const bool q_live = __any(q0 + li < Tq && (!q_mask || q_mask[(int64_t)b * Tq + (q0 + li < Tq ? q0 + li : Tq - 1)] != 0));
    // a workgroup without a single live query (the padded tail of the sequence, or tiles past Tq) does not stream K / V
    if (!__syncthreads_or(q_live)) {
        const int tq = q0 + li;
        if (tq < Tq) {
            float* orow = out + ((int64_t)b * Tq + tq) * ldo + h * HD;
            for (int c = lh * 4; c < HD; c += 8) *reinterpret_cast<float4*>(orow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    // Q^T fragments: lane (query li, half lh) holds d = 16s + 8*lh + 0..7 of its query, hi and lo
    bf16x8 qh[KS], ql[KS];
    {
        const int tq = q0 + li;
        const char* qr = reinterpret_cast<const char*>(q + ((int64_t)b * Tq + (tq < Tq ? tq : Tq - 1)) * ldq) + h * HD * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int off = vrd::pair_index(16 * s + 8 * lh) * 2;
            qh[s] = *reinterpret_cast<const bf16x8*>(qr + off);
            ql[s] = *reinterpret_cast<const bf16x8*>(qr + off + 64);
        }
    }

    // DMA of tile kt: instruction j covers plane j / (32/RPI) rows (j % (32/RPI))*RPI ...; planes k_hi, k_lo, v_hi, v_lo
    const int rin = lane / G::CPR, pch = lane % G::CPR;
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int j = wave + NW * i;
            if (j < G::N_DMA) {
                const int plane = j / (32 / G::RPI), rb = j % (32 / G::RPI);
                const int row = rb * G::RPI + rin;                 // key inside the tile
                const int key = kt * 32 + row;
                const int sw = plane < 2 ? G::kswz(row) : G::vswz(row);
                const int lc = pch ^ sw;                           // logical 16-byte chunk of the plane row
                const char* base = (plane < 2 ? kb : vb) + (int64_t)key * ldkv * 4 + ((plane & 1) ? 64 : 0);
                // branch-free select (as a ternary on pointers hipcc emits an exec-masked branch per DMA)
                const uintptr_t pa = reinterpret_cast<uintptr_t>(base + (lc >> 2) * 128 + (lc & 3) * 16);
                const uintptr_t pz = reinterpret_cast<uintptr_t>(zero_src + lc * 16);
                const char* src = reinterpret_cast<const char*>(pz ^ ((pa ^ pz) & (uintptr_t)0 - (uintptr_t)(key < Tk)));
                __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(lds + buf * G::STAGE + plane * G::PLANE + rb * 1024), 16, 0, 0);
            }
        }
    };

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
    float m_run = -INFINITY, l_part = 0.f;

    // per-lane LDS offsets: K fragment row = key li; V^T transposed reads: lane 4q+p of a 16-lane group addresses
    // row r0 + q, columns c0 + 4p .. 4p+3, where c0 = 32*dt + 16*((lane >> 4) & 1) and r0 = 16s + 8*part + 4*lh
    const int krow = li * G::ROWB;
    const int vq = (lane >> 2) & 3, vp = lane & 3;
    const int vcol0 = 16 * ((lane >> 4) & 1) + 4 * vp;          // column inside a 32-wide d tile

    const int nkt = (Tk + 31) / 32;
    // key bias of the whole row of tiles, once: an ordinary global load inside the loop would make the compiler
    // drain the LDS-DMA queue (vmcnt(0)) at its first use, every tile
    // ... and, per tile of 32 keys, whether any key is valid: a tile of masked keys only adds exp(-inf) = 0 to every
    // sum and leaves the running maxima alone, so it is not loaded or multiplied at all (padding behind the pair's
    // frames: 32 of 288 keys at the benchmark shape, most of a max_seq_len batch of short pairs)
    int* const tile_on = reinterpret_cast<int*>(kbias + nkt * 32);
    for (int key = tid; key < nkt * 32; key += NW * 64) {
        const bool ok = key < Tk && (!kv_mask || kv_mask[(int64_t)b * Tk + key]);
        kbias[key] = ok ? 0.f : -INFINITY;
        const unsigned long long bal = __ballot(ok);
        if ((lane & 31) == 0) tile_on[key >> 5] = ((bal >> (lane & 32)) & 0xffffffffull) != 0ull;
    }
    __syncthreads();
    unsigned long long act = ~0ull;                      // rows of more than 64 tiles: every tile is visited
    if (nkt <= 64) act = __ballot(lane < nkt && tile_on[lane < nkt ? lane : 0] != 0);
    auto next_on = [&](int from) {                       // first tile >= from that has a valid key, or nkt
        if (from >= nkt) return nkt;
        if (nkt > 64) return from;
        const unsigned long long m = act & (~0ull << from);
        return m ? (int)__builtin_ctzll(m) : nkt;
    };
    int kt = next_on(0);
    if (kt < nkt) issue(kt, 0);
    for (int it = 0; kt < nkt; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                           // ... and everybody else's; the previous tile is fully consumed
        const int kt_next = next_on(kt + 1);
        if (kt_next < nkt) issue(kt_next, (it + 1) & 1);
        const char* st = lds + (it & 1) * G::STAGE;
        const float* kbs = kbias + kt * 32;
        if (!q_live) {
            kt = kt_next;
            continue;
        }

        // ---- S^T = K . Q^T (three products per k16 step)
        f32x16 sacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
        // the K fragments of step s+1 are requested before the three MFMAs of step s (one chain of dependent
        // MFMAs: nothing else would hide the LDS latency)
        bf16x8 kh = *reinterpret_cast<const bf16x8*>(st + krow + ((lh ^ G::kswz(li)) * 16));
        bf16x8 kl = *reinterpret_cast<const bf16x8*>(st + G::PLANE + krow + ((lh ^ G::kswz(li)) * 16));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 nh = kh, nl = kl;
            if (s + 1 < KS) {
                const int off = krow + (((2 * (s + 1) + lh) ^ G::kswz(li)) * 16);
                nh = *reinterpret_cast<const bf16x8*>(st + off);
                nl = *reinterpret_cast<const bf16x8*>(st + G::PLANE + off);
            }
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], sacc, 0, 0, 0);
            kh = nh;
            kl = nl;
        }

        // ---- online softmax for query column li; this lane holds keys (e&3) + 8*(e>>2) + 4*lh
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = sacc[e] * scale + kbs[(e & 3) + 8 * (e >> 2) + 4 * lh];
            mx = fmaxf(mx, sacc[e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sacc[e] = __expf(sacc[e] - m_use);
            psum += sacc[e];
        }
        l_part = l_part * alpha + psum;
        m_run = m_new;
        // rescale the running output only if some query's maximum moved (the kernel is instruction-bound: ~550
        // instructions per tile and wave, 16 * DT of them this rescale; after the first tiles the maxima rarely move)
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
        }

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pf[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = sacc[8 * s + j];
            bf16x8 ph, pl;
            split8(pf, ph, pl);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                bf16x8 vh, vl;
#ifdef VRD_ATTN_NO_TR
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int key = 16 * s + 8 * (jj >> 2) + 4 * lh + (jj & 3);
                    const int col = 32 * d + li;
                    const int off = key * G::ROWB + ((((col * 2) >> 4) ^ G::vswz(key)) * 16) + ((col * 2) & 15);
                    vh[jj] = *reinterpret_cast<const __bf16*>(st + 2 * G::PLANE + off);
                    vl[jj] = *reinterpret_cast<const __bf16*>(st + 3 * G::PLANE + off);
                }
#else
                s16x8 rh, rl;
#pragma unroll
                for (int part = 0; part < 2; ++part) {
                    const int row = 16 * s + 8 * part + 4 * lh + vq;                  // key row this lane addresses
                    const int col = 32 * d + vcol0;                                   // first of its 4 columns
                    const int off = row * G::ROWB + ((((col * 2) >> 4) ^ G::vswz(row)) * 16) + ((col * 2) & 15);
                    const s16x4 th = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 2 * G::PLANE + off));
                    const s16x4 tl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(st + 3 * G::PLANE + off));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        rh[4 * part + j] = th[j];
                        rl[4 * part + j] = tl[j];
                    }
                }
                vh = __builtin_bit_cast(bf16x8, rh);
                vl = __builtin_bit_cast(bf16x8, rl);
#endif
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, oacc[d], 0, 0, 0);
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, oacc[d], 0, 0, 0);
            }
        }
        kt = kt_next;
    }

    const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
    const float inv = q_live ? 1.0f / l_tot : 0.f;
    const int tq = q0 + li;
    if (tq < Tq) {
        float* orow = out + ((int64_t)b * Tq + tq) * ldo;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = h * HD + 32 * d + 8 * g + 4 * lh;           // registers 4g..4g+3 are d = 32d + 8g + 4lh + 0..3
                const float4 val = make_float4(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv,
                                               oacc[d][4 * g + 3] * inv);
                if (pair_out) vrd::store_pair4(orow, c, width, val);
                else *reinterpret_cast<float4*>(orow + c) = val;
            }
    }
}

template <int HD, int NW>
int launch(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask, const uint8_t* q_mask, int B, int Tq,
           int Tk, int n_head, float scale, float* out, int64_t ldo, int pair_out, hipStream_t s) {
    auto kern = attn_flash_x3_kernel<HD, NW>;
    constexpr size_t lds_max = 2 * AG<HD>::STAGE + (4096 + 128) * sizeof(float);       // key bias + tile flags for Tk <= 4096
    const size_t lds = 2 * AG<HD>::STAGE + (size_t)((Tk + 31) / 32) * 33 * sizeof(float);
    if (lds > lds_max) {
        vrd::set_error("vrd_attention_pair: Tk = %d exceeds the 4096 keys the key-bias row is sized for", Tk);
        return -1;
    }
    if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(kern), lds_max, "vrd_attention_pair")) return rc;
    const int tiles = (Tq + 31) / 32;
    const int q_blocks = (tiles + NW - 1) / NW;
    hipLaunchKernelGGL(kern, dim3((unsigned)q_blocks * n_head * B), dim3(NW * 64), lds, s, q, ldq, k, v, ldkv, kv_mask, q_mask, Tq, Tk,
                       n_head * HD, scale, out, ldo, pair_out, q_blocks, n_head);
    return 0;
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

}  // namespace

extern "C" int vrd_attention_pair(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                                  const uint8_t* kv_mask, const uint8_t* q_mask, int B, int Tq, int Tk, int n_head, int head_dim,
                                  float* out, int64_t ldo, int out_pair, void* stream) {
    VRD_CHECK_ARG(q && k && v && out, "vrd_attention_pair: null pointer");
    VRD_CHECK_ARG(head_dim == 64 || head_dim == 128, "vrd_attention_pair: head_dim must be 64 or 128 (got %d)", head_dim);
    VRD_CHECK_ARG(B > 0 && B <= 65535 && Tq > 0 && Tk > 0 && n_head > 0 && n_head <= 65535, "vrd_attention_pair: bad sizes");
    const int width = n_head * head_dim;
    VRD_CHECK_ARG(ldq >= width && ldkv >= width && ldo >= width && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 &&
                      aligned16(q) && aligned16(k) && aligned16(v) && aligned16(out),
                  "vrd_attention_pair: rows must be 16-byte aligned pair rows of width n_head*head_dim");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float scale = 1.0f / sqrtf((float)head_dim);
    vrd::ProfScope prof(VRD_K_ATTN_FLASH, s, 4.0 * B * (double)n_head * Tq * Tk * head_dim,
                        4.0 * B * (double)width * (2.0 * Tq + 2.0 * Tk));
    // waves (32-query tiles) per workgroup: every workgroup streams the whole K / V row of its (b, h), and the kernel is
    // bound by that stream (LDS-DMA issue), so fewer, fuller workgroups win even when the last one is mostly empty:
    // 288 queries = 9 tiles run as 4 + 4 + 1 (the lone tile is the padded tail at the benchmark shape and exits at
    // once) 17 % faster than as 3 + 3 + 3.  Two workgroups of 4 waves fill a CU's registers (247 VGPRs per wave); ONE
    // workgroup of 8 waves (a single stream per (b, h)) was measured 70 % slower: nothing runs while it waits at its
    // barrier for a tile.
    static const int nw_env = [] { const char* e = getenv("VRD_FLASH_NW"); return e ? atoi(e) : 0; }();
    const int nw = nw_env == 3 || nw_env == 4 ? nw_env : 4;
    int rc;
    if (head_dim == 128) rc = nw == 3 ? launch<128, 3>(q, ldq, k, v, ldkv, kv_mask, q_mask, B, Tq, Tk, n_head, scale, out, ldo, out_pair, s)
                                      : launch<128, 4>(q, ldq, k, v, ldkv, kv_mask, q_mask, B, Tq, Tk, n_head, scale, out, ldo, out_pair, s);
    else rc = nw == 3 ? launch<64, 3>(q, ldq, k, v, ldkv, kv_mask, q_mask, B, Tq, Tk, n_head, scale, out, ldo, out_pair, s)
                      : launch<64, 4>(q, ldq, k, v, ldkv, kv_mask, q_mask, B, Tq, Tk, n_head, scale, out, ldo, out_pair, s);
    if (rc) return rc;
    VRD_LAUNCH_CHECK();
    return 0;
}
