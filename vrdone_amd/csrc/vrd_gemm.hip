// Dense 1-D convolution (k = 1 or 3) as an implicit GEMM on the gfx950 f32 MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate).
//
//   C[r, n] = epilogue( sum_k A'[r, k] * W[n, k] ),  A'[r, tap*Cin + ci] = A[r + tap - taps/2, ci]
//
// rows r = b*T + t are channels-last activations; the tap shift stays inside one length-T
// sequence (zero padding), which is the reference's Conv1d(padding=k//2) per sample.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup, K step 16, four waves in a 2 x 2
// grid, each owning a 64 x 64 sub-tile = 2 x 2 MFMA accumulators of 32 x 32 (64 VGPRs).
// Operand tiles are staged global -> registers -> LDS (double buffered, one barrier per K
// step) in k-major order [k][row] so every ds_read_b32 of a fragment is conflict free:
// lane l reads row (l & 31) of k = 2*step + (l >> 5), which is exactly the A/B operand
// map of the 32x32x2 instruction.  The f32 MFMA issues once per 64 cycles per SIMD, so
// four LDS reads per four MFMAs leave the LDS idle; the kernel is bound by the MFMA pipe.
//
// Workgroups are renumbered so that the ones dealt to one XCD (blockIdx % 8) walk
// consecutive output tiles: the N-tiles of one 128-row activation panel then share that
// XCD's L2 instead of each XCD fetching the panel from HBM.
#include "vrd_common.h"
#include "vrd_gemm_epilogue.h"

namespace {

using vrd::f32x16;

constexpr int BM = 128, BN = 128;
constexpr int LDM = 132;   // LDS row pitch (floats): 128 + 4 keeps rows 16-B aligned; 2-way write conflicts are free

constexpr size_t lds_bytes(int bk) { return (size_t)4 * bk * LDM * sizeof(float); }   // a[2][bk][LDM] + b[2][bk][LDM]

template <bool VEC, int TAPS>
__device__ __forceinline__ void load_a(const vrd_gemm_args& p, int64_t r, int t_in_seq, int k, int K, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r >= p.M) return;
    if (VEC) {
        if (k >= K) return;
        int tap = 0, ci = k;
        if (TAPS == 3) {
            tap = (k >= p.Cin) + (k >= 2 * p.Cin);
            ci = k - tap * p.Cin;
            int tt = t_in_seq + tap - 1;
            if (tt < 0 || tt >= p.T) return;
        }
        const float4 x = *reinterpret_cast<const float4*>(p.A + (r + tap - (TAPS == 3 ? 1 : 0)) * p.lda + ci);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int kk = k + j;
            if (kk >= K) continue;
            int tap = 0, ci = kk;
            if (TAPS == 3) {
                tap = kk / p.Cin;
                ci = kk - tap * p.Cin;
                int tt = t_in_seq + tap - 1;
                if (tt < 0 || tt >= p.T) continue;
            }
            v[j] = p.A[(r + tap - (TAPS == 3 ? 1 : 0)) * p.lda + ci];
        }
    }
}

template <bool VEC>
__device__ __forceinline__ void load_w(const vrd_gemm_args& p, int n, int k, int K, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (n >= p.N) return;
    if (VEC) {
        if (k >= K) return;
        const float4 x = *reinterpret_cast<const float4*>(p.W + (int64_t)n * K + k);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < K) v[j] = p.W[(int64_t)n * K + k + j];
    }
}

// workgroups per CU the register budget is set for (3 x 168 registers; at the compiler's own choice of 208 two fit and the
// MFMA pipes idle whenever both are outside their K loops: 0.71 -> 0.75 of the f32 MFMA peak with three)
#ifndef VRD_F32_WAVES
#define VRD_F32_WAVES 3
#endif
// sum over the tiles that skipped their contraction (padding map) of K * tile columns: 2 * 128 * this = FLOPs that were launched
// but not executed (the profile keeps executed and launched work apart, as for the 256 x 256 split kernel)
__device__ unsigned long long g_f32_skipped_kn;

template <bool VEC, int TAPS, int BK, bool STAGED>
__global__ __launch_bounds__(256, VRD_F32_WAVES) void gemm_f32_mfma_kernel(vrd_gemm_args p, int tiles_m, int tiles_n, unsigned* rflag) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*lds_a)[BK][LDM] = reinterpret_cast<float (*)[BK][LDM]>(smem);
    float (*lds_b)[BK][LDM] = reinterpret_cast<float (*)[BK][LDM]>(smem + 2 * BK * LDM);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int K = p.Cin * TAPS;
    const int nkt = (K + BK - 1) / BK;
    // XCD-aware (bijective) renumbering of the workgroup id.  (Persistent workgroups -- a grid of CUs x 3 walking the tiles --
    // were measured: 414 against 397 ms per step in the f32 mode; not kept.)
    const int nwg = tiles_m * tiles_n;
    const int vb = blockIdx.x;
    const int xcd = vb & 7, q = nwg >> 3, rem = nwg & 7;
    const int lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (vb >> 3);
    const int tm = lid / tiles_n, tn = lid - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    // Padding map (vrd_gemm_args.row_blocks, the 256 x 256 split kernel's: vrd_gemm_x3_big.hip): the tile's rows are four 32-row
    // blocks, slots tm*4 .. tm*4+3 of the block list; inside a segment of the list the blocks holding valid frames come first, so
    // a tile is a contraction tile or, behind those, a tile of fully padded blocks that only runs the epilogue on a zero
    // accumulator (the reference's value wherever row_mask zeroes the row).
    const int32_t* blist = p.row_blocks;
    bool contract = true;
    int64_t brow[4];                                    // first row of the tile's four blocks (p.M: a block outside the matrix)
#pragma unroll
    for (int j = 0; j < 4; ++j) brow[j] = m0 + 32 * j;
    if (blist) {
        const int nblk = (int)(p.M >> 5), seg_len = p.row_block_seg_len;
#pragma unroll
        for (int j = 0; j < 4; ++j) brow[j] = tm * 4 + j < nblk ? (int64_t)vrd::uniform_load(blist + tm * 4 + j) * 32 : p.M;
        const int seg = (tm * 4) / seg_len;
        contract = tm * 4 < nblk && tm * 4 - seg * seg_len < vrd::uniform_load(p.row_blocks_active + seg);
        if (!contract && tid == 0 && tm * 4 < nblk)
            atomicAdd(&g_f32_skipped_kn, (unsigned long long)(p.Cin * TAPS) * (unsigned)(p.N - n0 < BN ? p.N - n0 : BN));
    }

    // staging assignment: NP (row, 4-wide k chunk) pieces of each operand per thread
    constexpr int NP = BK / 8, KQ = BK / 4;
    int srow[NP], skq[NP], st[NP];
    int64_t arow[NP];                                   // the piece's row of A
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int f = tid + 256 * i;
        srow[i] = f / KQ;
        skq[i] = (f % KQ) * 4;
        const int b4 = srow[i] >> 5;
        arow[i] = (b4 == 0 ? brow[0] : b4 == 1 ? brow[1] : b4 == 2 ? brow[2] : brow[3]) + (srow[i] & 31);
        st[i] = (TAPS == 3 && arow[i] < p.M) ? (int)(arow[i] % p.T) : 0;
    }

    float ra[NP][4], rb[NP][4];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            load_a<VEC, TAPS>(p, arow[i], st[i], kt * BK + skq[i], K, ra[i]);
            load_w<VEC>(p, n0 + srow[i], kt * BK + skq[i], K, rb[i]);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lds_a[buf][skq[i] + j][srow[i]] = ra[i][j];
                lds_b[buf][skq[i] + j][srow[i]] = rb[i][j];
            }
    };

    if (contract) {
    fetch(0);
    stage(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kt + 1);
        // fragments of step s+1 are read from LDS while the four MFMAs of step s execute
        float fa[2][2], fb[2][2];
        fa[0][0] = lds_a[cur][lh][wm * 64 + li];
        fa[0][1] = lds_a[cur][lh][wm * 64 + 32 + li];
        fb[0][0] = lds_b[cur][lh][wn * 64 + li];
        fb[0][1] = lds_b[cur][lh][wn * 64 + 32 + li];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int c = s & 1, nx = c ^ 1;
            if (s + 1 < BK / 2) {
                fa[nx][0] = lds_a[cur][2 * s + 2 + lh][wm * 64 + li];
                fa[nx][1] = lds_a[cur][2 * s + 2 + lh][wm * 64 + 32 + li];
                fb[nx][0] = lds_b[cur][2 * s + 2 + lh][wn * 64 + li];
                fb[nx][1] = lds_b[cur][2 * s + 2 + lh][wn * 64 + 32 + li];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][0], fb[c][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][1], fb[c][1], acc[1][1], 0, 0, 0);
            // pin the interleave: the two paired LDS reads of the next step, then this step's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (kt + 1 < nkt) stage(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    }       // contract

    vrd::gemm_epilogue<STAGED>(p, acc, smem, wm ? brow[2] : brow[0], n0 + wn * 64, wave, lane, rflag, wm ? brow[3] : brow[1]);
}

inline bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0; }

template <bool VEC, int TAPS, int BK, bool STAGED>
int launch_variant(const vrd_gemm_args& a, int tiles_m, int tiles_n, hipStream_t s) {
    auto kern = gemm_f32_mfma_kernel<VEC, TAPS, BK, STAGED>;
    constexpr size_t lds = lds_bytes(BK);
    if (lds > 48 * 1024)        // opt in to a large dynamic LDS carve (per variant and device)
        if (int rc = vrd::reserve_lds(reinterpret_cast<const void*>(kern), lds, "vrd_gemm")) return rc;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m, tiles_n, a.c_pair == VRD_PAIR_F16 ? vrd::range_flag() : nullptr);
    return 0;
}

template <int BK, bool STAGED>
int launch_bk(const vrd_gemm_args& a, bool vec, int tiles_m, int tiles_n, hipStream_t s) {
    if (a.taps == 1) return vec ? launch_variant<true, 1, BK, STAGED>(a, tiles_m, tiles_n, s)
                                : launch_variant<false, 1, BK, STAGED>(a, tiles_m, tiles_n, s);
    return vec ? launch_variant<true, 3, BK, STAGED>(a, tiles_m, tiles_n, s)
               : launch_variant<false, 3, BK, STAGED>(a, tiles_m, tiles_n, s);
}

}  // namespace

namespace vrd {
// FLOPs of contractions the exact-f32 kernel skipped through padding maps since the last call (reads and clears the device counter)
double take_f32_skipped_flops() {
    unsigned long long v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f32_skipped_kn), sizeof(v)) != hipSuccess) return 0.0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f32_skipped_kn), &zero, sizeof(zero));
    return 2.0 * BM * (double)v;
}
int launch_gemm_x3(const vrd_gemm_args& a, bool staged, hipStream_t s);
bool gemm_x3_dma_ok(const vrd_gemm_args& a, bool staged);
int launch_gemm_x3_dma(const vrd_gemm_args& a, hipStream_t s);
int launch_gemm_x3_big(const vrd_gemm_args& a, hipStream_t s);
int launch_gemm_x3_big_batch(const vrd_gemm_args* a, int count, hipStream_t s);
}  // namespace vrd

namespace {
// which split-precision kernel serves these arguments (shape, layout and alignment rules; see the kernels' files)
struct X3Choice {
    bool x3, dma, big;
};
X3Choice choose_x3(const vrd_gemm_args* a, bool vec, bool staged) {
    const int K = a->Cin * a->taps;
    const bool x3 = a->a_pair_width > 0 || (a->W_split && vec && (K % 32 == 0) && aligned16(a->W_split));
    static const int dma_env = [] { const char* e = getenv("VRD_X3_DMA"); return e ? atoi(e) : 1; }();
    // the 128 x 256 DMA kernel runs one workgroup per CU: below ~2 rounds of tiles the 128 x 128 kernel (two
    // workgroups per CU, four times the tiles) fills the chip better
    static const int64_t dma_min_tiles = [] { const char* e = getenv("VRD_X3_DMA_MIN_TILES"); return e ? atoll(e) : 512; }();
    const bool dma = x3 && dma_env && vrd::gemm_x3_dma_ok(*a, staged) &&
                     ((a->M + 127) / 128) * ((a->N + 255) / 256) >= dma_min_tiles;
    // the 256 x 256 kernel (DMA-issue cost per MFMA a third lower) once there are about two rounds of its tiles
    static const int64_t big_min_tiles = [] { const char* e = getenv("VRD_X3_BIG_MIN_TILES"); return e ? atoll(e) : 512; }();
    const bool big = dma && a->N >= 256 && K >= 96 && vrd::gemm_epilogue_lean_ok(*a) &&      // (its K loop is written for >= 3 steps)
                     (a->taps == 1 || a->T >= 32) &&      // k = 3: the kernel steps its sequence position by 8 rows per piece
                     (reinterpret_cast<uintptr_t>(a->A) & 127u) == 0 && (reinterpret_cast<uintptr_t>(a->W_split) & 127u) == 0 &&
                     ((a->M + 255) / 256) * ((a->N + 255) / 256) >= big_min_tiles;
    return X3Choice{x3, dma, big};
}
}  // namespace

// Host-side validation shared by vrd_gemm and vrd_gemm_batch: nothing is launched for a problem that fails it.
namespace {
int validate_gemm_args(const vrd_gemm_args* a) {
    VRD_CHECK_ARG(a->A && a->W && a->C, "vrd_gemm: null operand");
    VRD_CHECK_ARG(a->M >= 0 && a->N > 0 && a->Cin > 0, "vrd_gemm: bad sizes M=%lld N=%d Cin=%d",
                  (long long)a->M, a->N, a->Cin);
    VRD_CHECK_ARG(a->taps == 1 || a->taps == 3, "vrd_gemm: taps must be 1 or 3 (got %d)", a->taps);
    VRD_CHECK_ARG(a->taps == 1 || (a->T > 0 && a->M % a->T == 0),
                  "vrd_gemm: k=3 conv needs M (%lld) to be a multiple of T (%d)", (long long)a->M, a->T);
    VRD_CHECK_ARG(a->lda >= a->Cin && a->ldc >= a->N, "vrd_gemm: leading dimension too small");
    VRD_CHECK_ARG(a->act >= 0 && a->act <= 2, "vrd_gemm: bad activation %d", a->act);
    VRD_CHECK_ARG(!a->res || a->ldres >= a->N, "vrd_gemm: ldres too small");
    VRD_CHECK_ARG(!a->res2 || a->ldres2 >= a->N, "vrd_gemm: ldres2 too small");
    VRD_CHECK_ARG(!a->c_pair || (a->N % 32 == 0 && a->ldc % 32 == 0 && aligned16(a->C)),
                  "vrd_gemm: pair output needs N %% 32 == 0 and rows that start on a 128-byte block");
    VRD_CHECK_ARG(a->a_pair_width == 0 || (a->W_split && a->Cin % 32 == 0 && a->lda % 32 == 0 &&
                                           aligned16(a->A) && aligned16(a->W_split)),
                  "vrd_gemm: pair-row A needs W_split, Cin %% 32 == 0 and rows that start on a 128-byte block");
    VRD_CHECK_ARG(a->split_fmt == 0 || a->split_fmt == VRD_PAIR_BF16 || a->split_fmt == VRD_PAIR_F16, "vrd_gemm: bad split_fmt %d", a->split_fmt);
    VRD_CHECK_ARG(a->split_fmt != VRD_PAIR_F16 || !a->W_split || a->w_scale, "vrd_gemm: a VRD_PAIR_F16 W_split needs w_scale");
    VRD_CHECK_ARG(a->c_pair == VRD_PAIR_NONE || a->c_pair == VRD_PAIR_BF16 || a->c_pair == VRD_PAIR_F16, "vrd_gemm: bad c_pair %d", a->c_pair);
    VRD_CHECK_ARG(!a->a_scale || (a->split_fmt == VRD_PAIR_F16 && a->a_pair_width == 0),
                  "vrd_gemm: a_scale goes with f32-row A in the VRD_PAIR_F16 format");
    VRD_CHECK_ARG(!a->c_pair || !a->W_split || a->c_pair == (a->split_fmt ? a->split_fmt : (int)VRD_PAIR_BF16),
                  "vrd_gemm: pair output (format %d) of a split-precision GEMM must be in its operand format (%d)", a->c_pair, a->split_fmt);
    VRD_CHECK_ARG(!a->row_blocks || (a->row_blocks_active && a->M % 32 == 0 && a->row_block_seg_len >= 8 &&
                                     a->row_block_seg_len % 8 == 0),
                  "vrd_gemm: a row-block list needs its active counts, M %% 32 == 0 and a segment length that is a multiple of 8 (M = %lld, seg_len %d)",
                  (long long)a->M, a->row_block_seg_len);
    return 0;
}
}  // namespace

extern "C" int vrd_gemm(const vrd_gemm_args* a, void* stream) {
    VRD_CHECK_ARG(a != nullptr, "vrd_gemm: null args");
    if (int rc = validate_gemm_args(a)) return rc;
    if (a->M == 0) return 0;
    const int64_t tiles_m64 = (a->M + BM - 1) / BM;
    const int tiles_n = (a->N + BN - 1) / BN;
    VRD_CHECK_ARG(tiles_m64 * tiles_n < (int64_t)1 << 31, "vrd_gemm: grid too large");
    const int tiles_m = (int)tiles_m64;
    const int K = a->Cin * a->taps;
    const bool vec = (a->Cin % 4 == 0) && (a->lda % 4 == 0) && aligned16(a->A) && aligned16(a->W);
    // float4 epilogue needs 16-byte aligned output / residual rows
    const bool staged = (a->ldc % 4 == 0) && aligned16(a->C) && (!a->res || (a->ldres % 4 == 0 && aligned16(a->res))) &&
                        (!a->res2 || (a->ldres2 % 4 == 0 && aligned16(a->res2)));
    static const int bk_env = [] { const char* e = getenv("VRD_GEMM_BK"); return e ? atoi(e) : 0; }();
    const int bk = bk_env == 32 ? 32 : 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double flops = 2.0 * (double)a->M * a->N * K;
    const double bytes = 4.0 * ((double)a->M * a->Cin + (double)a->N * K + (double)a->M * a->N *
                                (1.0 + (a->res ? 1.0 : 0.0) + (a->res2 ? 1.0 : 0.0)));
    const X3Choice pick = choose_x3(a, vec, staged);
    const bool x3 = pick.x3, dma = pick.dma, big = pick.big;
    vrd::ProfScope prof(big ? VRD_K_GEMM_X3_BIG : dma ? VRD_K_GEMM_X3_DMA : (x3 ? VRD_K_GEMM_X3 : VRD_K_GEMM), s, flops, bytes);
    if (x3) {
        int rc3 = big ? vrd::launch_gemm_x3_big(*a, s)
                      : dma ? vrd::launch_gemm_x3_dma(*a, s) : vrd::launch_gemm_x3(*a, staged, s);
        if (rc3) return rc3;
        VRD_LAUNCH_CHECK();
        return 0;
    }
    // exact f32 products on the unscaled W: the split operand's format and scale (the epilogue's accumulator factor) do not apply
    // -- a call that carries W_split but does not qualify for a split kernel (K % 32, alignment) lands here
    vrd_gemm_args f = *a;
    f.split_fmt = 0;
    f.w_scale = nullptr;
    f.a_scale = nullptr;
    VRD_CHECK_ARG(!f.c_pair || f.c_pair == VRD_PAIR_BF16 || f.c_pair == VRD_PAIR_F16, "vrd_gemm: bad c_pair");
    int rc;
    if (bk == 32) rc = staged ? launch_bk<32, true>(f, vec, tiles_m, tiles_n, s) : launch_bk<32, false>(f, vec, tiles_m, tiles_n, s);
    else          rc = staged ? launch_bk<16, true>(f, vec, tiles_m, tiles_n, s) : launch_bk<16, false>(f, vec, tiles_m, tiles_n, s);
    if (rc) return rc;
    VRD_LAUNCH_CHECK();
    return 0;
}

// Several GEMMs as one launch where possible: problems that differ only in A, W / W_split, bias and C and that the
// 256 x 256 kernel takes (the q / k / v projections of an attention block) run as one grid; anything else runs one by one.
extern "C" int vrd_gemm_batch(const vrd_gemm_args* a, int count, void* stream) {
    VRD_CHECK_ARG(a != nullptr && count >= 1 && count <= 4, "vrd_gemm_batch: 1..4 problems (got %d)", count);
    for (int i = 0; i < count; ++i)
        if (int rc = validate_gemm_args(&a[i])) return rc;
    bool same = count > 1 && a[0].M > 0;
    for (int i = 1; i < count && same; ++i) {
        const vrd_gemm_args &x = a[i], &y = a[0];
        same = x.lda == y.lda && x.ldc == y.ldc && x.M == y.M && x.N == y.N && x.Cin == y.Cin && x.taps == y.taps && x.T == y.T &&
               x.act == y.act && x.row_mask == y.row_mask && x.scale == y.scale && x.res == y.res && x.ldres == y.ldres &&
               x.res_masked == y.res_masked && x.res2 == y.res2 && x.ldres2 == y.ldres2 && x.a_pair_width == y.a_pair_width &&
               x.c_pair == y.c_pair && x.row_blocks == y.row_blocks && x.row_blocks_active == y.row_blocks_active &&
               x.row_block_seg_len == y.row_block_seg_len && (x.bias != nullptr) == (y.bias != nullptr) &&
               (x.W_split != nullptr) == (y.W_split != nullptr) && x.split_fmt == y.split_fmt;
    }
    static const int batch_env = [] { const char* e = getenv("VRD_GEMM_BATCH"); return e ? atoi(e) : 1; }();
    if (same && batch_env) {
        bool all_big = true;
        for (int i = 0; i < count && all_big; ++i) {
            const vrd_gemm_args* p = &a[i];
            if (!(p->A && p->W && p->C && p->lda >= p->Cin && p->ldc >= p->N)) { all_big = false; break; }
            const bool vec = (p->Cin % 4 == 0) && (p->lda % 4 == 0) && aligned16(p->A) && aligned16(p->W);
            const bool staged = (p->ldc % 4 == 0) && aligned16(p->C) && (!p->res || (p->ldres % 4 == 0 && aligned16(p->res))) &&
                                (!p->res2 || (p->ldres2 % 4 == 0 && aligned16(p->res2)));
            all_big = choose_x3(p, vec, staged).big && (!p->c_pair || (p->N % 32 == 0 && p->ldc % 32 == 0)) &&
                      (!p->row_blocks || (p->row_blocks_active && p->row_block_seg_len >= 8 && p->row_block_seg_len % 8 == 0));
        }
        if (all_big) {
            hipStream_t s = static_cast<hipStream_t>(stream);
            const int K = a[0].Cin * a[0].taps;
            const double flops = 2.0 * (double)a[0].M * a[0].N * K * count;
            const double bytes = 4.0 * count * ((double)a[0].M * a[0].Cin + (double)a[0].N * K + (double)a[0].M * a[0].N *
                                                (1.0 + (a[0].res ? 1.0 : 0.0) + (a[0].res2 ? 1.0 : 0.0)));
            vrd::ProfScope prof(VRD_K_GEMM_X3_BIG, s, flops, bytes);
            int rc = vrd::launch_gemm_x3_big_batch(a, count, s);
            if (rc) return rc;
            VRD_LAUNCH_CHECK();
            return 0;
        }
    }
    for (int i = 0; i < count; ++i) {
        int rc = vrd_gemm(&a[i], stream);
        if (rc) return rc;
    }
    return 0;
}
