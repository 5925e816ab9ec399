/*
 * vrdone_hip.h -- C ABI of libvrdone_hip.so: hand-written gfx950 (MI355X) kernels for the
 * VrdONE relation-encoding hot path (MaskVRD._mask_vrd and its eval post-processing).
 *
 * The reference has no FFI; its boundary for this path is the Python module surface of
 * models/ (SURVEY 8b).  Each entry point below replaces the ATen call sites of the
 * reference functions cited next to it (paths relative to the reference checkout) and is
 * bound from Python with ctypes (vrdone_amd/_hip.py; INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - every activation tensor is fp32, channels-last: a matrix of `rows` x `C` with an
 *    explicit leading dimension `ld*` (floats between consecutive rows); row r = b*T + t.
 *    Leading dimensions let producers write straight into concatenation buffers.
 *  - "pair" rows (split-precision modes only): a tensor whose only consumer is a GEMM (or the global attention) may be
 *    produced in the operand format of the split-precision MFMA: the 4*C bytes of a C-channel row (C % 32 == 0,
 *    row start 16-byte aligned) hold, per block of 32 channels, [32 x 16-bit hi | 32 x 16-bit lo]: one 128-byte line
 *    per 32 channels.  Same leading dimension as the f32 row.  Producers take an `out_pair` argument (enum
 *    vrd_pair_format: 0 = plain f32 rows); vrd_gemm takes `a_pair_width` (> 0: A is pair rows in the format
 *    `split_fmt`; 0 for f32 input).
 *  - masks are uint8 (0/1), one byte per row (the reference's (B,1,T) bool mask).
 *  - all pointers are device pointers owned by the caller; the library allocates no
 *    device memory and keeps no state besides the optional profiling event list.
 *  - `stream` is a hipStream_t; work is queued asynchronously on it.
 *  - return value: 0 on success, negative on error (message via vrd_last_error()).
 *    Arguments are validated on the host before any launch.
 */
#ifndef VRDONE_HIP_H
#define VRDONE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRD_ABI_VERSION 34

enum vrd_act { VRD_ACT_NONE = 0, VRD_ACT_RELU = 1, VRD_ACT_GELU = 2 };

/* Element format of pair rows and of the split weight operand (W_split).  Both replace an f32 product a*w by
 * a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the 16-bit MFMA with f32 accumulation.
 *   VRD_PAIR_BF16 ("bf16x3"): hi = bf16(x), lo = bf16(x - hi).  ~17 significand bits per product; the full f32 range.
 *   VRD_PAIR_F16  ("f16x3"):  hi = f16(y), lo = f16(y - hi) of y = x * 2^e, round to nearest: ~22 significand bits per
 *       product -- the reference-grade mode: end to end its distance to a float64 run of the reference is within
 *       1.0-1.3x the distance of the reference's own float32 run (tests/golden/mask_vrd_f64.npz).  e is the fixed
 *       VRD_F16_ACT_EXP for activations (|x| < 4094 representable; larger magnitudes come out as NaN, never as a
 *       silently wrong number) and a per-tensor power of two for weights (vrd_split_weight: max |w| * 2^e in
 *       [2^14, 2^15)); the GEMM epilogue multiplies the accumulator by the exact power of two that undoes both. */
enum vrd_pair_format { VRD_PAIR_NONE = 0, VRD_PAIR_BF16 = 1, VRD_PAIR_F16 = 2 };
#define VRD_F16_ACT_EXP 4

/* kernel families, for vrd_prof_read() */
enum vrd_kernel_id {
    VRD_K_GEMM = 0, VRD_K_LAYERNORM = 1, VRD_K_DWCONV_LN = 2, VRD_K_LOCAL_ATTN = 3,
    VRD_K_ATTN_SMALL = 4, VRD_K_ATTN_FLASH = 5, VRD_K_POOL = 6, VRD_K_MASK_HEAD = 7,
    VRD_K_TRANSPOSE = 8, VRD_K_POSTPROC = 9, VRD_K_GEMM_X3 = 10, VRD_K_GEMM_X3_DMA = 11, VRD_K_GEMM_X3_BIG = 12,
    VRD_K_BACKWARD = 13,
    VRD_K_COUNT = 14
};

int vrd_abi_version(void);
const char* vrd_last_error(void);

/* ---- operand range of the VRD_PAIR_F16 format ----
 * A value x with |x * 2^VRD_F16_ACT_EXP| >= 65,520 does not fit the f16 planes of a pair row: it is stored as hi = inf,
 * lo = -inf and poisons every product it enters, but a NaN does not reach the outputs reliably (the ReLU behind the embedding
 * LayerNorms and the max-pools of the branch blocks drop it: fmax(NaN, x) = x).  Every kernel that WRITES pair rows in this
 * format, or splits f32 rows into them while staging, therefore reports: it ORs a tag naming its family into one 32-bit flag
 * word per device -- 1 boundary tensors (vrd_bct_to_btc, vrd_pack_pairs, vrd_gather_pairs), 2 vrd_layernorm, 4 vrd_dwconv_ln,
 * 8 pair-row outputs of vrd_gemm / vrd_gemm_batch, 16 f32 rows split inside vrd_gemm, 32 pair-row outputs of vrd_local_attn /
 * vrd_attention.  (vrd_attention_pair needs none: its outputs are averages of value rows that were checked when they were
 * written.)  The word lives in device memory for the life of the process; the caller reads it together with a call's
 * results (an asynchronous 4-byte copy on the same stream), repeats the call in another precision if it is non-zero, and
 * clears it with a 4-byte memset.  The reference computes in float32 and has no such limit (models/blocks.py:728-737).
 * `flag` receives the device address of the CURRENT device's word. */
int vrd_f16_range_flag(void** flag);

/* ---- profiling: HIP events around every launch of a family, on the launch stream ---- */
int vrd_prof_enable(int on);
int vrd_prof_reset(void);
/* synchronises the recorded events; ms = summed kernel time, flops = summed algorithmic
 * FLOPs (GEMM / attention families; 0 elsewhere), bytes = summed algorithmic HBM bytes. */
int vrd_prof_read(int kernel_id, double* ms, int64_t* launches, double* flops, double* bytes);
/* Of the FLOPs vrd_prof_read reports for the family (what the launches were sized for), the part that padding maps
 * (vrd_gemm_args.row_blocks) made the kernels skip: executed = flops - flops_skipped. */
int vrd_prof_read_skipped(int kernel_id, double* flops_skipped);
/* Restrict event recording to the kernel families whose bit (1 << vrd_kernel_id) is set (default: all).  Every
 * recorded launch costs two event records, which also keep consecutive kernels from overlapping: ~3 us per launch,
 * 8 % of a 256-pair step with all ~520 launches recorded, 1 % of a 2048-pair step. */
int vrd_prof_select(unsigned long long family_mask);

/* ---- layout change at the boundary ------------------------------------------------------
 * (B, C_total, T) -> rows (b*T+t) of `dst`, channels [c0, c0+count) of the source.
 * Replaces the channel slicing of models/backbones.py:161-166 / :329-341 and the
 * transposing copy of models/maskvrd.py:382-385.
 * T_src (0: = T): frames per channel row of the source, of which the first T are converted; src_batch (nullable, B
 * device int32): output sequence b is source sequence src_batch[b].  Together: a batch of selected pairs at a shorter
 * padded length straight from the caller's (B_all, C_total, T_src) tensor (MaskVRD's tight padding). */
int vrd_bct_to_btc(const float* src, int B, int C_total, int T, int c0, int count,
                   float* dst, int64_t ld_dst, int out_pair /* enum vrd_pair_format */, int T_src, const int32_t* src_batch,
                   void* stream);
/* Eval batching (models/maskvrd.py:363-414 + backbones.py:161-166) without the (B, C_in, T) intermediate: pair p
 * is an (L_p, C_in) frame-major matrix src[p] (how dataloaders/vidvrd.py:693 builds it, before its permute view)
 * laid out [s_vis V | o_vis V | (s_clip Cc | o_clip Cc) | so_box S | s_box E | o_box E].  Writes the backbone's
 * channels-last operand buffers, zero rows for t >= lens[p]:
 *   vis  (2, P, T, V)   subject rows then object rows (pair rows when pair_wide)
 *   clip (2, P, T, Cc)  only when Cc > 0 (pair rows when pair_wide)
 *   so_box (P, T, S), ent (2, P, T, E)   f32
 * `src` is a device array of P device pointers. */
typedef struct {
    const float* const* src;
    const int32_t* lens;
    int32_t P, C_in, T, V, Cc, S, E;
    float* vis;
    float* clip;
    float* so_box;
    float* ent;
    int32_t pair_wide;      /* enum vrd_pair_format of vis / clip */
} vrd_pack_args;
int vrd_pack_pairs(const vrd_pack_args* a, void* stream);

/* Eval batching from PER-TRACKLET features (what dataloaders/vidvrd.py:652-693 + utils/misc.py:158-217 do on the host,
 * one (L, C_in) matrix per pair): vis (sum L, V), clip (sum L, Cc) or NULL and boxes (sum L, 4; x0 y0 x1 y1, already
 * clamped to the frame, dataloaders/vidvrd.py:567-571) hold every tracklet's frames once; pair p covers `lens[p]` frames,
 * frame t being row s_row[p] + t*stride of the subject's arrays and o_row[p] + t*stride of the object's (the dataloader's
 * feat[start_offset::feat_stride] slicing, :678-692).  Writes the same operand buffers as vrd_pack_pairs (zero rows for
 * t >= lens[p]); so_box = the 5 subject-object box features, ent = the 8 entity box features of subject then object rows
 * (boxes normalised by the frame size w x h, first differences along the sub-sampled frames).
 * out_vis == NULL: only the box features are written (the wide rows then come from vrd_assemble_pairs). */
typedef struct {
    const float* vis;
    const float* clip;
    const float* boxes;
    const int64_t* s_row;
    const int64_t* o_row;
    const int32_t* lens;
    int32_t P, T, V, Cc, stride;
    float w, h;
    float* out_vis;
    float* out_clip;
    float* out_so_box;
    float* out_ent;
    int32_t pair_wide;      /* enum vrd_pair_format of out_vis / out_clip */
} vrd_gather_args;
int vrd_gather_pairs(const vrd_gather_args* a, void* stream);

/* Entity-stage rows of a batch of P pairs put together from rows computed ONCE PER TRACKLET (second half).
 * The reference runs the embedding convs, the visual/box fusion and the first stem block on the subject and the object
 * of every pair (models/backbones.py:172-214), although every op there sees one tracklet only and reaches at most `reach`
 * frames to either side: frame t of a pair with n frames equals the same frame computed on the whole tracklet when
 * reach <= t < n - reach; only the frames within `reach` of the pair's window edges (zero padding there) differ.
 *   streams   (., D)        rows of the per-tracklet computation; stream_row[e] = the row holding frame 0 of entity e
 *                           (e = p for the subject, P + p for the object of pair p)
 *   snippets  (4P, L, D)    the same stage run on short pieces at the window edges, `piece` >= 2*reach frames each, in
 *                           buffers of L > piece frames: [subject start P | subject end P | object start P | object end P].
 *                           A start piece holds frames [0, min(n, piece)); an end piece holds frames [n - piece, n) followed by
 *                           padding -- like the pair's own rows when n < T: the reference's LayerNorms turn a padded frame
 *                           into their bias, which the next conv reads (models/backbones.py:196-207, blocks.py:828-860) -- or,
 *                           when n == T (no padded frame follows in the pair's batch either), frames [n - L, n) filling the
 *                           buffer.  End pieces are unused when n <= piece.
 *   out       (2P, T, D)    subject rows then object rows; zero for t >= n */
typedef struct {
    const float* streams;
    const float* snippets;
    const int64_t* stream_row;
    const int32_t* lens;
    int32_t P, T, D, L, piece, reach;
    float* out;
} vrd_assemble_args;
int vrd_assemble_pairs(const vrd_assemble_args* a, void* stream);

/* rows (b*T+t) x C (leading dim ld_src) -> (B, C, T) contiguous. */
int vrd_btc_to_bct(const float* src, int64_t ld_src, int B, int C, int T, float* dst, void* stream);

/* ---- dense 1-D convolution as (implicit) GEMM on f32 MFMA -------------------------------
 * C[r, n] = epilogue( sum_{tap, ci} A[r + tap - taps/2, ci] * W[n, tap*Cin + ci] )
 * taps = 1: pointwise conv (nn.Conv1d k=1: models/blocks.py:728-737,1054-1060,45-48;
 *           models/local_transformer.py:133-142; models/fpns.py:199-201;
 *           models/predictor.py:73,79,83).
 * taps = 3: dense k=3 conv with zero padding inside each length-T sequence
 *           (models/backbones.py:59-63,74,127,309-313 via models/blocks.py:99).
 * epilogue: v = acc + bias[n]; v = act(v); v *= row_mask[r]; v *= scale[n];
 *           v += res[r, n] * (res_masked ? row_mask[r] : 1); v += res2[r, n].
 * That covers MaskedConv1D's mask (blocks.py:111), GELU/ReLU (blocks.py:59,1056),
 * AffineDropPath in eval (blocks.py:1148-1149) and the residual forms of
 * blocks.py:1074-1076 and local_transformer.py:815,829,833; res2 is the SOS stream
 * update of backbones.py:220-221.
 * W is [N][K] row-major with K = taps*Cin, tap-major (the caller re-packs a (N,Cin,3)
 * Conv1d weight once).
 * Precision: with W_split == NULL every product is an exact f32 MFMA product.  With W_split set
 * (vrd_split_weight's output in the format `split_fmt`) and K % 32 == 0, 16-byte aligned operands, the
 * kernel splits f32 activations the same way on the fly (or takes pair rows) and forms a*w ~= a_hi*w_hi + a_hi*w_lo +
 * a_lo*w_hi on the 16-bit MFMA with f32 accumulation (enum vrd_pair_format); shapes that do
 * not qualify silently use the f32 kernel. */
typedef struct {
    const float* A;  int64_t lda;
    const float* W;
    const float* bias;
    float* C;  int64_t ldc;
    int64_t M;  int32_t N;  int32_t Cin;  int32_t taps;  int32_t T;
    int32_t act;
    const uint8_t* row_mask;
    const float* scale;
    const float* res;  int64_t ldres;  int32_t res_masked;
    const float* res2;  int64_t ldres2;
    const uint16_t* W_split; /* optional 16-bit split of W in pair-row blocks, (N, K/32, [32 hi | 32 lo]), K % 32 == 0:
                                enables the split-precision MFMA path */
    int32_t a_pair_width;   /* > 0: A rows are pair rows in the format split_fmt (needs W_split, Cin % 32 == 0, lda % 32 == 0) */
    int32_t c_pair;         /* enum vrd_pair_format: write C as pair rows of width N (N % 32 == 0, ldc % 32 == 0); a format
                               other than VRD_PAIR_NONE must equal split_fmt */
    /* Optional padding skip (M % 32 == 0): row_blocks = a permutation of the M/32 indices of 32-row blocks in
     * segments of row_block_seg_len entries (a multiple of 8; the last segment may be shorter), each holding its
     * blocks with a valid frame first (vrd_row_blocks); row_blocks_active[segment] = how many those are (device
     * memory).  A 256-row tile takes eight consecutive list entries; tiles behind a segment's active blocks skip the
     * contraction and get the epilogue of a zero accumulator: exactly the reference value where row_mask zeroes the
     * row (res / res2 still added), bias-only filler otherwise -- only for GEMMs whose masked rows no valid row
     * reads (q/k/v projections feeding masked attention, the MLP hidden).  Kernels without the indirection ignore
     * it and compute every row. */
    const int32_t* row_blocks;
    const int32_t* row_blocks_active;
    int32_t row_block_seg_len;
    int32_t split_fmt;      /* format of W_split (and of pair-row A): VRD_PAIR_BF16 (also when 0) or VRD_PAIR_F16 */
    const float* w_scale;   /* VRD_PAIR_F16 only: device pointer to the factor that turns the accumulator of the scaled
                               operands back into the product, 2^-(e_w + VRD_F16_ACT_EXP) -- element 0 of vrd_split_weight's
                               `scale` output for W_split */
    const float* a_scale;   /* VRD_PAIR_F16 with f32-row A (a_pair_width == 0) only, nullable: device pointer to two floats
                               (2^e, 2^-e), vrd_absmax_scale's output for A: the rows are split as f16 planes of A * 2^e instead
                               of A * 2^VRD_F16_ACT_EXP, for operands without a known range -- the gradients of the training
                               step's input-gradient GEMMs (the reference differentiates in float32, train.py:182-186) */
} vrd_gemm_args;
int vrd_gemm(const vrd_gemm_args* a, void* stream);
/* `count` (1..4) GEMMs of an array of argument structs.  Problems that differ only in A, W / W_split, bias and C and
 * that the 256 x 256 split-precision kernel takes -- the q / k / v projections of one attention block
 * (models/blocks.py:935-947, models/local_transformer.py:157-161) -- run as ONE grid (one ragged last round of tiles
 * instead of three); anything else is the same as calling vrd_gemm on each. */
int vrd_gemm_batch(const vrd_gemm_args* a, int count, void* stream);

/* Padding map of a channels-last activation matrix: rows = B*T flat rows with validity mask[rows] (models/maskvrd.py
 * :386-392 builds it as t < len_b), rows % 32 == 0.  The rows/32 block indices are dealt into segments of seg_len
 * entries of order[] (the last may be shorter; at most 64 segments): segment s = its proportional share of the blocks
 * with at least one valid row (ascending), then its share of the fully padded ones (ascending); n_active[s] = number
 * of the former.  For vrd_gemm choose seg_len = 8 * ceil(ceil(rows/32 / 8) / 8): about one segment per XCD of its
 * tile order, so every XCD gets the same amount of padding to skip.  The reference computes every padded frame
 * (T_pad = 288 for 256 valid frames; max_seq_len for short pairs). */
int vrd_row_blocks(const uint8_t* mask, int64_t rows, int seg_len, int32_t* order, int32_t* n_active, void* stream);

/* ---- dense conv with FEW input channels * mask -> [LayerNorm -> ReLU] as one row kernel (round 6) ----------------------
 * The box-feature embeddings: MaskedConv1D(n_bbox_entity = 8 -> 512, k = 3) -> LayerNorm -> ReLU (models/backbones.py:174-178 with
 * blocks.py:45-48, 143-158) and MaskedConv1D(n_bbox_so = 5 -> 512, k = 3) (backbones.py:208-210).  With taps * Cin <= 32 the
 * contraction is 24 (15) multiply-adds per output: as a GEMM it ran on the exact-f32 MFMA kernel, bound by writing its 2.4 GB
 * output, which the LayerNorm then read back and wrote again; here a wave computes a row's N outputs on the vector units (f32 FMAs,
 * the row's inputs through the scalar cache, the weights from LDS), normalises them in registers and stores them once.
 *   y[r, n] = act(LN_n((bias[n] + sum_{tap, ci} W[n, ci, tap] * x[r + tap - taps/2, ci]) * row_mask[r]))
 * x: rows of Cin f32 (leading dimension ldx); rows outside r's length-T sequence read 0 (Conv1d zero padding); W: the Conv1d
 * parameter (N, Cin, taps) contiguous; N in {256, 512}; taps in {1, 3}; taps * Cin <= 32 and (3 + taps) * Cin <= 64 (the inputs of four
 * output rows are one wave-wide load); gamma / beta NULL: no LayerNorm;
 * y: f32 rows or (out_pair) pair rows of width N. */
typedef struct {
    const float* x;
    int64_t ldx;
    int64_t rows;
    int Cin, taps, T, N;
    const float* w;
    const float* bias;              /* nullable */
    const uint8_t* row_mask;        /* nullable: rows */
    const float* gamma;             /* nullable (with beta): N */
    const float* beta;
    int relu;
    float* y;
    int64_t ldy;
    int out_pair;                   /* enum vrd_pair_format */
} vrd_conv_ln_args;
int vrd_conv_ln(const vrd_conv_ln_args* a, void* stream);

/* ---- channel LayerNorm (models/blocks.py:143-158), C in {256, 512} ----------------------
 * y[r,:] = LN(x[r,:]) * gamma + beta; optional ReLU (backbones.py:174); optional
 * post_add[(r % add_period), :] added after the affine (query_pos of
 * models/local_transformer.py:809,820). */
int vrd_layernorm(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C,
                  const float* gamma, const float* beta, int relu,
                  const float* post_add, int64_t ld_add, int add_period, int out_pair /* enum vrd_pair_format */, void* stream);

/* A ragged row space (host memory): the rows of an activation are `count` groups of sequences back to back, group i =
 * n[i] sequences of T[i] frames from row row[i] on -- the buckets of equal padded length of MaskVRD's tight padding
 * (vrdone_amd/models/ragged.py; the reference pads every pair of a batch to one length, models/maskvrd.py:363-414).
 * Kernels that walk sequences take it in place of (B, T): one launch for all groups. */
#define VRD_MAX_SEGS 32
typedef struct {
    int32_t count;                 /* 1 .. VRD_MAX_SEGS */
    int32_t n[VRD_MAX_SEGS];
    int32_t T[VRD_MAX_SEGS];
    int64_t row[VRD_MAX_SEGS];
} vrd_row_segs;

/* ---- depthwise conv (+ nearest-upsample add) * mask -> LayerNorm, fused -----------------
 * for o in [0, n_out): y_o[b,t',:] = LN_o( mask_out[b,t'] * (bias_o + sum_k w_o[c,g,k] *
 *        xin[b, stride*t' + k - ksize/2, group_in*c + g]) ), xin = x (+ x_up[b, t/2, :]), or, with pre_gamma set,
 *        xin = LayerNorm(x; pre_gamma, pre_beta) (the ln1 of models/blocks.py:1064 applied to every input row as it
 *        is read; rows outside [0, Tin) stay the zero padding of the convolution).
 * MaskedConv1D depthwise + LayerNorm pairs of models/blocks.py:927-933,
 * models/local_transformer.py:149-156, models/fpns.py:246-254 (group_in = 2 is the
 * top-level Conv1d(512,256,3,groups=256) of fpns.py:181-184), and mask_features
 * (fpns.py:226,256) with gamma = NULL (no LN) and a bias. */
typedef struct {
    const float* x;  int64_t ldx;
    const float* x_up;  int64_t ldx_up;
    int32_t B, Tin, C, ksize, stride, group_in;
    const uint8_t* mask_out;
    int32_t n_out;
    const float* w[3];
    const float* bias[3];
    const float* gamma[3];
    const float* beta[3];
    int32_t relu[3];
    float* y[3];
    int64_t ldy[3];
    int32_t out_pair[3];        /* enum vrd_pair_format per set */
    const float* pre_gamma;     /* optional LayerNorm on the input rows (needs group_in == 1, no x_up) */
    const float* pre_beta;
    const float* packed[3];     /* optional per set: the set's parameters as the kernel keeps them on chip,
                                   (group_in*ksize + 3) * C floats = taps [group_in][ksize][C] | bias [C] (zeros if
                                   none) | gamma [C] (ones) | beta [C] (zeros); replaces w / bias and the values of
                                   gamma / beta (whose pointers still say whether the set has a LayerNorm) */
    const vrd_row_segs* segs;   /* nullable (host memory): the INPUT rows as groups of sequences (T[i] % stride == 0; row[i] even when
                                   x_up or stride 2 is used) -- B and Tin are then ignored; group i's output rows start at
                                   row[i] / stride, its x_up rows at row[i] / 2 */
} vrd_dwconv_ln_args;
int vrd_dwconv_ln(const vrd_dwconv_ln_args* a, void* stream);

/* ---- banded (local-window) attention, models/blocks.py:950-986 ---------------------------
 * query t attends keys j in [t-half_win, t+half_win] within [0,T); masked keys get -1e4,
 * masked query rows give 0.  q is scaled by head_dim^-0.5 inside.  C = n_head*head_dim = 512.
 * rel_pe: NULL, or the relative position bias [n_head][2*half_win+1] added to the scaled scores
 * before the key mask (`use_rel_pe`, models/blocks.py:739-743,957-958). */
int vrd_local_attn(const float* q, const float* k, const float* v, int64_t ld,
                   const uint8_t* mask, const float* rel_pe, int B, int T, int C, int n_head, int half_win,
                   float* out, int64_t ldo, int out_pair, void* stream);
/* The same over a ragged row space (segs: groups of sequences, see vrd_row_segs) in one launch. */
int vrd_local_attn_segs(const float* q, const float* k, const float* v, int64_t ld,
                        const uint8_t* mask, const float* rel_pe, const vrd_row_segs* segs, int C, int n_head, int half_win,
                        float* out, int64_t ldo, int out_pair, void* stream);

/* ---- global masked attention, models/local_transformer.py:163-183 and :44-63 -------------
 * out[b,tq,h,:] = softmax_j(q.k_j / sqrt(hd) | kv_mask[b,j]) . v_j ; keys with mask 0 get
 * -inf.  kv_mask may be NULL (all keys valid).  `algo`: 0 = auto, 1 = generic VALU kernel
 * (any shape; the predictor's 9-query attention), 2 = f32-MFMA flash kernel (hd in {64,128}).
 * out_pair is honoured by the flash kernel only (row width n_head*head_dim). */
int vrd_attention(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                  const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, int head_dim,
                  float* out, int64_t ldo, int algo, int out_pair, void* stream);

/* Same attention for pair-row q, k, v (split-precision modes; rows of width n_head*head_dim written by the projection
 * GEMMs with c_pair): both contractions as three 16-bit MFMA products, f32 softmax.  head_dim in {64, 128}.
 * pair_fmt: the format of q, k, v (VRD_PAIR_BF16 / VRD_PAIR_F16); out_pair: 0 or the same format.
 * q_mask (optional, B*Tq bytes): query rows the caller zeroes afterwards anyway (the output projection's row mask,
 * local_transformer.py:183); tiles of 32 queries without a valid one are not computed and read 0. */
int vrd_attention_pair(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv,
                       const uint8_t* kv_mask, const uint8_t* q_mask, int B, int Tq, int Tk, int n_head, int head_dim,
                       float* out, int64_t ldo, int out_pair, int pair_fmt, void* stream);

/* ---- MaxPool1d(3,2,1) skip * mask (models/blocks.py:1040-1046,1074) and mask[::2] ------- */
int vrd_maxpool_mask(const float* x, int64_t ldx, int B, int Tin, int C, const uint8_t* mask_in,
                     float* y, int64_t ldy, uint8_t* mask_out, void* stream);

/* ---- mask head, models/predictor.py:103-104,110-111 --------------------------------------
 * seg[b,q,t] = sum_c emb[b,q,c] * feat[b,t,c]; -10 where out_mask[b,t] == 0.
 * emb: (B*Q) x Dp rows; feat: (B*T) x Dp rows; seg: (B, Q, T) contiguous. */
int vrd_mask_head(const float* emb, int64_t ld_emb, const float* feat, int64_t ld_feat,
                  const uint8_t* out_mask, int B, int Q, int T, int Dp, float fill,
                  float* seg, void* stream);

/* ---- eval post-processing, models/maskvrd.py:247-309 -------------------------------------
 * Per pair p and query q: softmax over classes, top-k of classes 1.., and the [first,last]
 * frame with sigmoid(mask) > 0.5 inside the valid length.  Outputs, per (p, q):
 *   top_score[p,q,k], top_cat[p,q,k] (class id, already +1), seg_first[p,q], seg_last[p,q]
 *   (-1 when the query's mask is empty).
 * logits: (P, Q, K1); masks: (P, Q, T); valid_len: (P) int32. */
int vrd_postprocess(const float* logits, const float* masks, const int32_t* valid_len,
                    int P, int Q, int K1, int T, int topk,
                    float* top_score, int32_t* top_cat, int32_t* seg_first, int32_t* seg_last,
                    void* stream);

/* ====================================================================================================================
 * Backward kernels (training step: the reference differentiates its ATen graph with autograd, train.py:186;
 * models/maskvrd.py:168-198).  All f32 rows (no pair rows).  Parameter gradients are ACCUMULATED (+=) into buffers the
 * caller zeroes; where several workgroups add to one element the order (last bits) varies between runs.
 * The forward of a training step runs the same forward kernels, less fused, so that every op has saved inputs:
 *   y = vrd_gemm(x, W, b, row_mask)                       dense conv, mask only
 *   u = vrd_activation(y)                                        GELU / ReLU
 *   out = vrd_rowcol_scale(u, scale, row_scale, res...)     AffineDropPath scale * per-sample keep factor + residuals
 * ==================================================================================================================== */

/* Weight gradient of vrd_gemm's convolution (models/blocks.py:99 under autograd):
 *   dW[n, tap*Cin + ci] += sum_r G[r, n] * row_mask[r] * X[r + tap - taps/2, ci]      (rows outside r's length-T sequence: 0)
 * dW is (N, taps*Cin) tap-major like vrd_gemm's W.  f32 MFMA, exact f32 products. */
int vrd_gemm_wgrad(const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* row_mask, int64_t M, int N, int Cin,
                   int taps, int T, float* dW, void* stream);

/* The same gradient in the split precision of the forward path (bf16x3 mode): every product as g_lo x_hi + g_hi x_lo + g_hi x_hi
 * on v_mfma_f32_32x32x16_bf16, f32 accumulate (~2^-17 relative product error, ~5x the rate of the exact-f32 MFMA).  Same
 * arguments and accumulation rules as vrd_gemm_wgrad; dbias (nullable, N floats) additionally receives the bias gradient
 * dbias[n] += sum_r G[r, n] * row_mask[r] in the same pass (exact f32 sums).
 * The rows are cut into chunks that fill the chip, one workgroup per (chunk, 128 x 128 tile of dW).  `scratch` (nullable; 16-byte
 * aligned, `scratch_floats` floats, never read before it is written) takes the chunks' partial tiles, which a second launch sums
 * into dW in chunk order: 4 * CUs * 16,384 + N * taps * Cin floats always suffice.  Without it, or when it is too small, the
 * partial tiles are added to dW with float atomics (slower: L2 retires about one float atomic per clock and channel; and the
 * order of the additions then varies from run to run).
 * g_scale (nullable): device pointer to {2^e, 2^-e} from vrd_absmax_scale(G): the products are then formed on f16 planes --
 * G * 2^e, X * 2^VRD_F16_ACT_EXP -- at ~2^-22 relative error (the f16x3 mode's backward); NULL: bf16 planes, ~2^-17. */
int vrd_gemm_wgrad_x3(const float* G, int64_t ldg, const float* X, int64_t ldx, const uint8_t* row_mask, int64_t M, int N, int Cin,
                      int taps, int T, float* dW, float* dbias, float* scratch, int64_t scratch_floats, const float* g_scale,
                      void* stream);

/* scale[0] = 2^e, scale[1] = 2^-e with e such that max |x| * 2^e lies in [2^13, 2^14) over the (rows x cols) matrix x (e = 0 for
 * an all-zero or non-finite matrix; |e| <= 100): the power-of-two factor that puts a tensor of unknown range -- a gradient -- into
 * the f16 planes of VRD_PAIR_F16 with headroom.  One launch; `scale` is VRD_ABSMAX_SCALE_FLOATS floats of device memory that the
 * caller zeroed once: element 3 is the workgroup ticket, which the last workgroup leaves zeroed again, elements 4.. take the
 * workgroups' partial maxima (at most one workgroup per CU); readers use elements 0 and 1 only. */
#define VRD_ABSMAX_SCALE_FLOATS 516
int vrd_absmax_scale(const float* x, int64_t ldx, int64_t rows, int cols, float* scale, void* stream);

/* out[c] += sum_r a[r, c] * (b ? b[brow(r), c * b_cstride + b_coffset] : 1) * (row_mask ? row_mask[r] : 1) * (row_scale ?
 * row_scale[r] : 1), brow(r = s*T + t) = s * (b_rstride*T) + b_rstride*t + shift when that stays inside the sequence
 * (else the row contributes 0).  Bias gradients (b = NULL), AffineDropPath scale gradients (b = the branch value),
 * depthwise-conv weight gradients (b = conv input, shift = k - ksize/2, b_rstride = stride, b_cstride = inputs per group).
 * `scratch` (nullable, 16-byte aligned; 1,024 * C floats always suffice) as in vrd_layernorm_bwd, used when b is NULL or lies on the
 * rows of a (b_cstride = b_rstride = 1, no offset, no shift) and the rows are float4-aligned: the row blocks' partial sums, added
 * up by a second launch instead of one float atomic per column and workgroup. */
int vrd_colsum(const float* a, int64_t lda, const float* b, int64_t ldb, int b_cstride, int b_coffset, int b_rstride, int shift,
               int T, const uint8_t* row_mask, const float* row_scale, int64_t rows, int C, float* out, float* scratch,
               int64_t scratch_floats, void* stream);

/* Weight and bias gradient of a depthwise MaskedConv1D (models/blocks.py:91-113 under autograd; k = 1 / 3, `stride`, group_in
 * = 1 or 2 inputs per group) in one pass over dD (rows x C, rows = B * T output rows):
 *   dw[(c * group_in + g) * ksize + kk] += sum_r dD[r, c] * row_mask[r] * x[in_row(r, kk), c * group_in + g]   (the Conv1d weight's own
 *   (C, group_in, ksize) layout),  dbias[c] += sum_r dD[r, c] * row_mask[r]
 * in_row(r = s*T + t, kk) = s * stride*T + stride*t + kk - ksize/2 where that stays inside sequence s; dbias may be NULL.
 * `scratch` (nullable; 1,024 * 4 C floats always suffice): as in vrd_colsum, for ksize 3, group_in 1 and float4-aligned rows. */
int vrd_dwconv_wgrad(const float* dD, int64_t lddd, const float* x, int64_t ldx, int ksize, int stride, int group_in, int T,
                     const uint8_t* row_mask, int64_t rows, int C, float* dw, float* dbias, float* scratch, int64_t scratch_floats,
                     void* stream);

/* out[r,c] = v[r,c] * col_scale[c] * row_scale[r] * row_mask[r] + res[r,c] * (res_masked ? row_mask[r] : 1) + res2[r,c]
 * (every factor / term optional).  Training form of the affine drop-path residual: models/blocks.py:1074-1076 with
 * AffineDropPath (:1148) = scale[c] * keep[b] / keep_prob (drop_path, :1107-1120), local_transformer.py:815,829,833;
 * also its gradient w.r.t. v (res = NULL) and the row masking of an upstream gradient. */
int vrd_rowcol_scale(const float* v, int64_t ldv, int64_t rows, int C, const float* col_scale, const float* row_scale,
                     const uint8_t* row_mask, const float* res, int64_t ldres, int res_masked, const float* res2, int64_t ldres2,
                     float* out, int64_t ldo, void* stream);

/* dy == NULL: out = act(x) (VRD_ACT_RELU / VRD_ACT_GELU, erf form: models/blocks.py:59,1056);  else out = dy * act'(x). */
int vrd_activation(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int C, int act, float* out, int64_t ldo,
            void* stream);

/* Channel LayerNorm backward (models/blocks.py:143-158 under autograd), C in {256, 512}; with relu != 0 the forward was
 * ReLU(LN(x)).  dx written; dgamma / dbeta (C floats each) accumulated.  `scratch` (nullable, 16-byte aligned;
 * ceil(rows / 32) * 2 C floats always suffice, ~520 * 2 C up to 131,072 rows) takes the workgroups' partial column sums, which a second launch adds up; without it every workgroup
 * ends in one float atomic per channel, and atomics on one address are retired one after the other. */
int vrd_layernorm_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t rows, int C, const float* gamma,
                      const float* beta, int relu, float* dx, int64_t lddx, float* dgamma, float* dbeta, float* scratch,
                      int64_t scratch_floats, void* stream);

/* Input gradient of the depthwise convolution of vrd_dwconv_ln (without its LayerNorms: in a training step those run as
 * separate vrd_layernorm calls): dD[o] = gradient w.r.t. the masked conv output of set o, (B*Tin/stride, C) rows;
 * w[o] = the (C, group_in, ksize) Conv1d weight.  dx (B*Tin, C*group_in) written; with dx_up also the gradient of the
 * nearest-x2-upsampled addend x_up (B*Tin/2 rows): dx_up[b, t] = dx[b, 2t] + dx[b, 2t+1] (models/fpns.py:252). */
typedef struct {
    const float* dD[3];  int64_t lddd[3];
    const float* w[3];
    int32_t n_out, B, Tin, C, ksize, stride, group_in;
    const uint8_t* mask_out;
    float* dx;  int64_t lddx;
    float* dx_up;  int64_t lddx_up;
} vrd_dwconv_bwd_args;
int vrd_dwconv_bwd(const vrd_dwconv_bwd_args* a, void* stream);

/* Banded attention backward (vrd_local_attn; models/blocks.py:950-986): dq, dk, dv (leading dimension ldd) from the
 * forward inputs and dO.  scratch: 2 * B*T * n_head * (2*half_win+1) floats; on return its second half holds dS
 * [B*T][n_head][2*half_win+1], the gradient w.r.t. the biased scores: d rel_pe is its sum over the rows. */
int vrd_local_attn_bwd(const float* q, const float* k, const float* v, int64_t ld, const float* dO, int64_t lddo,
                       const uint8_t* mask, const float* rel_pe, int B, int T, int C, int n_head, int half_win,
                       float* dq, float* dk, float* dv,
                       int64_t ldd, float* scratch, void* stream);

/* Global attention backward, first half (vrd_attention; models/local_transformer.py:44-63,163-183): the probabilities
 * P (B, n_head, Tq, Tk) and the gradient dS w.r.t. the scaled scores.  Then dq = head_dim^-0.5 * dS K, dk = head_dim^-0.5
 * * dS^T Q, dv = P^T dO are three vrd_bmm calls.  Tk <= 1024, head_dim <= 128. */
int vrd_attn_bwd_probs(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* dO, int64_t lddo,
                       const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, int head_dim, float* P, float* dS, void* stream);

/* Second half of the same on matrices (long sequences; round 3): with P = head_dim^-0.5 * Q K^T and dS = dO V^T already formed by
 * two vrd_bmm products (both (B, n_head, Tq, Tk)), turn them in place into P = softmax_j(P | kv_mask) (masked keys: 0) and
 * dS = P * (dS - sum_j P dS).  Tk <= 1024. */
int vrd_attn_bwd_softmax(float* P, float* dS, const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, void* stream);

/* Global attention backward without the (B, n_head, Tq, Tk) matrices (round 4): dq, dk, dv from the forward's inputs, its output
 * `out` and dO, the scores recomputed tile by tile in the bf16 split of the other backward GEMMs (two kernels: dq -- which also
 * leaves every query's log-sum-exp and sum_d dO O in `scratch` -- then dk / dv).  head_dim 64; q / out / dO / dq rows of
 * leading dimension ldq / ldo / ldo / ldq, k / v / dk / dv of ldkv; scratch: 2 * B * n_head * Tq floats.  The five-product form
 * above stays for other head sizes and for the exact-f32 mode.
 * o_scale / v_scale (both or neither): vrd_absmax_scale's outputs for dO and v: the products are then formed on f16 planes (the
 * f16x3 mode) -- q, k, v, P at 2^VRD_F16_ACT_EXP, dO at o_scale[0], dS at o_scale[0] * v_scale[0] * 2^-17, which bounds it below
 * 2^15 whatever the scores are -- at ~2^-22 relative error; NULL: bf16 planes, ~2^-17. */
int vrd_attention_bwd(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const float* out, const float* dO,
                      int64_t ldo, const uint8_t* kv_mask, int B, int Tq, int Tk, int n_head, int head_dim, float* dq, float* dk,
                      float* dv, const float* lse /* (B, n_head, Tq) from vrd_attention_rows, or NULL: recomputed */, float* scratch,
                      const float* o_scale, const float* v_scale, void* stream);
/* The forward of that pair for a training step: vrd_attention on f32 rows in split precision (fmt: VRD_PAIR_BF16 / VRD_PAIR_F16 --
 * the operands are split while they are staged, no pair rows), f32 rows out, plus every query's log-sum-exp of the scaled
 * scores, lse (B, n_head, Tq), which the backward then does not recompute.  head_dim 64. */
int vrd_attention_rows(const float* q, int64_t ldq, const float* k, const float* v, int64_t ldkv, const uint8_t* kv_mask, int B, int Tq,
                       int Tk, int n_head, int head_dim, int fmt, float* out, int64_t ldo, float* lse, void* stream);

/* Strided batched matmul, f32: C[z][i][n] (= or +=) alpha * sum_k A[z][i][k] * B[z][k][n], z = (z0 < Z0, z1 < Z1); every
 * operand is addressed by (stride of z0, stride of z1, stride of its row index, stride of its column index) in floats.
 * Products of at least 32 x 32 x 16 run as 64 x 64 tiles on the matrix cores (exact f32 MFMA, k ascending per output); smaller
 * ones one thread per output (fastest when B and C are contiguous along n). */
typedef struct {
    const float* A;  int64_t a_z0, a_z1, a_row, a_col;
    const float* B;  int64_t b_z0, b_z1, b_row, b_col;
    float* C;  int64_t c_z0, c_z1, c_row, c_col;
    int32_t Z0, Z1, M, N, K;
    float alpha;
    int32_t accumulate;
} vrd_bmm_args;
int vrd_bmm(const vrd_bmm_args* a, void* stream);

/* MaxPool1d(3,2,1)(x) * mask[::2] backward (vrd_maxpool_mask; models/blocks.py:1040-1046,1074): the gradient goes to
 * the first maximum of each window (ATen's rule). */
int vrd_maxpool_bwd(const float* x, int64_t ldx, const float* dy, int64_t lddy, int B, int Tin, int C, const uint8_t* mask_in, float* dx,
                    int64_t lddx, void* stream);

/* ---- training tail on the device ----------------------------------------------------------------------
 * Hungarian assignment of the matcher (models/maskvrd.py:484-492: `.cpu()` + scipy linear_sum_assignment per pair).
 * cost: (sum N, Q) rows = relations (leading dimension ld), row r of pair p = first[p] + r, count[p] = N_p <= Q <= 16.
 * query_of[first[p] + r] = the query relation r of pair p is assigned to (the minimum-cost assignment; pairs with
 * count 0 are skipped).  Double-precision potentials; the result equals scipy's whenever the optimum is unique. */
int vrd_assign(const float* cost, int64_t ld, const int32_t* first, const int32_t* count, int P, int Q, int32_t* query_of, void* stream);

/* EMA of a whole state dict in one launch (utils/train_utils.py:21-29): for every tensor t and element i,
 * ema[t][i] = decay * ema[t][i] + one_minus_decay * model[t][i], both products and the sum rounded to f32 (the reference's
 * tensor expression, bit for bit).  ema / model: device arrays of device pointers; numel: elements per tensor; the launch
 * is cut into chunks of 4096 elements: chunk c works on tensor chunk_tensor[c], elements chunk_index[c]*4096 ... */
int vrd_ema_update(float* const* ema, const float* const* model, const int64_t* numel, const int32_t* chunk_tensor,
                   const int32_t* chunk_index, int n_chunks, float decay, float one_minus_decay, void* stream);

/* ---- the training criterion for all decoder layers of a step --------------------------------------------
 * Replaces the tensor code of models/maskvrd.py:417-496 (bipartite_match: the three cost matrices), :498-588 (loss_labels,
 * loss_masks on the final head and the auxiliary heads) and models/losses.py:4-354 (masked focal / dice, plain and fuzzy
 * targets).  Layer l's predictions: logits[l] (B, Q, K1), masks[l] (B, Q, T), contiguous f32.  Ground truth of the batch:
 * G relations; owner[g] = pair of relation g; tgt_ids[g] its predicate class; tgt_masks (G, T) 0/1; out_valid (B, T) the
 * pairs' valid frames; segs (G, 2) [start, end) frames -- non-null selects the fuzzy targets of losses.py:214-227 with
 * scale_range.  alpha < 0 switches the focal weighting off, like the reference's. */
typedef struct {
    const float* logits[4];
    const float* masks[4];
    int32_t n_layers, B, Q, K1, T, G;
    const uint8_t* out_valid;
    const int64_t* tgt_ids;
    const float* tgt_masks;
    const int32_t* owner;
    const int32_t* segs;
    float scale_range, alpha, gamma;
    float w_class, w_mask, w_dice;            /* cost weights (vrd_criterion_costs only) */
} vrd_criterion_args;

typedef struct {
    float* logits[4];                         /* (B, Q, K1) each, every element written */
    float* masks[4];                          /* (B, Q, T) each, every element written (0 on unmatched rows / padded frames) */
} vrd_criterion_grads;

/* cost (n_layers, G, Q): w_class * (-log softmax(logits_l[owner[g], q])[tgt_ids[g]]) + w_mask * focal cost + w_dice * dice cost
 * of giving relation g to query q of its own pair (entry [owner[g]*Q + q, g] of the reference's matrices, maskvrd.py:447-481).
 * Feed it to vrd_assign with the pair tables repeated per layer. */
int vrd_criterion_costs(const vrd_criterion_args* a, float* cost, void* stream);
/* out (n_layers, 4): [class-weighted cross-entropy over all (pair, query) rows (target = the matched relation's class, else 0;
 * F.cross_entropy(..., weight=class_weight)), sum_g focal_g / num_masks, sum_g dice_g / num_masks, sum of the class weights
 * (the backward's normaliser)] for query_of (n_layers, G) = the assignment (entries < 0 are read as query 0).  Sums run in a
 * fixed order: the result does not depend on scheduling. */
int vrd_criterion_losses(const vrd_criterion_args* a, const int32_t* query_of, const float* class_weight, float num_masks, float* out,
                         void* stream);
/* Gradients of sum_l (gout[l][0] * class_l + gout[l][1] * focal_l + gout[l][2] * dice_l) with respect to every layer's logits and
 * masks; gout (n_layers, 3), fwd_out = vrd_criterion_losses' output for the same arguments. */
int vrd_criterion_backward(const vrd_criterion_args* a, const int32_t* query_of, const float* class_weight, float num_masks,
                           const float* fwd_out, const float* gout, const vrd_criterion_grads* grads, void* stream);

/* The split-precision GEMMs' weight operand (vrd_gemm_args.W_split) from an f32 weight: logical matrix
 * W'[r][tap*Q + q] = src[r*sr + tap*st + q*sq]  (R rows, K = taps*Q columns, K % 32 == 0; strides in floats, may be negative)
 * -> out (R, K/32, 2, 32) 16-bit: per block of 32 columns [32 x hi | 32 x lo].
 * fmt = VRD_PAIR_BF16: hi = bf16(w), lo = bf16(w - hi), one launch; scale unused (may be NULL).
 * fmt = VRD_PAIR_F16: hi = f16(y), lo = f16(y - hi) of y = w * 2^e_w with e_w chosen on the device so that max |w| * 2^e_w lies in
 *   [2^14, 2^15) (e_w in [-113, 100]; an all-zero weight: e_w = 0): three launches (clear, max, split), no host synchronisation.
 *   scale: 4 device floats written: [0] = 2^-(e_w + VRD_F16_ACT_EXP) (vrd_gemm_args.w_scale), [1] = 2^e_w, [2] scratch, [3] unused.
 * Forward operand of a Conv1d weight (N, Cin, k): R = N, Q = Cin, sr = Cin*k, st = 1, sq = k.  Operand of its input-gradient
 * GEMM (autograd of models/blocks.py:91-113; k = 3: taps flipped): R = Cin, Q = N, src = w + (k - 1), sr = k, st = -1,
 * sq = Cin*k.  In a training step every weight changes every step, so both are rebuilt per step: as tensor expressions that
 * was ~11 elementwise launches per weight. */
int vrd_split_weight(const float* src, int R, int Q, int taps, int64_t sr, int64_t st, int64_t sq, uint16_t* out, int fmt, float* scale,
                     void* stream);

/* The same for many weights in ONE launch (a training step re-splits every conv weight, forward and input-gradient form, after
 * each optimiser update).  jobs: DEVICE array; chunk c of the launch is one 32 x 32 tile of job chunk_job[c]: with KB = taps*Q/32
 * blocks per row, rows 32 * (chunk_index[c] / KB) .. + 31 and K block chunk_index[c] % KB (both device int32 arrays of n_chunks
 * entries; a job is covered by ceil(R / 32) * KB chunks). */
typedef struct {
    const float* src;      /* element (r, tap, q) at src[r*sr + tap*st + q*sq] */
    uint16_t* out;         /* (R, taps*Q/32, 2, 32) 16-bit */
    int32_t R, Q, taps, fmt;   /* fmt: enum vrd_pair_format; 0 is read as VRD_PAIR_BF16 */
    int64_t sr, st, sq;
    float* scale;          /* VRD_PAIR_F16: the job's 4 scale floats (see vrd_split_weight); else unused */
} vrd_split_job;
int vrd_split_weights(const vrd_split_job* jobs, int n_jobs, const int32_t* chunk_job, const int32_t* chunk_index, int n_chunks,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VRDONE_HIP_H */
