/*
 * renderih_amd.h -- C ABI of librenderih_amd.so (MI355X / gfx950 hot-path kernels).
 *
 * The reference (adwardlee/RenderIH) has no native boundary on this path: its hot path is
 * torch operators called from Python (models/encoder.py, models/model_attn/{gcn,self_attn,img_attn,inter_attn,DualGraph}.py, models/decoder.py,
 * models/manolayer.py), which dispatch to cuDNN/cuBLAS.  This header is the boundary a maintainer
 * binds instead (ctypes stub in INTEGRATION.md): each entry point names the reference op group
 * (file:line) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where said) owned by the caller
 *     (PyTorch's caching allocator); the library never allocates or frees device memory;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it, no host sync;
 *   - return value: 0 = ok, otherwise a hipError_t (>0) or a negative argument-check code;
 *     no C++ exception crosses the ABI;
 *   - activations are NHWC ("pixels x channels", channel fastest); a matrix is the H=W=1 case.
 */
#ifndef RENDERIH_AMD_H
#define RENDERIH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIH_OK 0
#define RIH_EINVAL (-1)

/* ------------------------------------------------------------------------------------------------
 * rih_gemm: C = epilogue(alpha * A (*) B), fp32 in / fp32 out / fp32 accumulate on one of three MFMA engines (field `engine`):
 * 0 = native v_mfma_f32_32x32x2_f32, 1 = six bf16 products (v_mfma_f32_32x32x16_bf16), 2 (default where both operands have bounds) =
 * three fp16 products on a scaled two-term split (v_mfma_f32_32x32x16_f16).
 *
 * One kernel family covers every dense contraction of the path:
 *   conv2d forward / data-gradient (A gathered im2col-style from an NHWC tensor)
 *       replaces torch conv2d in models/encoder.py:52,112-116,172 + torchvision resnet50,
 *       models/model_attn/img_attn.py:62 (patch conv)
 *   conv2d / linear weight-gradient (A = im2col^T, split-K over pixels)
 *   nn.Linear forward/backward          models/model_attn/{gcn.py:66,self_attn.py:66-76,inter_attn.py:85-107}
 *   attention QK^T, PV and their gradients (two-level batch strides = batch x head)
 *       models/model_attn/self_attn.py:70-74, inter_attn.py:93-107
 *
 * a_mode 0: A(m,k): m = (img,ho,wo) output pixel, k = (kh,kw,ci); element =
 *           X[img][(ho*strideA - padH + kh)/upS][(wo*strideA - padW + kw)/upS][ci]  (0 when out of range or
 *           not divisible by upS; upS>1 expresses the data-gradient of a strided conv).  K-contiguous.
 * a_mode 1: A(m,k): m = (kh,kw,ci), k = (img,ho,wo) -- the transpose gather, M-contiguous (weight grad).
 * b_mode 0: B(k,n) = Bp[k*ldb + n]   (row-major [K][N])
 * b_mode 1: B(k,n) = Bp[n*ldb + k]   ([N][K], e.g. an nn.Linear / 1x1-conv weight as stored)
 * Epilogue (splitk==1): C[m*ldc+n] = act(alpha*acc + bias[n] + R[m*ldr+n]); with splitk>1 raw partial sums
 *           go to C + split*sCsplit and rih_splitk_reduce finishes.
 */
typedef struct rih_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    const float* bias;   /* [N] or NULL */
    const float* R;      /* residual [M][ldr] or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr;
    int32_t a_mode, b_mode;
    /* batching: grid.z = nb1 * nb2 * splitk */
    int32_t nb1, nb2;
    int64_t sA1, sA2, sB1, sB2, sC1, sC2;
    int32_t splitk;      /* >= 1 */
    int32_t kchunk;      /* K range per split, multiple of 32 */
    int64_t sCsplit;
    float alpha;
    int32_t relu;
    /* im2col geometry of A (a plain matrix is H=W=Ho=Wo=KH=KW=1, strideA=upS=1, pad=0, Cin=K or M) */
    int32_t H, W, Cin, Ho, Wo, KH, KW, strideA, upS, padH, padW;
    int32_t tile;        /* 0: 128x128, 1: 128x64, 2: 64x64, 3: 128x32 (engine 0 only),
                            4: 256x128 software-pipelined (engine 1 only; operands must satisfy the split fast-path
                               preconditions documented at gemm_split_kernel, else RIH_EINVAL) */
    int32_t engine;      /* 0: native f32 MFMA (v_mfma_f32_32x32x2_f32, 157 TF peak);
                            1: fp32 emulated on the bf16 MFMA pipe: each operand split into three bf16 terms
                               (hi+mid+lo, 24 significand bits), six v_mfma_f32_32x32x16_bf16 products, fp32
                               accumulate -- same accuracy as engine 0 (error <= ~2^-23 relative per product),
                               2.5 PF / 6 = 417 TF peak.  Tiles 0,1,2 only (tile 3 always runs engine 0).
                            2: fp32 emulated on the fp16 MFMA pipe with THREE products: each operand is scaled by a power
                               of two (from amax_a / amax_b below) and split into two fp16 terms hi + 2^-11 lo (22-23
                               significand bits), hi*hi and hi*lo + lo*hi accumulate in two fp32 accumulators
                               (v_mfma_f32_32x32x16_f16), combined and un-scaled in the epilogue; 2.5 PF / 3 = 833 TF
                               peak.  Exists on the fast path of tiles 0,1,2 (a_mode / b_mode 0 or 1); a descriptor that asks
                               for it elsewhere runs engine 1 (rih_gemm_engine tells which). */
    /* Strided output rows (cS > 1; a_mode 0, splitk 1, no residual): GEMM row m = (img, i, j) over (Ho, Wo) is stored
     * to pixel (img, i*cS + cOH, j*cS + cOW) of a [*, cH, cW, ldc] tensor.  Used to compute the data gradient of a
     * stride-s convolution as s*s dense sub-convolutions, one per output parity class (each with the kernel taps
     * kh = kh0 + s*t that can reach that class, packed by rih_pack_conv_weight_sub), instead of one convolution over
     * a zero-stuffed gradient (upS) in which (s*s-1)/(s*s) of the products are structural zeros.  cS <= 1: off. */
    int32_t cS, cOH, cOW, cH, cW;
    /* a_mode 1 only: A(ones_row, k) = 1 for every k < K, whatever memory holds (rows > ones_row up to M are junk).
     * Appending this row to a weight-gradient GEMM makes C(ones_row, n) = sum_k B(k, n), the bias gradient, so that
     * no separate column-sum pass over dy is needed (rih_splitk_reduce_bias picks the row up).  0 = off. */
    int32_t ones_row;
    int32_t reserved0;   /* keeps the 64-bit fields below aligned; set 0 */
    /* Per-nb1-slice strides (in floats) of bias and R.  With nb1 = 2 and sB1 / sBias1 = the distance between the left-
     * and the right-hand parameter tensors, the two per-hand nn.Linear layers of the decoder
     * (models/model_attn/DualGraph.py:83-89 runs them one after the other) execute as ONE launch on activations
     * stacked [2][rows][K].  0 = bias / R shared by all slices. */
    int64_t sBias1, sR1;
    /* Optional statistics epilogue (split engine's fast path, a_mode 0, no split-K, no batch): per block of
     * rih_gemm_stats_rows(desc) GEMM rows the per-column mean and centred sum of squares of the stored values,
     * stats[ceil(M / rows)][2][N] -- the training statistics of the BatchNorm behind a convolution without a pass over its
     * output (rih_bn_stats_from_blocks merges them).  NULL = off.  rih_gemm returns RIH_EINVAL when stats is set and the
     * descriptor does not take that path; rih_gemm_stats_rows (stats field ignored) returns 0 for such a descriptor. */
    float* stats;
    /* Dropout in the epilogue (ABI 12; plain row-major GEMMs = nn.Linear on the split engine's fast path, no split-K, no stats):
     * C = dropout(act(alpha A B + bias)) + R with the mask stream of rih_add_dropout over the output -- element e (offset from C
     * in floats; the output is expected contiguous from C) is kept and scaled by 1 / (1 - drop_p) iff
     * rih_hash(drop_seed + *drop_seed_dev, e) >= drop_p * 2^32 -- so that rih_gemm with drop_p equals rih_gemm followed by
     * rih_add_dropout(R, ., drop_p, drop_seed, drop_seed_dev) bit for bit and rih_dropout_bwd re-draws the mask.  relu and R
     * together are refused.  drop_p = 0: off.  rih_gemm_dropout_ok(desc) tells in advance whether a descriptor takes that path
     * (rih_gemm returns RIH_EINVAL for drop_p != 0 otherwise). */
    float drop_p;
    int32_t reserved1;
    uint64_t drop_seed;
    const uint64_t* drop_seed_dev;
    /* Engine 2 only (ABI 13): DEVICE pointers to a BOUND BLOCK each (RIH_BOUND_FLOATS floats, see rih_absmax): an upper bound
     * of max|A| resp. max|B| over everything the launch reads (all batch slices).  The kernel derives the power-of-two operand
     * scale from it (s * bound in [2^14, 2^15)), so the bound is read at run time -- a step captured in a hipGraph follows the
     * data.  Any upper bound is correct: a bound 2^k too large costs k of the ~29 binades below the maximum that keep full
     * precision; a bound that is too SMALL can overflow fp16 (inf / NaN in the result).  Written by rih_absmax or by the
     * producing kernel (rih_bn_apply, rih_bn_bwd: `amax` arguments).  NULL = the operand is known to lie inside
     * [-2^15, 2^15] (scale 1). */
    const float* amax_a;
    const float* amax_b;
    /* Segmented A (ABI 13): A = [A | a_seg[0] | a_seg[1] | a_seg[2]] along K -- columns [0, k_seg[0]) come from A (pitch lda),
     * columns [k_seg[i], k_seg[i+1]) from a_seg[i] (pitch lda_seg[i]; the last piece ends at K); unused pieces NULL.  The
     * channel concatenation in front of a 1x1 convolution (models/encoder.py:165-173: `torch.cat((hms_fmaps[i], dp_fmaps[i],
     * img_fmaps[i]), dim=1)` -> conv1x1) is then read in place instead of being materialised.  Plain row-major pieces only
     * (a_mode 0 without im2col geometry), b_mode 1, one batch slice, no split-K; every k_seg a multiple of 32, pieces 16-byte
     * aligned; engines 1 and 2 on the fast path of tiles 0..2 (statistics epilogue available), RIH_EINVAL otherwise. */
    const float* a_seg[3];
    int32_t lda_seg[3];
    int32_t k_seg[3];
} rih_gemm_desc;
int rih_gemm_stats_rows(const rih_gemm_desc* d);
int rih_gemm_dropout_ok(const rih_gemm_desc* d);
/* The engine rih_gemm would run `d` on (0 / 1 / 2), or -1 for an invalid descriptor. */
int rih_gemm_engine(const rih_gemm_desc* d);
/* Bound block: RIH_BOUND_FLOATS floats = 64 partial maxima at a stride of 32 floats (one per 128-byte line; the other floats are
 * unused); THE BOUND IS THE MAXIMUM OF THE 64.  A producer merges its candidates into the lines with atomic maxima (64 lines so
 * that the same-address atomics of thousands of workgroups do not queue up behind one word), a consumer (rih_gemm engine 2) reads
 * the 64 words with one vector load per wavefront.  The caller zeroes a block (all 2048 floats, or at least the 64 words) before
 * its first producer runs; several producers may share a block.
 * rih_absmax: block <- max(block, max_i |x[i]|) over n floats (NaNs are ignored); one pass at streaming rate.
 * rih_absmax_multi: the same for n tensors in ceil(n / 120) launches (`descs` is HOST memory, read before the call returns):
 * the convolution weights of a training step. */
#define RIH_BOUND_FLOATS 2048
int rih_absmax(const float* x, int64_t n, float* out, void* stream);
typedef struct rih_absmax_desc {
    const float* x;
    float* out;             /* bound block */
    int64_t n;
} rih_absmax_desc;
int rih_absmax_multi(const rih_absmax_desc* descs, int n, void* stream);

int rih_gemm(const rih_gemm_desc* d, void* stream);

/* Grouped launch: n independent rih_gemm problems that take the SAME kernel variant run as ONE launch -- e.g. all weight
 * gradients of the decoder's nn.Linear layers of a backward stage (models/model_attn/gcn.py:99-110, self_attn.py:17-33: 116
 * GEMM launches of ~20 us per ResNet50 step, each a few hundred workgroups with short reductions) -- so that the chip stays
 * full across problems and the dependent-launch floor is paid once.  Because the problems then fill the machine together,
 * the caller can also use far fewer split-K slices per problem than a lone launch needs (less partial-slab traffic).
 *   rih_gemm_multi_variant(d): >= 0 = the variant id of a descriptor that can ride in a grouped launch (split engine's fast
 *     path; weight-gradient GEMMs a_mode 1 / b_mode 0 on 64x64 or 128x128 tiles), -1 otherwise.
 *   rih_gemm_multi_table_bytes / rih_gemm_multi_pack: size of, and fill, the launch table in HOST memory for n descriptors of
 *     one variant (returns the variant, or a negative error code; *total_blocks = grid size).  The table holds the prepared
 *     kernel arguments of every problem and a block -> problem map.
 *   rih_gemm_multi_launch(dev_table, variant, total_blocks, stream): the launch; `dev_table` = the packed table in DEVICE
 *     memory (the caller's stream-ordered copy; it must stay unchanged until the kernel has run).  No allocation, no sync. */
int rih_gemm_multi_variant(const rih_gemm_desc* d);
int64_t rih_gemm_multi_table_bytes(const rih_gemm_desc* descs, int n);
int rih_gemm_multi_pack(const rih_gemm_desc* descs, int n, void* host_table, int32_t* total_blocks);
int rih_gemm_multi_launch(const void* dev_table, int variant, int total_blocks, void* stream);

/* Sum split-K partials P[S][M][N] (M = taps*Cin rows ordered (tap,ci)) into a weight gradient laid out
 * like the parameter: dst[(n*CinValid + ci)*taps + tap]  (OIHW for convs, [out][in] for nn.Linear).
 * Rows with ci >= CinValid (channel padding) are dropped.  accumulate!=0 adds to dst.  One launch, fixed summation
 * order (deterministic). */
int rih_splitk_reduce(float* P, int S, int M, int N, float* dst, int Cin, int taps, int CinValid,
                      int accumulate, void* stream);
/* Same on slabs of Mp >= M rows; with db != NULL (Mp >= M+1) slab row M -- produced by a weight-gradient GEMM whose A
 * operand carries an all-ones row there (rih_gemm_desc.ones_row) -- is summed into the bias gradient db[N]. */
int rih_splitk_reduce_bias(const float* P, int S, int Mp, int M, int N, float* dst, int Cin, int taps, int CinValid,
                           int accumulate, float* db, void* stream);
/* Any number of independent reductions (each as rih_splitk_reduce_bias) in ceil(n/56) launches: the split-K partials of
 * the weight gradients of a whole backward stage are summed at the END of the stage instead of one small launch behind
 * every weight-gradient GEMM (157 launches of ~8 us per ResNet50 step; a dependent kernel costs >= 4.5 us on this
 * platform whatever its size).  `descs` is HOST memory, read before the call returns.  Same fixed summation order. */
typedef struct rih_reduce_desc {
    const float* P;         /* [S][Mp][N] partial slabs */
    float* dst;             /* parameter-layout gradient, see rih_splitk_reduce */
    float* db;              /* optional bias gradient [N] from slab row M, or NULL */
    int32_t S, Mp, M, N, Cin, taps, CinValid, accumulate;
    int32_t CinPitch;       /* 0: dst is the whole parameter [N][CinValid][taps]; > 0: dst points at the first of CinValid columns of a
                               wider parameter [N][CinPitch][taps] (the column slice of one part of a concatenated input, a_seg) */
    int32_t reserved;
} rih_reduce_desc;
int rih_splitk_reduce_multi(const rih_reduce_desc* descs, int n, void* stream);
/* nb independent reductions in one launch: slice b reads P + b*sP and writes dst + b*sDst, db + b*sDb (the split-K
 * partials of an nb1-batched weight-gradient GEMM, e.g. the paired left/right-hand layers). */
int rih_splitk_reduce_bias_batched(const float* P, int S, int Mp, int M, int N, float* dst, int Cin, int taps,
                                   int CinValid, int accumulate, float* db, int nb, int64_t sP, int64_t sDst,
                                   int64_t sDb, void* stream);

/* Training statistics (what rih_bn_stats produces, running buffers included) from the per-row-block (mean, M2) pairs that
 * rih_gemm's statistics epilogue writes (rih_gemm_desc.stats: part[T][2][C], T = ceil(rows / rows_per_block),
 * rows_per_block = rih_gemm_stats_rows(desc)): no pass over the convolution output; one launch. */
int rih_bn_stats_from_blocks(const float* part, int T, int C, int rows, int rows_per_block, float eps, float momentum,
                             float* mean, float* invstd, float* running_mean, float* running_var, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Halo-resident 3x3 convolution, stride 1, padding 1 (csrc/rih_conv3.hip, ABI 15) -- the 3x3 convolutions of the trunk's
 * Bottlenecks and of the aux decoders (torchvision Bottleneck.conv2 via models/encoder.py:81-83; models/encoder.py:44-54) and
 * their data gradients, on engine 2's arithmetic (fp32 = three fp16 MFMA products, see rih_gemm_desc.engine).  A workgroup owns
 * an 8 x 32 pixel patch and 128 (64) output channels, loads and converts each 32-channel chunk of the (8+2) x (32+2) input
 * halo ONCE and runs the nine taps on shifted windows of that LDS image; the weights arrive as two pre-split fp16 planes
 * ("H2": dst[n][k / 8][plane][8 halves], k = (tap, channel), written once per training step by rih_h2_multi) through LDS-DMA.
 *   y[img][i][j][n] = act(sum_{kh,kw,c} x[img][i+kh-1][j+kw-1][c] * W(n, (kh*3+kw)*C + c) (+ r[img][i][j][n]))        (zero padding)
 * Forward: W from rih_h2_desc.for_dgrad 0 (n = co).  Data gradient: x = dy, C = Cout, N = CinPad, for_dgrad 1 (flipped taps).
 * Preconditions (rih_conv3x3_ok returns 1): C % 32 == 0, N % 32 == 0, (H % 8 == 0 and W % 32 == 0: patches of 8 x 32 pixels) or
 * (H % 16 == 0 and W % 16 == 0: 16 x 16), Kpad == 9 * C, 16-byte aligned x / w_h2 / y / stats, ldx / ldy % 4 == 0, one image
 * below 2 GiB.  amax_x / amax_w: bound blocks (rih_absmax) -- amax_w must be the block the H2 planes were scaled with.  stats
 * (optional): [imgs * H * W / rows][2][N] (mean, M2) per block of rows = rih_conv3x3_stats_rows(desc) output rows (64; 32 when
 * N is not a multiple of 64), the format of rih_gemm_desc.stats (a block is a piece of a patch, every block full).
 * Not bit-identical to rih_gemm on the same convolution: the reduction runs chunk-major, (c / 32, tap, c % 32). */
typedef struct rih_conv3_desc {
    const float* x;         /* NHWC [imgs][H][W][ldx] */
    const void* w_h2;       /* [N][Kpad / 8][2][8] fp16 */
    float* y;               /* NHWC [imgs][H][W][ldy] */
    float* stats;           /* or NULL */
    const float* amax_x;
    const float* amax_w;
    int32_t imgs, H, W, C, N, ldx, ldy, Kpad, relu;
    int32_t ldr;            /* (ABI 19) pitch of r */
    const float* r;         /* (ABI 19) optional residual NHWC [imgs][H][W][ldr], added before the ReLU: the skip gradient that joins the
                             * data gradient of a residual block's first 3x3 convolution (torchvision BasicBlock.conv1; reference
                             * models/hrnet: BasicBlock.forward `out += residual`).  16-byte aligned, ldr >= N, ldr % 4 == 0; not
                             * together with stats; rih_stem takes none.  NULL: none */
} rih_conv3_desc;
int rih_conv3x3_ok(const rih_conv3_desc* d);
int rih_conv3x3_stats_rows(const rih_conv3_desc* d);
int rih_conv3x3(const rih_conv3_desc* d, void* stream);
typedef struct rih_h2_desc {
    const float* w;         /* OIHW parameter */
    void* dst;              /* [N][Kpad / 8][2][8] fp16: N = Cout (for_dgrad 0) or CinPad (1), Kpad >= KH*KW*CinPad (resp. *Cout), % 32 */
    const float* amax;      /* bound block of w */
    int32_t Cout, Cin, KH, KW, CinPad, for_dgrad, Kpad;
} rih_h2_desc;
/* any number of H2 weight operands in ceil(n / 48) launches (`descs` is HOST memory, read before the call returns) */
int rih_h2_multi(const rih_h2_desc* descs, int n, void* stream);

/* Short-reduction streaming GEMM (csrc/rih_conv3.hip `panel_kernel`, ABI 16): C[M][N] = act(A[M][K] W[N][K]^T (+ R)) for the 1x1
 * convolutions with K = 64 or 128 on large maps (torchvision Bottleneck.conv3 of layer1 / layer2 forward, conv1's data
 * gradient) -- HBM streams that the tiled kernels of rih_gemm ran at 2-3 TB/s.  Persistent workgroups (one per CU) keep their
 * column block of the H2 weight planes (rih_h2_multi with KH = KW = 1: forward n = co, k = ci; data gradient for_dgrad 1: n = ci,
 * k = co) in LDS and walk over row tiles with the next tile's rows in flight.  Engine-2 arithmetic (amax_a / amax_w as
 * rih_conv3_desc).  Preconditions (rih_panel_ok returns 1): K in {64, 128}, N % 64 == 0 (% 128 at K = 128), M % 128 == 0,
 * 16-byte aligned a / w_h2 / c / r / stats, lda / ldc / ldr % 4 == 0, M * lda * 4 < 2^31, not stats and r together.
 * stats (optional): [M / rows][2][N] per block of rows = rih_panel_stats_rows(desc) consecutive rows (rih_gemm_desc.stats format). */
typedef struct rih_panel_desc {
    const float* a;         /* [M][lda] */
    const void* w_h2;       /* [N][K / 8][2][8] fp16 */
    float* c;               /* [M][ldc] */
    const float* r;         /* [M][ldr] or NULL */
    float* stats;           /* or NULL */
    const float* amax_a;
    const float* amax_w;
    int32_t M, N, K, lda, ldc, ldr, relu;
} rih_panel_desc;
int rih_panel_ok(const rih_panel_desc* d);
int rih_panel_stats_rows(const rih_panel_desc* d);
int rih_panel(const rih_panel_desc* d, void* stream);
/* Long-K plain-row GEMM (csrc/rih_conv3.hip rows_kernel, ABI 18) -- the 1x1 convolutions with K >= 256 of Bottleneck.conv1 / conv3
 * and their data gradients (torchvision Bottleneck via models/encoder.py:75-89,107-126): the same descriptor and the same
 * arithmetic as rih_panel (engine 2, H2 weight planes), as 512-thread workgroups on 256 x 128 / 128 x 128 / 256 x 64 / 128 x 64 tiles
 * with the weights staged by LDS-DMA and three A stages in LDS.  Preconditions (rih_rows_ok returns 1): K % 32 == 0, K >= 64,
 * N % 64 == 0, M % 128 == 0, pitches % 4 == 0, 16-byte aligned operands, M * lda * 4 < 2^31, not both stats and r.
 * stats (optional): [M / rows][2][N] per block of rows = rih_rows_stats_rows(desc) consecutive rows (64 or 32). */
/* The stem convolution (ABI 18): 7 x 7 / stride 2 / padding 3 over a FOUR-channel NHWC image batch (encoder.resnet.conv1 of
 * torchvision ResNet-50 via models/encoder.py:107-116, on the 3 -> 4 channel padded input of rih_nchw_to_nhwc) on the rows kernel with
 * an im2col loader (one tap of one output pixel = one float4).  Descriptor = rih_conv3_desc: x [imgs][H][W][4] (C = ldx = 4), w_h2 =
 * H2 planes of the OIHW weight with CinPad 4 (rih_h2_desc: KH = KW = 7, Kpad = 224), y [imgs][H/2][W/2][ldy], N = 64, stats per 64
 * rows.  Preconditions (rih_stem_ok returns 1): H, W even, imgs * H/2 * W/2 % 256 == 0, the image batch < 2 GiB, 16-byte aligned. */
int rih_stem_ok(const rih_conv3_desc* d);
int rih_stem(const rih_conv3_desc* d, void* stream);
int rih_rows_ok(const rih_panel_desc* d);
int rih_rows_stats_rows(const rih_panel_desc* d);
int rih_rows(const rih_panel_desc* d, void* stream);

/* Finish a forward split-K GEMM: C[m*ldc+n] = act(alpha * sum_s P[s][m][n] + bias[n] + R[m*ldr+n]). */
int rih_splitk_finish(const float* P, int S, int M, int N, float* C, int ldc, const float* bias, const float* R,
                      int ldr, float alpha, int relu, void* stream);

/* Repack an OIHW conv weight for rih_gemm: dst[((kh*KW+kw)*Ci_pad + ci)*Cout + co] (forward, b_mode 0) or,
 * with for_dgrad!=0, dst[(((KH-1-kh)*KW + (KW-1-kw))*Cout + co)*Ci_pad + ci] (flipped, in/out swapped). */
int rih_pack_conv_weight(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int CinPad,
                         int for_dgrad, void* stream);
/* Any number of weight operands packed in ceil(n/56) launches (every 3x3 / 7x7 / patch convolution of a training step needs
 * its forward and its data-gradient operand once per step: 58 launches of ~6 us otherwise).  mode 0 = rih_pack_conv_weight
 * (for_dgrad 0), mode 1 = rih_pack_conv_weight_sub (the full flipped operand is kh0 = kw0 = 0, step 1, Th = KH, Tw = KW).
 * `descs` is HOST memory, read before the call returns. */
typedef struct rih_pack_desc {
    const float* w;         /* OIHW parameter */
    float* dst;
    int32_t Cout, Cin, KH, KW, CinPad, mode, kh0, kw0, step, Th, Tw, reserved;
} rih_pack_desc;
int rih_pack_conv_weight_multi(const rih_pack_desc* descs, int n, void* stream);
/* Tap subset for one parity class of a strided data gradient:
 * dst[((th*Tw + tw)*Cout + co)*CinPad + ci] = w[co][ci][kh0 + step*(Th-1-th)][kw0 + step*(Tw-1-tw)]. */
int rih_pack_conv_weight_sub(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int CinPad, int kh0,
                             int kw0, int step, int Th, int Tw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Layout, pooling, resampling   (models/encoder.py:107-113 stem, :31 nn.Upsample, :155-158 avgpool)   */
int rih_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream);
int rih_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, void* stream);
int rih_maxpool3x3s2_fwd(const float* x, float* y, int8_t* arg, int N, int H, int W, int C, void* stream);
int rih_maxpool3x3s2_bwd(const float* dy, const int8_t* arg, float* dx, int N, int H, int W, int C, void* stream);
int rih_avgpool_fwd(const float* x, float* y, int N, int HW, int C, void* stream);
int rih_avgpool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream);
int rih_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int rih_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* bilinear x`factor`, align_corners=True: [N,H,W,C] -> [N,factor*H,factor*W,C]
 * (F.interpolate in models/encoder.py:228-230: the HRNet branches are brought to 64x64 with x2 / x4 / x8) */
int rih_upsample_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int factor, void* stream);
int rih_upsample_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int factor, void* stream);
/* nearest x`factor` upsample fused with the running sum of HighResolutionModule's fuse layer
 * (models/model_zoo/hrnet.py:181-183, 229-236): y = add + up(x) (add may be NULL); C % 4 == 0.
 * Backward of the upsample branch: dx = sum of dy over each factor x factor block (the `add` branch gets dy). */
int rih_nearest_up_add_fwd(const float* x, const float* add, float* y, int N, int H, int W, int C, int factor,
                           void* stream);
int rih_nearest_up_bwd(const float* dy, float* dx, int N, int H, int W, int C, int factor, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (NHWC, rows = N*H*W)   (torchvision resnet bn*, models/encoder.py:54, model_zoo/__init__.py:57)
 * ws: caller workspace, >= rih_bn_ws_floats(rows, C) floats.                                           */
int64_t rih_bn_ws_floats(int rows, int C);
/* training statistics: mean[C], invstd[C]; updates running_mean/var (momentum, unbiased var) when non-NULL */
int rih_bn_stats(const float* x, int rows, int C, float eps, float momentum, float* mean, float* invstd,
                 float* running_mean, float* running_var, float* ws, void* stream);
/* eval statistics from running buffers */
int rih_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                      float* invstd, void* stream);
/* y = act((x-mean)*invstd*gamma + beta + residual).  amax (optional, ABI 13): a bound block (rih_absmax) that receives max|y| --
 * the operand bound of the convolution that reads y (rih_gemm_desc.amax_a, engine 2) at no extra pass; zeroed by the caller. */
int rih_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                 const float* residual, float* y, int rows, int C, int relu, uint8_t* relu_mask, float* amax, void* stream);
/* relu_mask (optional, relu != 0): rows*C/4 bytes, byte q = sign pattern of output quad q (bit e set: element 4q+e > 0).
 * The backward then reads one byte per quad instead of the 16 bytes of y (it needs nothing else of y).
 * backward of the above (training statistics): given dy and the ReLU pattern (relu_mask, or the forward output y when
 * relu_mask == NULL), produces dx, dgamma, dbeta and, when dres != NULL, the residual-branch gradient (= masked dy).
 * frozen_stats bit 0: eval-mode backward (statistics are constants).  Bit 1: the BatchNorm's input x is itself a ReLU output
 * (Conv -> ReLU -> BN, models/encoder.py:52-54): dx is additionally gated by x > 0, so the convolution's backward needs no
 * separate ReLU pass over it. */
int rih_bn_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
               const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta, int rows, int C,
               int relu, int frozen_stats, float* ws, const uint8_t* relu_mask, float* amax_dx, void* stream);
/* amax_dx (optional, ABI 13): a bound block that receives max|dx|, as `amax` of rih_bn_apply -- dx is the gradient operand of
 * the producing convolution's data- and weight-gradient GEMMs. */

/* ------------------------------------------------------------------------------------------------
 * Row-wise ops on [rows][D] matrices (decoder)                                                         */
/* y = act(LayerNorm(x (+ x2)) * g + b); saves mean/rstd per row   (nn.LayerNorm eps=1e-6, gcn.py:91-97 ...) */
int rih_layernorm_fwd(const float* x, const float* x2, const float* g, const float* b, float* y, float* mean,
                      float* rstd, int rows, int D, float eps, int relu, void* stream);
/* dx (= gradient wrt x and x2), dg/db accumulated via ws (>= 2*nblk*D floats, nblk = rih_ln_nblk(rows)).
 * dres (may be NULL): [rows][D] added to dx -- the gradient arriving over a skip connection around the norm
 * (x + f(LN(x)), self_attn.py:26-33, 84-85), so that autograd needs no separate accumulation pass. */
int rih_ln_nblk(int rows);
/* rih_layernorm_bwd[_grouped] with dg == db == NULL leaves the parameter gradients as partials in ws ([groups][nblk][2][D]);
 * rih_ln_param_final_multi finishes any number of them (one descriptor per group: ws + g*nblk*2*D, dg + g*D, db + g*D) in
 * ceil(n/100) launches -- a backward stage's 50 LayerNorms cost one finishing launch instead of 50 (>= 4.5 us each).
 * `descs` is HOST memory, read before the call returns. */
typedef struct rih_ln_final_desc {
    const float* ws;
    float* dg;
    float* db;
    int32_t D, nblk;
} rih_ln_final_desc;
int rih_ln_param_final_multi(const rih_ln_final_desc* descs, int n, void* stream);
int rih_layernorm_bwd(const float* dy, const float* x, const float* x2, const float* y, const float* g,
                      const float* mean, const float* rstd, const float* dres, float* dx, float* dg, float* db, int rows,
                      int D, int relu, float* ws, void* stream);
/* `groups` independent LayerNorms in one launch (the left- and the right-hand layer of DualGraph.py:83-89 on
 * activations stacked [groups][rows][D]): group q uses the parameters g + q*sG, b + q*sB (distances in floats between
 * the two layers' parameter tensors, may be negative) and, in the backward, writes dg + q*D, db + q*D; ws >= groups * 2*nblk*D floats. */
int rih_layernorm_fwd_grouped(const float* x, const float* x2, const float* g, const float* b, float* y, float* mean,
                              float* rstd, int groups, int rows, int D, int64_t sG, int64_t sB, float eps, int relu,
                              void* stream);
int rih_layernorm_bwd_grouped(const float* dy, const float* x, const float* x2, const float* y, const float* g,
                              const float* mean, const float* rstd, const float* dres, float* dx, float* dg, float* db,
                              int groups, int rows, int D, int64_t sG, int relu, float* ws, void* stream);
/* softmax over the last dim of [rows][ld] (first `cols` entries), optional inverted dropout with a
 * counter-based hash RNG (csrc/rih_hash.h: element idx of stream `seed` is kept iff mix32(lo32(idx) ^ hi32(idx) * 0x85EBCA6B ^
 * key(seed)) >= p * 2^32): P (pre-dropout probabilities) and Pd (post-dropout, may alias P when p==0).
 * Every dropout entry point takes the stream id of the mask as `seed` plus an optional DEVICE word `seed_dev` (may be
 * NULL) that is added to it on the GPU: a training step captured in a hipGraph advances that word inside the graph
 * and gets a fresh mask at every replay, with no host involvement. */
int rih_softmax_fwd(const float* S, float* P, float* Pd, int64_t rows, int cols, int ld, float drop_p,
                    uint64_t seed, const uint64_t* seed_dev, void* stream);
/* dS = alpha * P * (dPd*mask/(1-p) - sum_j(dPd*mask/(1-p) * P)), written in place over dPd */
int rih_softmax_bwd(const float* P, float* dPd, int64_t rows, int cols, int ld, float drop_p, uint64_t seed,
                    const uint64_t* seed_dev, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / gather                                                                                 */
/* Adam / AdamW step (torch.optim.Adam / AdamW semantics, amsgrad off) over a table of parameter tensors in ONE launch:
 * replaces the optimizer step of core/gcn_trainer.py:127 (torch.optim.Adam: 24 multi-tensor launches of ~37 us per ResNet50
 * step, 4.7x the HBM time of the 7 passes it needs).  `table` (DEVICE memory): one entry per tensor; block b of the launch
 * updates elements [blk_chunk[b]*C, (blk_chunk[b]+1)*C) of tensor blk_tensor[b], C = rih_adam_chunk() (both DEVICE int32
 * arrays of nblocks entries, built once by the caller).  step >= 1 is the 1-based step count of the bias corrections. */
typedef struct rih_adam_entry {
    float* p;           /* parameter, updated in place */
    const float* g;     /* gradient */
    float* m;           /* exp_avg */
    float* v;           /* exp_avg_sq */
    int64_t n;          /* elements */
} rih_adam_entry;
int rih_adam_multi(const rih_adam_entry* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int nblocks, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, int adamw, void* stream);
int rih_adam_chunk(void);

/* y = a + dropout(b) (inverted dropout, p may be 0); b_bcast_rows>0: b is [b_bcast_rows][D] broadcast over the batch */
int rih_add_dropout(const float* a, const float* b, float* y, int64_t n, int D, int b_bcast_rows, float drop_p,
                    uint64_t seed, const uint64_t* seed_dev, void* stream);
/* backward of dropout: dx = dy * mask/(1-p) */
int rih_dropout_bwd(const float* dy, float* dx, int64_t n, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                    void* stream);
/* y = max(x,0);  dx = dy * (y>0) */
int rih_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int rih_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* column sums of [rows][C] (bias gradients); ws >= rih_colsum_ws_floats(rows,C) */
int64_t rih_colsum_ws_floats(int rows, int C);
int rih_colsum(const float* x, int rows, int C, int ldx, float* out, int accumulate, float* ws, void* stream);
/* out[b][i][:] = x[b][idx[i]][:]  (graph_perm gathers, nearest upsample along V, token slicing) */
int rih_gather_rows(const float* x, const int32_t* idx, float* y, int B, int Vin, int Vout, int D, void* stream);
/* dx[b][idx[i]][:] += dy[b][i][:]  with dx zero-initialised by the callee (deterministic, via inverse lists) */
int rih_scatter_rows_add(const float* dy, const int32_t* inv_ptr, const int32_t* inv_idx, float* dx, int B, int Vin,
                         int Vout, int D, void* stream);

/* Orthographic projection uv = (scale*img)*xyz[:2] + (trans*img/2 + img/2)   (utils/manoutils.py:26-44)
 * v[B][V][3], scale[B], trans[B][2] -> out[B][V][2]; backward gives dv (z component 0), dscale, dtrans. */
int rih_project_fwd(const float* v, const float* scale, const float* trans, float* out, int B, int V,
                    float img_size, void* stream);
int rih_project_bwd(const float* dout, const float* v, const float* scale, float* dv, float* dscale,
                    float* dtrans, int B, int V, float img_size, void* stream);

/* Chebyshev K=2 feature build: y[b][v][2f+0] = x[b][v][f], y[b][v][2f+1] = sum_j L[v][j] x[b][j][f]
 * with L in CSR (models/model_attn/gcn.py:34-69).  Backward takes CSR of L^T. */
int rih_cheby_fwd(const float* x, const int32_t* indptr, const int32_t* indices, const float* vals, float* y,
                  int B, int V, int F, void* stream);
int rih_cheby_bwd(const float* dy, const int32_t* t_indptr, const int32_t* t_indices, const float* t_vals,
                  float* dx, int B, int V, int F, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MANO linear-blend-skinning layer   (models/manolayer.py:250-322)
 * Constant model (uploaded once by the caller, device pointers):
 *   comps[45][45] (PCA basis rows), hands_mean[45], shapedirs[778][3][10], posedirs[778][3][135],
 *   v_template[778][3], J_reg[16][778] (dense), weights[778][16], parent[16] (host array).
 * Per call: root[B][9], pose (PCA coeffs [B][ncomp] if ncomp>0, else rotation matrices [B][15][9]),
 *   shape[B][10], trans[B][3] or NULL, scale[B] or NULL.  Outputs v[B][778][3], j[B][21][3].
 * packed: >= rih_mano_pack_floats() floats (16-byte aligned) written by rih_mano_pack from the model buffers: the blend
 *   bases as one k-major matrix [148][2496] (posedirs | shapedirs | v_template, coordinates padded to 13 tiles of 192) and
 *   the joint regressor folded onto template and shape basis, and (ABI 17) a compact [148][48] copy of the basis columns of the
 *   13 special vertices (finger tips, new_skel).  Re-pack whenever a model buffer changes (the reference's
 *   callers mutate shapedirs in place, dataset/interhand.py:22-25).
 * rih_mano_fwd variant 0: ONE fused launch (workgroup = 64-vertex tile of the packed basis pinned in LDS x group of hand
 *   chunks; pose chain per chunk, blend GEMM on v_mfma_f32_16x16x4_f32, skinning).  ws may be NULL (inference: only v and j
 *   are written); with ws (>= rih_mano_ws_floats(B) floats, kept by the caller) it also stores what rih_mano_bwd needs.
 *   From 256 hand chunks (4096 hands) on, variant 0 runs the same kernel hand-chunk major: a workgroup does the pose chain of
 *   its 16 hands once and streams the 13 basis tiles through the LDS buffer from L2 (the tile-major form repeats the pose
 *   work in each of the 13 tiles: 47 % of its cycles; measured 259 -> 212 us for 4096 hands).  variant 2 / 3 force the hand-major / tile-major form (tests, timing).
 *   variant 1: the two-kernel forward of round 1 (needs ws; kept for A/B timing).                                      */
typedef struct rih_mano_model {
    const float* comps;
    const float* hands_mean;
    const float* shapedirs;
    const float* posedirs;
    const float* v_template;
    const float* J_reg;
    const float* weights;
    int32_t parent[16];
} rih_mano_model;

int64_t rih_mano_ws_floats(int B);
int64_t rih_mano_pack_floats(void);
int rih_mano_debug_stamps(long long* device_buf_13x16);   /* development aid: phase timestamps of the fused forward; NULL = off */
int rih_mano_pack(const rih_mano_model* m, float* packed, void* stream);
int rih_mano_fwd(const rih_mano_model* m, const float* packed, const float* root, const float* pose, int ncomp,
                 const float* shape, const float* trans, const float* scale, int center_idx, int new_skel, float* v,
                 float* j, float* ws, int B, int variant, void* stream);
/* gradients wrt root/pose/shape/trans/scale (any may be NULL) given dv[B][778][3], dj[B][21][3] and the forward's ws (16-byte
 * aligned: the hand's SE3s are read with 16-byte scalar loads).
 * Three launches (round 5): (1) per hand the skinning / kinematic-chain part -- one workgroup per hand; the SE3s as wave-uniform
 * scalar operands, dG = W^T M on v_mfma_f32_16x16x4_f32; (2) the two contractions with the blend bases (pose-blend and shape
 * gradients: 0.33 MFLOP per hand against 1.26 MB of basis) TILE major: a workgroup pins one of the 13 tiles of `packed` in LDS
 * and walks over 16-hand chunks of dv_tpose, writing partial sums; (3) per hand (one wavefront) the sum of the 13 partials in
 * tile order, the pose-blend term, Rodrigues backward, PCA projection, shape gradient.  ws_bwd (>= rih_mano_bwd_ws_floats(B)
 * floats, 16-byte aligned) carries dv_tpose / rotation and rest-joint gradients / the partial sums between them.
 * 4096 hands: 175 us (rounds 3-4: 362); 128 hands: 39 us (174).  ws_bwd == NULL runs the one-kernel backward of rounds 1-2
 * (every workgroup re-reads the whole basis from L2: 1.17 ms for 4096 hands; A/B partner). */
int rih_mano_bwd(const rih_mano_model* m, const float* packed, const float* root, const float* pose, int ncomp,
                 const float* shape, const float* trans, const float* scale, int center_idx, int new_skel, const float* dv,
                 const float* dj, const float* ws, float* d_root, float* d_pose, float* d_shape, float* d_trans,
                 float* d_scale, float* ws_bwd, int B, void* stream);
int64_t rih_mano_bwd_ws_floats(int B);

/* ------------------------------------------------------------------------------------------------
 * Fused mesh loss   (core/Loss.py:68-164 GraphLoss.calc_loss, :201-277 calc_loss_GCN; aux loss disabled there)
 * Constant topology of one hand (device pointers, uploaded once by the caller):
 *   faces[F][3]; vptr[V+1], vlist[3F]: for every vertex the list of (face*3 + corner) it belongs to, ascending;
 *   J[NJ][V] dense joint regressor incl. the 5 finger tips, reordered (Loss.py:38-53); perm[Vc*pool] = graph_perm
 *   (vert_to_GCN); pool = 2^k consecutive graph-order vertices are average-pooled pairwise into one coarse vertex.
 * rih_mesh_loss (one launch per hand): predictions v3d_pred[B][V][3], v2d_pred[B][V][2], c3d_pred[B][Vc][3],
 *   c2d_pred[B][Vc][2]; labels v3d_gt, v2d_gt; gt_shift[B][3] or NULL is added to v3d_gt (root_rel of the right hand).
 *   term_weights[7] (DEVICE array, read by the kernel at run time so that a captured hipGraph follows in-place updates
 *   such as the epoch gate of the edge term; order v2d, v3d, joint, normal, edge, coarse-3d, coarse-2d) = weight of the
 *   raw SUM of each term in the total, i.e. LOSS_WEIGHT / element count / 2 (hand average).  Writes the gradient of the total
 *   with respect to the four prediction tensors and partial[B][8] (raw sums of the seven terms per image).
 * rih_mesh_loss_final: out[0] = total over both hands; out[1..7] = the seven terms as the reference reports them
 *   (mean over elements, averaged over the hands); counts[7] = element counts (DEVICE array). */
typedef struct rih_mesh_topo {
    const int32_t* faces;
    const int32_t* vptr;
    const int32_t* vlist;
    const float* J;
    const int32_t* perm;
    int32_t V, F, NJ, Vc, pool;
} rih_mesh_topo;

int rih_mesh_loss(const rih_mesh_topo* topo, const float* v3d_pred, const float* v2d_pred, const float* c3d_pred,
                  const float* c2d_pred, const float* v3d_gt, const float* v2d_gt, const float* gt_shift,
                  const float* term_weights, float img_size, float* g_v3d, float* g_v2d, float* g_c3d, float* g_c2d,
                  float* partial, int B, void* stream);
int rih_mesh_loss_final(const float* partial_left, const float* partial_right, int B, const float* term_weights,
                        const float* counts, float* out, void* stream);

/* library / device info.  RIH_ABI_VERSION is bumped whenever a struct layout or a signature of this header changes;
 * rih_version() returns the value the library was compiled with and rih_abi_sizes() the sizeof of EVERY by-pointer struct, in
 * this order: gemm desc, mano model, mesh topo, hconv desc, reduce desc, pack desc, ln final desc, adam entry, absmax desc,
 * conv3 desc, h2 desc, panel desc (RIH_ABI_NSIZES values), so a host binding can refuse a stale binary instead of handing it
 * mis-laid-out structs. */
#define RIH_ABI_VERSION 19
#define RIH_ABI_NSIZES 12
int rih_version(void);
int rih_abi_sizes(int32_t* out10);
const char* rih_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Attention without a score matrix in memory (csrc/rih_flash.hip) -- the default path of SelfAttn / inter_attn
 * (models/model_attn/self_attn.py:70-76: `attn = softmax(q k^T / sqrt(d_q)); attn = dropout(attn); out = attn v`;
 * inter_attn.py:93-107: the two cross-hand directions).  q / k / v are head-sliced in place: row pitches q_ld / kv_ld floats,
 * head h at column h*d, slice b at row b*Sq (b*Sk); d in {16, 32, 64}; B*heads <= 65535.
 *   rih_flash_attention_fwd: out[b][i][h*d + c] = sum_j Pd[i][j] v[j][c], Pd = dropout(softmax_j(alpha q_i . k_j)); one launch;
 *     lse[B*heads][Sq] receives log2(sum_j 2^(alpha log2(e) q_i . k_j)) -- all the backward needs besides q, k, v, out.
 *     Dropout: element (row r = (b*heads + h)*Sq + i, column j) is kept iff hash(seed (+ *seed_dev), r*Sk + j) >= p * 2^32, kept
 *     values scaled by 1/(1-p) -- the mask of rih_softmax_fwd, bit for bit.
 *   rih_flash_attention_bwd: dq, dk, dv (row pitches dq_ld / dkv_ld, same head slicing) from dO [B][Sq][do_ld] and the saved
 *     out / lse, recomputing the probabilities tile by tile; two launches (query side, then key side); Dws [B*heads][Sq] is
 *     workspace (receives rowsum(dO o out)).  Every element of the dq / dk / dv head slices is written exactly once. */
int rih_flash_attention_fwd(const float* q, int q_ld, const float* k, const float* v, int kv_ld, int B, int heads, int Sq,
                            int Sk, int d, float alpha, float drop_p, uint64_t seed, const uint64_t* seed_dev, float* out,
                            int ld_out, float* lse, void* stream);
int rih_flash_attention_bwd(const float* dO, int do_ld, const float* O, int o_ld, const float* q, int q_ld, const float* k,
                            const float* v, int kv_ld, int B, int heads, int Sq, int Sk, int d, float alpha, float drop_p,
                            uint64_t seed, const uint64_t* seed_dev, const float* lse, float* Dws, float* dq, int dq_ld,
                            float* dk, float* dv, int dkv_ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused attention forward, first generation (keeps P / Pd in memory; superseded by rih_flash_attention_*, kept as an opt-in
 * experiment RIH_FUSED_ATTN=1): out = softmax(alpha q k^T)
 * [dropout] v per (image, head) in ONE launch; q / k / v are head-sliced in place (row pitches q_ld / kv_ld, head h at
 * column h*d), out [B][Sq][ld_out] at column h*d.  P and Pd [B][heads][Sq][ldP] receive the probabilities before / after
 * dropout for the backward (Pd may equal P when drop_p == 0).  d in {16, 32, 64}, Sk <= 320.  Dropout mask and seed as
 * rih_softmax_fwd.  Not yet measured on hardware (see csrc/rih_attn.hip). */
int rih_attention_fwd_fused(const float* q, int q_ld, const float* k, const float* v, int kv_ld, int B, int heads, int Sq,
                            int Sk, int d, float alpha, float drop_p, uint64_t seed, const uint64_t* seed_dev, float* P,
                            float* Pd, int ldP, float* out, int ld_out, void* stream);
/* Query side of the attention backward in ONE launch (replaces the dO V^T GEMM, rih_softmax_bwd and the dS K GEMM):
 * dS = alpha * P * (m*dPd - rowsum(m*dPd*P)) with dPd = dO V^T and m the regenerated dropout mask (x 1/(1-p)), written to
 * dS [B][heads][Sq][ldP] for the key-side products (dK = dS^T q, dV = Pd^T dO stay batched GEMMs), and dq = dS k written
 * to dq [B][Sq][dq_ld] at column h*d.  dO [B][Sq][do_ld] head-sliced like q.  Same limits as the forward. */
int rih_attention_bwd_dq_fused(const float* dO, int do_ld, const float* k, const float* v, int kv_ld, int B, int heads,
                               int Sq, int Sk, int d, float alpha, float drop_p, uint64_t seed, const uint64_t* seed_dev,
                               const float* P, float* dS, int ldP, float* dq, int dq_ld, void* stream);
/* Key side of the attention backward in ONE launch (replaces two transposed batched GEMMs): dv = Pd^T dO and
 * dk = dS^T q, written to [B][Sk][dkv_ld] at column h*d.  Pd / dS [B][heads][Sq][ldP] (dS from the kernel above). */
int rih_attention_bwd_dkv_fused(const float* dO, int do_ld, const float* q, int q_ld, int B, int heads, int Sq, int Sk,
                                int d, const float* Pd, const float* dS, int ldP, float* dk, float* dv, int dkv_ld,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp16-storage inference backbone (BASELINE configs[4]; csrc/rih_half.hip).  Eval-mode only: every BatchNorm of
 * models/encoder.py is folded -- Conv->BN(->ReLU) (trunk, encoder.py:107-116) into weights (rih_hpack_conv_weight's
 * `scale`) and `bias`; Conv->ReLU->BN (aux decoders encoder.py:52-54, mid convs model_zoo/__init__.py:56-62) into
 * post_scale / post_shift.  Activations are NHWC fp16 with a pixel pitch (ldx / ldr / ldy, in elements) so that channel
 * slices of a wider tensor can be read and written in place (the channel concatenations of encoder.py:160-171 are never
 * copied).  fp32 accumulation (v_mfma_f32_32x32x16_f16), fp32 epilogue:
 *   y[pix][co] = post( relu?( sum_k x[...] w[co][k] + bias[co] + res[pix][co] ) ),  post(u) = u*post_scale[co] + post_shift[co]
 * Requirements: Cin % 8 == 0, ldx % 8 == 0, x / w / zero 16-byte aligned, Kpad % 64 == 0, Ho/Wo consistent with the
 * geometry; `zero` points at >= 16 bytes of zeros in device memory (source of every padded operand chunk).  16-byte
 * epilogue accesses are used when y / res rows are aligned (ldy % 8 (fp16) or % 4 (fp32), ldr % 8), scalar otherwise. */
typedef struct rih_hconv_desc {
    const void* x;            /* fp16 [N][H][W][ldx], channels [0, Cin) */
    const void* w;            /* fp16 [Cout][Kpad] from rih_hpack_conv_weight, k = (kh, kw, ci) */
    const void* zero;
    const float* bias;        /* [Cout] or NULL */
    const float* post_scale;  /* [Cout] or NULL (both or neither) */
    const float* post_shift;
    const void* res;          /* fp16 [pixels][ldr] or NULL */
    void* y;                  /* [pixels][ldy] fp16, or fp32 when out_f32 */
    int32_t N, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;
    int32_t ldx, ldr, ldy, Kpad;
    int32_t relu, out_f32;
} rih_hconv_desc;
int rih_hconv(const rih_hconv_desc* d, void* stream);
/* fp32 OIHW -> fp16 [Cout][Kpad], k = (kh*KW + kw)*CinPad + ci, times scale[co] when scale != NULL; zero padded. */
int rih_hpack_conv_weight(const float* w, const float* scale, void* dst, int Cout, int Cin, int KH, int KW, int CinPad,
                          int Kpad, void* stream);
/* Eval BatchNorm as an affine map: scale = gamma/sqrt(var+eps), shift = beta - mean*scale + scale*conv_bias
 * (gamma / beta / conv_bias may be NULL = 1 / 0 / 0). */
int rih_hbn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                 float eps, float* scale, float* shift, int C, void* stream);
/* [N][C][H][W] fp32 (C <= 8) -> [N][H][W][8] fp16, channels >= C zero. */
int rih_himage_nchw_to_nhwc8(const float* img, void* out, int N, int C, int H, int W, void* stream);
/* fp16 NHWC 3x3 stride-2 pad-1 max-pool, bilinear x2 (align_corners=True), global average pool (fp32 [N][C] out). */
int rih_hmaxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int ldx, int ldy, void* stream);
int rih_hupsample2x(const void* x, void* y, int N, int H, int W, int C, int ldx, int ldy, void* stream);
int rih_havgpool(const void* x, float* y, int N, int HW, int C, int ldx, void* stream);

/* Contact deviation (apps/eval_interhand.py:481-490, utils/eval_metrics.py:30-50 `compute_cdev`): per sample, the mean over
 * the ground-truth right-hand vertices whose nearest ground-truth left-hand vertex is within `contact` (3e-3 m) of
 * |pred_left[nearest] - pred_right[v]|; NaN when there is no such vertex (the reference's nanmean then skips the sample).
 * All meshes [B][V][3], V <= 1024; out [B]. */
int rih_cdev(const float* pred_left, const float* pred_right, const float* gt_left, const float* gt_right, int B, int V,
             float contact, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input preparation of a batch resident on the GPU (csrc/rih_input.hip) -- core/loader.py:104-219
 * `handDataset.process_data` and utils/manoutils.py:214-260 (`imgUtils.data_augmentation`, `add_noise`).
 * rih_prepare_images: src [B][S][S][3] uint8 BGR (decoded crops).  minv [B][6] = the INVERSE (destination -> source)
 *   of the matrix the reference hands to cv.warpAffine, in double precision (NULL: no warp); 8-bit INTER_LINEAR /
 *   BORDER_CONSTANT fixed-point algorithm of OpenCV.  bright [B][4] = per-channel gain a_b, a_g, a_r and offset b of
 *   `add_noise` (NULL: none).  flip [B] (NULL: none).  Outputs (each may be NULL): aug_u8 [B][S][S][3] the augmented
 *   crop, ori [B][3][S][S] = BGR / 255, norm [B][3][S][S] = ImageNet-normalised RGB, norm_h8 [B][S][S][8] fp16 NHWC
 *   (the input layout of rih_hconv's first layer).
 * rih_prepare_labels: p2 [B][NP][2], p3 [B][NP][3] with NP = 2*(NV+NJ) points ordered [verts_left, joints_left,
 *   verts_right, joints_right]; A [B][6] the first two rows of the float32 affine matrix (NULL: identity), R [B][9] the
 *   in-plane rotation (NULL: identity); root-relative to joint `root_joint` of each hand, scaled so that the mean
 *   |joint root - joint 0| equals bone_length (<= 0: off); flipped samples mirror x (2-D: img_size - x), negate
 *   root_rel y/z as the reference does, and swap the hands in the output. */
int rih_prepare_images(const uint8_t* src, int B, int S, const double* minv, const double* bright, const uint8_t* flip,
                       uint8_t* aug_u8, float* ori, float* norm, void* norm_h8, void* stream);
int rih_prepare_labels(const float* p2, const float* p3, int B, int NV, int NJ, const float* A, const float* R,
                       const uint8_t* flip, float bone_length, int root_joint, float img_size, float* o2, float* o3,
                       float* root_rel, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Interior distance field of closed triangle meshes (csrc/rih_sdf.hip) -- the reference's only native kernel,
 * pose_data_optimize/sdf/sdf/csrc/sdf_cuda_kernel.cu:242-335 (`sdf_cuda`).  faces [F][3] int32 (shared by the batch),
 * vertices [B][V][3] inside [-1,1]^3, phi [B][G][G][G] indexed [b][z][y][x]: distance from the voxel centre
 * -1 + (i + 0.5) * 2/(G-1) to the closest triangle when the centre is inside the mesh (odd crossing count towards the
 * corner (-1,-1,-1)), else 0.  No gradient (as in the reference). */
int rih_sdf(float* phi, const int32_t* faces, const float* vertices, int B, int F, int V, int G, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MANO parameter head of the reference's `load_new_model` network (common/myhand/decoder_lijun_mano.py:112-160,247-300)
 * nn.Hardswish and scale*tanh (the ParamRegressor MLP, `F.tanh(shape) * 3`): elementwise, n floats. */
int rih_hardswish_fwd(const float* x, float* y, int64_t n, void* stream);
int rih_hardswish_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int rih_tanh_scale_fwd(const float* x, float* y, int64_t n, float scale, void* stream);
int rih_tanh_scale_bwd(const float* dy, const float* y, float* dx, int64_t n, float scale, void* stream);
/* x [n][6] (viewed (3,2): two column vectors) -> Gram-Schmidt rotation matrix R [n][3][3] (decoder_lijun_mano.py:118-125)
 * and its axis-angle vector aa [n][3] through the quaternion (common/myhand/utils/comm.py:176-324, NaN -> 0).
 * Backward: dx = J_R^T dR + J_aa^T daa (either gradient may be NULL), exact (forward-mode dual numbers). */
int rih_rot6d_fwd(const float* x, float* R, float* aa, int n, void* stream);
int rih_rot6d_bwd(const float* x, const float* dR, const float* daa, float* dx, int n, void* stream);
/* axis-angle [n][3] -> rotation matrix [n][3][3], angle = |axis| + 1e-8 (common/utils/manolayer.py:32-48) */
int rih_rodrigues_fwd(const float* a, float* R, int n, void* stream);
int rih_rodrigues_bwd(const float* a, const float* dR, float* da, int n, void* stream);
/* vout = (v - j[root]) * s, s = target / |j[ja] - j[jb]| per mesh (decoder_lijun_mano.py:262-267: root-centred,
 * bone-length-normalised MANO mesh, target 0.095 m); sout [B].  Backward: dv [B][V][3], dj [B][NJ][3] from dvout and
 * (may be NULL) dsout. */
int rih_center_scale_fwd(const float* v, const float* j, int B, int V, int NJ, int root, int ja, int jb, float target,
                         float* vout, float* sout, void* stream);
int rih_center_scale_bwd(const float* v, const float* j, const float* dvout, const float* dsout, int B, int V, int NJ,
                         int root, int ja, int jb, float target, float* dv, float* dj, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation metrics of one hand (apps/eval_interhand.py:334-415, common/utils/intag_eval.py:92-143,217-283):
 * joints = Jreg [NJ][V] x vertices unless given (j_pred / j_gt may be NULL), both point sets made relative to joint
 * `root_idx`, prediction rescaled by |gt bone| / |pred bone| (bone = joints bone_a, bone_b);
 *   j_err_ori [B][NJ], v_err_ori [B][V]   |pred - gt| before rescaling      (any of the four may be NULL)
 *   j_err     [B][NJ], v_err     [B][V]   |pred * scale - gt|
 *   pa        [B][2]                      Procrustes-aligned mean joint / vertex error (PA-MPJPE, PA-MPVPE): optimal
 *                                         similarity transform of the root-relative prediction onto the ground truth
 *   j_pred_out [B][NJ][3] (may be NULL)   the joints regressed from v_pred
 * V <= 1024, NJ <= 32.  One workgroup per image; no workspace. */
int rih_hand_metrics(const float* v_pred, const float* v_gt, const float* j_pred, const float* j_gt, const float* Jreg,
                     int B, int V, int NJ, int root_idx, int bone_a, int bone_b, float* j_err_ori, float* v_err_ori,
                     float* j_err, float* v_err, float* pa, float* j_pred_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RENDERIH_AMD_H */
