/*
 * renderih_amd_experiments.h -- entry points of the two EXPERIMENT sources, which are not part of the default
 * librenderih_amd.so (renderih_amd/_build.py: RIH_BUILD_EXPERIMENTS=1 compiles csrc/rih_gemm3.hip and csrc/rih_chain.hip in).
 * Both were built, parity-tested and measured in rounds 2-3 and lost: the P3 GEMM is power-bound like engine 1 of rih_gemm
 * (DESIGN 3.11), the row-chain kernel is 1.1 ms per step slower than the launches it fuses.  Same conventions as
 * renderih_amd.h (plain C ABI, device pointers + sizes + stream, int return codes).
 */
#ifndef RENDERIH_AMD_EXPERIMENTS_H
#define RENDERIH_AMD_EXPERIMENTS_H
#include "renderih_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Conversion-free split-bf16 GEMM on pre-split "P3" operands (csrc/rih_gemm3.hip).  Replaces rih_gemm for the ResNet trunk
 * of models/encoder.py:107-116 (torchvision resnet50 convolutions, forward and data gradient): the BatchNorm kernels that
 * produce an activation / gradient write it in P3, the weights are converted once per step, and the GEMM stages operands
 * global -> LDS by LDS-DMA with one barrier per 32-deep k-tile and converts nothing.
 *
 * P3 format of a row-major fp32 matrix [rows][C], C % 8 == 0, row pitch ld channels (ld % 8 == 0): 6 bytes per element;
 *   the 16-byte unit (row r, channel group g = c/8, plane p: 0 hi, 1 mid, 2 lo) holds 8 bf16 at byte
 *   ((r*ld/8 + g)*3 + p)*16, where x = hi + mid + lo with round-to-nearest at every level (|x - hi - mid - lo| <= 2^-24 |x|).
 *
 * rih_gemm_p3: C[m][n] = act(sum_k A(m,k) B(n,k) + bias[n] + R[m][n]); A(m,k): m = (img, ho, wo), k = (kh, kw, c),
 *   element X[img][ho*stride - padH + kh][wo*stride - padW + kw][c] of the P3 tensor X[*][H][W][lda] (0 outside);
 *   B = P3 [N][ldb] with ldb >= K = KH*KW*Cin.  Cin % 32 == 0, KH*KW <= 32, X smaller than 4 GiB.  A plain matrix product is
 *   H = Ho = rows, W = Wo = 1 (or any factorisation), KH = KW = stride = 1, pad 0.  `zero`: >= 16 readable zero bytes
 *   (16-byte aligned) that padding lanes load instead of the operand.  cS..cW as in rih_gemm_desc (strided output rows).
 *   stats != NULL (requires M % tile rows == 0): additionally writes per-column statistics of the stored values,
 *   stats[(tile_m*N + n)*2 + {0,1}] = (mean, sum of squared deviations from that mean) over the tile's rows; merge with
 *   rih_bn_stats_merge (Chan's formula in double) -> the batch statistics of nn.BatchNorm2d without a pass over C.
 *   tile 0: 256x128, 1: 128x128, 2: 128x64 (rows per tile: rih_gemm_p3_tile_rows).
 * rih_p3_from_f32: fp32 [rows][C] (pitch ldx) -> P3 (pitch ldo).  rih_p3_conv_weight: OIHW weight -> P3 [N][Kpad] as forward
 *   operand (for_dgrad 0: N = Cout, k = (tap, ci < CinPad)) or as flipped data-gradient operand of the tap subset
 *   kh0 + step*t, kw0 + step*t' (for_dgrad 1: N = CinPad, k = ((th, tw), co)); same conventions as rih_presplit_conv_weight. */
typedef struct rih_gemm_p3_desc {
    const void* A;
    const void* B;
    const void* zero;
    float* C;
    const float* bias;   /* [N] or NULL */
    const float* R;      /* residual [M][ldr] or NULL (not with cS > 1) */
    float* stats;        /* [ceil(M/rows)][N][2] or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr;
    int32_t H, W, Cin, Ho, Wo, KH, KW, stride, padH, padW;
    int32_t cS, cOH, cOW, cH, cW;
    int32_t relu;
    int32_t tile;
    int32_t layout;      /* 0: interleaved P3 (above); 1: slab-major "P3S": [C/32][rows][12 units] for A (rows = all pixels of the
                            tensor, lda ignored) and [K/32][N][12 units] for B -- the k-tile of consecutive pixels is contiguous,
                            so the LDS-DMA stream consists of whole 128-byte lines */
} rih_gemm_p3_desc;
int rih_gemm_p3(const rih_gemm_p3_desc* d, void* stream);
int rih_gemm_p3_tile_rows(int tile);
int rih_p3_from_f32(const float* x, int64_t rows, int C, int ldx, void* out, int ldo, int layout, void* stream);
int rih_p3_conv_weight(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int CinPad, int for_dgrad, int kh0,
                       int kw0, int step, int Th, int Tw, int Kpad, int layout, void* stream);
/* part [T][C][2] from rih_gemm_p3 (tiles of rows_per_tile rows) -> what rih_bn_stats produces: mean[C],
 * invstd[C] = 1 / sqrt(biased var + eps) and, when running_mean / running_var != NULL, their momentum update with the unbiased
 * variance (nn.BatchNorm2d training forward).  Chan's merge in double, one wavefront per channel. */
int rih_bn_stats_from_tiles(const float* part, int T, int C, int rows_per_tile, float eps, float momentum, float* mean,
                            float* invstd, float* running_mean, float* running_var, void* stream);
/* part [T][C][2] (tiles of rows_per_tile rows) -> mean[C], biased variance var[C] */
int rih_bn_stats_merge(const float* part, int T, int C, int rows_per_tile, float* mean, float* var, void* stream);


/* ------------------------------------------------------------------------------------------------
 * rih_chain: a chain of row-wise layers in ONE launch (csrc/rih_chain.hip) -- the Linear -> (dropout, add) -> LayerNorm ->
 * ReLU -> Linear sequences of the mesh decoder (models/model_attn/self_attn.py:17-33 `MLP_res_block`, :66-85 the attention
 * output projection + skip + `ff`; inter_attn.py:85-125; gcn.py:99-110 `GCN_ResBlock`) and their backward sequences.
 * Every operator of such a chain maps token rows to token rows independently, so a workgroup keeps a block of `rblk` rows in
 * LDS from the first load to the last store and runs the operators of `op[]` on it in order; the matrix products use the
 * exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) with the weight operand streamed from L2.  What a standalone-launch sequence pays
 * per operator -- a dependent launch (>= 4.5 us), a round trip of the activations through L2, a 64x64 GEMM tile with a
 * 4-k-tile main loop -- is paid once per chain.
 *
 * Activations are [nhands][rows][*] row-major (hands stacked, renderih_amd/attn.py); the workgroup of (row block, hand h)
 * addresses row r of a tensor with pitch `ld` at p + ((long long)h * rows + r) * ld, and parameters at p_i + h * s_i (s_i = the
 * distance in floats from the left to the right hand's parameter; 0 = shared).  The block state is `cur` [rblk][width] (the
 * running activation) and one optional second buffer `kept`.  Operators (kind, and what the other fields mean):
 *   RIH_CH_LOAD      cur = p0 rows (n columns, pitch ld); width = n
 *   RIH_CH_STORE     p0 rows (pitch ld) = cur
 *   RIH_CH_ADD       cur += p0 rows (pitch ld)
 *   RIH_CH_KEEP      kept = cur
 *   RIH_CH_ADD_KEPT  cur += kept
 *   RIH_CH_GEMM      cur[:, :n] = cur[:, :k] x B (+ p1[n] bias) (ReLU if flags & RIH_CHF_RELU); B(kk, j) = p0[j*k + kk] (an
 *                    nn.Linear weight [n][k] as stored) or, with RIH_CHF_BT, p0[kk*n + j] ([k][n]: the same weight in the data
 *                    gradient).  With RIH_CHF_OUT_GLOBAL the result goes to p2 rows (pitch ld) instead of `cur` (n is then not
 *                    limited by the LDS block; cur keeps its value).  With RIH_CHF_A_GLOBAL the left operand is read from
 *                    p3 rows (pitch `lda`) instead of `cur` (k is then not limited by the block either: the QKV data gradient,
 *                    k = 3 D; data-gradient form only).  k % 64 == 0, n % 32 == 0.  Fused epilogue of a product that stays in
 *                    the block, applied in this order to v = acc + bias (ReLU): RIH_CHF_EPI_MASKNZ v = (p4 rows (pitch lde)
 *                    != 0) ? v * f1 : 0; RIH_CHF_EPI_DROPOUT as RIH_CH_DROPOUT (f0, seed) on the [nhands][rows][n] result;
 *                    RIH_CHF_EPI_ADD v += p4 rows (pitch lde; not together with MASKNZ); RIH_CHF_EPI_ADD_KEPT v += kept;
 *                    RIH_CHF_EPI_STORE p2 rows (pitch ld) = v; RIH_CHF_EPI_KEEP kept = v.  (Each is what the separate operator
 *                    does; fused, its memory operand is requested before the product starts instead of after it.)
 *   RIH_CH_DROPOUT   cur = keep ? cur / (1 - f0) : 0, element (h, r, c) of the [nhands][rows][width] tensor kept iff
 *                    hash(seed (+ *seed_dev), ((h*rows + r)*width + c)) >= f0 * 2^32 -- the mask of rih_add_dropout /
 *                    rih_dropout_bwd on the same tensor, bit for bit
 *   RIH_CH_MASKNZ    cur = (p0 rows (pitch ld) != 0) ? cur * f0 : 0   (backward of ReLU [+ dropout] through the saved output)
 *   RIH_CH_LN        cur = LayerNorm(cur; gamma p0, beta p1, eps f0) (ReLU if RIH_CHF_RELU); mean / rstd of row (h, r) to
 *                    p2[h*rows + r] / p3[...] when p2 != NULL
 *   RIH_CH_LN_BWD    cur (= dy) -> dx of that LayerNorm: p0 = its input rows (pitch ld), p1 / p2 = mean / rstd, p3 = gamma;
 *                    the block's partial parameter gradients go to p4 + h*s4 laid out [nblk][2][width] (nblk = ceil(rows/rblk);
 *                    [.][0] = d gamma, [.][1] = d beta) for rih_ln_param_final_multi
 * Limits: 1 <= nops <= RIH_CHAIN_MAXOPS; rblk in {32, 64}; ldw >= every width + 4, ldw % 4 == 0, rblk * ldw <= 8448; all widths % 4 == 0; row pitches % 4 == 0 and pointers 16-byte aligned. */
#define RIH_CHAIN_MAXOPS 16
enum {
    RIH_CH_LOAD = 1, RIH_CH_STORE = 2, RIH_CH_ADD = 3, RIH_CH_KEEP = 4, RIH_CH_ADD_KEPT = 5, RIH_CH_GEMM = 6,
    RIH_CH_DROPOUT = 7, RIH_CH_MASKNZ = 8, RIH_CH_LN = 9, RIH_CH_LN_BWD = 10
};
#define RIH_CHF_RELU 1
#define RIH_CHF_BT 2
#define RIH_CHF_OUT_GLOBAL 4
#define RIH_CHF_A_GLOBAL 8
#define RIH_CHF_EPI_DROPOUT 16
#define RIH_CHF_EPI_ADD 32
#define RIH_CHF_EPI_ADD_KEPT 64
#define RIH_CHF_EPI_STORE 128
#define RIH_CHF_EPI_KEEP 256
#define RIH_CHF_EPI_MASKNZ 512
typedef struct rih_chain_op {
    int32_t kind, flags, n, k;
    int32_t ld, lda, lde, reserved;
    float f0, f1;
    uint64_t seed;
    const void* p0;
    const void* p1;
    void* p2;
    void* p3;
    void* p4;
    int64_t s0, s1, s2, s3, s4;
} rih_chain_op;
typedef struct rih_chain_desc {
    int32_t nops, rows, nhands, rblk;
    int32_t ldw, reserved;
    const uint64_t* seed_dev;
    rih_chain_op op[RIH_CHAIN_MAXOPS];
} rih_chain_desc;
int rih_chain(const rih_chain_desc* desc, void* stream);
/* The argument check of rih_chain on its own (no launch): 0 = launchable; -1 = a header field; -4 = the block does not fit the
 * LDS; otherwise -(100 * (index of the offending operator + 1) + reason), reason 1 = a pointer is missing / misaligned (or a
 * parameter stride is not a multiple of 4), 2 = a width / pitch / probability, 3 = operator order, 4 = too wide for the block. */
int rih_chain_check(const rih_chain_desc* desc);

#ifdef __cplusplus
}
#endif
#endif
