// rih_chain.hip -- a chain of row-wise decoder layers in ONE launch (include/renderih_amd.h: rih_chain).
//
// The mesh decoder (models/model_attn/self_attn.py:17-33, :66-85; inter_attn.py:85-125; gcn.py:99-110) is a sequence of
// small operators on token rows -- Linear (K, N <= 256), dropout, residual add, LayerNorm, ReLU -- that the standalone
// kernels run as one dependent launch each: >= 4.5 us of launch floor per operator, the activations through L2 between any
// two of them, and GEMMs whose main loop is 2..8 k-tiles long (20-60 TF on a 64x64 tile).  Every one of these operators maps
// a token row to a token row independently, so here a workgroup owns a block of 32 or 64 rows, keeps it in LDS from the first
// load to the last store and interprets the operator list of the descriptor on it.  The backward sequences are chains of the
// same operators (data-gradient GEMMs on the transposed weight view, mask re-draws, LayerNorm backward), so one kernel serves
// both directions; the weight gradients stay batched GEMMs over the tensors a chain stores on its way.
//
//   * state: `cur` [rblk][ldw] fp32 (the running activation) and an optional second buffer `kept` (skip connections);
//     ldw = widest activation + 4 floats, so that the 16-byte operand reads of 32 consecutive rows spread over the banks.
//   * matrix products: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), one 32x32 output block per wavefront and pass; the A operand
//     is read from `cur` (one ds_read_b128 per four MFMAs), the B operand -- the weight -- goes from L2 straight into
//     registers, a whole 128-deep slice of the reduction (16 x 16 bytes per lane) requested before the first MFMA, so that a
//     product pays ONE memory round trip, not one per k-tile.  The k index is permuted inside a group of 8 (lane half h, step
//     t <-> k = 8j + 4h + t) identically for both operands, which lets each operand fetch be one contiguous 16-byte access.
//   * everything else (dropout masks from the counter hash of rih_hash.h, LayerNorm forward / backward with 4 or 8 lanes per
//     row, residual adds) works on the block in LDS with 16-byte accesses.
// Determinism: no atomics; the LayerNorm parameter gradients leave the kernel as per-block partial sums in a fixed order
// (finished by rih_ln_param_final_multi like those of rih_layernorm_bwd).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd_experiments.h"
#include "rih_hash.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));    // (register arrays of HIP's float4 -- a struct of unions -- end up in scratch memory)
constexpr int TPB = 256;
constexpr int CAP_BIG = 8448;     // floats of `cur` (and of `kept`): 64 rows x (128 + 4); 32 rows x (256 + 4) fits too
constexpr int CAP_SMALL = 4224;   // 32 rows x (128 + 4): two workgroups per CU next to the weight tiles

__device__ __forceinline__ int acc_row(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }
__device__ __forceinline__ uint32_t ch_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

struct Blk {
    int h, row0, nrows, R, ldw;
    long long rowbase;          // h * rows + row0: index of the block's first row in a hands-stacked tensor
};

// ---- nxt[:, :n] = cur[:, :k] x B (+ bias) (ReLU), or the same product written to global rows ------------------------------
// A wavefront owns the 32-column blocks cb = wave, wave + 4, ... of the result and, per block, ALL row blocks of the workgroup
// (one weight tile feeds one or two accumulators).  The weight streams in k-chunks of KC floats: a chunk of a column block is
// 32 x KC, requested from L2 with fully used 128-byte lines (16 bytes per lane and request), parked in the wavefront's private
// LDS tile and read back in MFMA operand order; the requests of the next chunk are issued before the MFMAs of the current one,
// so the L2 round trip hides behind the arithmetic.  (Operand fetches straight from L2 -- each lane its own 16 bytes of a row
// -- use a quarter of every line they pull through the CU's L1: measured 3.5 x the MFMA time on the decoder's D = 256 blocks.)
//   NT weight ([n][k] as nn.Linear stores it): tile [32 columns][KC + 4], 16-byte writes and operand reads;
//   BT weight ([k][n], the data gradient):     tile [KC][32 columns], 16-byte writes, four 4-byte operand reads per group.
// The result goes to the block's OTHER activation buffer (the caller swaps the two), so a finished column block leaves the
// registers at once and the loop body exists once.  AG: the A operand comes from global rows p3 (pitch lda) instead of `cur`.
#define WT_ISSUE(n0_, k0_)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {      /* n % 32 == 0 and k % KC == 0: every tile is full */       \
        const int idx_ = lane + 64 * i_;                                                                                \
        if (!BT) q[i_] = *reinterpret_cast<const f4*>(W + (long long)((n0_) + idx_ / (KC / 4)) * K + (k0_) + 4 * (idx_ % (KC / 4))); \
        else q[i_] = *reinterpret_cast<const f4*>(W + (long long)((k0_) + (idx_ >> 3)) * N + (n0_) + 4 * (idx_ & 7)); \
    }
#define WT_COMMIT()                                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {                                                                 \
        const int idx_ = lane + 64 * i_;                                                                                \
        if (!BT) *reinterpret_cast<f4*>(wt + (idx_ / (KC / 4)) * (KC + 4) + 4 * (idx_ % (KC / 4))) = q[i_];          \
        else *reinterpret_cast<f4*>(wt + 4 * idx_) = q[i_];       /* [kr][32]: idx = kr * 8 + column quad */         \
    }

template <bool AG, bool BT, int NRB, int KC>
__device__ __forceinline__ void ch_gemm(const rih_chain_op& op, const float* __restrict__ cur, float* __restrict__ nxt,
                                        float* __restrict__ wtile, const Blk& b, int tid) {
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int K = op.k, N = op.n, ldw = b.ldw;
    const float* __restrict__ W = reinterpret_cast<const float*>(op.p0) + b.h * op.s0;
    const float* __restrict__ bias = op.p1 ? reinterpret_cast<const float*>(op.p1) + b.h * op.s1 : nullptr;
    const bool outg = (op.flags & RIH_CHF_OUT_GLOBAL) != 0, relu = (op.flags & RIH_CHF_RELU) != 0;
    const int ncb = N >> 5, nchunk = K / KC;
    float* __restrict__ wt = wtile + wave * (32 * (KC + 4));
    float* __restrict__ dst = outg ? reinterpret_cast<float*>(op.p2) + b.rowbase * op.ld : nullptr;
    // A rows of this lane: LDS rows l31 (+32), or -- AG -- global rows (those behind the block's last one re-read it: finite
    // values that nothing stores)
    const float* a0 = cur + l31 * ldw + 4 * lhi;
    long long astep = 32 * ldw;
    if (AG) {
        const float* A = reinterpret_cast<const float*>(op.p3) + 4 * lhi;
        const int r0 = min(l31, b.nrows - 1), r1 = min(32 + l31, b.nrows - 1);
        a0 = A + (b.rowbase + r0) * (long long)op.lda;
        astep = (long long)(r1 - r0) * op.lda;
    }
    // Every workgroup streams the same weight: started at the same chunk, the CUs of an XCD would all ask the same few L2
    // channels for the same lines at the same time (rows of a power-of-two pitch: a 32 x KC tile lives in 4 of 16 channels --
    // measured 1.3 TB/s of weight traffic for the whole chip).  Each workgroup therefore walks the chunks of a column block
    // from its own starting point (the k order of a row's sum depends on its block index, not on the run).
    const int rot = (int)(blockIdx.x + blockIdx.y) % nchunk;
    constexpr int NV = 32 * KC / 256;       // 16-byte requests per lane and chunk
    f4 q[NV];
    if (wave < ncb) WT_ISSUE(wave * 32, rot * KC)
    for (int cb = wave; cb < ncb; cb += 4) {
        floatx16 acc[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        int kc = rot;                               // chunk being multiplied
        for (int ch = 0; ch < nchunk; ++ch) {
            WT_COMMIT()
            __builtin_amdgcn_wave_barrier();        // (the tile is read by other lanes of this wavefront than wrote it)
            const int kn = (kc + 1 == nchunk) ? 0 : kc + 1;
            if (ch + 1 < nchunk) {
                WT_ISSUE(cb * 32, kn * KC)
            } else if (cb + 4 < ncb) {
                WT_ISSUE((cb + 4) * 32, rot * KC)
            }
#pragma unroll
            for (int j = 0; j < KC / 8; ++j) {
                float4 bv;
                if (!BT) {
                    bv = *reinterpret_cast<const float4*>(wt + l31 * (KC + 4) + 8 * j + 4 * lhi);
                } else {
                    const float* wp = wt + (8 * j + 4 * lhi) * 32 + l31;
                    bv = make_float4(wp[0], wp[32], wp[64], wp[96]);
                }
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const float4 a = *reinterpret_cast<const float4*>(a0 + rb * astep + kc * KC + 8 * j);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bv.x, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bv.y, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bv.z, acc[rb], 0, 0, 0);
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bv.w, acc[rb], 0, 0, 0);
                }
            }
            kc = kn;
        }
        const int col = cb * 32 + l31;
        const float bvs = bias ? bias[col] : 0.f;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + acc_row(r, lhi);
                float v = acc[rb][r] + bvs;
                if (relu) v = fmaxf(v, 0.f);
                if (outg) {
                    if (row < b.nrows) dst[(long long)row * op.ld + col] = v;
                } else {
                    nxt[row * ldw + col] = v;
                }
            }
    }
}

#undef WT_ISSUE
#undef WT_COMMIT

// ---- LayerNorm forward on the block: T = 256 / R lanes per row -----------------------------------------------------------------
__device__ __forceinline__ float group_sum(float v, int T) {
    for (int m = 1; m < T; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ void ch_ln(const rih_chain_op& op, float* __restrict__ cur, const Blk& b, int tid, int width) {
    const int T = TPB / b.R, r = tid / T, sub = tid % T, w4 = width >> 2;
    const float* __restrict__ g = reinterpret_cast<const float*>(op.p0) + b.h * op.s0;
    const float* __restrict__ be = reinterpret_cast<const float*>(op.p1) + b.h * op.s1;
    float4* row = reinterpret_cast<float4*>(cur + r * b.ldw);
    float s = 0.f;
    for (int c4 = sub; c4 < w4; c4 += T) {
        const float4 v = row[c4];
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float m = group_sum(s, T) / (float)width;
    float q = 0.f;
    for (int c4 = sub; c4 < w4; c4 += T) {
        const float4 v = row[c4];
        const float a0 = v.x - m, a1 = v.y - m, a2 = v.z - m, a3 = v.w - m;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rs = 1.f / sqrtf(group_sum(q, T) / (float)width + op.f0);
    const bool relu = (op.flags & RIH_CHF_RELU) != 0;
    for (int c4 = sub; c4 < w4; c4 += T) {
        const float4 v = row[c4];
        const float4 gg = *reinterpret_cast<const float4*>(g + 4 * c4), bb = *reinterpret_cast<const float4*>(be + 4 * c4);
        float4 o;
        o.x = (v.x - m) * rs * gg.x + bb.x;
        o.y = (v.y - m) * rs * gg.y + bb.y;
        o.z = (v.z - m) * rs * gg.z + bb.z;
        o.w = (v.w - m) * rs * gg.w + bb.w;
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        row[c4] = o;
    }
    if (op.p2 != nullptr && sub == 0 && r < b.nrows) {
        reinterpret_cast<float*>(op.p2)[b.rowbase + r] = m;
        reinterpret_cast<float*>(op.p3)[b.rowbase + r] = rs;
    }
}

// ---- LayerNorm backward on the block: cur = dy -> dx; the block's d gamma / d beta partial sums to the workspace -------------
__device__ __forceinline__ void ch_ln_bwd(const rih_chain_op& op, float* __restrict__ cur, float* __restrict__ stat,
                                          const Blk& b, int tid, int width, int nblk) {
    const float* __restrict__ x = reinterpret_cast<const float*>(op.p0) + b.rowbase * op.ld;
    const float* __restrict__ g = reinterpret_cast<const float*>(op.p3) + b.h * op.s3;
    float* smean = stat;
    float* srstd = stat + 64;
    if (tid < b.R) {
        const bool ok = tid < b.nrows;
        smean[tid] = ok ? reinterpret_cast<const float*>(op.p1)[b.rowbase + tid] : 0.f;
        srstd[tid] = ok ? reinterpret_cast<const float*>(op.p2)[b.rowbase + tid] : 0.f;
    }
    __syncthreads();
    // (1) parameter-gradient partial sums: a lane per column, the block's rows in order
    float* ws = reinterpret_cast<float*>(op.p4) + b.h * op.s4 + (long long)blockIdx.x * 2 * width;
    (void)nblk;
    for (int c = tid; c < width; c += TPB) {
        float sg = 0.f, sb = 0.f;
        for (int r = 0; r < b.nrows; ++r) {
            const float dy = cur[r * b.ldw + c];
            const float xh = (x[(long long)r * op.ld + c] - smean[r]) * srstd[r];
            sg += dy * xh;
            sb += dy;
        }
        ws[c] = sg;
        ws[width + c] = sb;
    }
    __syncthreads();
    // (2) dx = rstd * (g dy - mean(g dy) - xhat mean(g dy xhat)), T lanes per row
    const int T = TPB / b.R, r = tid / T, sub = tid % T, w4 = width >> 2;
    float4* row = reinterpret_cast<float4*>(cur + r * b.ldw);
    const float m = smean[r], rs = srstd[r];
    const bool ok = r < b.nrows;
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)(ok ? r : 0) * op.ld);
    float s1 = 0.f, s2 = 0.f;
    for (int c4 = sub; c4 < w4; c4 += T) {
        const float4 dy = row[c4], gg = *reinterpret_cast<const float4*>(g + 4 * c4);
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) xv = xr[c4];
        const float h0 = (xv.x - m) * rs, h1 = (xv.y - m) * rs, h2 = (xv.z - m) * rs, h3 = (xv.w - m) * rs;
        const float t0 = dy.x * gg.x, t1 = dy.y * gg.y, t2 = dy.z * gg.z, t3 = dy.w * gg.w;
        s1 += (t0 + t1) + (t2 + t3);
        s2 += (t0 * h0 + t1 * h1) + (t2 * h2 + t3 * h3);
    }
    s1 = group_sum(s1, T) / (float)width;
    s2 = group_sum(s2, T) / (float)width;
    for (int c4 = sub; c4 < w4; c4 += T) {
        const float4 dy = row[c4], gg = *reinterpret_cast<const float4*>(g + 4 * c4);
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) xv = xr[c4];
        float4 o;
        o.x = rs * (dy.x * gg.x - s1 - (xv.x - m) * rs * s2);
        o.y = rs * (dy.y * gg.y - s1 - (xv.y - m) * rs * s2);
        o.z = rs * (dy.z * gg.z - s1 - (xv.z - m) * rs * s2);
        o.w = rs * (dy.w * gg.w - s1 - (xv.w - m) * rs * s2);
        row[c4] = o;
    }
}

template <int CUR_F, int KEPT_F, int KC, int NRB>
__global__ __launch_bounds__(TPB, (CUR_F <= CAP_SMALL ? 2 : 1)) void chain_kernel(const rih_chain_desc d) {
    __shared__ float4 cur4[CUR_F / 4];
    __shared__ float4 nxt4[CUR_F / 4];
    __shared__ float4 kept4[(KEPT_F > 0 ? KEPT_F : 4) / 4];
    __shared__ float stat[128];
    __shared__ float4 wtile4[4 * 32 * (KC + 4) / 4];
    float* wtile = reinterpret_cast<float*>(wtile4);
    float* cur = reinterpret_cast<float*>(cur4);
    float* nxt = reinterpret_cast<float*>(nxt4);
    float* kept = reinterpret_cast<float*>(kept4);
    const int tid = threadIdx.x;
    Blk b;
    b.h = blockIdx.y;
    b.R = d.rblk;
    b.ldw = d.ldw;
    b.row0 = blockIdx.x * d.rblk;
    b.nrows = min(d.rblk, d.rows - b.row0);
    b.rowbase = (long long)b.h * d.rows + b.row0;
    const int nblk = gridDim.x;
    uint64_t seed_add = 0;
    if (d.seed_dev != nullptr) seed_add = *d.seed_dev;
    int width = 0;
    for (int i = 0; i < d.nops; ++i) {
        const rih_chain_op& op = d.op[i];
        switch (op.kind) {
        case RIH_CH_LOAD: {
            width = op.n;
            const int w4 = width >> 2;
            const float* src = reinterpret_cast<const float*>(op.p0) + b.rowbase * op.ld;
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < b.nrows) v = *reinterpret_cast<const float4*>(src + (long long)r * op.ld + c);
                *reinterpret_cast<float4*>(cur + r * b.ldw + c) = v;
            }
            break;
        }
        case RIH_CH_STORE: {
            const int w4 = width >> 2;
            float* dst = reinterpret_cast<float*>(const_cast<void*>(op.p0)) + b.rowbase * op.ld;
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                if (r < b.nrows) *reinterpret_cast<float4*>(dst + (long long)r * op.ld + c) =
                    *reinterpret_cast<const float4*>(cur + r * b.ldw + c);
            }
            break;
        }
        case RIH_CH_ADD: {
            const int w4 = width >> 2;
            const float* src = reinterpret_cast<const float*>(op.p0) + b.rowbase * op.ld;
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                if (r < b.nrows) {
                    const float4 u = *reinterpret_cast<const float4*>(src + (long long)r * op.ld + c);
                    float4* p = reinterpret_cast<float4*>(cur + r * b.ldw + c);
                    float4 v = *p;
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    *p = v;
                }
            }
            break;
        }
        case RIH_CH_KEEP:
        case RIH_CH_ADD_KEPT: {
            const int w4 = width >> 2;
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                float4* pc = reinterpret_cast<float4*>(cur + r * b.ldw + c);
                float4* pk = reinterpret_cast<float4*>(kept + r * b.ldw + c);
                if (op.kind == RIH_CH_KEEP) {
                    *pk = *pc;
                } else {
                    float4 v = *pc;
                    const float4 u = *pk;
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    *pc = v;
                }
            }
            break;
        }
        case RIH_CH_GEMM: {
            // the fused epilogue's memory operand (residual rows / the mask's saved activation) is requested BEFORE the product
            // and consumed after it: its round trip hides behind the arithmetic
            constexpr int PF = 8;               // 16-byte items per thread: rblk * n / 4 / 256 <= 8
            const bool epi = (op.flags & (RIH_CHF_EPI_DROPOUT | RIH_CHF_EPI_ADD | RIH_CHF_EPI_ADD_KEPT | RIH_CHF_EPI_STORE |
                                          RIH_CHF_EPI_KEEP | RIH_CHF_EPI_MASKNZ)) != 0;
            const bool eload = (op.flags & (RIH_CHF_EPI_ADD | RIH_CHF_EPI_MASKNZ)) != 0;
            const int n4 = op.n >> 2;
            f4 pf[PF];
            if (eload) {
                const float* src = reinterpret_cast<const float*>(op.p4) + b.rowbase * op.lde;
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    const int idx = tid + TPB * t, r = idx / n4, c = (idx - r * n4) << 2;
                    pf[t] = f4{0.f, 0.f, 0.f, 0.f};
                    if (idx < b.R * n4 && r < b.nrows) pf[t] = *reinterpret_cast<const f4*>(src + (long long)r * op.lde + c);
                }
            }
            if (op.flags & RIH_CHF_A_GLOBAL) ch_gemm<true, true, NRB, KC>(op, cur, nxt, wtile, b, tid);
            else if (op.flags & RIH_CHF_BT) ch_gemm<false, true, NRB, KC>(op, cur, nxt, wtile, b, tid);
            else ch_gemm<false, false, NRB, KC>(op, cur, nxt, wtile, b, tid);
            if (epi) {
                __syncthreads();
                const uint64_t key = rih_seed_key(op.seed + seed_add);
                const uint32_t thr = ch_thresh(op.f0);
                const float ks = 1.f / (1.f - op.f0);
                float* dst = (op.flags & RIH_CHF_EPI_STORE) ? reinterpret_cast<float*>(op.p2) + b.rowbase * op.ld : nullptr;
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    const int idx = tid + TPB * t, r = idx / n4, c = (idx - r * n4) << 2;
                    if (idx < b.R * n4) {
                        f4* pn = reinterpret_cast<f4*>(nxt + r * b.ldw + c);
                        f4 v = *pn;
                        if (op.flags & RIH_CHF_EPI_MASKNZ) {
                            v.x = pf[t].x != 0.f ? v.x * op.f1 : 0.f;
                            v.y = pf[t].y != 0.f ? v.y * op.f1 : 0.f;
                            v.z = pf[t].z != 0.f ? v.z * op.f1 : 0.f;
                            v.w = pf[t].w != 0.f ? v.w * op.f1 : 0.f;
                        }
                        if (op.flags & RIH_CHF_EPI_DROPOUT) {
                            const uint64_t e = (uint64_t)(b.rowbase + r) * (uint64_t)op.n + (uint64_t)c;
                            v.x = rih_hash_k64(key, e) >= thr ? v.x * ks : 0.f;
                            v.y = rih_hash_k64(key, e + 1) >= thr ? v.y * ks : 0.f;
                            v.z = rih_hash_k64(key, e + 2) >= thr ? v.z * ks : 0.f;
                            v.w = rih_hash_k64(key, e + 3) >= thr ? v.w * ks : 0.f;
                        }
                        if (op.flags & RIH_CHF_EPI_ADD) v += pf[t];
                        if (op.flags & RIH_CHF_EPI_ADD_KEPT) v += *reinterpret_cast<const f4*>(kept + r * b.ldw + c);
                        if (dst != nullptr && r < b.nrows) *reinterpret_cast<f4*>(dst + (long long)r * op.ld + c) = v;
                        if (op.flags & RIH_CHF_EPI_KEEP) *reinterpret_cast<f4*>(kept + r * b.ldw + c) = v;
                        *pn = v;
                    }
                }
            }
            if (!(op.flags & RIH_CHF_OUT_GLOBAL)) {     // the result is the new running activation
                float* t = cur;
                cur = nxt;
                nxt = t;
            }
            if (!(op.flags & RIH_CHF_OUT_GLOBAL)) width = op.n;
            break;
        }
        case RIH_CH_DROPOUT: {
            const int w4 = width >> 2;
            const uint64_t key = rih_seed_key(op.seed + seed_add);
            const uint32_t thr = ch_thresh(op.f0);
            const float ks = 1.f / (1.f - op.f0);
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                const uint64_t e = (uint64_t)(b.rowbase + r) * (uint64_t)width + (uint64_t)c;
                float4* p = reinterpret_cast<float4*>(cur + r * b.ldw + c);
                float4 v = *p;
                v.x = rih_hash_k64(key, e) >= thr ? v.x * ks : 0.f;
                v.y = rih_hash_k64(key, e + 1) >= thr ? v.y * ks : 0.f;
                v.z = rih_hash_k64(key, e + 2) >= thr ? v.z * ks : 0.f;
                v.w = rih_hash_k64(key, e + 3) >= thr ? v.w * ks : 0.f;
                *p = v;
            }
            break;
        }
        case RIH_CH_MASKNZ: {
            const int w4 = width >> 2;
            const float* src = reinterpret_cast<const float*>(op.p0) + b.rowbase * op.ld;
            for (int idx = tid; idx < b.R * w4; idx += TPB) {
                const int r = idx / w4, c = (idx - r * w4) << 2;
                float4* p = reinterpret_cast<float4*>(cur + r * b.ldw + c);
                float4 v = *p, u = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < b.nrows) u = *reinterpret_cast<const float4*>(src + (long long)r * op.ld + c);
                v.x = u.x != 0.f ? v.x * op.f0 : 0.f;
                v.y = u.y != 0.f ? v.y * op.f0 : 0.f;
                v.z = u.z != 0.f ? v.z * op.f0 : 0.f;
                v.w = u.w != 0.f ? v.w * op.f0 : 0.f;
                *p = v;
            }
            break;
        }
        case RIH_CH_LN:
            ch_ln(op, cur, b, tid, width);
            break;
        case RIH_CH_LN_BWD:
            ch_ln_bwd(op, cur, stat, b, tid, width, nblk);
            break;
        default:
            break;
        }
        __syncthreads();
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Static walk of the program: widths, buffer needs, pointer / pitch requirements.  0 = launchable; otherwise -(100 * (index of
// the offending operator + 1) + reason), reason 1 = pointer missing / misaligned, 2 = width or pitch, 3 = operator order,
// 4 = does not fit the LDS block; -1 = header fields.  *keeps_out = the program uses the second buffer.
static int chain_check(const rih_chain_desc& d, bool* keeps_out) {
    if (d.nops < 1 || d.nops > RIH_CHAIN_MAXOPS || d.rows < 1 || d.nhands < 1 || d.nhands > 65535) return RIH_EINVAL;
    if ((d.rblk != 32 && d.rblk != 64) || d.ldw < 8 || d.ldw % 4 != 0) return RIH_EINVAL;
    int width = 0;
    bool keeps = false;
#define CH_BAD(reason) return -(100 * (i + 1) + (reason))
    for (int i = 0; i < d.nops; ++i) {
        const rih_chain_op& op = d.op[i];
        switch (op.kind) {
        case RIH_CH_LOAD:
            if (!op.p0 || !al16(op.p0)) CH_BAD(1);
            if (op.n < 4 || op.n % 4 != 0 || op.ld < op.n || op.ld % 4 != 0) CH_BAD(2);
            width = op.n;
            break;
        case RIH_CH_STORE:
        case RIH_CH_ADD:
        case RIH_CH_MASKNZ:
            if (width == 0) CH_BAD(3);
            if (!op.p0 || !al16(op.p0)) CH_BAD(1);
            if (op.ld < width || op.ld % 4 != 0) CH_BAD(2);
            break;
        case RIH_CH_KEEP:
            if (width == 0) CH_BAD(3);
            keeps = true;
            break;
        case RIH_CH_ADD_KEPT:
            if (!keeps || width == 0) CH_BAD(3);
            break;
        case RIH_CH_GEMM: {
            if (op.flags & RIH_CHF_A_GLOBAL) {
                if (!(op.flags & RIH_CHF_BT)) CH_BAD(3);       // (only the data-gradient form exists)
                if (!op.p3 || !al16(op.p3)) CH_BAD(1);
                if (op.lda < op.k || op.lda % 4 != 0) CH_BAD(2);
            } else {
                if (width == 0) CH_BAD(3);
                if (op.k != width) CH_BAD(2);
            }
            if (!op.p0 || !al16(op.p0) || op.s0 % 4 != 0) CH_BAD(1);
            if (op.k % 64 != 0 || op.n < 32 || op.n % 32 != 0) CH_BAD(2);
            if (op.flags & RIH_CHF_OUT_GLOBAL) {
                if (!op.p2) CH_BAD(1);
                if (op.ld < op.n) CH_BAD(2);
                if (op.flags & (RIH_CHF_EPI_DROPOUT | RIH_CHF_EPI_ADD | RIH_CHF_EPI_ADD_KEPT | RIH_CHF_EPI_STORE | RIH_CHF_EPI_KEEP |
                                RIH_CHF_EPI_MASKNZ))
                    CH_BAD(3);
            } else {
                if ((op.flags & RIH_CHF_EPI_ADD) && (op.flags & RIH_CHF_EPI_MASKNZ)) CH_BAD(3);
                if (op.flags & (RIH_CHF_EPI_ADD | RIH_CHF_EPI_MASKNZ)) {
                    if (!op.p4) CH_BAD(1);
                    if (op.lde < op.n) CH_BAD(2);
                }
                if (op.flags & RIH_CHF_EPI_STORE) {
                    if (!op.p2) CH_BAD(1);
                    if (op.ld < op.n) CH_BAD(2);
                }
                if ((op.flags & RIH_CHF_EPI_DROPOUT) && !(op.f0 >= 0.f && op.f0 < 1.f)) CH_BAD(2);
                if ((op.flags & RIH_CHF_EPI_ADD_KEPT) && !keeps) CH_BAD(3);
                if (op.flags & RIH_CHF_EPI_KEEP) keeps = true;
                width = op.n;
            }
            break;
        }
        case RIH_CH_DROPOUT:
            if (width == 0) CH_BAD(3);
            if (!(op.f0 >= 0.f && op.f0 < 1.f)) CH_BAD(2);
            break;
        case RIH_CH_LN:
            if (width == 0) CH_BAD(3);
            if (!op.p0 || !op.p1 || ((op.p2 == nullptr) != (op.p3 == nullptr)) || !al16(op.p0) || !al16(op.p1) ||
                op.s0 % 4 != 0 || op.s1 % 4 != 0)
                CH_BAD(1);
            break;
        case RIH_CH_LN_BWD:
            if (width == 0) CH_BAD(3);
            if (!op.p0 || !op.p1 || !op.p2 || !op.p3 || !op.p4 || !al16(op.p0) || !al16(op.p3) || op.s3 % 4 != 0) CH_BAD(1);
            if (op.ld < width || op.ld % 4 != 0) CH_BAD(2);
            break;
        default:
            CH_BAD(3);
        }
        if (width + 4 > d.ldw) CH_BAD(4);
    }
#undef CH_BAD
    const int need = d.rblk * d.ldw;
    if (need > CAP_BIG) return -4;
    *keeps_out = keeps;
    return 0;
}

}  // namespace

extern "C" int rih_chain_check(const rih_chain_desc* desc) {
    if (!desc) return RIH_EINVAL;
    bool keeps = false;
    return chain_check(*desc, &keeps);
}

extern "C" int rih_chain(const rih_chain_desc* desc, void* stream) {
    if (!desc) return RIH_EINVAL;
    const rih_chain_desc& d = *desc;
    bool keeps = false;
    const int bad = chain_check(d, &keeps);
    if (bad != 0) return bad;
    const int need = d.rblk * d.ldw;
    const dim3 grid((d.rows + d.rblk - 1) / d.rblk, d.nhands), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    // small blocks: 32-deep weight chunks (18 KB of tiles) keep two workgroups on a CU; big ones 64-deep
#define RIH_CHAIN_LAUNCH(C_, K_, KC_)                                                                     \
    {                                                                                                    \
        if (d.rblk == 32) hipLaunchKernelGGL((chain_kernel<C_, K_, KC_, 1>), grid, block, 0, s, d);       \
        else hipLaunchKernelGGL((chain_kernel<C_, K_, KC_, 2>), grid, block, 0, s, d);                    \
    }
    if (!keeps) {
        if (need <= CAP_SMALL) RIH_CHAIN_LAUNCH(CAP_SMALL, 0, 32)
        else RIH_CHAIN_LAUNCH(CAP_BIG, 0, 64)
    } else {
        if (need <= CAP_SMALL) RIH_CHAIN_LAUNCH(CAP_SMALL, CAP_SMALL, 32)
        else RIH_CHAIN_LAUNCH(CAP_BIG, CAP_BIG, 64)
    }
#undef RIH_CHAIN_LAUNCH
    return (int)hipGetLastError();
}
