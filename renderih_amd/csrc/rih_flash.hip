// rih_flash.hip -- attention of the mesh decoder (models/model_attn/self_attn.py:70-76, inter_attn.py:93-107) for gfx950
// WITHOUT a score matrix in memory: out = softmax(alpha q k^T) [dropout] v per (image, head), forward in one launch, backward in
// two.  The unfused path (ops.py: batched QK^T GEMM -> softmax kernel -> batched PV GEMM; five launches backward) writes and
// re-reads [B][heads][Sq][Sk] probabilities four times per direction -- 204 MB per tensor at the finest level (Sq = Sk = 316,
// 512 (image, hand, head) slices) -- and its K = 16..64 products run the 64x64-tile GEMM at 20 TF.  Here:
//
//   * forward: a wavefront owns 32 queries; keys / values stream through LDS in tiles of 32 (shared by the workgroup's four
//     wavefronts); the score tile is computed TRANSPOSED (S^T = K Q^T: MFMA rows = keys, columns = queries), so that a lane
//     holds ONE query's column: the running row maximum / sum of the online softmax are per-lane scalars, reduced over the 16
//     accumulator registers in-lane plus ONE xor-32 shuffle per tile (the row-major layout needs 80 shuffles per tile), and the
//     rescale of the output accumulator is a per-lane multiply.  P^T goes through a per-wavefront LDS tile into the operand
//     layout of O^T += V^T Pd^T.  Saved for the backward: the output and one log-sum-exp word per query -- no P, no Pd.
//   * backward, query side (dq): recomputes S^T and dP^T = V dO^T per key tile, forms dS^T = alpha P^T o (mask dP^T - D) with
//     D = rowsum(dO o O) (per lane), accumulates dQ^T += K^T dS^T.  Also writes D for the key-side kernel.
//   * backward, key side (dk, dv): a wavefront owns 32 keys and walks the query tiles: S = Q K^T and dP = dO V^T with the keys
//     as MFMA columns (K, V rows preloaded in registers), P / dS through LDS into dV += Pd^T dO, dK += dS^T Q.
//   Dropout masks are recomputed from the counter-based hash of rih_softmax_fwd (same element index: bit-identical masks).
//
// Arithmetic: native f32 MFMA (v_mfma_f32_32x32x2_f32) -- exact fp32 products like the reference; the products are small
// (<= 2 * 316^2 * 64 per slice), the kernels are bound by exp / LDS / launch count, not by the matrix pipe.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"
#include "rih_hash.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int TPB = 256;
constexpr float LOG2E = 1.4426950408889634f;

// mask word of element idx (rih_hash.h); `key` = rih_seed_key(seed), computed once per thread.  The element count of a score
// tensor (B * heads * Sq * Sk) fits 32 bits for every decoder shape; the 64-bit form is the same function on wider indices.
__device__ __forceinline__ uint32_t fl_hash(uint64_t key, long long idx, bool wide) {
    return wide ? rih_hash_k64(key, (uint64_t)idx) : rih_hash_k32(key, (uint32_t)idx);
}
__device__ __forceinline__ uint32_t fl_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
// accumulator register r of lane (l31, lhi) holds row acc_row(r, lhi), column l31 of a 32x32 MFMA tile
__device__ __forceinline__ int acc_row(int r, int lhi) { return (r & 3) + 8 * (r >> 2) + 4 * lhi; }

// Two [32][DH] operand tiles (rows r0.. of a and of b, row pitches lda / ldb floats, rows >= nrows read as zero) on their way
// into LDS, in two halves so that the global loads of tile j+1 fly during the arithmetic of tile j: `issue` puts them into
// registers (16-byte loads when the operands allow), `commit` writes the registers to the LDS tiles.
template <int DH>
struct TileRegs {
    static constexpr int NV = (32 * DH / 4 + TPB - 1) / TPB;        // float4 slots per thread and operand
    float4 a[NV], b[NV];
};
template <int DH>
__device__ __forceinline__ void issue_tiles(TileRegs<DH>& t, const float* __restrict__ a, long long lda,
                                            const float* __restrict__ b, long long ldb, int r0, int nrows, int tid, bool vec) {
    constexpr int Q = DH / 4;
#pragma unroll
    for (int i = 0; i < TileRegs<DH>::NV; ++i) {
        const int idx = tid + i * TPB;
        const int rr = idx / Q, c = 4 * (idx - rr * Q), row = r0 + rr;
        t.a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        t.b[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < 32 * Q && row < nrows) {
            const float* pa = a + (long long)row * lda + c;
            const float* pb = b + (long long)row * ldb + c;
            if (vec) {
                t.a[i] = *reinterpret_cast<const float4*>(pa);
                t.b[i] = *reinterpret_cast<const float4*>(pb);
            } else {
                t.a[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
                t.b[i] = make_float4(pb[0], pb[1], pb[2], pb[3]);
            }
        }
    }
}
template <int DH>
__device__ __forceinline__ void commit_tiles(const TileRegs<DH>& t, float (*As)[DH + 1], float (*Bs)[DH + 1], int tid) {
    constexpr int Q = DH / 4;
#pragma unroll
    for (int i = 0; i < TileRegs<DH>::NV; ++i) {
        const int idx = tid + i * TPB;
        if (idx < 32 * Q) {
            const int rr = idx / Q, c = 4 * (idx - rr * Q);
            As[rr][c] = t.a[i].x; As[rr][c + 1] = t.a[i].y; As[rr][c + 2] = t.a[i].z; As[rr][c + 3] = t.a[i].w;
            Bs[rr][c] = t.b[i].x; Bs[rr][c + 1] = t.b[i].y; Bs[rr][c + 2] = t.b[i].z; Bs[rr][c + 3] = t.b[i].w;
        }
    }
}

// Store a transposed accumulator (rows = channels, columns = this lane's query / row index `row`) to dst[row][c0 + channel].
template <int DH, int CT>
__device__ __forceinline__ void store_T(const floatx16 (&o)[CT], float* __restrict__ dst, int lhi, bool vec) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * ct + 8 * g + 4 * lhi;        // channels c .. c+3 = registers 4g .. 4g+3
            if (c >= DH) continue;
            if (vec) {
                *reinterpret_cast<float4*>(dst + c) = make_float4(o[ct][4 * g], o[ct][4 * g + 1], o[ct][4 * g + 2], o[ct][4 * g + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[c + e] = o[ct][4 * g + e];
            }
        }
}

template <int DH>
__global__ __launch_bounds__(TPB) void flash_fwd_kernel(const float* __restrict__ q, int q_ld, const float* __restrict__ k,
                                                        const float* __restrict__ v, int kv_ld, int heads, int Sq, int Sk,
                                                        float alpha, float drop_p, uint64_t seed,
                                                        const uint64_t* __restrict__ seed_dev, float* __restrict__ out,
                                                        int ld_out, float* __restrict__ lse, int vec_out, int vec_in) {
    constexpr int CT = (DH + 31) / 32;
    __shared__ float Ks[32][DH + 1];
    __shared__ float Vs[32][DH + 1];
    __shared__ float Ps[TPB / 64][32][33];
    if (seed_dev != nullptr) seed += *seed_dev;
    const uint64_t hkey = rih_seed_key(seed);
    const bool wide = (long long)gridDim.y * Sq * Sk > 0xffffffffLL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int qr = blockIdx.x * 128 + wave * 32 + l31;          // this lane's query
    const bool qok = qr < Sq;
    const float* qb = q + (long long)b * Sq * q_ld + h * DH;
    const float* kb = k + (long long)b * Sk * kv_ld + h * DH;
    const float* vb = v + (long long)b * Sk * kv_ld + h * DH;
    const int nt = (Sk + 31) / 32;
    const float alpha2 = alpha * LOG2E;
    const long long ridx = (long long)bh * Sq + qr;
    const uint32_t thr = fl_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;

    float qa[DH / 2];           // B operand of S^T = K Q^T: Q[qr][2t + lhi]
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) qa[t] = qok ? qb[(long long)qr * q_ld + 2 * t + lhi] : 0.f;

    float m = -INFINITY, l = 0.f;       // running maximum (log2 domain) and this half-wave's share of the running sum
    floatx16 o[CT];                     // O^T: register r = channel acc_row(r, lhi) + 32 ct of query qr
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;

    TileRegs<DH> tr;
    issue_tiles<DH>(tr, kb, kv_ld, vb, kv_ld, 0, Sk, tid, vec_in != 0);
    for (int j = 0; j < nt; ++j) {
        __syncthreads();                                        // the previous tile's operands have been consumed
        commit_tiles<DH>(tr, Ks, Vs, tid);
        __syncthreads();
        if (j + 1 < nt) issue_tiles<DH>(tr, kb, kv_ld, vb, kv_ld, 32 * (j + 1), Sk, tid, vec_in != 0);   // in flight during tile j
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31][2 * t + lhi], qa[t], acc, 0, 0, 0);
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = 32 * j + acc_row(r, lhi);
            acc[r] = (key < Sk) ? acc[r] * alpha2 : -INFINITY;
            tmax = fmaxf(tmax, acc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);                     // finite: key 32 j is always valid
        const float corr = exp2f(m - m_new);                    // 0 on the first tile (m = -inf)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = acc_row(r, lhi);
            float p = exp2f(acc[r] - m_new);                    // 0 for keys past Sk
            psum += p;
            if (drop_p > 0.f) {
                const bool keep = fl_hash(hkey, ridx * Sk + 32 * j + kr, wide) >= thr;
                p = keep ? p * keep_scale : 0.f;
            }
            Ps[wave][kr][l31] = p;
        }
        l = l * corr + psum;
        m = m_new;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][r] *= corr;
        __syncthreads();                                        // Pd^T tile visible to the wavefront's other lanes
#pragma unroll
        for (int t = 0; t < 16; ++t) {      // A: V^T[channel l31 (+32 ct)][key 2t + lhi];  B: Pd^T[key 2t + lhi][query l31]
            const float pb = Ps[wave][2 * t + lhi][l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = l31 + 32 * ct;
                const float av = (c < DH) ? Vs[2 * t + lhi][c] : 0.f;
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, pb, o[ct], 0, 0, 0);
            }
        }
    }
    const float ltot = l + __shfl_xor(l, 32, 64);
    const float inv = 1.f / ltot;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= inv;
    if (qok) {
        store_T<DH, CT>(o, out + ((long long)b * Sq + qr) * ld_out + h * DH, lhi, vec_out != 0);
        if (lhi == 0) lse[ridx] = m + log2f(ltot);              // log2 of the row's sum of 2^(alpha2 s)
    }
}

// Backward, query side.  dq[q][c] = sum_key dS[q][key] K[key][c];  also D[q] = sum_c dO[q][c] O[q][c] for the key-side kernel.
template <int DH>
__global__ __launch_bounds__(TPB) void flash_bwd_dq_kernel(const float* __restrict__ dO, int do_ld, const float* __restrict__ O,
                                                           int o_ld, const float* __restrict__ q, int q_ld,
                                                           const float* __restrict__ k, const float* __restrict__ v, int kv_ld,
                                                           int heads, int Sq, int Sk, float alpha, float drop_p, uint64_t seed,
                                                           const uint64_t* __restrict__ seed_dev, const float* __restrict__ lse,
                                                           float* __restrict__ Dout, float* __restrict__ dq, int dq_ld,
                                                           int vec_out, int vec_in) {
    constexpr int CT = (DH + 31) / 32;
    __shared__ float Ks[32][DH + 1];
    __shared__ float Vs[32][DH + 1];
    __shared__ float Ps[TPB / 64][32][33];
    if (seed_dev != nullptr) seed += *seed_dev;
    const uint64_t hkey = rih_seed_key(seed);
    const bool wide = (long long)gridDim.y * Sq * Sk > 0xffffffffLL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int qr = blockIdx.x * 128 + wave * 32 + l31;
    const bool qok = qr < Sq;
    const float* qb = q + (long long)b * Sq * q_ld + h * DH;
    const float* dob = dO + (long long)b * Sq * do_ld + h * DH;
    const float* ob = O + (long long)b * Sq * o_ld + h * DH;
    const float* kb = k + (long long)b * Sk * kv_ld + h * DH;
    const float* vb = v + (long long)b * Sk * kv_ld + h * DH;
    const int nt = (Sk + 31) / 32;
    const float alpha2 = alpha * LOG2E;
    const long long ridx = (long long)bh * Sq + qr;
    const uint32_t thr = fl_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;

    float qa[DH / 2], da[DH / 2];
    float dpart = 0.f;
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) {
        qa[t] = qok ? qb[(long long)qr * q_ld + 2 * t + lhi] : 0.f;
        da[t] = qok ? dob[(long long)qr * do_ld + 2 * t + lhi] : 0.f;
        dpart += da[t] * (qok ? ob[(long long)qr * o_ld + 2 * t + lhi] : 0.f);
    }
    const float D = dpart + __shfl_xor(dpart, 32, 64);
    const float L2 = qok ? lse[ridx] : 0.f;
    if (qok && lhi == 0) Dout[ridx] = D;

    floatx16 g[CT];                     // dQ^T
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[ct][r] = 0.f;

    TileRegs<DH> tr;
    issue_tiles<DH>(tr, kb, kv_ld, vb, kv_ld, 0, Sk, tid, vec_in != 0);
    for (int j = 0; j < nt; ++j) {
        __syncthreads();
        commit_tiles<DH>(tr, Ks, Vs, tid);
        __syncthreads();
        if (j + 1 < nt) issue_tiles<DH>(tr, kb, kv_ld, vb, kv_ld, 32 * (j + 1), Sk, tid, vec_in != 0);
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31][2 * t + lhi], qa[t], s, 0, 0, 0);        // S^T = K Q^T
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[l31][2 * t + lhi], da[t], dp, 0, 0, 0);      // dPd^T = V dO^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = acc_row(r, lhi);
            const int key = 32 * j + kr;
            const float p = (key < Sk) ? exp2f(s[r] * alpha2 - L2) : 0.f;
            float d = dp[r];
            if (drop_p > 0.f) d = (fl_hash(hkey, ridx * Sk + key, wide) >= thr) ? d * keep_scale : 0.f;
            Ps[wave][kr][l31] = alpha * p * (d - D);            // dS^T
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {      // A: K^T[channel][key 2t + lhi];  B: dS^T[key 2t + lhi][query l31]
            const float sb = Ps[wave][2 * t + lhi][l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = l31 + 32 * ct;
                const float ak = (c < DH) ? Ks[2 * t + lhi][c] : 0.f;
                g[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, sb, g[ct], 0, 0, 0);
            }
        }
    }
    if (qok) store_T<DH, CT>(g, dq + ((long long)b * Sq + qr) * dq_ld + h * DH, lhi, vec_out != 0);
}

// Backward, key side: dv[key][c] = sum_q Pd[q][key] dO[q][c],  dk[key][c] = sum_q dS[q][key] Q[q][c].
template <int DH>
__global__ __launch_bounds__(TPB) void flash_bwd_dkv_kernel(const float* __restrict__ dO, int do_ld, const float* __restrict__ q,
                                                            int q_ld, const float* __restrict__ k, const float* __restrict__ v,
                                                            int kv_ld, int heads, int Sq, int Sk, float alpha, float drop_p,
                                                            uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                            const float* __restrict__ lse, const float* __restrict__ Din,
                                                            float* __restrict__ dk, float* __restrict__ dv, int dkv_ld,
                                                            int vec_in) {
    constexpr int CT = (DH + 31) / 32;
    __shared__ float Qs[32][DH + 1];
    __shared__ float Os[32][DH + 1];
    __shared__ float Ps[TPB / 64][32][33];
    __shared__ float Ss[TPB / 64][32][33];
    __shared__ float Ls[32], Ds[32];
    if (seed_dev != nullptr) seed += *seed_dev;
    const uint64_t hkey = rih_seed_key(seed);
    const bool wide = (long long)gridDim.y * Sq * Sk > 0xffffffffLL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int key0 = blockIdx.x * 128 + wave * 32;
    const int key = key0 + l31;                                 // this lane's key (column of S = Q K^T)
    const bool kok = key < Sk;
    const float* qb = q + (long long)b * Sq * q_ld + h * DH;
    const float* dob = dO + (long long)b * Sq * do_ld + h * DH;
    const float* kb = k + (long long)b * Sk * kv_ld + h * DH;
    const float* vb = v + (long long)b * Sk * kv_ld + h * DH;
    const float alpha2 = alpha * LOG2E;
    const uint32_t thr = fl_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;

    float ka[DH / 2], va[DH / 2];       // B operands: K^T[channel 2t + lhi][key], V^T[...][key]
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) {
        ka[t] = kok ? kb[(long long)key * kv_ld + 2 * t + lhi] : 0.f;
        va[t] = kok ? vb[(long long)key * kv_ld + 2 * t + lhi] : 0.f;
    }
    floatx16 gv[CT], gk[CT];            // dV, dK: register r = key key0 + acc_row(r, lhi), channel l31 + 32 ct
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gv[ct][r] = 0.f; gk[ct][r] = 0.f; }

    TileRegs<DH> tr;
    float lreg = 0.f, dreg = 0.f;       // threads 0..31: the tile's log-sum-exp / D words, prefetched like the operand tiles
    auto issue_rows = [&](int q0) {
        if (tid < 32) {
            const bool ok = q0 + tid < Sq;
            lreg = ok ? lse[(long long)bh * Sq + q0 + tid] : 0.f;
            dreg = ok ? Din[(long long)bh * Sq + q0 + tid] : 0.f;
        }
    };
    issue_tiles<DH>(tr, qb, q_ld, dob, do_ld, 0, Sq, tid, vec_in != 0);
    issue_rows(0);
    for (int q0 = 0; q0 < Sq; q0 += 32) {
        __syncthreads();
        commit_tiles<DH>(tr, Qs, Os, tid);
        if (tid < 32) { Ls[tid] = lreg; Ds[tid] = dreg; }
        __syncthreads();
        if (q0 + 32 < Sq) {
            issue_tiles<DH>(tr, qb, q_ld, dob, do_ld, q0 + 32, Sq, tid, vec_in != 0);
            issue_rows(q0 + 32);
        }
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) {
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[l31][2 * t + lhi], ka[t], s, 0, 0, 0);        // S = Q K^T
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Os[l31][2 * t + lhi], va[t], dp, 0, 0, 0);      // dPd = dO V^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qrow = acc_row(r, lhi);
            const int qq = q0 + qrow;
            const bool ok = qq < Sq && kok;
            float p = ok ? exp2f(s[r] * alpha2 - Ls[qrow]) : 0.f;
            float d = dp[r];
            float pd = p;
            if (drop_p > 0.f) {
                const bool keep = fl_hash(hkey, ((long long)bh * Sq + qq) * Sk + key, wide) >= thr;
                pd = keep ? p * keep_scale : 0.f;
                d = keep ? d * keep_scale : 0.f;
            }
            Ps[wave][qrow][l31] = pd;
            Ss[wave][qrow][l31] = alpha * p * (d - Ds[qrow]);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t) {      // A: Pd^T / dS^T [key l31][query 2t + lhi];  B: dO / Q [query 2t + lhi][channel]
            const float ap = Ps[wave][2 * t + lhi][l31];
            const float as = Ss[wave][2 * t + lhi][l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = l31 + 32 * ct;
                const float bo = (c < DH) ? Os[2 * t + lhi][c] : 0.f;
                const float bq = (c < DH) ? Qs[2 * t + lhi][c] : 0.f;
                gv[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap, bo, gv[ct], 0, 0, 0);
                gk[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bq, gk[ct], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = l31 + 32 * ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = key0 + acc_row(r, lhi);
            if (kr < Sk && c < DH) {
                const long long at = ((long long)b * Sk + kr) * dkv_ld + h * DH + c;
                dv[at] = gv[ct][r];
                dk[at] = gk[ct][r];
            }
        }
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int rih_flash_attention_fwd(const float* q, int q_ld, const float* k, const float* v, int kv_ld, int B, int heads,
                                       int Sq, int Sk, int d, float alpha, float drop_p, uint64_t seed,
                                       const uint64_t* seed_dev, float* out, int ld_out, float* lse, void* stream) {
    if (!q || !k || !v || !out || !lse || B < 1 || heads < 1 || Sq < 1 || Sk < 1) return RIH_EINVAL;
    if (q_ld < d || kv_ld < d || ld_out < heads * d || drop_p < 0.f || drop_p >= 1.f) return RIH_EINVAL;
    if ((long long)B * heads > 65535) return RIH_EINVAL;
    const dim3 grid((Sq + 127) / 128, B * heads), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    const int vec = (al16(out) && ld_out % 4 == 0 && d % 4 == 0) ? 1 : 0;
    const int vin = (al16(k) && al16(v) && kv_ld % 4 == 0) ? 1 : 0;
#define RIH_FL(D_) hipLaunchKernelGGL((flash_fwd_kernel<D_>), grid, block, 0, s, q, q_ld, k, v, kv_ld, heads, Sq, Sk, alpha,   \
                                      drop_p, seed, seed_dev, out, ld_out, lse, vec, vin)
    if (d == 16) RIH_FL(16);
    else if (d == 32) RIH_FL(32);
    else if (d == 64) RIH_FL(64);
    else return RIH_EINVAL;
#undef RIH_FL
    return (int)hipGetLastError();
}

extern "C" int rih_flash_attention_bwd(const float* dO, int do_ld, const float* O, int o_ld, const float* q, int q_ld,
                                       const float* k, const float* v, int kv_ld, int B, int heads, int Sq, int Sk, int d,
                                       float alpha, float drop_p, uint64_t seed, const uint64_t* seed_dev, const float* lse,
                                       float* Dws, float* dq, int dq_ld, float* dk, float* dv, int dkv_ld, void* stream) {
    if (!dO || !O || !q || !k || !v || !lse || !Dws || !dq || !dk || !dv || B < 1 || heads < 1 || Sq < 1 || Sk < 1)
        return RIH_EINVAL;
    if (q_ld < d || kv_ld < d || do_ld < heads * d || o_ld < heads * d || dq_ld < d || dkv_ld < d) return RIH_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || (long long)B * heads > 65535) return RIH_EINVAL;
    const dim3 gq((Sq + 127) / 128, B * heads), gk((Sk + 127) / 128, B * heads), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    const int vec = (al16(dq) && dq_ld % 4 == 0 && d % 4 == 0) ? 1 : 0;
    const int vkv = (al16(k) && al16(v) && kv_ld % 4 == 0) ? 1 : 0;
    const int vqo = (al16(q) && al16(dO) && q_ld % 4 == 0 && do_ld % 4 == 0) ? 1 : 0;
#define RIH_FL(D_)                                                                                                             \
    {                                                                                                                          \
        hipLaunchKernelGGL((flash_bwd_dq_kernel<D_>), gq, block, 0, s, dO, do_ld, O, o_ld, q, q_ld, k, v, kv_ld, heads, Sq, Sk,  \
                           alpha, drop_p, seed, seed_dev, lse, Dws, dq, dq_ld, vec, vkv);                                      \
        hipLaunchKernelGGL((flash_bwd_dkv_kernel<D_>), gk, block, 0, s, dO, do_ld, q, q_ld, k, v, kv_ld, heads, Sq, Sk, alpha,   \
                           drop_p, seed, seed_dev, lse, Dws, dk, dv, dkv_ld, vqo);                                             \
    }
    if (d == 16) RIH_FL(16)
    else if (d == 32) RIH_FL(32)
    else if (d == 64) RIH_FL(64)
    else return RIH_EINVAL;
#undef RIH_FL
    return (int)hipGetLastError();
}
