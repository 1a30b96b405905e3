// rih_attn.hip -- fused attention forward for gfx950: softmax(alpha Q K^T) [dropout] V for one (image, head) and a block
// of 128 query rows per workgroup, without the score matrix making a round trip through memory.
//
// Today (renderih_amd/ops.py::_attn_forward) this is three launches -- a batched QK^T GEMM that writes S, a softmax
// kernel that reads S and writes P (and the dropped copy Pd), a batched PV GEMM that reads Pd.  Here the 32 x Sk scores of
// a wavefront's 32 query rows stay in registers (Sk <= 320: ten MFMA accumulator tiles), the row softmax is done on the
// accumulator layout with 5 xor-shuffles per reduction, P / Pd are written once (the backward kernels of ops.py still
// consume them), and P V runs from registers through a small per-wavefront LDS transpose.  K and V are streamed in
// 32-key tiles through LDS, shared by the four wavefronts.  fp32 MFMA (v_mfma_f32_32x32x2_f32): the products are tiny
// (<= 4 MFLOP per workgroup), the kernel is latency / traffic bound.
//
// The query side of the backward (dO V^T -> softmax backward -> dS K) has the same shape and is the second kernel of
// this file; the key-side products (dV = Pd^T dO, dK = dS^T Q) reduce over query rows, i.e. across those workgroups, and
// are a third kernel blocked over keys: 2 launches for the backward instead of 5.
//
// STATUS: written after the round's GPU budget was spent; verified on the HIP-on-CPU harness
// (tests/test_kernels_on_cpu.py::test_fused_attention_forward) against the unfused path and torch (outputs, gradients,
// dropout masks); not yet run or measured on a GPU, therefore OFF by default (ops.FUSED_ATTN / RIH_FUSED_ATTN=1).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"
#include "rih_hash.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int TPB = 256;
constexpr int NTMAX = 10;           // key tiles of 32: Sk <= 320

__device__ __forceinline__ uint32_t attn_hash(uint64_t seed, uint64_t idx) { return rih_hash(seed, idx); }
__device__ __forceinline__ uint32_t attn_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}
// reductions over the 32 lanes that hold one accumulator row (lanes with equal lane>>5)
__device__ __forceinline__ float row_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int DH>
__global__ __launch_bounds__(TPB) void attn_fwd_fused_kernel(const float* __restrict__ q, int q_ld,
                                                             const float* __restrict__ k, const float* __restrict__ v,
                                                             int kv_ld, int heads, int Sq, int Sk, float alpha,
                                                             float drop_p, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_dev, float* __restrict__ P,
                                                             float* __restrict__ Pd, int ldP, float* __restrict__ out,
                                                             int ld_out) {
    constexpr int CT = (DH + 31) / 32;              // 32-wide column tiles of the output
    __shared__ float Ks[32][DH + 1];
    __shared__ float Vs[32][DH + 1];
    __shared__ float Ps[TPB / 64][32][33];
    if (seed_dev != nullptr) seed += *seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float* qb = q + (long long)b * Sq * q_ld + h * DH;
    const float* kb = k + (long long)b * Sk * kv_ld + h * DH;
    const float* vb = v + (long long)b * Sk * kv_ld + h * DH;
    const int nt = (Sk + 31) / 32;

    // A operand of Q K^T: lane l supplies Q[q0 + l%32][2t + l/32]
    float qa[DH / 2];
    {
        const int qr = q0 + l31;
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) qa[t] = (qr < Sq) ? qb[(long long)qr * q_ld + 2 * t + lhi] : 0.f;
    }

    // ---- pass 1: scores of the wavefront's 32 rows against every key tile
    floatx16 s[NTMAX];
#pragma unroll
    for (int j = 0; j < NTMAX; ++j) {
        if (j < nt) {                               // uniform over the block
            __syncthreads();
            for (int i = tid; i < 32 * DH; i += TPB) {
                const int kr = i / DH, c = i - kr * DH, key = 32 * j + kr;
                Ks[kr][c] = (key < Sk) ? kb[(long long)key * kv_ld + c] : 0.f;
            }
            __syncthreads();
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < DH / 2; ++t)        // B operand: lane l supplies K[key l%32][channel 2t + l/32]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[t], Ks[l31][2 * t + lhi], acc, 0, 0, 0);
            s[j] = acc;
        }
    }

    // ---- row softmax on the accumulator layout: register r of lane l = row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
    float mx[16], sm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mx[r] = -INFINITY;
#pragma unroll
    for (int j = 0; j < NTMAX; ++j)
        if (j < nt) {
            const bool valid = 32 * j + l31 < Sk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = valid ? s[j][r] * alpha : -INFINITY;
                s[j][r] = x;
                mx[r] = fmaxf(mx[r], x);
            }
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) { mx[r] = row_max(mx[r]); sm[r] = 0.f; }
#pragma unroll
    for (int j = 0; j < NTMAX; ++j)
        if (j < nt) {
            const bool valid = 32 * j + l31 < Sk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = valid ? expf(s[j][r] - mx[r]) : 0.f;
                s[j][r] = e;
                sm[r] += e;
            }
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[r] = 1.f / row_sum(sm[r]);

    // probabilities out (P for the backward, Pd = dropped copy that multiplies V); s[] then holds Pd
    const uint32_t thr = attn_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll
    for (int j = 0; j < NTMAX; ++j)
        if (j < nt) {
            const int col = 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float pr = s[j][r] * sm[r];
                if (row < Sq && col < Sk) {
                    const long long ridx = (long long)bh * Sq + row;
                    P[ridx * ldP + col] = pr;
                    if (drop_p > 0.f) {
                        const bool keep = attn_hash(seed, (uint64_t)(ridx * Sk + col)) >= thr;
                        pr = keep ? pr * keep_scale : 0.f;
                        Pd[ridx * ldP + col] = pr;
                    } else if (Pd != P) {
                        Pd[ridx * ldP + col] = pr;
                    }
                } else {
                    pr = 0.f;
                }
                s[j][r] = pr;
            }
        }

    // ---- pass 2: O = Pd V, key tile by key tile; Pd goes through LDS to turn accumulator layout into operand layout
    floatx16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
#pragma unroll
    for (int j = 0; j < NTMAX; ++j) {
        if (j < nt) {
            __syncthreads();
            for (int i = tid; i < 32 * DH; i += TPB) {
                const int kr = i / DH, c = i - kr * DH, key = 32 * j + kr;
                Vs[kr][c] = (key < Sk) ? vb[(long long)key * kv_ld + c] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Ps[wave][(r & 3) + 8 * (r >> 2) + 4 * lhi][l31] = s[j][r];
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {          // A: Pd[row l%32][key 2t + l/32];  B: V[key 2t + l/32][channel l%32 (+32 ct)]
                const float a = Ps[wave][l31][2 * t + lhi];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = l31 + 32 * ct;
                    const float bval = (c < DH) ? Vs[2 * t + lhi][c] : 0.f;
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bval, o[ct], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = l31 + 32 * ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (row < Sq && c < DH) out[((long long)b * Sq + row) * ld_out + h * DH + c] = o[ct][r];
        }
    }
}

// Backward, query side: for a block of 128 query rows of one (image, head)
//   dPd = dO V^T  ->  d = mask * dPd,  D = rowsum(d * P),  dS = alpha * P * (d - D)  (= rih_softmax_bwd)  ->  dQ = dS K
// in one launch (today: a batched GEMM that writes dPd, the softmax-backward kernel that rewrites it in place, a
// batched GEMM that reads it).  dS is still written once, for the key-side products dK = dS^T Q and dV = Pd^T dO, which
// remain batched GEMMs.  Same structure as the forward kernel: dO rows are the A operand of pass 1 against V tiles, the
// 32 x Sk tile of dS stays in accumulator registers, pass 2 multiplies it with K tiles through the LDS transpose.
template <int DH>
__global__ __launch_bounds__(TPB) void attn_bwd_dq_fused_kernel(const float* __restrict__ dO, int do_ld,
                                                                const float* __restrict__ k, const float* __restrict__ v,
                                                                int kv_ld, int heads, int Sq, int Sk, float alpha,
                                                                float drop_p, uint64_t seed,
                                                                const uint64_t* __restrict__ seed_dev,
                                                                const float* __restrict__ P, float* __restrict__ dS, int ldP,
                                                                float* __restrict__ dq, int dq_ld) {
    constexpr int CT = (DH + 31) / 32;
    __shared__ float Ts[32][DH + 1];                // V tile in pass 1, K tile in pass 2
    __shared__ float Ps[TPB / 64][32][33];
    if (seed_dev != nullptr) seed += *seed_dev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float* dob = dO + (long long)b * Sq * do_ld + h * DH;
    const float* kb = k + (long long)b * Sk * kv_ld + h * DH;
    const float* vb = v + (long long)b * Sk * kv_ld + h * DH;
    const int nt = (Sk + 31) / 32;
    float da[DH / 2];
    {
        const int qr = q0 + l31;
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) da[t] = (qr < Sq) ? dob[(long long)qr * do_ld + 2 * t + lhi] : 0.f;
    }
    floatx16 s[NTMAX];
#pragma unroll
    for (int j = 0; j < NTMAX; ++j) {
        if (j < nt) {
            __syncthreads();
            for (int i = tid; i < 32 * DH; i += TPB) {
                const int kr = i / DH, c = i - kr * DH, key = 32 * j + kr;
                Ts[kr][c] = (key < Sk) ? vb[(long long)key * kv_ld + c] : 0.f;
            }
            __syncthreads();
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int t = 0; t < DH / 2; ++t)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[t], Ts[l31][2 * t + lhi], acc, 0, 0, 0);
            s[j] = acc;
        }
    }
    // d = mask * dPd, D = rowsum(d * P); P is read again in the second sweep (an L2 hit) rather than held in 160 more
    // registers
    const uint32_t thr = attn_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    float dot[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dot[r] = 0.f;
#pragma unroll
    for (int j = 0; j < NTMAX; ++j)
        if (j < nt) {
            const int col = 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float pr = 0.f, d = 0.f;
                if (row < Sq && col < Sk) {
                    const long long ridx = (long long)bh * Sq + row;
                    pr = P[ridx * ldP + col];
                    d = s[j][r];
                    if (drop_p > 0.f) d = (attn_hash(seed, (uint64_t)(ridx * Sk + col)) >= thr) ? d * keep_scale : 0.f;
                }
                s[j][r] = d;
                dot[r] += d * pr;
            }
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) dot[r] = row_sum(dot[r]);
#pragma unroll
    for (int j = 0; j < NTMAX; ++j)
        if (j < nt) {
            const int col = 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float g = 0.f;
                if (row < Sq && col < Sk) {
                    const long long at = ((long long)bh * Sq + row) * ldP + col;
                    g = alpha * P[at] * (s[j][r] - dot[r]);
                    dS[at] = g;
                }
                s[j][r] = g;
            }
        }
    // dQ = dS K
    floatx16 o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
#pragma unroll
    for (int j = 0; j < NTMAX; ++j) {
        if (j < nt) {
            __syncthreads();
            for (int i = tid; i < 32 * DH; i += TPB) {
                const int kr = i / DH, c = i - kr * DH, key = 32 * j + kr;
                Ts[kr][c] = (key < Sk) ? kb[(long long)key * kv_ld + c] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Ps[wave][(r & 3) + 8 * (r >> 2) + 4 * lhi][l31] = s[j][r];
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float a = Ps[wave][l31][2 * t + lhi];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = l31 + 32 * ct;
                    const float bval = (c < DH) ? Ts[2 * t + lhi][c] : 0.f;
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bval, o[ct], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = l31 + 32 * ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = q0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (row < Sq && c < DH) dq[((long long)b * Sq + row) * dq_ld + h * DH + c] = o[ct][r];
        }
    }
}

// Backward, key side: dV = Pd^T dO and dK = dS^T Q for a block of 128 keys of one (image, head) -- the two products that
// reduce over the query rows.  A wavefront owns 32 keys; the A operands are read straight from the [Sq][ldP] matrices
// (lane = key: 128-byte coalesced rows, 16 query rows per lane batched ahead of the MFMAs), the B operands (dO and Q rows)
// are staged in LDS in blocks of 32 query rows and shared by the four wavefronts.
template <int DH>
__global__ __launch_bounds__(TPB) void attn_bwd_dkv_fused_kernel(const float* __restrict__ dO, int do_ld,
                                                                 const float* __restrict__ q, int q_ld, int heads, int Sq, int Sk,
                                                                 const float* __restrict__ Pd, const float* __restrict__ dS,
                                                                 int ldP, float* __restrict__ dk, float* __restrict__ dv,
                                                                 int dkv_ld) {
    constexpr int CT = (DH + 31) / 32;
    __shared__ float Os[32][DH + 1];
    __shared__ float Qs[32][DH + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int key = blockIdx.x * 128 + wave * 32 + l31;                 // this lane's A-operand row
    const float* dob = dO + (long long)b * Sq * do_ld + h * DH;
    const float* qb = q + (long long)b * Sq * q_ld + h * DH;
    const float* pdb = Pd + (long long)bh * Sq * ldP;
    const float* dsb = dS + (long long)bh * Sq * ldP;
    floatx16 av[CT], ak[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { av[ct][r] = 0.f; ak[ct][r] = 0.f; }
    for (int q0 = 0; q0 < Sq; q0 += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * DH; i += TPB) {
            const int qr = i / DH, c = i - qr * DH, qq = q0 + qr;
            Os[qr][c] = (qq < Sq) ? dob[(long long)qq * do_ld + c] : 0.f;
            Qs[qr][c] = (qq < Sq) ? qb[(long long)qq * q_ld + c] : 0.f;
        }
        float pa[16], sa[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int qq = q0 + 2 * t + lhi;
            const bool ok = qq < Sq && key < Sk;
            pa[t] = ok ? pdb[(long long)qq * ldP + key] : 0.f;
            sa[t] = ok ? dsb[(long long)qq * ldP + key] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = l31 + 32 * ct;
                const float bo = (c < DH) ? Os[2 * t + lhi][c] : 0.f;
                const float bq = (c < DH) ? Qs[2 * t + lhi][c] : 0.f;
                av[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[t], bo, av[ct], 0, 0, 0);
                ak[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[t], bq, ak[ct], 0, 0, 0);
            }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = l31 + 32 * ct;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = blockIdx.x * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (kr < Sk && c < DH) {
                const long long at = ((long long)b * Sk + kr) * dkv_ld + h * DH + c;
                dv[at] = av[ct][r];
                dk[at] = ak[ct][r];
            }
        }
    }
}

}  // namespace

extern "C" int rih_attention_fwd_fused(const float* q, int q_ld, const float* k, const float* v, int kv_ld, int B, int heads,
                                       int Sq, int Sk, int d, float alpha, float drop_p, uint64_t seed,
                                       const uint64_t* seed_dev, float* P, float* Pd, int ldP, float* out, int ld_out,
                                       void* stream) {
    if (!q || !k || !v || !P || !Pd || !out || B < 1 || heads < 1 || Sq < 1 || Sk < 1) return RIH_EINVAL;
    if (Sk > 32 * NTMAX || ldP < Sk || q_ld < d || kv_ld < d || ld_out < heads * d) return RIH_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && Pd == P)) return RIH_EINVAL;
    if ((long long)B * heads > 65535) return RIH_EINVAL;
    const dim3 grid((Sq + 127) / 128, B * heads), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16)
        hipLaunchKernelGGL((attn_fwd_fused_kernel<16>), grid, block, 0, s, q, q_ld, k, v, kv_ld, heads, Sq, Sk, alpha, drop_p,
                           seed, seed_dev, P, Pd, ldP, out, ld_out);
    else if (d == 32)
        hipLaunchKernelGGL((attn_fwd_fused_kernel<32>), grid, block, 0, s, q, q_ld, k, v, kv_ld, heads, Sq, Sk, alpha, drop_p,
                           seed, seed_dev, P, Pd, ldP, out, ld_out);
    else if (d == 64)
        hipLaunchKernelGGL((attn_fwd_fused_kernel<64>), grid, block, 0, s, q, q_ld, k, v, kv_ld, heads, Sq, Sk, alpha, drop_p,
                           seed, seed_dev, P, Pd, ldP, out, ld_out);
    else
        return RIH_EINVAL;
    return (int)hipGetLastError();
}

extern "C" int rih_attention_bwd_dq_fused(const float* dO, int do_ld, const float* k, const float* v, int kv_ld, int B,
                                          int heads, int Sq, int Sk, int d, float alpha, float drop_p, uint64_t seed,
                                          const uint64_t* seed_dev, const float* P, float* dS, int ldP, float* dq, int dq_ld,
                                          void* stream) {
    if (!dO || !k || !v || !P || !dS || !dq || B < 1 || heads < 1 || Sq < 1 || Sk < 1) return RIH_EINVAL;
    if (Sk > 32 * NTMAX || ldP < Sk || do_ld < heads * d || kv_ld < d || dq_ld < d) return RIH_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || (long long)B * heads > 65535) return RIH_EINVAL;
    const dim3 grid((Sq + 127) / 128, B * heads), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16)
        hipLaunchKernelGGL((attn_bwd_dq_fused_kernel<16>), grid, block, 0, s, dO, do_ld, k, v, kv_ld, heads, Sq, Sk, alpha,
                           drop_p, seed, seed_dev, P, dS, ldP, dq, dq_ld);
    else if (d == 32)
        hipLaunchKernelGGL((attn_bwd_dq_fused_kernel<32>), grid, block, 0, s, dO, do_ld, k, v, kv_ld, heads, Sq, Sk, alpha,
                           drop_p, seed, seed_dev, P, dS, ldP, dq, dq_ld);
    else if (d == 64)
        hipLaunchKernelGGL((attn_bwd_dq_fused_kernel<64>), grid, block, 0, s, dO, do_ld, k, v, kv_ld, heads, Sq, Sk, alpha,
                           drop_p, seed, seed_dev, P, dS, ldP, dq, dq_ld);
    else
        return RIH_EINVAL;
    return (int)hipGetLastError();
}

extern "C" int rih_attention_bwd_dkv_fused(const float* dO, int do_ld, const float* q, int q_ld, int B, int heads, int Sq,
                                           int Sk, int d, const float* Pd, const float* dS, int ldP, float* dk, float* dv,
                                           int dkv_ld, void* stream) {
    if (!dO || !q || !Pd || !dS || !dk || !dv || B < 1 || heads < 1 || Sq < 1 || Sk < 1) return RIH_EINVAL;
    if (ldP < Sk || do_ld < heads * d || q_ld < d || dkv_ld < d || (long long)B * heads > 65535) return RIH_EINVAL;
    const dim3 grid((Sk + 127) / 128, B * heads), block(TPB);
    hipStream_t s = (hipStream_t)stream;
    if (d == 16)
        hipLaunchKernelGGL((attn_bwd_dkv_fused_kernel<16>), grid, block, 0, s, dO, do_ld, q, q_ld, heads, Sq, Sk, Pd, dS, ldP, dk, dv, dkv_ld);
    else if (d == 32)
        hipLaunchKernelGGL((attn_bwd_dkv_fused_kernel<32>), grid, block, 0, s, dO, do_ld, q, q_ld, heads, Sq, Sk, Pd, dS, ldP, dk, dv, dkv_ld);
    else if (d == 64)
        hipLaunchKernelGGL((attn_bwd_dkv_fused_kernel<64>), grid, block, 0, s, dO, do_ld, q, q_ld, heads, Sq, Sk, Pd, dS, ldP, dk, dv, dkv_ld);
    else
        return RIH_EINVAL;
    return (int)hipGetLastError();
}
