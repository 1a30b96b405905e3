// rih_mano.hip -- MANO linear-blend-skinning layer (models/manolayer.py:250-322) for gfx950.
//
// Forward = ONE kernel (mano_fused_kernel) over a packed basis:
//   rih_mano_pack      (once per change of the model buffers; the host caches it on the buffers' version counters) lays the
//                      blend bases out as one k-major matrix Bmat[148][2496]: rows 0..134 posedirs, 135..144 shapedirs,
//                      145 v_template, 146..147 zero, columns = 778 x 3 coordinates padded to 13 x 192; and folds the joint
//                      regressor into Jt = J_reg v_template [16][3], Js = J_reg shapedirs [16][3][10].
//   mano_fused_kernel  workgroup = (tile of 64 vertices, group of hand chunks).  The tile of the basis (148 x 192 fp32 =
//                      111 KB) is pinned in LDS for the whole workgroup.  Per chunk of 16 hands: (1) PCA -> axis-angle ->
//                      Rodrigues, rest joints Jt + Js beta, 16-joint SE3 chain, tips, centre / scale / translation, all in
//                      LDS (redone by each of the 13 vertex tiles: 3 kFLOP per hand against 90 kFLOP of tile work -- cheaper
//                      than a round trip through memory and a second launch); (2) the blend GEMM
//                      v_tpose[16 hands][192] = [pose feature | beta | 1][16][148] x Bmat tile on v_mfma_f32_16x16x4_f32
//                      (exact fp32: a k-ordered fmaf chain); (3) skinning, lane = vertex, SE3s read from LDS.
//                      Inference writes only v and j; with ws != NULL (training) it also fills the backward's workspace.
//   (mano_pose_kernel / mano_vertex_kernel: the two-kernel forward of round 1, kept as `variant 1` for A/B timing.)
// Backward = one workgroup per hand (mano_bwd_kernel): reverse LBS, chain, blend shapes, Rodrigues, PCA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"

namespace {

constexpr int NV = 778, NVC = 778 * 3, NJ = 16, NPF = 135;
constexpr int TILE_V = 64;
constexpr int NTILES = (NV + TILE_V - 1) / TILE_V;   // 13

// per-hand workspace layout (floats)
constexpr int OFF_R = 0;            // [16][9]   R[0] = root, R[1..15] = local joint rotations
constexpr int OFF_JT = 144;         // [16][3]   rest joints
constexpr int OFF_G = 192;          // [16][12]  global SE3 (3x4 row-major)
constexpr int OFF_POST = 384;       // centre[3], scale, trans[3], pad
constexpr int OFF_J21C = 392;       // [21][3]   joints - centre (before scale/trans), +1 pad
constexpr int OFF_VS = 456;         // [778*3]   v_shaped
constexpr int OFF_VT = OFF_VS + NVC;     // v_tpose (shape + pose blend)
constexpr int OFF_VSC = OFF_VT + NVC;    // v_skinned - centre
constexpr int WS_STRIDE = OFF_VSC + NVC + 2;   // 7460

__constant__ int c_new_order[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};
__constant__ int c_special[13] = {745, 317, 444, 556, 673, 63, 144, 271, 220, 148, 290, 770, 83};
// new_skel: joint 5 <- (63,144), 9 <- (271,220), 13 <- (148,290), 17 <- (770,83)   (manolayer.py:316-320)
__constant__ int c_ns_joint[4] = {5, 9, 13, 17};

struct Model {
    const float* comps;
    const float* hands_mean;
    const float* shapedirs;
    const float* posedirs;
    const float* v_template;
    const float* J_reg;
    const float* weights;
    int parent[16];
    // derived on the host (to_model): tree depth of every joint ([16] = maximum) and the children of every joint in index order
    // (CSR) -- the backward used to rebuild both with chains of dependent loads from the kernel argument at its start
    int depth[17];
    int cstart[17];
    int child[16];
};

// Wave-uniform operands that an EARLIER kernel wrote: reading them through the constant address space makes the compiler use
// scalar loads (s_load_dwordx*) into SGPRs instead of per-lane requests.  (The host build of the test harness defines it empty.)
#ifndef RIH_CONST_AS
#define RIH_CONST_AS __attribute__((address_space(4)))
#endif

__device__ __forceinline__ void rodrigues_fwd(const float* ax, float* R) {
    // manolayer.py:32-48: angle = ||axis|| + 1e-8, R = I + sin K + (1-cos) K^2
    const float n = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const float th = n + 1e-8f;
    const float a0 = ax[0] / th, a1 = ax[1] / th, a2 = ax[2] / th;
    const float s = sinf(th), c1 = 1.f - cosf(th);
    const float K[9] = {0.f, -a2, a1, a2, 0.f, -a0, -a1, a0, 0.f};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float k2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) k2 += K[r * 3 + k] * K[k * 3 + c];
            R[r * 3 + c] = ((r == c) ? 1.f : 0.f) + s * K[r * 3 + c] + c1 * k2;
        }
}

__device__ __forceinline__ void rodrigues_bwd(const float* ax, const float* D, float* dax) {
    const float n = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const float th = n + 1e-8f;
    const float a[3] = {ax[0] / th, ax[1] / th, ax[2] / th};
    const float s = sinf(th), c = cosf(th), c1 = 1.f - c;
    const float K[9] = {0.f, -a[2], a[1], a[2], 0.f, -a[0], -a[1], a[0], 0.f};
    const float aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
    float dth = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const float k2 = a[r] * a[cc] - ((r == cc) ? aa : 0.f);
            dth += D[r * 3 + cc] * (c * K[r * 3 + cc] + s * k2);
        }
    const float tr = D[0] + D[4] + D[8];
    const float sk[3] = {D[7] - D[5], D[2] - D[6], D[3] - D[1]};
    float da[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float rowdot = 0.f, coldot = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) { rowdot += D[k * 3 + j] * a[j]; coldot += D[j * 3 + k] * a[j]; }
        da[k] = s * sk[k] + c1 * (rowdot + coldot - 2.f * a[k] * tr);
    }
    const float dadot = da[0] * ax[0] + da[1] * ax[1] + da[2] * ax[2];
    const float coef = (n > 0.f) ? (dth - dadot / (th * th)) / n : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) dax[k] = da[k] / th + coef * ax[k];
}

// ------------------------------------------------------------------------------------------------ forward A
__global__ __launch_bounds__(256) void mano_pose_kernel(Model m, const float* __restrict__ root,
                                                        const float* __restrict__ pose, int ncomp,
                                                        const float* __restrict__ shape,
                                                        const float* __restrict__ trans,
                                                        const float* __restrict__ scale, int center_idx, int new_skel,
                                                        float* __restrict__ jout, float* __restrict__ ws) {
    __shared__ float s_axis[48];
    __shared__ float s_R[NJ * 9];
    __shared__ float s_beta[10];
    __shared__ float s_vs[NVC];
    __shared__ float s_jt[NJ * 3];
    __shared__ float s_G[NJ * 12];
    __shared__ float s_src[21 * 3];      // 16 chain joints + 5 tips (skinned, un-centred)
    __shared__ float s_sp[13 * 3];       // skinned special vertices
    __shared__ float s_post[8];
    const int b = blockIdx.x, t = threadIdx.x;
    float* w = ws + (long long)b * WS_STRIDE;

    if (t < 10) s_beta[t] = shape[b * 10 + t];
    if (ncomp > 0 && t < 45) {
        float a = m.hands_mean[t];
        for (int c = 0; c < ncomp; ++c) a += pose[(long long)b * ncomp + c] * m.comps[c * 45 + t];
        s_axis[t] = a;
    }
    __syncthreads();
    if (t < NJ) {
        float R[9];
        if (t == 0) {
            for (int e = 0; e < 9; ++e) R[e] = root[(long long)b * 9 + e];
        } else if (ncomp > 0) {
            rodrigues_fwd(&s_axis[(t - 1) * 3], R);
        } else {
            for (int e = 0; e < 9; ++e) R[e] = pose[((long long)b * 15 + (t - 1)) * 9 + e];
        }
        for (int e = 0; e < 9; ++e) { s_R[t * 9 + e] = R[e]; w[OFF_R + t * 9 + e] = R[e]; }
    }
    for (int i = t; i < NVC; i += 256) {
        float v = m.v_template[i];
#pragma unroll
        for (int s = 0; s < 10; ++s) v += m.shapedirs[i * 10 + s] * s_beta[s];
        s_vs[i] = v;
        w[OFF_VS + i] = v;
    }
    __syncthreads();
    {   // joint regression: thread -> (joint = t/16, part = t%16)
        const int j = t >> 4, part = t & 15;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int v = part; v < NV; v += 16) {
            const float wv = m.J_reg[j * NV + v];
            a0 += wv * s_vs[v * 3 + 0];
            a1 += wv * s_vs[v * 3 + 1];
            a2 += wv * s_vs[v * 3 + 2];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            a0 += __shfl_xor(a0, o, 64);
            a1 += __shfl_xor(a1, o, 64);
            a2 += __shfl_xor(a2, o, 64);
        }
        if (part == 0) {
            s_jt[j * 3 + 0] = a0; s_jt[j * 3 + 1] = a1; s_jt[j * 3 + 2] = a2;
            w[OFF_JT + j * 3 + 0] = a0; w[OFF_JT + j * 3 + 1] = a1; w[OFF_JT + j * 3 + 2] = a2;
        }
    }
    __syncthreads();
    if (t == 0) {   // kinematic chain (manolayer.py:274-289)
        for (int i = 0; i < NJ; ++i) {
            const float* R = &s_R[i * 9];
            const float* jv = &s_jt[i * 3];
            float tl[3];
            for (int r = 0; r < 3; ++r) tl[r] = jv[r] - (R[r * 3] * jv[0] + R[r * 3 + 1] * jv[1] + R[r * 3 + 2] * jv[2]);
            float* G = &s_G[i * 12];
            if (i == 0) {
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) G[r * 4 + c] = R[r * 3 + c];
                    G[r * 4 + 3] = tl[r];
                }
                for (int r = 0; r < 3; ++r) s_src[r] = jv[r];
            } else {
                const float* P = &s_G[m.parent[i] * 12];
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c)
                        G[r * 4 + c] = P[r * 4] * R[c] + P[r * 4 + 1] * R[3 + c] + P[r * 4 + 2] * R[6 + c];
                    G[r * 4 + 3] = P[r * 4] * tl[0] + P[r * 4 + 1] * tl[1] + P[r * 4 + 2] * tl[2] + P[r * 4 + 3];
                    s_src[i * 3 + r] = P[r * 4] * jv[0] + P[r * 4 + 1] * jv[1] + P[r * 4 + 2] * jv[2] + P[r * 4 + 3];
                }
            }
        }
    }
    __syncthreads();
    if (t < NJ * 12) w[OFF_G + t] = s_G[t];
    // special vertices: pose blend (thread -> (k, c)), then skinning (thread -> k)
    __shared__ float s_spt[13 * 3];
    if (t < 39) {
        const int k = t / 3, c = t % 3;
        const int vc = c_special[k] * 3 + c;
        float v = s_vs[vc];
        for (int p = 0; p < NPF; ++p) {
            const float pf = s_R[9 + p] - (((p % 9) % 4 == 0) ? 1.f : 0.f);
            v += m.posedirs[(long long)vc * NPF + p] * pf;
        }
        s_spt[t] = v;
    }
    __syncthreads();
    if (t < 13) {
        const int v = c_special[t];
        float T[12];
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float wj = m.weights[v * NJ + j];
            for (int e = 0; e < 12; ++e) T[e] += wj * s_G[j * 12 + e];
        }
        const float* x = &s_spt[t * 3];
        for (int r = 0; r < 3; ++r) s_sp[t * 3 + r] = T[r * 4] * x[0] + T[r * 4 + 1] * x[1] + T[r * 4 + 2] * x[2] + T[r * 4 + 3];
    }
    __syncthreads();
    if (t < 15) s_src[48 + t] = s_sp[t];     // tips = special[0..4]
    __syncthreads();
    if (t == 0) {
        float ctr[3] = {0.f, 0.f, 0.f};
        if (center_idx >= 0) for (int c = 0; c < 3; ++c) ctr[c] = s_src[c_new_order[center_idx] * 3 + c];
        s_post[0] = ctr[0]; s_post[1] = ctr[1]; s_post[2] = ctr[2];
        s_post[3] = scale ? scale[b] : 1.f;
        for (int c = 0; c < 3; ++c) s_post[4 + c] = trans ? trans[b * 3 + c] : 0.f;
        s_post[7] = 0.f;
    }
    __syncthreads();
    if (t < 8) w[OFF_POST + t] = s_post[t];
    if (t < 63) {
        const int k = t / 3, c = t % 3;
        const float jc = s_src[c_new_order[k] * 3 + c] - s_post[c];
        w[OFF_J21C + t] = jc;
        float o = jc * s_post[3] + s_post[4 + c];
        if (new_skel) {
            for (int q = 0; q < 4; ++q)
                if (c_ns_joint[q] == k) {
                    const float va = (s_sp[(5 + 2 * q) * 3 + c] - s_post[c]) * s_post[3] + s_post[4 + c];
                    const float vb = (s_sp[(6 + 2 * q) * 3 + c] - s_post[c]) * s_post[3] + s_post[4 + c];
                    o = (va + vb) / 2.f;
                }
        }
        jout[(long long)b * 63 + t] = o;
    }
}

// ------------------------------------------------------------------------------------------------ forward B
__global__ __launch_bounds__(256) void mano_vertex_kernel(Model m, float* __restrict__ vout, float* __restrict__ ws,
                                                          int B, int hands_per_block) {
    __shared__ float s_pd[TILE_V * 405];      // posedirs tile pinned in LDS (103,680 B)
    const int tile = blockIdx.x;
    const int v0 = tile * TILE_V;
    const int nrows = min(TILE_V, NV - v0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    {
        const float* src = m.posedirs + (long long)v0 * 405;
        const int n = nrows * 405;
        for (int i = t; i < n; i += 256) s_pd[i] = src[i];
    }
    __syncthreads();
    const int v = v0 + lane;
    const bool valid = lane < nrows;
    float wgt[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wgt[j] = valid ? m.weights[v * NJ + j] : 0.f;
    const float* pd = &s_pd[lane * 405];
    const int h0 = blockIdx.y * hands_per_block;
    const int h1 = min(B, h0 + hands_per_block);
    for (int hh = h0 + wave; hh < h1; hh += 4) {
        const int h = __builtin_amdgcn_readfirstlane(hh);
        float* __restrict__ w = ws + (long long)h * WS_STRIDE;
        const float* __restrict__ Rl = w + OFF_R + 9;       // local rotations of joints 1..15 -> pose feature
        float vt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) vt[c] = valid ? w[OFF_VS + v * 3 + c] : 0.f;
        if (valid) {
#pragma unroll 5
            for (int p = 0; p < NPF; ++p) {
                const float pf = Rl[p] - (((p % 9) % 4 == 0) ? 1.f : 0.f);
                vt[0] += pd[p] * pf;
                vt[1] += pd[NPF + p] * pf;
                vt[2] += pd[2 * NPF + p] * pf;
            }
        }
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
        const float* __restrict__ G = w + OFF_G;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] += wgt[j] * G[j * 12 + e];
        const float cx = w[OFF_POST + 0], cy = w[OFF_POST + 1], cz = w[OFF_POST + 2];
        const float sc = w[OFF_POST + 3];
        const float tx = w[OFF_POST + 4], ty = w[OFF_POST + 5], tz = w[OFF_POST + 6];
        if (valid) {
            const float x = T[0] * vt[0] + T[1] * vt[1] + T[2] * vt[2] + T[3] - cx;
            const float y = T[4] * vt[0] + T[5] * vt[1] + T[6] * vt[2] + T[7] - cy;
            const float z = T[8] * vt[0] + T[9] * vt[1] + T[10] * vt[2] + T[11] - cz;
            float* o = vout + ((long long)h * NV + v) * 3;
            o[0] = x * sc + tx;
            o[1] = y * sc + ty;
            o[2] = z * sc + tz;
            w[OFF_VT + v * 3 + 0] = vt[0]; w[OFF_VT + v * 3 + 1] = vt[1]; w[OFF_VT + v * 3 + 2] = vt[2];
            w[OFF_VSC + v * 3 + 0] = x; w[OFF_VSC + v * 3 + 1] = y; w[OFF_VSC + v * 3 + 2] = z;
        }
    }
}

// ------------------------------------------------------------------------------------------------ packed basis
constexpr int KP = 148;                 // 135 pose features + 10 betas + 1 (template) + 2 zero rows
constexpr int NCP = NTILES * 192;       // 2496 padded coordinates
constexpr int PK_JT = KP * NCP;         // Jt[16][3]
constexpr int PK_JS = PK_JT + 48;       // Js[16][3][10]
constexpr int PK_SP = PK_JS + 480;      // [KP][48]: the basis columns of the 13 special vertices' 39 coordinates, compact (9 zero columns)
constexpr int PK_FLOATS = PK_SP + KP * 48;
constexpr int BW_STRIDE = NCP + 240;          // per-hand hand-off of the split backward: dv_tpose | dR[144] | (48 unused) | djt[48]

__global__ void mano_pack_basis_kernel(Model m, float* __restrict__ pk) {
    const long long total = (long long)KP * NCP;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / NCP), n = (int)(i - (long long)k * NCP);
        float v = 0.f;
        if (n < NVC) {
            if (k < NPF) v = m.posedirs[(long long)n * NPF + k];
            else if (k < NPF + 10) v = m.shapedirs[n * 10 + (k - NPF)];
            else if (k == NPF + 10) v = m.v_template[n];
        }
        pk[i] = v;
    }
}
// one wavefront per output: Jt[j][c] = sum_v J_reg[j][v] v_template[v][c];  Js[j][c][s] = sum_v J_reg[j][v] shapedirs[v][c][s]
// the basis columns of the special vertices gathered into a compact [KP][48] matrix (after mano_pack_basis_kernel): phase 1d of the
// fused forward used to gather them from the big matrix -- 37 requests per lane at a 10 KB stride, ~24 cache lines per instruction
__global__ void mano_pack_special_kernel(float* __restrict__ pk) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < KP * 48; idx += gridDim.x * blockDim.x) {
        const int k = idx / 48, n = idx - k * 48;
        pk[PK_SP + idx] = (n < 39) ? pk[(long long)k * NCP + c_special[n / 3] * 3 + n % 3] : 0.f;
    }
}

__global__ void mano_pack_joints_kernel(Model m, float* __restrict__ pk) {
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= 528) return;
    float a = 0.f;
    if (o < 48) {
        const int j = o / 3, c = o % 3;
        for (int v = lane; v < NV; v += 64) a += m.J_reg[j * NV + v] * m.v_template[v * 3 + c];
    } else {
        const int q = o - 48, j = q / 30, c = (q / 10) % 3, sdx = q % 10;
        for (int v = lane; v < NV; v += 64) a += m.J_reg[j * NV + v] * m.shapedirs[(v * 3 + c) * 10 + sdx];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) pk[PK_JT + o] = a;
}

// ------------------------------------------------------------------------------------------------ fused forward
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// A [rows][192] tile of a row-major matrix (row pitch src_ld floats) into LDS rows of pitch PITCH, by all 256 threads: the
// 16-byte requests go out in batches of 12 per thread BEFORE the first LDS store (a load -> store loop waits one L2 round trip
// per iteration: 28 dependent round trips per basis tile, ~26 us -- what the tile streaming of the hand-major forward and of
// the backward blend kernel used to cost).  PITCH % 4 == 0: 16-byte stores; otherwise (even pitch) two 8-byte stores.
// SWZ (the forward's basis tile, pitch 192 = 3 x 64 banks): the 16-float column block of row k is stored at block ^ (k & 3), so
// that the blend GEMM's operand read -- 4 consecutive rows x 16 consecutive columns per wave instruction -- hits 64 different
// banks instead of 16 four times (round 4; a padded pitch does not fit: the kernel uses 162.7 of 163.8 KB).
template <int PITCH, bool SWZ = false>
__device__ __forceinline__ void load_tile_192(float* __restrict__ dst, const float* __restrict__ src, long long src_ld, int rows,
                                              int t) {
    constexpr int NB = 12;
    const int total = rows * 48;
    for (int base = 0; base < total; base += 256 * NB) {
        floatx4 r[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = base + t + 256 * u;
            r[u] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (i < total) {
                const int k = i / 48, q = i - k * 48;
                r[u] = *reinterpret_cast<const floatx4*>(src + (long long)k * src_ld + 4 * q);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = base + t + 256 * u;
            if (i < total) {
                const int k = i / 48, q = i - k * 48;
                float* d = dst + k * PITCH + (SWZ ? ((4 * q) ^ ((k & 3) << 4)) : 4 * q);
                if (PITCH % 4 == 0) {
                    *reinterpret_cast<floatx4*>(d) = r[u];
                } else {
                    *reinterpret_cast<floatx2*>(d) = floatx2{r[u].x, r[u].y};
                    *reinterpret_cast<floatx2*>(d + 2) = floatx2{r[u].z, r[u].w};
                }
            }
        }
    }
}

// Skinning FMAs with the SE3 element as a ROW-BROADCAST operand (DPP row_newbcast: every row of 16 lanes reads lane K of its own
// row): T[c] += G[(j, c)] * w[j] for four joints j and the twelve elements c, where the 48 values G[(j, c)] of the four joints sit in
// three registers G0..G2 whose lane l holds flat element 16 r + (l & 15) -- the same sixteen values in each of the four rows.  One
// v_fmac_f32_dpp per product: no LDS read, no move.  (The compiler does not fold __builtin_amdgcn_update_dpp into the FMA -- it
// emits v_mov_b32_dpp + v_fma -- so the block is inline assembly; the leading `s_nop 4` covers what the hazard recognizer cannot see
// inside an asm statement: the two wait states a DPP read needs after a VALU write of its source AND the five a DPP instruction needs
// after a VALU write of EXEC (v_cmpx; ADVICE round 5 -- the compiler emits no such sequence in front of the block today, but that is
// code generation, not a guarantee).)  The host build of the test
// harness supplies the same arithmetic with an emulated lane exchange.
#ifndef RIH_SKIN_GROUP
#define RIH_SKIN_GROUP(T, G0, G1, G2, W0, W1, W2, W3)                                                                             \
    asm("s_nop 4\n"                                                                                                               \
        "v_fmac_f32_dpp %0, %12, %15 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %1, %12, %15 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %2, %12, %15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %3, %12, %15 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %4, %12, %15 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %5, %12, %15 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %6, %12, %15 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %7, %12, %15 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %8, %12, %15 row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %9, %12, %15 row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %10, %12, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %11, %12, %15 row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %0, %12, %16 row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %1, %12, %16 row_newbcast:13 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %2, %12, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %3, %12, %16 row_newbcast:15 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %4, %13, %16 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %5, %13, %16 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %6, %13, %16 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %7, %13, %16 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %8, %13, %16 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %9, %13, %16 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %10, %13, %16 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %11, %13, %16 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %0, %13, %17 row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %1, %13, %17 row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %2, %13, %17 row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %3, %13, %17 row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %4, %13, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %5, %13, %17 row_newbcast:13 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %6, %13, %17 row_newbcast:14 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %7, %13, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %8, %14, %17 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %9, %14, %17 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %10, %14, %17 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %11, %14, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %0, %14, %18 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %1, %14, %18 row_newbcast:5 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %2, %14, %18 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %3, %14, %18 row_newbcast:7 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %4, %14, %18 row_newbcast:8 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %5, %14, %18 row_newbcast:9 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %6, %14, %18 row_newbcast:10 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %7, %14, %18 row_newbcast:11 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %8, %14, %18 row_newbcast:12 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %9, %14, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %10, %14, %18 row_newbcast:14 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %11, %14, %18 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"                                                                                                                         \
        : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]), "+v"(T[4]), "+v"(T[5]), "+v"(T[6]), "+v"(T[7]), "+v"(T[8]), "+v"(T[9]),   \
          "+v"(T[10]), "+v"(T[11])                                                                                                 \
        : "v"(G0), "v"(G1), "v"(G2), "v"(W0), "v"(W1), "v"(W2), "v"(W3))
#endif

constexpr int HC = 16;                  // hands per chunk = one 16-row MFMA block
constexpr int LDPF = 149;               // odd row stride of the pose-feature operand: conflict-free column reads
constexpr int GST = 200;                // per-hand SE3s (16 x 12) + post (8)
constexpr int SCR = 392;                // per-hand phase-1 scratch: axis 48 | R 144 | jt 48 | src 63 | sp 39 | spt 39 | pad
constexpr int S_AX = 0, S_RR = 48, S_JT = 192, S_SRC = 240, S_SP = 303, S_SPT = 342;
constexpr int FUSED_LDS_FLOATS = KP * 192 + HC * LDPF + HC * GST + HC * SCR + 528 + 20;    // 40,676 floats = 162,704 B
static_assert(FUSED_LDS_FLOATS * 4 <= 163840, "LDS budget");
static_assert(HC * 192 <= HC * SCR, "the v_tpose tile aliases the phase-1 scratch");

template <bool HM>
__global__ __launch_bounds__(256) void mano_fused_kernel(Model m, const float* __restrict__ pk,
                                                         const float* __restrict__ root, const float* __restrict__ pose,
                                                         int ncomp, const float* __restrict__ shape,
                                                         const float* __restrict__ trans, const float* __restrict__ scale,
                                                         int center_idx, int new_skel, float* __restrict__ vout,
                                                         float* __restrict__ jout, float* __restrict__ ws, int B,
                                                         long long* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) float smem[FUSED_LDS_FLOATS];
    // phase timestamps (shader clock) of the first chunk of the workgroups (tile, group 0): development aid, dbg == NULL otherwise
#define RIH_STAMP(i_) do { if (dbg != nullptr && threadIdx.x == 0 && blockIdx.y == 0 && chunk == 0) dbg[blockIdx.x * 16 + (i_)] = clock64(); } while (0)
#define RIH_STAMP2(i_) do { if (dbg != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && chunk == 0) dbg[16 + (i_)] = clock64(); } while (0)
    float* s_B = smem;                          // [KP][192] basis tile
    float* s_pf = s_B + KP * 192;               // [HC][LDPF] pose feature | beta | 1 | 0 0
    float* s_G = s_pf + HC * LDPF;              // [HC][GST]
    float* s_scr = s_G + HC * GST;              // [HC][SCR] phase-1 scratch, later [HC][192] v_tpose tile
    float* s_J = s_scr + HC * SCR;              // Jt[48] | Js[480]: the joint regressor folded onto template / shape basis
    int* s_depth = reinterpret_cast<int*>(s_J + 528);      // tree depth of the 16 joints, [16] = maximum
    // Two workgroup shapes.  !HM (small batches): workgroup = (vertex tile blockIdx.x, group of hand chunks blockIdx.y), the
    // basis tile pinned once; the per-hand pose work of a chunk is repeated by each of the 13 tiles (47 % of a workgroup's
    // cycles, profiles/r02/mano_phases_m8.log).  HM (hand-chunk major, >= 256 chunks): workgroup = group of hand chunks, pose
    // work once per chunk, then the 13 basis tiles are streamed through the same LDS buffer from L2 (1.46 MB, L2-resident).
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile_fixed = HM ? 0 : (int)blockIdx.x;
    // does this workgroup need the 13 special vertices (tips / new_skel)?  Tile 0 writes the joints; every tile needs the
    // centre joint, which is a tip when new_order[center] >= 16.
    const bool centre_is_tip = center_idx >= 0 && c_new_order[center_idx] >= 16;
    const bool first_tile = HM || tile_fixed == 0;        // this workgroup writes the joints / joint-side workspace
    const bool need_special = first_tile || centre_is_tip;

    // the basis tile: rows are 192 contiguous floats of Bmat
    auto load_basis = [&](int tile) { load_tile_192<192, true>(s_B, pk + tile * 192, NCP, KP, t); };
    if (!HM) load_basis(tile_fixed);
    for (int i = t; i < 528; i += 256) s_J[i] = pk[PK_JT + i];
    if (t <= NJ) s_depth[t] = m.depth[t];
    __syncthreads();
    float wgt[NJ];
    auto load_weights = [&](int tile) {
        const int vv = tile * TILE_V + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // the vertex's 16 weights: four 16-byte requests (rounds 1-4: sixteen dword loads)
            const floatx4 w4 = (vv < NV) ? *reinterpret_cast<const floatx4*>(m.weights + vv * NJ + 4 * q) : floatx4{0.f, 0.f, 0.f, 0.f};
            wgt[4 * q] = w4.x; wgt[4 * q + 1] = w4.y; wgt[4 * q + 2] = w4.z; wgt[4 * q + 3] = w4.w;
        }
    };
    if (!HM) load_weights(tile_fixed);

    const int nchunks = (B + HC - 1) / HC;
    for (int chunk = HM ? blockIdx.x : blockIdx.y; chunk < nchunks; chunk += HM ? gridDim.x : gridDim.y) {
        const int h0 = chunk * HC;
        // the basis columns of the special vertices (phase 1d) are requested NOW: they depend on nothing, and behind phases 1a-1c
        // their (cold) round trip is hidden -- issued in 1d itself they were 14 k of the 20 k cycles of that phase
        float bv_sp[KP / 4];
        if (need_special && wave < 3) {
            const int kq = lane >> 4, n = wave * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) bv_sp[ks] = pk[PK_SP + (4 * ks + kq) * 48 + n];       // (compact copy: rih_mano_pack)
        }
        __syncthreads();                        // previous chunk's skinning is done with s_G / s_scr
        RIH_STAMP(0);
        // ---- phase 1a: axis-angle = hands_mean + pose x comps as a 16 x 48 x 48 product on the f32 MFMA (waves 0..2 own 16
        //      outputs each; all operand loads are issued before the first MFMA), beta and the constant operand columns
        if (ncomp > 0 && wave < 3) {
            const int hl = lane & 15, kq = lane >> 4, e = wave * 16 + (lane & 15);
            float av[12], bv[12];
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) {
                const int k = 4 * ks + kq;
                av[ks] = (k < ncomp && h0 + hl < B) ? pose[(long long)(h0 + hl) * ncomp + k] : 0.f;
                bv[ks] = (k < ncomp && e < 45) ? m.comps[k * 45 + e] : 0.f;
            }
            floatx4 ax = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 12; ++ks) ax = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], ax, 0, 0, 0);
            const float mean = (e < 45) ? m.hands_mean[e] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s_scr[(4 * kq + r) * SCR + S_AX + e] = ax[r] + mean;
        }
        for (int i = t; i < HC * 13; i += 256) {
            const int hl = i / 13, e = i - hl * 13, h = h0 + hl;
            float x = 0.f;
            if (h < B) x = (e < 10) ? shape[h * 10 + e] : (e == 10 ? 1.f : 0.f);
            s_pf[hl * LDPF + NPF + e] = x;
        }
        __syncthreads();
        RIH_STAMP(1);
        // ---- phase 1b: rotations (thread = (hand, joint)) -> pose feature and local transforms L_j = [R_j | jt_j - R_j jt_j]
        //      (parked in the SE3 slots), rest joints jt = Jt + Js beta
        for (int i = t; i < HC * 48; i += 256) {
            const int hl = i / 48, e = i - hl * 48;
            float a = s_J[e];
#pragma unroll
            for (int sdx = 0; sdx < 10; ++sdx) a += s_J[48 + e * 10 + sdx] * s_pf[hl * LDPF + NPF + sdx];
            s_scr[hl * SCR + S_JT + e] = a;
            if (ws != nullptr && first_tile && h0 + hl < B) ws[(long long)(h0 + hl) * WS_STRIDE + OFF_JT + e] = a;
        }
        __syncthreads();
        if (t < HC * NJ) {
            const int hl = t >> 4, j = t & 15, h = h0 + hl;
            float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
            if (h < B) {
                if (j == 0) {
                    for (int e = 0; e < 9; ++e) R[e] = root[(long long)h * 9 + e];
                } else if (ncomp > 0) {
                    rodrigues_fwd(&s_scr[hl * SCR + S_AX + (j - 1) * 3], R);
                } else {
                    for (int e = 0; e < 9; ++e) R[e] = pose[((long long)h * 15 + (j - 1)) * 9 + e];
                }
            }
            const float* jv = &s_scr[hl * SCR + S_JT + j * 3];
            float* L = &s_G[hl * GST + j * 12];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) L[r * 4 + c] = R[r * 3 + c];
                L[r * 4 + 3] = jv[r] - (R[r * 3] * jv[0] + R[r * 3 + 1] * jv[1] + R[r * 3 + 2] * jv[2]);
            }
            if (j > 0)
                for (int e = 0; e < 9; ++e) s_pf[hl * LDPF + (j - 1) * 9 + e] = (h < B) ? R[e] - ((e % 4 == 0) ? 1.f : 0.f) : 0.f;
            if (ws != nullptr && first_tile && h < B)
                for (int e = 0; e < 9; ++e) ws[(long long)h * WS_STRIDE + OFF_R + j * 9 + e] = R[e];
        }
        __syncthreads();
        RIH_STAMP(2);
        // ---- phase 1c: kinematic chain by tree level (manolayer.py:274-289): G_j = G_parent(j) L_j, thread = (hand, joint,
        //      row); a level reads its parents' finished SE3s, computes into registers, then overwrites its own slots
        for (int lvl = 1; lvl <= s_depth[16]; ++lvl) {
            float o[3][4];
#pragma unroll
            for (int q = 0; q < 3; ++q) {               // HC * 48 = 768 (hand, joint, row) items over 256 threads
                const int i = t + 256 * q;
                const int hl = i / 48, jr = i - hl * 48, j = jr / 3, r = jr - j * 3;
                if (s_depth[j] == lvl) {
                    const float* P = &s_G[hl * GST + m.parent[j] * 12 + r * 4];
                    const float* L = &s_G[hl * GST + j * 12];
                    o[q][0] = P[0] * L[0] + P[1] * L[4] + P[2] * L[8];
                    o[q][1] = P[0] * L[1] + P[1] * L[5] + P[2] * L[9];
                    o[q][2] = P[0] * L[2] + P[1] * L[6] + P[2] * L[10];
                    o[q][3] = P[0] * L[3] + P[1] * L[7] + P[2] * L[11] + P[3];
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int i = t + 256 * q;
                const int hl = i / 48, jr = i - hl * 48, j = jr / 3, r = jr - j * 3;
                if (s_depth[j] == lvl) {
                    float* Gj = &s_G[hl * GST + j * 12 + r * 4];
                    Gj[0] = o[q][0]; Gj[1] = o[q][1]; Gj[2] = o[q][2]; Gj[3] = o[q][3];
                }
            }
            __syncthreads();
        }
        // chain joint positions: src_j = G_j applied to the rest joint (= G_parent jt_j + t_parent)
        for (int i = t; i < HC * 48; i += 256) {
            const int hl = i / 48, jr = i - hl * 48, j = jr / 3, r = jr - j * 3;
            const float* Gj = &s_G[hl * GST + j * 12 + r * 4];
            const float* jv = &s_scr[hl * SCR + S_JT + j * 3];
            s_scr[hl * SCR + S_SRC + jr] = Gj[0] * jv[0] + Gj[1] * jv[1] + Gj[2] * jv[2] + Gj[3];
        }
        __syncthreads();
        RIH_STAMP(3);
        // ---- phase 1d: the 13 special vertices (5 tips + 8 new_skel): blend on the MFMA with the basis columns read straight
        //      from their compact copy in the packed buffer (waves 0..2 own 16 of the 39 coordinates each), then skinning
        if (need_special) {
            if (wave < 3) {
                const int kq = lane >> 4, n = wave * 16 + (lane & 15);
                const float* a_rd = s_pf + (lane & 15) * LDPF + kq;
                float av[KP / 4];
#pragma unroll
                for (int ks = 0; ks < KP / 4; ++ks) av[ks] = a_rd[4 * ks];      // (all LDS reads first: see blend_steps)
                floatx4 sp = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KP / 4; ++ks) sp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv_sp[ks], sp, 0, 0, 0);
                if (n < 39)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_scr[(4 * kq + r) * SCR + S_SPT + n] = sp[r];
            }
            RIH_STAMP2(0);
            __syncthreads();
            RIH_STAMP2(1);
            for (int i = t; i < HC * 13; i += 256) {
                const int hl = i / 13, q = i - hl * 13, vv = c_special[q];
                // the vertex's sixteen weights in four 16-byte requests up front (the joint loop used to wait for one dword load
                // per joint: sixteen L2 round trips in a row, most of the 20 k cycles this phase took), the SE3s as 16-byte LDS reads
                float wv[NJ];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const floatx4 w4 = *reinterpret_cast<const floatx4*>(m.weights + vv * NJ + 4 * u);
                    wv[4 * u] = w4.x; wv[4 * u + 1] = w4.y; wv[4 * u + 2] = w4.z; wv[4 * u + 3] = w4.w;
                }
                float T[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float* Gj = &s_G[hl * GST + j * 12];
                    const float4 g0 = *reinterpret_cast<const float4*>(Gj), g1 = *reinterpret_cast<const float4*>(Gj + 4),
                                 g2 = *reinterpret_cast<const float4*>(Gj + 8);
                    T[0] += wv[j] * g0.x; T[1] += wv[j] * g0.y; T[2] += wv[j] * g0.z; T[3] += wv[j] * g0.w;
                    T[4] += wv[j] * g1.x; T[5] += wv[j] * g1.y; T[6] += wv[j] * g1.z; T[7] += wv[j] * g1.w;
                    T[8] += wv[j] * g2.x; T[9] += wv[j] * g2.y; T[10] += wv[j] * g2.z; T[11] += wv[j] * g2.w;
                }
                const float* x = &s_scr[hl * SCR + S_SPT + q * 3];
                for (int r = 0; r < 3; ++r)
                    s_scr[hl * SCR + S_SP + q * 3 + r] = T[r * 4] * x[0] + T[r * 4 + 1] * x[1] + T[r * 4 + 2] * x[2] + T[r * 4 + 3];
            }
            RIH_STAMP2(2);
            __syncthreads();
            RIH_STAMP2(3);
            for (int i = t; i < HC * 15; i += 256) {
                const int hl = i / 15, e = i - hl * 15;
                s_scr[hl * SCR + S_SRC + 48 + e] = s_scr[hl * SCR + S_SP + e];       // tips = special[0..4]
            }
            __syncthreads();
        }
        RIH_STAMP(4);
        // ---- phase 1e: centre / scale / translation; tile 0 writes the 21 joints (and the joint-side workspace)
        if (t < HC) {
            const int h = h0 + t;
            float* post = &s_G[t * GST + 192];
            for (int c = 0; c < 3; ++c) post[c] = (center_idx >= 0) ? s_scr[t * SCR + S_SRC + c_new_order[center_idx] * 3 + c] : 0.f;
            post[3] = (scale != nullptr && h < B) ? scale[h] : 1.f;
            for (int c = 0; c < 3; ++c) post[4 + c] = (trans != nullptr && h < B) ? trans[h * 3 + c] : 0.f;
            post[7] = 0.f;
        }
        __syncthreads();
        if (first_tile) {
            for (int i = t; i < HC * 63; i += 256) {
                const int hl = i / 63, e = i - hl * 63, h = h0 + hl;
                if (h >= B) continue;
                const int k = e / 3, c = e % 3;
                const float* post = &s_G[hl * GST + 192];
                const float jc = s_scr[hl * SCR + S_SRC + c_new_order[k] * 3 + c] - post[c];
                float o = jc * post[3] + post[4 + c];
                if (new_skel) {
                    for (int q = 0; q < 4; ++q)
                        if (c_ns_joint[q] == k) {
                            const float va = (s_scr[hl * SCR + S_SP + (5 + 2 * q) * 3 + c] - post[c]) * post[3] + post[4 + c];
                            const float vb = (s_scr[hl * SCR + S_SP + (6 + 2 * q) * 3 + c] - post[c]) * post[3] + post[4 + c];
                            o = (va + vb) / 2.f;
                        }
                }
                jout[(long long)h * 63 + e] = o;
                if (ws != nullptr) ws[(long long)h * WS_STRIDE + OFF_J21C + e] = jc;
            }
            if (ws != nullptr)
                for (int i = t; i < HC * GST; i += 256) {
                    const int hl = i / GST, e = i - hl * GST, h = h0 + hl;
                    if (h < B) ws[(long long)h * WS_STRIDE + OFF_G + e] = s_G[i];     // OFF_POST = OFF_G + 192
                }
            __syncthreads();            // the scratch (src / sp) is overwritten by the v_tpose tile below
        }
        RIH_STAMP(5);
        // ---- phase 2: v_tpose[16][192] = operand[16][148] x Bmat tile; wave w owns coordinate blocks 3w .. 3w+2 (k-steps of 4)
        floatx4 acc[3];
        // Round 5: the operands of several k-steps at a time in registers, the next group requested before the products of the
        // current one.  The compiler's own schedule read ONE k-step ahead -- ds_read, s_waitcnt, v_mfma, 111 times per tile: every
        // product waited out an LDS round trip (the workgroup is alone on its CU, one wavefront per SIMD, nothing else to run).
        auto blend_steps = [&](int ks0, int ks1) {
            constexpr int GS = 7;
            const int kq = lane >> 4;                       // row 4 ks + kq: its column blocks sit at block ^ kq (load_tile_192 SWZ)
            const float* a_rd = s_pf + (lane & 15) * LDPF + kq;
            const float* b_rd = s_B + kq * 192 + (lane & 15);
            int cb[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) cb[j] = (wave * 48 + 16 * j) ^ (kq << 4);
            float av[2][GS], bv[2][GS][3];
            auto fetch = [&](int buf, int g0) {
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    const int ks = g0 + i;
                    if (ks < ks1) {
                        av[buf][i] = a_rd[4 * ks];
#pragma unroll
                        for (int j = 0; j < 3; ++j) bv[buf][i][j] = b_rd[4 * ks * 192 + cb[j]];
                    }
                }
            };
            fetch(0, ks0);
            int buf = 0;
#pragma unroll
            for (int g0 = ks0; g0 < ks1; g0 += GS, buf ^= 1) {
                if (g0 + GS < ks1) fetch(buf ^ 1, g0 + GS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GS; ++i) {
                    if (g0 + i < ks1) {
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][i], bv[buf][i][j], acc[j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // C/D layout: column = lane & 15 (coordinate), row = 4 * (lane >> 4) + r (hand)
        auto park_vtpose = [&]() {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_scr[(4 * (lane >> 4) + r) * 192 + wave * 48 + 16 * j + (lane & 15)] = acc[j][r];
        };
        // ---- phase 3: skinning, lane = vertex, wave w takes hands w, w+4, ...
        auto skin = [&](int tile) {
            const int v = tile * TILE_V + lane;
            const bool valid = v < NV;
            // Round 5: the hand's 192 SE3 elements in twelve registers (lane l holds element 16 r + (l & 15)); the products take
            // them as DPP row-broadcast operands (RIH_SKIN_GROUP).  Rounds 1-4 fetched every element for every lane from LDS --
            // 48 broadcast ds_read_b128 per hand, 192 per wavefront and tile at 8 cycles of the CU's one LDS pipe each: 6.1 k of
            // the 7.8 k cycles the skinning of a tile took (tools/mano_phases.py).
            // (measured and not kept on the LDS form in round 4: v_pk_fma_f32 121 -> 128.6 us at 4096 hands; T as a
            // [64 vertices x 16 joints] x [16 x 12] product on the f32 MFMA with a quad transpose-reduce 147-150 us
            // (profiles/r04/c19, c20); round 5: SE3 reads pipelined through registers / lane = (hand, vertex) with the SE3s in
            // registers: both beyond 512 registers, 128-130 us (profiles/r05/mano/c15_forward_experiments_stdout.txt))
            // The operands of the NEXT hand (twelve SE3 registers, three v_tpose coordinates, seven post-transform scalars) are
            // requested before the products of the current one: one wavefront per SIMD, nothing else hides an LDS round trip.
            float g[2][12], vt[2][3], post[2][7];
            auto fetch = [&](int buf, int hl) {
                const float* G = &s_G[hl * GST];
#pragma unroll
                for (int r = 0; r < 12; ++r) g[buf][r] = G[16 * r + (lane & 15)];
#pragma unroll
                for (int k = 0; k < 3; ++k) vt[buf][k] = s_scr[hl * 192 + lane * 3 + k];
#pragma unroll
                for (int e = 0; e < 7; ++e) post[buf][e] = G[192 + e];
            };
            auto hand = [&](int buf, int hl) {
                const int h = h0 + hl;
                float T[12];
#pragma unroll
                for (int e = 0; e < 12; ++e) T[e] = 0.f;
                RIH_SKIN_GROUP(T, g[buf][0], g[buf][1], g[buf][2], wgt[0], wgt[1], wgt[2], wgt[3]);
                RIH_SKIN_GROUP(T, g[buf][3], g[buf][4], g[buf][5], wgt[4], wgt[5], wgt[6], wgt[7]);
                RIH_SKIN_GROUP(T, g[buf][6], g[buf][7], g[buf][8], wgt[8], wgt[9], wgt[10], wgt[11]);
                RIH_SKIN_GROUP(T, g[buf][9], g[buf][10], g[buf][11], wgt[12], wgt[13], wgt[14], wgt[15]);
                const float vt0 = vt[buf][0], vt1 = vt[buf][1], vt2 = vt[buf][2];
                const float* ps = post[buf];
                // (the three dword stores per output at a 12-byte stride stay: sending the 192 consecutive floats of a tile back
                // through LDS to leave as 8-byte stores of consecutive lanes was measured in round 4 -- 121 -> 184 us at 4096
                // hands: three dependent LDS round trips per hand cost far more than the scattered stores)
                if (valid && h < B) {
                    const float x = T[0] * vt0 + T[1] * vt1 + T[2] * vt2 + T[3] - ps[0];
                    const float y = T[4] * vt0 + T[5] * vt1 + T[6] * vt2 + T[7] - ps[1];
                    const float z = T[8] * vt0 + T[9] * vt1 + T[10] * vt2 + T[11] - ps[2];
                    float* o = vout + ((long long)h * NV + v) * 3;
                    o[0] = x * ps[3] + ps[4];
                    o[1] = y * ps[3] + ps[5];
                    o[2] = z * ps[3] + ps[6];
                    if (ws != nullptr) {
                        float* w = ws + (long long)h * WS_STRIDE;
                        w[OFF_VT + v * 3 + 0] = vt0; w[OFF_VT + v * 3 + 1] = vt1; w[OFF_VT + v * 3 + 2] = vt2;
                        w[OFF_VSC + v * 3 + 0] = x; w[OFF_VSC + v * 3 + 1] = y; w[OFF_VSC + v * 3 + 2] = z;
                    }
                }
            };
            fetch(0, wave);
#pragma unroll 1
            for (int hp = 0; hp < HC / 8; ++hp) {           // two hands per iteration: the register sets alternate statically
                const int hl = wave + 8 * hp;
                fetch(1, hl + 4);
                hand(0, hl);
                if (hp + 1 < HC / 8) fetch(0, hl + 8);
                hand(1, hl + 4);
            }
        };
        if (!HM) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
            blend_steps(0, KP / 4);
            park_vtpose();
            __syncthreads();
            RIH_STAMP(6);
            skin(tile_fixed);
        } else {
            // Hand-major: every workgroup streams the same 13 tiles (each starts at its own, so that the CUs of an XCD do not all
            // ask the same L2 channels for the same lines at the same time).  Round 4: the tile arrives in two halves of its
            // k-range (rows 0..75 = 19 k-steps, rows 76..147 = 18), and the NEXT half is always in flight -- global -> registers
            // -- while the MFMAs (and, for the first half of the next tile, the skinning) of the current one run; it lands in the
            // LDS rows its predecessor has finished with.  One basis tile of LDS as before (two would be 222 KB); the chain of
            // round trips load -> blend -> skin per tile (13 x ~9 us of a 127 us forward at 4096 hands) loses its load link.
            constexpr int H0 = 76, KS0 = H0 / 4, NHR = 15;          // 76 x 48 = 3648 16-byte items = 14.25 per thread
            floatx4 hr[NHR];
            // a thread's items of a half tile: item i = t + 256 u -> row i / 48, 16-byte column i % 48 -- the same for every
            // tile and both halves, so the source offsets are computed once per chunk (they were recomputed, with their 64-bit
            // address arithmetic, 30 times per tile)
            int src_off[NHR];
#pragma unroll
            for (int u = 0; u < NHR; ++u) {
                const int i = t + 256 * u, k = i / 48, q = i - k * 48;
                src_off[u] = k * NCP + 4 * q;
            }
            auto issue_half = [&](int tile, int half) {
                const int total = (half ? KP - H0 : H0) * 48;
                const float* src = pk + (half ? H0 : 0) * NCP + tile * 192;
#pragma unroll
                for (int u = 0; u < NHR; ++u) {
                    hr[u] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (t + 256 * u < total) hr[u] = *reinterpret_cast<const floatx4*>(src + src_off[u]);
                }
            };
            auto land_half = [&](int half) {
                const int r0 = half ? H0 : 0, total = (half ? KP - H0 : H0) * 48;
#pragma unroll
                for (int u = 0; u < NHR; ++u) {
                    const int i = t + 256 * u;
                    if (i < total) {
                        const int k = i / 48, q = i - k * 48;
                        *reinterpret_cast<floatx4*>(s_B + (r0 + k) * 192 + ((4 * q) ^ (((r0 + k) & 3) << 4))) = hr[u];
                    }
                }
            };
            const int tile0 = (int)blockIdx.x % NTILES;
            issue_half(tile0, 0);
            __syncthreads();            // phase 1 is done with its scratch; the previous chunk's last blend with the basis rows
            land_half(0);
            __syncthreads();
            for (int tl = 0; tl < NTILES; ++tl) {
                const int tile = (tl + (int)blockIdx.x) % NTILES;
                // (stamps 8..14: the steps of the third tile, tools/mano_phases.py)
                if (tl == 2) RIH_STAMP(8);
                issue_half(tile, 1);
                load_weights(tile);
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
                blend_steps(0, KS0);
                if (tl == 2) RIH_STAMP(9);
                land_half(1);           // rows 76..147: the previous tile's second half was finished before its skinning
                __syncthreads();
                if (tl == 2) RIH_STAMP(10);
                if (tl + 1 < NTILES) issue_half((tile + 1) % NTILES, 0);
                blend_steps(KS0, KP / 4);
                if (tl == 2) RIH_STAMP(11);
                park_vtpose();
                __syncthreads();        // v_tpose complete; every wave is done with rows 0..75
                if (tl == 0) RIH_STAMP(6);
                if (tl == 2) RIH_STAMP(12);
                skin(tile);
                if (tl == 2) RIH_STAMP(13);
                if (tl + 1 < NTILES) land_half(0);
                __syncthreads();        // next tile's first half landed; the skinning is done with the v_tpose tile
                if (tl == 2) RIH_STAMP(14);
            }
        }
        RIH_STAMP(7);
    }
#undef RIH_STAMP
}

// ------------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void mano_bwd_kernel(Model m, const float* __restrict__ pose, int ncomp,
                                                       int center_idx, int new_skel, int has_scale,
                                                       const float* __restrict__ dv, const float* __restrict__ dj,
                                                       const float* __restrict__ ws, float* __restrict__ d_root,
                                                       float* __restrict__ d_pose, float* __restrict__ d_shape,
                                                       float* __restrict__ d_trans, float* __restrict__ d_scale,
                                                       float* __restrict__ wsb, long long* __restrict__ dbg) {
    // phase timestamps of workgroup 0 (development aid, tools/mano_phases.py): slots 208.. of the rih_mano_debug_stamps buffer
#define RIH_BSTAMP(i_) do { if (dbg != nullptr && threadIdx.x == 0 && blockIdx.x == 0) dbg[208 + (i_)] = clock64(); } while (0)
    RIH_BSTAMP(0);
    __shared__ float s_dvs[NVC];           // dv_eff -> dv_skin -> dv_tpose -> dv_shaped (in place)
    __shared__ float s_M[NV * 6];          // per vertex: dv_skin (3) | v_tpose (3) -- the factors of M_v = dv_skin x [v_t;1]
    __shared__ __attribute__((aligned(16))) float s_G[NJ * 12];
    __shared__ float s_R[NJ * 9], s_jt[NJ * 3];
    __shared__ float s_dG[NJ * 12], s_dR[NJ * 9], s_djt[NJ * 3];
    __shared__ float s_djeff[63], s_dsrc[63];
    __shared__ float s_dpf[NPF];
    __shared__ float s_axis[48], s_dax[48];
    __shared__ float s_red[4];
    __shared__ float s_sh[250];
    __shared__ float s_part[4 * NJ * 12];
    __shared__ float s_red4[16];
    __shared__ int s_depth[NJ + 1];        // tree depth of the joints, [16] = maximum
    __shared__ int s_par[NJ], s_cstart[NJ + 1], s_child[NJ];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t <= NJ) { s_depth[t] = m.depth[t]; s_cstart[t] = m.cstart[t]; }
    if (t < NJ) { s_par[t] = m.parent[t]; s_child[t] = m.child[t]; }
    const float* w = (const float*)__builtin_assume_aligned(ws + (long long)b * WS_STRIDE, 16);     // (WS_STRIDE % 4 == 0)
    const float* dvb = dv + (long long)b * NVC;
    const float* djb = dj + (long long)b * 63;

    if (t < NJ * 12) { s_G[t] = w[OFF_G + t]; s_dG[t] = 0.f; }
    if (t < NJ * 9) { s_R[t] = w[OFF_R + t]; s_dR[t] = 0.f; }
    if (t < NJ * 3) { s_jt[t] = w[OFF_JT + t]; s_djt[t] = 0.f; }
    if (t < 63) {
        float d = djb[t];
        if (new_skel) {
            const int k = t / 3;
            if (k == 5 || k == 9 || k == 13 || k == 17) d = 0.f;
        }
        s_djeff[t] = d;
    }
    if (wsb == nullptr && ncomp > 0 && t < 45) {        // (split backward: the finish kernel computes the axis-angle vectors)
        float a = m.hands_mean[t];
        for (int c = 0; c < ncomp; ++c) a += pose[(long long)b * ncomp + c] * m.comps[c * 45 + t];
        s_axis[t] = a;
    }
    const float sc = w[OFF_POST + 3];
    RIH_BSTAMP(1);
    // 1. effective vertex gradient (new_skel joints are averages of output vertices).  Round 5: the 2 x 10 global requests of a
    //    thread go out together (the loop used to wait one round trip per iteration: 17 k of the workgroup's 120 k cycles)
    float p_sv0 = 0.f, p_sv1 = 0.f, p_sv2 = 0.f, p_dot = 0.f;
    {
        constexpr int NIT = (NVC + 255) / 256;
        float dreg[NIT], wreg[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = t + 256 * u;
            dreg[u] = (i < NVC) ? dvb[i] : 0.f;
            wreg[u] = (i < NVC) ? w[OFF_VSC + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = t + 256 * u;
            if (i >= NVC) continue;
            float d = dreg[u];
            const int v = i / 3, c = i - v * 3;
            if (new_skel) {
                for (int q = 0; q < 8; ++q)
                    if (c_special[5 + q] == v) d += 0.5f * djb[c_ns_joint[q >> 1] * 3 + c];
            }
            s_dvs[i] = d;
            if (c == 0) p_sv0 += d; else if (c == 1) p_sv1 += d; else p_sv2 += d;
            p_dot += d * wreg[u];
        }
    }
    __syncthreads();
    if (t < 63) p_dot += s_djeff[t] * w[OFF_J21C + t];
    float sv0, sv1, sv2, dot;               // four block sums behind ONE pair of barriers
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            p_sv0 += __shfl_xor(p_sv0, o, 64); p_sv1 += __shfl_xor(p_sv1, o, 64);
            p_sv2 += __shfl_xor(p_sv2, o, 64); p_dot += __shfl_xor(p_dot, o, 64);
        }
        __syncthreads();
        if ((t & 63) == 0) { s_red4[(t >> 6) * 4] = p_sv0; s_red4[(t >> 6) * 4 + 1] = p_sv1; s_red4[(t >> 6) * 4 + 2] = p_sv2; s_red4[(t >> 6) * 4 + 3] = p_dot; }
        __syncthreads();
        sv0 = (s_red4[0] + s_red4[4]) + (s_red4[8] + s_red4[12]);
        sv1 = (s_red4[1] + s_red4[5]) + (s_red4[9] + s_red4[13]);
        sv2 = (s_red4[2] + s_red4[6]) + (s_red4[10] + s_red4[14]);
        dot = (s_red4[3] + s_red4[7]) + (s_red4[11] + s_red4[15]);
    }
    float sj[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 21; ++k) { sj[0] += s_djeff[k * 3]; sj[1] += s_djeff[k * 3 + 1]; sj[2] += s_djeff[k * 3 + 2]; }
    const float st[3] = {sv0 + sj[0], sv1 + sj[1], sv2 + sj[2]};
    if (t == 0) {
        if (d_trans) for (int c = 0; c < 3; ++c) d_trans[b * 3 + c] = st[c];
        if (d_scale && has_scale) d_scale[b] = dot;
    }
    RIH_BSTAMP(2);
    // 3. joint-side gradients in `src` order (16 chain joints + 5 tips)
    if (t < 63) {
        const int k = t / 3, c = t % 3;
        float d = sc * s_djeff[t];
        if (center_idx >= 0 && k == center_idx) d -= sc * st[c];
        s_dsrc[c_new_order[k] * 3 + c] = d;
    }
    __syncthreads();
    RIH_BSTAMP(3);
    // 4. through the skinning: dv_skin = sc dv_eff (+ the tip joints' gradients at their vertices); dv_tpose = T_v^T dv_skin;
    //    M_v = dv_skin x [v_t;1].  Round 5: the sixteen SE3s of the hand are WAVE-UNIFORM operands read from the workspace by
    //    scalar loads (constant address space: the forward wrote them, nobody writes them here) -- lane = vertex used to fetch
    //    each of them from LDS with 48 broadcast ds_read_b128 per vertex, and with four workgroups per CU the LDS pipe (8
    //    cycles per such instruction) was what the phase waited for: 65 k of the workgroup's 120 k cycles.  The separate
    //    dv_skin pass over the coordinates (14 k cycles) is folded in: a thread scales its own vertices.
    {
        const RIH_CONST_AS float* gG = (const RIH_CONST_AS float*)(w + OFF_G);
        float wv[4][NJ], Tr[4][9], d[4][3];
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            const int v = min(t + 256 * vi, NV - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // the vertex's 16 weights: four 16-byte requests
                const float4 w4 = *reinterpret_cast<const float4*>(m.weights + v * NJ + 4 * q);
                wv[vi][4 * q] = w4.x; wv[vi][4 * q + 1] = w4.y; wv[vi][4 * q + 2] = w4.z; wv[vi][4 * q + 3] = w4.w;
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) Tr[vi][e] = 0.f;
        }
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            const int v = t + 256 * vi;
            if (v >= NV) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x = sc * s_dvs[v * 3 + c];
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    if (c_special[q] == v) x += s_dsrc[(16 + q) * 3 + c];
                d[vi][c] = x;
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float g[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) g[e] = gG[j * 12 + e];
#pragma unroll
            for (int vi = 0; vi < 4; ++vi) {
                Tr[vi][0] += wv[vi][j] * g[0]; Tr[vi][1] += wv[vi][j] * g[1]; Tr[vi][2] += wv[vi][j] * g[2];
                Tr[vi][3] += wv[vi][j] * g[4]; Tr[vi][4] += wv[vi][j] * g[5]; Tr[vi][5] += wv[vi][j] * g[6];
                Tr[vi][6] += wv[vi][j] * g[8]; Tr[vi][7] += wv[vi][j] * g[9]; Tr[vi][8] += wv[vi][j] * g[10];
            }
        }
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            const int v = t + 256 * vi;
            if (v >= NV) continue;
            // M_v = dv_skin x [v_t; 1] is rank one: keep its two factors (19 KB for the mesh instead of 37 KB -- four workgroups
            // per CU instead of two) and form the products in step 5
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s_M[v * 6 + c] = d[vi][c];
                s_M[v * 6 + 3 + c] = w[OFF_VT + v * 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                s_dvs[v * 3 + c] = Tr[vi][c] * d[vi][0] + Tr[vi][3 + c] * d[vi][1] + Tr[vi][6 + c] * d[vi][2];     // dv_tpose
        }
    }
    __syncthreads();
    RIH_BSTAMP(4);
    // 5. dG = W^T M, [16 joints] x [778 vertices] x [12 elements], on v_mfma_f32_16x16x4_f32: A = four rows of the skinning
    //    weights (lane = (joint, vertex k): 256 contiguous bytes per instruction, ALL 49 requests of a lane issued before the
    //    first product), B = M_v = dv_skin x [v_t; 1] formed from its two factors in LDS (lane = (element, vertex k)); each
    //    wavefront takes every fourth group of four vertices, the four partial sums are joined through LDS in wavefront order.
    //    (Rounds 2-4: lane = (joint, three elements) with eight weight requests in flight -- 25 dependent L2 round trips per
    //    wavefront, 63 k of the workgroup's 104 k cycles; the phase timings of round 4 had this phase under the skinning's label.)
    {
        const int lane = t & 63, wave = t >> 6, kq = lane >> 4, col = lane & 15;
        constexpr int NKS = (NV + 3) / 4;               // 195 groups of four vertices
        constexpr int PER = (NKS + 3) / 4;              // 49 per wavefront
        float wa[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int v = 4 * (wave + 4 * i) + kq;
            wa[i] = (v < NV) ? m.weights[v * NJ + col] : 0.f;
        }
        const int r = col >> 2, c = col & 3;            // element (r, c) of the 3 x 4 block; columns 12..15 of the product are unused
        floatx4 acc0 = floatx4{0.f, 0.f, 0.f, 0.f}, acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int v = 4 * (wave + 4 * i) + kq;
            float bv = 0.f;
            if (v < NV && col < 12) {
                const float* f = &s_M[v * 6];
                bv = (c < 3) ? f[r] * f[3 + c] : f[r];
            }
            if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], bv, acc1, 0, 0, 0);
            else       acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], bv, acc0, 0, 0, 0);
        }
        // C/D layout: column = lane & 15 (element), row = 4 * (lane >> 4) + q (joint)
        __syncthreads();                    // s_dvs is needed below; the partials go to a separate scratch
        if (col < 12) {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_part[wave * 192 + (4 * kq + q) * 12 + col] = acc0[q] + acc1[q];
        }
        __syncthreads();
        if (t < NJ * 12) s_dG[t] = (s_part[t] + s_part[192 + t]) + (s_part[384 + t] + s_part[576 + t]);
    }
    __syncthreads();
    RIH_BSTAMP(5);
    // 6. joint-position gradients through the chain: dG_parent += d(joint i) x [jt_i; 1]^T (gathered per parent, children in
    //    index order), djt_i += R_parent^T d(joint i).  (Rounds 1-2 ran steps 6-7 on ONE thread: ~2400 dependent LDS accesses,
    //    ~80 us of the 150 us a workgroup took.)
    if (t < NJ * 12) {
        const int p = t / 12, e = t - p * 12, r = e >> 2, c = e & 3;
        float a = 0.f;
        for (int q = s_cstart[p]; q < s_cstart[p + 1]; ++q) {
            const int i = s_child[q];
            a += s_dsrc[i * 3 + r] * (c < 3 ? s_jt[i * 3 + c] : 1.f);
        }
        s_dG[t] += a;
    } else if (t < NJ * 12 + NJ * 3) {
        const int i = (t - NJ * 12) / 3, c = (t - NJ * 12) - i * 3;
        if (i == 0) {
            s_djt[c] += s_dsrc[c];
        } else {
            const float* P = &s_G[s_par[i] * 12];
            const float* dji = &s_dsrc[i * 3];
            s_djt[i * 3 + c] += P[c] * dji[0] + P[4 + c] * dji[1] + P[8 + c] * dji[2];
        }
    }
    __syncthreads();
    // 7a. global-transform gradients up the tree, one level at a time: a parent gathers dRg_i R_i^T + dtg_i tl_i^T (and dtg_i)
    //     from its children (index order), whose own gradients are complete by then
    for (int L = s_depth[16]; L >= 1; --L) {
        if (t < NJ * 12) {
            const int p = t / 12, e = t - p * 12, r = e >> 2, c = e & 3;
            if (s_depth[p] == L - 1) {
                float a = 0.f;
                for (int q = s_cstart[p]; q < s_cstart[p + 1]; ++q) {
                    const int i = s_child[q];
                    const float* R = &s_R[i * 9];
                    const float* jv = &s_jt[i * 3];
                    const float* dGi = &s_dG[i * 12];
                    if (c < 3) {
                        const float tl = jv[c] - (R[c * 3] * jv[0] + R[c * 3 + 1] * jv[1] + R[c * 3 + 2] * jv[2]);
                        a += dGi[r * 4] * R[c * 3] + dGi[r * 4 + 1] * R[c * 3 + 1] + dGi[r * 4 + 2] * R[c * 3 + 2] + dGi[r * 4 + 3] * tl;
                    } else {
                        a += dGi[r * 4 + 3];
                    }
                }
                s_dG[t] += a;
            }
        }
        __syncthreads();
    }
    // 7b. local terms of every joint in parallel: dR_i = dRl - dtl jt^T, djt_i += dtl - R_i^T dtl, with
    //     dRl = P_rot^T dRg, dtl = P_rot^T dtg (P = the parent's global transform; identity for the root)
    if (t < NJ * 12) {
        const int i = t / 12, e = t - i * 12, r = e >> 2, c = e & 3;
        const float* dGi = &s_dG[i * 12];
        float dtl[3];
        if (i > 0) {
            const float* P = &s_G[s_par[i] * 12];
            for (int q = 0; q < 3; ++q) dtl[q] = P[q] * dGi[3] + P[4 + q] * dGi[7] + P[8 + q] * dGi[11];
            if (c < 3) {
                const float dRl = P[r] * dGi[c] + P[4 + r] * dGi[4 + c] + P[8 + r] * dGi[8 + c];
                s_dR[i * 9 + r * 3 + c] = dRl - dtl[r] * s_jt[i * 3 + c];
            }
        } else {
            for (int q = 0; q < 3; ++q) dtl[q] = dGi[q * 4 + 3];
            if (c < 3) s_dR[r * 3 + c] = dGi[r * 4 + c] - dtl[r] * s_jt[c];
        }
        if (c == 3) {       // (three lanes per joint: r plays the coordinate)
            const float* R = &s_R[i * 9];
            s_djt[i * 3 + r] += dtl[r] - (R[r] * dtl[0] + R[3 + r] * dtl[1] + R[6 + r] * dtl[2]);
        }
    }
    __syncthreads();
    RIH_BSTAMP(6);
    if (wsb != nullptr) {
        // Split backward (rih_mano_bwd, default): the two contractions with the blend bases -- steps 8 and 10, 0.33 MFLOP per
        // hand against 1.26 MB of basis that one workgroup per hand re-reads from L2 (5.2 GB for 4096 hands: the 1.17 ms of
        // round 2) -- are left to mano_bwd_blend_kernel, which runs them as MFMA products of 16-hand chunks against basis
        // tiles in LDS.  Step 9 disappears: d_shape = shapedirs^T (dv_tpose + J_reg^T djt) = shapedirs^T dv_tpose + Js^T djt with
        // the folded regressor Js of the packed basis (480 MACs per hand instead of 37 k loads of J_reg).  Handed over per hand:
        // dv_tpose (coordinates padded to the packed basis' 2496), the joint-rotation gradients before the pose-blend term, the
        // axis-angle vector and the rest-joint gradients.
        float* o = wsb + (long long)b * BW_STRIDE;
        for (int i = t; i < NCP; i += 256) o[i] = (i < NVC) ? s_dvs[i] : 0.f;                       // dv_tpose
        if (t < NJ * 9) o[NCP + t] = s_dR[t];
        if (t < 48) o[NCP + 192 + t] = s_djt[t];
        if (t < 9 && d_root) d_root[(long long)b * 9 + t] = s_dR[t];
        RIH_BSTAMP(7);
        return;
    }
    // 8. pose-blend gradient: dpf[p] = sum_vc posedirs[vc][p] * dv_tpose[vc]
    if (t < NPF) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int vc = 0;
        for (; vc + 3 < NVC; vc += 4) {
            a0 += m.posedirs[(long long)vc * NPF + t] * s_dvs[vc];
            a1 += m.posedirs[(long long)(vc + 1) * NPF + t] * s_dvs[vc + 1];
            a2 += m.posedirs[(long long)(vc + 2) * NPF + t] * s_dvs[vc + 2];
            a3 += m.posedirs[(long long)(vc + 3) * NPF + t] * s_dvs[vc + 3];
        }
        for (; vc < NVC; ++vc) a0 += m.posedirs[(long long)vc * NPF + t] * s_dvs[vc];
        s_dpf[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (t < NPF) s_dR[9 + t] += s_dpf[t];
    // 9. joint-regressor term -> dv_shaped
    for (int i = t; i < NVC; i += 256) {
        const int v = i / 3, c = i - v * 3;
        float a = s_dvs[i];
        for (int j = 0; j < NJ; ++j) a += m.J_reg[j * NV + v] * s_djt[j * 3 + c];
        s_dvs[i] = a;
    }
    __syncthreads();
    // 10. shape gradient
    if (t < 250) {
        const int s = t % 10, part = t / 10;
        float a = 0.f;
        for (int vc = part; vc < NVC; vc += 25) a += m.shapedirs[vc * 10 + s] * s_dvs[vc];
        s_sh[t] = a;
    }
    __syncthreads();
    if (t < 10 && d_shape) {
        float a = 0.f;
        for (int part = 0; part < 25; ++part) a += s_sh[part * 10 + t];
        d_shape[b * 10 + t] = a;
    }
    // 11. rotations
    if (t < 9 && d_root) d_root[(long long)b * 9 + t] = s_dR[t];
    if (d_pose) {
        if (ncomp > 0) {
            if (t < 15) rodrigues_bwd(&s_axis[t * 3], &s_dR[(t + 1) * 9], &s_dax[t * 3]);
            __syncthreads();
            if (t < ncomp) {
                float a = 0.f;
                for (int k = 0; k < 45; ++k) a += s_dax[k] * m.comps[t * 45 + k];
                d_pose[(long long)b * ncomp + t] = a;
            }
        } else if (t < NPF) {
            d_pose[(long long)b * NPF + t] = s_dR[9 + t];
        }
    }
}

// ---- backward, parts 2 and 3: the contractions with the blend bases ----------------------------------------------------------
//   out[hand][148] = dv_tpose[hand][2496] x Bmat^T: columns 0..134 = the pose-feature gradient, 135..144 = the vertex part of
//   the shape gradient (the joint-regressor part is Js^T djt, added in the epilogue).
// Round 5 -- TILE major: a workgroup pins ONE of the 13 basis tiles ([148][192], zero rows up to 160; LDS pitch 196: the
// k-strided operand reads of 16 rows are bank-conflict free) and walks over chunks of 16 hands; what streams is the chunk's
// 12 KB tile of dv_tpose (the next one in flight in registers), not the 111 KB basis tile.  Rounds 3-4 ran this hand-chunk major
// like the large-batch forward -- every workgroup streamed all 13 basis tiles from L2, ~11 us per tile of which the ten
// 16 x 16 x 192 products on v_mfma_f32_16x16x4_f32 are 2.2: 145 us for 4096 hands, and 8 workgroups x 13 serial tiles for 128
// hands.  (The forward has a reason to be hand major there -- its pose chain would be repeated per tile; this product has none.)
// The price is that a hand's 148 sums now come from 13 workgroups: each writes its partial [16 hands][160] and
// mano_bwd_finish_kernel adds them in tile order (deterministic) in front of the epilogue it took over: pose-blend term into the
// joint-rotation gradients, Rodrigues backward, PCA projection (or the rotation-matrix gradients as they are), shape gradient.
constexpr int BL_P = 196;                       // LDS row pitch: 196 mod 64 = 4, so the 16 rows x 4 k of an operand read (bank =
                                                // 4 row + k) hit 64 different banks (194, rounds 2-3: pairs of lanes collided)
constexpr int BL_ROWS = 160;                    // basis rows incl. zero padding to ten 16-column blocks
constexpr int PART_W = 256;                     // floats of a (tile, hand) partial row: 128 whole sums + 4 k-quarters x 32
__global__ __launch_bounds__(256) void mano_bwd_blend_kernel(const float* __restrict__ pk, const float* __restrict__ wsb,
                                                             float* __restrict__ part, int B) {
    __shared__ __attribute__((aligned(16))) float s_B[BL_ROWS * BL_P];      // 125,440 B
    __shared__ __attribute__((aligned(16))) float s_V[2][HC * BL_P];        //  25,088 B: dv_tpose tiles of two chunks
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x, group = blockIdx.y, ngroups = gridDim.y;
    const int nchunks = (B + HC - 1) / HC;
    floatx4 hv[3];                              // 16 hands x 48 sixteen-byte items = 3 per thread
    auto issue = [&](int chunk) {
        const int h0 = chunk * HC;
        const float* vsrc = wsb + (long long)h0 * BW_STRIDE + tile * 192;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = t + 256 * u, k = i / 48, q = i - k * 48;
            hv[u] = floatx4{0.f, 0.f, 0.f, 0.f};        // rows of hands behind the end of the batch are zero
            if (h0 + k < B) hv[u] = *reinterpret_cast<const floatx4*>(vsrc + (long long)k * BW_STRIDE + 4 * q);
        }
    };
    auto land = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int i = t + 256 * u, k = i / 48, q = i - k * 48;
            *reinterpret_cast<floatx4*>(&s_V[buf][k * BL_P + 4 * q]) = hv[u];
        }
    };
    if (group < nchunks) issue(group);
    load_tile_192<BL_P>(s_B, pk + tile * 192, NCP, KP, t);
    for (int i = t; i < (BL_ROWS - KP) * BL_P; i += 256) s_B[KP * BL_P + i] = 0.f;     // rows 148..159 are zero
    if (group < nchunks) land(0);
    __syncthreads();
    int buf = 0;
    for (int chunk = group; chunk < nchunks; chunk += ngroups, buf ^= 1) {
        const bool more = chunk + ngroups < nchunks;
        if (more) issue(chunk + ngroups);
        // The ten column blocks of 16 basis rows over four wavefronts: blocks w and w + 4 whole, and the k-quarter w of blocks 8
        // and 9 (120 products each; 3 + 3 + 2 + 2 whole blocks left two wavefronts waiting at the barrier) -- four independent
        // accumulation chains that share the A operand read.
        floatx4 acc0 = floatx4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        {
            const float* a_rd = &s_V[buf][(lane & 15) * BL_P + (lane >> 4)];
            const float* b0 = s_B + (wave * 16 + (lane & 15)) * BL_P + (lane >> 4);
            const float* b1 = b0 + 64 * BL_P;
            const float* b8 = s_B + (128 + (lane & 15)) * BL_P + (lane >> 4) + 48 * wave;       // the wavefront's k-quarter
            const float* b9 = b8 + 16 * BL_P;
            // operands of twelve k-steps at a time in registers, the NEXT twelve requested before the products of the current
            // ones (the compiler's own schedule asked LDS two k-steps ahead and waited out its latency every four products)
            float av[2][12], bv0[2][12], bv1[2][12], bv8[12], bv9[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) { av[0][i] = a_rd[4 * i]; bv0[0][i] = b0[4 * i]; bv1[0][i] = b1[4 * i]; }
#pragma unroll
            for (int i = 0; i < 12; ++i) { bv8[i] = b8[4 * i]; bv9[i] = b9[4 * i]; }
#pragma unroll
            for (int seg = 0; seg < 4; ++seg) {
                const int cur = seg & 1, nxt = cur ^ 1;
                if (seg < 3) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int k = 4 * (12 * (seg + 1) + i);
                        av[nxt][i] = a_rd[k]; bv0[nxt][i] = b0[k]; bv1[nxt][i] = b1[k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (seg == wave) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv0[cur][i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv1[cur][i], acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv8[i], acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv9[i], acc3, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv0[cur][i], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][i], bv1[cur][i], acc1, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // C/D layout: column = lane & 15 (basis row within the block), row = 4 * (lane >> 4) + r (hand).  Partial row of a hand:
        // [0..127] the whole blocks, [128 + 32 q + c] the k-quarter q of columns 128 + c
        const int h0 = chunk * HC;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int h = h0 + 4 * (lane >> 4) + r;
            if (h >= B) continue;
            float* pr = part + ((long long)tile * B + h) * PART_W;
            pr[wave * 16 + (lane & 15)] = acc0[r];
            pr[64 + wave * 16 + (lane & 15)] = acc1[r];
            pr[128 + 32 * wave + (lane & 15)] = acc2[r];
            pr[144 + 32 * wave + (lane & 15)] = acc3[r];
        }
        if (more) land(buf ^ 1);        // the other buffer: its last readers passed the barrier of the previous iteration
        __syncthreads();
    }
}

// One WAVEFRONT per hand, four hands per workgroup: every global request of a hand (3 x 13 partial sums, rotation gradients, axis,
// rest-joint gradients) leaves in one batch, the PCA basis and the folded regressor are staged in LDS by the whole workgroup.
// (First version of this round: a workgroup per 16 hands looping ten times over 13 requests, then 45-step loops on global
// memory -- 24 us for 8 workgroups, 28 us for 256: a chain of round trips.)
constexpr int FH = 4;
__global__ __launch_bounds__(256) void mano_bwd_finish_kernel(Model m, const float* __restrict__ pk, int ncomp,
                                                              const float* __restrict__ pose, const float* __restrict__ wsb,
                                                              const float* __restrict__ part,
                                                              float* __restrict__ d_pose, float* __restrict__ d_shape, int B) {
    __shared__ float s_comps[45 * 45], s_js[480];
    __shared__ float s_dR[FH][144], s_ax[FH][48], s_dax[FH][48], s_djt[FH][48], s_out[FH][BL_ROWS], s_pose[FH][48];
    const int t = threadIdx.x, lane = t & 63, hl = t >> 6, h = blockIdx.x * FH + hl;
    const bool live = h < B;
    const float* o = wsb + (long long)(live ? h : 0) * BW_STRIDE + NCP;
    // partial sums of the hand: columns lane and 64 + lane whole (13 tiles); columns 128 + (lane & 31) in four k-quarters, of
    // which the lane's half wave takes two
    float p[2][NTILES], pq[2][NTILES], rr[3], ps = 0.f, hm = 0.f, dj = 0.f;
    const int c32 = lane & 31, qh = lane >> 5;
#pragma unroll
    for (int tl = 0; tl < NTILES; ++tl) {
        const float* pr = part + ((long long)tl * B + (live ? h : 0)) * PART_W;
        p[0][tl] = live ? pr[lane] : 0.f;
        p[1][tl] = live ? pr[64 + lane] : 0.f;
        pq[0][tl] = live ? pr[128 + 32 * (2 * qh) + c32] : 0.f;
        pq[1][tl] = live ? pr[128 + 32 * (2 * qh + 1) + c32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) rr[u] = (live && lane + 64 * u < 144) ? o[lane + 64 * u] : 0.f;
    if (live && lane < 48) dj = o[192 + lane];
    if (ncomp > 0) {
        if (live && lane < ncomp) ps = pose[(long long)h * ncomp + lane];
        if (lane < 45) hm = m.hands_mean[lane];
        for (int i = t; i < ncomp * 45; i += 256) s_comps[i] = m.comps[i];
    }
    for (int i = t; i < 480; i += 256) s_js[i] = pk[PK_JS + i];
    {       // tile order, then quarter order: deterministic
        float a0 = p[0][0], a1 = p[1][0], q0 = pq[0][0], q1 = pq[1][0];
#pragma unroll
        for (int tl = 1; tl < NTILES; ++tl) { a0 += p[0][tl]; a1 += p[1][tl]; q0 += pq[0][tl]; q1 += pq[1][tl]; }
        float q = q0 + q1;
        q += __shfl_xor(q, 32, 64);
        s_out[hl][lane] = a0;
        s_out[hl][64 + lane] = a1;
        if (qh == 0) s_out[hl][128 + c32] = q;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (lane + 64 * u < 144) s_dR[hl][lane + 64 * u] = rr[u];
    if (lane < 48) { s_pose[hl][lane] = ps; s_djt[hl][lane] = dj; }
    __syncthreads();
    if (ncomp > 0 && lane < 48) {       // axis-angle = hands_mean + pose x comps (for the Rodrigues backward; ready at the next barrier)
        float a = hm;
        if (lane < 45)
            for (int c = 0; c < ncomp; ++c) a += s_pose[hl][c] * s_comps[c * 45 + lane];
        s_ax[hl][lane] = a;
    }
    if (lane < 10 && live && d_shape) {
        float a = s_out[hl][NPF + lane];
        for (int q = 0; q < 48; ++q) a += s_djt[hl][q] * s_js[q * 10 + lane];      // + Js^T djt
        d_shape[(long long)h * 10 + lane] = a;
    }
    if (d_pose == nullptr) return;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int e = lane + 64 * u;
        if (e < NPF) s_dR[hl][9 + e] += s_out[hl][e];
    }
    __syncthreads();
    if (ncomp > 0) {
        if (lane < 15) rodrigues_bwd(&s_ax[hl][lane * 3], &s_dR[hl][(lane + 1) * 9], &s_dax[hl][lane * 3]);
        __syncthreads();
        if (lane < ncomp && live) {
            float a = 0.f;
#pragma unroll 15
            for (int k = 0; k < 45; ++k) a += s_dax[hl][k] * s_comps[lane * 45 + k];
            d_pose[(long long)h * ncomp + lane] = a;
        }
    } else if (live) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int e = lane + 64 * u;
            if (e < NPF) d_pose[(long long)h * NPF + e] = s_dR[hl][9 + e];
        }
    }
}

Model to_model(const rih_mano_model* m) {
    Model o;
    o.comps = m->comps; o.hands_mean = m->hands_mean; o.shapedirs = m->shapedirs; o.posedirs = m->posedirs;
    o.v_template = m->v_template; o.J_reg = m->J_reg; o.weights = m->weights;
    for (int i = 0; i < 16; ++i) o.parent[i] = m->parent[i];
    // (model_ok: parent[0] < 0, 0 <= parent[i] < i -- the loops below end)
    int mx = 0, n = 0;
    for (int i = 0; i < 16; ++i) {
        int d = 0;
        for (int pp = o.parent[i]; pp >= 0 && d < 16; pp = o.parent[pp]) ++d;
        o.depth[i] = d;
        if (i > 0 && d > mx) mx = d;
    }
    o.depth[16] = mx;
    for (int p = 0; p < 16; ++p) {
        o.cstart[p] = n;
        for (int i = 1; i < 16; ++i)
            if (o.parent[i] == p && n < 15) o.child[n++] = i;
    }
    o.cstart[16] = n;
    for (int q = n; q < 16; ++q) o.child[q] = 0;
    return o;
}

bool model_ok(const rih_mano_model* m) {
    if (!m || !m->hands_mean || !m->shapedirs || !m->posedirs || !m->v_template || !m->J_reg || !m->weights) return false;
    if (m->parent[0] >= 0) return false;
    for (int i = 1; i < 16; ++i)
        if (m->parent[i] < 0 || m->parent[i] >= i) return false;
    return true;
}

}  // namespace

extern "C" int64_t rih_mano_ws_floats(int B) { return B > 0 ? (int64_t)B * WS_STRIDE : 0; }
// per hand: the hand-off of the per-hand kernel (BW_STRIDE) + the 13 tiles' partial sums of the blend products
extern "C" int64_t rih_mano_bwd_ws_floats(int B) { return B > 0 ? (int64_t)B * (BW_STRIDE + NTILES * PART_W) : 0; }

static long long* g_mano_dbg = nullptr;
// development aid: device buffer of 13 x 16 int64 that receives phase timestamps of the fused forward (NULL = off)
extern "C" int rih_mano_debug_stamps(long long* buf) { g_mano_dbg = buf; return 0; }

extern "C" int64_t rih_mano_pack_floats(void) { return PK_FLOATS; }

extern "C" int rih_mano_pack(const rih_mano_model* m, float* packed, void* stream) {
    if (!model_ok(m) || !packed || ((uintptr_t)packed % 16) != 0) return RIH_EINVAL;
    const Model mm = to_model(m);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mano_pack_basis_kernel, dim3(1024), dim3(256), 0, s, mm, packed);
    hipLaunchKernelGGL(mano_pack_joints_kernel, dim3(132), dim3(256), 0, s, mm, packed);
    hipLaunchKernelGGL(mano_pack_special_kernel, dim3(28), dim3(256), 0, s, packed);
    return (int)hipGetLastError();
}

extern "C" int rih_mano_fwd(const rih_mano_model* m, const float* packed, const float* root, const float* pose, int ncomp,
                            const float* shape, const float* trans, const float* scale, int center_idx, int new_skel,
                            float* v, float* j, float* ws, int B, int variant, void* stream) {
    if (!model_ok(m) || !root || !pose || !shape || !v || !j || B < 1) return RIH_EINVAL;
    if (ncomp < 0 || ncomp > 45 || center_idx >= 21) return RIH_EINVAL;
    if (ncomp > 0 && !m->comps) return RIH_EINVAL;
    const Model mm = to_model(m);
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0 || variant == 2 || variant == 3) {
        // ONE launch.  Workgroups = 13 vertex tiles x hand groups; about two workgroups per CU in total, so that the 111 KB
        // basis tile a workgroup pins is amortised over several 16-hand chunks when the batch is large.
        if (!packed || ((uintptr_t)packed % 16) != 0) return RIH_EINVAL;
        const int nchunks = (B + HC - 1) / HC;
        if (variant == 2 || (variant == 0 && nchunks >= 256)) {
            // hand-chunk major: the pose work of a chunk once, basis tiles streamed from L2 (one workgroup per CU: LDS)
            hipLaunchKernelGGL(mano_fused_kernel<true>, dim3(nchunks < 256 ? nchunks : 256), dim3(256), 0, s, mm, packed, root,
                               pose, ncomp, shape, trans, scale, center_idx, new_skel, v, j, ws, B, g_mano_dbg);
            return (int)hipGetLastError();
        }
        int groups = (512 + NTILES - 1) / NTILES;
        if (groups > nchunks) groups = nchunks;
        hipLaunchKernelGGL(mano_fused_kernel<false>, dim3(NTILES, groups), dim3(256), 0, s, mm, packed, root, pose, ncomp,
                           shape, trans, scale, center_idx, new_skel, v, j, ws, B, g_mano_dbg);
        return (int)hipGetLastError();
    }
    if (variant != 1 || !ws) return RIH_EINVAL;     // round-1 two-kernel forward (A/B timing): needs the workspace
    hipLaunchKernelGGL(mano_pose_kernel, dim3(B), dim3(256), 0, s, mm, root, pose, ncomp, shape, trans, scale,
                       center_idx, new_skel, j, ws);
    // enough workgroups to cover 256 CUs, as few posedirs-tile reloads as that allows
    int chunks = (1024 + NTILES - 1) / NTILES;
    if (chunks > (B + 3) / 4) chunks = (B + 3) / 4;
    if (chunks < 1) chunks = 1;
    const int hpb = (B + chunks - 1) / chunks;
    chunks = (B + hpb - 1) / hpb;
    hipLaunchKernelGGL(mano_vertex_kernel, dim3(NTILES, chunks), dim3(256), 0, s, mm, v, ws, B, hpb);
    return (int)hipGetLastError();
}

extern "C" int rih_mano_bwd(const rih_mano_model* m, const float* packed, const float* root, const float* pose, int ncomp,
                            const float* shape, const float* trans, const float* scale, int center_idx, int new_skel,
                            const float* dv, const float* dj, const float* ws, float* d_root, float* d_pose,
                            float* d_shape, float* d_trans, float* d_scale, float* ws_bwd, int B, void* stream) {
    (void)root; (void)shape; (void)trans;
    if (!model_ok(m) || !pose || !dv || !dj || !ws || B < 1) return RIH_EINVAL;
    if (ncomp < 0 || ncomp > 45 || center_idx >= 21) return RIH_EINVAL;
    if (ncomp > 0 && !m->comps) return RIH_EINVAL;
    if (ws_bwd != nullptr && (!packed || ((uintptr_t)packed & 15) || ((uintptr_t)ws_bwd & 15))) return RIH_EINVAL;
    if ((uintptr_t)ws & 15) return RIH_EINVAL;          // the per-hand kernel reads the hand's SE3s with 16-byte scalar loads
    const Model mm = to_model(m);
    hipStream_t s = (hipStream_t)stream;
    // ws_bwd == NULL: the one-kernel backward of round 1 (one workgroup per hand does everything; kept for A/B timing)
    hipLaunchKernelGGL(mano_bwd_kernel, dim3(B), dim3(256), 0, s, mm, pose, ncomp, center_idx, new_skel, scale ? 1 : 0, dv, dj,
                       ws, d_root, d_pose, d_shape, d_trans, d_scale, ws_bwd, g_mano_dbg);
    if (ws_bwd != nullptr && (d_pose != nullptr || d_shape != nullptr)) {
        const int nchunks = (B + HC - 1) / HC;
        // 13 tiles x groups of chunks: as many groups as keep the launch within one workgroup per CU (LDS: one fits) -- 19 on
        // 256 CUs; a 20th would put four workgroups into a second round
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            cus < 1)
            cus = 256;
        int groups = cus / NTILES;
        if (groups < 1) groups = 1;
        if (groups > nchunks) groups = nchunks;
        float* part = ws_bwd + (long long)B * BW_STRIDE;
        hipLaunchKernelGGL(mano_bwd_blend_kernel, dim3(NTILES, groups), dim3(256), 0, s, packed, ws_bwd, part, B);
        hipLaunchKernelGGL(mano_bwd_finish_kernel, dim3((B + FH - 1) / FH), dim3(256), 0, s, mm, packed, ncomp, pose, ws_bwd, part,
                           d_pose, d_shape, B);
    }
    return (int)hipGetLastError();
}
