// rih_loss.hip -- fused training loss of the mesh decoder (core/Loss.py: GraphLoss.calc_loss + calc_loss_GCN) for gfx950.
//
// One workgroup per (image, hand).  The hand's predicted and ground-truth meshes (2 x 778 x 3 floats) live in LDS; the
// kernel produces in one pass (a) the raw sums of the seven loss terms of this hand and (b) the gradient of the weighted
// total with respect to every prediction tensor, so that neither the forward nor the backward of the loss launches
// anything else:
//   vertex terms      SmoothL1(v3d), MSE(v2d / img * 2 - 1)                                   (Loss.py:108-110)
//   joint term        SmoothL1(J v3d_pred, J v3d_gt), J = 21 x 778 regressor incl. tips      (Loss.py:38-53, 105-111)
//   face terms        normal consistency <e_pred / |e_pred|, n_gt> and edge length, 1538 faces x 3 edges  (Loss.py:68-102)
//   coarse terms      SmoothL1 / MSE of the 252-vertex prediction against the ground truth gathered into graph order
//                     and average-pooled pairwise (vert_to_GCN + mesh_downsample, Loss.py:139-164)
// Face gradients are written per face to LDS and gathered per vertex through a vertex->(face, corner) adjacency list
// in a fixed order: no atomics, bit-reproducible.  Bound: latency (a few hundred KB per workgroup); it replaces ~600
// torch elementwise / index / fill launches per training step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXV = 800, MAXF = 1600, MAXJ = 24, MAXPOOL = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the 256 threads of the block; the result is valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float sl1(float x) {          // nn.SmoothL1Loss(beta=1)
    const float a = fabsf(x);
    return a < 1.f ? 0.5f * x * x : a - 0.5f;
}
__device__ __forceinline__ float dsl1(float x) { return fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f); }

struct Topo {
    const int32_t* faces;
    const int32_t* vptr;
    const int32_t* vlist;
    const float* J;
    const int32_t* perm;
    int V, F, NJ, Vc, pool;
};

struct Weights {    // weight of each term in the total, already divided by its element count and by 2 (hand average)
    float v2d, v3d, joint, norm, edge, c3d, c2d;
};

__global__ __launch_bounds__(TPB) void mesh_loss_kernel(Topo tp, const float* __restrict__ v3d_pred,
                                                        const float* __restrict__ v2d_pred,
                                                        const float* __restrict__ c3d_pred,
                                                        const float* __restrict__ c2d_pred,
                                                        const float* __restrict__ v3d_gt,
                                                        const float* __restrict__ v2d_gt,
                                                        const float* __restrict__ gt_shift,
                                                        const float* __restrict__ wdev, float img,
                                                        float* __restrict__ g_v3d, float* __restrict__ g_v2d,
                                                        float* __restrict__ g_c3d, float* __restrict__ g_c2d,
                                                        float* __restrict__ partial) {
    __shared__ float s_vp[MAXV * 3], s_vg[MAXV * 3];
    __shared__ float s_fg[MAXF * 9];
    __shared__ float s_gj[MAXJ * 3];
    __shared__ float s_red[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const int V = tp.V, F = tp.F;
    // term weights live in device memory so that a captured hipGraph follows the caller's epoch gate (edge term)
    const Weights w{wdev[0], wdev[1], wdev[2], wdev[3], wdev[4], wdev[5], wdev[6]};
    const float* vp = v3d_pred + (long long)b * V * 3;
    const float* vg = v3d_gt + (long long)b * V * 3;
    float sh[3] = {0.f, 0.f, 0.f};
    if (gt_shift != nullptr) {
        sh[0] = gt_shift[b * 3 + 0];
        sh[1] = gt_shift[b * 3 + 1];
        sh[2] = gt_shift[b * 3 + 2];
    }
    for (int i = t; i < V * 3; i += TPB) {
        s_vp[i] = vp[i];
        s_vg[i] = vg[i] + sh[i % 3];
    }
    __syncthreads();

    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // v2d, v3d, joint, norm, edge, c3d, c2d (raw sums)
    constexpr int VPT = (MAXV + TPB - 1) / TPB;              // vertices per thread
    float gv[VPT][3];
    // ---- vertex terms
    const float s2 = 2.f / img;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int v = t + TPB * k;
        gv[k][0] = gv[k][1] = gv[k][2] = 0.f;
        if (v < V) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = s_vp[v * 3 + c] - s_vg[v * 3 + c];
                acc[1] += sl1(d);
                gv[k][c] = w.v3d * dsl1(d);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long o = ((long long)b * V + v) * 2 + c;
                const float d = (v2d_pred[o] * s2 - 1.f) - (v2d_gt[o] * s2 - 1.f);
                acc[0] += d * d;
                g_v2d[o] = w.v2d * 2.f * d * s2;
            }
        }
    }
    // ---- joint term: jp = J vp, jg = J vg  (21 x 3 each)
    for (int j = 0; j < tp.NJ; ++j) {
        float p[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
        const float* Jr = tp.J + (long long)j * V;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
            const int v = t + TPB * k;
            if (v < V) {
                const float jw = Jr[v];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p[c] += jw * s_vp[v * 3 + c];
                    g[c] += jw * s_vg[v * 3 + c];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = block_sum(p[c], s_red) - block_sum(g[c], s_red);
            if (t == 0) {
                acc[2] += sl1(d);
                s_gj[j * 3 + c] = w.joint * dsl1(d);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int v = t + TPB * k;
        if (v < V) {
            for (int j = 0; j < tp.NJ; ++j) {
                const float jw = tp.J[(long long)j * V + v];
#pragma unroll
                for (int c = 0; c < 3; ++c) gv[k][c] += jw * s_gj[j * 3 + c];
            }
        }
    }
    // ---- face terms: per face the gradient with respect to its three edge vectors, into LDS
    for (int f = t; f < F; f += TPB) {
        const int i0 = tp.faces[f * 3 + 0], i1 = tp.faces[f * 3 + 1], i2 = tp.faces[f * 3 + 2];
        float ep[3][3], eg[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p0 = s_vp[i0 * 3 + c], p1 = s_vp[i1 * 3 + c], p2 = s_vp[i2 * 3 + c];
            const float q0 = s_vg[i0 * 3 + c], q1 = s_vg[i1 * 3 + c], q2 = s_vg[i2 * 3 + c];
            ep[0][c] = p0 - p1; ep[1][c] = p1 - p2; ep[2][c] = p2 - p0;
            eg[0][c] = q0 - q1; eg[1][c] = q1 - q2; eg[2][c] = q2 - q0;
        }
        float n[3] = {eg[0][1] * eg[1][2] - eg[0][2] * eg[1][1], eg[0][2] * eg[1][0] - eg[0][0] * eg[1][2],
                      eg[0][0] * eg[1][1] - eg[0][1] * eg[1][0]};
        const float nl = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-12f);      // F.normalize eps
        n[0] /= nl; n[1] /= nl; n[2] /= nl;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float len = sqrtf(ep[e][0] * ep[e][0] + ep[e][1] * ep[e][1] + ep[e][2] * ep[e][2]);
            const float leng = sqrtf(eg[e][0] * eg[e][0] + eg[e][1] * eg[e][1] + eg[e][2] * eg[e][2]);
            const float den = fmaxf(len, 1e-12f);
            const float u[3] = {ep[e][0] / den, ep[e][1] / den, ep[e][2] / den};
            const float d = u[0] * n[0] + u[1] * n[1] + u[2] * n[2];
            acc[3] += sl1(d);
            const float dl = len - leng;
            acc[4] += sl1(dl);
            const float gn = w.norm * dsl1(d), ge = w.edge * dsl1(dl);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // d<u,n>/de = (n - u <u,n>) / |e|   (|e| > eps);   d|e|/de = e / |e| (0 at e = 0, as torch)
                const float dn = (len > 1e-12f) ? (n[c] - u[c] * d) / den : n[c] / den;
                const float de = (len > 0.f) ? ep[e][c] / len : 0.f;
                s_fg[f * 9 + e * 3 + c] = gn * dn + ge * de;
            }
        }
    }
    __syncthreads();
    // ---- gather face gradients per vertex (fixed order), write the 3-D vertex gradient
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int v = t + TPB * k;
        if (v < V) {
            for (int q = tp.vptr[v]; q < tp.vptr[v + 1]; ++q) {
                const int fc = tp.vlist[q];
                const int f = fc / 3, c0 = fc - f * 3;
                // corner c0 of the face is the head of edge c0 (e_c0 = v_c0 - v_{c0+1}) and the tail of edge c0-1
                const float* ga = s_fg + f * 9 + c0 * 3;
                const float* gb = s_fg + f * 9 + ((c0 + 2) % 3) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) gv[k][c] += ga[c] - gb[c];
            }
            float* o = g_v3d + ((long long)b * V + v) * 3;
            o[0] = gv[k][0]; o[1] = gv[k][1]; o[2] = gv[k][2];
        }
    }
    // ---- coarse level: ground truth in graph order, average-pooled pairwise log2(pool) times
    for (int kc = t; kc < tp.Vc; kc += TPB) {
        float a3[MAXPOOL][3], a2[MAXPOOL][2];
        for (int i = 0; i < tp.pool; ++i) {
            const int v = tp.perm[kc * tp.pool + i];
#pragma unroll
            for (int c = 0; c < 3; ++c) a3[i][c] = s_vg[v * 3 + c];
#pragma unroll
            for (int c = 0; c < 2; ++c) a2[i][c] = v2d_gt[((long long)b * V + v) * 2 + c];
        }
        for (int m = tp.pool; m > 1; m >>= 1)
            for (int i = 0; i < m / 2; ++i) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a3[i][c] = (a3[2 * i][c] + a3[2 * i + 1][c]) * 0.5f;
#pragma unroll
                for (int c = 0; c < 2; ++c) a2[i][c] = (a2[2 * i][c] + a2[2 * i + 1][c]) * 0.5f;
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long long o = ((long long)b * tp.Vc + kc) * 3 + c;
            const float d = c3d_pred[o] - a3[0][c];
            acc[5] += sl1(d);
            g_c3d[o] = w.c3d * dsl1(d);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long long o = ((long long)b * tp.Vc + kc) * 2 + c;
            const float d = (c2d_pred[o] * s2 - 1.f) - (a2[0][c] * s2 - 1.f);
            acc[6] += d * d;
            g_c2d[o] = w.c2d * 2.f * d * s2;
        }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const float s = block_sum(acc[i], s_red);
        if (t == 0) partial[b * 8 + i] = s;
    }
}

// out[0] = total; out[1..7] = the seven terms as the reference reports them (mean over elements, averaged over hands)
__global__ void mesh_loss_final_kernel(const float* __restrict__ pl, const float* __restrict__ pr, int B,
                                       const float* __restrict__ wdev, const float* __restrict__ cdev,
                                       float* __restrict__ out) {
    const int i = threadIdx.x;
    __shared__ float s_t[8];
    if (i < 7) {
        float sl = 0.f, sr = 0.f;
        for (int b = 0; b < B; ++b) { sl += pl[b * 8 + i]; sr += pr[b * 8 + i]; }
        const float wi = wdev[i], ci = cdev[i];
        s_t[i] = wi * (sl + sr);
        out[1 + i] = 0.5f * (sl + sr) / ci;
    }
    __syncthreads();
    if (i == 0) out[0] = ((s_t[0] + s_t[1]) + (s_t[2] + s_t[3])) + ((s_t[4] + s_t[5]) + s_t[6]);
}

}  // namespace

extern "C" int rih_mesh_loss(const rih_mesh_topo* tp, const float* v3d_pred, const float* v2d_pred, const float* c3d_pred,
                             const float* c2d_pred, const float* v3d_gt, const float* v2d_gt, const float* gt_shift,
                             const float* term_weights, float img_size, float* g_v3d, float* g_v2d, float* g_c3d,
                             float* g_c2d, float* partial, int B, void* stream) {
    if (!tp || !v3d_pred || !v2d_pred || !c3d_pred || !c2d_pred || !v3d_gt || !v2d_gt || !term_weights || !g_v3d ||
        !g_v2d || !g_c3d || !g_c2d || !partial || B < 1)
        return RIH_EINVAL;
    if (tp->V < 1 || tp->V > MAXV || tp->F < 1 || tp->F > MAXF || tp->NJ < 1 || tp->NJ > MAXJ || tp->Vc < 1 ||
        tp->pool < 1 || tp->pool > MAXPOOL || (tp->pool & (tp->pool - 1)) != 0 || img_size <= 0.f)
        return RIH_EINVAL;
    Topo t{tp->faces, tp->vptr, tp->vlist, tp->J, tp->perm, tp->V, tp->F, tp->NJ, tp->Vc, tp->pool};
    hipLaunchKernelGGL(mesh_loss_kernel, dim3(B), dim3(TPB), 0, (hipStream_t)stream, t, v3d_pred, v2d_pred, c3d_pred,
                       c2d_pred, v3d_gt, v2d_gt, gt_shift, term_weights, img_size, g_v3d, g_v2d, g_c3d, g_c2d, partial);
    return (int)hipGetLastError();
}

extern "C" int rih_mesh_loss_final(const float* partial_left, const float* partial_right, int B, const float* term_weights,
                                   const float* counts, float* out, void* stream) {
    if (!partial_left || !partial_right || !term_weights || !counts || !out || B < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(mesh_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial_left, partial_right, B,
                       term_weights, counts, out);
    return (int)hipGetLastError();
}
