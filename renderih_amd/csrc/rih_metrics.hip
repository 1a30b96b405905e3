// rih_metrics.hip -- evaluation metrics of the hand-mesh predictions for gfx950: joint regression, root alignment,
// bone-length rescaling, per-joint / per-vertex errors and the Procrustes-aligned errors (PA-MPJPE, PA-MPVPE).
//
// Replaces, per hand, the torch sequence of apps/eval_interhand.py:334-415 / common/utils/intag_eval.py:217-283
// (`eval_hand2`) including `batch_compute_similarity_transform_torch` (torch.svd on the CPU in the reference:
// intag_eval.py:196-208 moves the tensors to the host first), so that evaluation stays on the GPU.
//
// One workgroup (256 threads) per image: the predicted and the ground-truth mesh (2 x 778 x 3 floats) sit in LDS; the
// 21 joints are 63 wavefront-wide dot products over the vertices; the similarity alignment needs only 16 moments of
// the point pairs (block-reduced in double), after which one thread solves the 4x4 eigen-problem of
// rih_procrustes.h and all threads apply the transform.  Latency-bound (19 KB in, 6.4 KB out per workgroup).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"
#include "rih_procrustes.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXV = 1024, MAXJ = 32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Shared {
    float pv[MAXV * 3], gv[MAXV * 3];
    float pj[MAXJ * 3], gj[MAXJ * 3];
    double red[TPB / 64][16];
    double xf[13];          // R (9, row-major), scale, t (3)
};

// J (NJ x V) times a mesh in LDS: each wavefront takes joints wave, wave+4, ...; lanes stride over the vertices
__device__ void regress_joints(const float* __restrict__ Jreg, const float* mesh, float* joints, int V, int NJ) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = wave; j < NJ; j += TPB / 64) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int v = lane; v < V; v += 64) {
            const float w = Jreg[(long long)j * V + v];
            a0 += w * mesh[3 * v];
            a1 += w * mesh[3 * v + 1];
            a2 += w * mesh[3 * v + 2];
        }
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        a2 = wave_sum(a2);
        if (lane == 0) { joints[3 * j] = a0; joints[3 * j + 1] = a1; joints[3 * j + 2] = a2; }
    }
}

// Errors of one point set (joints or vertices) + its similarity-aligned mean error.
//   x1 = P - root_p, x2 = G - root_g;  err_ori = |x1 - x2|, err = |x1 * sc - x2|;
//   *pa = mean |s R x1 + t - x2| with (s, R, t) the optimal similarity transform of x1 onto x2.
__device__ void point_set_errors(Shared& sh, const float* P, const float* G, int n, const float root_p[3],
                                 const float root_g[3], float sc, float* __restrict__ err_ori, float* __restrict__ err,
                                 float* __restrict__ pa) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        float x1[3], x2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { x1[a] = P[3 * i + a] - root_p[a]; x2[a] = G[3 * i + a] - root_g[a]; }
        float eo = 0.f, es = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float d0 = x1[a] - x2[a], d1 = x1[a] * sc - x2[a];
            eo += d0 * d0;
            es += d1 * d1;
        }
        if (err_ori != nullptr) err_ori[i] = sqrtf(eo);
        if (err != nullptr) err[i] = sqrtf(es);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            acc[a] += (double)x1[a];
            acc[3 + a] += (double)x2[a];
            acc[15] += (double)x1[a] * (double)x1[a];
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[6 + 3 * a + b] += (double)x1[a] * (double)x2[b];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double s = wave_sum_d(acc[k]);
        if (lane == 0) sh.red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m[16];
        for (int k = 0; k < 16; ++k) m[k] = sh.red[0][k] + sh.red[1][k] + sh.red[2][k] + sh.red[3][k];
        double s12[3][3], R[3][3], s, t[3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) s12[a][b] = m[6 + 3 * a + b];
        rih_similarity_from_moments(n, &m[0], &m[3], s12, m[15], R, &s, t);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) sh.xf[3 * a + b] = R[a][b];
        sh.xf[9] = s;
        sh.xf[10] = t[0]; sh.xf[11] = t[1]; sh.xf[12] = t[2];
    }
    __syncthreads();
    double e = 0.0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        double x1[3], x2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { x1[a] = (double)(P[3 * i + a] - root_p[a]); x2[a] = (double)(G[3 * i + a] - root_g[a]); }
        double d2 = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double h = sh.xf[9] * (sh.xf[3 * a] * x1[0] + sh.xf[3 * a + 1] * x1[1] + sh.xf[3 * a + 2] * x1[2]) +
                             sh.xf[10 + a] - x2[a];
            d2 += h * h;
        }
        e += sqrt(d2);
    }
    e = wave_sum_d(e);
    __syncthreads();            // sh.red is reused
    if (lane == 0) sh.red[wave][0] = e;
    __syncthreads();
    if (threadIdx.x == 0) *pa = (float)((sh.red[0][0] + sh.red[1][0] + sh.red[2][0] + sh.red[3][0]) / n);
    __syncthreads();
}

__global__ __launch_bounds__(TPB) void hand_metrics_kernel(const float* __restrict__ v_pred, const float* __restrict__ v_gt,
                                                           const float* __restrict__ j_pred, const float* __restrict__ j_gt,
                                                           const float* __restrict__ Jreg, int V, int NJ, int root_idx,
                                                           int bone_a, int bone_b, float* __restrict__ j_err_ori,
                                                           float* __restrict__ v_err_ori, float* __restrict__ j_err,
                                                           float* __restrict__ v_err, float* __restrict__ pa,
                                                           float* __restrict__ j_pred_out) {
    __shared__ Shared sh;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < V * 3; i += TPB) {
        sh.pv[i] = v_pred[(long long)b * V * 3 + i];
        sh.gv[i] = v_gt[(long long)b * V * 3 + i];
    }
    if (j_pred != nullptr)
        for (int i = threadIdx.x; i < NJ * 3; i += TPB) sh.pj[i] = j_pred[(long long)b * NJ * 3 + i];
    if (j_gt != nullptr)
        for (int i = threadIdx.x; i < NJ * 3; i += TPB) sh.gj[i] = j_gt[(long long)b * NJ * 3 + i];
    __syncthreads();
    if (j_pred == nullptr) regress_joints(Jreg, sh.pv, sh.pj, V, NJ);
    if (j_gt == nullptr) regress_joints(Jreg, sh.gv, sh.gj, V, NJ);
    __syncthreads();
    if (j_pred_out != nullptr)
        for (int i = threadIdx.x; i < NJ * 3; i += TPB) j_pred_out[(long long)b * NJ * 3 + i] = sh.pj[i];
    float root_p[3], root_g[3], lp = 0.f, lg = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        root_p[a] = sh.pj[3 * root_idx + a];
        root_g[a] = sh.gj[3 * root_idx + a];
        const float dp = sh.pj[3 * bone_a + a] - sh.pj[3 * bone_b + a], dg = sh.gj[3 * bone_a + a] - sh.gj[3 * bone_b + a];
        lp += dp * dp;
        lg += dg * dg;
    }
    const float sc = sqrtf(lg) / sqrtf(lp);             // length_gt / length_pred   (intag_eval.py:226-244)
    point_set_errors(sh, sh.pj, sh.gj, NJ, root_p, root_g, sc, j_err_ori ? j_err_ori + (long long)b * NJ : nullptr,
                     j_err ? j_err + (long long)b * NJ : nullptr, pa + 2 * b);
    point_set_errors(sh, sh.pv, sh.gv, V, root_p, root_g, sc, v_err_ori ? v_err_ori + (long long)b * V : nullptr,
                     v_err ? v_err + (long long)b * V : nullptr, pa + 2 * b + 1);
}

}  // namespace

extern "C" int rih_hand_metrics(const float* v_pred, const float* v_gt, const float* j_pred, const float* j_gt,
                                const float* Jreg, int B, int V, int NJ, int root_idx, int bone_a, int bone_b,
                                float* j_err_ori, float* v_err_ori, float* j_err, float* v_err, float* pa,
                                float* j_pred_out, void* stream) {
    if (!v_pred || !v_gt || !pa || B < 1 || V < 3 || V > MAXV || NJ < 3 || NJ > MAXJ) return RIH_EINVAL;
    if ((!j_pred || !j_gt) && !Jreg) return RIH_EINVAL;
    if (root_idx < 0 || root_idx >= NJ || bone_a < 0 || bone_a >= NJ || bone_b < 0 || bone_b >= NJ) return RIH_EINVAL;
    hipLaunchKernelGGL(hand_metrics_kernel, dim3(B), dim3(TPB), 0, (hipStream_t)stream, v_pred, v_gt, j_pred, j_gt, Jreg, V,
                       NJ, root_idx, bone_a, bone_b, j_err_ori, v_err_ori, j_err, v_err, pa, j_pred_out);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Contact deviation of apps/eval_interhand.py:481-490 (utils/eval_metrics.py:30-50 `compute_idx` / `compute_cdev`): every
// ground-truth right-hand vertex finds its nearest ground-truth left-hand vertex (pytorch3d knn_points, K = 1); where that
// distance is within `contact` (3 mm) the predicted meshes should keep the pair together: out[b] = mean over those
// vertices of |pred_left[nearest] - pred_right[v]|, NaN when the hands do not touch.  One workgroup per sample; the left
// hand's vertices are staged in LDS (V x 12 bytes) and every lane scans them as broadcasts for its right-hand vertices.
namespace {
constexpr int CDEV_TPB = 256;
constexpr int CDEV_MAXV = 1024;

__global__ __launch_bounds__(CDEV_TPB) void cdev_kernel(const float* __restrict__ pred_l, const float* __restrict__ pred_r,
                                                        const float* __restrict__ gt_l, const float* __restrict__ gt_r, int V,
                                                        float contact, float* __restrict__ out) {
    __shared__ float L[CDEV_MAXV * 3];
    __shared__ float rs[CDEV_TPB];
    __shared__ int rc[CDEV_TPB];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* gl = gt_l + (long long)b * V * 3;
    for (int i = tid; i < V * 3; i += CDEV_TPB) L[i] = gl[i];
    __syncthreads();
    float sum = 0.f;
    int cnt = 0;
    for (int v = tid; v < V; v += CDEV_TPB) {
        const float* p = gt_r + ((long long)b * V + v) * 3;
        const float x = p[0], y = p[1], z = p[2];
        float best = INFINITY;
        int arg = 0;
        for (int j = 0; j < V; ++j) {
            const float dx = x - L[3 * j], dy = y - L[3 * j + 1], dz = z - L[3 * j + 2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { best = d2; arg = j; }                  // first minimum wins
        }
        if (!(sqrtf(best) > contact)) {
            const float* a = pred_l + ((long long)b * V + arg) * 3;
            const float* c = pred_r + ((long long)b * V + v) * 3;
            const float ex = a[0] - c[0], ey = a[1] - c[1], ez = a[2] - c[2];
            sum += sqrtf(ex * ex + ey * ey + ez * ez);
            ++cnt;
        }
    }
    rs[tid] = sum;
    rc[tid] = cnt;
    __syncthreads();
    for (int s = CDEV_TPB / 2; s > 0; s >>= 1) {
        if (tid < s) { rs[tid] += rs[tid + s]; rc[tid] += rc[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) out[b] = rs[0] / (float)rc[0];                     // 0 / 0 = NaN: no contact in this sample (nanmean)
}
}  // namespace

extern "C" int rih_cdev(const float* pred_left, const float* pred_right, const float* gt_left, const float* gt_right, int B,
                        int V, float contact, float* out, void* stream) {
    if (!pred_left || !pred_right || !gt_left || !gt_right || !out || B < 1 || V < 1 || V > CDEV_MAXV) return RIH_EINVAL;
    hipLaunchKernelGGL(cdev_kernel, dim3(B), dim3(CDEV_TPB), 0, (hipStream_t)stream, pred_left, pred_right, gt_left, gt_right, V,
                       contact, out);
    return (int)hipGetLastError();
}
