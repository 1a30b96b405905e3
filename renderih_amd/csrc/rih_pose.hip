// rih_pose.hip -- the MANO parameter head of the reference's `load_new_model` network for gfx950
// (common/myhand/decoder_lijun_mano.py:112-160, 247-300): Hardswish / 3*tanh activations of the ParamRegressor MLP,
// rot6d -> rotation matrix -> axis-angle of the 16 joints, axis-angle -> root rotation matrix, and the root-centred,
// bone-length-normalised MANO mesh.  All tensors here are tiny (B x 16 joints, B x 778 vertices): one thread per joint
// resp. one workgroup per mesh; every kernel is latency-bound.  The rotation maths lives in rih_pose_math.h (generic
// scalar: float for the forward, dual numbers for the exact vector-Jacobian products of the backward).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"
#include "rih_pose_math.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

inline int blocks_for(long long n) {
    long long b = (n + TPB - 1) / TPB;
    return (int)(b < 1 ? 1 : (b > 65535 ? 65535 : b));
}

__global__ void hardswish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB)
        y[i] = rih_hardswish(x[i]);
}
__global__ void hardswish_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                     long long n) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB)
        dx[i] = dy[i] * rih_hardswish_grad(x[i]);
}
__global__ void tanh_scale_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float scale) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB)
        y[i] = scale * tanhf(x[i]);
}
__global__ void tanh_scale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                      long long n, float scale) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        const float t = y[i] / scale;
        dx[i] = dy[i] * scale * (1.f - t * t);
    }
}

__global__ void rot6d_fwd_kernel(const float* __restrict__ x, float* __restrict__ R, float* __restrict__ aa, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float xi[6], Ri[9], ai[3];
    for (int k = 0; k < 6; ++k) xi[k] = x[6 * i + k];
    rih_rot6d_to_rotmat_aa<float>(xi, Ri, ai);
    for (int k = 0; k < 9; ++k) R[9 * i + k] = Ri[k];
    for (int k = 0; k < 3; ++k) aa[3 * i + k] = ai[k];
}
__global__ void rot6d_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dR, const float* __restrict__ daa,
                                 float* __restrict__ dx, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float xi[6], g[6], dRi[9], dai[3];
    for (int k = 0; k < 6; ++k) xi[k] = x[6 * i + k];
    if (dR != nullptr)
        for (int k = 0; k < 9; ++k) dRi[k] = dR[9 * i + k];
    if (daa != nullptr)
        for (int k = 0; k < 3; ++k) dai[k] = daa[3 * i + k];
    rih_rot6d_vjp(xi, dR != nullptr ? dRi : nullptr, daa != nullptr ? dai : nullptr, g);
    for (int k = 0; k < 6; ++k) dx[6 * i + k] = g[k];
}
__global__ void rodrigues_fwd_kernel(const float* __restrict__ a, float* __restrict__ R, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float ai[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]}, Ri[9];
    rih_rodrigues<float>(ai, Ri);
    for (int k = 0; k < 9; ++k) R[9 * i + k] = Ri[k];
}
__global__ void rodrigues_bwd_kernel(const float* __restrict__ a, const float* __restrict__ dR, float* __restrict__ da, int n) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    float ai[3] = {a[3 * i], a[3 * i + 1], a[3 * i + 2]}, dRi[9], g[3];
    for (int k = 0; k < 9; ++k) dRi[k] = dR[9 * i + k];
    rih_rodrigues_vjp(ai, dRi, g);
    for (int k = 0; k < 3; ++k) da[3 * i + k] = g[k];
}

// out_v = (v - j[root]) * s,  s = target / |j[a] - j[b]|      (decoder_lijun_mano.py:262-267)
__global__ __launch_bounds__(TPB) void center_scale_fwd_kernel(const float* __restrict__ v, const float* __restrict__ j,
                                                               int V, int NJ, int root, int ja, int jb, float target,
                                                               float* __restrict__ vout, float* __restrict__ sout) {
    const int b = blockIdx.x;
    const float* jj = j + (long long)b * NJ * 3;
    float r[3], l2 = 0.f;
    for (int k = 0; k < 3; ++k) {
        r[k] = jj[3 * root + k];
        const float d = jj[3 * ja + k] - jj[3 * jb + k];
        l2 += d * d;
    }
    const float s = target / sqrtf(l2);
    if (threadIdx.x == 0) sout[b] = s;
    for (int i = threadIdx.x; i < V * 3; i += TPB)
        vout[(long long)b * V * 3 + i] = (v[(long long)b * V * 3 + i] - r[i % 3]) * s;
}
// dv = s G;  dj[root] -= s sum G;  ds = sum G.(v - r) + gs;  dL = -s/L ds;  dj[a] += dL (ja - jb)/L, dj[b] -= the same
__global__ __launch_bounds__(TPB) void center_scale_bwd_kernel(const float* __restrict__ v, const float* __restrict__ j,
                                                               const float* __restrict__ G, const float* __restrict__ gs,
                                                               int V, int NJ, int root, int ja, int jb, float target,
                                                               float* __restrict__ dv, float* __restrict__ dj) {
    __shared__ float red[TPB / 64][4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* jj = j + (long long)b * NJ * 3;
    float r[3], d[3], l2 = 0.f;
    for (int k = 0; k < 3; ++k) {
        r[k] = jj[3 * root + k];
        d[k] = jj[3 * ja + k] - jj[3 * jb + k];
        l2 += d[k] * d[k];
    }
    const float L = sqrtf(l2), s = target / L;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};        // sum G (3), sum G.(v - r)
    for (int i = threadIdx.x; i < V * 3; i += TPB) {
        const float g = G[(long long)b * V * 3 + i];
        dv[(long long)b * V * 3 + i] = s * g;
        acc[i % 3] += g;
        acc[3] += g * (v[(long long)b * V * 3 + i] - r[i % 3]);
    }
    for (int k = 0; k < 4; ++k) {
        const float t = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = t;
    }
    for (int i = threadIdx.x; i < NJ * 3; i += TPB) dj[(long long)b * NJ * 3 + i] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4];
        for (int k = 0; k < 4; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        float* o = dj + (long long)b * NJ * 3;
        const float ds = t[3] + (gs != nullptr ? gs[b] : 0.f);
        const float dL = -s / L * ds;
        for (int k = 0; k < 3; ++k) {
            o[3 * root + k] -= s * t[k];
            o[3 * ja + k] += dL * d[k] / L;
            o[3 * jb + k] -= dL * d[k] / L;
        }
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)
#define LAUNCH_RET() return (int)hipGetLastError()

extern "C" int rih_hardswish_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(hardswish_fwd_kernel, dim3(blocks_for(n)), dim3(TPB), 0, STREAM, x, y, (long long)n);
    LAUNCH_RET();
}
extern "C" int rih_hardswish_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(hardswish_bwd_kernel, dim3(blocks_for(n)), dim3(TPB), 0, STREAM, dy, x, dx, (long long)n);
    LAUNCH_RET();
}
extern "C" int rih_tanh_scale_fwd(const float* x, float* y, int64_t n, float scale, void* stream) {
    if (!x || !y || n < 1 || scale == 0.f) return RIH_EINVAL;
    hipLaunchKernelGGL(tanh_scale_fwd_kernel, dim3(blocks_for(n)), dim3(TPB), 0, STREAM, x, y, (long long)n, scale);
    LAUNCH_RET();
}
extern "C" int rih_tanh_scale_bwd(const float* dy, const float* y, float* dx, int64_t n, float scale, void* stream) {
    if (!dy || !y || !dx || n < 1 || scale == 0.f) return RIH_EINVAL;
    hipLaunchKernelGGL(tanh_scale_bwd_kernel, dim3(blocks_for(n)), dim3(TPB), 0, STREAM, dy, y, dx, (long long)n, scale);
    LAUNCH_RET();
}
extern "C" int rih_rot6d_fwd(const float* x, float* R, float* aa, int n, void* stream) {
    if (!x || !R || !aa || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(rot6d_fwd_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, STREAM, x, R, aa, n);
    LAUNCH_RET();
}
extern "C" int rih_rot6d_bwd(const float* x, const float* dR, const float* daa, float* dx, int n, void* stream) {
    if (!x || !dx || (!dR && !daa) || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(rot6d_bwd_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, STREAM, x, dR, daa, dx, n);
    LAUNCH_RET();
}
extern "C" int rih_rodrigues_fwd(const float* a, float* R, int n, void* stream) {
    if (!a || !R || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(rodrigues_fwd_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, STREAM, a, R, n);
    LAUNCH_RET();
}
extern "C" int rih_rodrigues_bwd(const float* a, const float* dR, float* da, int n, void* stream) {
    if (!a || !dR || !da || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(rodrigues_bwd_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, STREAM, a, dR, da, n);
    LAUNCH_RET();
}
extern "C" int rih_center_scale_fwd(const float* v, const float* j, int B, int V, int NJ, int root, int ja, int jb,
                                    float target, float* vout, float* sout, void* stream) {
    if (!v || !j || !vout || !sout || B < 1 || V < 1 || NJ < 1) return RIH_EINVAL;
    if (root < 0 || root >= NJ || ja < 0 || ja >= NJ || jb < 0 || jb >= NJ) return RIH_EINVAL;
    hipLaunchKernelGGL(center_scale_fwd_kernel, dim3(B), dim3(TPB), 0, STREAM, v, j, V, NJ, root, ja, jb, target, vout, sout);
    LAUNCH_RET();
}
extern "C" int rih_center_scale_bwd(const float* v, const float* j, const float* dvout, const float* dsout, int B, int V,
                                    int NJ, int root, int ja, int jb, float target, float* dv, float* dj, void* stream) {
    if (!v || !j || !dvout || !dv || !dj || B < 1 || V < 1 || NJ < 1) return RIH_EINVAL;
    if (root < 0 || root >= NJ || ja < 0 || ja >= NJ || jb < 0 || jb >= NJ) return RIH_EINVAL;
    hipLaunchKernelGGL(center_scale_bwd_kernel, dim3(B), dim3(TPB), 0, STREAM, v, j, dvout, dsout, V, NJ, root, ja, jb, target,
                       dv, dj);
    LAUNCH_RET();
}
