// rih_sdf.hip -- voxelised interior distance field of a closed triangle mesh: the reference's only native kernel
// (pose_data_optimize/sdf/sdf/csrc/sdf_cuda_kernel.cu:242-308, used by sdf/sdf_loss.py for inter-hand penetration), SURVEY 8f
// rank 4.  phi[b][k][j][i] = distance from the centre of voxel (i, j, k) of a G^3 grid over [-1, 1]^3 to the nearest
// triangle if the centre is inside mesh b (odd number of crossings of the segment from the centre towards the corner
// (-1,-1,-1), Moeller-Trumbore), else 0.
//
// The reference runs one thread per voxel, each walking all F faces and gathering 3 x 3 floats per face through the index
// list from global memory.  Here a workgroup of 256 voxels stages the faces' nine coordinates in LDS once per tile of 128
// faces (gathered cooperatively, one face per lane) and every lane then reads them as LDS broadcasts: F x 36 bytes per
// workgroup from L2 instead of per thread.  The per-face arithmetic follows the reference's order of operations so that the
// CPU restatement (oracle/sdf_oracle.py) agrees to rounding.  Compute-bound on fp32 VALU (~150 flops per voxel-face pair).
// STATUS: harness-verified against the oracle; not yet run on a GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"

namespace {

constexpr int TPB = 256;
constexpr int FT = 128;          // faces per LDS tile

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float dist3(const float* a, const float* b) {
    float l = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = a[i] - b[i]; l += d * d; }
    return sqrtf(l);
}

// sdf_cuda_kernel.cu:69-88
__device__ __forceinline__ float point_segment(const float* x0, const float* x1, const float* x2, float* r) {
    const float dx[3] = {x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]};
    const float m2 = dot3(dx, dx);
    float s12 = (dot3(x2, dx) - dot3(x0, dx)) / m2;
    s12 = s12 < 0.f ? 0.f : (s12 > 1.f ? 1.f : s12);
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = s12 * x1[i] + (1.f - s12) * x2[i];
    return dist3(x0, r);
}

// sdf_cuda_kernel.cu:158-236: closest point of triangle x1 x2 x3 to x0
__device__ __forceinline__ void point_triangle(const float* x0, const float* x1, const float* x2, const float* x3, float* r) {
    float x13[3], x23[3], x03[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { x13[i] = x1[i] - x3[i]; x23[i] = x2[i] - x3[i]; x03[i] = x0[i] - x3[i]; }
    const float m13 = dot3(x13, x13), m23 = dot3(x23, x23), d = dot3(x13, x23);
    const float invdet = 1.f / fmaxf(m13 * m23 - d * d, 1e-30f);
    const float a = dot3(x13, x03), b = dot3(x23, x03);
    const float w23 = invdet * (m23 * a - d * b), w31 = invdet * (m13 * b - d * a), w12 = 1.f - w23 - w31;
    if (w23 >= 0.f && w31 >= 0.f && w12 >= 0.f) {
#pragma unroll
        for (int i = 0; i < 3; ++i) r[i] = w23 * x1[i] + w31 * x2[i] + w12 * x3[i];
        return;
    }
    float r1[3], r2[3], d1, d2;
    if (w23 > 0.f) { d1 = point_segment(x0, x1, x2, r1); d2 = point_segment(x0, x1, x3, r2); }
    else if (w31 > 0.f) { d1 = point_segment(x0, x1, x2, r1); d2 = point_segment(x0, x2, x3, r2); }
    else { d1 = point_segment(x0, x1, x3, r1); d2 = point_segment(x0, x2, x3, r2); }
    const bool first = d1 < d2;
#pragma unroll
    for (int i = 0; i < 3; ++i) r[i] = first ? r1[i] : r2[i];
}

// sdf_cuda_kernel.cu:91-153 (Moeller-Trumbore): does the ray orig + t dir hit the triangle, and at which t
__device__ __forceinline__ bool ray_triangle(const float* orig, const float* dir, const float* v0, const float* v1,
                                             const float* v2, float* t) {
    float e1[3], e2[3], tv[3], pv[3], qv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { e1[i] = v1[i] - v0[i]; e2[i] = v2[i] - v0[i]; }
    pv[0] = dir[1] * e2[2] - dir[2] * e2[1]; pv[1] = dir[2] * e2[0] - dir[0] * e2[2]; pv[2] = dir[0] * e2[1] - dir[1] * e2[0];
    const float det = dot3(e1, pv);
    if (det > -0.000001f && det < 0.000001f) return false;
    const float inv = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 3; ++i) tv[i] = orig[i] - v0[i];
    const float u = dot3(tv, pv) * inv;
    if (u < 0.f || u > 1.f) return false;
    qv[0] = tv[1] * e1[2] - tv[2] * e1[1]; qv[1] = tv[2] * e1[0] - tv[0] * e1[2]; qv[2] = tv[0] * e1[1] - tv[1] * e1[0];
    const float v = dot3(dir, qv) * inv;
    if (v < 0.f || u + v > 1.f) return false;
    *t = dot3(e2, qv) * inv;
    return true;
}

__global__ __launch_bounds__(TPB) void sdf_kernel(float* __restrict__ phi, const int32_t* __restrict__ faces,
                                                  const float* __restrict__ vertices, int F, int V, int G) {
    __shared__ float tri[FT][9];
    const int vox = G * G * G;
    const int b = blockIdx.y;
    const int tid = blockIdx.x * TPB + threadIdx.x;
    const bool live = tid < vox;
    const int i = tid % G, j = (tid / G) % G, k = tid / (G * G);
    const float dx = 2.f / (float)(G - 1);
    const float c[3] = {-1.f + ((float)i + 0.5f) * dx, -1.f + ((float)j + 0.5f) * dx, -1.f + ((float)k + 0.5f) * dx};
    const float dir[3] = {-1.f - c[0], -1.f - c[1], -1.f - c[2]};
    const float* vb = vertices + (long long)b * V * 3;
    int hits = 0;
    float best = 1000.f;
    for (int f0 = 0; f0 < F; f0 += FT) {
        __syncthreads();
        if (threadIdx.x < FT && f0 + (int)threadIdx.x < F) {
            const int32_t* fc = faces + 3 * (f0 + threadIdx.x);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float* vp = vb + 3 * fc[p];
                tri[threadIdx.x][3 * p] = vp[0];
                tri[threadIdx.x][3 * p + 1] = vp[1];
                tri[threadIdx.x][3 * p + 2] = vp[2];
            }
        }
        __syncthreads();
        const int n = min(FT, F - f0);
        if (live)
            for (int f = 0; f < n; ++f) {
                const float* t = tri[f];
                float r[3], tt;
                point_triangle(c, t, t + 3, t + 6, r);
                best = fminf(best, dist3(c, r));
                if (ray_triangle(c, dir, t, t + 3, t + 6, &tt) && tt >= 0.f) ++hits;
            }
    }
    if (live) phi[(long long)b * vox + tid] = (hits & 1) ? best : 0.f;
}

}  // namespace

extern "C" int rih_sdf(float* phi, const int32_t* faces, const float* vertices, int B, int F, int V, int G, void* stream) {
    if (!phi || !faces || !vertices || B < 1 || B > 65535 || F < 1 || V < 3 || G < 2 || G > 1024) return RIH_EINVAL;
    const long long vox = (long long)G * G * G;
    hipLaunchKernelGGL(sdf_kernel, dim3((unsigned)((vox + TPB - 1) / TPB), B), dim3(TPB), 0, (hipStream_t)stream, phi, faces,
                       vertices, F, V, G);
    return (int)hipGetLastError();
}
