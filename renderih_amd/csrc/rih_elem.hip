// rih_elem.hip -- HBM-bound kernels of the RenderIH pose network for gfx950: layout changes, pooling,
// bilinear resampling, BatchNorm (training statistics, apply, backward), LayerNorm, softmax(+dropout),
// residual/dropout adds, column sums, row gathers and the Chebyshev graph feature build.
//
// Everything is NHWC / row-major with the channel (feature) dimension fastest, so a wavefront's 64 lanes
// touch consecutive addresses; float4 (16 B/lane) accesses wherever the channel count allows.  Row
// reductions (LayerNorm, softmax) use one wavefront per row and DPP/xor shuffles over 64 lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/renderih_amd.h"
#include "rih_hash.h"

namespace {

constexpr int TPB = 256;

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline int grid_for(long long n, int per_thread = 1) {
    long long b = (n + (long long)TPB * per_thread - 1) / ((long long)TPB * per_thread);
    if (b < 1) b = 1;
    if (b > 16384) b = 16384;
    return (int)b;
}

#define GRID_STRIDE(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based RNG for dropout: 32 uniform bits from (seed, element index) -- rih_hash.h
__device__ __forceinline__ uint32_t drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0.0) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

// ------------------------------------------------------------------------------------------------ operand bounds (rih_gemm engine 2)
// A bound block (include/renderih_amd.h: RIH_BOUND_FLOATS) = 64 partial maxima, one per 128-byte line; the bound is the maximum
// of the 64.  Non-negative floats order like their bit patterns, so a merge is an integer atomicMax (NaN candidates are dropped
// by fmaxf before they get there).  Why 64 lines: same-address device-scope atomics retire at ~7-10 ns each on this chip (they
// are served behind the XCDs' L2s).  One word for the whole tensor cost the BatchNorm kernels dearly -- one atomic per wavefront
// (65536 per launch): 6.5 -> 60 ms per step; one per block with a read in front of it: still +27 us on a 30 us launch
// (profiles/r04/ab/train_e2.log, step_trace_c2_*.txt) -- because the blocks of a streaming kernel finish in herds that all
// read the same stale word.  Spread over 64 lines (block b -> line b & 63) the herd is 64 short queues that drain in parallel.
// Every thread of the block must call it (one barrier).
constexpr int BOUND_SLOTS = 64, BOUND_STRIDE = 32;
static_assert(BOUND_SLOTS * BOUND_STRIDE == RIH_BOUND_FLOATS, "bound block layout");
__device__ __forceinline__ void amax_publish(float* out, float v) {
    __shared__ float amax_red[TPB / 64];
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = amax_red[0];
#pragma unroll
        for (int w = 1; w < TPB / 64; ++w) m = fmaxf(m, amax_red[w]);
        const unsigned bits = __float_as_uint(m);
        unsigned* o = reinterpret_cast<unsigned*>(out) + ((blockIdx.x + blockIdx.y * 7u) & (BOUND_SLOTS - 1)) * BOUND_STRIDE;
        if (m > 0.f && bits > __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(o, bits);
    }
}
__device__ __forceinline__ float absmax_span(const float* __restrict__ x, long long n, long long first, long long step) {
    float m = 0.f;
    const long long nq = n >> 2;
    if (((uintptr_t)x & 15) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (long long i = first; i < nq; i += step) {
            const float4 v = x4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (long long i = 4 * nq + first; i < n; i += step) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (long long i = first; i < n; i += step) m = fmaxf(m, fabsf(x[i]));
    }
    return m;
}
__global__ __launch_bounds__(TPB) void absmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    amax_publish(out, absmax_span(x, n, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x));
}
// many tensors in one launch (rih_absmax_multi: the convolution weights of a training step); descriptors by value
constexpr int ABSMAX_PACK = 120;
struct AbsmaxPack {
    rih_absmax_desc d[ABSMAX_PACK];
    int first[ABSMAX_PACK + 1];
    int n;
};
static_assert(sizeof(AbsmaxPack) <= 4096, "kernel argument limit");
__global__ __launch_bounds__(TPB) void absmax_multi_kernel(const AbsmaxPack pk) {
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < pk.n && b >= pk.first[k + 1]) ++k;
    const int nb = pk.first[k + 1] - pk.first[k];
    const float m = absmax_span(pk.d[k].x, pk.d[k].n, (long long)(b - pk.first[k]) * TPB + threadIdx.x, (long long)nb * TPB);
    amax_publish(pk.d[k].out, m);
}

// ------------------------------------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                    int Cpad) {
    const long long total = (long long)N * H * W * Cpad;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cpad);
        const long long pix = i / Cpad;
        const int hw = (int)(pix % ((long long)H * W));
        const int n = (int)(pix / ((long long)H * W));
        y[i] = (c < C) ? x[((long long)n * C + c) * H * W + hw] : 0.f;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                    int ldx) {
    const long long total = (long long)N * C * H * W;
    GRID_STRIDE(i, total) {
        const int hw = (int)(i % ((long long)H * W));
        const long long t = i / ((long long)H * W);
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        y[i] = x[((long long)n * H * W + hw) * ldx + c];
    }
}

// ------------------------------------------------------------------------------------------------ pooling
// Vector width of the pooling / resampling kernels: V = 4 (16-byte accesses, channel quads) whenever C % 4 == 0, else scalar.
template <int V> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<4> { using type = float4; };

template <int V>
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int8_t* __restrict__ arg, int N,
                                   int H, int W, int C, int Ho, int Wo) {
    using T = typename VecT<V>::type;
    const int CV = C / V;
    const long long total = (long long)N * Ho * Wo * CV;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[V];
        int bi[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if (hi < 0 || hi >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if (wi < 0 || wi >= W) continue;
                const T v = *reinterpret_cast<const T*>(x + (((long long)n * H + hi) * W + wi) * C + c);
                const float* pv = reinterpret_cast<const float*>(&v);
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (pv[e] > best[e] || pv[e] != pv[e]) { best[e] = pv[e]; bi[e] = kh * 3 + kw; }
            }
        }
        if constexpr (V == 4) {
            reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
            reinterpret_cast<unsigned*>(arg)[i] = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        } else {
            y[i] = best[0];
            arg[i] = (int8_t)bi[0];
        }
    }
}
template <int V>
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int8_t* __restrict__ arg,
                                   float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    using T = typename VecT<V>::type;
    const int CV = C / V;
    const long long total = (long long)N * H * W * CV;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int wi = (int)(t % W);
        t /= W;
        const int hi = (int)(t % H);
        const int n = (int)(t / H);
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.f;
        for (int ho = hi / 2; ho <= (hi + 1) / 2; ++ho) {
            if (ho >= Ho) continue;
            const int kh = hi + 1 - 2 * ho;
            if (kh < 0 || kh > 2) continue;
            for (int wo = wi / 2; wo <= (wi + 1) / 2; ++wo) {
                if (wo >= Wo) continue;
                const int kw = wi + 1 - 2 * wo;
                if (kw < 0 || kw > 2) continue;
                const long long o = (((long long)n * Ho + ho) * Wo + wo) * C + c;
                const T d = *reinterpret_cast<const T*>(dy + o);
                const float* pd = reinterpret_cast<const float*>(&d);
                unsigned a;
                if constexpr (V == 4) a = *reinterpret_cast<const unsigned*>(arg + o);
                else a = (unsigned)(unsigned char)arg[o];
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if ((int)((a >> (8 * e)) & 0xffu) == kh * 3 + kw) s[e] += pd[e];
            }
        }
        if constexpr (V == 4) reinterpret_cast<float4*>(dx)[i] = make_float4(s[0], s[1], s[2], s[3]);
        else dx[i] = s[0];
    }
}

__global__ void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const long long total = (long long)N * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int n = (int)(i / C);
        float s = 0.f;
        for (int p = 0; p < HW; ++p) s += x[((long long)n * HW + p) * C + c];
        y[i] = s / (float)HW;
    }
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int HW, int C) {
    const long long total = (long long)N * HW * C;
    const float inv = 1.f / (float)HW;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long long)HW * C));
        dx[i] = dy[(long long)n * C + c] * inv;
    }
}

// bilinear xF (F = 2, 4, 8 ...), align_corners=True (nn.Upsample in models/encoder.py:51; F.interpolate in
// models/encoder.py:228-230 for the HRNet heads)
__device__ __forceinline__ void bil_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = src - (float)i0;
}

template <int V>
__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                      int F) {
    using T = typename VecT<V>::type;
    const int Ho = F * H, Wo = F * W, CV = C / V;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)N * Ho * Wo * CV;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        int h0, h1, w0, w1;
        float lh, lw;
        bil_src(ho, sh, H, h0, h1, lh);
        bil_src(wo, sw, W, w0, w1, lw);
        const float* b = x + (long long)n * H * W * C + c;
        const T x00 = *reinterpret_cast<const T*>(b + ((long long)h0 * W + w0) * C);
        const T x01 = *reinterpret_cast<const T*>(b + ((long long)h0 * W + w1) * C);
        const T x10 = *reinterpret_cast<const T*>(b + ((long long)h1 * W + w0) * C);
        const T x11 = *reinterpret_cast<const T*>(b + ((long long)h1 * W + w1) * C);
        const float hl0 = 1.f - lh, wl0 = 1.f - lw;
        T o;
#pragma unroll
        for (int e = 0; e < V; ++e)
            reinterpret_cast<float*>(&o)[e] =
                hl0 * (wl0 * reinterpret_cast<const float*>(&x00)[e] + lw * reinterpret_cast<const float*>(&x01)[e]) +
                lh * (wl0 * reinterpret_cast<const float*>(&x10)[e] + lw * reinterpret_cast<const float*>(&x11)[e]);
        reinterpret_cast<T*>(y)[i] = o;
    }
}
template <int V>
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W,
                                      int C, int F) {
    using T = typename VecT<V>::type;
    const int Ho = F * H, Wo = F * W, CV = C / V;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)N * H * W * CV;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int wi = (int)(t % W);
        t /= W;
        const int hi = (int)(t % H);
        const int n = (int)(t / H);
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.f;
        for (int ho = max(0, F * hi - F - 1); ho <= min(Ho - 1, F * hi + 2 * F); ++ho) {
            int h0, h1;
            float lh;
            bil_src(ho, sh, H, h0, h1, lh);
            const float wh = ((h0 == hi) ? (1.f - lh) : 0.f) + ((h1 == hi) ? lh : 0.f);
            if (wh == 0.f) continue;
            for (int wo = max(0, F * wi - F - 1); wo <= min(Wo - 1, F * wi + 2 * F); ++wo) {
                int w0, w1;
                float lw;
                bil_src(wo, sw, W, w0, w1, lw);
                const float ww = ((w0 == wi) ? (1.f - lw) : 0.f) + ((w1 == wi) ? lw : 0.f);
                if (ww == 0.f) continue;
                const T d = *reinterpret_cast<const T*>(dy + (((long long)n * Ho + ho) * Wo + wo) * C + c);
#pragma unroll
                for (int e = 0; e < V; ++e) s[e] += wh * ww * reinterpret_cast<const float*>(&d)[e];
            }
        }
        if constexpr (V == 4) reinterpret_cast<float4*>(dx)[i] = make_float4(s[0], s[1], s[2], s[3]);
        else dx[i] = s[0];
    }
}

// nearest xF upsample fused with the accumulation of HighResolutionModule's fuse sum (model_zoo/hrnet.py:181-183,
// 229-236): y[n, ho, wo, :] = add[n, ho, wo, :] + x[n, ho/F, wo/F, :]
__global__ void nearest_up_add_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ y,
                                      int N, int H, int W, int C4, int F) {
    const int Ho = F * H, Wo = F * W;
    const long long total = (long long)N * Ho * Wo * C4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* a4 = reinterpret_cast<const float4*>(add);
    float4* y4 = reinterpret_cast<float4*>(y);
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float4 v = x4[(((long long)n * H + ho / F) * W + wo / F) * C4 + c];
        if (add != nullptr) {
            const float4 a = a4[i];
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        y4[i] = v;
    }
}
// dx[n, h, w, :] = sum of dy over the FxF block
__global__ void nearest_up_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C4,
                                      int F) {
    const int Wo = F * W;
    const long long total = (long long)N * H * W * C4;
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    float4* x4 = reinterpret_cast<float4*>(dx);
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int a = 0; a < F; ++a)
            for (int b = 0; b < F; ++b) {
                const float4 v = d4[(((long long)n * F * H + (F * h + a)) * Wo + (F * w + b)) * C4 + c];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        x4[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------ column reductions
// Shared helper for per-channel reductions over rows of an [rows][C] matrix (C % 4 == 0):
// block = 256 threads = cgb channel-quads x rt row-threads; grid = (ceil(C/4/cgb), nchunk).
struct ColGeom {
    int cgb, rt, gx, nchunk, rows_per_chunk;
};
inline ColGeom col_geom(int rows, int C) {
    ColGeom g;
    const int CG = C / 4;
    g.cgb = CG < 64 ? CG : 64;
    int p = 1;
    while (p * 2 <= g.cgb) p *= 2;       // power of two <= cgb so that 256 % cgb == 0
    g.cgb = p;
    g.rt = TPB / g.cgb;
    g.gx = (CG + g.cgb - 1) / g.cgb;
    int want = 1024 / g.gx;
    if (want < 1) want = 1;
    int maxc = (rows + g.rt * 4 - 1) / (g.rt * 4);
    if (maxc < 1) maxc = 1;
    g.nchunk = want < maxc ? want : maxc;
    if (g.nchunk > 256) g.nchunk = 256;
    g.rows_per_chunk = (rows + g.nchunk - 1) / g.nchunk;
    g.nchunk = (rows + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return g;
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// reduce `v1`,`v2` (per-thread float4 partials) across the rt row-threads of a block; thread ry==0 returns the sum
template <int NV>
__device__ __forceinline__ void block_col_reduce(float4 (&v)[NV], int cg, int ry, int cgb, int rt) {
    __shared__ float4 red[TPB];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        red[ry * cgb + cg] = v[k];
        __syncthreads();
        for (int s = rt >> 1; s > 0; s >>= 1) {
            if (ry < s) red[ry * cgb + cg] = f4add(red[ry * cgb + cg], red[(ry + s) * cgb + cg]);
            __syncthreads();
        }
        v[k] = red[cg];
        __syncthreads();
    }
}

// BN training statistics, pass 1: shifted sums  sum(x - x0), sum((x - x0)^2) per chunk, x0 = first row
__device__ __forceinline__ void bn_stats_partial_body(const float* __restrict__ x, int rows, int C, int cgb,
                                                      int rt, int rows_per_chunk, float* __restrict__ ws) {
    const int cg = threadIdx.x % cgb, ry = threadIdx.x / cgb;
    const int g = blockIdx.x * cgb + cg;
    const int CG = C / 4;
    const int chunk = blockIdx.y, nchunk = gridDim.y;
    float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    if (g < CG) {
        const float4 sh = *reinterpret_cast<const float4*>(x + g * 4);
        const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
        auto add = [&](const float4 v) {
            const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
            acc[0].x += dx; acc[0].y += dy; acc[0].z += dz; acc[0].w += dw;
            acc[1].x += dx * dx; acc[1].y += dy * dy; acc[1].z += dz * dz; acc[1].w += dw * dw;
        };
        const float* px = x + g * 4;
        int r = r0 + ry;
        // four independent 16-byte loads in flight per lane (one load per iteration leaves HBM at ~2 TB/s)
        for (; r + 3 * rt < r1; r += 4 * rt) {
            const float4 v0 = *reinterpret_cast<const float4*>(px + (long long)r * C);
            const float4 v1 = *reinterpret_cast<const float4*>(px + (long long)(r + rt) * C);
            const float4 v2 = *reinterpret_cast<const float4*>(px + (long long)(r + 2 * rt) * C);
            const float4 v3 = *reinterpret_cast<const float4*>(px + (long long)(r + 3 * rt) * C);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; r < r1; r += rt) add(*reinterpret_cast<const float4*>(px + (long long)r * C));
    }
    block_col_reduce<2>(acc, cg, ry, cgb, rt);
    if (ry == 0 && g < CG) {
        *reinterpret_cast<float4*>(ws + ((long long)0 * nchunk + chunk) * C + g * 4) = acc[0];
        *reinterpret_cast<float4*>(ws + ((long long)1 * nchunk + chunk) * C + g * 4) = acc[1];
    }
}
__global__ __launch_bounds__(TPB) void bn_stats_partial_kernel(const float* __restrict__ x, int rows, int C, int cgb,
                                                               int rt, int rows_per_chunk, float* __restrict__ ws) {
    bn_stats_partial_body(x, rows, C, cgb, rt, rows_per_chunk, ws);
}

// one wavefront per channel: lanes stride over the chunk partials, fp64 xor-shuffle reduction
__global__ __launch_bounds__(TPB) void bn_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ ws,
                                                             int rows, int C, int nchunk, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ rmean, float* __restrict__ rvar) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < nchunk; k += 64) {
        s1 += (double)ws[((long long)0 * nchunk + k) * C + c];
        s2 += (double)ws[((long long)1 * nchunk + k) * C + c];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    const double n = (double)rows;
    const double d = s1 / n;
    const double m = (double)x[c] + d;
    double var = s2 / n - d * d;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean != nullptr) {
        const double unb = (rows > 1) ? var * n / (n - 1.0) : var;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ res, float* __restrict__ y,
                                long long nquads, int C, int relu, unsigned char* __restrict__ mask,
                                float* __restrict__ amax) {
    const int CG = C / 4;
    float am = 0.f;         // max |y| of this thread (-> *amax: the operand bound of the GEMM that reads y, rih_gemm engine 2)
    GRID_STRIDE(i, nquads) {
        const int g = (int)(i % CG);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 m = reinterpret_cast<const float4*>(mean)[g];
        const float4 is = reinterpret_cast<const float4*>(invstd)[g];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[g];
        const float4 be = reinterpret_cast<const float4*>(beta)[g];
        float4 o;
        o.x = (v.x - m.x) * (is.x * ga.x) + be.x;
        o.y = (v.y - m.y) * (is.y * ga.y) + be.y;
        o.z = (v.z - m.z) * (is.z * ga.z) + be.z;
        o.w = (v.w - m.w) * (is.w * ga.w) + be.w;
        if (res != nullptr) {
            const float4 r = reinterpret_cast<const float4*>(res)[i];
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            // the backward needs only the sign pattern of the output: one byte per quad instead of re-reading y (16 bytes)
            if (mask != nullptr)
                mask[i] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax != nullptr) amax_publish(amax, am);
}

// Training statistics from per-row-block (mean, M2) pairs -- the statistics epilogue of rih_gemm (rih_gemm_desc.stats:
// part[T][2][C], block k covers rows [k*rpb, min(rows, (k+1)*rpb))).  One workgroup per channel; Chan's merge written as three
// double sums: n*mean = sum n_k mean_k, M2 = sum M2_k + sum n_k mean_k^2 - n mean^2.
// (A 16-channel x 16-lane workgroup that reads each 64-byte sector of partials once was tried in round 3: 41 us per launch
// against 10 us -- it has 16 x fewer threads in flight; one workgroup per channel stays.)
__global__ __launch_bounds__(TPB) void bn_blocks_final_kernel(const float* __restrict__ part, int T, int C, int rows, int rpb,
                                                              float eps, float momentum, float* __restrict__ mean,
                                                              float* __restrict__ invstd, float* __restrict__ rmean,
                                                              float* __restrict__ rvar) {
    __shared__ double red[3][TPB / 64];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int k = threadIdx.x; k < T; k += TPB) {
        const double nk = (double)((k == T - 1) ? rows - (T - 1) * rpb : rpb);
        const double mk = (double)part[((long long)k * 2 + 0) * C + c];
        s1 += nk * mk;
        s2 += (double)part[((long long)k * 2 + 1) * C + c];
        s3 += nk * mk * mk;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    s3 = wave_sum_d(s3);
    if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; red[2][wave] = s3; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = s2 = s3 = 0.0;
    for (int w = 0; w < TPB / 64; ++w) { s1 += red[0][w]; s2 += red[1][w]; s3 += red[2][w]; }
    const double n = (double)rows;
    const double m = s1 / n;
    double var = (s2 + s3 - n * m * m) / n;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean != nullptr) {
        const double unb = (rows > 1) ? var * n / (n - 1.0) : var;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

// BN backward pass 1: per-channel sum(dym), sum(dym * xhat), dym = dy * (y>0) when relu
__global__ __launch_bounds__(TPB) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, int rows, int C, int cgb,
                                                             int rt, int rows_per_chunk, int relu,
                                                             float* __restrict__ ws,
                                                             const unsigned char* __restrict__ mask) {
#include "rih_bn_bwd_partial.inc"
}

// finalize: out0[C] = sum dym, out1[C] = sum dym*xhat (also the dbeta / dgamma outputs); one wavefront per channel
__global__ __launch_bounds__(TPB) void two_sum_final_kernel(const float* __restrict__ ws, int C, int nchunk,
                                                            float* __restrict__ out0, float* __restrict__ out1) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < nchunk; k += 64) {
        s1 += (double)ws[((long long)0 * nchunk + k) * C + c];
        s2 += (double)ws[((long long)1 * nchunk + k) * C + c];
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane == 0) { out0[c] = (float)s1; out1[c] = (float)s2; }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ sum_dy, const float* __restrict__ sum_dyxh,
                                    float* __restrict__ dx, float* __restrict__ dres, long long nquads, int C,
                                    float inv_rows, int relu, int frozen, const unsigned char* __restrict__ mask,
                                    float* __restrict__ amax) {
    const int CG = C / 4;
    float am = 0.f;         // max |dx| of this thread (-> *amax, see bn_apply_kernel)
    GRID_STRIDE(i, nquads) {
        const int g = (int)(i % CG);
        float4 d = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            if (mask != nullptr) {
                const unsigned bits = mask[i];
                if (!(bits & 1u)) d.x = 0.f;
                if (!(bits & 2u)) d.y = 0.f;
                if (!(bits & 4u)) d.z = 0.f;
                if (!(bits & 8u)) d.w = 0.f;
            } else {
                const float4 yy = reinterpret_cast<const float4*>(y)[i];
                if (!(yy.x > 0.f)) d.x = 0.f;
                if (!(yy.y > 0.f)) d.y = 0.f;
                if (!(yy.z > 0.f)) d.z = 0.f;
                if (!(yy.w > 0.f)) d.w = 0.f;
            }
        }
        if (dres != nullptr) reinterpret_cast<float4*>(dres)[i] = d;
        const float4 is = reinterpret_cast<const float4*>(invstd)[g];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[g];
        // `frozen` bit 1: the input of this BatchNorm is a ReLU output (Conv -> ReLU -> BN of models/encoder.py:52-54): the
        // gradient leaves already gated by x > 0, and the convolution's backward skips its own ReLU pass over it
        const bool in_relu = (frozen & 2) != 0;
        float4 o;
        if (frozen & 1) {
            o.x = d.x * is.x * ga.x; o.y = d.y * is.y * ga.y; o.z = d.z * is.z * ga.z; o.w = d.w * is.w * ga.w;
            if (in_relu) {
                const float4 v = reinterpret_cast<const float4*>(x)[i];
                if (!(v.x > 0.f)) o.x = 0.f;
                if (!(v.y > 0.f)) o.y = 0.f;
                if (!(v.z > 0.f)) o.z = 0.f;
                if (!(v.w > 0.f)) o.w = 0.f;
            }
        } else {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const float4 m = reinterpret_cast<const float4*>(mean)[g];
            const float4 s1 = reinterpret_cast<const float4*>(sum_dy)[g];
            const float4 s2 = reinterpret_cast<const float4*>(sum_dyxh)[g];
            o.x = (d.x - s1.x * inv_rows - ((v.x - m.x) * is.x) * (s2.x * inv_rows)) * (is.x * ga.x);
            o.y = (d.y - s1.y * inv_rows - ((v.y - m.y) * is.y) * (s2.y * inv_rows)) * (is.y * ga.y);
            o.z = (d.z - s1.z * inv_rows - ((v.z - m.z) * is.z) * (s2.z * inv_rows)) * (is.z * ga.z);
            o.w = (d.w - s1.w * inv_rows - ((v.w - m.w) * is.w) * (s2.w * inv_rows)) * (is.w * ga.w);
            if (in_relu) {
                if (!(v.x > 0.f)) o.x = 0.f;
                if (!(v.y > 0.f)) o.y = 0.f;
                if (!(v.z > 0.f)) o.z = 0.f;
                if (!(v.w > 0.f)) o.w = 0.f;
            }
        }
        reinterpret_cast<float4*>(dx)[i] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax != nullptr) amax_publish(amax, am);
}

// generic column sum (bias gradients): any C, scalar path; block = 64 channels x 4 row-threads
__global__ __launch_bounds__(TPB) void colsum_partial_kernel(const float* __restrict__ x, int rows, int C, int ldx,
                                                             int rows_per_chunk, float* __restrict__ ws) {
    __shared__ float red[TPB];
    const int cl = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int chunk = blockIdx.y;
    float s = 0.f;
    if (c < C) {
        const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
        for (int r = r0 + ry; r < r1; r += 4) s += x[(long long)r * ldx + c];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (ry == 0 && c < C) ws[(long long)chunk * C + c] = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
}

__global__ __launch_bounds__(TPB) void colsum_final_kernel(const float* __restrict__ ws, int C, int nchunk,
                                                           float* __restrict__ out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    double s = 0.0;
    for (int k = lane; k < nchunk; k += 64) s += (double)ws[(long long)k * C + c];
    s = wave_sum_d(s);
    if (lane == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

inline int colsum_nchunk(int rows, int C) {
    const int gx = (C + 63) / 64;
    int want = 512 / gx;
    if (want < 1) want = 1;
    if (want > 128) want = 128;
    int maxc = (rows + 63) / 64;
    if (maxc < 1) maxc = 1;
    return want < maxc ? want : maxc;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
constexpr int LN_MAXPER = 16;   // D <= 1024

__global__ __launch_bounds__(TPB) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int rows, int D, float eps,
                                                            int relu, long long sG, long long sB) {
    {   // blockIdx.y = group (left / right hand): `rows` rows each, stacked activations, parameter sets sG / sB apart
        const long long go = (long long)blockIdx.y * rows;
        x += go * D; y += go * D; mean += go; rstd += go;
        if (x2 != nullptr) x2 += go * D;
        g += blockIdx.y * sG; b += blockIdx.y * sB;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nper = (D + 63) / 64;
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        float v[LN_MAXPER];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPER; ++i) {
            const int e = lane + 64 * i;
            float t = 0.f;
            if (i < nper && e < D) {
                t = x[(long long)r * D + e];
                if (x2 != nullptr) t += x2[(long long)r * D + e];
            }
            v[i] = t;
            s += t;
        }
        const float m = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPER; ++i) {
            const int e = lane + 64 * i;
            if (i < nper && e < D) { const float d = v[i] - m; q += d * d; }
        }
        const float var = wave_sum(q) / (float)D;
        const float rs = 1.f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXPER; ++i) {
            const int e = lane + 64 * i;
            if (i < nper && e < D) {
                float o = (v[i] - m) * rs * g[e] + b[e];
                if (relu) o = fmaxf(o, 0.f);
                y[(long long)r * D + e] = o;
            }
        }
        if (lane == 0) { mean[r] = m; rstd[r] = rs; }
    }
}

__global__ __launch_bounds__(TPB) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ x2, const float* __restrict__ y,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ dres, float* __restrict__ dx,
                                                            float* __restrict__ ws, int rows, int D, int relu,
                                                            long long sG) {
    {
        const long long go = (long long)blockIdx.y * rows;
        dy += go * D; x += go * D; dx += go * D; mean += go; rstd += go;
        if (x2 != nullptr) x2 += go * D;
        if (y != nullptr) y += go * D;
        if (dres != nullptr) dres += go * D;
        g += blockIdx.y * sG;
        ws += (long long)blockIdx.y * gridDim.x * 2 * D;
    }
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nper = (D + 63) / 64;
    float dgacc[LN_MAXPER], dbacc[LN_MAXPER];
#pragma unroll
    for (int i = 0; i < LN_MAXPER; ++i) { dgacc[i] = 0.f; dbacc[i] = 0.f; }
    for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
        const float m = mean[r], rs = rstd[r];
        float xh[LN_MAXPER], gd[LN_MAXPER];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXPER; ++i) {
            const int e = lane + 64 * i;
            xh[i] = 0.f;
            gd[i] = 0.f;
            if (i < nper && e < D) {
                float t = x[(long long)r * D + e];
                if (x2 != nullptr) t += x2[(long long)r * D + e];
                float d = dy[(long long)r * D + e];
                if (relu && !(y[(long long)r * D + e] > 0.f)) d = 0.f;
                const float h = (t - m) * rs;
                xh[i] = h;
                dgacc[i] += d * h;
                dbacc[i] += d;
                const float gdy = d * g[e];
                gd[i] = gdy;
                c1 += gdy;
                c2 += gdy * h;
            }
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXPER; ++i) {
            const int e = lane + 64 * i;
            if (i < nper && e < D) {
                float v = rs * (gd[i] - c1 - xh[i] * c2);
                if (dres != nullptr) v += dres[(long long)r * D + e];      // gradient of a skip connection around the norm
                dx[(long long)r * D + e] = v;
            }
        }
    }
    // reduce dg/db over the 4 waves of the block, write ws[blk][{dg,db}][D]
#pragma unroll
    for (int i = 0; i < LN_MAXPER; ++i) {
        if (i < nper) {
            const int e = lane + 64 * i;
            red[wave][lane] = dgacc[i];
            __syncthreads();
            if (wave == 0 && e < D)
                ws[((long long)blockIdx.x * 2 + 0) * D + e] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            __syncthreads();
            red[wave][lane] = dbacc[i];
            __syncthreads();
            if (wave == 0 && e < D)
                ws[((long long)blockIdx.x * 2 + 1) * D + e] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            __syncthreads();
        }
    }
}

// ---- 16-byte LayerNorm kernels for D = 64 / 128 / 256 / 512 / 1024 (every LayerNorm of the mesh decoder).
// LPR lanes own one row (float4 per lane, VPT of them): a wavefront normalises 64 / LPR rows at once, the row reductions are
// xor-shuffles inside the LPR-lane group.  Against the generic kernels above (one 4-byte element per lane and step, one row
// per wavefront): a quarter of the memory instructions and up to four rows in flight per wavefront -- these launches are
// latency-bound (8 .. 33 MB tensors), not HBM-bound.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float4 f4z() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int LPR, int VPT>
__global__ __launch_bounds__(TPB) void layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                                const float* __restrict__ g, const float* __restrict__ b,
                                                                float* __restrict__ y, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int rows, float eps, int relu,
                                                                long long sG, long long sB) {
    constexpr int D = LPR * 4 * VPT, RPW = 64 / LPR;
    {
        const long long go = (long long)blockIdx.y * rows;
        x += go * D; y += go * D; mean += go; rstd += go;
        if (x2 != nullptr) x2 += go * D;
        g += blockIdx.y * sG; b += blockIdx.y * sB;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    float4 gg[VPT], bb[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        gg[q] = *reinterpret_cast<const float4*>(g + (l + LPR * q) * 4);
        bb[q] = *reinterpret_cast<const float4*>(b + (l + LPR * q) * 4);
    }
    for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < rows; r0 += gridDim.x * 4 * RPW) {
        const int r = r0 + sub;
        const bool ok = r < rows;
        const long long ro = (long long)(ok ? r : 0) * D;
        float4 v[VPT];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int c = (l + LPR * q) * 4;
            float4 t = *reinterpret_cast<const float4*>(x + ro + c);
            if (x2 != nullptr) {
                const float4 u = *reinterpret_cast<const float4*>(x2 + ro + c);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[q] = t;
            s += (t.x + t.y) + (t.z + t.w);
        }
        const float m = group_sum<LPR>(s) / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const float a0 = v[q].x - m, a1 = v[q].y - m, a2 = v[q].z - m, a3 = v[q].w - m;
            qq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float var = group_sum<LPR>(qq) / (float)D;
        const float rs = 1.f / sqrtf(var + eps);
        if (ok) {
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                float4 o;
                o.x = (v[q].x - m) * rs * gg[q].x + bb[q].x;
                o.y = (v[q].y - m) * rs * gg[q].y + bb[q].y;
                o.z = (v[q].z - m) * rs * gg[q].z + bb[q].z;
                o.w = (v[q].w - m) * rs * gg[q].w + bb[q].w;
                if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                *reinterpret_cast<float4*>(y + ro + (l + LPR * q) * 4) = o;
            }
            if (l == 0) { mean[r] = m; rstd[r] = rs; }
        }
    }
}

template <int LPR, int VPT>
__global__ __launch_bounds__(TPB) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ x2, const float* __restrict__ y,
                                                                const float* __restrict__ g, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ dres, float* __restrict__ dx,
                                                                float* __restrict__ ws, int rows, int relu, long long sG) {
    constexpr int D = LPR * 4 * VPT, RPW = 64 / LPR;
    {
        const long long go = (long long)blockIdx.y * rows;
        dy += go * D; x += go * D; dx += go * D; mean += go; rstd += go;
        if (x2 != nullptr) x2 += go * D;
        if (y != nullptr) y += go * D;
        if (dres != nullptr) dres += go * D;
        g += blockIdx.y * sG;
        ws += (long long)blockIdx.y * gridDim.x * 2 * D;
    }
    __shared__ float4 red[4][LPR * VPT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, l = lane % LPR;
    float4 gg[VPT], dgacc[VPT], dbacc[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
        gg[q] = *reinterpret_cast<const float4*>(g + (l + LPR * q) * 4);
        dgacc[q] = f4z();
        dbacc[q] = f4z();
    }
    for (int r0 = (blockIdx.x * 4 + wave) * RPW; r0 < rows; r0 += gridDim.x * 4 * RPW) {
        const int r = r0 + sub;
        const bool ok = r < rows;
        const long long ro = (long long)(ok ? r : 0) * D;
        const float m = mean[ok ? r : 0], rs = rstd[ok ? r : 0];
        float4 h[VPT], gd[VPT];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            const int c = (l + LPR * q) * 4;
            float4 t = *reinterpret_cast<const float4*>(x + ro + c);
            if (x2 != nullptr) {
                const float4 u = *reinterpret_cast<const float4*>(x2 + ro + c);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            float4 d = *reinterpret_cast<const float4*>(dy + ro + c);
            if (relu) {
                const float4 yy = *reinterpret_cast<const float4*>(y + ro + c);
                if (!(yy.x > 0.f)) d.x = 0.f;
                if (!(yy.y > 0.f)) d.y = 0.f;
                if (!(yy.z > 0.f)) d.z = 0.f;
                if (!(yy.w > 0.f)) d.w = 0.f;
            }
            if (!ok) d = f4z();
            h[q] = make_float4((t.x - m) * rs, (t.y - m) * rs, (t.z - m) * rs, (t.w - m) * rs);
            dgacc[q].x += d.x * h[q].x; dgacc[q].y += d.y * h[q].y; dgacc[q].z += d.z * h[q].z; dgacc[q].w += d.w * h[q].w;
            dbacc[q].x += d.x; dbacc[q].y += d.y; dbacc[q].z += d.z; dbacc[q].w += d.w;
            gd[q] = make_float4(d.x * gg[q].x, d.y * gg[q].y, d.z * gg[q].z, d.w * gg[q].w);
            c1 += (gd[q].x + gd[q].y) + (gd[q].z + gd[q].w);
            c2 += (gd[q].x * h[q].x + gd[q].y * h[q].y) + (gd[q].z * h[q].z + gd[q].w * h[q].w);
        }
        c1 = group_sum<LPR>(c1) / (float)D;
        c2 = group_sum<LPR>(c2) / (float)D;
        if (ok) {
#pragma unroll
            for (int q = 0; q < VPT; ++q) {
                const int c = (l + LPR * q) * 4;
                float4 o;
                o.x = rs * (gd[q].x - c1 - h[q].x * c2);
                o.y = rs * (gd[q].y - c1 - h[q].y * c2);
                o.z = rs * (gd[q].z - c1 - h[q].z * c2);
                o.w = rs * (gd[q].w - c1 - h[q].w * c2);
                if (dres != nullptr) {
                    const float4 e = *reinterpret_cast<const float4*>(dres + ro + c);
                    o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
                }
                *reinterpret_cast<float4*>(dx + ro + c) = o;
            }
        }
    }
    // dg / db: over the 64 / LPR rows of the wavefront by shuffles, over the 4 wavefronts through LDS -> ws[blk][{dg,db}][D]
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int q = 0; q < VPT; ++q) {
            float4 a = pass == 0 ? dgacc[q] : dbacc[q];
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                a.x += __shfl_xor(a.x, o, 64); a.y += __shfl_xor(a.y, o, 64);
                a.z += __shfl_xor(a.z, o, 64); a.w += __shfl_xor(a.w, o, 64);
            }
            if (sub == 0) red[wave][l + LPR * q] = a;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < LPR * VPT; c += TPB) {
            const float4 a = red[0][c], b2 = red[1][c], c2 = red[2][c], d2 = red[3][c];
            *reinterpret_cast<float4*>(ws + ((long long)blockIdx.x * 2 + pass) * D + c * 4) =
                make_float4((a.x + b2.x) + (c2.x + d2.x), (a.y + b2.y) + (c2.y + d2.y), (a.z + b2.z) + (c2.z + d2.z),
                            (a.w + b2.w) + (c2.w + d2.w));
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(TPB) void ln_param_final_kernel(const float* __restrict__ ws, int D, int nblk,
                                                             float* __restrict__ dg, float* __restrict__ db) {
    ws += (long long)blockIdx.y * nblk * 2 * D;     // group: dg / db are [groups][D]
    dg += (long long)blockIdx.y * D;
    db += (long long)blockIdx.y * D;
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (e >= D) return;
    double a = 0.0, b = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        a += (double)ws[((long long)k * 2 + 0) * D + e];
        b += (double)ws[((long long)k * 2 + 1) * D + e];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) { dg[e] = (float)a; db[e] = (float)b; }
}

// ------------------------------------------------------------------------------------------------ softmax
constexpr int SM_MAXPER = 16;   // cols <= 1024

__global__ __launch_bounds__(TPB) void softmax_fwd_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                          float* __restrict__ Pd, long long rows, int cols, int ld,
                                                          float drop_p, uint64_t seed,
                                                          const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nper = (cols + 63) / 64;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        float v[SM_MAXPER];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < SM_MAXPER; ++i) {
            const int j = lane + 64 * i;
            v[i] = -INFINITY;
            if (i < nper && j < cols) { v[i] = S[r * ld + j]; mx = fmaxf(mx, v[i]); }
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < SM_MAXPER; ++i) {
            const int j = lane + 64 * i;
            if (i < nper && j < cols) { v[i] = expf(v[i] - mx); s += v[i]; }
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < SM_MAXPER; ++i) {
            const int j = lane + 64 * i;
            if (i < nper && j < cols) {
                const float pr = v[i] * inv;
                P[r * ld + j] = pr;
                if (drop_p > 0.f) {
                    const bool keep = rih_hash(seed, (uint64_t)(r * cols + j)) >= thr;
                    Pd[r * ld + j] = keep ? pr * keep_scale : 0.f;
                } else if (Pd != P) {
                    Pd[r * ld + j] = pr;
                }
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dPd,
                                                          long long rows, int cols, int ld, float drop_p, uint64_t seed,
                                                          const uint64_t* __restrict__ seed_dev,
                                                          float alpha) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nper = (cols + 63) / 64;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        float pv[SM_MAXPER], dp[SM_MAXPER];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < SM_MAXPER; ++i) {
            const int j = lane + 64 * i;
            pv[i] = 0.f;
            dp[i] = 0.f;
            if (i < nper && j < cols) {
                pv[i] = P[r * ld + j];
                float d = dPd[r * ld + j];
                if (drop_p > 0.f) {
                    const bool keep = rih_hash(seed, (uint64_t)(r * cols + j)) >= thr;
                    d = keep ? d * keep_scale : 0.f;
                }
                dp[i] = d;
                dot += d * pv[i];
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < SM_MAXPER; ++i) {
            const int j = lane + 64 * i;
            if (i < nper && j < cols) dPd[r * ld + j] = alpha * pv[i] * (dp[i] - dot);
        }
    }
}

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void add_dropout_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                   long long n, long long bmod, float drop_p, uint64_t seed,
                                   const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    GRID_STRIDE(i, n) {
        float bv = b[bmod > 0 ? (i % bmod) : i];
        if (drop_p > 0.f) bv = (rih_hash(seed, (uint64_t)i) >= thr) ? bv * keep_scale : 0.f;
        y[i] = (a != nullptr ? a[i] : 0.f) + bv;
    }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long n, float drop_p,
                                   uint64_t seed, const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    GRID_STRIDE(i, n) {
        float d = dy[i];
        if (drop_p > 0.f) d = (rih_hash(seed, (uint64_t)i) >= thr) ? d * keep_scale : 0.f;
        dx[i] = d;
    }
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                long long n) {
    GRID_STRIDE(i, n) dx[i] = (y[i] > 0.f) ? dy[i] : 0.f;
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ idx, float* __restrict__ y,
                                   int B, int Vin, int Vout, int D) {
    const long long total = (long long)B * Vout * D;
    GRID_STRIDE(i, total) {
        const int d = (int)(i % D);
        const long long t = i / D;
        const int v = (int)(t % Vout);
        const int b = (int)(t / Vout);
        y[i] = x[((long long)b * Vin + idx[v]) * D + d];
    }
}

__global__ void scatter_rows_add_kernel(const float* __restrict__ dy, const int32_t* __restrict__ inv_ptr,
                                        const int32_t* __restrict__ inv_idx, float* __restrict__ dx, int B, int Vin,
                                        int Vout, int D) {
    const long long total = (long long)B * Vin * D;
    GRID_STRIDE(i, total) {
        const int d = (int)(i % D);
        const long long t = i / D;
        const int v = (int)(t % Vin);
        const int b = (int)(t / Vin);
        float s = 0.f;
        for (int k = inv_ptr[v]; k < inv_ptr[v + 1]; ++k) s += dy[((long long)b * Vout + inv_idx[k]) * D + d];
        dx[i] = s;
    }
}

// scalar forms: any F
__global__ void cheby_fwd_scalar_kernel(const float* __restrict__ x, const int32_t* __restrict__ indptr,
                                 const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                 float* __restrict__ y, int B, int V, int F) {
    const long long total = (long long)B * V * F;
    GRID_STRIDE(i, total) {
        const int f = (int)(i % F);
        const long long t = i / F;
        const int v = (int)(t % V);
        const int b = (int)(t / V);
        const float* xb = x + (long long)b * V * F + f;
        float s = 0.f;
        for (int k = indptr[v]; k < indptr[v + 1]; ++k) s += vals[k] * xb[(long long)indices[k] * F];
        float2 o;
        o.x = xb[(long long)v * F];
        o.y = s;
        reinterpret_cast<float2*>(y)[i] = o;
    }
}

__global__ void cheby_bwd_scalar_kernel(const float* __restrict__ dy, const int32_t* __restrict__ indptr,
                                 const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                 float* __restrict__ dx, int B, int V, int F) {
    const long long total = (long long)B * V * F;
    GRID_STRIDE(i, total) {
        const int f = (int)(i % F);
        const long long t = i / F;
        const int v = (int)(t % V);
        const int b = (int)(t / V);
        const float* db = dy + (long long)b * V * 2 * F + 2 * f;
        float s = db[(long long)v * 2 * F];
        for (int k = indptr[v]; k < indptr[v + 1]; ++k) s += vals[k] * db[(long long)indices[k] * 2 * F + 1];
        dx[i] = s;
    }
}

// four features per lane (F % 4 == 0 in every layer of the decoder): 16-byte gathers of the neighbour rows, two 16-byte
// stores of the interleaved (x, Lx) pairs; the CSR row (indptr / indices / vals) is read once per four outputs
__global__ void cheby_fwd_kernel(const float* __restrict__ x, const int32_t* __restrict__ indptr,
                                 const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                 float* __restrict__ y, int B, int V, int F) {
    const int F4 = F >> 2;
    const long long total = (long long)B * V * F4;
    GRID_STRIDE(i, total) {
        const int f4 = (int)(i % F4);
        const long long t = i / F4;
        const int v = (int)(t % V);
        const int b = (int)(t / V);
        const float4* xb = reinterpret_cast<const float4*>(x + (long long)b * V * F) + f4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // four neighbours per round: their (index, weight) pairs, then their rows, are all in flight before the first FMA
        // (a one-neighbour loop is a chain of two dependent loads per step: the kernel is latency-bound, not HBM-bound)
        const int kend = indptr[v + 1];
        for (int k = indptr[v]; k < kend; k += 4) {
            int idx[4];
            float w[4];
            float4 n[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = k + j < kend;
                idx[j] = in ? indices[k + j] : v;
                w[j] = in ? vals[k + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) n[j] = xb[(long long)idx[j] * F4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < kend) { s.x += w[j] * n[j].x; s.y += w[j] * n[j].y; s.z += w[j] * n[j].z; s.w += w[j] * n[j].w; }
        }
        const float4 c = xb[(long long)v * F4];
        float4* o = reinterpret_cast<float4*>(y) + 2 * i;
        o[0] = make_float4(c.x, s.x, c.y, s.y);
        o[1] = make_float4(c.z, s.z, c.w, s.w);
    }
}

__global__ void cheby_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ indptr,
                                 const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                 float* __restrict__ dx, int B, int V, int F) {
    const int F4 = F >> 2;
    const long long total = (long long)B * V * F4;
    GRID_STRIDE(i, total) {
        const int f4 = (int)(i % F4);
        const long long t = i / F4;
        const int v = (int)(t % V);
        const int b = (int)(t / V);
        // dy row = [B][V][2F] as float4 pairs: (d x0, d Lx0, d x1, d Lx1), (d x2, d Lx2, d x3, d Lx3)
        const float4* db = reinterpret_cast<const float4*>(dy + (long long)b * V * 2 * F) + 2 * f4;
        const float4 a0 = db[(long long)v * 2 * F4], a1 = db[(long long)v * 2 * F4 + 1];
        float4 s = make_float4(a0.x, a0.z, a1.x, a1.z);
        const int kend = indptr[v + 1];
        for (int k = indptr[v]; k < kend; k += 4) {      // four neighbours in flight per round, see cheby_fwd_kernel
            int idx[4];
            float w[4];
            float4 n0[4], n1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = k + j < kend;
                idx[j] = in ? indices[k + j] : v;
                w[j] = in ? vals[k + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                n0[j] = db[(long long)idx[j] * 2 * F4];
                n1[j] = db[(long long)idx[j] * 2 * F4 + 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < kend) { s.x += w[j] * n0[j].y; s.y += w[j] * n0[j].w; s.z += w[j] * n1[j].y; s.w += w[j] * n1[j].w; }
        }
        reinterpret_cast<float4*>(dx)[i] = s;
    }
}

// orthographic projection uv = (scale*img) * xyz[:2] + (trans*img/2 + img/2)   (utils/manoutils.py:26-44)
__global__ void project_fwd_kernel(const float* __restrict__ v, const float* __restrict__ scale,
                                   const float* __restrict__ trans, float* __restrict__ out, int B, int V, float img) {
    const long long total = (long long)B * V;
    GRID_STRIDE(i, total) {
        const int b = (int)(i / V);
        const float s = scale[b] * img;
        const float t0 = trans[b * 2 + 0] * img / 2.f + img / 2.f;
        const float t1 = trans[b * 2 + 1] * img / 2.f + img / 2.f;
        out[i * 2 + 0] = s * v[i * 3 + 0] + t0;
        out[i * 2 + 1] = s * v[i * 3 + 1] + t1;
    }
}

__global__ __launch_bounds__(TPB) void project_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                          const float* __restrict__ scale, float* __restrict__ dv,
                                                          float* __restrict__ dscale, float* __restrict__ dtrans,
                                                          int V, float img) {
    __shared__ float red[3][TPB / 64];
    const int b = blockIdx.x;
    const float s = scale[b] * img;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < V; i += TPB) {
        const long long o = (long long)b * V + i;
        const float d0 = dout[o * 2 + 0], d1 = dout[o * 2 + 1];
        dv[o * 3 + 0] = s * d0;
        dv[o * 3 + 1] = s * d1;
        dv[o * 3 + 2] = 0.f;
        a0 += d0 * v[o * 3 + 0] + d1 * v[o * 3 + 1];
        a1 += d0;
        a2 += d1;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; red[2][wave] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        for (int w = 0; w < TPB / 64; ++w) { r0 += red[0][w]; r1 += red[1][w]; r2 += red[2][w]; }
        dscale[b] = r0 * img;
        dtrans[b * 2 + 0] = r1 * img / 2.f;
        dtrans[b * 2 + 1] = r2 * img / 2.f;
    }
}

// Parameter gradients of many LayerNorms finished by ONE launch (rih_ln_param_final_multi): the same per-channel reduction as
// ln_param_final_kernel, the descriptors by value in the kernel argument (see splitk_reduce_multi_kernel in rih_gemm.hip).
constexpr int LN_FINAL_PACK = 100;
struct LnFinalPack {
    rih_ln_final_desc d[LN_FINAL_PACK];
    int first[LN_FINAL_PACK + 1];
    int n;
};
static_assert(sizeof(LnFinalPack) <= 4096, "kernel argument limit");
__global__ __launch_bounds__(TPB) void ln_param_final_multi_kernel(const LnFinalPack pk) {
    const int bidx = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < pk.n && bidx >= pk.first[i + 1]) ++i;
    const rih_ln_final_desc& d = pk.d[i];
    const int lane = threadIdx.x & 63;
    const int e = (bidx - pk.first[i]) * (TPB / 64) + (threadIdx.x >> 6);
    if (e >= d.D) return;
    double a = 0.0, b = 0.0;
    for (int k = lane; k < d.nblk; k += 64) {
        a += (double)d.ws[((long long)k * 2 + 0) * d.D + e];
        b += (double)d.ws[((long long)k * 2 + 1) * d.D + e];
    }
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    if (lane == 0) { d.dg[e] = (float)a; d.db[e] = (float)b; }
}

// ---- Chebyshev feature build through LDS (V <= CH_MAXV vertices: the 63 / 126 / 252-vertex levels of the mesh decoder).
// The gather kernels above read every vertex row ~7 times (once per neighbour; the backward even 2 x 16 bytes per neighbour for
// 16 useful ones): 0.46 GB of L2 traffic for a 33 MB tensor, 36 us per launch.  Here a block owns one sample and a slice of
// CH_FS channel quads, stages the V rows of the slice in LDS once (coalesced) and gathers from LDS.  Same summation order as
// the gather kernels (bit-identical results).
constexpr int CH_MAXV = 256, CH_FS = 4, CH_ITEMS = CH_MAXV * CH_FS / TPB;
__global__ __launch_bounds__(TPB) void cheby_fwd_lds_kernel(const float* __restrict__ x, const int32_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, float* __restrict__ y, int V,
                                                            int F) {
    __shared__ float4 t[CH_MAXV * CH_FS];
    const int F4 = F >> 2, f4base = blockIdx.x * CH_FS, b = blockIdx.y;
    const int nfs = min(CH_FS, F4 - f4base);
    const float4* xb = reinterpret_cast<const float4*>(x + (long long)b * V * F) + f4base;
    for (int i = threadIdx.x; i < V * CH_FS; i += TPB) {
        const int v = i / CH_FS, j = i % CH_FS;
        if (j < nfs) t[i] = xb[(long long)v * F4 + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < V * CH_FS; i += TPB) {
        const int v = i / CH_FS, j = i % CH_FS;
        if (j >= nfs) continue;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = indptr[v]; k < indptr[v + 1]; ++k) {
            const float w = vals[k];
            const float4 n = t[indices[k] * CH_FS + j];
            s.x += w * n.x; s.y += w * n.y; s.z += w * n.z; s.w += w * n.w;
        }
        const float4 c = t[i];
        float4* o = reinterpret_cast<float4*>(y) + 2 * (((long long)b * V + v) * F4 + f4base + j);
        o[0] = make_float4(c.x, s.x, c.y, s.y);
        o[1] = make_float4(c.z, s.z, c.w, s.w);
    }
}
__global__ __launch_bounds__(TPB) void cheby_bwd_lds_kernel(const float* __restrict__ dy, const int32_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, float* __restrict__ dx, int V,
                                                            int F) {
    __shared__ float4 t[CH_MAXV * CH_FS];       // the Laplacian half (d Lx) of the slice's gradient rows
    const int F4 = F >> 2, f4base = blockIdx.x * CH_FS, b = blockIdx.y;
    const int nfs = min(CH_FS, F4 - f4base);
    const float4* db = reinterpret_cast<const float4*>(dy + (long long)b * V * 2 * F) + 2 * f4base;
    float4 own[CH_ITEMS];                       // the identity half (d x) of the item this thread finishes below
#pragma unroll
    for (int it = 0; it < CH_ITEMS; ++it) {
        const int i = threadIdx.x + it * TPB;
        const int v = i / CH_FS, j = i % CH_FS;
        own[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < V * CH_FS && j < nfs) {
            const float4 a0 = db[(long long)v * 2 * F4 + 2 * j], a1 = db[(long long)v * 2 * F4 + 2 * j + 1];
            own[it] = make_float4(a0.x, a0.z, a1.x, a1.z);
            t[i] = make_float4(a0.y, a0.w, a1.y, a1.w);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CH_ITEMS; ++it) {
        const int i = threadIdx.x + it * TPB;
        const int v = i / CH_FS, j = i % CH_FS;
        if (i >= V * CH_FS || j >= nfs) continue;
        float4 s = own[it];
        for (int k = indptr[v]; k < indptr[v + 1]; ++k) {
            const float w = vals[k];
            const float4 n = t[indices[k] * CH_FS + j];
            s.x += w * n.x; s.y += w * n.y; s.z += w * n.z; s.w += w * n.w;
        }
        reinterpret_cast<float4*>(dx)[((long long)b * V + v) * F4 + f4base + j] = s;
    }
}

// ---- 16-byte softmax kernels (cols % 4 == 0, ld % 4 == 0: the 316-token attention of the finest mesh level, 64 / 256-token
// ones): a lane owns column quads 4*lane + 256*i.  Same hash index per element as the scalar kernels (identical masks).
template <int NV>
__global__ __launch_bounds__(TPB) void softmax_fwd_vec_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                              float* __restrict__ Pd, long long rows, int cols, int ld,
                                                              float drop_p, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_dev) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        float4 v[NV];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            v[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            if (c < cols) {
                v[i] = *reinterpret_cast<const float4*>(S + r * ld + c);
                mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
            }
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (4 * lane + 256 * i < cols) {
                v[i].x = expf(v[i].x - mx); v[i].y = expf(v[i].y - mx); v[i].z = expf(v[i].z - mx); v[i].w = expf(v[i].w - mx);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
        s = wave_sum(s);
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (c < cols) {
                const float4 pr = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
                *reinterpret_cast<float4*>(P + r * ld + c) = pr;
                if (drop_p > 0.f) {
                    const uint64_t base = (uint64_t)(r * cols + c);
                    float4 o;
                    o.x = rih_hash(seed, base) >= thr ? pr.x * keep_scale : 0.f;
                    o.y = rih_hash(seed, base + 1) >= thr ? pr.y * keep_scale : 0.f;
                    o.z = rih_hash(seed, base + 2) >= thr ? pr.z * keep_scale : 0.f;
                    o.w = rih_hash(seed, base + 3) >= thr ? pr.w * keep_scale : 0.f;
                    *reinterpret_cast<float4*>(Pd + r * ld + c) = o;
                } else if (Pd != P) {
                    *reinterpret_cast<float4*>(Pd + r * ld + c) = pr;
                }
            }
        }
    }
}
template <int NV>
__global__ __launch_bounds__(TPB) void softmax_bwd_vec_kernel(const float* __restrict__ P, float* __restrict__ dPd,
                                                              long long rows, int cols, int ld, float drop_p, uint64_t seed,
                                                              const uint64_t* __restrict__ seed_dev, float alpha) {
    if (seed_dev != nullptr) seed += *seed_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t thr = drop_thresh(drop_p);
    const float keep_scale = (drop_p > 0.f) ? 1.f / (1.f - drop_p) : 1.f;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        float4 pv[NV], dp[NV];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            dp[i] = pv[i];
            if (c < cols) {
                pv[i] = *reinterpret_cast<const float4*>(P + r * ld + c);
                float4 d = *reinterpret_cast<const float4*>(dPd + r * ld + c);
                if (drop_p > 0.f) {
                    const uint64_t base = (uint64_t)(r * cols + c);
                    d.x = rih_hash(seed, base) >= thr ? d.x * keep_scale : 0.f;
                    d.y = rih_hash(seed, base + 1) >= thr ? d.y * keep_scale : 0.f;
                    d.z = rih_hash(seed, base + 2) >= thr ? d.z * keep_scale : 0.f;
                    d.w = rih_hash(seed, base + 3) >= thr ? d.w * keep_scale : 0.f;
                }
                dp[i] = d;
                dot += (d.x * pv[i].x + d.y * pv[i].y) + (d.z * pv[i].z + d.w * pv[i].w);
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = 4 * lane + 256 * i;
            if (c < cols)
                *reinterpret_cast<float4*>(dPd + r * ld + c) =
                    make_float4(alpha * pv[i].x * (dp[i].x - dot), alpha * pv[i].y * (dp[i].y - dot),
                                alpha * pv[i].z * (dp[i].z - dot), alpha * pv[i].w * (dp[i].w - dot));
        }
    }
}

// ------------------------------------------------------------------------------------------------ optimizer
// Adam / AdamW over a TABLE of parameter tensors in one launch.  A block owns one 4096-element chunk of one tensor
// (blk_tensor / blk_chunk); 16-byte accesses when the tensor's four pointers allow.  The step is HBM-bound: 4 reads
// + 3 writes per element.  Same update as torch.optim.Adam(amsgrad=False, maximize=False):
//   g += wd * p  (Adam)  |  p *= 1 - lr * wd  (AdamW);  m += (g - m)(1 - b1);  v = b2 v + (1 - b2) g^2;
//   p -= (lr / bias1) * m / (sqrt(v) / sqrt(bias2) + eps)
constexpr int ADAM_CHUNK = 4096;
struct AdamScalars {
    float lr, beta1, beta2, eps, wd, step_size, inv_sqrt_bias2;
    int adamw;
};
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& a) {
    if (a.wd != 0.f) {
        if (a.adamw) p *= 1.f - a.lr * a.wd;
        else g += a.wd * p;
    }
    m += (g - m) * (1.f - a.beta1);
    v = v * a.beta2 + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bias2 + a.eps;
    p -= a.step_size * (m / denom);
}
__global__ __launch_bounds__(TPB) void adam_multi_kernel(const rih_adam_entry* __restrict__ table,
                                                         const int* __restrict__ blk_tensor,
                                                         const int* __restrict__ blk_chunk, const AdamScalars a) {
    const rih_adam_entry e = table[blk_tensor[blockIdx.x]];
    const long long base = (long long)blk_chunk[blockIdx.x] * ADAM_CHUNK;
    const long long n = e.n;
    const bool vec = ((((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) == 0);
    if (vec) {
#pragma unroll
        for (int u = 0; u < ADAM_CHUNK / (TPB * 4); ++u) {
            const long long i = base + ((long long)u * TPB + threadIdx.x) * 4;
            if (i + 3 < n) {
                float4 p = *reinterpret_cast<float4*>(e.p + i);
                const float4 g = *reinterpret_cast<const float4*>(e.g + i);
                float4 m = *reinterpret_cast<float4*>(e.m + i);
                float4 v = *reinterpret_cast<float4*>(e.v + i);
                adam_one(p.x, g.x, m.x, v.x, a);
                adam_one(p.y, g.y, m.y, v.y, a);
                adam_one(p.z, g.z, m.z, v.z, a);
                adam_one(p.w, g.w, m.w, v.w, a);
                *reinterpret_cast<float4*>(e.p + i) = p;
                *reinterpret_cast<float4*>(e.m + i) = m;
                *reinterpret_cast<float4*>(e.v + i) = v;
            } else {
                for (long long j = i; j < n && j < i + 4; ++j) adam_one(e.p[j], e.g[j], e.m[j], e.v[j], a);
            }
        }
    } else {
        for (int u = threadIdx.x; u < ADAM_CHUNK; u += TPB) {
            const long long j = base + u;
            if (j < n) adam_one(e.p[j], e.g[j], e.m[j], e.v[j], a);
        }
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)
#define LAUNCH_RET() return (int)hipGetLastError()

extern "C" int rih_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream) {
    if (!x || !y || N < 1 || C < 1 || H < 1 || W < 1 || Cpad < C) return RIH_EINVAL;
    const long long total = (long long)N * H * W * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(TPB), 0, STREAM, x, y, N, C, H, W, Cpad);
    LAUNCH_RET();
}
extern "C" int rih_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, void* stream) {
    if (!x || !y || N < 1 || C < 1 || H < 1 || W < 1 || ldx < C) return RIH_EINVAL;
    const long long total = (long long)N * H * W * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(TPB), 0, STREAM, x, y, N, C, H, W, ldx);
    LAUNCH_RET();
}
extern "C" int rih_maxpool3x3s2_fwd(const float* x, float* y, int8_t* arg, int N, int H, int W, int C, void* stream) {
    if (!x || !y || !arg || N < 1 || H < 1 || W < 1 || C < 1) return RIH_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * C;
    if (C % 4 == 0 && al16(x) && al16(y) && al16(arg))
        hipLaunchKernelGGL(maxpool_fwd_kernel<4>, dim3(grid_for(total / 4)), dim3(TPB), 0, STREAM, x, y, arg, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<1>, dim3(grid_for(total)), dim3(TPB), 0, STREAM, x, y, arg, N, H, W, C, Ho, Wo);
    LAUNCH_RET();
}
extern "C" int rih_maxpool3x3s2_bwd(const float* dy, const int8_t* arg, float* dx, int N, int H, int W, int C,
                                    void* stream) {
    if (!dy || !dx || !arg || N < 1 || H < 1 || W < 1 || C < 1) return RIH_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long total = (long long)N * H * W * C;
    if (C % 4 == 0 && al16(dy) && al16(dx) && al16(arg))
        hipLaunchKernelGGL(maxpool_bwd_kernel<4>, dim3(grid_for(total / 4)), dim3(TPB), 0, STREAM, dy, arg, dx, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(grid_for(total)), dim3(TPB), 0, STREAM, dy, arg, dx, N, H, W, C, Ho, Wo);
    LAUNCH_RET();
}
extern "C" int rih_avgpool_fwd(const float* x, float* y, int N, int HW, int C, void* stream) {
    if (!x || !y || N < 1 || HW < 1 || C < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long long)N * C)), dim3(TPB), 0, STREAM, x, y, N, HW, C);
    LAUNCH_RET();
}
extern "C" int rih_avgpool_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream) {
    if (!dy || !dx || N < 1 || HW < 1 || C < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long long)N * HW * C)), dim3(TPB), 0, STREAM, dy, dx, N, HW, C);
    LAUNCH_RET();
}
extern "C" int rih_upsample_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int factor, void* stream) {
    if (!x || !y || N < 1 || H < 1 || W < 1 || C < 1 || factor < 1 || factor > 64) return RIH_EINVAL;
    const long long total = (long long)N * factor * factor * H * W * C;
    if (C % 4 == 0 && al16(x) && al16(y))
        hipLaunchKernelGGL(upsample2x_fwd_kernel<4>, dim3(grid_for(total / 4)), dim3(TPB), 0, STREAM, x, y, N, H, W, C, factor);
    else
        hipLaunchKernelGGL(upsample2x_fwd_kernel<1>, dim3(grid_for(total)), dim3(TPB), 0, STREAM, x, y, N, H, W, C, factor);
    LAUNCH_RET();
}
extern "C" int rih_upsample_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int factor, void* stream) {
    if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 1 || factor < 1 || factor > 64) return RIH_EINVAL;
    const long long total = (long long)N * H * W * C;
    if (C % 4 == 0 && al16(dy) && al16(dx))
        hipLaunchKernelGGL(upsample2x_bwd_kernel<4>, dim3(grid_for(total / 4)), dim3(TPB), 0, STREAM, dy, dx, N, H, W, C, factor);
    else
        hipLaunchKernelGGL(upsample2x_bwd_kernel<1>, dim3(grid_for(total)), dim3(TPB), 0, STREAM, dy, dx, N, H, W, C, factor);
    LAUNCH_RET();
}
extern "C" int rih_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    return rih_upsample_bilinear_fwd(x, y, N, H, W, C, 2, stream);
}
extern "C" int rih_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    return rih_upsample_bilinear_bwd(dy, dx, N, H, W, C, 2, stream);
}
extern "C" int rih_nearest_up_add_fwd(const float* x, const float* add, float* y, int N, int H, int W, int C, int factor,
                                      void* stream) {
    if (!x || !y || N < 1 || H < 1 || W < 1 || C < 4 || (C % 4) != 0 || factor < 1 || factor > 64) return RIH_EINVAL;
    hipLaunchKernelGGL(nearest_up_add_kernel, dim3(grid_for((long long)N * factor * factor * H * W * (C / 4))), dim3(TPB),
                       0, STREAM, x, add, y, N, H, W, C / 4, factor);
    LAUNCH_RET();
}
extern "C" int rih_nearest_up_bwd(const float* dy, float* dx, int N, int H, int W, int C, int factor, void* stream) {
    if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 4 || (C % 4) != 0 || factor < 1 || factor > 64) return RIH_EINVAL;
    hipLaunchKernelGGL(nearest_up_bwd_kernel, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(TPB), 0, STREAM, dy, dx,
                       N, H, W, C / 4, factor);
    LAUNCH_RET();
}

extern "C" int64_t rih_bn_ws_floats(int rows, int C) {
    if (rows < 1 || C < 4 || (C % 4) != 0) return 0;
    const ColGeom g = col_geom(rows, C);
    return (int64_t)2 * g.nchunk * C + 2 * C;
}
extern "C" int rih_bn_stats(const float* x, int rows, int C, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, float* ws, void* stream) {
    if (!x || !mean || !invstd || !ws || rows < 1 || C < 4 || (C % 4) != 0) return RIH_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RIH_EINVAL;
    const ColGeom g = col_geom(rows, C);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(g.gx, g.nchunk), dim3(TPB), 0, STREAM, x, rows, C, g.cgb, g.rt,
                       g.rows_per_chunk, ws);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 3) / 4), dim3(TPB), 0, STREAM, x, ws, rows, C, g.nchunk, eps,
                       momentum, mean, invstd, running_mean, running_var);
    LAUNCH_RET();
}
extern "C" int rih_bn_stats_from_blocks(const float* part, int T, int C, int rows, int rows_per_block, float eps,
                                        float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                                        void* stream) {
    if (!part || !mean || !invstd || T < 1 || C < 1 || rows < 1 || rows_per_block < 1) return RIH_EINVAL;
    if ((long long)(T - 1) * rows_per_block >= rows || (long long)T * rows_per_block < rows) return RIH_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RIH_EINVAL;
    hipLaunchKernelGGL(bn_blocks_final_kernel, dim3(C), dim3(TPB), 0, STREAM, part, T, C, rows, rows_per_block, eps, momentum,
                       mean, invstd, running_mean, running_var);
    LAUNCH_RET();
}
extern "C" int rih_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                                 float* invstd, void* stream) {
    if (!running_mean || !running_var || !mean || !invstd || C < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 127) / 128), dim3(128), 0, STREAM, running_mean, running_var, C,
                       eps, mean, invstd);
    LAUNCH_RET();
}
extern "C" int rih_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* residual, float* y, int rows, int C, int relu,
                            uint8_t* relu_mask, float* amax, void* stream) {
    if (!x || !mean || !invstd || !gamma || !beta || !y || rows < 1 || C < 4 || (C % 4) != 0) return RIH_EINVAL;
    const long long nq = (long long)rows * (C / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(nq)), dim3(TPB), 0, STREAM, x, mean, invstd, gamma, beta, residual,
                       y, nq, C, relu, relu_mask, amax);
    LAUNCH_RET();
}
extern "C" int rih_bn_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                          const float* gamma, float* dx, float* dres, float* dgamma, float* dbeta, int rows, int C,
                          int relu, int frozen_stats, float* ws, const uint8_t* relu_mask, float* amax_dx, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !dx || !dgamma || !dbeta || !ws) return RIH_EINVAL;
    if (relu && !y && !relu_mask) return RIH_EINVAL;
    if (rows < 1 || C < 4 || (C % 4) != 0) return RIH_EINVAL;
    const ColGeom g = col_geom(rows, C);
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(g.gx, g.nchunk), dim3(TPB), 0, STREAM, dy, x, y, mean, invstd, rows, C,
                       g.cgb, g.rt, g.rows_per_chunk, relu, ws, relu_mask);
    hipLaunchKernelGGL(two_sum_final_kernel, dim3((C + 3) / 4), dim3(TPB), 0, STREAM, ws, C, g.nchunk, dbeta, dgamma);
    const long long nq = (long long)rows * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(nq)), dim3(TPB), 0, STREAM, dy, x, y, mean, invstd, gamma,
                       dbeta, dgamma, dx, dres, nq, C, 1.f / (float)rows, relu, frozen_stats, relu_mask, amax_dx);
    LAUNCH_RET();
}

extern "C" int rih_ln_nblk(int rows) {
    // 4 rows (wavefronts) per block pass; enough blocks that a wavefront walks only a few rows: each row is a
    // load -> shuffle-reduce -> store dependency chain (~1.5 us), so few resident waves means latency-bound
    int n = (rows + 15) / 16;
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    return n;
}
extern "C" int rih_layernorm_fwd_grouped(const float* x, const float* x2, const float* g, const float* b, float* y,
                                         float* mean, float* rstd, int groups, int rows, int D, int64_t sG, int64_t sB,
                                         float eps, int relu, void* stream) {
    if (!x || !g || !b || !y || !mean || !rstd || rows < 1 || D < 1 || D > 64 * LN_MAXPER) return RIH_EINVAL;
    if (groups < 1 || groups > 65535) return RIH_EINVAL;
    int blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    const bool vec = al16(x) && al16(g) && al16(b) && al16(y) && (x2 == nullptr || al16(x2)) && sG % 4 == 0 && sB % 4 == 0;
#define RIH_LN_FWD(LPR_, VPT_)                                                                                          \
    {                                                                                                                   \
        int bl = (rows + 4 * (64 / LPR_) - 1) / (4 * (64 / LPR_));                                                      \
        if (bl > 8192) bl = 8192;                                                                                       \
        hipLaunchKernelGGL((layernorm_fwd_vec_kernel<LPR_, VPT_>), dim3(bl, groups), dim3(TPB), 0, STREAM, x, x2, g, b, y, \
                           mean, rstd, rows, eps, relu, (long long)sG, (long long)sB);                                  \
    }
    if (vec && D == 64) RIH_LN_FWD(16, 1)
    else if (vec && D == 128) RIH_LN_FWD(32, 1)
    else if (vec && D == 256) RIH_LN_FWD(64, 1)
    else if (vec && D == 512) RIH_LN_FWD(64, 2)
    else if (vec && D == 1024) RIH_LN_FWD(64, 4)
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(blocks, groups), dim3(TPB), 0, STREAM, x, x2, g, b, y, mean, rstd, rows,
                           D, eps, relu, (long long)sG, (long long)sB);
#undef RIH_LN_FWD
    LAUNCH_RET();
}
extern "C" int rih_layernorm_fwd(const float* x, const float* x2, const float* g, const float* b, float* y, float* mean,
                                 float* rstd, int rows, int D, float eps, int relu, void* stream) {
    return rih_layernorm_fwd_grouped(x, x2, g, b, y, mean, rstd, 1, rows, D, 0, 0, eps, relu, stream);
}
extern "C" int rih_layernorm_bwd_grouped(const float* dy, const float* x, const float* x2, const float* y, const float* g,
                                         const float* mean, const float* rstd, const float* dres, float* dx, float* dg,
                                         float* db, int groups, int rows, int D, int64_t sG, int relu, float* ws,
                                         void* stream) {
    if (!dy || !x || !g || !mean || !rstd || !dx || !ws || ((dg == nullptr) != (db == nullptr))) return RIH_EINVAL;
    if (relu && !y) return RIH_EINVAL;
    if (rows < 1 || D < 1 || D > 64 * LN_MAXPER || groups < 1 || groups > 65535) return RIH_EINVAL;
    const int nblk = rih_ln_nblk(rows);
    const bool vec = al16(dy) && al16(x) && al16(g) && al16(dx) && al16(ws) && (x2 == nullptr || al16(x2)) &&
                     (y == nullptr || al16(y)) && (dres == nullptr || al16(dres)) && sG % 4 == 0;
#define RIH_LN_BWD(LPR_, VPT_)                                                                                           \
    hipLaunchKernelGGL((layernorm_bwd_vec_kernel<LPR_, VPT_>), dim3(nblk, groups), dim3(TPB), 0, STREAM, dy, x, x2, y, g, \
                       mean, rstd, dres, dx, ws, rows, relu, (long long)sG);
    if (vec && D == 64) RIH_LN_BWD(16, 1)
    else if (vec && D == 128) RIH_LN_BWD(32, 1)
    else if (vec && D == 256) RIH_LN_BWD(64, 1)
    else if (vec && D == 512) RIH_LN_BWD(64, 2)
    else if (vec && D == 1024) RIH_LN_BWD(64, 4)
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk, groups), dim3(TPB), 0, STREAM, dy, x, x2, y, g, mean, rstd, dres,
                           dx, ws, rows, D, relu, (long long)sG);
#undef RIH_LN_BWD
    if (dg != nullptr)      // NULL: the caller finishes the parameter gradients later with rih_ln_param_final_multi
        hipLaunchKernelGGL(ln_param_final_kernel, dim3((D + 3) / 4, groups), dim3(TPB), 0, STREAM, ws, D, nblk, dg, db);
    LAUNCH_RET();
}
extern "C" int rih_ln_param_final_multi(const rih_ln_final_desc* descs, int n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return RIH_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!descs[i].ws || !descs[i].dg || !descs[i].db || descs[i].D < 1 || descs[i].nblk < 1) return RIH_EINVAL;
    for (int base = 0; base < n; base += LN_FINAL_PACK) {
        LnFinalPack pk;
        pk.n = (n - base < LN_FINAL_PACK) ? n - base : LN_FINAL_PACK;
        int total = 0;
        for (int i = 0; i < pk.n; ++i) {
            pk.d[i] = descs[base + i];
            pk.first[i] = total;
            total += (descs[base + i].D + 3) / 4;
        }
        pk.first[pk.n] = total;
        hipLaunchKernelGGL(ln_param_final_multi_kernel, dim3(total), dim3(TPB), 0, STREAM, pk);
    }
    LAUNCH_RET();
}
extern "C" int rih_layernorm_bwd(const float* dy, const float* x, const float* x2, const float* y, const float* g,
                                 const float* mean, const float* rstd, const float* dres, float* dx, float* dg, float* db,
                                 int rows, int D, int relu, float* ws, void* stream) {
    return rih_layernorm_bwd_grouped(dy, x, x2, y, g, mean, rstd, dres, dx, dg, db, 1, rows, D, 0, relu, ws, stream);
}
extern "C" int rih_softmax_fwd(const float* S, float* P, float* Pd, int64_t rows, int cols, int ld, float drop_p,
                               uint64_t seed, const uint64_t* seed_dev, void* stream) {
    if (!S || !P || !Pd || rows < 1 || cols < 1 || cols > 64 * SM_MAXPER || ld < cols) return RIH_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return RIH_EINVAL;
    if (drop_p > 0.f && Pd == P) return RIH_EINVAL;
    long long blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    const bool vec = cols % 4 == 0 && ld % 4 == 0 && al16(S) && al16(P) && al16(Pd);
#define RIH_SM(NV_) hipLaunchKernelGGL(softmax_fwd_vec_kernel<NV_>, dim3((int)blocks), dim3(TPB), 0, STREAM, S, P, Pd, \
                                       (long long)rows, cols, ld, drop_p, seed, seed_dev)
    if (vec && cols <= 256) RIH_SM(1);
    else if (vec && cols <= 512) RIH_SM(2);
    else if (vec) RIH_SM(4);
    else
        hipLaunchKernelGGL(softmax_fwd_kernel, dim3((int)blocks), dim3(TPB), 0, STREAM, S, P, Pd, (long long)rows, cols, ld,
                           drop_p, seed, seed_dev);
#undef RIH_SM
    LAUNCH_RET();
}
extern "C" int rih_softmax_bwd(const float* P, float* dPd, int64_t rows, int cols, int ld, float drop_p, uint64_t seed,
                               const uint64_t* seed_dev, float alpha, void* stream) {
    if (!P || !dPd || rows < 1 || cols < 1 || cols > 64 * SM_MAXPER || ld < cols) return RIH_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return RIH_EINVAL;
    long long blocks = (rows + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    const bool vec = cols % 4 == 0 && ld % 4 == 0 && al16(P) && al16(dPd);
#define RIH_SM(NV_) hipLaunchKernelGGL(softmax_bwd_vec_kernel<NV_>, dim3((int)blocks), dim3(TPB), 0, STREAM, P, dPd,   \
                                       (long long)rows, cols, ld, drop_p, seed, seed_dev, alpha)
    if (vec && cols <= 256) RIH_SM(1);
    else if (vec && cols <= 512) RIH_SM(2);
    else if (vec) RIH_SM(4);
    else
        hipLaunchKernelGGL(softmax_bwd_kernel, dim3((int)blocks), dim3(TPB), 0, STREAM, P, dPd, (long long)rows, cols, ld,
                           drop_p, seed, seed_dev, alpha);
#undef RIH_SM
    LAUNCH_RET();
}

extern "C" int rih_adam_multi(const rih_adam_entry* table, const int32_t* blk_tensor, const int32_t* blk_chunk, int nblocks,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step, int adamw,
                              void* stream) {
    if (!table || !blk_tensor || !blk_chunk || nblocks < 1 || step < 1) return RIH_EINVAL;
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return RIH_EINVAL;
    AdamScalars a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay; a.adamw = adamw;
    // bias corrections in double on the host, as torch's Python loop computes them
    const double b1 = 1.0 - pow((double)beta1, (double)step), b2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / b1);
    a.inv_sqrt_bias2 = (float)(1.0 / sqrt(b2));
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)nblocks), dim3(TPB), 0, STREAM, table, blk_tensor, blk_chunk, a);
    LAUNCH_RET();
}
extern "C" int rih_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int rih_add_dropout(const float* a, const float* b, float* y, int64_t n, int D, int b_bcast_rows,
                               float drop_p, uint64_t seed, const uint64_t* seed_dev, void* stream) {
    if (!b || !y || n < 1 || D < 1 || b_bcast_rows < 0 || drop_p < 0.f || drop_p >= 1.f) return RIH_EINVAL;
    const long long bmod = (long long)b_bcast_rows * D;
    hipLaunchKernelGGL(add_dropout_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, STREAM, a, b, y, (long long)n, bmod, drop_p,
                       seed, seed_dev);
    LAUNCH_RET();
}
extern "C" int rih_dropout_bwd(const float* dy, float* dx, int64_t n, float drop_p, uint64_t seed,
                               const uint64_t* seed_dev, void* stream) {
    if (!dy || !dx || n < 1 || drop_p < 0.f || drop_p >= 1.f) return RIH_EINVAL;
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, STREAM, dy, dx, (long long)n, drop_p, seed,
                       seed_dev);
    LAUNCH_RET();
}
extern "C" int rih_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, STREAM, x, y, (long long)n);
    LAUNCH_RET();
}
extern "C" int rih_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    if (!dy || !y || !dx || n < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, STREAM, dy, y, dx, (long long)n);
    LAUNCH_RET();
}
extern "C" int64_t rih_colsum_ws_floats(int rows, int C) {
    if (rows < 1 || C < 1) return 0;
    return (int64_t)colsum_nchunk(rows, C) * C;
}
extern "C" int rih_colsum(const float* x, int rows, int C, int ldx, float* out, int accumulate, float* ws,
                          void* stream) {
    if (!x || !out || !ws || rows < 1 || C < 1 || ldx < C) return RIH_EINVAL;
    const int nchunk = colsum_nchunk(rows, C);
    const int rpc = (rows + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, nchunk), dim3(TPB), 0, STREAM, x, rows, C, ldx, rpc, ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 3) / 4), dim3(TPB), 0, STREAM, ws, C, nchunk, out, accumulate);
    LAUNCH_RET();
}
extern "C" int rih_gather_rows(const float* x, const int32_t* idx, float* y, int B, int Vin, int Vout, int D,
                               void* stream) {
    if (!x || !idx || !y || B < 1 || Vin < 1 || Vout < 1 || D < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long long)B * Vout * D)), dim3(TPB), 0, STREAM, x, idx, y, B,
                       Vin, Vout, D);
    LAUNCH_RET();
}
extern "C" int rih_scatter_rows_add(const float* dy, const int32_t* inv_ptr, const int32_t* inv_idx, float* dx, int B,
                                    int Vin, int Vout, int D, void* stream) {
    if (!dy || !inv_ptr || !inv_idx || !dx || B < 1 || Vin < 1 || Vout < 1 || D < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(grid_for((long long)B * Vin * D)), dim3(TPB), 0, STREAM, dy, inv_ptr,
                       inv_idx, dx, B, Vin, Vout, D);
    LAUNCH_RET();
}
extern "C" int rih_cheby_fwd(const float* x, const int32_t* indptr, const int32_t* indices, const float* vals, float* y,
                             int B, int V, int F, void* stream) {
    if (!x || !indptr || !indices || !vals || !y || B < 1 || V < 1 || F < 1) return RIH_EINVAL;
    if (F % 4 == 0 && V <= CH_MAXV && B <= 65535 && al16(x) && al16(y))
        hipLaunchKernelGGL(cheby_fwd_lds_kernel, dim3((F / 4 + CH_FS - 1) / CH_FS, B), dim3(TPB), 0, STREAM, x, indptr, indices,
                           vals, y, V, F);
    else if (F % 4 == 0)
        hipLaunchKernelGGL(cheby_fwd_kernel, dim3(grid_for((long long)B * V * (F / 4))), dim3(TPB), 0, STREAM, x, indptr,
                           indices, vals, y, B, V, F);
    else
        hipLaunchKernelGGL(cheby_fwd_scalar_kernel, dim3(grid_for((long long)B * V * F)), dim3(TPB), 0, STREAM, x, indptr,
                           indices, vals, y, B, V, F);
    LAUNCH_RET();
}
extern "C" int rih_cheby_bwd(const float* dy, const int32_t* t_indptr, const int32_t* t_indices, const float* t_vals,
                             float* dx, int B, int V, int F, void* stream) {
    if (!dy || !t_indptr || !t_indices || !t_vals || !dx || B < 1 || V < 1 || F < 1) return RIH_EINVAL;
    if (F % 4 == 0 && V <= CH_MAXV && B <= 65535 && al16(dy) && al16(dx))
        hipLaunchKernelGGL(cheby_bwd_lds_kernel, dim3((F / 4 + CH_FS - 1) / CH_FS, B), dim3(TPB), 0, STREAM, dy, t_indptr,
                           t_indices, t_vals, dx, V, F);
    else if (F % 4 == 0)
        hipLaunchKernelGGL(cheby_bwd_kernel, dim3(grid_for((long long)B * V * (F / 4))), dim3(TPB), 0, STREAM, dy, t_indptr,
                           t_indices, t_vals, dx, B, V, F);
    else
        hipLaunchKernelGGL(cheby_bwd_scalar_kernel, dim3(grid_for((long long)B * V * F)), dim3(TPB), 0, STREAM, dy,
                           t_indptr, t_indices, t_vals, dx, B, V, F);
    LAUNCH_RET();
}

extern "C" int rih_project_fwd(const float* v, const float* scale, const float* trans, float* out, int B, int V,
                               float img_size, void* stream) {
    if (!v || !scale || !trans || !out || B < 1 || V < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(project_fwd_kernel, dim3(grid_for((long long)B * V)), dim3(TPB), 0, STREAM, v, scale, trans, out,
                       B, V, img_size);
    LAUNCH_RET();
}
extern "C" int rih_project_bwd(const float* dout, const float* v, const float* scale, float* dv, float* dscale,
                               float* dtrans, int B, int V, float img_size, void* stream) {
    if (!dout || !v || !scale || !dv || !dscale || !dtrans || B < 1 || V < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(project_bwd_kernel, dim3(B), dim3(TPB), 0, STREAM, dout, v, scale, dv, dscale, dtrans, V,
                       img_size);
    LAUNCH_RET();
}

extern "C" int rih_absmax(const float* x, int64_t n, float* out, void* stream) {
    if (!x || !out || n < 0) return RIH_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n, 16)), dim3(TPB), 0, STREAM, x, (long long)n, out);
    LAUNCH_RET();
}
extern "C" int rih_absmax_multi(const rih_absmax_desc* descs, int n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return RIH_EINVAL;
    for (int i = 0; i < n; ++i)
        if (!descs[i].x || !descs[i].out || descs[i].n < 1) return RIH_EINVAL;
    for (int base = 0; base < n; base += ABSMAX_PACK) {
        AbsmaxPack pk;
        pk.n = (n - base < ABSMAX_PACK) ? n - base : ABSMAX_PACK;
        int total = 0;
        for (int i = 0; i < pk.n; ++i) {
            pk.d[i] = descs[base + i];
            pk.first[i] = total;
            long long nb = (descs[base + i].n + 16 * TPB - 1) / (16 * TPB);     // 16 elements per thread
            total += (int)(nb > 256 ? 256 : nb);
        }
        pk.first[pk.n] = total;
        hipLaunchKernelGGL(absmax_multi_kernel, dim3(total), dim3(TPB), 0, STREAM, pk);
    }
    LAUNCH_RET();
}

extern "C" int rih_version(void) { return RIH_ABI_VERSION; }
extern "C" int rih_abi_sizes(int32_t* out9) {      // RIH_ABI_NSIZES values
    if (!out9) return RIH_EINVAL;
    out9[0] = (int32_t)sizeof(rih_gemm_desc);
    out9[1] = (int32_t)sizeof(rih_mano_model);
    out9[2] = (int32_t)sizeof(rih_mesh_topo);
    out9[3] = (int32_t)sizeof(rih_hconv_desc);
    out9[4] = (int32_t)sizeof(rih_reduce_desc);
    out9[5] = (int32_t)sizeof(rih_pack_desc);
    out9[6] = (int32_t)sizeof(rih_ln_final_desc);
    out9[7] = (int32_t)sizeof(rih_adam_entry);
    out9[8] = (int32_t)sizeof(rih_absmax_desc);
    out9[9] = (int32_t)sizeof(rih_conv3_desc);
    out9[10] = (int32_t)sizeof(rih_h2_desc);
    out9[11] = (int32_t)sizeof(rih_panel_desc);
    return 0;
}
extern "C" const char* rih_arch(void) { return "gfx950"; }
