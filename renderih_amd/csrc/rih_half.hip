// rih_half.hip -- fp16-storage inference backbone for gfx950 (BASELINE configs[4]: "batch=256 fp16, hipGraph-captured").
//
// In eval mode every BatchNorm of the CNN encoder is a per-channel affine map, so Conv->BN(->ReLU) (torchvision trunk,
// models/encoder.py:107-116) folds into the convolution's weights and bias, and Conv->ReLU->BN (aux decoders
// models/encoder.py:52-54, mid convs models/model_zoo/__init__.py:56-62) into a post-activation scale/shift: the whole
// encoder becomes a chain of convolutions with fused epilogues plus a max-pool, three bilinear upsamples and an average
// pool.  Activations and weights are stored as fp16 (half the HBM traffic of the fp32 path), products accumulate in fp32
// on v_mfma_f32_32x32x16_f16 (one MFMA pass instead of the split engine's six), epilogue arithmetic is fp32.
//
// hconv_kernel: implicit GEMM  Y[pixel][cout] = sum_k A[pixel][k] W[cout][k],  k = (kh, kw, cin), NHWC fp16 input.
//   * 128 x BN output tile per workgroup (BN = 128 or 64), 4 wavefronts as 2 x 2, each 64 x BN/2 = 2 x (BN/64) MFMA tiles;
//     k-tiles of 64 (4 MFMA k-steps), two LDS stages.
//   * Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass).  One
//     wave-instruction fills 8 rows x 128 B; the LDS image is lane-linear, so the XOR swizzle that makes the ds_read_b128
//     operand fetches bank-conflict free (16-byte chunk c of row r lives at chunk c ^ ((r>>1)&7)) is applied on the SOURCE
//     address.  A lane whose chunk is padding (conv halo, pixel >= M, k >= K, cout >= Cout) reads a 16-byte zero page instead.
//   * the (tap, channel) walk of the im2col gather is per lane and incremental (no division in the loop); Cin % 8 == 0 so a
//     chunk never straddles a tap.
//   * epilogue: accumulators -> LDS (fp32) -> each lane owns 8 consecutive output channels of a pixel: + bias + residual
//     (16-byte load) -> ReLU -> post scale/shift -> one 16-byte fp16 store (or two fp32 stores for the tensors handed to
//     the fp32 mesh decoder).
// STATUS: written after the round's GPU budget was spent; verified on the HIP-on-CPU harness (tests/test_half.py) against
// torch on fp16-rounded operands; not yet run or measured on a GPU.  Opt-in (HandNET_GCN.use_fp16_backbone()).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/renderih_amd.h"

namespace {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int TPB = 256;
constexpr int BM = 128;
constexpr int BKH = 64;                 // halves per k-tile = 128 bytes per LDS row
constexpr int ROWB = 128;               // bytes per LDS row
constexpr int STAGE_LD = 36;            // epilogue staging: floats per row (32 + pad, keeps 16-byte alignment)

struct HConvArgs {
    const half_t* x;
    const half_t* w;
    const half_t* zero;
    const float* bias;
    const float* post_scale;
    const float* post_shift;
    const half_t* res;
    void* y;
    int M, Cout, Kpad;
    int H, W, Cin, KH, KW, stride, pad, Ho, Wo;
    int ldx, ldr, ldy, relu, out_f32, vec, cvec;
};

__device__ __forceinline__ int xcd_chunk(int bid, int nwg) {       // each XCD (private L2) gets a contiguous run of tiles
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ void glds16(const half_t* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ok ? a : b on addresses WITHOUT control flow: hipcc otherwise sinks each LDS-DMA into both arms of a branch (one arm has a
// scalar base) and the loader becomes a chain of divergent branches
__device__ __forceinline__ const half_t* pick(bool ok, const half_t* a, const half_t* b) {
    const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
    return (const half_t*)(((uintptr_t)a & m) | ((uintptr_t)b & ~m));
}

__device__ __forceinline__ float clamp_h(float v) { return fminf(fmaxf(v, -65504.f), 65504.f); }

// GLDS = false: the same data movement through registers (16-byte global loads issued before a tile's MFMAs, ds_write_b128
// after them) -- the conventional staging, kept as the A/B partner and fallback of the LDS-DMA path (RIH_HCONV_GLDS=0).
template <int BN, bool GLDS>
__global__ __launch_bounds__(TPB, 2) void hconv_kernel(const HConvArgs p) {
    constexpr int WN = BN / 2, TN = WN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / 32, NB = BN / 32;               // LDS-DMA instructions per wave and k-tile
    static_assert(2 * STAGE >= 4 * 64 * STAGE_LD * 4, "epilogue staging must fit in the operand stages");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tilesN = (p.Cout + BN - 1) / BN;
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int m0 = (bid / tilesN) * BM, n0 = (bid % tilesN) * BN;

    // ------------------------------------------------------------ loader state: rows 32 i + 8 wave + lane/8, one chunk
    const int lrow = 8 * wave + (lane >> 3);
    const int chunk = (lane & 7) ^ ((lrow >> 1) & 7);       // logical 8-half chunk this lane fetches (same for all i)
    int hi0[NA], wi0[NA], pix[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + 32 * i + lrow;
        if (m < p.M) {
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int img = t / p.Ho;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            pix[i] = (img * p.H + hi0[i]) * p.W + wi0[i];
        } else {
            hi0[i] = -(1 << 24);                            // every tap fails the range test
            wi0[i] = 0;
            pix[i] = 0;
        }
    }
    int kh, kw, ci;
    {
        const int k = 8 * chunk;
        const int tap = k / p.Cin;
        ci = k - tap * p.Cin;
        kh = tap / p.KW;
        kw = tap - kh * p.KW;
    }
    const half_t* wsrc[NB];                                 // rows beyond Cout: the zero page, step 0
    int wstep[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + 32 * i + lrow;
        wsrc[i] = (n < p.Cout) ? p.w + (long long)n * p.Kpad + 8 * chunk : p.zero;
        wstep[i] = (n < p.Cout) ? BKH : 0;
    }
    const int ntiles = p.Kpad / BKH;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 regs[NA + NB];                                    // register staging (GLDS = false only)
    auto stage = [&](int buf) {                             // issue the loads of the next k-tile
        unsigned char* a_dst = smem + buf * STAGE + (8 * wave) * ROWB + lane * 16;
        const bool kok = kh < p.KH;
        const int tapoff = kh * p.W + kw;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = (int)kok & (int)((unsigned)(hi0[i] + kh) < (unsigned)p.H) & (int)((unsigned)(wi0[i] + kw) < (unsigned)p.W);
            const long long off = (long long)(pix[i] + tapoff) * p.ldx + ci;
            const half_t* src = pick(ok, p.x + off, p.zero);
            if (GLDS) glds16(src, a_dst + 32 * i * ROWB);
            else regs[i] = *(const u32x4*)src;
        }
        unsigned char* b_dst = a_dst + A_BYTES;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (GLDS) glds16(wsrc[i], b_dst + 32 * i * ROWB);
            else regs[NA + i] = *(const u32x4*)wsrc[i];
            wsrc[i] += wstep[i];
        }
        ci += BKH;                                          // advance the (tap, channel) walk by one k-tile
        while (ci >= p.Cin) {
            ci -= p.Cin;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    auto land = [&](int buf) {                              // GLDS = false: registers -> the same lane-linear LDS image
        if (GLDS) return;
        unsigned char* a_dst = smem + buf * STAGE + (8 * wave) * ROWB + lane * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) *(u32x4*)(a_dst + 32 * i * ROWB) = regs[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *(u32x4*)(a_dst + A_BYTES + 32 * i * ROWB) = regs[NA + i];
    };

    floatx16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0);
    // ------------------------------------------------------------ what the epilogue will need from memory, requested NOW: the
    // lane's 8 output channels per column block j (bias, post scale / shift) and the residual of every element it will write
    // (4 pixels per j).  They land while the k-loop runs; issued in the epilogue they cost one exposed memory latency each
    // (measured: the 1x1 convolutions with a residual ran at 0.25 of their HBM roofline, profiles/r04/hconv_sweep_c10.log)
    const int ech = lane & 3, erow = lane >> 2;             // item = lane + 64 q: channel block ech, row erow + 16 q of the wave's 64
    const int em0 = m0 + wm * 64 + erow, en0 = n0 + wn * WN + 8 * ech;
    float bia[TN][8], psc[TN][8], psh[TN][8];
    half8 rres[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = en0 + j * 32;
        if (p.cvec && n + 8 <= p.Cout) {                      // 16-byte loads (the build does not merge scalar ones)
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f), o = make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 b0 = p.bias ? *(const float4*)(p.bias + n) : z, b1 = p.bias ? *(const float4*)(p.bias + n + 4) : z;
            const float4 s0 = p.post_scale ? *(const float4*)(p.post_scale + n) : o, s1 = p.post_scale ? *(const float4*)(p.post_scale + n + 4) : o;
            const float4 t0 = p.post_scale ? *(const float4*)(p.post_shift + n) : z, t1 = p.post_scale ? *(const float4*)(p.post_shift + n + 4) : z;
            bia[j][0] = b0.x; bia[j][1] = b0.y; bia[j][2] = b0.z; bia[j][3] = b0.w; bia[j][4] = b1.x; bia[j][5] = b1.y; bia[j][6] = b1.z; bia[j][7] = b1.w;
            psc[j][0] = s0.x; psc[j][1] = s0.y; psc[j][2] = s0.z; psc[j][3] = s0.w; psc[j][4] = s1.x; psc[j][5] = s1.y; psc[j][6] = s1.z; psc[j][7] = s1.w;
            psh[j][0] = t0.x; psh[j][1] = t0.y; psh[j][2] = t0.z; psh[j][3] = t0.w; psh[j][4] = t1.x; psh[j][5] = t1.y; psh[j][6] = t1.z; psh[j][7] = t1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = n + e < p.Cout;
                bia[j][e] = (p.bias != nullptr && ok) ? p.bias[n + e] : 0.f;
                psc[j][e] = (p.post_scale != nullptr && ok) ? p.post_scale[n + e] : 1.f;
                psh[j][e] = (p.post_scale != nullptr && ok) ? p.post_shift[n + e] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = em0 + 16 * q;
#pragma unroll
            for (int e = 0; e < 8; ++e) rres[j][q][e] = (half_t)0.f;
            if (p.res != nullptr && p.vec && m < p.M && n + 8 <= p.Cout) rres[j][q] = *(const half8*)(p.res + (long long)m * p.ldr + n);
        }
    }
    land(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) stage(buf ^ 1);          // lands while this tile is multiplied
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            const int c = 2 * s + lhi;                      // lanes 0-31 supply k 0..7 of the step, lanes 32-63 k 8..15
            half8 a[2], b[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm * 64 + i * 32 + l31;
                a[i] = *(const half8*)(As + r * ROWB + 16 * (c ^ ((r >> 1) & 7)));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WN + j * 32 + l31;
                b[j] = *(const half8*)(Bs + r * ROWB + 16 * (c ^ ((r >> 1) & 7)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ntiles) land(buf ^ 1);
        __syncthreads();                                    // tile t+1 has landed, nobody still reads tile t
    }

    // ------------------------------------------------------------ epilogue through LDS: 8 output channels per lane
    float* st = (float*)smem + wave * 64 * STAGE_LD;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (j > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * STAGE_LD + l31] = acc[i][j][r];
        __syncthreads();
        const int n = en0 + j * 32;
        const bool full = p.vec && n + 8 <= p.Cout;
        const int cnt = n >= p.Cout ? 0 : (full ? 8 : min(8, p.Cout - n));
        // pass 1: every value of this lane for this j is finished in registers before its first store -- a store through y may
        // alias any later load as far as the compiler knows, and a load issued after it waits out a full memory latency
        float v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = erow + 16 * q;
            const float4 v0 = *(const float4*)(st + row * STAGE_LD + 8 * ech), v1 = *(const float4*)(st + row * STAGE_LD + 8 * ech + 4);
            const float t[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const int m = em0 + 16 * q;
            float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.res != nullptr && m < p.M && cnt > 0) {
                if (full) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) rv[e] = (float)rres[j][q][e];
                } else {
                    const half_t* rp = p.res + (long long)m * p.ldr + n;
                    for (int e = 0; e < cnt; ++e) rv[e] = (float)rp[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float u = t[e] + rv[e] + bia[j][e];
                if (p.relu) u = fmaxf(u, 0.f);
                v[q][e] = u * psc[j][e] + psh[j][e];
            }
        }
        // pass 2: the stores
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = em0 + 16 * q;
            if (m >= p.M || cnt == 0) continue;
            if (p.out_f32) {
                float* yp = (float*)p.y + (long long)m * p.ldy + n;
                if (full) {
                    *(float4*)yp = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
                    *(float4*)(yp + 4) = make_float4(v[q][4], v[q][5], v[q][6], v[q][7]);
                } else {
                    for (int e = 0; e < cnt; ++e) yp[e] = v[q][e];
                }
            } else {
                half_t* yp = (half_t*)p.y + (long long)m * p.ldy + n;
                if (full) {
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)clamp_h(v[q][e]);
                    *(half8*)yp = h;
                } else {
                    for (int e = 0; e < cnt; ++e) yp[e] = (half_t)clamp_h(v[q][e]);
                }
            }
        }
    }
}

// fp32 OIHW weights -> fp16 [Cout][Kpad], k = (kh, kw, ci) with ci padded to CinPad, times an optional per-cout scale (the
// folded BatchNorm); zero beyond K
__global__ void hpack_weight_kernel(const float* __restrict__ w, const float* __restrict__ scale, half_t* __restrict__ dst,
                                    int Cout, int Cin, int KH, int KW, int CinPad, int Kpad) {
    const long long total = (long long)Cout * Kpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), n = (int)(i / Kpad);
        const int tap = k / CinPad, ci = k - tap * CinPad;
        float v = 0.f;
        if (tap < KH * KW && ci < Cin) {
            const int kh = tap / KW, kw = tap - kh * KW;
            v = w[(((long long)n * Cin + ci) * KH + kh) * KW + kw];
            if (scale != nullptr) v *= scale[n];
        }
        dst[i] = (half_t)clamp_h(v);
    }
}

// BatchNorm (eval) -> scale = gamma / sqrt(var + eps), shift = beta - mean * scale (+ scale * conv_bias)
__global__ void hbn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                const float* __restrict__ var, const float* __restrict__ conv_bias, float eps,
                                float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = (gamma != nullptr ? gamma[c] : 1.f) / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = (beta != nullptr ? beta[c] : 0.f) - mean[c] * s + (conv_bias != nullptr ? conv_bias[c] * s : 0.f);
}

// NCHW fp32 image -> NHWC fp16 with the channels padded to 8 (one 16-byte store per pixel)
__global__ void himage_kernel(const float* __restrict__ img, half_t* __restrict__ out, int N, int C, int HW) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pxl = (int)(i % HW), n = (int)(i / HW);
        half8 h;
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = (c < C) ? (half_t)clamp_h(img[((long long)n * C + c) * HW + pxl]) : (half_t)0.f;
        *(half8*)(out + i * 8) = h;
    }
}

// 3x3 stride 2 pad 1 max-pool (torchvision trunk), 8 channels per lane
__global__ void hmaxpool_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int N, int H, int W, int C, int ldx,
                                int ldy, int Ho, int Wo) {
    const int C8 = C / 8;
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        long long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = 2 * ho - 1 + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = 2 * wo - 1 + kw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const half8 h = *(const half8*)(x + (((long long)n * H + hi) * W + wi) * ldx + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)h[e]);
            }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)m[e];
        *(half8*)(y + (((long long)n * Ho + ho) * Wo + wo) * ldy + c) = o;
    }
}

// bilinear x2, align_corners=True (nn.Upsample of models/encoder.py:51), same source-index arithmetic as the fp32 kernel
__device__ __forceinline__ void hbil_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
    const float src = scale * (float)o;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l1 = src - (float)i0;
}
__global__ void hupsample2x_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int N, int H, int W, int C, int ldx,
                                   int ldy) {
    const int Ho = 2 * H, Wo = 2 * W, C8 = C / 8;
    const float sh = (Ho > 1) ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = (Wo > 1) ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        long long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho), n = (int)(t / Ho);
        int h0, h1, w0, w1;
        float lh, lw;
        hbil_src(ho, sh, H, h0, h1, lh);
        hbil_src(wo, sw, W, w0, w1, lw);
        const half_t* b = x + (long long)n * H * W * ldx + c;
        const half8 x00 = *(const half8*)(b + ((long long)h0 * W + w0) * ldx), x01 = *(const half8*)(b + ((long long)h0 * W + w1) * ldx);
        const half8 x10 = *(const half8*)(b + ((long long)h1 * W + w0) * ldx), x11 = *(const half8*)(b + ((long long)h1 * W + w1) * ldx);
        const float hl0 = 1.f - lh, wl0 = 1.f - lw;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = (half_t)(hl0 * (wl0 * (float)x00[e] + lw * (float)x01[e]) + lh * (wl0 * (float)x10[e] + lw * (float)x11[e]));
        *(half8*)(y + (((long long)n * Ho + ho) * Wo + wo) * ldy + c) = o;
    }
}

// global average pool of an NHWC fp16 map -> fp32 [N][C]: one wavefront per (image, 64-channel group) strides the pixels
__global__ void havgpool_kernel(const half_t* __restrict__ x, float* __restrict__ y, int N, int HW, int C, int ldx) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), n = blockIdx.y;
    const int part = threadIdx.x >> 6;                      // 4 pixel partitions
    __shared__ float red[4][64];
    float s = 0.f;
    if (c < C)
        for (int p = part; p < HW; p += 4) s += (float)x[((long long)n * HW + p) * ldx + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C) y[(long long)n * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW;
}

inline int hgrid(long long n) { return (int)((n + 255) / 256 < 65535 * 16 ? (n + 255) / 256 : 65535 * 16); }

// ------------------------------------------------------------------------------------------------ halo-resident 3x3 (round 5)
// hconv3_halo_kernel: the stride-1 / pad-1 3x3 convolutions of the fp16 backbone with the input halo resident in LDS -- the
// fp16 sibling of csrc/rih_conv3.hip.  A workgroup (512 threads, 8 wavefronts as 4 x 2, wave tile 64 x BN/2) owns a patch of 256
// pixels of one image (8 x 32, or 16 x 16 where the width is a multiple of 16 only) and BN = 128 / 64 output channels; per
// 64-channel chunk the (rows + 2) x (width + 2) halo (128 bytes per pixel) is fetched ONCE by LDS-DMA and the nine taps run on
// shifted windows of it -- the implicit GEMM above fetches every pixel nine times per chunk (its 3x3 launches sat at 0.23-0.36 of
// their rooflines, LDS-DMA bound at an occupancy of two: profiles/r04/c11_hconv_sweep.log).  Weights: the same [Cout][Kpad] fp16
// operand (k = (tap, ci)), one 128-byte row segment per output channel and k-tile, LDS-DMA, double-buffered; one barrier per
// k-tile.  LDS images: unit j (8 halves) of a pixel at position j ^ ((halo column >> 1) & 7), of a weight row at
// j ^ ((row >> 1) & 7), applied on the SOURCE address (the LDS-DMA destination is lane-linear): conflict-free ds_read_b128
// operand fetches for every tap shift (tests/test_kernels_on_cpu.py::test_conv3_lds_image_is_conflict_free).
// Epilogue as hconv_kernel (bias, ReLU, post scale / shift, fp16 or fp32 output); no residual (no 3x3 convolution has one).
// Preconditions (hconv3_ok): KH = KW = 3, stride 1, pad 1, Cin % 64 == 0, Cout % 64 == 0, (H % 8 == 0 and W % 32 == 0) or
// (H % 16 == 0 and W % 16 == 0), no residual, 16-byte aligned output rows.
constexpr int H3_NT = 512;
constexpr int H3_ASTAGE = 2752 * 16;        // 340 pixels x 8 units, rounded up to whole wave instructions (43 x 64 units)

template <int TW, int BN>
__global__ __launch_bounds__(H3_NT, 2) void hconv3_halo_kernel(const HConvArgs p) {
    constexpr int TH = 256 / TW, HWP = TW + 2, HP = (TH + 2) * HWP;
    constexpr int TN = BN / 64;
    constexpr int B_STAGE = BN * ROWB;
    constexpr int NPA = (HP * 8 + H3_NT - 1) / H3_NT;       // 6: the last instruction of the last issuing wave lands in the padding
    constexpr int NPB = (BN * 8) / H3_NT;                   // 2 / 1
    static_assert(((HP * 8 + 63) / 64) * 64 * 16 <= H3_ASTAGE && (BN * 8) % H3_NT == 0, "stage sizes");
    constexpr int SMEM = 2 * H3_ASTAGE + 2 * B_STAGE;
    static_assert(SMEM >= 8 * 32 * STAGE_LD * 4, "epilogue staging fits");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * H3_ASTAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_x = p.W / TW, tiles_y = p.H / TH, nblk = p.Cout / BN;
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int nb = bid % nblk;
    int rest = bid / nblk;
    const int tx_t = rest % tiles_x;
    rest /= tiles_x;
    const int ty_t = rest % tiles_y;
    const int img = rest / tiles_y;
    const int y0 = ty_t * TH, x0 = tx_t * TW, n0 = nb * BN;
    const int lty = (TW == 32) ? 0 : (l31 >> 4), ltx = l31 & (TW - 1);

    // A: LDS unit U = pass * NT + tid (halo pixel hp = U / 8, position U % 8) holds source unit j = pos ^ ((hx >> 1) & 7)
    const half_t* a_src[NPA];
    bool a_on[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int U = i * H3_NT + tid;
        const int hp = U >> 3, pos = U & 7;
        a_src[i] = p.zero;
        a_on[i] = (i * H3_NT + (tid & ~63)) < HP * 8;       // wave-uniform: this wave's instruction starts inside the halo
        if (hp < HP) {
            const int hy = hp / HWP, hx = hp - hy * HWP;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                a_src[i] = p.x + ((long long)(img * p.H + y) * p.W + x) * p.ldx + 8 * (pos ^ ((hx >> 1) & 7));
        }
    }
    const half_t* b_src[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int U = i * H3_NT + tid;
        const int n = U >> 3, j = (U & 7) ^ ((n >> 1) & 7);
        b_src[i] = p.w + (long long)(n0 + n) * p.Kpad + 8 * j;
    }
    auto issue_A = [&](int c0, unsigned char* dst) {        // the halo of channels [c0, c0 + 64); padding lanes read the zero page
#pragma unroll
        for (int i = 0; i < NPA; ++i)
            if (a_on[i]) glds16(pick(a_src[i] != p.zero, a_src[i] + c0, p.zero), dst + (i * H3_NT + tid) * 16);
    };
    auto issue_B = [&](int kofs, unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < NPB; ++i) glds16(b_src[i] + kofs, dst + (i * H3_NT + tid) * 16);
    };
    int b_rd[4][TN];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            const int n = wn * (BN / 2) + jj * 32 + l31;
            b_rd[s][jj] = n * ROWB + (((2 * s + lhi) ^ ((n >> 1) & 7)) << 4);
        }
    constexpr int BLKROWS = 32 / TW;
    const int a_hp0 = (wm * 2 * BLKROWS + lty) * HWP + ltx;

    floatx16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // epilogue constants, requested before the k-loop (see hconv_kernel): lane -> 8 output channels of a 32-column block
    const int ech = lane & 3, erow = lane >> 2;
    const int en0 = n0 + wn * (BN / 2) + 8 * ech;
    float bia[TN][8], psc[TN][8], psh[TN][8];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = en0 + j * 32;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bia[j][e] = p.bias != nullptr ? p.bias[n + e] : 0.f;
            psc[j][e] = p.post_scale != nullptr ? p.post_scale[n + e] : 1.f;
            psh[j][e] = p.post_scale != nullptr ? p.post_shift[n + e] : 0.f;
        }
    }

    const int nchunk = p.Cin / 64, ntile = nchunk * 9;
    issue_A(0, Abuf);
    issue_B(0, Bbuf);
    __syncthreads();
    int kt = 0;
    for (int c = 0; c < nchunk; ++c) {
        const unsigned char* As = Abuf + (c & 1) * H3_ASTAGE;
#pragma unroll 1
        for (int t = 0; t < 9; ++t, ++kt) {
            if (kt + 1 < ntile) {
                const int t1 = (t == 8) ? 0 : t + 1, c1 = (t == 8) ? c + 1 : c;
                issue_B(t1 * p.Cin + c1 * 64, Bbuf + ((kt + 1) & 1) * B_STAGE);
            }
            if (t == 0 && c + 1 < nchunk) issue_A((c + 1) * 64, Abuf + ((c + 1) & 1) * H3_ASTAGE);   // (last read in chunk c - 1)
            const unsigned char* Bs = Bbuf + (kt & 1) * B_STAGE;
            const int kh = t / 3, kw = t - kh * 3;
            const int hp_t = a_hp0 + kh * HWP + kw;
            const int sw = ((ltx + kw) >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                half8 a[2], b[TN];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *(const half8*)(As + (hp_t + i * BLKROWS * HWP) * ROWB + (((2 * s + lhi) ^ sw) << 4));
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) b[jj] = *(const half8*)(Bs + b_rd[s][jj]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[jj], acc[i][jj], 0, 0, 0);
            }
            __syncthreads();        // k-tile kt + 1 (and, at t == 0, the next chunk's halo) has landed; nobody still reads kt's stages
        }
    }

    // ------------------------------------------------------------ epilogue: 32 x 32 blocks through this wave's LDS slice
    float* st = (float*)smem + wave * 32 * STAGE_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = wm * 2 + i;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (i + j > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * lhi) * STAGE_LD + l31] = acc[i][j][r];
            __builtin_amdgcn_wave_barrier();
            const int n = en0 + j * 32;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = erow + 16 * q;                  // row of the 32-row block -> pixel (ty, tx) of the patch
                const int ty = (TW == 32) ? blk : 2 * blk + (row >> 4), tx = row & (TW - 1);
                const float4 v0 = *(const float4*)(st + row * STAGE_LD + 8 * ech), v1 = *(const float4*)(st + row * STAGE_LD + 8 * ech + 4);
                const float t8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float u = t8[e] + bia[j][e];
                    if (p.relu) u = fmaxf(u, 0.f);
                    v[e] = u * psc[j][e] + psh[j][e];
                }
                const long long m = ((long long)img * p.H + y0 + ty) * p.W + x0 + tx;
                if (p.out_f32) {
                    float* yp = (float*)p.y + m * p.ldy + n;
                    *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    half8 h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)clamp_h(v[e]);
                    *(half8*)((half_t*)p.y + m * p.ldy + n) = h;
                }
            }
        }
    }
}

// 0: not a shape of hconv3_halo_kernel; else the patch width (32 / 16)
inline int hconv3_tw(const rih_hconv_desc* d) {
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->res != nullptr) return 0;
    if (d->Cin % 64 != 0 || d->Cout % 64 != 0 || d->Kpad < 9 * d->Cin) return 0;
    if (d->ldy % (d->out_f32 ? 4 : 8) != 0 || ((uintptr_t)d->y & 15) != 0) return 0;
    if (d->H % 8 == 0 && d->W % 32 == 0) return 32;
    if (d->H % 16 == 0 && d->W % 16 == 0) return 16;
    return 0;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rih_hconv(const rih_hconv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y || !d->zero) return RIH_EINVAL;
    if (d->N < 1 || d->H < 1 || d->W < 1 || d->Cin < 8 || d->Cin % 8 != 0 || d->Cout < 1 || d->KH < 1 || d->KW < 1) return RIH_EINVAL;
    if (d->stride < 1 || d->pad < 0 || d->Ho < 1 || d->Wo < 1) return RIH_EINVAL;
    if (d->Ho != (d->H + 2 * d->pad - d->KH) / d->stride + 1 || d->Wo != (d->W + 2 * d->pad - d->KW) / d->stride + 1) return RIH_EINVAL;
    const long long K = (long long)d->KH * d->KW * d->Cin;
    if (d->Kpad % BKH != 0 || d->Kpad < K) return RIH_EINVAL;
    if (d->ldx < d->Cin || d->ldx % 8 != 0 || d->ldy < d->Cout || !al16(d->x) || !al16(d->w) || !al16(d->zero)) return RIH_EINVAL;
    if (d->res != nullptr && d->ldr < d->Cout) return RIH_EINVAL;
    if ((d->post_scale == nullptr) != (d->post_shift == nullptr)) return RIH_EINVAL;
    const long long M = (long long)d->N * d->Ho * d->Wo;
    if (M > 0x7fffffffLL || (long long)d->N * d->H * d->W > 0x7fffffffLL) return RIH_EINVAL;
    HConvArgs a;
    a.x = (const half_t*)d->x; a.w = (const half_t*)d->w; a.zero = (const half_t*)d->zero;
    a.bias = d->bias; a.post_scale = d->post_scale; a.post_shift = d->post_shift;
    a.res = (const half_t*)d->res; a.y = d->y;
    a.M = (int)M; a.Cout = d->Cout; a.Kpad = d->Kpad;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.Ho = d->Ho; a.Wo = d->Wo; a.ldx = d->ldx; a.ldr = d->ldr; a.ldy = d->ldy; a.relu = d->relu; a.out_f32 = d->out_f32;
    // 16-byte epilogue accesses need aligned rows on every tensor they touch
    a.vec = al16(d->y) && d->ldy % (d->out_f32 ? 4 : 8) == 0 && (d->res == nullptr || (al16(d->res) && d->ldr % 8 == 0));
    a.cvec = (d->bias == nullptr || al16(d->bias)) && (d->post_scale == nullptr || (al16(d->post_scale) && al16(d->post_shift)));
    hipStream_t s = (hipStream_t)stream;
    // stride-1 3x3 convolutions on whole patches: the halo-resident kernel (RIH_HCONV_HALO=0: the implicit GEMM below)
    static const bool halo = [] { const char* e = getenv("RIH_HCONV_HALO"); return !(e && e[0] == '0'); }();
    const int tw = halo ? hconv3_tw(d) : 0;
    if (tw != 0) {
        const long long patches = (long long)d->N * (d->H / (256 / tw)) * (d->W / tw);
        const int bn = (d->Cout % 128 == 0 && patches * (d->Cout / 128) >= 256) ? 128 : 64;
        const long long grid = patches * (d->Cout / bn);
        if (grid > 0x7fffffffLL) return RIH_EINVAL;
        if (tw == 32 && bn == 128) hipLaunchKernelGGL((hconv3_halo_kernel<32, 128>), dim3((unsigned)grid), dim3(H3_NT), 0, s, a);
        else if (tw == 32) hipLaunchKernelGGL((hconv3_halo_kernel<32, 64>), dim3((unsigned)grid), dim3(H3_NT), 0, s, a);
        else if (bn == 128) hipLaunchKernelGGL((hconv3_halo_kernel<16, 128>), dim3((unsigned)grid), dim3(H3_NT), 0, s, a);
        else hipLaunchKernelGGL((hconv3_halo_kernel<16, 64>), dim3((unsigned)grid), dim3(H3_NT), 0, s, a);
        return (int)hipGetLastError();
    }
    const long long tilesM = (M + BM - 1) / BM;
    static const bool glds = [] { const char* e = getenv("RIH_HCONV_GLDS"); return !(e && e[0] == '0'); }();
    // 128x64 tiles (three workgroups per CU instead of two; A is re-read per 64 columns) also for wide outputs when the reduction
    // is ONE k-tile: measured per shape at B = 256 (profiles/r04/c13_hconv_sweep_*.log) 1x1 64 -> 256 at 64x64 297 -> 272 us with
    // the residual, 279 -> 254 without; everywhere (RIH_HCONV_BN=64) the 3x3 convolutions lose 14-45 % and a forward 8 %.
    // RIH_HCONV_BN=128 switches the rule off.
    static const int force_bn = [] { const char* e = getenv("RIH_HCONV_BN"); return e ? atoi(e) : 0; }();
    const bool narrow = force_bn == 64 || (force_bn != 128 && d->Kpad <= BKH);
    if (d->Cout > 64 && !narrow) {
        const long long tiles = tilesM * ((d->Cout + 127) / 128);
        if (tiles > 0x7fffffffLL) return RIH_EINVAL;
        if (glds) hipLaunchKernelGGL((hconv_kernel<128, true>), dim3((unsigned)tiles), dim3(TPB), 0, s, a);
        else hipLaunchKernelGGL((hconv_kernel<128, false>), dim3((unsigned)tiles), dim3(TPB), 0, s, a);
    } else {
        const long long tiles = tilesM * ((d->Cout + 63) / 64);
        if (tiles > 0x7fffffffLL) return RIH_EINVAL;
        if (glds) hipLaunchKernelGGL((hconv_kernel<64, true>), dim3((unsigned)tiles), dim3(TPB), 0, s, a);
        else hipLaunchKernelGGL((hconv_kernel<64, false>), dim3((unsigned)tiles), dim3(TPB), 0, s, a);
    }
    return (int)hipGetLastError();
}

extern "C" int rih_hpack_conv_weight(const float* w, const float* scale, void* dst, int Cout, int Cin, int KH, int KW,
                                     int CinPad, int Kpad, void* stream) {
    if (!w || !dst || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || CinPad < Cin || CinPad % 8 != 0) return RIH_EINVAL;
    if (Kpad % BKH != 0 || Kpad < KH * KW * CinPad) return RIH_EINVAL;
    hipLaunchKernelGGL(hpack_weight_kernel, dim3(hgrid((long long)Cout * Kpad)), dim3(256), 0, (hipStream_t)stream, w, scale,
                       (half_t*)dst, Cout, Cin, KH, KW, CinPad, Kpad);
    return (int)hipGetLastError();
}

extern "C" int rih_hbn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                            const float* conv_bias, float eps, float* scale, float* shift, int C, void* stream) {
    if (!mean || !var || !scale || !shift || C < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(hbn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var,
                       conv_bias, eps, scale, shift, C);
    return (int)hipGetLastError();
}

extern "C" int rih_himage_nchw_to_nhwc8(const float* img, void* out, int N, int C, int H, int W, void* stream) {
    if (!img || !out || N < 1 || C < 1 || C > 8 || H < 1 || W < 1 || !al16(out)) return RIH_EINVAL;
    hipLaunchKernelGGL(himage_kernel, dim3(hgrid((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, img, (half_t*)out,
                       N, C, H * W);
    return (int)hipGetLastError();
}

extern "C" int rih_hmaxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int ldx, int ldy, void* stream) {
    if (!x || !y || N < 1 || H < 1 || W < 1 || C < 8 || C % 8 != 0 || ldx < C || ldy < C || ldx % 8 != 0 || ldy % 8 != 0 ||
        !al16(x) || !al16(y))
        return RIH_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(hmaxpool_kernel, dim3(hgrid((long long)N * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, N, H, W, C, ldx, ldy, Ho, Wo);
    return (int)hipGetLastError();
}

extern "C" int rih_hupsample2x(const void* x, void* y, int N, int H, int W, int C, int ldx, int ldy, void* stream) {
    if (!x || !y || N < 1 || H < 1 || W < 1 || C < 8 || C % 8 != 0 || ldx < C || ldy < C || ldx % 8 != 0 || ldy % 8 != 0 ||
        !al16(x) || !al16(y))
        return RIH_EINVAL;
    hipLaunchKernelGGL(hupsample2x_kernel, dim3(hgrid((long long)N * 4 * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, N, H, W, C, ldx, ldy);
    return (int)hipGetLastError();
}

extern "C" int rih_havgpool(const void* x, float* y, int N, int HW, int C, int ldx, void* stream) {
    if (!x || !y || N < 1 || N > 65535 || HW < 1 || C < 1 || ldx < C) return RIH_EINVAL;
    hipLaunchKernelGGL(havgpool_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, y, N, HW,
                       C, ldx);
    return (int)hipGetLastError();
}
