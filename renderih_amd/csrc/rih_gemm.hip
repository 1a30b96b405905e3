// rih_gemm.hip -- fp32 MFMA GEMM / implicit-GEMM convolution for gfx950 (MI355X, CDNA4).
//
// One LDS-tiled kernel family on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TF peak) covers the dense
// contractions of the RenderIH pose network: conv forward, conv data-gradient, conv/linear
// weight-gradient (split-K), nn.Linear, and the attention GEMMs (see include/renderih_amd.h).
//
// Block = 256 threads = 4 wavefronts (64 lanes) in a 2x2 grid; block tile BM x BN x 32, each wave owns
// (BM/2)x(BN/2) as 32x32 MFMA tiles.  A and B tiles are staged k-major in LDS ([k][m], [k][n]) so an
// MFMA operand fetch is one conflict-free ds_read_b32 per lane (lanes 0-31 -> consecutive m, lanes
// 32-63 -> next k).  Global->register prefetch of tile t+1 is issued before the MFMAs of tile t.
// blockIdx.x is remapped so that each XCD (private L2) works on a contiguous run of tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* R;
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int nb2, splitk, kchunk;
    long long sA1, sA2, sB1, sB2, sC1, sC2, sCsplit;
    float alpha;
    int relu;
    int H, W, Cin, Ho, Wo, KH, KW, strideA, upS, padH, padW;
    int vecA, vecB;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "each XCD gets a contiguous chunk" remap (blocks are dispatched round-robin over 8 XCDs)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int LDAS = BM + 4;
    constexpr int LDBS = BN + 4;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NPA = BM / 32, NPB = BN / 32;

    __shared__ __attribute__((aligned(16))) float smem[BK * (LDAS + LDBS)];
    float* As = smem;
    float* Bs = smem + BK * LDAS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int tilesN = (p.N + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;

    const int z = blockIdx.z;
    const int split = z % p.splitk;
    const int bz = z / p.splitk;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    const float* __restrict__ A = p.A + b1 * p.sA1 + b2 * p.sA2;
    const float* __restrict__ B = p.B + b1 * p.sB1 + b2 * p.sB2;
    float* __restrict__ C = p.C + b1 * p.sC1 + b2 * p.sC2 + split * p.sCsplit;

    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    // ------------------------------------------------------------------ A loader state
    // AMODE 0: thread -> (row = tid/8 (+32 per pass), k-quad = tid%8)
    // AMODE 1: thread -> (k-row = tid/(BM/4) (+256/(BM/4) per pass), m-quad = tid%(BM/4))
    long long a_base[NPA];
    int a_hi0[NPA], a_wi0[NPA];
    int a_kh, a_kw, a_ci;           // AMODE 0: running (kh,kw,ci) of this thread's k-quad; AMODE 1: fixed tap of m-quad
    int a_mvalid = 0;               // AMODE 1: number of valid elements in this thread's m-quad (0..4)
    constexpr int QA = BM / 4, RA = 256 / QA;
    if (AMODE == 0) {
        const int arow = tid >> 3, aq = tid & 7;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = m0 + arow + 32 * i;
            if (m < p.M) {
                const int wo = m % p.Wo;
                const int t = m / p.Wo;
                const int ho = t % p.Ho;
                const int img = t / p.Ho;
                a_base[i] = (long long)img * p.H * p.W * p.lda;
                a_hi0[i] = ho * p.strideA - p.padH;
                a_wi0[i] = wo * p.strideA - p.padW;
            } else {
                a_base[i] = 0;
                a_hi0[i] = -(1 << 28);
                a_wi0[i] = -(1 << 28);
            }
        }
        const int kg = kbeg + aq * 4;
        const int tap = kg / p.Cin;
        a_ci = kg - tap * p.Cin;
        a_kh = tap / p.KW;
        a_kw = tap - a_kh * p.KW;
    } else {
        const int amq = tid % QA;
        const int mm = m0 + amq * 4;
        const int tap = mm / p.Cin;
        a_ci = mm - tap * p.Cin;
        a_kh = tap / p.KW;
        a_kw = tap - a_kh * p.KW;
        a_mvalid = max(0, min(4, p.M - mm));
#pragma unroll
        for (int i = 0; i < NPA; ++i) { a_base[i] = 0; a_hi0[i] = 0; a_wi0[i] = 0; }
    }

    float4 areg[NPA], breg[NPB];

    auto load_A = [&](int ktile) {
        if (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                float4 v = zero4();
                if (a_kh < p.KH) {
                    int hi = a_hi0[i] + a_kh, wi = a_wi0[i] + a_kw;
                    bool ok = (hi >= 0) && (wi >= 0);
                    if (p.upS > 1) {
                        ok = ok && (hi % p.upS == 0) && (wi % p.upS == 0);
                        hi /= p.upS;
                        wi /= p.upS;
                    }
                    ok = ok && (hi < p.H) && (wi < p.W);
                    if (ok) {
                        const float* ptr = A + a_base[i] + ((long long)hi * p.W + wi) * p.lda + a_ci;
                        const int rem = p.Cin - a_ci;      // valid elements of this quad inside the tap
                        if (p.vecA) {
                            v = *reinterpret_cast<const float4*>(ptr);
                            if (rem < 4) {
                                if (rem < 2) v.y = 0.f;
                                if (rem < 3) v.z = 0.f;
                                v.w = 0.f;
                            }
                        } else {
                            v.x = ptr[0];
                            if (rem > 1) v.y = ptr[1];
                            if (rem > 2) v.z = ptr[2];
                            if (rem > 3) v.w = ptr[3];
                        }
                    }
                }
                areg[i] = v;
            }
        } else {
            const int akr = tid / QA;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                float4 v = zero4();
                const int k = ktile + akr + RA * i;
                if (k < kend && a_mvalid > 0 && a_kh < p.KH) {
                    const int wo = k % p.Wo;
                    const int t = k / p.Wo;
                    const int ho = t % p.Ho;
                    const int img = t / p.Ho;
                    const int hi = ho * p.strideA - p.padH + a_kh;
                    const int wi = wo * p.strideA - p.padW + a_kw;
                    if (hi >= 0 && wi >= 0 && hi < p.H && wi < p.W) {
                        const float* ptr = A + ((long long)(img * p.H + hi) * p.W + wi) * p.lda + a_ci;
                        if (p.vecA) {
                            v = *reinterpret_cast<const float4*>(ptr);
                            if (a_mvalid < 4) {
                                if (a_mvalid < 2) v.y = 0.f;
                                if (a_mvalid < 3) v.z = 0.f;
                                v.w = 0.f;
                            }
                        } else {
                            v.x = ptr[0];
                            if (a_mvalid > 1) v.y = ptr[1];
                            if (a_mvalid > 2) v.z = ptr[2];
                            if (a_mvalid > 3) v.w = ptr[3];
                        }
                    }
                }
                areg[i] = v;
            }
        }
    };

    auto advance_A = [&]() {
        if (AMODE == 0) {
            a_ci += BK;
            while (a_ci >= p.Cin && a_kh < p.KH) {
                a_ci -= p.Cin;
                if (++a_kw == p.KW) { a_kw = 0; ++a_kh; }
            }
        }
    };

    auto store_A = [&]() {
        if (AMODE == 0) {
            const int arow = tid >> 3, aq = tid & 7;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                float* dst = As + (aq * 4) * LDAS + arow + 32 * i;
                dst[0] = areg[i].x;
                dst[LDAS] = areg[i].y;
                dst[2 * LDAS] = areg[i].z;
                dst[3 * LDAS] = areg[i].w;
            }
        } else {
            const int amq = tid % QA, akr = tid / QA;
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                *reinterpret_cast<float4*>(As + (akr + RA * i) * LDAS + amq * 4) = areg[i];
        }
    };

    // ------------------------------------------------------------------ B loader
    constexpr int QB = BN / 4, RB = 256 / QB;
    auto load_B = [&](int ktile) {
        if (BMODE == 0) {
            const int bnq = tid % QB, bkr = tid / QB;
            const int n = n0 + bnq * 4;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                float4 v = zero4();
                const int k = ktile + bkr + RB * i;
                if (k < kend && n < p.N) {
                    const float* ptr = B + (long long)k * p.ldb + n;
                    if (p.vecB && n + 3 < p.N) {
                        v = *reinterpret_cast<const float4*>(ptr);
                    } else {
                        v.x = ptr[0];
                        if (n + 1 < p.N) v.y = ptr[1];
                        if (n + 2 < p.N) v.z = ptr[2];
                        if (n + 3 < p.N) v.w = ptr[3];
                    }
                }
                breg[i] = v;
            }
        } else {
            const int brow = tid >> 3, bq = tid & 7;
            const int k = ktile + bq * 4;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                float4 v = zero4();
                const int n = n0 + brow + 32 * i;
                if (n < p.N && k < kend) {
                    const float* ptr = B + (long long)n * p.ldb + k;
                    if (p.vecB && k + 3 < kend) {
                        v = *reinterpret_cast<const float4*>(ptr);
                    } else {
                        v.x = ptr[0];
                        if (k + 1 < kend) v.y = ptr[1];
                        if (k + 2 < kend) v.z = ptr[2];
                        if (k + 3 < kend) v.w = ptr[3];
                    }
                }
                breg[i] = v;
            }
        }
    };

    auto store_B = [&]() {
        if (BMODE == 0) {
            const int bnq = tid % QB, bkr = tid / QB;
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                *reinterpret_cast<float4*>(Bs + (bkr + RB * i) * LDBS + bnq * 4) = breg[i];
        } else {
            const int brow = tid >> 3, bq = tid & 7;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                float* dst = Bs + (bq * 4) * LDBS + brow + 32 * i;
                dst[0] = breg[i].x;
                dst[LDBS] = breg[i].y;
                dst[2 * LDBS] = breg[i].z;
                dst[3 * LDBS] = breg[i].w;
            }
        }
    };

    // ------------------------------------------------------------------ main loop
    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ntiles > 0) {
        load_A(kbeg);
        load_B(kbeg);
        store_A();
        store_B();
    }
    __syncthreads();

    const int l31 = lane & 31, lhi = lane >> 5;
    const float* a_rd = As + lhi * LDAS + wm * WM + l31;
    const float* b_rd = Bs + lhi * LDBS + wn * WN + l31;

    for (int t = 0; t < ntiles; ++t) {
        const bool more = (t + 1 < ntiles);
        if (more) {
            advance_A();
            load_A(kbeg + (t + 1) * BK);
            load_B(kbeg + (t + 1) * BK);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a_rd[(kk * 2) * LDAS + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b_rd[(kk * 2) * LDBS + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_A();
            store_B();
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue
    // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool raw = (p.splitk > 1);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + l31;
            const float bv = (!raw && p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m < p.M && n < p.N) {
                    float v = acc[i][j][r];
                    if (!raw) {
                        v = v * p.alpha + bv;
                        if (p.R != nullptr) v += p.R[(long long)m * p.ldr + n];
                        if (p.relu) v = fmaxf(v, 0.f);
                    }
                    C[(long long)m * p.ldc + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN>
int launch_tile(const GemmArgs& a, int a_mode, int b_mode, dim3 grid, hipStream_t s) {
    dim3 block(256);
    if (a_mode == 0 && b_mode == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 0>), grid, block, 0, s, a);
    else if (a_mode == 0 && b_mode == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 1>), grid, block, 0, s, a);
    else if (a_mode == 1 && b_mode == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 0>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 1>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

__global__ void splitk_reduce_kernel(const float* __restrict__ P, int S, int M, int N, float* __restrict__ dst,
                                     int Cin, int taps, int CinValid, int accumulate) {
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (long long)m * N);
        const int tap = m / Cin, ci = m - tap * Cin;
        if (ci >= CinValid) continue;
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += P[(long long)k * total + i];
        const long long o = ((long long)n * CinValid + ci) * taps + tap;
        dst[o] = accumulate ? dst[o] + s : s;
    }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin,
                                        int KH, int KW, int CinPad, int for_dgrad) {
    const long long total = (long long)KH * KW * CinPad * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int co, ci, tap;
        if (!for_dgrad) {   // dst[(tap*CinPad + ci)*Cout + co]
            co = (int)(i % Cout);
            const long long t = i / Cout;
            ci = (int)(t % CinPad);
            tap = (int)(t / CinPad);
        } else {            // dst[(tap'*Cout + co)*CinPad + ci], tap' = flipped tap
            ci = (int)(i % CinPad);
            const long long t = i / CinPad;
            co = (int)(t % Cout);
            tap = KH * KW - 1 - (int)(t / Cout);
        }
        float v = 0.f;
        if (ci < Cin) v = w[((long long)co * Cin + ci) * (KH * KW) + tap];
        dst[i] = v;
    }
}

}  // namespace

extern "C" int rih_gemm(const rih_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return RIH_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K < 0) return RIH_EINVAL;
    if (d->splitk < 1 || d->nb1 < 1 || d->nb2 < 1) return RIH_EINVAL;
    if (d->splitk > 1 && (d->kchunk <= 0 || d->kchunk % BK != 0)) return RIH_EINVAL;
    if (d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->upS < 1 || d->strideA < 1)
        return RIH_EINVAL;
    if (d->a_mode == 1 && d->upS != 1) return RIH_EINVAL;
    if (d->KH * d->KW > 1 && (d->Cin % 4) != 0) return RIH_EINVAL;   // quads must not straddle taps
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.R = d->R;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr;
    a.nb2 = d->nb2; a.splitk = d->splitk;
    a.kchunk = (d->splitk > 1) ? d->kchunk : ((d->K + BK - 1) / BK) * BK + BK;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.sCsplit = d->sCsplit;
    a.alpha = d->alpha; a.relu = d->relu;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.KH = d->KH; a.KW = d->KW;
    a.strideA = d->strideA; a.upS = d->upS; a.padH = d->padH; a.padW = d->padW;
    const bool a16 = ((uintptr_t)d->A % 16 == 0) && (d->lda % 4 == 0) && (d->sA1 % 4 == 0) && (d->sA2 % 4 == 0);
    const bool b16 = ((uintptr_t)d->B % 16 == 0) && (d->ldb % 4 == 0) && (d->sB1 % 4 == 0) && (d->sB2 % 4 == 0);
    a.vecA = a16 ? 1 : 0;
    a.vecB = b16 ? 1 : 0;
    int bm = 128, bn = 128;
    if (d->tile == 1) { bm = 128; bn = 64; }
    else if (d->tile == 2) { bm = 64; bn = 64; }
    else if (d->tile != 0) return RIH_EINVAL;
    const long long tiles = (long long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn);
    const long long gz = (long long)d->nb1 * d->nb2 * d->splitk;
    if (tiles > 0x7fffffffLL || gz > 65535) return RIH_EINVAL;
    dim3 grid((unsigned)tiles, 1, (unsigned)gz);
    hipStream_t s = (hipStream_t)stream;
    if (d->tile == 0) return launch_tile<128, 128>(a, d->a_mode, d->b_mode, grid, s);
    if (d->tile == 1) return launch_tile<128, 64>(a, d->a_mode, d->b_mode, grid, s);
    return launch_tile<64, 64>(a, d->a_mode, d->b_mode, grid, s);
}

extern "C" int rih_splitk_reduce(const float* P, int S, int M, int N, float* dst, int Cin, int taps, int CinValid,
                                 int accumulate, void* stream) {
    if (!P || !dst || S < 1 || M < 1 || N < 1 || Cin < 1 || taps < 1 || CinValid < 1) return RIH_EINVAL;
    const long long total = (long long)M * N;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, S, M, N, dst, Cin,
                       taps, CinValid, accumulate);
    return (int)hipGetLastError();
}

extern "C" int rih_pack_conv_weight(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int CinPad,
                                    int for_dgrad, void* stream) {
    if (!w || !dst || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || CinPad < Cin) return RIH_EINVAL;
    const long long total = (long long)KH * KW * CinPad * Cout;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin,
                       KH, KW, CinPad, for_dgrad);
    return (int)hipGetLastError();
}
