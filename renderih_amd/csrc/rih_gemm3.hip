// rih_gemm3.hip -- conversion-free split-bf16 GEMM / implicit-GEMM convolution on pre-split "P3" operands (gfx950).
//
// Same arithmetic as the split engine of rih_gemm.hip (fp32 value x = hi + mid + lo as three bf16 terms, product
// = lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-grade
// results, ceiling 2.5 PF / 6 = 417 TF), but the operands ARRIVE split: the producer of an activation (BatchNorm apply /
// backward-apply, rih_p3_from_f32) and the once-per-step weight pass write the P3 format, so this kernel converts
// nothing and stages nothing through registers.  Measured motivation (profiles/r02/presplit_bench_m1.log): removing the
// conversion VALU from the register-staged 128x128 kernels changes nothing (150 TF either way) -- those kernels are
// bound by their two barriers per k-tile and the synchronous convert/ds_write section, not by VALU issue.
//
// P3 format of a row-major matrix [rows][C] (C % 8 == 0, row pitch ld channels, ld % 8 == 0):
//   16-byte unit (row r, channel group g = c/8, plane p in {hi, mid, lo}) = 8 bf16 at byte ((r*ld/8 + g)*3 + p)*16,
//   i.e. the three planes of 8 consecutive channels are 48 contiguous bytes and a 32-channel k-tile of one row is 192
//   contiguous bytes (12 units = three 64-byte lines).  6 bytes per element.
//
// Slab-major variant "P3S" (desc.layout = 1): [C/32 slabs][rows][12 units] -- the 192 bytes of a row's k-tile are followed by
//   the NEXT ROW's 192 bytes of the same k-tile, so the operand stream of a k-tile is contiguous over consecutive pixels and an
//   LDS-DMA wave instruction (64 units = 5.3 rows) fetches whole 128-byte lines.  Measured motivation
//   (profiles/r02/p3_variants_m4.log): with the interleaved layout every request is a 64-byte segment; the same bytes as
//   full-line requests run 198 TF instead of 166 TF.  LDS image row-major [row][12 units], unit j of a row rotated to position
//   (j + ((row >> 2) & 3)) % 12: conflict-free ds_read_b128 (searched exhaustively over the 16-lane read groups).
//
// gemm_p3_kernel: C[m][n] = sum_k A[m][k] B[n][k],  A = P3 activation with im2col gather (k = (kh, kw, c), one tap per
//   k-tile of 32 since the channel count is a multiple of 32), B = P3 weight [N][K].
//   * 8 wavefronts (4 x 2), block tile 256x128 / 128x128 / 128x64, wave tile (TM x TN) 32x32 MFMA blocks.
//   * Operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPRs, no ds_write.  Two LDS stages; the loads of
//     k-tile t+1 are issued right after the barrier that opens k-tile t and have the whole tile (48 MFMAs per wave) to land:
//     ONE barrier per k-tile, nothing synchronous in between.
//   * LDS image of a stage: 64-byte quads; quad (row, jq) (units 4 jq .. 4 jq + 3 of the row) lives at quad slot
//     jq * ROWS + row and its unit u sits at position (u + (row >> 2)) & 3 inside the quad.  An LDS-DMA wave instruction
//     fills 16 quads = 16 rows x 64 contiguous source bytes (whole 64-byte lines on the source side: the rotation only
//     permutes the four lanes of a quad), and a ds_read_b128 operand fetch (one unit of 32 consecutive rows per half
//     wave) touches 16 distinct 16-byte bank groups per 16-lane group: conflict free.
//   * A lane whose unit is padding (conv halo, row >= M, n >= N) reads a 16-byte zero page instead.
//   * epilogue: + bias, + residual, ReLU, strided rows (parity classes of a strided data gradient), and optionally the
//     per-column BatchNorm statistics of the tile (mean and centred sum of squares over the tile's rows, merged later by
//     rih_bn_stats_merge with Chan's formula) -- the separate statistics pass over the activation disappears.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/renderih_amd_experiments.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

struct G3Args {
    const unsigned char* A;
    const unsigned char* B;
    const unsigned char* zero;
    float* C;
    const float* bias;
    const float* R;
    float* stats;
    int M, N, K;
    int lda, ldb, ldc, ldr;
    int H, W, Ho, Wo, KH, KW, stride, padH, padW, Cin;
    int cS, cOH, cOW, cH, cW;
    int relu;
    unsigned a_slab, b_slab;        // P3S: bytes per 32-channel slab of A (pixels * 192) and of B (N * 192)
};

__device__ __forceinline__ int xcd_remap3(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ void glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ok ? a : b on addresses without control flow (hipcc otherwise sinks each LDS-DMA into both arms of a branch)
__device__ __forceinline__ const unsigned char* pick(bool ok, const unsigned char* a, const unsigned char* b) {
    const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
    return (const unsigned char*)(((uintptr_t)a & m) | ((uintptr_t)b & ~m));
}

__device__ __forceinline__ long long c_row3(const G3Args& p, int m) {
    if (p.cS <= 1) return m;
    const int j = m % p.Wo;
    const int t = m / p.Wo;
    const int i = t % p.Ho;
    const int img = t / p.Ho;
    return ((long long)img * p.cH + (i * p.cS + p.cOH)) * p.cW + (j * p.cS + p.cOW);
}

// SLAB = false: interleaved P3 operands, LDS image of 64-byte quads; SLAB = true: slab-major P3S operands, row-major LDS image.
// (The loader ablations of profiles/r02/p3_variants_m4.log -- linear full-line source, no loads, zero page -- were timing
// variants of this kernel at commit 20d9156..; they are not kept in the product.)
template <int WGM, int WGN, int TM, int TN, bool SLAB = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_p3_kernel(const G3Args p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int A_BYTES = BM * 192, B_BYTES = BN * 192, STAGE = A_BYTES + B_BYTES;
    constexpr int QPI = NT / 4;                                   // quads filled by one LDS-DMA instruction of the block
    constexpr int NIA = (3 * BM + QPI - 1) / QPI, NIB = (3 * BN + QPI - 1) / QPI;     // = ceil(12 BM / NT): same in both layouts
    static_assert((3 * BM) % 16 == 0 && (3 * BN) % 16 == 0 && (12 * BM) % 64 == 0 && (12 * BN) % 64 == 0,
                  "a wave instruction must not straddle the A/B regions");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tilesN = (p.N + BN - 1) / BN;
    const int bid = xcd_remap3(blockIdx.x, gridDim.x);
    const int tm = bid / tilesN;
    const int m0 = tm * BM, n0 = (bid % tilesN) * BN;

    // ------------------------------------------------------------ loader state
    // interleaved: instruction i of the A region fills quad Q = i*QPI + tid/4 = (jq, row); this lane supplies position tid&3
    //   of it, i.e. unit j = 4 jq + ((pos - (row>>2)) & 3) of the row's k-tile.
    // slab-major: instruction i fills units U = i*NT + tid of the row-major image: row = U / 12, position U % 12, i.e. unit
    //   j = (position - ((row>>2)&3)) mod 12; 64 consecutive lanes cover 5.3 consecutive rows = one contiguous source range.
    const int pos = tid & 3;
    unsigned a_off[NIA], a_val[NIA];
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
        int row, j;
        bool live;
        if (SLAB) {
            const int U = i * NT + tid;
            row = U / 12;
            j = (U - row * 12 - ((row >> 2) & 3) + 12) % 12;
            live = row < BM;
        } else {
            const int Q = i * QPI + (tid >> 2);
            row = Q % BM;
            j = 4 * (Q / BM) + ((pos - (row >> 2)) & 3);
            live = Q < 3 * BM;
        }
        const int m = m0 + row;
        a_off[i] = 0;
        a_val[i] = 0;
        if (live && m < p.M) {
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int img = t / p.Ho;
            const int hi0 = ho * p.stride - p.padH, wi0 = wo * p.stride - p.padW;
            if (SLAB) a_off[i] = (unsigned)(((img * p.H + hi0) * p.W + wi0) * 192 + j * 16);
            else a_off[i] = (unsigned)(((img * p.H + hi0) * p.W + wi0) * p.lda * 6 + j * 16);      // wraps for hi0/wi0 < 0
            unsigned bits = 0;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W)
                        bits |= 1u << (kh * p.KW + kw);
            a_val[i] = bits;
        }
    }
    const unsigned char* b_src[NIB];
    int b_step[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        int row, j;
        bool live;
        if (SLAB) {
            const int U = i * NT + tid;
            row = U / 12;
            j = (U - row * 12 - ((row >> 2) & 3) + 12) % 12;
            live = row < BN;
        } else {
            const int Q = i * QPI + (tid >> 2);
            row = Q % BN;
            j = 4 * (Q / BN) + ((pos - (row >> 2)) & 3);
            live = Q < 3 * BN;
        }
        const int n = n0 + row;
        const bool ok = live && n < p.N;
        if (SLAB) {
            b_src[i] = ok ? p.B + ((long long)n * 192 + j * 16) : p.zero;
            b_step[i] = ok ? (int)p.b_slab : 0;
        } else {
            b_src[i] = ok ? p.B + ((long long)n * p.ldb * 6 + j * 16) : p.zero;
            b_step[i] = ok ? 192 : 0;
        }
    }
    // wave-uniform walk over (tap, channel): one tap per k-tile (Cin % 32 == 0)
    int u_tap = 0, u_ci = 0, u_kh = 0, u_kw = 0;
    const int ntiles = p.K / 32;

    auto issue = [&](int buf) {                                  // LDS-DMA of the next k-tile into stage `buf`
        unsigned char* dst = smem + buf * STAGE + tid * 16;
        const unsigned tapoff = SLAB ? (unsigned)((u_kh * p.W + u_kw) * 192) + (unsigned)(u_ci >> 5) * p.a_slab
                                     : (unsigned)(((u_kh * p.W + u_kw) * p.lda + u_ci) * 6);
        const unsigned tapbit = 1u << (u_tap & 31);
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            // the last instruction of a region may be partial; the test is wave-uniform in both layouts
            const bool part = SLAB ? ((i + 1) * NT > 12 * BM) : ((i + 1) * QPI > 3 * BM);
            if (!part || (SLAB ? (i * NT + tid) < 12 * BM : (i * QPI + (tid >> 2)) < 3 * BM)) {
                const bool ok = (a_val[i] & tapbit) != 0u;
                glds16(pick(ok, p.A + (a_off[i] + tapoff), p.zero), dst + i * NT * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const bool part = SLAB ? ((i + 1) * NT > 12 * BN) : ((i + 1) * QPI > 3 * BN);
            if (!part || (SLAB ? (i * NT + tid) < 12 * BN : (i * QPI + (tid >> 2)) < 3 * BN)) {
                glds16(b_src[i], dst + A_BYTES + i * NT * 16);
                b_src[i] += b_step[i];
            }
        }
        u_ci += 32;
        if (u_ci >= p.Cin) {
            u_ci = 0;
            ++u_tap;
            if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
        }
    };

    // operand fetch offsets: lane (l31, lhi) reads unit j = (2 s + lhi) * 3 + plane of rows (wave base + 32 i + l31)
    int a_rd[6], b_rd[6];
    {
        const int ra = wm * TM * 32 + l31, rb = wn * TN * 32 + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int j = (2 * s + lhi) * 3 + pl;
                if (SLAB) {
                    a_rd[s * 3 + pl] = ra * 192 + ((j + ((ra >> 2) & 3)) % 12) * 16;
                    b_rd[s * 3 + pl] = A_BYTES + rb * 192 + ((j + ((rb >> 2) & 3)) % 12) * 16;
                } else {
                    a_rd[s * 3 + pl] = ((j >> 2) * BM + ra) * 64 + (((j & 3) + (ra >> 2)) & 3) * 16;
                    b_rd[s * 3 + pl] = A_BYTES + ((j >> 2) * BN + rb) * 64 + (((j & 3) + (rb >> 2)) & 3) * 16;
                }
            }
    }
    constexpr int RBLK = SLAB ? 32 * 192 : 2048;           // byte distance of the next 32-row block of a wave tile

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ntiles > 0) issue(0);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                                    // k-tile t has landed; nobody still reads the other stage
        if (t + 1 < ntiles) issue((t + 1) & 1);             // lands while k-tile t is multiplied
        const unsigned char* st = smem + (t & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    av[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + a_rd[s * 3 + pl] + i * RBLK));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bv[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + b_rd[s * 3 + pl] + j * RBLK));
            }
            // smallest terms first; consecutive MFMAs rotate over the TM x TN accumulators
#define RIH_P3_TERM(PA_, PB_)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA_][i], bv[PB_][j], acc[i][j], 0, 0, 0);
            RIH_P3_TERM(2, 0)
            RIH_P3_TERM(0, 2)
            RIH_P3_TERM(1, 1)
            RIH_P3_TERM(1, 0)
            RIH_P3_TERM(0, 1)
            RIH_P3_TERM(0, 0)
#undef RIH_P3_TERM
        }
    }

    // ------------------------------------------------------------ epilogue
    // C/D layout of v_mfma_f32_32x32x16_bf16: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
            const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                rv[r] = (p.R != nullptr && m < p.M && n < p.N) ? p.R[(long long)m * p.ldr + n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc[i][j][r] + bv + rv[r];
                if (p.relu) v = fmaxf(v, 0.f);
                acc[i][j][r] = v;
                if (m < p.M && n < p.N) p.C[c_row3(p, m) * p.ldc + n] = v;
            }
        }
    }
    if (p.stats != nullptr) {
        // per-column statistics of this tile's BM rows (the host guarantees M % BM == 0): wave level first (mean, then the
        // centred sum of squares -- no cancellation), then Chan's merge of the WGM waves that share the columns
        __syncthreads();                                    // the operand stages are free now
        float* sm = reinterpret_cast<float*>(smem);         // [WGM][BN][2]
        constexpr float inv_rows = 1.f / (TM * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) s1 += acc[i][j][r];
            s1 += __shfl_xor(s1, 32);
            const float mean = s1 * inv_rows;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[i][j][r] - mean;
                    s2 += d * d;
                }
            s2 += __shfl_xor(s2, 32);
            if (lhi == 0) {
                const int c = (wn * TN + j) * 32 + l31;
                sm[(wm * BN + c) * 2] = mean;
                sm[(wm * BN + c) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float mean = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) mean += sm[(w * BN + tid) * 2];
            mean *= 1.f / WGM;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) {
                const float d = sm[(w * BN + tid) * 2] - mean;
                m2 += sm[(w * BN + tid) * 2 + 1] + (float)(TM * 32) * d * d;
            }
            float* o = p.stats + ((long long)tm * p.N + n0 + tid) * 2;
            o[0] = mean;
            o[1] = m2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ P3 producers
__device__ __forceinline__ unsigned pk_bf16_3(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};        // v_cvt_pk_bf16_f32 (RNE); a in the low half
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2_3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16_3(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact in fp32
    m = pk_bf16_3(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = pk_bf16_3(ra, rb);
}
// eight fp32 values -> the 48 bytes (hi | mid | lo) of one channel group
__device__ __forceinline__ void store_p3_group(unsigned char* dst, const float* v) {
    uint4 h, m, l;
    split2_3(v[0], v[1], h.x, m.x, l.x);
    split2_3(v[2], v[3], h.y, m.y, l.y);
    split2_3(v[4], v[5], h.z, m.z, l.z);
    split2_3(v[6], v[7], h.w, m.w, l.w);
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = h;
    d[1] = m;
    d[2] = l;
}

__global__ void p3_from_f32_kernel(const float* __restrict__ x, long long rows, int C, int ldx, unsigned char* __restrict__ out,
                                   int ldo) {
    const int G = C / 8;
    const long long total = rows * G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / G;
        const int g = (int)(i - r * G);
        const float4 v0 = *reinterpret_cast<const float4*>(x + r * ldx + 8 * g);
        const float4 v1 = *reinterpret_cast<const float4*>(x + r * ldx + 8 * g + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        store_p3_group(out + (r * (ldo / 8) + g) * 48, v);
    }
}

// slab-major P3S: out[((c/32) * rows + r) * 12 + ((c%32)/8) * 3 + plane] (16-byte units); thread = (slab, row, channel group)
__global__ void p3s_from_f32_kernel(const float* __restrict__ x, long long rows, int C, int ldx, unsigned char* __restrict__ out) {
    const long long total = rows * (C / 8);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kg = (int)(i & 3);
        const long long t = i >> 2;
        const long long r = t % rows;
        const int slab = (int)(t / rows);
        const float* px = x + r * ldx + slab * 32 + kg * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(px);
        const float4 v1 = *reinterpret_cast<const float4*>(px + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        store_p3_group(out + (((long long)slab * rows + r) * 12 + kg * 3) * 16, v);
    }
}

// OIHW conv weight -> P3 [N][Kpad]:
//   for_dgrad 0: n = co, k = (tap, ci < CinPad)                       (forward operand)
//   for_dgrad 1: n = ci < CinPad, k = ((th, tw), co), taps flipped, subset (kh0 + step*t, kw0 + step*t')  (data gradient)
struct P3WArgs {
    const float* w;
    unsigned char* dst;
    int N, K, Kpad, for_dgrad;
    int Cout, Cin, KH, KW, CinPad, kh0, kw0, step, Th, Tw;
};
__device__ __forceinline__ float p3w_fetch(const P3WArgs& a, int n, int k) {
    if (k >= a.K) return 0.f;
    if (!a.for_dgrad) {
        const int tap = k / a.CinPad, ci = k - tap * a.CinPad;
        return ci < a.Cin ? a.w[((long long)n * a.Cin + ci) * (a.KH * a.KW) + tap] : 0.f;
    }
    const int co = k % a.Cout, t = k / a.Cout;
    const int tw = t % a.Tw, th = t / a.Tw;
    const int kh = a.kh0 + a.step * (a.Th - 1 - th), kw = a.kw0 + a.step * (a.Tw - 1 - tw);
    return n < a.Cin ? a.w[(((long long)co * a.Cin + n) * a.KH + kh) * a.KW + kw] : 0.f;
}
__global__ void p3_weight_kernel(const P3WArgs a, int slab_major) {
    const int G = a.Kpad / 8;
    const long long total = (long long)a.N * G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i / G), g = (int)(i - (long long)n * G);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p3w_fetch(a, n, 8 * g + e);
        // interleaved: [n][k/8][plane]; slab-major: [k/32][n][(k%32)/8][plane]
        const long long unit = slab_major ? (((long long)(g >> 2) * a.N + n) * 4 + (g & 3)) : i;
        store_p3_group(a.dst + unit * 48, v);
    }
}

// Chan merge of the per-tile statistics written by gemm_p3_kernel: one wavefront per channel.
// part [T][C][2] (mean, centred sum of squares over `rows_per_tile` rows) -> mean[C], var[C] (biased).
__global__ void bn_stats_merge_kernel(const float* __restrict__ part, int T, int C, int rows_per_tile, float* __restrict__ mean_out,
                                      float* __restrict__ var_out) {
    const int c = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)part[((long long)t * C + c) * 2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean = s / T;
    double m2 = 0.0;
    for (int t = lane; t < T; t += 64) {
        const double d = (double)part[((long long)t * C + c) * 2] - mean;
        m2 += (double)part[((long long)t * C + c) * 2 + 1] + (double)rows_per_tile * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if (lane == 0) {
        mean_out[c] = (float)mean;
        var_out[c] = (float)(m2 / ((double)T * rows_per_tile));
    }
}

// the same merge with nn.BatchNorm2d's finalisation (bn_stats_final_kernel of rih_elem.hip)
__global__ void bn_stats_from_tiles_kernel(const float* __restrict__ part, int T, int C, int rows_per_tile, float eps,
                                           float momentum, float* __restrict__ mean_out, float* __restrict__ invstd,
                                           float* __restrict__ rmean, float* __restrict__ rvar) {
    const int c = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int t = lane; t < T; t += 64) s += (double)part[((long long)t * C + c) * 2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean = s / T;
    double m2 = 0.0;
    for (int t = lane; t < T; t += 64) {
        const double d = (double)part[((long long)t * C + c) * 2] - mean;
        m2 += (double)part[((long long)t * C + c) * 2 + 1] + (double)rows_per_tile * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if (lane != 0) return;
    const double n = (double)T * rows_per_tile;
    const double var = m2 / n;
    mean_out[c] = (float)mean;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean != nullptr) {
        const double unb = (n > 1.0) ? var * n / (n - 1.0) : var;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * mean);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

template <int WGM, int WGN, int TM, int TN, bool SLAB = false>
int launch_p3(const G3Args& a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    if (tiles > 0x7fffffffLL) return RIH_EINVAL;
    hipLaunchKernelGGL((gemm_p3_kernel<WGM, WGN, TM, TN, SLAB>), dim3((unsigned)tiles), dim3(64 * WGM * WGN), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int rih_gemm_p3_tile_rows(int tile) { return tile == 0 ? 256 : (tile == 1 || tile == 2) ? 128 : -1; }

extern "C" int rih_gemm_p3(const rih_gemm_p3_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C || !d->zero) return RIH_EINVAL;
    if (d->M < 1 || d->N < 1 || d->K < 32 || d->K % 32 != 0 || d->Cin < 32 || d->Cin % 32 != 0) return RIH_EINVAL;
    if (d->lda % 8 != 0 || d->lda < d->Cin || d->ldb % 8 != 0 || d->ldb < d->K || d->ldc < 1) return RIH_EINVAL;
    if (d->KH < 1 || d->KW < 1 || d->KH * d->KW > 32 || d->K != d->KH * d->KW * d->Cin) return RIH_EINVAL;
    if (d->H < 1 || d->W < 1 || d->Ho < 1 || d->Wo < 1 || d->stride < 1 || d->M % (d->Ho * d->Wo) != 0) return RIH_EINVAL;
    if (((uintptr_t)d->A | (uintptr_t)d->B | (uintptr_t)d->zero) % 16 != 0) return RIH_EINVAL;
    const long long imgs = d->M / ((long long)d->Ho * d->Wo);
    if (imgs * d->H * d->W * (long long)d->lda * 6 >= (1ll << 32)) return RIH_EINVAL;    // 32-bit byte offsets into A
    if (d->R != nullptr && (d->ldr < d->N || d->cS > 1)) return RIH_EINVAL;
    if (d->cS > 1 && (d->cH < 1 || d->cW < 1 || d->cOH < 0 || d->cOW < 0)) return RIH_EINVAL;
    const int bm = rih_gemm_p3_tile_rows(d->tile);
    if (bm < 0) return RIH_EINVAL;
    if (d->stats != nullptr && (d->M % bm != 0 || d->cS > 1)) return RIH_EINVAL;
    G3Args a;
    a.A = (const unsigned char*)d->A; a.B = (const unsigned char*)d->B; a.zero = (const unsigned char*)d->zero;
    a.C = d->C; a.bias = d->bias; a.R = d->R; a.stats = d->stats;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr;
    a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.KH = d->KH; a.KW = d->KW; a.stride = d->stride;
    a.padH = d->padH; a.padW = d->padW; a.Cin = d->Cin;
    a.cS = d->cS; a.cOH = d->cOH; a.cOW = d->cOW; a.cH = d->cH; a.cW = d->cW;
    a.relu = d->relu;
    hipStream_t s = (hipStream_t)stream;
    if (d->layout == 1) {           // slab-major P3S operands
        const long long npix = imgs * d->H * d->W;
        if (npix * 192 >= (1ll << 32) || (long long)d->N * 192 >= (1ll << 31)) return RIH_EINVAL;
        a.a_slab = (unsigned)(npix * 192);
        a.b_slab = (unsigned)((long long)d->N * 192);
        if (d->tile == 0) return launch_p3<4, 2, 2, 2, true>(a, s);
        if (d->tile == 1) return launch_p3<4, 2, 1, 2, true>(a, s);
        return launch_p3<4, 2, 1, 1, true>(a, s);
    }
    if (d->layout != 0) return RIH_EINVAL;
    a.a_slab = a.b_slab = 0;
    if (d->tile == 0) return launch_p3<4, 2, 2, 2>(a, s);       // 256 x 128
    if (d->tile == 1) return launch_p3<4, 2, 1, 2>(a, s);       // 128 x 128
    return launch_p3<4, 2, 1, 1>(a, s);                         // 128 x 64
}

extern "C" int rih_p3_from_f32(const float* x, int64_t rows, int C, int ldx, void* out, int ldo, int layout, void* stream) {
    if (!x || !out || rows < 1 || C < 8 || C % 8 != 0 || ldx < C || ldx % 4 != 0 || ldo < C || ldo % 8 != 0) return RIH_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out) % 16 != 0) return RIH_EINVAL;
    const long long total = rows * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (layout == 1) {              // slab-major: whole 32-channel slabs
        if (C % 32 != 0) return RIH_EINVAL;
        hipLaunchKernelGGL(p3s_from_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, C, ldx,
                           (unsigned char*)out);
        return (int)hipGetLastError();
    }
    if (layout != 0) return RIH_EINVAL;
    hipLaunchKernelGGL(p3_from_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)rows, C, ldx,
                       (unsigned char*)out, ldo);
    return (int)hipGetLastError();
}

extern "C" int rih_p3_conv_weight(const float* w, void* dst, int Cout, int Cin, int KH, int KW, int CinPad, int for_dgrad,
                                  int kh0, int kw0, int step, int Th, int Tw, int Kpad, int layout, void* stream) {
    if (layout != 0 && layout != 1) return RIH_EINVAL;
    if (!w || !dst || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || CinPad < Cin || Kpad % 32 != 0) return RIH_EINVAL;
    P3WArgs a = {};
    a.w = w; a.dst = (unsigned char*)dst; a.Kpad = Kpad; a.for_dgrad = for_dgrad ? 1 : 0;
    a.Cout = Cout; a.Cin = Cin; a.KH = KH; a.KW = KW; a.CinPad = CinPad;
    if (!for_dgrad) {
        a.N = Cout; a.K = KH * KW * CinPad;
    } else {
        if (step < 1 || Th < 1 || Tw < 1 || kh0 < 0 || kw0 < 0 || kh0 + step * (Th - 1) >= KH || kw0 + step * (Tw - 1) >= KW)
            return RIH_EINVAL;
        a.N = CinPad; a.K = Th * Tw * Cout;
        a.kh0 = kh0; a.kw0 = kw0; a.step = step; a.Th = Th; a.Tw = Tw;
    }
    if (Kpad < a.K) return RIH_EINVAL;
    const long long total = (long long)a.N * (Kpad / 8);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(p3_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, layout);
    return (int)hipGetLastError();
}

extern "C" int rih_bn_stats_merge(const float* part, int T, int C, int rows_per_tile, float* mean, float* var, void* stream) {
    if (!part || !mean || !var || T < 1 || C < 1 || rows_per_tile < 1) return RIH_EINVAL;
    hipLaunchKernelGGL(bn_stats_merge_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, T, C, rows_per_tile,
                       mean, var);
    return (int)hipGetLastError();
}

extern "C" int rih_bn_stats_from_tiles(const float* part, int T, int C, int rows_per_tile, float eps, float momentum, float* mean,
                                       float* invstd, float* running_mean, float* running_var, void* stream) {
    if (!part || !mean || !invstd || T < 1 || C < 1 || rows_per_tile < 1) return RIH_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return RIH_EINVAL;
    hipLaunchKernelGGL(bn_stats_from_tiles_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, T, C,
                       rows_per_tile, eps, momentum, mean, invstd, running_mean, running_var);
    return (int)hipGetLastError();
}
