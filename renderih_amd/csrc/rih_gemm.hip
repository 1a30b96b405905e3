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
#include <stdlib.h>
#include "../../include/renderih_amd.h"
#include "rih_hash.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

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
    long long sBias1, sR1;          // per-nb1-slice strides of bias / R (paired left/right-hand GEMMs)
    float alpha;
    int relu;
    int H, W, Cin, Ho, Wo, KH, KW, strideA, upS, padH, padW;
    int vecA, vecB;
    unsigned a_bytes, b_bytes;      // extent of one batch slice of A / B (split fast path: buffer range check)
    int cS, cOH, cOW, cH, cW;       // strided output rows (parity classes of a strided-conv data gradient)
    int ones_row;                   // a_mode 1: A(ones_row, k) = 1 for every valid k (bias gradient row); 0 = off
    int epi_vec;                    // C, R, bias and every stride involved are 16-byte aligned: the epilogue may use 16-byte accesses
    float* stats;                   // split fast path, a_mode 0, splitk 1: per wave-row-block column sums [M / WM][2][N] (or NULL)
    // dropout in the epilogue (DROP variants of the split fast path; appended last: the older kernels' kernarg offsets stay):
    // element e of the output (offset from C in floats) is kept iff rih_hash(drop_seed + *drop_seed_dev, e) >= drop_thr
    unsigned drop_thr;              // 0 = off
    float drop_scale;               // 1 / (1 - p)
    unsigned long long drop_seed;
    const unsigned long long* drop_seed_dev;    // device-resident addend of the seed (hipGraph replay), or NULL
    // engine 2 (two-term fp16 split): device-resident upper bounds of |A|, |B| (bound blocks, rih_absmax), or NULL = 1.0
    const float* amax_a;
    const float* amax_b;
    // segmented A (rih_gemm_desc.a_seg, plain a_mode 0 on the fast path): columns [kseg[i-1], kseg[i]) of A come from Aseg[i-1]
    // (pitch ldaseg[i-1], extent aseg_bytes[i-1]); Aseg[0] == NULL: one operand
    const float* Aseg[3];
    int ldaseg[3];
    int kseg[3];
    unsigned aseg_bytes[3];
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective "each XCD gets a contiguous chunk" remap (blocks are dispatched round-robin over 8 XCDs)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void set_elem(float4& v, int e, float x) {
    if (e == 0) v.x = x; else if (e == 1) v.y = x; else if (e == 2) v.z = x; else v.w = x;
}

// Row of C that GEMM row m = (img, i, j) over (Ho, Wo) is stored to.  cS <= 1: row m.  cS > 1: pixel
// (img, i*cS + cOH, j*cS + cOW) of a [*, cH, cW] tensor -- one parity class of a strided convolution's data gradient.
__device__ __forceinline__ long long c_row(const GemmArgs& p, int m) {
    if (p.cS <= 1) return m;
    const int j = m % p.Wo;
    const int t = m / p.Wo;
    const int i = t % p.Ho;
    const int img = t / p.Ho;
    return ((long long)img * p.cH + (i * p.cS + p.cOH)) * p.cW + (j * p.cS + p.cOW);
}

// ---- split engine (ENGINE 1): fp32 operands are split on the way into LDS into three bf16 planes
// x = hi + mid + lo (round-to-nearest at every level, residual <= 2^-24 |x|) and the product is formed on the bf16
// MFMA pipe as hi*hi + (hi*mid + mid*hi) + (mid*mid + hi*lo + lo*hi) with fp32 accumulation: six
// v_mfma_f32_32x32x16_bf16 per 32x32x16 block, the dropped terms are <= 2^-24 relative -- fp32-grade results at
// 2.5 PF / 6 = 417 TF instead of the 157 TF of the native f32 MFMA.
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const bf16x2 v = {(__bf16)a, (__bf16)b};        // v_cvt_pk_bf16_f32 (RNE); a in the low half
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);      // exact in fp32
    m = pk_bf16(ra, rb);
    ra -= __uint_as_float(m << 16);
    rb -= __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(ra, rb);
}
// LDS operand tile of the split engine: per plane [rows][16 dwords] (one row = 32 k as bf16, no padding).  The 16-byte
// chunk c (8 consecutive k) of logical row m lives at physical row m ^ ((m>>4)&1), chunk c ^ ((m>>2)&3): the b128
// operand fetches (one chunk per lane, 32 consecutive rows per half-wave) and the b64 stores of the K-contiguous
// loaders are bank-conflict free; the transposing (K-strided) loaders, whose 16-lane store groups hit 16 different
// row-quads at one k-quad, pay the unavoidable 2-way store conflict (8 distinct bank pairs exist for a fixed k-quad).
__device__ __forceinline__ int lds_row(int m) { return (m ^ ((m >> 4) & 1)) * 16; }
__device__ __forceinline__ int lds_swz(int m) { return (m >> 2) & 3; }

// ---- split engine 2 (ENGINE 2): fp32 operands are scaled by a power of two s (so that s * max|x| lies in [2^14, 2^15), well
// inside the fp16 range) and split on the way into LDS into TWO fp16 planes, hi = fp16(s x) and lo = fp16((s x - hi) * 2^11)
// (round-to-nearest both; |s x - hi - 2^-11 lo| <= 2^-23 |s x|, and lo keeps its 11 bits down to |s x| = 2^-14 * 2^-11 thanks
// to the 2^11 pre-scale -- the error-corrected tensor-core SGEMM scheme of Ootomo & Yokota).  The product is formed with THREE
// v_mfma_f32_32x32x16_f16 per 32x32x16 block: hi*hi into one fp32 accumulator, hi*lo + lo*hi into a second one; the epilogue
// combines acc0 + 2^-11 acc1 and undoes the operand scales (exact: powers of two).  The dropped lo*lo term is <= 2^-22
// relative.  Half the matrix-pipe work (and energy) per fp32 FLOP of the six-product bf16 engine: 2.5 PF / 3 = 833 TF.
// The scale comes from a device-resident upper bound of max|x| (GemmArgs.amax_a / amax_b: a bound block written by the kernel
// that produced the operand, or by rih_absmax): any upper bound is correct, a loose one only costs range at the bottom (full 22-bit
// precision for |x| >= 2^-29 * bound).
__device__ __forceinline__ float e2_scale(const float* amax, bool at_least_one) {
    if (amax == nullptr) return 1.f;
    // the bound block: 64 partial maxima, one per 128-byte line (include/renderih_amd.h: rih_absmax) -- one vector load per
    // wavefront and an xor-shuffle maximum; the result is wave-uniform
    float a = amax[(threadIdx.x & 63) * (RIH_BOUND_FLOATS / 64)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
    a = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a)));
    if (at_least_one) a = fmaxf(a, 1.f);            // the all-ones row of a weight-gradient's A operand must stay in range
    const int e = (int)((__float_as_uint(a) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;             // zero / denormal bound (an all-zero operand), or inf / NaN (garbage either way)
    int se = 268 - e;                               // 2^(14 - (e - 127)), biased
    se = se > 253 ? 253 : se;                       // keep 1/s a normal number
    return __uint_as_float((unsigned)se << 23);
}
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};     // RNE; a in the low half
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split2h(float a, float b, float s, unsigned& h, unsigned& l) {
#if defined(__HIP_DEVICE_COMPILE__)
    // six mixed-precision FMAs per pair (hipcc's own selection for the C form below takes ten): the f16 result of
    // v_fma_mix{lo,hi}_f16 is the RNE conversion of the exact product (a power-of-two scaling), v_fma_mix_f32 reads the f16 half
    // back as an addend, so the residual a*s - hi is one instruction and exact
    float ra, rb;
    const float k2048 = 2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "s"(s), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "s"(s), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(ra), "s"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(rb), "s"(k2048));
#else       /* host build of tests/hipcpu: the same arithmetic in C */
    a *= s;
    b *= s;
    const f16x2 hv = {(_Float16)a, (_Float16)b};
    h = __builtin_bit_cast(unsigned, hv);
    l = pk_f16((a - (float)hv.x) * 2048.f, (b - (float)hv.y) * 2048.f);        // the differences are exact in fp32
#endif
}

// Epilogue shared by the kernels below.  Accumulators -> LDS (one 32x32 block per wave at a time; the operand tiles are dead
// by then) -> each lane owns 4 consecutive columns of a row: one 16-byte residual load and one 16-byte store per lane, 8 lanes
// per 128-byte row segment -- a quarter of the store instructions of the column-per-lane C/D layout.  The store phase of these
// kernels is issue-bound (all CUs write their tiles in lock-step, MI355X_MICROARCH.md "epilogue store tail"): measured on the
// B = 64 step, GEMM family 24.0 -> 22.3 ms (profiles/r02/bench_m14_wide_epilogue.log).
// `stg`: this wave's 32 x SLD floats of LDS; the caller guarantees a __syncthreads() since the last operand read.
constexpr int SLD = 36;             // floats per staged row: 32 + pad, keeps float4 alignment
// STATS: also the per-column (mean, centred sum of squares M2) of the stored values over the wave's TM*32 rows -- shifted sums per
// lane, Chan's pairwise merge across lanes: no E[x^2] - mean^2 cancellation -- written to p.stats[row block][2][N]; the BatchNorm
// that follows the convolution merges the blocks in double (rih_bn_stats_from_blocks).  The training statistics then cost no
// pass over the activation (csrc/rih_elem.hip: bn_stats_partial_kernel reads it once).
// DROP (rih_gemm_desc.drop_p > 0; plain a_mode-0 GEMMs = nn.Linear): v = dropout(act(alpha acc + bias)) + R -- the mask stream
// of rih_add_dropout over the output tensor (element index = offset from desc.C), so the fused form equals
// rih_gemm followed by rih_add_dropout(R, ., p, seed) bit for bit and rih_dropout_bwd re-draws the same mask.
// E2 (engine 2): the staged value is (acc + 2^-11 acc1) * inv_a * inv_b -- the correction accumulator folded in and the operand
// scales undone (also for the raw split-K slabs, whose reduction knows nothing of scales).
template <int TM, int TN, bool STATS = false, bool DROP = false, bool E2 = false>
__device__ __forceinline__ void store_tiles_wide(const GemmArgs& p, floatx16 (&acc)[TM][TN], float* stg, float* __restrict__ C,
                                                 const float* __restrict__ biasp, const float* __restrict__ Rp, int mbase,
                                                 int nbase, int lane, floatx16 (*acc1)[TN] = nullptr, float inv_a = 1.f,
                                                 float inv_b = 1.f) {
    const bool raw = (p.splitk > 1);
    const bool vec = p.epi_vec != 0;
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned long long drop_key = 0ull;
    if (DROP) drop_key = rih_seed_key(p.drop_seed + (p.drop_seed_dev != nullptr ? *p.drop_seed_dev : 0ull));
    // per lane and column block: shift (the lane's first stored row), sums of (v - shift) and of its square, row count
    float4 ssh[STATS ? TN : 1], ssum[STATS ? TN : 1], ssq[STATS ? TN : 1];
    float scnt[STATS ? TN : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) { ssh[j] = zero4(); ssum[j] = zero4(); ssq[j] = zero4(); scnt[j] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // (the staging rows are this wavefront's own: a wave-level ordering point is enough, the four waves of the workgroup
            // store their blocks without waiting for each other -- round 4; before: two workgroup barriers per 32x32 block;
            // same-box A/B 1936.8 / 1937.1 against 1937.2 / 1932.5 images/s: neutral, kept as the simpler form)
            if (i + j > 0) __builtin_amdgcn_wave_barrier();         // the previous block has been read back
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                if (E2) v = fmaf(acc1[i][j][r], 0x1p-11f, v) * inv_a * inv_b;
                stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + l31] = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
                const int m = mbase + i * 32 + row, n = nbase + j * 32 + c4;
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(stg + row * SLD + c4);
                float* crow = C + c_row(p, m) * p.ldc + n;
                const bool full = vec && n + 3 < p.N;
                if (!raw) {
                    float4 b4 = zero4(), r4 = zero4();
                    if (full) {
                        if (biasp != nullptr) b4 = *reinterpret_cast<const float4*>(biasp + n);
                        if (Rp != nullptr) r4 = *reinterpret_cast<const float4*>(Rp + (long long)m * p.ldr + n);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < p.N) {
                                if (biasp != nullptr) set_elem(b4, e, biasp[n + e]);
                                if (Rp != nullptr) set_elem(r4, e, Rp[(long long)m * p.ldr + n + e]);
                            }
                    }
                    if (DROP) {
                        v.x = v.x * p.alpha + b4.x;
                        v.y = v.y * p.alpha + b4.y;
                        v.z = v.z * p.alpha + b4.z;
                        v.w = v.w * p.alpha + b4.w;
                        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        const unsigned long long e0 = (unsigned long long)(crow - p.C);
                        v.x = (rih_hash_k64(drop_key, e0) >= p.drop_thr) ? v.x * p.drop_scale : 0.f;
                        v.y = (rih_hash_k64(drop_key, e0 + 1) >= p.drop_thr) ? v.y * p.drop_scale : 0.f;
                        v.z = (rih_hash_k64(drop_key, e0 + 2) >= p.drop_thr) ? v.z * p.drop_scale : 0.f;
                        v.w = (rih_hash_k64(drop_key, e0 + 3) >= p.drop_thr) ? v.w * p.drop_scale : 0.f;
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    } else {
                    v.x = v.x * p.alpha + b4.x + r4.x;
                    v.y = v.y * p.alpha + b4.y + r4.y;
                    v.z = v.z * p.alpha + b4.z + r4.z;
                    v.w = v.w * p.alpha + b4.w + r4.w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    }
                }
                if (full) {
                    *reinterpret_cast<float4*>(crow) = v;
                } else {
                    crow[0] = v.x;
                    if (n + 1 < p.N) crow[1] = v.y;
                    if (n + 2 < p.N) crow[2] = v.z;
                    if (n + 3 < p.N) crow[3] = v.w;
                }
                if (STATS) {        // (columns past N are never written out below)
                    if (scnt[j] == 0.f) ssh[j] = v;
                    scnt[j] += 1.f;
                    const float dx = v.x - ssh[j].x, dy = v.y - ssh[j].y, dz = v.z - ssh[j].z, dw = v.w - ssh[j].w;
                    ssum[j].x += dx; ssum[j].y += dy; ssum[j].z += dz; ssum[j].w += dw;
                    ssq[j].x += dx * dx; ssq[j].y += dy * dy; ssq[j].z += dz * dz; ssq[j].w += dw * dw;
                }
            }
        }
    }
    if (STATS) {
        // a lane holds (count, mean, centred sum of squares) of its <= 4*TM rows per column; the eight row-lanes (lane >> 3)
        // are merged pairwise with Chan's formula (three xor-shuffle rounds), lane >> 3 == 0 writes the block's (mean, M2)
        const long long rb = mbase / (TM * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float n = scnt[j];
            const float inv = n > 0.f ? 1.f / n : 0.f;
            float4 mean = make_float4(ssh[j].x + ssum[j].x * inv, ssh[j].y + ssum[j].y * inv, ssh[j].z + ssum[j].z * inv,
                                      ssh[j].w + ssum[j].w * inv);
            float4 m2 = make_float4(ssq[j].x - ssum[j].x * ssum[j].x * inv, ssq[j].y - ssum[j].y * ssum[j].y * inv,
                                    ssq[j].z - ssum[j].z * ssum[j].z * inv, ssq[j].w - ssum[j].w * ssum[j].w * inv);
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                const float nb = __shfl_xor(n, o, 64);
                const float nt = n + nb;
                const float wb = nt > 0.f ? nb / nt : 0.f;          // weight of the partner's mean
                const float cf = n * wb;                            // n * nb / nt
#define RIH_MERGE(c_)                                                        \
    {                                                                        \
        const float mb = __shfl_xor(mean.c_, o, 64), qb = __shfl_xor(m2.c_, o, 64); \
        const float dl = mb - mean.c_;                                       \
        mean.c_ += dl * wb;                                                  \
        m2.c_ += qb + dl * dl * cf;                                          \
    }
                RIH_MERGE(x) RIH_MERGE(y) RIH_MERGE(z) RIH_MERGE(w)
#undef RIH_MERGE
                n = nt;
            }
            const int nn = nbase + j * 32 + (lane & 7) * 4;
            if ((lane >> 3) == 0 && nn < p.N && mbase < p.M) {
                float* s0 = p.stats + (rb * 2 + 0) * p.N + nn;
                float* s1 = p.stats + (rb * 2 + 1) * p.N + nn;
                if (vec && nn + 3 < p.N) {
                    *reinterpret_cast<float4*>(s0) = mean;
                    *reinterpret_cast<float4*>(s1) = m2;
                } else {
                    const float a[4] = {mean.x, mean.y, mean.z, mean.w};
                    const float b[4] = {m2.x, m2.y, m2.z, m2.w};
                    for (int e = 0; e < 4 && nn + e < p.N; ++e) { s0[e] = a[e]; s1[e] = b[e]; }
                }
            }
        }
    }
}

template <int BM, int BN, int AMODE, int BMODE, bool VEC, int ENGINE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs p) {
    constexpr int LDAS = BM + 4;
    constexpr int LDBS = BN + 4;
    constexpr int PLANE_A = BM * 16, PLANE_B = BN * 16;     // split engine: dwords per bf16 plane
    constexpr int WGN = (BN >= 64) ? 2 : 1;             // wave grid: WGM x WGN = 4 waves
    constexpr int WGM = 4 / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NPA = BM / 32, NPB = BN / 32;

    constexpr int OPER_FLOATS = ENGINE ? 3 * (PLANE_A + PLANE_B) : BK * (LDAS + LDBS);
    constexpr int SMEM_FLOATS = OPER_FLOATS > 4 * 32 * 36 ? OPER_FLOATS : 4 * 32 * 36;      // >= the epilogue's staging area
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* As = smem;
    float* Bs = smem + (ENGINE ? 3 * PLANE_A : BK * LDAS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tilesN = (p.N + BN - 1) / BN;
    // Workgroups are dispatched x-fastest and round-robin over the 8 XCDs.  Remap so that each XCD (private L2) gets a
    // contiguous run of output tiles -- and, for a split-K launch, of (split, tile) pairs with the tile fastest, so
    // that all tiles reading one K-range of the operands sit on the same XCD instead of fetching it eight times.
    int bid, z = blockIdx.z;
    if (p.splitk > 1 && gridDim.z == (unsigned)p.splitk) {
        const int c = xcd_remap(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
        bid = c % gridDim.x;
        z = c / gridDim.x;
    } else {
        bid = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;

    const int split = z % p.splitk;
    const int bz = z / p.splitk;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    const float* __restrict__ A = p.A + b1 * p.sA1 + b2 * p.sA2;
    const float* __restrict__ B = p.B + b1 * p.sB1 + b2 * p.sB2;
    float* __restrict__ C = p.C + b1 * p.sC1 + b2 * p.sC2 + split * p.sCsplit;
    const float* __restrict__ biasp = p.bias != nullptr ? p.bias + b1 * p.sBias1 : nullptr;
    const float* __restrict__ Rp = p.R != nullptr ? p.R + b1 * p.sR1 : nullptr;

    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    // ------------------------------------------------------------------ A loader state
    // AMODE 0: thread -> (row = tid/8 (+32 per pass), k-quad = tid%8)
    // AMODE 1: thread -> (k-row = tid/(BM/4) (+256/(BM/4) per pass), m-quad = tid%(BM/4))
    unsigned a_base[NPA];           // element offset of the row's image (host guarantees the A extent < 2^31)
    int a_hw0[NPA];                 // packed (hi0 | wi0 << 16), hi0/wi0 = top-left input coordinate of the row's window
    int a_kh, a_kw, a_ci;           // AMODE 0: running (kh,kw,ci) of this thread's k-quad; AMODE 1: fixed tap of m-quad
    int a_mvalid = 0;               // AMODE 1: number of valid elements in this thread's m-quad (0..4)
    int a_one = -1;                 // AMODE 1: element of the m-quad that is the all-ones row, or -1
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
                a_base[i] = (unsigned)img * (unsigned)(p.H * p.W) * (unsigned)p.lda;
                const int hi0 = ho * p.strideA - p.padH, wi0 = wo * p.strideA - p.padW;
                a_hw0[i] = (hi0 & 0xffff) | (wi0 << 16);
            } else {
                a_base[i] = 0;
                a_hw0[i] = (int)0x80008000;     // hi0 = wi0 = -32768: never in range
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
        // (rows from ones_row on are not in memory: the all-ones row is synthesised below, the rows behind it are padding)
        a_mvalid = max(0, min(4, (p.ones_row > 0 ? p.ones_row : p.M) - mm));
        if (p.ones_row > 0 && p.ones_row >= mm && p.ones_row < mm + 4) a_one = p.ones_row - mm;
#pragma unroll
        for (int i = 0; i < NPA; ++i) { a_base[i] = 0; a_hw0[i] = 0; }
    }

    // Loads are branch-free: every lane always issues its loads (an invalid element reads a safe in-bounds dummy
    // address) and a 4-bit-per-load validity mask is applied when the registers are written to LDS.  Nothing touches a
    // loaded value before that point, so all global loads of a k-tile are in flight together across the MFMA phase.
    float4 areg[NPA], breg[NPB];
    unsigned amask = 0, bmask = 0;          // bit (4*i + e): element e of load i is valid

    auto ld4 = [&](const float* ptr, unsigned bits) -> float4 {
        float4 v;
        if (VEC) {      // 16-byte aligned rows: one dwordx4 (a partially valid tail quad stays inside the row pitch)
            v = *reinterpret_cast<const float4*>(ptr);
        } else {        // unaligned / odd pitch: four scalar loads, an invalid element re-reads element 0
            v.x = ptr[0];
            v.y = ptr[(bits >> 1) & 1u];
            v.z = ptr[((bits >> 2) & 1u) * 2u];
            v.w = ptr[((bits >> 3) & 1u) * 3u];
        }
        return v;
    };
    auto mask4 = [](float4 v, unsigned bits) -> float4 {
        v.x = (bits & 1u) ? v.x : 0.f;
        v.y = (bits & 2u) ? v.y : 0.f;
        v.z = (bits & 4u) ? v.z : 0.f;
        v.w = (bits & 8u) ? v.w : 0.f;
        return v;
    };
    auto first_bits = [](int nvalid) -> unsigned {      // mask of the first min(nvalid,4) elements
        return nvalid >= 4 ? 15u : (nvalid <= 0 ? 0u : ((1u << nvalid) - 1u));
    };

    auto load_A = [&](int ktile) {
        amask = 0;
        if (AMODE == 0) {
            const unsigned kbits = (a_kh < p.KH) ? first_bits(p.Cin - a_ci) : 0u;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                int hi = (int)(short)(a_hw0[i] & 0xffff) + a_kh, wi = (a_hw0[i] >> 16) + a_kw;
                bool ok = (hi >= 0) && (wi >= 0) && (kbits != 0u);
                if (p.upS > 1) {
                    ok = ok && (hi % p.upS == 0) && (wi % p.upS == 0);
                    hi /= p.upS;
                    wi /= p.upS;
                }
                ok = ok && (hi < p.H) && (wi < p.W);
                const unsigned bits = ok ? kbits : 0u;
                const unsigned off = ok ? (a_base[i] + (unsigned)(hi * p.W + wi) * (unsigned)p.lda + (unsigned)a_ci) : 0u;
                areg[i] = ld4(A + off, bits);
                amask |= bits << (4 * i);
            }
        } else {
            const int akr = tid / QA;
            const unsigned mbits = (a_kh < p.KH) ? first_bits(a_mvalid) : 0u;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const int k = ktile + (ENGINE ? akr * NPA + i : akr + RA * i);
                const int kc = min(k, kend - 1);
                const int wo = kc % p.Wo;
                const int t = kc / p.Wo;
                const int ho = t % p.Ho;
                const int img = t / p.Ho;
                const int hi = ho * p.strideA - p.padH + a_kh;
                const int wi = wo * p.strideA - p.padW + a_kw;
                const bool ok = (k < kend) && (mbits != 0u) && hi >= 0 && wi >= 0 && hi < p.H && wi < p.W;
                const unsigned bits = ok ? mbits : 0u;
                const unsigned off = ok ? ((unsigned)((img * p.H + hi) * p.W + wi) * (unsigned)p.lda + (unsigned)a_ci) : 0u;
                areg[i] = ld4(A + off, bits);
                amask |= bits << (4 * i);
                if (a_one >= 0 && k < kend) {       // the all-ones row of the bias gradient
                    set_elem(areg[i], a_one, 1.f);
                    amask |= (1u << a_one) << (4 * i);
                }
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

    // split engine: store NP k-consecutive values of one operand row as bf16 hi/mid/lo (NP = 4: ds_write_b64 per
    // plane at dword offset `o` (even); NP = 2: ds_write_b32)
    auto put4 = [](float* base, int plane, int o, float x0, float x1, float x2, float x3) {
        unsigned h0, m0, l0, h1, m1, l1;
        split2(x0, x1, h0, m0, l0);
        split2(x2, x3, h1, m1, l1);
        unsigned* u = reinterpret_cast<unsigned*>(base) + o;
        *reinterpret_cast<uint2*>(u) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(u + plane) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(u + 2 * plane) = make_uint2(l0, l1);
    };
    auto put2 = [](float* base, int plane, int o, float x0, float x1) {
        unsigned h, m, l;
        split2(x0, x1, h, m, l);
        unsigned* u = reinterpret_cast<unsigned*>(base) + o;
        u[0] = h;
        u[plane] = m;
        u[2 * plane] = l;
    };
    // K-contiguous loader (thread = row tid/8 (+32 per pass), k-quad tid%8) -> one b64 per plane
    auto store_kcontig = [&](float* base, int plane, const float4* reg, unsigned mask, int npass) {
        const int row = tid >> 3, q = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < npass) {
                const float4 v = mask4(reg[i], (mask >> (4 * i)) & 15u);
                const int m = row + 32 * i;
                put4(base, plane, lds_row(m) + 4 * ((q >> 1) ^ lds_swz(m)) + 2 * (q & 1), v.x, v.y, v.z, v.w);
            }
        }
    };
    // K-strided loader (thread = row-quad tid%Q, k-group tid/Q, loads i = 0..NP-1 are k-consecutive): transpose in
    // registers, one b64 (NP = 4) or b32 (NP = 2) per row and plane
    auto store_kstrided = [&](float* base, int plane, const float4* reg, unsigned mask, int rows) {
        const int Q = rows / 4, mq = tid % Q, kr = tid / Q;
        if (rows == 128) {
            const float4 v0 = mask4(reg[0], mask & 15u), v1 = mask4(reg[1], (mask >> 4) & 15u),
                         v2 = mask4(reg[2], (mask >> 8) & 15u), v3 = mask4(reg[3], (mask >> 12) & 15u);
            const int sw = 4 * ((kr >> 1) ^ lds_swz(4 * mq)) + 2 * (kr & 1);     // lds_swz is the same for the 4 rows
            put4(base, plane, lds_row(4 * mq + 0) + sw, v0.x, v1.x, v2.x, v3.x);
            put4(base, plane, lds_row(4 * mq + 1) + sw, v0.y, v1.y, v2.y, v3.y);
            put4(base, plane, lds_row(4 * mq + 2) + sw, v0.z, v1.z, v2.z, v3.z);
            put4(base, plane, lds_row(4 * mq + 3) + sw, v0.w, v1.w, v2.w, v3.w);
        } else {    // rows == 64: two k per thread -> dword kr of the row
            const float4 v0 = mask4(reg[0], mask & 15u), v1 = mask4(reg[1], (mask >> 4) & 15u);
            const int sw = 4 * ((kr >> 2) ^ lds_swz(4 * mq)) + (kr & 3);
            put2(base, plane, lds_row(4 * mq + 0) + sw, v0.x, v1.x);
            put2(base, plane, lds_row(4 * mq + 1) + sw, v0.y, v1.y);
            put2(base, plane, lds_row(4 * mq + 2) + sw, v0.z, v1.z);
            put2(base, plane, lds_row(4 * mq + 3) + sw, v0.w, v1.w);
        }
    };

    auto store_A = [&]() {
        if (ENGINE) {
            if (AMODE == 0) store_kcontig(As, PLANE_A, areg, amask, NPA);
            else store_kstrided(As, PLANE_A, areg, amask, BM);
            return;
        }
        if (AMODE == 0) {
            const int arow = tid >> 3, aq = tid & 7;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const float4 v = mask4(areg[i], (amask >> (4 * i)) & 15u);
                float* dst = As + (aq * 4) * LDAS + arow + 32 * i;
                dst[0] = v.x;
                dst[LDAS] = v.y;
                dst[2 * LDAS] = v.z;
                dst[3 * LDAS] = v.w;
            }
        } else {
            const int amq = tid % QA, akr = tid / QA;
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                *reinterpret_cast<float4*>(As + (akr + RA * i) * LDAS + amq * 4) =
                    mask4(areg[i], (amask >> (4 * i)) & 15u);
        }
    };

    // ------------------------------------------------------------------ B loader
    constexpr int QB = BN / 4, RB = 256 / QB;
    auto load_B = [&](int ktile) {
        bmask = 0;
        if (BMODE == 0) {
            const int bnq = tid % QB, bkr = tid / QB;
            const int n = n0 + bnq * 4;
            const unsigned nbits = first_bits(p.N - n);
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int k = ktile + (ENGINE ? bkr * NPB + i : bkr + RB * i);
                const unsigned bits = (k < kend) ? nbits : 0u;
                const unsigned off = bits ? ((unsigned)k * (unsigned)p.ldb + (unsigned)n) : 0u;
                breg[i] = ld4(B + off, bits);
                bmask |= bits << (4 * i);
            }
        } else {
            const int brow = tid >> 3, bq = tid & 7;
            const int k = ktile + bq * 4;
            const unsigned kbits = first_bits(kend - k);
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int n = n0 + brow + 32 * i;
                const unsigned bits = (n < p.N) ? kbits : 0u;
                const unsigned off = bits ? ((unsigned)n * (unsigned)p.ldb + (unsigned)k) : 0u;
                breg[i] = ld4(B + off, bits);
                bmask |= bits << (4 * i);
            }
        }
    };

    auto store_B = [&]() {
        if (ENGINE) {
            if (BMODE == 1) store_kcontig(Bs, PLANE_B, breg, bmask, NPB);
            else store_kstrided(Bs, PLANE_B, breg, bmask, BN);
            return;
        }
        if (BMODE == 0) {
            const int bnq = tid % QB, bkr = tid / QB;
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                *reinterpret_cast<float4*>(Bs + (bkr + RB * i) * LDBS + bnq * 4) =
                    mask4(breg[i], (bmask >> (4 * i)) & 15u);
        } else {
            const int brow = tid >> 3, bq = tid & 7;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const float4 v = mask4(breg[i], (bmask >> (4 * i)) & 15u);
                float* dst = Bs + (bq * 4) * LDBS + brow + 32 * i;
                dst[0] = v.x;
                dst[LDBS] = v.y;
                dst[2 * LDBS] = v.z;
                dst[3 * LDBS] = v.w;
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

    // split engine operand addresses: lane (l31, lhi) fetches chunk 2*s + lhi (8 k) of its rows for k16-step s
    int sa_off[2], sb_off[2];
    {
        const int ma = wm * WM + l31, nb = wn * WN + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            sa_off[s] = lds_row(ma) + 4 * ((2 * s + lhi) ^ lds_swz(ma));
            sb_off[s] = lds_row(nb) + 4 * ((2 * s + lhi) ^ lds_swz(nb));
        }
    }

    for (int t = 0; t < ntiles; ++t) {
        const bool more = (t + 1 < ntiles);
        if (more) {
            advance_A();
            load_A(kbeg + (t + 1) * BK);
            load_B(kbeg + (t + 1) * BK);
        }
        if (ENGINE) {
            const unsigned* Au = reinterpret_cast<const unsigned*>(As);
            const unsigned* Bu = reinterpret_cast<const unsigned*>(Bs);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        av[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Au + pl * PLANE_A + sa_off[s] + i * 512));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bv[pl][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bu + pl * PLANE_B + sb_off[s] + j * 512));
                }
                // smallest terms first; consecutive MFMAs rotate over the TM x TN accumulators
#define RIH_SPLIT_TERM(PA_, PB_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] =      \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA_][i], bv[PB_][j], acc[i][j], 0, 0, 0);
                RIH_SPLIT_TERM(2, 0)
                RIH_SPLIT_TERM(0, 2)
                RIH_SPLIT_TERM(1, 1)
                RIH_SPLIT_TERM(1, 0)
                RIH_SPLIT_TERM(0, 1)
                RIH_SPLIT_TERM(0, 0)
#undef RIH_SPLIT_TERM
            }
        } else {
            float av[2][TM], bv[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[0][i] = a_rd[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[0][j] = b_rd[j * 32];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 2) {      // LDS operands of step kk+1 are requested before the MFMAs of step kk
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[nxt][i] = a_rd[((kk + 1) * 2) * LDAS + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[nxt][j] = b_rd[((kk + 1) * 2) * LDBS + j * 32];
                }
                // pin the order: hipcc otherwise sinks the reads next to their MFMAs and exposes the LDS latency
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (more) {
            store_A();
            store_B();
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue (store_tiles_wide)
    // C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); every path to here ends with a
    // barrier (before the loop, and as the loop's last statement)
    store_tiles_wide<TM, TN>(p, acc, smem + wave * (32 * SLD), C, biasp, Rp, m0 + wm * WM, n0 + wn * WN, lane);
}

template <int BM, int BN, bool VEC, int ENGINE>
int launch_tile_e(const GemmArgs& a, int a_mode, int b_mode, dim3 grid, hipStream_t s) {
    dim3 block(256);
    if (a_mode == 0 && b_mode == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 0, VEC, ENGINE>), grid, block, 0, s, a);
    else if (a_mode == 0 && b_mode == 1) hipLaunchKernelGGL((gemm_kernel<BM, BN, 0, 1, VEC, ENGINE>), grid, block, 0, s, a);
    else if (a_mode == 1 && b_mode == 0) hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 0, VEC, ENGINE>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<BM, BN, 1, 1, VEC, ENGINE>), grid, block, 0, s, a);
    return (int)hipGetLastError();
}

template <int BM, int BN>
int launch_tile(const GemmArgs& a, int a_mode, int b_mode, int engine, dim3 grid, hipStream_t s) {
    // the vector variant needs 16-byte aligned rows on both operands and whole quads along B's contiguous dim
    const bool vec = a.vecA && a.vecB;
    if constexpr (BN >= 64) {       // the split engine's operand tiles are 64 or 128 rows
        if (engine == 1)
            return vec ? launch_tile_e<BM, BN, true, 1>(a, a_mode, b_mode, grid, s)
                       : launch_tile_e<BM, BN, false, 1>(a, a_mode, b_mode, grid, s);
    }
    return vec ? launch_tile_e<BM, BN, true, 0>(a, a_mode, b_mode, grid, s)
               : launch_tile_e<BM, BN, false, 0>(a, a_mode, b_mode, grid, s);
}

// ================================================================================================
// Split engine, fast path.  Same math as gemm_kernel<..., ENGINE 1> (three-term bf16 split, six MFMA products) but
// with the loader VALU cut to the bone, because the split engine is VALU-bound otherwise:
//   * operands are fetched with raw buffer loads: an invalid quad (conv halo, row >= M/N, k >= kend) is given an
//     offset >= 2^31 and the hardware range check returns zeros -- no masks, no selects on the data path;
//   * every per-row quantity (window origin offset, per-tap validity bitmask, LDS store address) is computed once
//     in the prologue; per k-tile a load costs an add (+ a bit test and select for conv rows), the tap / channel
//     walk is wave-uniform (scalar unit).
// Preconditions (checked by rih_gemm, which otherwise uses the general kernel): 16-byte aligned operands, upS == 1,
// conv A-gather (AMODE 0, not plain) needs Cin % 32 == 0 and KH*KW <= 32; transpose gather (AMODE 1, not plain)
// needs Wo % 4 == 0; K % 4 == 0; N % 4 == 0 for BMODE 0; operand slices < 2 GiB.
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// The kernel body takes the block coordinates as arguments: gemm_split_kernel passes blockIdx / gridDim, the grouped launch
// (gemm_split_multi_kernel, rih_gemm_multi) the coordinates of a block inside ITS problem of a descriptor table.
// ENG 2: the two-term fp16 split (three MFMA products, see e2_scale / split2h above) instead of the three-term bf16 one; same
// loaders, LDS layout (two planes instead of three) and epilogue.
template <int BM, int BN, int AMODE, int BMODE, bool PLAIN, bool STATS = false, bool DROP = false, int ENG = 1,
          bool SEG = false>
__device__ __forceinline__ void gemm_split_body(const GemmArgs& p, const int blk_x, const int blk_z, const int grid_x,
                                                const int grid_z) {
    static_assert(BMODE == 0 || BMODE == 1, "B is row-major [K][N] (0) or [N][K] (1)");
    static_assert(!SEG || (AMODE == 0 && PLAIN), "a segmented A operand is a plain row-major one");
    constexpr int NPL = (ENG == 2) ? 2 : 3;         // 16-bit planes per operand
    // global->register prefetch depth: k-tiles in flight.  The 64x64 tile (decoder-sized problems: a handful of
    // k-tiles, 12 MFMAs each) is bound by the load round trip per k-tile, so it keeps three tiles in flight.
    // (measured in round 2: 3 tiles in flight for the 64x64 tile changes nothing on decoder-sized problems, the floor is elsewhere.)
    constexpr int PF = 1;
    constexpr int PLANE_A = BM * 16, PLANE_B = BN * 16;
    constexpr int WGN = 2, WGM = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NPA = BM / 32, NPB = BN / 32;
    constexpr int QA = BM / 4, QB = BN / 4;

    constexpr int OPER_DW = NPL * (PLANE_A + PLANE_B);
    constexpr int SMEM_DW = OPER_DW > 4 * 32 * SLD ? OPER_DW : 4 * 32 * SLD;       // >= the epilogue's staging area
    __shared__ __attribute__((aligned(16))) unsigned smem[SMEM_DW];
    unsigned* As = smem;
    unsigned* Bs = smem + NPL * PLANE_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // engine 2: power-of-two operand scales from the device-resident bounds (scalar loads, wave-uniform)
    float e2_sa = 1.f, e2_sb = 1.f;
    if (ENG == 2) {
        e2_sa = e2_scale(p.amax_a, AMODE == 1 && p.ones_row > 0);
        e2_sb = e2_scale(p.amax_b, false);
    }

    const int tilesN = (p.N + BN - 1) / BN;
    // Workgroups are dispatched x-fastest and round-robin over the 8 XCDs.  Remap so that each XCD (private L2) gets a
    // contiguous run of output tiles -- and, for a split-K launch, of (split, tile) pairs with the tile fastest, so
    // that all tiles reading one K-range of the operands sit on the same XCD instead of fetching it eight times.
    int bid, z = blk_z;
    if (p.splitk > 1 && grid_z == p.splitk) {
        const int c = xcd_remap(blk_x + grid_x * blk_z, grid_x * grid_z);
        bid = c % grid_x;
        z = c / grid_x;
    } else {
        bid = xcd_remap(blk_x, grid_x);
    }
    const int m0 = (bid / tilesN) * BM;
    const int n0 = (bid % tilesN) * BN;

    const int split = z % p.splitk;
    const int bz = z / p.splitk;
    const int b2 = bz % p.nb2, b1 = bz / p.nb2;
    float* __restrict__ C = p.C + b1 * p.sC1 + b2 * p.sC2 + split * p.sCsplit;
    const float* __restrict__ biasp = p.bias != nullptr ? p.bias + b1 * p.sBias1 : nullptr;
    const float* __restrict__ Rp = p.R != nullptr ? p.R + b1 * p.sR1 : nullptr;
    const __amdgpu_buffer_rsrc_t rA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + b1 * p.sA1 + b2 * p.sA2), (short)0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + b1 * p.sB1 + b2 * p.sB2), (short)0, (int)p.b_bytes, 0x00020000);
    // segmented A: one more resource per further segment (never read when SEG is off; NULL segments get an empty extent)
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)(SEG ? p.Aseg[0] : p.A), (short)0,
                                                                         SEG ? (int)p.aseg_bytes[0] : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(SEG && p.Aseg[1] ? p.Aseg[1] : p.A), (short)0,
                                                                         SEG && p.Aseg[1] ? (int)p.aseg_bytes[1] : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA3 = __builtin_amdgcn_make_buffer_rsrc((void*)(SEG && p.Aseg[2] ? p.Aseg[2] : p.A), (short)0,
                                                                         SEG && p.Aseg[2] ? (int)p.aseg_bytes[2] : 0, 0x00020000);

    const int kbeg = split * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    // ------------------------------------------------------------------ per-thread loader constants
    const int row8 = tid >> 3, q8 = tid & 7;                 // K-contiguous operands: row (+32 per pass), k-quad
    unsigned a_off[NPA];        // AMODE 0: byte offset of (row window origin, k-quad); AMODE 1: see below
    unsigned a_val[NPA];        // AMODE 0 conv: per-tap validity bits of the row
    int a_st[4];                // LDS store offsets (dwords)
    // AMODE 1 state: thread = (m-quad amq, k-group akr); its NPA loads are k-consecutive
    const int amq = tid % QA, akr = tid / QA;
    int a1_wo = 0, a1_ho = 0, a1_img = 0, a1_kh = 0, a1_kw = 0;
    unsigned a1_base = 0;       // byte offset of (channel quad) -- OOB when the m-quad is out of range
    int a1_one = -1;            // element of this thread's m-quad that is the all-ones (bias-gradient) row, or -1
    if (AMODE == 1 && p.ones_row > 0 && p.ones_row >= m0 + 4 * amq && p.ones_row < m0 + 4 * amq + 4)
        a1_one = p.ones_row - (m0 + 4 * amq);
    if (AMODE == 0) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = m0 + row8 + 32 * i;
            a_val[i] = 0;
            if (m >= p.M) {
                a_off[i] = OOB;
            } else if (PLAIN) {
                a_off[i] = ((unsigned)m * (unsigned)p.lda + 4u * q8) * 4u;
            } else {
                const int wo = m % p.Wo;
                const int t = m / p.Wo;
                const int ho = t % p.Ho;
                const int img = t / p.Ho;
                const int hi0 = ho * p.strideA - p.padH, wi0 = wo * p.strideA - p.padW;
                a_off[i] = (unsigned)((((img * p.H + hi0) * p.W + wi0) * p.lda + 4 * q8) * 4);
                unsigned bits = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W)
                            bits |= 1u << (kh * p.KW + kw);
                a_val[i] = bits;
            }
        }
        const int o = lds_row(row8) + 4 * ((q8 >> 1) ^ lds_swz(row8)) + 2 * (q8 & 1);
        a_st[0] = o; a_st[1] = o + 512; a_st[2] = o + 1024; a_st[3] = o + 1536;
    } else {
        const int mm = m0 + 4 * amq;
        if (mm >= p.M) {
            a1_base = OOB;
        } else if (PLAIN) {
            a1_base = ((unsigned)(akr * NPA) * (unsigned)p.lda + (unsigned)mm) * 4u;
        } else {
            const int tap = mm / p.Cin;
            a1_base = (unsigned)(mm - tap * p.Cin) * 4u;
            a1_kh = tap / p.KW;
            a1_kw = tap - a1_kh * p.KW;
            const int k0 = kbeg + akr * NPA;            // first pixel of this thread in the first k-tile
            a1_wo = k0 % p.Wo;
            const int t = k0 / p.Wo;
            a1_ho = t % p.Ho;
            a1_img = t / p.Ho;
        }
#pragma unroll
        for (int i = 0; i < NPA; ++i) { a_off[i] = 0; a_val[i] = 0; }
        const int fl = (amq >> 2) & 1;          // = lds_row's flip bit of rows 4*amq..4*amq+3
        const int sw = (NPA == 4) ? 4 * ((akr >> 1) ^ (amq & 3)) + 2 * (akr & 1) : 4 * ((akr >> 2) ^ (amq & 3)) + (akr & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) a_st[j] = (4 * amq + (j ^ fl)) * 16 + sw;
    }

    unsigned b_off[NPB];
    int b_st[4];
    const int bnq = tid % QB, bkr = tid / QB;
    if (BMODE == 1) {
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const int n = n0 + row8 + 32 * i;
            b_off[i] = (n < p.N) ? ((unsigned)n * (unsigned)p.ldb + 4u * q8) * 4u : OOB;
        }
        const int o = lds_row(row8) + 4 * ((q8 >> 1) ^ lds_swz(row8)) + 2 * (q8 & 1);
        b_st[0] = o; b_st[1] = o + 512; b_st[2] = o + 1024; b_st[3] = o + 1536;
    } else {        // BMODE 0
        const int n = n0 + 4 * bnq;
        const unsigned base = (n < p.N) ? ((unsigned)(bkr * NPB) * (unsigned)p.ldb + (unsigned)n) * 4u : OOB;
#pragma unroll
        for (int i = 0; i < NPB; ++i) b_off[i] = base + (unsigned)i * (unsigned)p.ldb * 4u;
        const int fl = (bnq >> 2) & 1;
        const int sw = (NPB == 4) ? 4 * ((bkr >> 1) ^ (bnq & 3)) + 2 * (bkr & 1) : 4 * ((bkr >> 2) ^ (bnq & 3)) + (bkr & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) b_st[j] = (4 * bnq + (j ^ fl)) * 16 + sw;
    }

    // wave-uniform walk over (tap, channel) for the conv A-gather: one tap per k-tile (Cin % 32 == 0)
    int u_tap = 0, u_ci = 0, u_kh = 0, u_kw = 0;
    if (AMODE == 0 && !PLAIN) {
        u_tap = kbeg / p.Cin;
        u_ci = kbeg - u_tap * p.Cin;
        u_kh = u_tap / p.KW;
        u_kw = u_tap - u_kh * p.KW;
    }

    float4 areg[PF][NPA], breg[PF][NPB];

    auto load_A = [&](int ktile, int st) {
        if (AMODE == 0) {
            if (PLAIN && SEG) {
                // the k-tile lies in ONE segment (boundaries are multiples of 32): wave-uniform choice of base, pitch and extent;
                // a channel concatenation (models/encoder.py:165-173) is read in place, never materialised
                const bool in_k = ktile + 4 * q8 < kend;
                auto seg_load = [&](const __amdgpu_buffer_rsrc_t r, const int lda_s, const int k0) {
                    const unsigned ku = in_k ? (unsigned)(ktile - k0 + 4 * q8) * 4u : OOB;
#pragma unroll
                    for (int i = 0; i < NPA; ++i)
                        areg[st][i] = bload4(r, (a_off[i] == OOB || ku == OOB) ? OOB
                                                    : (unsigned)(m0 + row8 + 32 * i) * (unsigned)lda_s * 4u + ku);
                };
                if (ktile < p.kseg[0]) seg_load(rA, p.lda, 0);
                else if (p.Aseg[1] == nullptr || ktile < p.kseg[1]) seg_load(rA1, p.ldaseg[0], p.kseg[0]);
                else if (p.Aseg[2] == nullptr || ktile < p.kseg[2]) seg_load(rA2, p.ldaseg[1], p.kseg[1]);
                else seg_load(rA3, p.ldaseg[2], p.kseg[2]);
            } else if (PLAIN) {
                const unsigned ku = (ktile + 4 * q8 < kend) ? (unsigned)ktile * 4u : OOB;
#pragma unroll
                for (int i = 0; i < NPA; ++i) areg[st][i] = bload4(rA, a_off[i] + ku);
            } else {
                const unsigned tapoff = (unsigned)(((u_kh * p.W + u_kw) * p.lda + u_ci) * 4);
                const unsigned tapbit = (ktile < kend) ? (1u << (u_tap & 31)) : 0u;
#pragma unroll
                for (int i = 0; i < NPA; ++i) areg[st][i] = bload4(rA, (a_val[i] & tapbit) ? a_off[i] + tapoff : OOB);
            }
        } else {
            if (PLAIN) {
                const unsigned ku = (ktile + akr * NPA < kend) ? (unsigned)ktile * (unsigned)p.lda * 4u : OOB;
#pragma unroll
                for (int i = 0; i < NPA; ++i) areg[st][i] = bload4(rA, a1_base + ku + (unsigned)i * (unsigned)p.lda * 4u);
                if (a1_one >= 0) {
                    const float one = (ktile + akr * NPA < kend) ? 1.f : 0.f;       // K % NPA == 0: all or nothing
#pragma unroll
                    for (int i = 0; i < NPA; ++i) set_elem(areg[st][i], a1_one, one);
                }
            } else {
                const int hi = a1_ho * p.strideA - p.padH + a1_kh;
                const int wi = a1_wo * p.strideA - p.padW + a1_kw;
                const bool rowok = (ktile + akr * NPA < kend) && (unsigned)hi < (unsigned)p.H;
                const unsigned rowoff = (unsigned)((((a1_img * p.H + hi) * p.W + wi) * p.lda) * 4) + a1_base;
#pragma unroll
                for (int i = 0; i < NPA; ++i) {
                    const bool ok = rowok && (unsigned)(wi + i * p.strideA) < (unsigned)p.W;
                    areg[st][i] = bload4(rA, ok ? rowoff + (unsigned)(i * p.strideA * p.lda * 4) : OOB);
                }
                if (a1_one >= 0) {
                    const float one = (ktile + akr * NPA < kend) ? 1.f : 0.f;
#pragma unroll
                    for (int i = 0; i < NPA; ++i) set_elem(areg[st][i], a1_one, one);
                }
            }
        }
    };
    auto advance_A = [&]() {        // move the loader state one k-tile forward
        if (AMODE == 0 && !PLAIN) {
            u_ci += BK;
            if (u_ci >= p.Cin) {
                u_ci = 0;
                ++u_tap;
                if (++u_kw == p.KW) { u_kw = 0; ++u_kh; }
            }
        }
        if (AMODE == 1 && !PLAIN) {
            a1_wo += BK;
            while (a1_wo >= p.Wo) { a1_wo -= p.Wo; ++a1_ho; }
            while (a1_ho >= p.Ho) { a1_ho -= p.Ho; ++a1_img; }
        }
    };
    auto load_B = [&](int ktile, int st) {
        if (BMODE == 1) {
            const unsigned ku = (ktile + 4 * q8 < kend) ? (unsigned)ktile * 4u : OOB;
#pragma unroll
            for (int i = 0; i < NPB; ++i) breg[st][i] = bload4(rB, b_off[i] + ku);
        } else {
            const unsigned ku = (ktile + bkr * NPB < kend) ? (unsigned)ktile * (unsigned)p.ldb * 4u : OOB;
#pragma unroll
            for (int i = 0; i < NPB; ++i) breg[st][i] = bload4(rB, b_off[i] + ku);
        }
    };

    // `sc`: engine 2's operand scale (unused by engine 1)
    auto put4 = [](unsigned* u, int plane, float sc, float x0, float x1, float x2, float x3) {
        if (ENG == 2) {
            unsigned h0, l0, h1, l1;
            split2h(x0, x1, sc, h0, l0);
            split2h(x2, x3, sc, h1, l1);
            *reinterpret_cast<uint2*>(u) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(u + plane) = make_uint2(l0, l1);
        } else {
            unsigned h0, m0_, l0, h1, m1, l1;
            split2(x0, x1, h0, m0_, l0);
            split2(x2, x3, h1, m1, l1);
            *reinterpret_cast<uint2*>(u) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(u + plane) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(u + 2 * plane) = make_uint2(l0, l1);
        }
    };
    auto put2 = [](unsigned* u, int plane, float sc, float x0, float x1) {
        if (ENG == 2) {
            unsigned h, l;
            split2h(x0, x1, sc, h, l);
            u[0] = h;
            u[plane] = l;
        } else {
            unsigned h, m, l;
            split2(x0, x1, h, m, l);
            u[0] = h;
            u[plane] = m;
            u[2 * plane] = l;
        }
    };
    auto store_kcontig = [&](unsigned* base, int plane, float sc, const float4* reg, const int* st, int npass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < npass) put4(base + st[i], plane, sc, reg[i].x, reg[i].y, reg[i].z, reg[i].w);
    };
    auto store_kstrided = [&](unsigned* base, int plane, float sc, const float4* reg, const int* st, int np) {
        if (np == 4) {
            put4(base + st[0], plane, sc, reg[0].x, reg[1].x, reg[2].x, reg[3].x);
            put4(base + st[1], plane, sc, reg[0].y, reg[1].y, reg[2].y, reg[3].y);
            put4(base + st[2], plane, sc, reg[0].z, reg[1].z, reg[2].z, reg[3].z);
            put4(base + st[3], plane, sc, reg[0].w, reg[1].w, reg[2].w, reg[3].w);
        } else {
            put2(base + st[0], plane, sc, reg[0].x, reg[1].x);
            put2(base + st[1], plane, sc, reg[0].y, reg[1].y);
            put2(base + st[2], plane, sc, reg[0].z, reg[1].z);
            put2(base + st[3], plane, sc, reg[0].w, reg[1].w);
        }
    };
    auto store_A = [&](int st) {
        if (AMODE == 0) store_kcontig(As, PLANE_A, e2_sa, areg[st], a_st, NPA);
        else store_kstrided(As, PLANE_A, e2_sa, areg[st], a_st, NPA);
    };
    auto store_B = [&](int st) {
        if (BMODE == 1) store_kcontig(Bs, PLANE_B, e2_sb, breg[st], b_st, NPB);
        else store_kstrided(Bs, PLANE_B, e2_sb, breg[st], b_st, NPB);
    };

    // ------------------------------------------------------------------ main loop
    floatx16 acc[TM][TN];
    floatx16 acc1[ENG == 2 ? TM : 1][TN];           // engine 2: hi*lo + lo*hi (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (ENG == 2) acc1[i][j][r] = 0.f;
            }

    // prologue: tiles 0..PF-1 in flight (a tile at or beyond kend arrives as zeros and is never multiplied)
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        if (j > 0) advance_A();
        load_A(kbeg + j * BK, j);
        load_B(kbeg + j * BK, j);
    }

    const int l31 = lane & 31, lhi = lane >> 5;
    int sa_off[2], sb_off[2];
    {
        const int ma = wm * WM + l31, nb = wn * WN + l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            sa_off[s] = lds_row(ma) + 4 * ((2 * s + lhi) ^ lds_swz(ma));
            sb_off[s] = lds_row(nb) + 4 * ((2 * s + lhi) ^ lds_swz(nb));
        }
    }

    for (int t = 0; t < ntiles; t += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (t + j < ntiles) {           // uniform
                // register set j holds tile t+j: convert it into LDS, then refill the set with tile t+j+PF so that
                // the load flies during the MFMAs of this and the next PF-1 tiles
                store_A(j);
                store_B(j);
                advance_A();
                load_A(kbeg + (t + j + PF) * BK, j);
                load_B(kbeg + (t + j + PF) * BK, j);
                __syncthreads();
                if constexpr (ENG == 2) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        f16x8 av[2][TM], bv[2][TN];
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                            for (int i = 0; i < TM; ++i)
                                av[pl][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(As + pl * PLANE_A + sa_off[s] + i * 512));
#pragma unroll
                            for (int jj = 0; jj < TN; ++jj)
                                bv[pl][jj] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Bs + pl * PLANE_B + sb_off[s] + jj * 512));
                        }
#define RIH_E2_TERM(ACC_, PA_, PB_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj) ACC_[i][jj] = \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(av[PA_][i], bv[PB_][jj], ACC_[i][jj], 0, 0, 0);
                        RIH_E2_TERM(acc1, 1, 0)
                        RIH_E2_TERM(acc, 0, 0)
                        RIH_E2_TERM(acc1, 0, 1)
#undef RIH_E2_TERM
                    }
                } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 av[3][TM], bv[3][TN];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            av[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(As + pl * PLANE_A + sa_off[s] + i * 512));
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
                            bv[pl][jj] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Bs + pl * PLANE_B + sb_off[s] + jj * 512));
                    }
#define RIH_SPLIT_TERM(PA_, PB_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj) acc[i][jj] =  \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA_][i], bv[PB_][jj], acc[i][jj], 0, 0, 0);
                    RIH_SPLIT_TERM(2, 0)
                    RIH_SPLIT_TERM(0, 2)
                    RIH_SPLIT_TERM(1, 1)
                    RIH_SPLIT_TERM(1, 0)
                    RIH_SPLIT_TERM(0, 1)
                    RIH_SPLIT_TERM(0, 0)
#undef RIH_SPLIT_TERM
                }
                }
                __syncthreads();
            }
        }
    }

    // ------------------------------------------------------------------ epilogue (store_tiles_wide; the main loop ended with a barrier)
    if constexpr (ENG == 2) {
        store_tiles_wide<TM, TN, STATS, DROP, true>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * SLD), C, biasp, Rp,
                                                    m0 + wm * WM, n0 + wn * WN, lane, acc1, 1.f / e2_sa, 1.f / e2_sb);
    } else {
        store_tiles_wide<TM, TN, STATS, DROP>(p, acc, reinterpret_cast<float*>(smem) + wave * (32 * SLD), C, biasp, Rp, m0 + wm * WM, n0 + wn * WN,
                                        lane);
    }
}

template <int BM, int BN, int AMODE, int BMODE, bool PLAIN, bool STATS = false, bool DROP = false, int ENG = 1,
          bool SEG = false>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const GemmArgs p) {
    gemm_split_body<BM, BN, AMODE, BMODE, PLAIN, STATS, DROP, ENG, SEG>(p, (int)blockIdx.x, (int)blockIdx.z, (int)gridDim.x,
                                                                              (int)gridDim.z);
}

// ---- grouped launch (rih_gemm_multi): n independent problems of ONE kernel variant in one launch.  The table lives in device
// memory (uploaded by the caller: the descriptors of e.g. all weight gradients of a backward stage do not fit the 4 KB kernel
// argument): header, then per group of 8 consecutive blocks the problem it belongs to, then the problems.  Every problem's block
// count is padded to a multiple of 8, so that the low three bits of a block's index inside its problem are still the XCD it
// was dispatched to (xcd_remap); the padding blocks exit at once.  All table reads are wave-uniform scalar loads from the
// constant address space, exactly like the kernel-argument reads of the single launch.
struct MultiProb {
    GemmArgs a;
    int first;          // first block of the problem in the launch (multiple of 8)
    int gx, gz;         // the problem's own grid (gx * gz <= its padded block count)
    int pad;
};
struct MultiHeader {
    int n, total_blocks, groups, pad;
};
#ifndef RIH_CONST_AS        /* (the host build of tests/hipcpu defines it empty) */
#if defined(__HIP_DEVICE_COMPILE__)
#define RIH_CONST_AS __attribute__((address_space(4)))
#else
#define RIH_CONST_AS        /* hipcc's host pass only parses the kernel */
#endif
#endif

template <int BM, int BN, int AMODE, int BMODE, bool PLAIN, int ENG = 1>
__global__ __launch_bounds__(256, 2) void gemm_split_multi_kernel(const unsigned char* __restrict__ table) {
    const int b = (int)blockIdx.x;
    const MultiHeader RIH_CONST_AS* hd = (const MultiHeader RIH_CONST_AS*)table;
    const unsigned short RIH_CONST_AS* grp = (const unsigned short RIH_CONST_AS*)(table + sizeof(MultiHeader));
    const int groups = hd->groups;
    const int pi = (int)grp[b >> 3];
    const MultiProb RIH_CONST_AS* pr =
        (const MultiProb RIH_CONST_AS*)(table + sizeof(MultiHeader) + (((size_t)groups * 2 + 15) & ~(size_t)15)) + pi;
    const int lb = b - pr->first;
    const int gx = pr->gx, gz = pr->gz;
    if (lb >= gx * gz) return;              // padding block
    const GemmArgs p = pr->a;               // scalar loads; the copy lives in SGPRs like a kernel argument
    gemm_split_body<BM, BN, AMODE, BMODE, PLAIN, false, false, ENG>(p, lb % gx, lb / gx, gx, gz);
}

template <int BM, int BN>
int launch_split_e2(const GemmArgs& a, int a_mode, int b_mode, bool plain, dim3 grid, hipStream_t s) {
    dim3 block(256);
#define RIH_L2(AM_, BM_, PL_, ST_, DR_) \
    hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AM_, BM_, PL_, ST_, DR_, 2>), grid, block, 0, s, a)
    if (a_mode > 1 || b_mode > 1) return RIH_EINVAL;
    if (a.Aseg[0] != nullptr) {         // segmented A: plain rows x [N][K] weight (checked by the caller), with / without statistics
        if (a.stats != nullptr) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 1, true, true, false, 2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 1, true, false, false, 2, true>), grid, block, 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.drop_thr != 0u) {
        if (b_mode == 0) RIH_L2(0, 0, true, false, true); else RIH_L2(0, 1, true, false, true);
    } else if (a.stats != nullptr) {
        if (b_mode == 0) { if (plain) RIH_L2(0, 0, true, true, false); else RIH_L2(0, 0, false, true, false); }
        else { if (plain) RIH_L2(0, 1, true, true, false); else RIH_L2(0, 1, false, true, false); }
    }
    else if (a_mode == 0 && b_mode == 0) { if (plain) RIH_L2(0, 0, true, false, false); else RIH_L2(0, 0, false, false, false); }
    else if (a_mode == 0 && b_mode == 1) { if (plain) RIH_L2(0, 1, true, false, false); else RIH_L2(0, 1, false, false, false); }
    else if (a_mode == 1 && b_mode == 0) { if (plain) RIH_L2(1, 0, true, false, false); else RIH_L2(1, 0, false, false, false); }
    else { if (plain) RIH_L2(1, 1, true, false, false); else RIH_L2(1, 1, false, false, false); }
#undef RIH_L2
    return (int)hipGetLastError();
}

template <int BM, int BN>
int launch_split(const GemmArgs& a, int a_mode, int b_mode, bool plain, dim3 grid, hipStream_t s) {
    dim3 block(256);
#define RIH_LS(AM_, BM_, PL_) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, AM_, BM_, PL_>), grid, block, 0, s, a)
    if (a_mode > 1 || b_mode > 1) return RIH_EINVAL;
    if (a.Aseg[0] != nullptr) {         // segmented A (see launch_split_e2)
        if (a.stats != nullptr) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 1, true, true, false, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 1, true, false, false, 1, true>), grid, block, 0, s, a);
        return (int)hipGetLastError();
    }
    if (a.drop_thr != 0u) {        // dropout epilogue: plain a_mode-0 GEMMs only (checked by the caller)
        if (b_mode == 0) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 0, true, false, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, 1, true, false, true>), grid, block, 0, s, a);
    }
    else if (a.stats != nullptr) {      // statistics epilogue: forward-type GEMMs only (checked by the caller)
#define RIH_LSS(BM_, PL_) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, 0, BM_, PL_, true>), grid, block, 0, s, a)
        if (b_mode == 0) { if (plain) RIH_LSS(0, true); else RIH_LSS(0, false); }
        else { if (plain) RIH_LSS(1, true); else RIH_LSS(1, false); }
#undef RIH_LSS
    }
    else if (a_mode == 0 && b_mode == 0) { if (plain) RIH_LS(0, 0, true); else RIH_LS(0, 0, false); }
    else if (a_mode == 0 && b_mode == 1) { if (plain) RIH_LS(0, 1, true); else RIH_LS(0, 1, false); }
    else if (a_mode == 1 && b_mode == 0) { if (plain) RIH_LS(1, 0, true); else RIH_LS(1, 0, false); }
    else { if (plain) RIH_LS(1, 1, true); else RIH_LS(1, 1, false); }
#undef RIH_LS
    return (int)hipGetLastError();
}


// One-launch split-K reduction for weight gradients: each 256-thread block sums an 8x32 tile of the S partial slabs
// P[s][Mp][N] in a fixed order (deterministic) and writes it transposed into the parameter layout through LDS; blocks
// past the tile range sum slab row M (the all-ones row of the A operand = column sums of dy) into the bias gradient.
__device__ __forceinline__ void splitk_reduce_block(const float* __restrict__ P, int S, int Mp, int M, int N,
                                                    float* __restrict__ dst, int Cin, int taps, int CinValid, int accumulate,
                                                    float* __restrict__ db, int ntiles, int bx, int CinPitch = 0) {
    // CinPitch > 0: dst is a column slice of a wider [N][CinPitch][taps] parameter (rih_reduce_desc.CinPitch)
    const int pitch = CinPitch > 0 ? CinPitch : CinValid;
    const long long slab = (long long)Mp * N;
    if (bx >= ntiles) {
        const int n = (bx - ntiles) * 256 + threadIdx.x;
        if (n < N) {
            const float* q = P + (long long)M * N + n;
            float s0 = 0.f, s1 = 0.f;
            int k = 0;
            for (; k + 1 < S; k += 2) { s0 += q[(long long)k * slab]; s1 += q[(long long)(k + 1) * slab]; }
            if (k < S) s0 += q[(long long)k * slab];
            db[n] = s0 + s1;
        }
        return;
    }
    // 8 (m) x 32 (n) tile per block, one output element per thread: four times the blocks of a 32x32 tile, and the
    // S-long reduction of a thread is a single stream of independent loads (the kernel is latency-bound: decoder
    // weight gradients are 64x64 .. 512x768 outputs summed over 30-130 slabs)
    __shared__ float tile[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tilesN = (N + 31) / 32;
    const int m0 = (bx / tilesN) * 8, n0 = (bx % tilesN) * 32;
    {
        const int m = m0 + ty, n = n0 + tx;
        const bool ok = (m < M && n < N);
        const float* q = P + (ok ? (long long)m * N + n : 0);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = 0;
        for (; k + 3 < S; k += 4) {
            a0 += q[(long long)k * slab];
            a1 += q[(long long)(k + 1) * slab];
            a2 += q[(long long)(k + 2) * slab];
            a3 += q[(long long)(k + 3) * slab];
        }
        for (; k < S; ++k) a0 += q[(long long)k * slab];
        tile[ty][tx] = ok ? (a0 + a1) + (a2 + a3) : 0.f;
    }
    __syncthreads();
    {
        const int m = m0 + (threadIdx.x & 7), n = n0 + (threadIdx.x >> 3);
        if (m < M && n < N) {
            const int tap = m / Cin, ci = m - tap * Cin;
            if (ci < CinValid) {
                const long long o = ((long long)n * pitch + ci) * taps + tap;
                const float v = tile[threadIdx.x & 7][threadIdx.x >> 3];
                dst[o] = accumulate ? dst[o] + v : v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_fused_kernel(const float* __restrict__ P, int S, int Mp, int M, int N,
                                                                  float* __restrict__ dst, int Cin, int taps, int CinValid,
                                                                  int accumulate, float* __restrict__ db, int ntiles,
                                                                  long long sP, long long sDst, long long sDb) {
    // blockIdx.y = independent reductions of one launch (paired left/right-hand weight gradients)
    P += blockIdx.y * sP;
    dst += blockIdx.y * sDst;
    if (db != nullptr) db += blockIdx.y * sDb;
    splitk_reduce_block(P, S, Mp, M, N, dst, Cin, taps, CinValid, accumulate, db, ntiles, (int)blockIdx.x);
}

// Many independent reductions in ONE launch (rih_splitk_reduce_multi): the descriptors travel by value in the kernel
// argument (<= 4 KB), so the launch is self-contained -- no device table to upload, and a hipGraph captures it as is.
// A block finds its descriptor by scanning the exclusive prefix of block counts (<= REDUCE_PACK scalar compares).
constexpr int REDUCE_PACK = 56;
struct ReducePack {
    rih_reduce_desc d[REDUCE_PACK];
    int first[REDUCE_PACK + 1];         // first block of descriptor i; first[n] = total
    int ntiles[REDUCE_PACK];
    int n;
};
static_assert(sizeof(ReducePack) <= 4096, "kernel argument limit");
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(const ReducePack pk) {
    const int b = (int)blockIdx.x;
    int i = 0;
    while (i + 1 < pk.n && b >= pk.first[i + 1]) ++i;
    const rih_reduce_desc& d = pk.d[i];
    splitk_reduce_block(d.P, d.S, d.Mp, d.M, d.N, d.dst, d.Cin, d.taps, d.CinValid, d.accumulate, d.db, pk.ntiles[i],
                        b - pk.first[i], d.CinPitch);
}

// Forward split-K finish: C[m*ldc+n] = act(alpha * sum_s P[s][m][n] + bias[n] + R[m*ldr+n])
__global__ void splitk_finish_kernel(const float* __restrict__ P, int S, int M, int N, float* __restrict__ C, int ldc,
                                     const float* __restrict__ bias, const float* __restrict__ R, int ldr, float alpha,
                                     int relu) {
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i - (long long)m * N);
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += P[(long long)k * total + i];
        s *= alpha;
        if (bias != nullptr) s += bias[n];
        if (R != nullptr) s += R[(long long)m * ldr + n];
        if (relu) s = fmaxf(s, 0.f);
        C[(long long)m * ldc + n] = s;
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

// tap subset of a conv weight for one parity class of a strided data gradient (see rih_pack_conv_weight_sub)
__global__ void pack_conv_weight_sub_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin,
                                            int KH, int KW, int CinPad, int kh0, int kw0, int step, int Th, int Tw) {
    const long long total = (long long)Th * Tw * Cout * CinPad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % CinPad);
        long long t = i / CinPad;
        const int co = (int)(t % Cout);
        t /= Cout;
        const int tw = (int)(t % Tw), th = (int)(t / Tw);
        const int kh = kh0 + step * (Th - 1 - th), kw = kw0 + step * (Tw - 1 - tw);
        dst[i] = (ci < Cin) ? w[(((long long)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
    }
}
// Every weight operand of a step packed by ONE launch (rih_pack_conv_weight_multi; descriptors by value, see
// splitk_reduce_multi_kernel): mode 0 = forward operand [(tap, ci)][co], mode 1 = data-gradient operand of the tap subset
// (kh0, kw0, step, Th, Tw), flipped, [((th, tw), co)][ci] -- the same element maps as the two kernels above.
constexpr int PACK_PACK = 56;
struct PackPack {
    rih_pack_desc d[PACK_PACK];
    int first[PACK_PACK + 1];
    int n;
};
static_assert(sizeof(PackPack) <= 4096, "kernel argument limit");
__global__ __launch_bounds__(256) void pack_conv_weight_multi_kernel(const PackPack pk) {
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < pk.n && b >= pk.first[k + 1]) ++k;
    const rih_pack_desc& d = pk.d[k];
    const int nb = pk.first[k + 1] - pk.first[k];
    const int taps = d.KH * d.KW;
    const long long total = (d.mode == 0) ? (long long)taps * d.CinPad * d.Cout : (long long)d.Th * d.Tw * d.Cout * d.CinPad;
    for (long long i = (long long)(b - pk.first[k]) * 256 + threadIdx.x; i < total; i += (long long)nb * 256) {
        float v = 0.f;
        if (d.mode == 0) {
            const int co = (int)(i % d.Cout);
            const long long t = i / d.Cout;
            const int ci = (int)(t % d.CinPad), tap = (int)(t / d.CinPad);
            if (ci < d.Cin) v = d.w[((long long)co * d.Cin + ci) * taps + tap];
        } else {
            const int ci = (int)(i % d.CinPad);
            long long t = i / d.CinPad;
            const int co = (int)(t % d.Cout);
            t /= d.Cout;
            const int tw = (int)(t % d.Tw), th = (int)(t / d.Tw);
            const int kh = d.kh0 + d.step * (d.Th - 1 - th), kw = d.kw0 + d.step * (d.Tw - 1 - tw);
            if (ci < d.Cin) v = d.w[(((long long)co * d.Cin + ci) * d.KH + kh) * d.KW + kw];
        }
        d.dst[i] = v;
    }
}

}  // namespace

extern "C" int rih_pack_conv_weight_multi(const rih_pack_desc* descs, int n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return RIH_EINVAL;
    for (int i = 0; i < n; ++i) {
        const rih_pack_desc& d = descs[i];
        if (!d.w || !d.dst || d.Cout < 1 || d.Cin < 1 || d.KH < 1 || d.KW < 1 || d.CinPad < d.Cin || (d.mode != 0 && d.mode != 1))
            return RIH_EINVAL;
        if (d.mode == 1 && (d.step < 1 || d.Th < 1 || d.Tw < 1 || d.kh0 < 0 || d.kw0 < 0 ||
                            d.kh0 + d.step * (d.Th - 1) >= d.KH || d.kw0 + d.step * (d.Tw - 1) >= d.KW))
            return RIH_EINVAL;
    }
    for (int base = 0; base < n; base += PACK_PACK) {
        PackPack pk;
        pk.n = (n - base < PACK_PACK) ? n - base : PACK_PACK;
        int total = 0;
        for (int i = 0; i < pk.n; ++i) {
            const rih_pack_desc& d = descs[base + i];
            const long long el = (d.mode == 0) ? (long long)d.KH * d.KW * d.CinPad * d.Cout : (long long)d.Th * d.Tw * d.Cout * d.CinPad;
            long long nb = (el + 4095) / 4096;        // 16 elements per thread: these are latency-, not bandwidth-sized
            if (nb > 1024) nb = 1024;
            pk.d[i] = d;
            pk.first[i] = total;
            total += (int)nb;
        }
        pk.first[pk.n] = total;
        hipLaunchKernelGGL(pack_conv_weight_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, pk);
    }
    return (int)hipGetLastError();
}
extern "C" int rih_pack_conv_weight_sub(const float* w, float* dst, int Cout, int Cin, int KH, int KW, int CinPad,
                                        int kh0, int kw0, int step, int Th, int Tw, void* stream) {
    if (!w || !dst || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || CinPad < Cin || step < 1 || Th < 1 || Tw < 1 ||
        kh0 < 0 || kw0 < 0 || kh0 + step * (Th - 1) >= KH || kw0 + step * (Tw - 1) >= KW)
        return RIH_EINVAL;
    const long long total = (long long)Th * Tw * Cout * CinPad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_weight_sub_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin,
                       KH, KW, CinPad, kh0, kw0, step, Th, Tw);
    return (int)hipGetLastError();
}



struct PreparedGemm {       // what gemm_impl would launch on the split engine's fast path (tiles 0..2), for rih_gemm_multi
    GemmArgs a;
    int gx, gz, tile, a_mode, b_mode, plain, engine;
};

static int gemm_impl(const rih_gemm_desc* d, void* stream, int* stats_rows, PreparedGemm* prep = nullptr, int* engine_out = nullptr) {
    if (!d || !d->A || !d->B || !d->C) return RIH_EINVAL;
    if (d->M <= 0 || d->N <= 0 || d->K < 0) return RIH_EINVAL;
    if (d->splitk < 1 || d->nb1 < 1 || d->nb2 < 1) return RIH_EINVAL;
    if (d->splitk > 1 && (d->kchunk <= 0 || d->kchunk % BK != 0)) return RIH_EINVAL;
    if (d->Cin <= 0 || d->KH <= 0 || d->KW <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->upS < 1 || d->strideA < 1)
        return RIH_EINVAL;
    if (d->a_mode == 1 && d->upS != 1) return RIH_EINVAL;
    if (d->KH * d->KW > 1 && (d->Cin % 4) != 0) return RIH_EINVAL;   // quads must not straddle taps
    {   // 32-bit element offsets inside one batch slice of A and B; window coordinates packed in 16 bits
        const long long rowsA = (d->a_mode != 1) ? (long long)d->M : (long long)d->K;
        const long long imgs = (rowsA + (long long)d->Ho * d->Wo - 1) / ((long long)d->Ho * d->Wo);
        if (imgs * d->H * d->W * (long long)d->lda >= (1ll << 31)) return RIH_EINVAL;
        const long long rowsB = (d->b_mode == 0) ? (long long)d->K : (long long)d->N;
        if (rowsB * (long long)d->ldb >= (1ll << 31)) return RIH_EINVAL;
        if (d->b_mode < 0 || d->b_mode > 1 || d->a_mode < 0 || d->a_mode > 1) return RIH_EINVAL;
        if (d->H > 16000 || d->W > 16000 || d->Ho > 16000 || d->Wo > 16000 || d->strideA > 64 || d->padH > 64 ||
            d->padW > 64 || d->KH > 64 || d->KW > 64)
            return RIH_EINVAL;
    }
    GemmArgs a;
    a.A = d->A; a.B = d->B; a.C = d->C; a.bias = d->bias; a.R = d->R;
    a.M = d->M; a.N = d->N; a.K = d->K;
    a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc; a.ldr = d->ldr;
    a.nb2 = d->nb2; a.splitk = d->splitk;
    a.kchunk = (d->splitk > 1) ? d->kchunk : ((d->K + BK - 1) / BK) * BK + BK;
    a.sA1 = d->sA1; a.sA2 = d->sA2; a.sB1 = d->sB1; a.sB2 = d->sB2; a.sC1 = d->sC1; a.sC2 = d->sC2;
    a.sBias1 = d->sBias1; a.sR1 = d->sR1;
    a.sCsplit = d->sCsplit;
    a.alpha = d->alpha; a.relu = d->relu;
    a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.KH = d->KH; a.KW = d->KW;
    a.strideA = d->strideA; a.upS = d->upS; a.padH = d->padH; a.padW = d->padW;
    const bool a16 = ((uintptr_t)d->A % 16 == 0) && (d->lda % 4 == 0) && (d->sA1 % 4 == 0) && (d->sA2 % 4 == 0);
    const bool b16 = ((uintptr_t)d->B % 16 == 0) && (d->ldb % 4 == 0) && (d->sB1 % 4 == 0) && (d->sB2 % 4 == 0);
    a.vecA = a16 ? 1 : 0;
    a.vecB = b16 ? 1 : 0;
    a.a_bytes = a.b_bytes = 0;
    a.cS = d->cS; a.cOH = d->cOH; a.cOW = d->cOW; a.cH = d->cH; a.cW = d->cW;
    a.ones_row = d->ones_row;
    a.stats = d->stats;
    a.drop_thr = 0u; a.drop_scale = 1.f; a.drop_seed = 0ull; a.drop_seed_dev = nullptr;
    a.amax_a = d->amax_a; a.amax_b = d->amax_b;
    for (int i = 0; i < 3; ++i) { a.Aseg[i] = nullptr; a.ldaseg[i] = 0; a.kseg[i] = 0; a.aseg_bytes[i] = 0; }
    bool seg = false;
    if (d->a_seg[0] != nullptr) {
        // A = [A | a_seg[0] | a_seg[1] | a_seg[2]] along K: plain row-major pieces (a_mode 0, no im2col), b_mode 1, one batch slice,
        // no split-K, every boundary a multiple of 32, 16-byte aligned pieces; the split engines' fast path only
        seg = true;
        int prev = 0;
        for (int i = 0; i < 3; ++i) {
            if (d->a_seg[i] == nullptr) {
                for (int j = i; j < 3; ++j) if (d->a_seg[j] != nullptr) return RIH_EINVAL;
                break;
            }
            const int k0 = d->k_seg[i], k1 = (i < 2 && d->a_seg[i + 1] != nullptr) ? d->k_seg[i + 1] : d->K;
            if (k0 <= prev || k0 % 32 != 0 || k1 <= k0 || d->lda_seg[i] < k1 - k0 || d->lda_seg[i] % 4 != 0 ||
                ((uintptr_t)d->a_seg[i] % 16) != 0)
                return RIH_EINVAL;
            const long long bytes = ((long long)(d->M - 1) * d->lda_seg[i] + (k1 - k0)) * 4ll;
            if (bytes >= (1ll << 31)) return RIH_EINVAL;
            a.Aseg[i] = d->a_seg[i]; a.ldaseg[i] = d->lda_seg[i]; a.kseg[i] = k0; a.aseg_bytes[i] = (unsigned)bytes;
            prev = k0;
        }
        if (d->a_mode != 0 || d->b_mode != 1 || d->nb1 * d->nb2 != 1 || d->splitk != 1 || d->cS > 1 || d->drop_p != 0.f ||
            d->tile > 2 || d->engine < 1 || d->lda < d->k_seg[0] ||
            !(d->KH == 1 && d->KW == 1 && d->strideA == 1 && d->padH == 0 && d->padW == 0 && d->H == d->Ho && d->W == d->Wo))
            return RIH_EINVAL;
    }
    if (d->drop_p != 0.f) {
        if (!(d->drop_p > 0.f && d->drop_p < 1.f)) return RIH_EINVAL;
        double t = (double)d->drop_p * 4294967296.0;            // = drop_thresh() of csrc/rih_elem.hip
        if (t > 4294967295.0) t = 4294967295.0;
        a.drop_thr = (unsigned)t;
        a.drop_scale = 1.f / (1.f - d->drop_p);
        a.drop_seed = d->drop_seed;
        a.drop_seed_dev = (const unsigned long long*)d->drop_seed_dev;
    }
    {
        const auto al4 = [](long long v) { return (v & 3) == 0; };
        a.epi_vec = ((uintptr_t)d->C % 16 == 0) && al4(d->ldc) && al4(d->N) && al4(d->sC1) && al4(d->sC2) && al4(d->sCsplit) &&
                    (d->R == nullptr || (((uintptr_t)d->R % 16 == 0) && al4(d->ldr) && al4(d->sR1))) &&
                    (d->bias == nullptr || (((uintptr_t)d->bias % 16 == 0) && al4(d->sBias1)));
    }
    if (d->ones_row != 0 && (d->a_mode != 1 || d->ones_row < 0 || d->ones_row >= d->M)) return RIH_EINVAL;
    if (d->cS > 1 && (d->splitk != 1 || d->a_mode == 1 || d->R != nullptr || d->cH < 1 || d->cW < 1 || d->cOH < 0 ||
                      d->cOW < 0 || d->nb1 * d->nb2 != 1))
        return RIH_EINVAL;
    int bm = 128, bn = 128;
    if (d->tile == 1) { bm = 128; bn = 64; }
    else if (d->tile == 2) { bm = 64; bn = 64; }
    else if (d->tile == 3) { bm = 128; bn = 32; }
    else if (d->tile != 0) return RIH_EINVAL;
    const long long tiles = (long long)((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn);
    const long long gz = (long long)d->nb1 * d->nb2 * d->splitk;
    if (tiles > 0x7fffffffLL || gz > 65535) return RIH_EINVAL;
    dim3 grid((unsigned)tiles, 1, (unsigned)gz);
    hipStream_t s = (hipStream_t)stream;
    if (d->engine < 0 || d->engine > 2) return RIH_EINVAL;
    // engine 2 exists on the split engines' fast path only (tiles 0..2, operands converted by the kernel): anything else that
    // asks for it runs engine 1 -- same fp32-grade result, the six-product arithmetic (rih_gemm_engine tells in advance)
    const bool e2 = d->engine == 2 && d->tile <= 2;
    const int engine = d->engine == 2 ? 1 : d->engine;
    if (engine_out != nullptr) *engine_out = engine;
    if (engine == 1 && d->tile != 3 && a16 && b16 && d->upS == 1 && d->K % 4 == 0) {
        // fast path of the split engine (see gemm_split_kernel for the preconditions)
        const bool plain = (d->KH == 1 && d->KW == 1 && d->strideA == 1 && d->padH == 0 && d->padW == 0 &&
                            d->H == d->Ho && d->W == d->Wo);
        const long long rowsA = (d->a_mode != 1) ? (long long)d->M : (long long)d->K;
        const long long imgs = (rowsA + (long long)d->Ho * d->Wo - 1) / ((long long)d->Ho * d->Wo);
        const int colsA = seg ? d->k_seg[0] : (d->a_mode != 1) ? d->K : (d->ones_row > 0 ? d->ones_row : d->M);
        const long long a_rows = plain ? rowsA : imgs * d->H * d->W;
        const long long a_bytes = plain ? ((rowsA - 1) * d->lda + colsA) * 4ll : a_rows * (long long)d->lda * 4ll;
        const long long rowsB = (d->b_mode == 0) ? (long long)d->K : (long long)d->N;
        const long long b_bytes = ((rowsB - 1) * d->ldb + ((d->b_mode == 0) ? d->N : d->K)) * 4ll;
        bool ok = a_bytes < (1ll << 31) && b_bytes < (1ll << 31) && d->K >= 1;
        if (d->a_mode != 1 && !plain) ok = ok && (d->Cin % 32 == 0) && (d->KH * d->KW <= 32);
        if (d->a_mode == 1) ok = ok && (d->M % 4 == 0) && (plain || (d->Wo % 4 == 0 && d->Cin % 4 == 0));
        if (d->b_mode == 0) ok = ok && (d->N % 4 == 0);
        if (d->stats != nullptr && !(ok && d->tile <= 2 && d->a_mode == 0 && d->splitk == 1 && gz == 1 &&
                                     d->cS <= 1))
            return RIH_EINVAL;      // the statistics epilogue exists on this path only (rih_gemm_stats_rows tells in advance)
        // the dropout epilogue exists for plain row-major GEMMs on this path only (rih_gemm_dropout_ok tells in advance)
        if (d->drop_p != 0.f && !(ok && d->tile <= 2 && d->a_mode == 0 && plain && d->splitk == 1 &&
                                  d->cS <= 1 && d->stats == nullptr && !(d->relu && d->R != nullptr) && prep == nullptr))
            return RIH_EINVAL;
        if (stats_rows != nullptr) {
            *stats_rows = (ok && d->tile <= 2 && d->a_mode == 0 && d->splitk == 1 && gz == 1 && d->cS <= 1)
                              ? bm / 2 : 0;
            return 0;
        }
        if (seg && (!ok || prep != nullptr)) return RIH_EINVAL;                      // (no other kernel reads a segmented A)
        if (ok) {
            a.a_bytes = (unsigned)a_bytes;
            a.b_bytes = (unsigned)b_bytes;
            if (engine_out != nullptr) *engine_out = e2 ? 2 : 1;
            if (prep != nullptr) {
                if (d->stats != nullptr) return RIH_EINVAL;
                prep->a = a;
                prep->gx = (int)grid.x; prep->gz = (int)grid.z;
                prep->tile = d->tile; prep->a_mode = d->a_mode; prep->b_mode = d->b_mode; prep->plain = plain ? 1 : 0;
                prep->engine = e2 ? 2 : 1;
                return 0;
            }
            if (engine_out != nullptr) return 0;                       // rih_gemm_engine: a query, no launch
            if (e2) {
                if (d->tile == 0) return launch_split_e2<128, 128>(a, d->a_mode, d->b_mode, plain, grid, s);
                if (d->tile == 1) return launch_split_e2<128, 64>(a, d->a_mode, d->b_mode, plain, grid, s);
                return launch_split_e2<64, 64>(a, d->a_mode, d->b_mode, plain, grid, s);
            }
            if (d->tile == 0) return launch_split<128, 128>(a, d->a_mode, d->b_mode, plain, grid, s);
            if (d->tile == 1) return launch_split<128, 64>(a, d->a_mode, d->b_mode, plain, grid, s);
            return launch_split<64, 64>(a, d->a_mode, d->b_mode, plain, grid, s);
        }
    }
    if (d->drop_p != 0.f || seg) return RIH_EINVAL;           // the general kernels have no dropout epilogue / segmented A
    if (stats_rows != nullptr) { *stats_rows = 0; return 0; }
    if (prep != nullptr) return RIH_EINVAL;                   // not a fast-path descriptor: no grouped launch
    if (d->stats != nullptr) return RIH_EINVAL;
    if (engine_out != nullptr) {                              // rih_gemm_engine: a query, no launch
        if (d->tile == 3) *engine_out = 0;
        return 0;
    }
    if (d->tile == 0) return launch_tile<128, 128>(a, d->a_mode, d->b_mode, engine, grid, s);
    if (d->tile == 1) return launch_tile<128, 64>(a, d->a_mode, d->b_mode, engine, grid, s);
    if (d->tile == 3) return launch_tile<128, 32>(a, d->a_mode, d->b_mode, 0, grid, s);
    return launch_tile<64, 64>(a, d->a_mode, d->b_mode, engine, grid, s);
}

extern "C" int rih_gemm(const rih_gemm_desc* d, void* stream) { return gemm_impl(d, stream, nullptr); }

// ---- grouped launch.  Variant id = tile * 8 + a_mode * 4 + b_mode * 2 + plain of the split engine's fast path; the variants
// instantiated for the grouped kernel are the weight-gradient ones (a_mode 1, b_mode 0; 64x64 and 128x128 tiles) and the
// forward-type 64x64 ones.
extern "C" int rih_gemm_multi_variant(const rih_gemm_desc* d) {
    PreparedGemm pg;
    if (gemm_impl(d, nullptr, nullptr, &pg) != 0) return -1;
    const int v = pg.tile * 8 + pg.a_mode * 4 + pg.b_mode * 2 + pg.plain;
    switch (v) {
        case 2 * 8 + 4 + 0 + 1: case 2 * 8 + 4 + 0 + 0:         // 64x64 weight gradients (plain / conv gather)
        case 0 * 8 + 4 + 0 + 1: case 0 * 8 + 4 + 0 + 0:         // 128x128 weight gradients
            return v + (pg.engine == 2 ? 64 : 0);               // + 64: the same variant on engine 2
        case 1 * 8 + 4 + 0 + 1: case 1 * 8 + 4 + 0 + 0:         // 128x64 weight gradients (round 6: <= 64 output channels), engine 2 only
            return pg.engine == 2 ? v + 64 : -1;
        default: return -1;
    }
}

// The engine rih_gemm would run `d` on: 0 / 1 / 2 (a descriptor asking for engine 2 off the fast path runs engine 1), or -1.
extern "C" int rih_gemm_engine(const rih_gemm_desc* d) {
    int e = -1;
    if (gemm_impl(d, nullptr, nullptr, nullptr, &e) != 0) return -1;
    return e;
}

extern "C" int64_t rih_gemm_multi_table_bytes(const rih_gemm_desc* descs, int n) {
    if (n < 1 || !descs) return 0;
    long long groups = 0;
    for (int i = 0; i < n; ++i) {
        PreparedGemm pg;
        if (gemm_impl(&descs[i], nullptr, nullptr, &pg) != 0) return 0;
        groups += ((long long)pg.gx * pg.gz + 7) / 8;
    }
    return (int64_t)(sizeof(MultiHeader) + ((groups * 2 + 15) & ~15ll) + (long long)n * sizeof(MultiProb));
}

// Fill `host_table` (rih_gemm_multi_table_bytes bytes of HOST memory) for n descriptors of one variant; returns the variant
// (>= 0) and the launch's block count, or a negative error.  The caller copies the table to the device (any stream-ordered
// copy in front of the launch; a pinned staging buffer under stream capture) and calls rih_gemm_multi_launch.
extern "C" int rih_gemm_multi_pack(const rih_gemm_desc* descs, int n, void* host_table, int32_t* total_blocks) {
    if (n < 1 || n > 65535 || !descs || !host_table || !total_blocks) return -RIH_EINVAL;
    int variant = -1;
    long long groups = 0;
    for (int i = 0; i < n; ++i) {
        const int v = rih_gemm_multi_variant(&descs[i]);
        if (v < 0 || (variant >= 0 && v != variant)) return -RIH_EINVAL;
        variant = v;
        PreparedGemm pg;
        gemm_impl(&descs[i], nullptr, nullptr, &pg);
        groups += ((long long)pg.gx * pg.gz + 7) / 8;
    }
    if (groups * 8 > 0x7fffffffLL) return -RIH_EINVAL;
    unsigned char* t = (unsigned char*)host_table;
    MultiHeader* hd = (MultiHeader*)t;
    unsigned short* grp = (unsigned short*)(t + sizeof(MultiHeader));
    MultiProb* pr = (MultiProb*)(t + sizeof(MultiHeader) + ((groups * 2 + 15) & ~15ll));
    long long g = 0;
    for (int i = 0; i < n; ++i) {
        PreparedGemm pg;
        gemm_impl(&descs[i], nullptr, nullptr, &pg);
        const long long ng = ((long long)pg.gx * pg.gz + 7) / 8;
        pr[i].a = pg.a;
        pr[i].first = (int)(g * 8);
        pr[i].gx = pg.gx;
        pr[i].gz = pg.gz;
        pr[i].pad = 0;
        for (long long k = 0; k < ng; ++k) grp[g + k] = (unsigned short)i;
        g += ng;
    }
    for (long long k = g; k < ((groups * 2 + 15) & ~15ll) / 2; ++k) grp[k] = 0;
    hd->n = n; hd->total_blocks = (int)(groups * 8); hd->groups = (int)groups; hd->pad = 0;
    *total_blocks = (int)(groups * 8);
    return variant;
}

extern "C" int rih_gemm_multi_launch(const void* dev_table, int variant, int total_blocks, void* stream) {
    if (!dev_table || total_blocks < 8 || (total_blocks & 7)) return RIH_EINVAL;
    const unsigned char* t = (const unsigned char*)dev_table;
    dim3 grid((unsigned)total_blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 2 * 8 + 4 + 1: hipLaunchKernelGGL((gemm_split_multi_kernel<64, 64, 1, 0, true>), grid, block, 0, s, t); break;
        case 2 * 8 + 4 + 0: hipLaunchKernelGGL((gemm_split_multi_kernel<64, 64, 1, 0, false>), grid, block, 0, s, t); break;
        case 0 * 8 + 4 + 1: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 128, 1, 0, true>), grid, block, 0, s, t); break;
        case 0 * 8 + 4 + 0: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 128, 1, 0, false>), grid, block, 0, s, t); break;
        case 64 + 2 * 8 + 4 + 1: hipLaunchKernelGGL((gemm_split_multi_kernel<64, 64, 1, 0, true, 2>), grid, block, 0, s, t); break;
        case 64 + 2 * 8 + 4 + 0: hipLaunchKernelGGL((gemm_split_multi_kernel<64, 64, 1, 0, false, 2>), grid, block, 0, s, t); break;
        case 64 + 0 * 8 + 4 + 1: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 128, 1, 0, true, 2>), grid, block, 0, s, t); break;
        case 64 + 0 * 8 + 4 + 0: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 128, 1, 0, false, 2>), grid, block, 0, s, t); break;
        case 64 + 1 * 8 + 4 + 1: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 64, 1, 0, true, 2>), grid, block, 0, s, t); break;
        case 64 + 1 * 8 + 4 + 0: hipLaunchKernelGGL((gemm_split_multi_kernel<128, 64, 1, 0, false, 2>), grid, block, 0, s, t); break;
        default: return RIH_EINVAL;
    }
    return (int)hipGetLastError();
}

// 1 when rih_gemm would run `d` with a dropout epilogue (drop_p of the descriptor is ignored: 0.5 is assumed when it is 0)
extern "C" int rih_gemm_dropout_ok(const rih_gemm_desc* d) {
    if (!d) return 0;
    rih_gemm_desc c = *d;
    if (c.drop_p == 0.f) c.drop_p = 0.5f;
    c.stats = nullptr;
    int rows = 0;
    return gemm_impl(&c, nullptr, &rows) == 0 ? 1 : 0;
}

extern "C" int rih_gemm_stats_rows(const rih_gemm_desc* d) {
    int rows = 0;
    if (gemm_impl(d, nullptr, &rows) != 0) return 0;
    return rows;
}
extern "C" int rih_splitk_reduce_bias_batched(const float* P, int S, int Mp, int M, int N, float* dst, int Cin, int taps,
                                              int CinValid, int accumulate, float* db, int nb, int64_t sP, int64_t sDst,
                                              int64_t sDb, void* stream) {
    if (!P || !dst || S < 1 || M < 1 || Mp < M || N < 1 || Cin < 1 || taps < 1 || CinValid < 1) return RIH_EINVAL;
    if (db && Mp < M + 1) return RIH_EINVAL;
    if (nb < 1 || nb > 65535) return RIH_EINVAL;
    const long long tiles = (long long)((M + 7) / 8) * ((N + 31) / 32);
    const long long blocks = tiles + (db ? (N + 255) / 256 : 0);
    if (blocks > 0x7fffffffLL) return RIH_EINVAL;
    hipLaunchKernelGGL(splitk_reduce_fused_kernel, dim3((unsigned)blocks, (unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       P, S, Mp, M, N, dst, Cin, taps, CinValid, accumulate, db, (int)tiles, (long long)sP,
                       (long long)sDst, (long long)sDb);
    return (int)hipGetLastError();
}

extern "C" int rih_splitk_reduce_multi(const rih_reduce_desc* descs, int n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return RIH_EINVAL;
    for (int i = 0; i < n; ++i) {
        const rih_reduce_desc& d = descs[i];
        if (!d.P || !d.dst || d.S < 1 || d.M < 1 || d.Mp < d.M || d.N < 1 || d.Cin < 1 || d.taps < 1 || d.CinValid < 1)
            return RIH_EINVAL;
        if (d.db && d.Mp < d.M + 1) return RIH_EINVAL;
        if (d.CinPitch != 0 && d.CinPitch < d.CinValid) return RIH_EINVAL;
    }
    for (int base = 0; base < n; base += REDUCE_PACK) {
        ReducePack pk;
        pk.n = (n - base < REDUCE_PACK) ? n - base : REDUCE_PACK;
        long long total = 0;
        for (int i = 0; i < pk.n; ++i) {
            const rih_reduce_desc& d = descs[base + i];
            const long long tiles = (long long)((d.M + 7) / 8) * ((d.N + 31) / 32);
            pk.d[i] = d;
            pk.ntiles[i] = (int)tiles;
            pk.first[i] = (int)total;
            total += tiles + (d.db ? (d.N + 255) / 256 : 0);
            if (total > 0x7fffffffLL) return RIH_EINVAL;
        }
        pk.first[pk.n] = (int)total;
        hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, pk);
    }
    return (int)hipGetLastError();
}

extern "C" int rih_splitk_reduce_bias(const float* P, int S, int Mp, int M, int N, float* dst, int Cin, int taps,
                                      int CinValid, int accumulate, float* db, void* stream) {
    return rih_splitk_reduce_bias_batched(P, S, Mp, M, N, dst, Cin, taps, CinValid, accumulate, db, 1, 0, 0, 0, stream);
}

extern "C" int rih_splitk_reduce(float* P, int S, int M, int N, float* dst, int Cin, int taps, int CinValid,
                                 int accumulate, void* stream) {
    return rih_splitk_reduce_bias(P, S, M, M, N, dst, Cin, taps, CinValid, accumulate, nullptr, stream);
}

extern "C" int rih_splitk_finish(const float* P, int S, int M, int N, float* C, int ldc, const float* bias,
                                 const float* R, int ldr, float alpha, int relu, void* stream) {
    if (!P || !C || S < 1 || M < 1 || N < 1 || ldc < N || (R && ldr < N)) return RIH_EINVAL;
    const long long total = (long long)M * N;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, P, S, M, N, C, ldc, bias, R,
                       ldr, alpha, relu);
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
