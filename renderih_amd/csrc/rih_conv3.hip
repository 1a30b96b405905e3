// rih_conv3.hip -- halo-resident 3x3 convolution (stride 1, padding 1) for gfx950 on rih_gemm's engine-2 arithmetic.
//
// Why (round-5).  The implicit-GEMM kernels of rih_gemm.hip walk a 3x3 convolution tap by tap: every one of the nine
// k-tiles of a 32-channel chunk fetches the 128 (or 256) A rows of the workgroup again from L2 -- the same input pixels, shifted by
// one -- and converts them again (fp32 -> two scaled fp16 planes, six VALU instructions per pair).  Round 4 measured those
// kernels bound by the SUM of MFMA, conversion VALU and LDS issue plus ~8 TB/s of L2 -> LDS operand traffic (2.4 GB per launch for
// 268 MB of compulsory bytes on the 64x64 128 -> 128 convolution).  Here a workgroup owns an 8 x 32 pixel patch of one image
// and BN output channels; per 32-channel chunk it loads the (8+2) x (32+2) input halo ONCE, converts it ONCE into the two fp16
// planes in LDS, and runs the nine taps' MFMAs on shifted windows of that LDS image: A traffic and A conversion per FLOP drop
// 6.8-fold (340 halo pixels for 9 x 256 operand rows).  The weights arrive PRE-SPLIT (two fp16 planes, interleaved per 8 k:
// "H2", written once per step by rih_h2_conv_weight / rih_h2_multi) and are staged global -> LDS by LDS-DMA: no VGPR round
// trip, no ds_write, no conversion for B at all.  One barrier per k-tile; B double-buffered, A double-buffered per chunk.
//
// Arithmetic = engine 2 of rih_gemm.hip, bit for bit per product: operands scaled by powers of two derived from device-resident
// bound blocks (e2_scale), x = hi + 2^-11 lo with fp16 hi / lo, three v_mfma_f32_32x32x16_f16 per 32x32x16 block (hi*hi into
// acc0, lo*hi + hi*lo into acc1), result (acc0 + 2^-11 acc1) / (s_a s_b).  The summation ORDER over k differs from the tap-major
// implicit GEMM (here chunk-major: (c / 32, tap, c % 32)), so results agree to fp32 round-off, not bitwise.
//
// Geometry: 512 threads = 8 wavefronts; a patch is 256 pixels of one image -- 8 rows x 32 pixels (W % 32 == 0) or 16 x 16 (maps
// whose width is a multiple of 16 only: the 16 x 16 maps of ResNet layer3 / HRNet branch 2) -- times BN = 128 / 64 / 32 output
// channels: waves 4 (M) x 2 (N) with wave tile 64 x BN/2 for BN >= 64, waves 8 x 1 with wave tile 32 x 32 for BN = 32 (HRNet's
// 32-channel branch).  LDS: A 2 x 340 x 128 B = 85 KB, B 2 x BN x 128 B <= 32 KB: one workgroup (two waves per SIMD) per CU.
// LDS images: a pixel (or a weight row) is 8 units of 16 bytes -- unit j = (k / 8) * 2 + plane -- stored at position
// j ^ ((halo column >> 1) & 7) (weights: j ^ ((row >> 1) & 7)): the ds_read_b128 operand fetches (32 consecutive pixels of one
// image row, or 2 x 16 of two rows; 32 consecutive weight rows; one unit index per half wave) touch 16 distinct bank groups per
// 16-lane group for every tap shift (checked exhaustively by tests/test_kernels_on_cpu.py::test_conv3_lds_image_is_conflict_free).
//
// Preconditions (rih_conv3x3_ok): C % 32 == 0, N % 32 == 0, (H % 8 == 0 and W % 32 == 0) or (H % 16 == 0 and W % 16 == 0),
// 16-byte aligned operands, pitches % 4 == 0, one image < 2 GiB.  Epilogue: optional ReLU, optional BatchNorm statistics per 64-row wave block ((mean, M2), the format of
// rih_gemm_desc.stats: rih_bn_stats_from_blocks merges them), or an optional residual added before the ReLU (RES variants, ABI 19:
// the skip gradient in the data gradient of a BasicBlock's first convolution).  No bias (no 3x3 convolution of the network has one
// on the training path); callers with one use rih_gemm.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/renderih_amd.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int A_STAGE = 340 * 128;                                       // bytes: the larger halo (10 x 34), 8 units of 16 B per pixel
constexpr int NT = 512;
constexpr unsigned OOB = 0x80000000u;
constexpr int SLD = 36;                                                  // epilogue staging pitch (floats)

struct C3Args {
    const float* x;
    const unsigned char* w;     // H2 planes [N][Kp / 8][2][8 halves]
    float* y;
    float* stats;
    const float* amax_x;
    const float* amax_w;
    int imgs, H, W, C, N, ldx, ldy, Kp, relu;
    int tiles_x, tiles_y, nblk;
    const float* r;             // RES: residual [imgs][H][W][ldr], added before the ReLU (the skip gradient of a BasicBlock's first conv)
    int ldr;
};

__device__ __forceinline__ int xcd_remap_c3(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// the same scale derivation as rih_gemm.hip's e2_scale (bound block: 64 partial maxima, one per 128-byte line)
__device__ __forceinline__ float c3_scale(const float* amax) {
    if (amax == nullptr) return 1.f;
    float a = amax[(threadIdx.x & 63) * (RIH_BOUND_FLOATS / 64)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
    a = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a)));
    const int e = (int)((__float_as_uint(a) >> 23) & 0xffu);
    if (e == 0 || e == 255) return 1.f;
    int se = 268 - e;
    se = se > 253 ? 253 : se;
    return __uint_as_float((unsigned)se << 23);
}

__device__ __forceinline__ unsigned c3_pk_f16(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
// (a, b) * s -> packed fp16 hi pair and packed fp16 pair of the 2^11-scaled residuals (rih_gemm.hip split2h)
__device__ __forceinline__ void c3_split2h(float a, float b, float s, unsigned& h, unsigned& l) {
#if defined(__HIP_DEVICE_COMPILE__)
    float ra, rb;
    const float k2048 = 2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "s"(s), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "s"(s), "v"(h));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(ra), "s"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(rb), "s"(k2048));
#else
    a *= s;
    b *= s;
    const f16x2 hv = {(_Float16)a, (_Float16)b};
    h = __builtin_bit_cast(unsigned, hv);
    l = c3_pk_f16((a - (float)hv.x) * 2048.f, (b - (float)hv.y) * 2048.f);
#endif
}

// c3_split2h pinned in program order (volatile): rows_kernel converts the rows of k-tile j + 1 BEHIND its counted wait -- hoisted
// into the MFMA phase (where the scheduler likes to put it) the conversion needs those rows at the top of the iteration and the
// prefetch is one k-tile deep again
__device__ __forceinline__ void c3_split2h_pinned(float a, float b, float s, unsigned& h, unsigned& l) {
#if defined(__HIP_DEVICE_COMPILE__)
    float ra, rb;
    const float k2048 = 2048.f;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(s));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(s));
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "s"(s), "v"(h));
    asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "s"(s), "v"(h));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(l) : "v"(ra), "s"(k2048));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(l) : "v"(rb), "s"(k2048));
#else
    c3_split2h(a, b, s, h, l);
#endif
}

__device__ __forceinline__ void c3_glds16(const unsigned char* src, unsigned char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// LDS-DMA data is published to the other wavefronts by a barrier only once the requesting wavefront's vmcnt has drained: hipcc
// places that wait in front of __syncthreads() today, but nothing guarantees it -- every barrier that publishes
// DMA data is preceded by this explicit wait (CK's block_sync_lds_direct_load does the same).
__device__ __forceinline__ void c3_dma_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// The same LDS-DMA instruction issued OUTSIDE hipcc's wait-count bookkeeping (rows_kernel): behind the builtin the compiler puts
// `s_waitcnt vmcnt(0)` in front of the next barrier / LDS read that may alias the destination, which also drains every register
// load in flight -- a prefetch of A rows two k-tiles ahead would be cut back to one.  Here the kernel places its waits itself
// (c3_vm_wait<N>()).  N = 0 everywhere since round 6: the design counted ("at most N outstanding" = everything but the N youngest
// has landed), which holds among register loads but NOT between register loads and LDS-DMA requests on gfx950 -- see the schedule
// comment in rows_kernel.  What the explicit placement still buys is ONE wait per k-tile, behind the MFMAs, instead of one in
// front of every barrier and LDS read the compiler cannot prove independent.  The compiler's own waits for register loads stay
// correct: operations it does not know of can only make its counts conservative.
__device__ __forceinline__ void c3_glds16_raw(const unsigned char* src, unsigned char* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_dst);      // lane 0's address; lane l lands at + 16 l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(base), "v"(src) : "memory");      // (m0: nothing else in rows_kernel uses it; hipcc rejects it as a clobber)
#else
    c3_glds16(src, lds_dst);
#endif
}
template <int N>
__device__ __forceinline__ void c3_vm_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

__device__ __forceinline__ float4 c3_bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// A 16-byte buffer load outside hipcc's wait-count bookkeeping, like c3_glds16_raw: the destination registers are the kernel's to
// wait for (c3_vm_wait) before their first use.  `rs` = the four words of a raw buffer resource (base, base_hi, num_records, flags);
// an offset beyond num_records returns zeros (the hardware range check, as with the builtin).
typedef float c3_f32x4 __attribute__((ext_vector_type(4)));
typedef int c3_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 c3_bload4_raw(c3_i32x4 rs, __amdgpu_buffer_rsrc_t r, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
    c3_f32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs) : "memory");
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return c3_bload4(r, off);
#endif
}

template <int TW, int BN, bool STATS, bool RES = false>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(const C3Args p) {
    static_assert(!(STATS && RES), "a launch has statistics (forward) or a residual (data gradient), not both");
    constexpr int TH = 256 / TW, HW_ = TW + 2, HP = (TH + 2) * HW_;     // patch 8 x 32 (halo 340 pixels) or 16 x 16 (324)
    constexpr int WGN = BN >= 64 ? 2 : 1, WGM = 8 / WGN;                // waves 4 x 2, or 8 x 1 for 32 output channels
    constexpr int TM = 8 / WGM;                                         // 32-row blocks per wave: 2 / 1
    constexpr int TN = BN / (32 * WGN);                                 // 32-column blocks per wave: 2 / 1 / 1
    constexpr int B_STAGE = BN * 128;
    constexpr int NPA = (HP * 8 + NT - 1) / NT;         // float4 quads of a halo chunk per thread: 6 (the last pass partial)
    constexpr int NPB = (BN * 8 + NT - 1) / NT;         // LDS-DMA units of a weight k-tile per thread: 2 / 1 / 1 (BN 32: waves 0-3 only)
    static_assert((BN * 8) % 64 == 0 && HP * 128 <= A_STAGE, "whole wave instructions; the halo fits its stage");
    constexpr int SMEM = 2 * A_STAGE + 2 * B_STAGE;
    static_assert(SMEM >= 8 * 32 * SLD * 4, "epilogue staging fits");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * A_STAGE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;

    // workgroup -> (image, patch, channel block); channel blocks fastest so that workgroups sharing a halo are neighbours
    const int bid = xcd_remap_c3((int)blockIdx.x, (int)gridDim.x);
    const int nb = bid % p.nblk;
    int rest = bid / p.nblk;
    const int tx_t = rest % p.tiles_x;
    rest /= p.tiles_x;
    const int ty_t = rest % p.tiles_y;
    const int img = rest / p.tiles_y;
    const int y0 = ty_t * TH, x0 = tx_t * TW, n0 = nb * BN;
    // 32-row MFMA block blk (0..7) of the patch, lane l31 -> pixel (ty, tx): one image row of 32, or two rows of 16
    const int lty = (TW == 32) ? 0 : (l31 >> 4), ltx = l31 & (TW - 1);

    const float sa = c3_scale(p.amax_x), sb = c3_scale(p.amax_w);

    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (long long)img * p.H * p.W * p.ldx), (short)0, (int)((long long)p.H * p.W * p.ldx * 4), 0x00020000);

    // ---------------------------------------------------------------- loader constants
    // A: quad q = pass * NT + tid -> halo pixel hp = q / 8, channel quad cq = q % 8 of the 32-channel chunk
    unsigned a_goff[NPA];       // byte offset of (pixel, channel quad) in the image, OOB for the zero padding / idle lanes
    int a_lds[NPA];             // LDS byte offset of the hi half-unit (the lo half-unit is at offset ^ 16), -1: idle lane
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = i * NT + tid;
        const int hp = q >> 3, cq = q & 7;
        a_goff[i] = OOB;
        a_lds[i] = -1;
        if (hp < HP) {
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                a_goff[i] = (unsigned)(((y * p.W + x) * p.ldx + 4 * cq) * 4);
            a_lds[i] = hp * 128 + ((((cq >> 1) * 2) ^ ((hx >> 1) & 7)) << 4) + (cq & 1) * 8;
        }
    }
    // B: unit U = pass * NT + tid -> weight row n = U / 8, LDS position U % 8 holds source unit j = pos ^ ((n >> 1) & 7)
    const unsigned char* b_src[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int U = (i * NT + tid) % (BN * 8);                                    // (BN 32: the idle upper half never issues)
        const int n = U >> 3, j = (U & 7) ^ ((n >> 1) & 7);
        b_src[i] = p.w + ((long long)(n0 + n) * (p.Kp >> 3)) * 32 + j * 16;         // + (k / 8) * 32 per k-tile
    }
    // operand fetch: A pixel rows of this wave: blocks blk = wm * TM + i; B rows wn * (BN / WGN) + jj * 32 + l31
    int b_rd[2][2][TN];         // [k-step][plane][block]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const int n = wn * (BN / WGN) + jj * 32 + l31;
                b_rd[s][pl][jj] = n * 128 + (((((2 * s + lhi) * 2) + pl) ^ ((n >> 1) & 7)) << 4);
            }
    // halo pixel of block 0 of this wave, lane l31, at tap (0, 0); block i adds i * BLKROWS * HW_
    constexpr int BLKROWS = 32 / TW;                    // image rows per 32-row block: 1 / 2
    const int a_hp0 = (wm * TM * BLKROWS + lty) * HW_ + ltx;

    float4 areg[NPA];
    auto load_A = [&](int c0) {                         // global -> registers: the halo of channels [c0, c0 + 32)
        const unsigned add = (unsigned)c0 * 4u;
#pragma unroll
        for (int i = 0; i < NPA; ++i) areg[i] = c3_bload4(rX, a_goff[i] == OOB ? OOB : a_goff[i] + add);
    };
    auto store_A = [&](unsigned char* dst) {            // registers -> two fp16 planes in LDS
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            if (a_lds[i] < 0) continue;
            unsigned h0, l0, h1, l1;
            c3_split2h(areg[i].x, areg[i].y, sa, h0, l0);
            c3_split2h(areg[i].z, areg[i].w, sa, h1, l1);
            *reinterpret_cast<uint2*>(dst + a_lds[i]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + (a_lds[i] ^ 16)) = make_uint2(l0, l1);
        }
    };
    auto issue_B = [&](int kofs, unsigned char* dst) {  // LDS-DMA: weight rows [n0, n0 + BN) x k [kofs, kofs + 32)
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            if ((i + 1) * NT <= BN * 8 || i * NT + tid < BN * 8)                    // (wave-uniform: BN * 8 is a multiple of 64)
                c3_glds16(b_src[i] + (long long)(kofs >> 3) * 32, dst + (i * NT + tid) * 16);
    };

    floatx16 acc[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int nchunk = p.C / 32;
    const int ntile = nchunk * 9;
    // prologue: halo of chunk 0 converted into A stage 0, weights of k-tile 0 in flight into B stage 0
    load_A(0);
    issue_B(0, Bbuf);
    store_A(Abuf);
    c3_dma_wait();
    __syncthreads();
    int kt = 0;
    for (int c = 0; c < nchunk; ++c) {
        const unsigned char* As = Abuf + (c & 1) * A_STAGE;
#pragma unroll 1
        for (int t = 0; t < 9; ++t, ++kt) {
            // k-tile kt = (chunk c, tap t): its weights have landed in B stage kt & 1, the halo of chunk c is in A stage c & 1
            if (kt + 1 < ntile) {
                const int t1 = (t == 8) ? 0 : t + 1, c1 = (t == 8) ? c + 1 : c;
                issue_B(t1 * p.C + c1 * 32, Bbuf + ((kt + 1) & 1) * B_STAGE);
            }
            if (t == 0 && c + 1 < nchunk) load_A((c + 1) * 32);          // lands during this chunk's taps
            const unsigned char* Bs = Bbuf + (kt & 1) * B_STAGE;
            const int kh = t / 3, kw = t - kh * 3;
            const int hp_t = a_hp0 + kh * HW_ + kw;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 av[2][TM], bv[2][TN];
                const int sw = ((ltx + kw) >> 1) & 7;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int base = (hp_t + i * BLKROWS * HW_) * 128;
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        av[pl][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(
                                                                  As + base + (((((2 * s + lhi) * 2) + pl) ^ sw) << 4)));
                }
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        bv[pl][jj] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Bs + b_rd[s][pl][jj]));
#define RIH_C3_TERM(ACC_, PA_, PB_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj) ACC_[i][jj] = \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(av[PA_][i], bv[PB_][jj], ACC_[i][jj], 0, 0, 0);
                RIH_C3_TERM(acc1, 1, 0)
                RIH_C3_TERM(acc, 0, 0)
                RIH_C3_TERM(acc1, 0, 1)
#undef RIH_C3_TERM
            }
            if (t == 8 && c + 1 < nchunk) store_A(Abuf + ((c + 1) & 1) * A_STAGE);   // (that stage was last read in chunk c - 1)
            c3_dma_wait();                              // the weights of k-tile kt + 1 have landed before the barrier publishes them
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- epilogue (the loop ended with a barrier: LDS is free)
    // accumulators -> this wave's 32 x SLD floats of LDS -> each lane owns 4 consecutive columns of a row: 16-byte stores
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
    const float inv_a = 1.f / sa, inv_b = 1.f / sb;     // (applied one after the other, like rih_gemm: exact powers of two)
    float4 ssh[STATS ? TN : 1], ssum[STATS ? TN : 1], ssq[STATS ? TN : 1];
    float scnt[STATS ? TN : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            ssh[j] = make_float4(0, 0, 0, 0); ssum[j] = make_float4(0, 0, 0, 0); ssq[j] = make_float4(0, 0, 0, 0); scnt[j] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int blk = wm * TM + i;                    // 32-row block of the patch: image row blk (TW 32) or rows 2 blk, 2 blk + 1
        float* ypatch = p.y + (((long long)img * p.H + y0) * p.W + x0) * p.ldy + n0 + wn * (BN / WGN);
        const float* rpatch = RES ? p.r + (((long long)img * p.H + y0) * p.W + x0) * p.ldr + n0 + wn * (BN / WGN) : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (i + j > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaf(acc1[i][j][r], 0x1p-11f, acc[i][j][r]) * inv_a * inv_b;
                stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + l31] = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;       // row of the 32-row block -> pixel (ty, tx)
                const int ty = (TW == 32) ? blk : 2 * blk + (row >> 4), tx = row & (TW - 1);
                float4 v = *reinterpret_cast<const float4*>(stg + row * SLD + c4);
                if (RES) {
                    const float4 rr = *reinterpret_cast<const float4*>(rpatch + ((long long)ty * p.W + tx) * p.ldr + j * 32 + c4);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(ypatch + ((long long)ty * p.W + tx) * p.ldy + j * 32 + c4) = v;
                if (STATS) {
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
        // per column: (mean, centred sum of squares) of this wave's 32 TM rows; a lane holds 4 TM rows, the eight row-lanes merge
        // pairwise with Chan's formula (store_tiles_wide of rih_gemm.hip); row block index = WGM * patch + wm
        const long long rb = (((long long)img * p.tiles_y + ty_t) * p.tiles_x + tx_t) * WGM + wm;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float n = scnt[j];
            const float in = n > 0.f ? 1.f / n : 0.f;
            float4 mean = make_float4(ssh[j].x + ssum[j].x * in, ssh[j].y + ssum[j].y * in, ssh[j].z + ssum[j].z * in,
                                      ssh[j].w + ssum[j].w * in);
            float4 m2 = make_float4(ssq[j].x - ssum[j].x * ssum[j].x * in, ssq[j].y - ssum[j].y * ssum[j].y * in,
                                    ssq[j].z - ssum[j].z * ssum[j].z * in, ssq[j].w - ssum[j].w * ssum[j].w * in);
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                const float nbr = __shfl_xor(n, o, 64);
                const float nt = n + nbr;
                const float wb = nt > 0.f ? nbr / nt : 0.f;
                const float cf = n * wb;
#define RIH_C3_MERGE(c_)                                                              \
    {                                                                                 \
        const float mb = __shfl_xor(mean.c_, o, 64), qb = __shfl_xor(m2.c_, o, 64);   \
        const float dl = mb - mean.c_;                                                \
        mean.c_ += dl * wb;                                                           \
        m2.c_ += qb + dl * dl * cf;                                                   \
    }
                RIH_C3_MERGE(x) RIH_C3_MERGE(y) RIH_C3_MERGE(z) RIH_C3_MERGE(w)
#undef RIH_C3_MERGE
                n = nt;
            }
            if ((lane >> 3) == 0) {
                const int nn = n0 + wn * (BN / WGN) + j * 32 + (lane & 7) * 4;
                *reinterpret_cast<float4*>(p.stats + (rb * 2 + 0) * p.N + nn) = mean;
                *reinterpret_cast<float4*>(p.stats + (rb * 2 + 1) * p.N + nn) = m2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ H2 weight operand
// OIHW conv weight -> two scaled fp16 planes interleaved per 8 k: dst[n][k / 8][plane][8 halves] (row pitch Kp * 4 bytes), zero
// padded to Kp.  for_dgrad 0: n = co, k = (tap, ci < CinPad) -- the forward operand; for_dgrad 1: n = ci < CinPad,
// k = ((th, tw), co), taps flipped -- the data-gradient operand (stride-1 gradient: all KH x KW taps).  Scale from the weight's
// bound block, derived like the consuming kernel derives it.
struct H2WArgs {
    const float* w;
    unsigned char* dst;
    const float* amax;
    int N, K, Kp, for_dgrad;
    int Cout, Cin, KH, KW, CinPad;
};
__device__ __forceinline__ float h2w_fetch(const H2WArgs& a, int n, int k) {
    if (k >= a.K) return 0.f;
    if (!a.for_dgrad) {
        const int tap = k / a.CinPad, ci = k - tap * a.CinPad;
        return ci < a.Cin ? a.w[((long long)n * a.Cin + ci) * (a.KH * a.KW) + tap] : 0.f;
    }
    const int co = k % a.Cout, t = k / a.Cout;
    const int tw = t % a.KW, th = t / a.KW;
    const int kh = a.KH - 1 - th, kw = a.KW - 1 - tw;
    return n < a.Cin ? a.w[(((long long)co * a.Cin + n) * a.KH + kh) * a.KW + kw] : 0.f;
}
__device__ __forceinline__ void h2w_span(const H2WArgs& a, long long first, long long stride) {
    const int G = a.Kp / 8;
    const long long total = (long long)a.N * G;
    const float sc = c3_scale(a.amax);
    for (long long i = first; i < total; i += stride) {
        const int n = (int)(i / G), g = (int)(i - (long long)n * G);
        uint4 h, l;
        c3_split2h(h2w_fetch(a, n, 8 * g + 0), h2w_fetch(a, n, 8 * g + 1), sc, h.x, l.x);
        c3_split2h(h2w_fetch(a, n, 8 * g + 2), h2w_fetch(a, n, 8 * g + 3), sc, h.y, l.y);
        c3_split2h(h2w_fetch(a, n, 8 * g + 4), h2w_fetch(a, n, 8 * g + 5), sc, h.z, l.z);
        c3_split2h(h2w_fetch(a, n, 8 * g + 6), h2w_fetch(a, n, 8 * g + 7), sc, h.w, l.w);
        uint4* d = reinterpret_cast<uint4*>(a.dst + i * 32);
        d[0] = h;
        d[1] = l;
    }
}
constexpr int H2_PACK = 48;
struct H2Pack {
    H2WArgs d[H2_PACK];
    int first[H2_PACK + 1];
    int n;
};
static_assert(sizeof(H2Pack) <= 4096, "kernel argument limit");
__global__ __launch_bounds__(256) void h2_weight_multi_kernel(const H2Pack pk) {
    const int b = (int)blockIdx.x;
    int k = 0;
    while (k + 1 < pk.n && b >= pk.first[k + 1]) ++k;
    const int nb = pk.first[k + 1] - pk.first[k];
    h2w_span(pk.d[k], (long long)(b - pk.first[k]) * 256 + threadIdx.x, (long long)nb * 256);
}

int h2_args(H2WArgs& a, const rih_h2_desc& d) {
    if (!d.w || !d.dst || !d.amax || d.Cout < 1 || d.Cin < 1 || d.KH < 1 || d.KW < 1 || d.CinPad < d.Cin || d.Kpad < 32 ||
        d.Kpad % 32 != 0 || ((uintptr_t)d.dst % 16) != 0)
        return RIH_EINVAL;
    a.w = d.w; a.dst = (unsigned char*)d.dst; a.amax = d.amax; a.Kp = d.Kpad; a.for_dgrad = d.for_dgrad ? 1 : 0;
    a.Cout = d.Cout; a.Cin = d.Cin; a.KH = d.KH; a.KW = d.KW; a.CinPad = d.CinPad;
    if (!a.for_dgrad) { a.N = d.Cout; a.K = d.KH * d.KW * d.CinPad; }
    else { a.N = d.CinPad; a.K = d.KH * d.KW * d.Cout; }
    return a.Kp < a.K ? RIH_EINVAL : RIH_OK;
}


// ------------------------------------------------------------------------------------------------ short-K streaming GEMM ("panel")
// rih_panel: C[M][N] = act(A[M][K] W[N][K]^T (+ R)) for the 1x1 convolutions with a SHORT reduction (K = 64 or 128) and a large
// map -- layer1 / layer2's conv3 (64 -> 256, 128 -> 512) forward and conv1's data gradient.  These launches move 3-5 bytes per
// FLOP: they are HBM streams, and on the tiled kernels of rih_gemm.hip they ran at 1.9-2.9 TB/s inside the captured step
// (profiles/r05/step_by_grid_c7.txt: 64 -> 256 at 64x64 115 us forward, 176 us as a data gradient, for 335 / 603 MB) where the
// BatchNorm kernels beside them stream at 5.3-6.6 TB/s -- every workgroup pays its own prologue (bound blocks, first operand
// round trip, weights re-staged and re-converted per tile) for two k-tiles of work.  Here a workgroup is PERSISTENT: it keeps
// its BN-column slice of the weights (pre-split "H2" planes, 64 KB, staged once by LDS-DMA) in LDS and walks over row tiles;
// the fp32 rows of tile t + 2 are in flight (global -> registers) and tile t + 1 is being converted into the other LDS stage
// while tile t is multiplied, and the accumulators leave through the consumed stage as 16-byte stores -- nothing but two
// barriers per tile between one tile's stores and the next tile's loads.  Arithmetic: engine 2 (see conv3x3_halo_kernel).
// Geometry: 512 threads = 8 wavefronts; tile BM x BN with BM * K = 8192 (128 rows at K = 64, 64 rows at K = 128) and
// BN = 256 / 128 / 64 (BN * K <= 16384); LDS = weights 64 KB + two row stages of 32 KB; the epilogue staging aliases the stage
// that was just multiplied.  LDS images: unit j = (k / 8) * 2 + plane of a row at position j ^ (row & 15) inside its 16-unit
// group (conflict-free ds_read_b128 for 32 consecutive rows).  Epilogue: optional residual, ReLU, BatchNorm statistics per
// 32 TM rows of a wave (rih_gemm_desc.stats format).  Preconditions (rih_panel_ok): K in {64, 128}, N % 64 == 0, M % 128 == 0,
// at least 256 (tile, column block) items, 16-byte aligned operands and pitches.
struct PanelArgs {
    const float* a;
    const unsigned char* w;
    float* c;
    const float* r;
    float* stats;
    const float* amax_a;
    const float* amax_w;
    int M, N, K, lda, ldc, ldr, relu;
    int nblk, mtiles;
};

template <int KT, int BN, bool STATS, bool RES>
__global__ __launch_bounds__(NT, 2) void panel_kernel(const PanelArgs p) {
    constexpr int K = 32 * KT;                          // 64 / 128
    constexpr int BM = 8192 / K;                        // 128 / 64
    constexpr int WGN = BN >= 128 ? 4 : 2, WGM = 8 / WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    static_assert(TM >= 1 && TN >= 1 && BN * K <= 16384, "tile shape");
    constexpr int ROWB = K * 4;                         // bytes per LDS row: K / 8 groups x 2 planes x 16 B
    constexpr int UPR = K / 4;                          // 16-byte units per row
    constexpr int A_ST = BM * ROWB;                     // 32 KB
    constexpr int B_BYTES = BN * ROWB;
    constexpr int NPA = (BM * K / 4) / NT;              // float4 of a row tile per thread: 4
    constexpr int NPB = (BN * UPR) / NT;                // LDS-DMA units of the weight slice per thread
    static_assert((BM * K / 4) % NT == 0 && (BN * UPR) % NT == 0 && 8 * 32 * 32 * 4 <= A_ST, "loader / staging geometry");
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 2 * A_ST];
    unsigned char* const Bs = smem;
    unsigned char* const Abuf = smem + B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const int G = (int)gridDim.x;
    const int id = xcd_remap_c3((int)blockIdx.x, G);   // neighbours on one XCD share the row tiles (their column blocks differ)
    const int nb = id % p.nblk, n0 = nb * BN;
    const int mstep = G / p.nblk;
    int mt = id / p.nblk;
    if (mt >= p.mtiles) return;                         // (whole workgroup: before any barrier)

    const float sa = c3_scale(p.amax_a), sb = c3_scale(p.amax_w);
    const float inv_a = 1.f / sa, inv_b = 1.f / sb;

    // weights: LDS unit U = (row n, position pos) holds source unit j = (pos & ~15) | ((pos & 15) ^ (n & 15)); staged once
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int U = i * NT + tid;
        const int n = U / UPR, pos = U % UPR;
        const int j = (pos & ~15) | ((pos & 15) ^ (n & 15));
        c3_glds16(p.w + ((long long)(n0 + n) * UPR + j) * 16, Bs + U * 16);
    }
    // rows: float4 q = pass * NT + tid of a tile -> row q / (K / 4), channel quad q % (K / 4)
    int a_row[NPA], a_lds[NPA];
    unsigned a_col[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = i * NT + tid;
        const int row = q / (K / 4), cq = q % (K / 4);
        a_row[i] = row;
        a_col[i] = (unsigned)cq * 16u;
        const int j = (cq >> 1) * 2;                    // hi unit of the 8-channel group; lo = j + 1
        a_lds[i] = row * ROWB + (((j & ~15) | ((j & 15) ^ (row & 15))) << 4) + (cq & 1) * 8;
    }
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)0x7fffffff, 0x00020000);
    float4 areg[NPA];
    auto load_A = [&](int tile) {                       // (a tile past the end arrives as zeros and is never multiplied)
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const long long off = ((long long)tile * BM + a_row[i]) * p.lda * 4 + a_col[i];
            areg[i] = c3_bload4(rA, (tile < p.mtiles && off < 0x7fffffffLL) ? (unsigned)off : OOB);
        }
    };
    auto store_A = [&](unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            unsigned h0, l0, h1, l1;
            c3_split2h(areg[i].x, areg[i].y, sa, h0, l0);
            c3_split2h(areg[i].z, areg[i].w, sa, h1, l1);
            *reinterpret_cast<uint2*>(dst + a_lds[i]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + (a_lds[i] ^ 16)) = make_uint2(l0, l1);
        }
    };
    // operand fetch offsets per (k-tile, k-step, plane): unit j = 8 kt + (2 s + lhi) * 2 + pl of rows (wave base + 32 i + l31)
    const int ra = wm * (32 * TM) + l31, rb = wn * (32 * TN) + l31;

    load_A(mt);
    store_A(Abuf);
    c3_dma_wait();                                      // the weight slice has landed
    load_A(mt + mstep);
    __syncthreads();                                    // weights published, stage 0 complete
    int st = 0;
    for (; mt < p.mtiles; mt += mstep, st ^= 1) {
        const unsigned char* As = Abuf + st * A_ST;
        // tile t + 1 (in registers since the last iteration) -> the other stage; tile t + 2 -> registers
        store_A(Abuf + (st ^ 1) * A_ST);
        load_A(mt + 2 * mstep);
        floatx16 acc[TM][TN], acc1[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 av[2][TM], bv[2][TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int j = 8 * kt + (2 * s + lhi) * 2 + pl;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row = ra + 32 * i;
                        av[pl][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(
                                                                  As + row * ROWB + (((j & ~15) | ((j & 15) ^ (row & 15))) << 4)));
                    }
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) {
                        const int n = rb + 32 * jj;
                        bv[pl][jj] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(
                                                                   Bs + n * ROWB + (((j & ~15) | ((j & 15) ^ (n & 15))) << 4)));
                    }
                }
#define RIH_PN_TERM(ACC_, PA_, PB_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj) ACC_[i][jj] = \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(av[PA_][i], bv[PB_][jj], ACC_[i][jj], 0, 0, 0);
                RIH_PN_TERM(acc1, 1, 0)
                RIH_PN_TERM(acc, 0, 0)
                RIH_PN_TERM(acc1, 0, 1)
#undef RIH_PN_TERM
            }
        __syncthreads();                                // stage st is consumed by every wave; stage st ^ 1 is complete
        // ---- epilogue through this wave's 32 x 32 floats of the consumed stage
        float* stg = reinterpret_cast<float*>(Abuf + st * A_ST) + wave * (32 * 32);
        const int mbase = mt * BM + wm * (32 * TM), nbase = n0 + wn * (32 * TN);
        float4 ssh[STATS ? TN : 1], ssum[STATS ? TN : 1], ssq[STATS ? TN : 1];
        float scnt[STATS ? TN : 1];
        if (STATS) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                ssh[j] = make_float4(0, 0, 0, 0); ssum[j] = make_float4(0, 0, 0, 0); ssq[j] = make_float4(0, 0, 0, 0); scnt[j] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 rv[RES ? 4 : 1];
                if (RES) {                              // the block's residual, requested before the staging round trip
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rv[q] = *reinterpret_cast<const float4*>(p.r + (long long)(mbase + i * 32 + (lane >> 3) + 8 * q) * p.ldr + nbase +
                                                                 j * 32 + (lane & 7) * 4);
                }
                if (i + j > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = fmaf(acc1[i][j][r], 0x1p-11f, acc[i][j][r]) * inv_a * inv_b;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
                    float4 v = *reinterpret_cast<const float4*>(stg + row * 32 + c4);
                    if (RES) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<float4*>(p.c + (long long)(mbase + i * 32 + row) * p.ldc + nbase + j * 32 + c4) = v;
                    if (STATS) {
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
            const long long rblk = mbase / (32 * TM);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float n = scnt[j];
                const float in = n > 0.f ? 1.f / n : 0.f;
                float4 mean = make_float4(ssh[j].x + ssum[j].x * in, ssh[j].y + ssum[j].y * in, ssh[j].z + ssum[j].z * in,
                                          ssh[j].w + ssum[j].w * in);
                float4 m2 = make_float4(ssq[j].x - ssum[j].x * ssum[j].x * in, ssq[j].y - ssum[j].y * ssum[j].y * in,
                                        ssq[j].z - ssum[j].z * ssum[j].z * in, ssq[j].w - ssum[j].w * ssum[j].w * in);
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    const float nbr = __shfl_xor(n, o, 64);
                    const float nt = n + nbr;
                    const float wb = nt > 0.f ? nbr / nt : 0.f;
                    const float cf = n * wb;
#define RIH_PN_MERGE(c_)                                                              \
    {                                                                                 \
        const float mb = __shfl_xor(mean.c_, o, 64), qb = __shfl_xor(m2.c_, o, 64);   \
        const float dl = mb - mean.c_;                                                \
        mean.c_ += dl * wb;                                                           \
        m2.c_ += qb + dl * dl * cf;                                                   \
    }
                    RIH_PN_MERGE(x) RIH_PN_MERGE(y) RIH_PN_MERGE(z) RIH_PN_MERGE(w)
#undef RIH_PN_MERGE
                    n = nt;
                }
                if ((lane >> 3) == 0) {
                    const int nn = nbase + j * 32 + (lane & 7) * 4;
                    *reinterpret_cast<float4*>(p.stats + (rblk * 2 + 0) * p.N + nn) = mean;
                    *reinterpret_cast<float4*>(p.stats + (rblk * 2 + 1) * p.N + nn) = m2;
                }
            }
        }
        __syncthreads();                                // the staging reads are done: the next iteration overwrites this stage
    }
}

int panel_bn(const rih_panel_desc* d) {
    const int cap = 16384 / d->K;                       // 256 (K = 64) / 128 (K = 128)
    if (d->N % 256 == 0 && cap >= 256) return 256;
    if (d->N % 128 == 0 && cap >= 128) return 128;
    return 64;
}
bool panel_ok(const rih_panel_desc* d) {
    if (!d || !d->a || !d->w_h2 || !d->c || !d->amax_a || !d->amax_w) return false;
    if ((d->K != 64 && d->K != 128) || d->N < 64 || d->N % 64 != 0 || d->M < 128 || d->M % 128 != 0) return false;
    if (d->K == 128 && d->N % 128 != 0) return false;   // (64-row tiles leave no 8-wave arrangement for a 64-column block)
    if (d->lda < d->K || d->lda % 4 != 0 || d->ldc < d->N || d->ldc % 4 != 0) return false;
    if (d->r != nullptr && (d->ldr < d->N || d->ldr % 4 != 0)) return false;
    if ((((uintptr_t)d->a | (uintptr_t)d->w_h2 | (uintptr_t)d->c | (uintptr_t)d->r | (uintptr_t)d->stats) % 16) != 0) return false;
    if ((long long)d->M * d->lda * 4 >= (1ll << 31)) return false;     // 31-bit byte offsets into A
    return true;        // (whether the launch has enough (tile, column block) items to fill the chip is the caller's planning)
}

template <int KT, int BN>
void panel_launch(const PanelArgs& a, unsigned grid, bool stats, bool res, hipStream_t s) {
    if (stats && !res) hipLaunchKernelGGL((panel_kernel<KT, BN, true, false>), dim3(grid), dim3(NT), 0, s, a);
    else if (!stats && res) hipLaunchKernelGGL((panel_kernel<KT, BN, false, true>), dim3(grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((panel_kernel<KT, BN, false, false>), dim3(grid), dim3(NT), 0, s, a);
}

// ------------------------------------------------------------------------------------------------ long-K plain-row GEMM ("rows")
// rih_rows: C[M][N] = act(A[M][K] W[N][K]^T (+ R)) for the 1x1 convolutions with a LONG reduction (K >= 256: Bottleneck.conv1 /
// conv3 of layer2-4 forward, their data gradients, the downsample branches' data gradients) -- round 6.  On the tiled kernels of
// rih_gemm.hip these launches ran at 105-200 TF/s (profiles/r05/gemm_dump_c5.json: 16x16 1024 -> 256 51 us = 169 TF/s, 256 -> 1024
// 55-67 us = 128-156 TF/s, 64x64 256 -> 64 82 us = 105 TF/s) where the halo-resident 3x3 kernel reaches 285-350: 256-thread
// workgroups with 128 x 64 tiles move 22 FLOP per operand byte out of L2, convert BOTH operands in every workgroup and have one
// k-tile in flight.  Here the recipe of conv3x3_halo_kernel is applied to plain rows: 512 threads, a BM x BN tile of 256 x 128
// (128 x 128, 256 x 64, 128 x 64 for problems that would not fill the chip otherwise), the weights as pre-split H2 planes staged
// global -> LDS by LDS-DMA (no conversion, no VGPR round trip for B), the A rows global -> registers -> converted ONCE per tile ->
// LDS, THREE A stages and two B stages so that the loads of k-tile t + 2 are issued at the top of k-tile t and stored at its
// end: every request has a whole k-tile of MFMAs to land, one barrier per k-tile, and nothing is in flight across a barrier
// except what that barrier publishes.  Arithmetic: engine 2, product for product (see conv3x3_halo_kernel); k-order identical to
// the tiled kernel's (k ascending in 32-deep tiles), so the two agree to the last bit wherever their accumulation order inside a
// tile agrees, and to fp32 round-off in general.
// LDS images: a row (of A or of W) is 8 units of 16 bytes, unit j = (k / 8) * 2 + plane at position j ^ ((row >> 1) & 7) -- the
// weight image of conv3x3_halo_kernel for both operands (conflict-free ds_read_b128 for 32 consecutive rows).
// Epilogue: optional residual (requested before the staging round trip), ReLU, BatchNorm statistics per 32 TM rows of a wave
// (rih_gemm_desc.stats format).  Preconditions (rih_rows_ok): K % 32 == 0, K >= 64, N % 64 == 0, M % 128 == 0, 16-byte aligned
// operands, pitches % 4 == 0, A < 2 GiB.
#ifndef RIH_ROWS_COUNTED_WAITS
#define RIH_ROWS_COUNTED_WAITS 0
#endif
struct RowsArgs {
    const float* a;
    const unsigned char* w;     // H2 planes [N][K / 8][2][8 halves]
    float* c;
    const float* r;
    float* stats;
    const float* amax_a;
    const float* amax_w;
    int M, N, K, lda, ldc, ldr, relu;
    int nblk, mtiles;
    int H, W, Ho, Wo;           // STEM: the 4-channel input image and the output map
};

// STEM (rih_stem): the A operand is the im2col of a 7 x 7 / stride 2 / padding 3 convolution over a FOUR-channel NHWC image
// (encoder.resnet.conv1 on the 3 -> 4 padded input, models/encoder.py:107-116): k = (tap, channel), a 32-deep k-tile = eight taps, and
// one tap of one output pixel is ONE float4 -- the loader's quad (row, cq) reads the pixel under tap 8 kt + cq of output pixel `row`
// (zero outside the image, zero for the taps 49..55 that pad K = 196 to 224); everything behind the loader is the rows kernel.
template <int BM, int BN, bool STATS, bool RES, bool STEM = false>
__global__ __launch_bounds__(NT, 2) void rows_kernel(const RowsArgs p) {
    // (two wavefronts per SIMD = 256 registers for every tile: a SPILL of a register that an inline-assembly load is still writing
    // would store garbage -- at the 128-register budget of two 128-row workgroups per CU two variants spilled 2-5 registers)
    constexpr int WGN = 2, WGM = 4;                     // waves 4 (M) x 2 (N)
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;               // 2 x 2, 1 x 2, 2 x 1, 1 x 1
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    constexpr int A_ST = BM * 128, B_ST = BN * 128;     // bytes per stage: 32-deep k-tile, two fp16 planes
    constexpr int NPA = (BM * 8) / NT;                  // float4 quads of an A k-tile per thread: 4 / 2
    constexpr int NPB = (BN * 8) / NT;                  // LDS-DMA units of a W k-tile per thread: 2 / 1
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "loader geometry: every thread issues the same number of requests");
    constexpr int SMEM = 2 * A_ST + 3 * B_ST;          // two A stages, three W stages
    static_assert(SMEM >= 8 * 32 * SLD * 4, "epilogue staging fits");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    unsigned char* const Abuf = smem;
    unsigned char* const Bbuf = smem + 2 * A_ST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    // workgroup -> (row tile, column block); column blocks fastest: the workgroups that share A rows are neighbours on one XCD
    const int id = xcd_remap_c3((int)blockIdx.x, (int)gridDim.x);
    const int nb = id % p.nblk, mt = id / p.nblk;
    const int m0 = mt * BM, n0 = nb * BN;

    const float sa = c3_scale(p.amax_a), sb = c3_scale(p.amax_w);

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, (short)0, (int)0x7fffffff, 0x00020000);
    // A: quad q = pass * NT + tid -> row q / 8, channel quad q % 8 of the 32-deep k-tile
    unsigned a_goff[NPA];
    int a_lds[NPA];
    int s_ih0[STEM ? NPA : 1], s_iw0[STEM ? NPA : 1];   // STEM: input coordinates under tap (0, 0) of the row's output pixel
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int q = i * NT + tid;
        const int row = q >> 3, cq = q & 7;
        if (STEM) {
            const int m = m0 + row, hw = p.Ho * p.Wo;
            const int img = m / hw, pix = m - img * hw, oh = pix / p.Wo, ow = pix - oh * p.Wo;
            s_ih0[i] = 2 * oh - 3;
            s_iw0[i] = 2 * ow - 3;
            // (wraps for negative coordinates; only used where the coordinates are inside the image)
            a_goff[i] = (m < p.M) ? (unsigned)((((long long)img * p.H + s_ih0[i]) * p.W + s_iw0[i]) * 16) : OOB;
        } else {
            a_goff[i] = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + 4 * cq) * 4) : OOB;   // (M * lda * 4 < 2^31)
        }
        a_lds[i] = row * 128 + ((((cq >> 1) * 2) ^ ((row >> 1) & 7)) << 4) + (cq & 1) * 8;
    }
    // W: unit U = pass * NT + tid -> row n = U / 8, LDS position U % 8 holds source unit j = pos ^ ((n >> 1) & 7)
    const unsigned char* b_src[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int U = i * NT + tid;
        const int n = U >> 3, j = (U & 7) ^ ((n >> 1) & 7);
        b_src[i] = p.w + ((long long)(n0 + n) * (p.K >> 3)) * 32 + j * 16;          // + (k / 8) * 32 per k-tile
    }
    // operand fetch offsets [k-step][plane][block]: rows wm * 32 TM + 32 i + l31 of A, wn * 32 TN + 32 jj + l31 of W
    int a_rd[2][2][TM], b_rd[2][2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * (32 * TM) + 32 * i + l31;
                a_rd[s][pl][i] = r * 128 + (((((2 * s + lhi) * 2) + pl) ^ ((r >> 1) & 7)) << 4);
            }
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) {
                const int n = wn * (32 * TN) + 32 * jj + l31;
                b_rd[s][pl][jj] = n * 128 + (((((2 * s + lhi) * 2) + pl) ^ ((n >> 1) & 7)) << 4);
            }
        }

    float4 areg[2][NPA];                                // two k-tiles of A rows in flight / landed (indexed by literal parity below)
    // Every request of the main loop -- the A rows (buffer loads) and the weights (LDS-DMA) -- is issued through inline assembly,
    // OUTSIDE hipcc's wait-count bookkeeping, and waited for with counted waits: the compiler's own placement drains vmcnt to 0 in
    // front of every barrier behind an LDS-DMA and in front of every conversion behind a conditional request, which leaves each
    // request ONE k-tile of MFMAs to land -- and a round trip to L2 / HBM on this part (0.7-2 us under load) is longer than a
    // k-tile (0.5-1 us).  Requests past the last k-tile are issued all the same (out-of-range offset: zeros, no memory access; the
    // weights of the last k-tile again into a free stage), so that the number of requests per iteration is a compile-time constant.
    const c3_i32x4 rsA = {(int)(unsigned)(uintptr_t)p.a, (int)((unsigned long long)(uintptr_t)p.a >> 32), (int)0x7fffffff, (int)0x00020000};
#define RIH_RW_LOAD_A(SET_, KT_)                        /* global -> registers: columns [32 KT_, 32 KT_ + 32) of the tile's rows */ \
    {                                                                                                                        \
        const bool in_ = (KT_) < nk;                                                                                          \
        if (STEM) {                                     /* this lane's tap of the k-tile: (kh, kw) = divmod(8 KT_ + cq, 7) */   \
            const int tap_ = 8 * (KT_) + (tid & 7), kh_ = (tap_ * 37) >> 8, kw_ = tap_ - 7 * kh_;                             \
            const unsigned add_ = (unsigned)((kh_ * p.W + kw_) * 16);                                                         \
            unsigned off_[NPA];                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                                \
                const bool ok_ = in_ && tap_ < 49 && a_goff[i] != OOB && (unsigned)(s_ih0[i] + kh_) < (unsigned)p.H &&        \
                                 (unsigned)(s_iw0[i] + kw_) < (unsigned)p.W;                                                  \
                off_[i] = ok_ ? a_goff[i] + add_ : OOB;                                                                       \
            }                                                                                                                \
            _Pragma("unroll") for (int i = 0; i < NPA; ++i) areg[SET_][i] = c3_bload4_raw(rsA, rA, off_[i]);                  \
        } else {                                                                                                             \
            const unsigned add_ = (unsigned)(KT_) * 128u;                                                                     \
            unsigned off_[NPA];                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NPA; ++i) off_[i] = (a_goff[i] == OOB || !in_) ? OOB : a_goff[i] + add_;    \
            _Pragma("unroll") for (int i = 0; i < NPA; ++i) areg[SET_][i] = c3_bload4_raw(rsA, rA, off_[i]);                  \
        }                                                                                                                    \
    }
#define RIH_RW_STORE_A(SET_, DST_)                      /* registers -> two fp16 planes in LDS */                            \
    {                                                                                                                        \
        unsigned char* const dst_ = (DST_);                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NPA; ++i) {                                                                    \
            unsigned h0, l0, h1, l1;                                                                                         \
            c3_split2h_pinned(areg[SET_][i].x, areg[SET_][i].y, sa, h0, l0);                                                 \
            c3_split2h_pinned(areg[SET_][i].z, areg[SET_][i].w, sa, h1, l1);                                                 \
            *reinterpret_cast<uint2*>(dst_ + a_lds[i]) = make_uint2(h0, h1);                                                 \
            *reinterpret_cast<uint2*>(dst_ + (a_lds[i] ^ 16)) = make_uint2(l0, l1);                                          \
        }                                                                                                                    \
    }
    const int nk = p.K >> 5;
    auto issue_B = [&](int kt, unsigned char* dst) {    // LDS-DMA: weight rows [n0, n0 + BN) x k-tile kt (clamped to the last one)
        const long long ko = (long long)(kt < nk ? kt : nk - 1) * 128;
#pragma unroll
        for (int i = 0; i < NPB; ++i) c3_glds16_raw(b_src[i] + ko, dst + (i * NT + tid) * 16);
    };

    floatx16 acc[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    auto multiply = [&](const unsigned char* As, const unsigned char* Bs) {         // one 32-deep k-tile out of LDS
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 av[2][TM], bv[2][TN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    av[pl][i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(As + a_rd[s][pl][i]));
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
                    bv[pl][jj] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Bs + b_rd[s][pl][jj]));
            }
#define RIH_RW_TERM(ACC_, PA_, PB_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int jj = 0; jj < TN; ++jj) ACC_[i][jj] = \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(av[PA_][i], bv[PB_][jj], ACC_[i][jj], 0, 0, 0);
            RIH_RW_TERM(acc1, 1, 0)
            RIH_RW_TERM(acc, 0, 0)
            RIH_RW_TERM(acc1, 0, 1)
#undef RIH_RW_TERM
        }
    };

    // Schedule.  Iteration j multiplies k-tile j (A in LDS stage j & 1, W in stage j % 3).  At its top it requests the weights AND the
    // A rows of k-tile j + 2 (weights: LDS-DMA into the stage read in iteration j - 1; rows: into the register set converted at the
    // end of iteration j - 1); after the MFMAs it waits, converts the rows of k-tile j + 1 into the other A stage and meets the
    // barrier that publishes both.
    // THE WAIT IS vmcnt(0).  The design was a counted wait -- vmcnt(NPB + NPA): "only this iteration's requests may be outstanding",
    // which leaves every request two k-tiles of MFMAs to land -- and it is WRONG on this part: with LDS-DMA requests and buffer
    // loads in one queue a counted wait does not guarantee that the OLDER buffer loads have returned.  Measured (round 6,
    // tools/r6_stress_rows.py, profiles/r06/rows/c14_*): 6 of 800 launches wrote rows computed from registers whose load had not
    // landed (always whole load instructions: 32 rows at a stride of 8), only on the HBM-bound 262144 x 256 -> 64 shapes; with
    // vmcnt(0) 0 of 800, and the step's gradients are bit-reproducible again.  The counted form (RIH_ROWS_COUNTED_WAITS=1) was 2.4 %
    // faster on the rows launches, 0.4 % on the step.  hipcc never emits a counted wait across an LDS-DMA request either.
    unsigned char* Bc = Bbuf;                           // W stage of k-tile j, j + 1, j + 2 (rotating)
    unsigned char* Bn = Bbuf + B_ST;
    unsigned char* Bf = Bbuf + 2 * B_ST;
    issue_B(0, Bc);
    issue_B(1, Bn);
    RIH_RW_LOAD_A(0, 0)
    RIH_RW_LOAD_A(1, 1)
    c3_vm_wait<RIH_ROWS_COUNTED_WAITS ? NPA : 0>();     // the weights of k-tiles 0 and 1 and the rows of k-tile 0 have landed
    RIH_RW_STORE_A(0, Abuf)
    __syncthreads();
    int j = 0;
#define RIH_RW_ITER(PAR_)                               /* PAR_ = j & 1 (a literal: the register sets are not indexable) */     \
    {                                                                                                                        \
        issue_B(j + 2, Bf);                                                                                                  \
        RIH_RW_LOAD_A(PAR_, j + 2)                                                                                           \
        multiply(Abuf + PAR_ * A_ST, Bc);                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        c3_vm_wait<RIH_ROWS_COUNTED_WAITS ? NPB + NPA : 0>();                                                                \
        if (j + 1 < nk) RIH_RW_STORE_A(1 - PAR_, Abuf + (1 - PAR_) * A_ST)                                                   \
        __syncthreads();                                                                                                     \
        unsigned char* const t_ = Bc;                                                                                        \
        Bc = Bn; Bn = Bf; Bf = t_;                                                                                           \
        ++j;                                                                                                                 \
    }
#pragma unroll 1
    while (j + 1 < nk) {
        RIH_RW_ITER(0)
        RIH_RW_ITER(1)
    }
    // The compiler does not know that the registers of an inline-assembly load are written LATER: a request whose result nothing
    // reads is a dead definition to it, and it may hand the destination registers to the next instruction that needs some -- the
    // load then lands on top of a live value.  (Found on the GPU in round 6: with an odd trip count the last iteration, a separate
    // copy of the loop body, issued its surplus out-of-range requests into dead registers that the MFMA operand fetches re-used;
    // results were wrong and changed from run to run, the host harness -- where a load completes at once -- passed.)  So: every
    // request of the loop has a consumer in the next iteration's conversion (possibly never executed, but live for the register
    // allocator), the odd trip count's last k-tile issues NO request, and everything still in flight is drained behind a scheduling
    // barrier before the first instruction of the epilogue.
    if (j < nk) {                                       // (an odd trip count ends on an even k-tile: A stage 0)
        c3_vm_wait<0>();
        __builtin_amdgcn_sched_barrier(0);
        multiply(Abuf, Bc);
        __syncthreads();
    }
    c3_vm_wait<0>();                                    // (the surplus requests of the last iterations: nothing may land behind this line)
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
#undef RIH_RW_ITER
#undef RIH_RW_LOAD_A
#undef RIH_RW_STORE_A

    // ---------------------------------------------------------------- epilogue (the loop ended with a barrier: LDS is free)
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
    const float inv_a = 1.f / sa, inv_b = 1.f / sb;
    const int mbase = m0 + wm * (32 * TM), nbase = n0 + wn * (32 * TN);
    float4 ssh[STATS ? TN : 1], ssum[STATS ? TN : 1], ssq[STATS ? TN : 1];
    float scnt[STATS ? TN : 1];
    if (STATS) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            ssh[j] = make_float4(0, 0, 0, 0); ssum[j] = make_float4(0, 0, 0, 0); ssq[j] = make_float4(0, 0, 0, 0); scnt[j] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float4 rv[RES ? 4 : 1];
            if (RES) {                                  // the block's residual, requested before the staging round trip
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    rv[q] = *reinterpret_cast<const float4*>(p.r + (long long)(mbase + i * 32 + (lane >> 3) + 8 * q) * p.ldr + nbase +
                                                             j * 32 + (lane & 7) * 4);
            }
            if (i + j > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg[((r & 3) + 8 * (r >> 2) + 4 * lhi) * SLD + l31] = fmaf(acc1[i][j][r], 0x1p-11f, acc[i][j][r]) * inv_a * inv_b;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (lane >> 3) + 8 * q, c4 = (lane & 7) * 4;
                float4 v = *reinterpret_cast<const float4*>(stg + row * SLD + c4);
                if (RES) { v.x += rv[q].x; v.y += rv[q].y; v.z += rv[q].z; v.w += rv[q].w; }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(p.c + (long long)(mbase + i * 32 + row) * p.ldc + nbase + j * 32 + c4) = v;
                if (STATS) {
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
        const long long rblk = mbase / (32 * TM);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float n = scnt[j];
            const float in = n > 0.f ? 1.f / n : 0.f;
            float4 mean = make_float4(ssh[j].x + ssum[j].x * in, ssh[j].y + ssum[j].y * in, ssh[j].z + ssum[j].z * in,
                                      ssh[j].w + ssum[j].w * in);
            float4 m2 = make_float4(ssq[j].x - ssum[j].x * ssum[j].x * in, ssq[j].y - ssum[j].y * ssum[j].y * in,
                                    ssq[j].z - ssum[j].z * ssum[j].z * in, ssq[j].w - ssum[j].w * ssum[j].w * in);
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
                const float nbr = __shfl_xor(n, o, 64);
                const float nt = n + nbr;
                const float wb = nt > 0.f ? nbr / nt : 0.f;
                const float cf = n * wb;
#define RIH_RW_MERGE(c_)                                                              \
    {                                                                                 \
        const float mb = __shfl_xor(mean.c_, o, 64), qb = __shfl_xor(m2.c_, o, 64);   \
        const float dl = mb - mean.c_;                                                \
        mean.c_ += dl * wb;                                                           \
        m2.c_ += qb + dl * dl * cf;                                                   \
    }
                RIH_RW_MERGE(x) RIH_RW_MERGE(y) RIH_RW_MERGE(z) RIH_RW_MERGE(w)
#undef RIH_RW_MERGE
                n = nt;
            }
            if ((lane >> 3) == 0) {
                const int nn = nbase + j * 32 + (lane & 7) * 4;
                *reinterpret_cast<float4*>(p.stats + (rblk * 2 + 0) * p.N + nn) = mean;
                *reinterpret_cast<float4*>(p.stats + (rblk * 2 + 1) * p.N + nn) = m2;
            }
        }
    }
}

// tile of a rows launch: the largest of 256 x 128, 128 x 128 (256 x 64 for N = 64), 128 x 64 that still gives every CU a workgroup
void rows_tile(const rih_panel_desc* d, int& bm, int& bn) {
    const bool n128 = d->N % 128 == 0, m256 = d->M % 256 == 0;
    const auto wgs = [&](int m, int n) { return (long long)(d->M / m) * (d->N / n); };
    if (n128 && m256 && wgs(256, 128) >= 256) { bm = 256; bn = 128; return; }
    if (n128 && wgs(128, 128) >= 256) { bm = 128; bn = 128; return; }
    if (m256 && wgs(256, 64) >= 256) { bm = 256; bn = 64; return; }
    if (wgs(128, 64) >= 256 || !n128) { bm = 128; bn = 64; return; }
    bm = 128; bn = 128;         // a small problem: fewer, larger tiles (the caller's planning decides whether it comes here at all)
}
bool rows_ok(const rih_panel_desc* d) {
    if (!d || !d->a || !d->w_h2 || !d->c || !d->amax_a || !d->amax_w) return false;
    if (d->K < 64 || d->K % 32 != 0 || d->N < 64 || d->N % 64 != 0 || d->M < 128 || d->M % 128 != 0) return false;
    if (d->lda < d->K || d->lda % 4 != 0 || d->ldc < d->N || d->ldc % 4 != 0) return false;
    if (d->r != nullptr && (d->ldr < d->N || d->ldr % 4 != 0)) return false;
    if ((((uintptr_t)d->a | (uintptr_t)d->w_h2 | (uintptr_t)d->c | (uintptr_t)d->r | (uintptr_t)d->stats) % 16) != 0) return false;
    if ((long long)d->M * d->lda * 4 >= (1ll << 31)) return false;     // 31-bit byte offsets into A
    return true;
}

bool stem_ok(const rih_conv3_desc* d) {
    if (!d || !d->x || !d->w_h2 || !d->y || !d->amax_x || !d->amax_w) return false;
    if (d->imgs < 1 || d->H < 8 || d->W < 8 || d->H % 2 != 0 || d->W % 2 != 0 || d->C != 4 || d->ldx != 4) return false;
    if (d->N != 64 || d->ldy < d->N || d->ldy % 4 != 0 || d->Kpad != 224 || d->r != nullptr) return false;
    if ((((uintptr_t)d->x | (uintptr_t)d->w_h2 | (uintptr_t)d->y | (uintptr_t)d->stats) % 16) != 0) return false;
    const long long M = (long long)d->imgs * (d->H / 2) * (d->W / 2);
    if (M % 256 != 0 || M >= (1ll << 31)) return false;
    return (long long)d->imgs * d->H * d->W * 16 < (1ll << 31);         // 31-bit byte offsets into the image batch
}

template <int BM, int BN>
void rows_launch(const RowsArgs& a, unsigned grid, bool stats, bool res, hipStream_t s) {
    if (stats && !res) hipLaunchKernelGGL((rows_kernel<BM, BN, true, false>), dim3(grid), dim3(NT), 0, s, a);
    else if (!stats && res) hipLaunchKernelGGL((rows_kernel<BM, BN, false, true>), dim3(grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((rows_kernel<BM, BN, false, false>), dim3(grid), dim3(NT), 0, s, a);
}

// patch width: 32 when the map allows it (one image row per 32-row block), else 16 (two rows per block); 0: not a shape of this kernel
int c3_tw(const rih_conv3_desc* d) {
    if (d->H % 8 == 0 && d->W % 32 == 0) return 32;
    if (d->H % 16 == 0 && d->W % 16 == 0) return 16;
    return 0;
}
// output-channel block: the widest that divides N; 128 -> 64 when the 128-wide grid would leave CUs without a workgroup
int c3_bn(const rih_conv3_desc* d, int tw) {
    const long long patches = (long long)d->imgs * (d->H / (256 / tw)) * (d->W / tw);
    if (d->N % 128 == 0 && patches * (d->N / 128) >= 256) return 128;
    if (d->N % 64 == 0) return 64;
    return 32;
}
bool c3_ok(const rih_conv3_desc* d) {
    if (!d || !d->x || !d->w_h2 || !d->y || !d->amax_x || !d->amax_w) return false;
    if (d->imgs < 1 || d->H < 8 || d->W < 16 || c3_tw(d) == 0) return false;
    if (d->C < 32 || d->C % 32 != 0 || d->N < 32 || d->N % 32 != 0) return false;
    if (d->ldx < d->C || d->ldx % 4 != 0 || d->ldy < d->N || d->ldy % 4 != 0 || d->Kpad != 9 * d->C) return false;
    if ((((uintptr_t)d->x | (uintptr_t)d->w_h2 | (uintptr_t)d->y | (uintptr_t)d->stats) % 16) != 0) return false;
    if ((long long)d->H * d->W * d->ldx * 4 >= (1ll << 31)) return false;
    if (d->r != nullptr && (d->stats != nullptr || d->ldr < d->N || d->ldr % 4 != 0 || ((uintptr_t)d->r % 16) != 0)) return false;
    const long long wg = (long long)d->imgs * (d->H / 8) * (d->W / 16) * (d->N / 32);        // (an upper bound of the grid)
    return wg < (1ll << 31);
}

template <int TW, int BN>
void c3_launch(const C3Args& a, unsigned grid, bool stats, hipStream_t s) {
    if (stats) hipLaunchKernelGGL((conv3x3_halo_kernel<TW, BN, true>), dim3(grid), dim3(NT), 0, s, a);
    else if (a.r != nullptr) hipLaunchKernelGGL((conv3x3_halo_kernel<TW, BN, false, true>), dim3(grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<TW, BN, false>), dim3(grid), dim3(NT), 0, s, a);
}

}  // namespace

extern "C" int rih_conv3x3_ok(const rih_conv3_desc* d) { return c3_ok(d) ? 1 : 0; }

/* rows of the output per BatchNorm statistics block (rih_bn_stats_from_blocks' rows_per_block) for this descriptor: 64, or 32
 * with 32-channel blocks; the blocks are pieces of the 256-pixel patches, not of consecutive rows -- the merge only needs
 * their row counts, and every block is full.  0: not a shape of this kernel. */
extern "C" int rih_conv3x3_stats_rows(const rih_conv3_desc* d) {
    if (!c3_ok(d)) return 0;
    return c3_bn(d, c3_tw(d)) >= 64 ? 64 : 32;
}

extern "C" int rih_conv3x3(const rih_conv3_desc* d, void* stream) {
    if (!c3_ok(d)) return RIH_EINVAL;
    C3Args a;
    a.x = d->x; a.w = (const unsigned char*)d->w_h2; a.y = d->y; a.stats = d->stats; a.amax_x = d->amax_x; a.amax_w = d->amax_w;
    a.imgs = d->imgs; a.H = d->H; a.W = d->W; a.C = d->C; a.N = d->N; a.ldx = d->ldx; a.ldy = d->ldy; a.Kp = d->Kpad;
    a.relu = d->relu ? 1 : 0;
    a.r = d->r; a.ldr = d->ldr;
    const int tw = c3_tw(d), bn = c3_bn(d, tw);
    a.tiles_x = d->W / tw; a.tiles_y = d->H / (256 / tw);
    a.nblk = d->N / bn;
    const unsigned grid = (unsigned)((long long)d->imgs * a.tiles_x * a.tiles_y * a.nblk);
    hipStream_t s = (hipStream_t)stream;
    const bool st = d->stats != nullptr;
    if (tw == 32) {
        if (bn == 128) c3_launch<32, 128>(a, grid, st, s);
        else if (bn == 64) c3_launch<32, 64>(a, grid, st, s);
        else c3_launch<32, 32>(a, grid, st, s);
    } else {
        if (bn == 128) c3_launch<16, 128>(a, grid, st, s);
        else if (bn == 64) c3_launch<16, 64>(a, grid, st, s);
        else c3_launch<16, 32>(a, grid, st, s);
    }
    return (int)hipGetLastError();
}


extern "C" int rih_panel_ok(const rih_panel_desc* d) { return panel_ok(d) && !(d->stats && d->r) ? 1 : 0; }

/* rows per BatchNorm statistics block of rih_panel for this descriptor (32 or 64), 0: not a shape of the kernel */
extern "C" int rih_panel_stats_rows(const rih_panel_desc* d) {
    if (!panel_ok(d)) return 0;
    const int bn = panel_bn(d), bm = 8192 / d->K;
    const int wgm = bn >= 128 ? 2 : 4;
    return bm / wgm;
}

extern "C" int rih_panel(const rih_panel_desc* d, void* stream) {
    if (!panel_ok(d) || (d->stats && d->r)) return RIH_EINVAL;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return RIH_EINVAL;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    PanelArgs a;
    a.a = d->a; a.w = (const unsigned char*)d->w_h2; a.c = d->c; a.r = d->r; a.stats = d->stats;
    a.amax_a = d->amax_a; a.amax_w = d->amax_w;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.relu = d->relu ? 1 : 0;
    const int bn = panel_bn(d), bm = 8192 / d->K;
    a.nblk = d->N / bn;
    a.mtiles = d->M / bm;
    // one persistent workgroup per CU (128 KB of LDS), a multiple of the column-block count so that a workgroup keeps ONE block
    long long g = (long long)cus / a.nblk * a.nblk;
    if (g < a.nblk) g = a.nblk;
    const long long items = (long long)a.mtiles * a.nblk;
    if (g > items) g = items;
    const unsigned grid = (unsigned)g;
    hipStream_t s = (hipStream_t)stream;
    const bool st = d->stats != nullptr, rs = d->r != nullptr;
    if (d->K == 64) {
        if (bn == 256) panel_launch<2, 256>(a, grid, st, rs, s);
        else if (bn == 128) panel_launch<2, 128>(a, grid, st, rs, s);
        else panel_launch<2, 64>(a, grid, st, rs, s);
    } else {
        panel_launch<4, 128>(a, grid, st, rs, s);
    }
    return (int)hipGetLastError();
}

extern "C" int rih_rows_ok(const rih_panel_desc* d) { return rows_ok(d) && !(d->stats && d->r) ? 1 : 0; }

/* rows per BatchNorm statistics block of rih_rows for this descriptor (64 with 256-row tiles, 32 with 128-row tiles), 0: not a
 * shape of the kernel */
extern "C" int rih_rows_stats_rows(const rih_panel_desc* d) {
    if (!rows_ok(d)) return 0;
    int bm, bn;
    rows_tile(d, bm, bn);
    return bm / 4;
}

extern "C" int rih_rows(const rih_panel_desc* d, void* stream) {
    if (!rows_ok(d) || (d->stats && d->r)) return RIH_EINVAL;
    RowsArgs a;
    a.a = d->a; a.w = (const unsigned char*)d->w_h2; a.c = d->c; a.r = d->r; a.stats = d->stats;
    a.amax_a = d->amax_a; a.amax_w = d->amax_w;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldc = d->ldc; a.ldr = d->ldr; a.relu = d->relu ? 1 : 0;
    a.H = a.W = a.Ho = a.Wo = 0;
    int bm, bn;
    rows_tile(d, bm, bn);
    a.nblk = d->N / bn;
    a.mtiles = d->M / bm;
    const unsigned grid = (unsigned)((long long)a.mtiles * a.nblk);
    hipStream_t s = (hipStream_t)stream;
    const bool st = d->stats != nullptr, rs = d->r != nullptr;
    if (bm == 256 && bn == 128) rows_launch<256, 128>(a, grid, st, rs, s);
    else if (bm == 128 && bn == 128) rows_launch<128, 128>(a, grid, st, rs, s);
    else if (bm == 256) rows_launch<256, 64>(a, grid, st, rs, s);
    else rows_launch<128, 64>(a, grid, st, rs, s);
    return (int)hipGetLastError();
}

extern "C" int rih_stem_ok(const rih_conv3_desc* d) { return stem_ok(d) ? 1 : 0; }

extern "C" int rih_stem(const rih_conv3_desc* d, void* stream) {
    if (!stem_ok(d)) return RIH_EINVAL;
    RowsArgs a;
    a.a = d->x; a.w = (const unsigned char*)d->w_h2; a.c = d->y; a.r = nullptr; a.stats = d->stats;
    a.amax_a = d->amax_x; a.amax_w = d->amax_w;
    a.H = d->H; a.W = d->W; a.Ho = d->H / 2; a.Wo = d->W / 2;
    a.M = d->imgs * a.Ho * a.Wo; a.N = d->N; a.K = d->Kpad; a.lda = 4; a.ldc = d->ldy; a.ldr = 0; a.relu = d->relu ? 1 : 0;
    a.nblk = 1;
    a.mtiles = a.M / 256;
    hipStream_t s = (hipStream_t)stream;
    if (d->stats != nullptr) hipLaunchKernelGGL((rows_kernel<256, 64, true, false, true>), dim3((unsigned)a.mtiles), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL((rows_kernel<256, 64, false, false, true>), dim3((unsigned)a.mtiles), dim3(NT), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int rih_h2_multi(const rih_h2_desc* descs, int n, void* stream) {
    if (n < 0 || (n > 0 && !descs)) return RIH_EINVAL;
    for (int i0 = 0; i0 < n; i0 += H2_PACK) {
        H2Pack pk;
        pk.n = (n - i0 < H2_PACK) ? n - i0 : H2_PACK;
        int total = 0;
        for (int i = 0; i < pk.n; ++i) {
            const int rc = h2_args(pk.d[i], descs[i0 + i]);
            if (rc != RIH_OK) return rc;
            const long long units = (long long)pk.d[i].N * (pk.d[i].Kp / 8);
            long long nb = (units + 255) / 256;
            if (nb > 512) nb = 512;
            pk.first[i] = total;
            total += (int)nb;
        }
        pk.first[pk.n] = total;
        hipLaunchKernelGGL(h2_weight_multi_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, pk);
    }
    return (int)hipGetLastError();
}
