// TEST INFRASTRUCTURE: host emulation of the gfx950 builtins used by renderih_amd/csrc/rih_gemm.hip (see hip_runtime.h).
//   * buffer resources + raw 128-bit buffer loads with the hardware range check (a dword at or beyond num_records reads 0)
//   * v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 as wavefront collectives with the register layouts of the ISA:
//       A (32 x K): lane l holds row l%32, k = kpl*(l/32) .. +kpl-1 (kpl = 1 resp. 8 values per lane)
//       B (K x 32): lane l holds column l%32, the same k range
//       C/D (32 x 32 fp32, 16 per lane): register r of lane l = row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
#pragma once
#include <cstdio>

#define RIH_CONST_AS        /* the constant address space of the grouped GEMM's table reads: plain memory here */

struct hipcpu_rsrc { const char* base; unsigned bytes; };
#define __amdgpu_buffer_rsrc_t hipcpu_rsrc
typedef unsigned hipcpu_u32x4 __attribute__((ext_vector_type(4)));
typedef float hipcpu_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 hipcpu_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hipcpu_f16x8 __attribute__((ext_vector_type(8)));

static inline hipcpu_rsrc hipcpu_make_rsrc(const void* p, unsigned num) { return hipcpu_rsrc{(const char*)p, num}; }
static inline hipcpu_u32x4 hipcpu_buffer_load_b128(hipcpu_rsrc r, unsigned off) {
    hipcpu_u32x4 v = {0u, 0u, 0u, 0u};
    for (int i = 0; i < 4; ++i) {
        const unsigned long long o = (unsigned long long)off + 4ull * i;
        if (o + 4ull <= r.bytes) { unsigned w; std::memcpy(&w, r.base + o, 4); v[i] = w; }
    }
    return v;
}
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) hipcpu_make_rsrc((const void*)(p), (unsigned)(num))
#define __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, aux) hipcpu_buffer_load_b128((r), (unsigned)(off))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

static inline hipcpu_f32x16 hipcpu_mfma_32x32x2_f32(float a, float b, hipcpu_f32x16 c) {
    float* mine = (float*)hipcpu::xchg_slot(hipcpu::S().cur->flat);
    mine[0] = a; mine[1] = b;
    hipcpu::wave_barrier();
    const int l = hipcpu::lane(), base = hipcpu::wave_base(), n = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k)
            acc += ((const float*)hipcpu::xchg_slot(base + m + 32 * k))[0] * ((const float*)hipcpu::xchg_slot(base + n + 32 * k))[1];
        c[r] = acc;
    }
    hipcpu::wave_barrier();
    return c;
}
static inline hipcpu_f32x16 hipcpu_mfma_32x32x16_bf16(hipcpu_bf16x8 a, hipcpu_bf16x8 b, hipcpu_f32x16 c) {
    unsigned char* mine = hipcpu::xchg_slot(hipcpu::S().cur->flat);
    std::memcpy(mine, &a, 16);
    std::memcpy(mine + 16, &b, 16);
    hipcpu::wave_barrier();
    const int l = hipcpu::lane(), base = hipcpu::wave_base(), n = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            hipcpu_bf16x8 av, bv;
            std::memcpy(&av, hipcpu::xchg_slot(base + m + 32 * (k >> 3)), 16);
            std::memcpy(&bv, hipcpu::xchg_slot(base + n + 32 * (k >> 3)) + 16, 16);
            acc += (float)av[k & 7] * (float)bv[k & 7];
        }
        c[r] = acc;
    }
    hipcpu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipcpu_mfma_32x32x2_f32((a), (b), (c))
// scheduling barrier of a wavefront (no instruction on the hardware, where the lanes run in lock-step): here the point at which
// every fiber of the wavefront has caught up -- LDS written by one lane and read by another needs it
#define __builtin_amdgcn_wave_barrier() hipcpu::wave_barrier()
// v_mfma_f32_16x16x4_f32: A (16 x 4): lane l holds row l%16, k = l/16; B (4 x 16): lane l holds column l%16, k = l/16;
// C/D (16 x 16 fp32, 4 per lane): register r of lane l = row 4*(l>>4) + r, column l&15.  A k-ordered fmaf chain like the hardware.
typedef float hipcpu_f32x4 __attribute__((ext_vector_type(4)));
static inline hipcpu_f32x4 hipcpu_mfma_16x16x4_f32(float a, float b, hipcpu_f32x4 c) {
    float* mine = (float*)hipcpu::xchg_slot(hipcpu::S().cur->flat);
    mine[0] = a; mine[1] = b;
    hipcpu::wave_barrier();
    const int l = hipcpu::lane(), base = hipcpu::wave_base(), n = l & 15, q = l >> 4;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * q + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = __builtin_fmaf(((const float*)hipcpu::xchg_slot(base + m + 16 * k))[0],
                                 ((const float*)hipcpu::xchg_slot(base + n + 16 * k))[1], acc);
        c[r] = acc;
    }
    hipcpu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipcpu_mfma_16x16x4_f32((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipcpu_mfma_32x32x16_bf16((a), (b), (c))
static inline hipcpu_f32x16 hipcpu_mfma_32x32x16_f16(hipcpu_f16x8 a, hipcpu_f16x8 b, hipcpu_f32x16 c) {
    unsigned char* mine = hipcpu::xchg_slot(hipcpu::S().cur->flat);
    std::memcpy(mine, &a, 16);
    std::memcpy(mine + 16, &b, 16);
    hipcpu::wave_barrier();
    const int l = hipcpu::lane(), base = hipcpu::wave_base(), n = l & 31, hi = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            hipcpu_f16x8 av, bv;
            std::memcpy(&av, hipcpu::xchg_slot(base + m + 32 * (k >> 3)), 16);
            std::memcpy(&bv, hipcpu::xchg_slot(base + n + 32 * (k >> 3)) + 16, 16);
            acc += (float)av[k & 7] * (float)bv[k & 7];
        }
        c[r] = acc;
    }
    hipcpu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipcpu_mfma_32x32x16_f16((a), (b), (c))

// LDS-DMA (global_load_lds_dwordx4): the hardware writes lane l's 16 bytes to M0 + 16*l with M0 taken from the FIRST lane's
// LDS pointer (the compiler emits v_readfirstlane) -- the per-lane pointers are NOT a scatter.  The emulation therefore
// exchanges the first lane's pointer and aborts if a lane's own pointer is not first + 16*lane.  The global bytes are read
// at issue, the LDS write is deferred until the issuing thread's next __syncthreads() (see Fiber::pending).
static inline void hipcpu_global_load_lds16(const void* g, void* lds, int size, int off) {
    if (size != 16 || off != 0) { std::fprintf(stderr, "hipcpu: global_load_lds emulated for 16-byte pieces only\n"); std::abort(); }
    void** mine = (void**)hipcpu::xchg_slot(hipcpu::S().cur->flat);
    mine[0] = lds;
    hipcpu::wave_barrier();
    void* first = ((void**)hipcpu::xchg_slot(hipcpu::wave_base()))[0];
    if ((unsigned char*)lds != (unsigned char*)first + 16 * hipcpu::lane()) {
        std::fprintf(stderr, "hipcpu: global_load_lds destination is not lane-linear (lane %d)\n", hipcpu::lane());
        std::abort();
    }
    hipcpu::Fiber::Pending q;
    q.dst = lds;
    std::memcpy(q.data, g, 16);
    hipcpu::S().cur->pending.push_back(q);
    hipcpu::wave_barrier();
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    hipcpu_global_load_lds16((const void*)(g), (void*)(l), (size), (off))
