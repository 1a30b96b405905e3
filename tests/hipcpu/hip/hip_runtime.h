// TEST INFRASTRUCTURE (not product): a minimal "HIP on the CPU" so that the kernel sources under renderih_amd/csrc/
// can be compiled for the host (clang++ -x c++ -I tests/hipcpu) and EXECUTED in the CPU test-suite -- the arithmetic,
// indexing, LDS staging, barriers and wavefront shuffles of the real kernels, on small problems, without a GPU.
//
// Execution model: blocks run one after the other; the threads of a block are cooperative fibers (ucontext) on one OS
// thread.  __syncthreads() and the 64-lane wavefront collectives (__shfl_xor, readfirstlane, MFMA) are counting
// barriers at which a fiber yields to the scheduler until every live fiber of the block (resp. wavefront) has arrived.
// `__shared__` becomes a function-local static (one block is resident at a time).  gfx950 builtins used by the GEMM
// kernels (buffer resources / raw buffer loads with range check, the two MFMA shapes) are emulated in hipcpu_gfx950.h.
// Nothing here models performance.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
// HIP's device-side integer min / max
template <typename T> static inline T min(T a, T b) { return b < a ? b : a; }
template <typename T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

namespace hipcpu {

constexpr int WAVE = 64;
constexpr size_t STACK = 512 * 1024;

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    dim3 tid;
    int flat = 0;
    bool done = true;
    // LDS-DMA pieces issued by this thread that have not "landed" yet: applied when the thread reaches __syncthreads()
    // (the vmcnt(0) hipcc puts in front of the barrier) -- the LATEST legal arrival, so a read before the barrier sees
    // stale LDS exactly as it may on the hardware
    struct Pending { void* dst; unsigned char data[16]; };
    std::vector<Pending> pending;
    void land() { for (auto& q : pending) std::memcpy(q.dst, q.data, 16); pending.clear(); }
};

struct State {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    Fiber* cur = nullptr;
    dim3 bdim, gdim, bidx;
    int live = 0, bar_count = 0;
    unsigned bar_gen = 0;
    std::vector<int> wave_live, wave_count;
    std::vector<unsigned> wave_gen;
    std::vector<unsigned char> xchg;        // per thread: 64 bytes of exchange space for wavefront collectives
    void (*body)(void*) = nullptr;
    void* body_arg = nullptr;
};
inline State& S() { static State s; return s; }

inline void yield() { State& s = S(); swapcontext(&s.cur->ctx, &s.sched); }

inline void block_barrier() {
    State& s = S();
    const unsigned gen = s.bar_gen;
    if (++s.bar_count == s.live) { s.bar_count = 0; ++s.bar_gen; return; }
    while (s.bar_gen == gen) yield();
}
inline void wave_barrier() {
    State& s = S();
    const int w = s.cur->flat / WAVE;
    const unsigned gen = s.wave_gen[w];
    if (++s.wave_count[w] == s.wave_live[w]) { s.wave_count[w] = 0; ++s.wave_gen[w]; return; }
    while (s.wave_gen[w] == gen) yield();
}
inline unsigned char* xchg_slot(int flat) { return S().xchg.data() + (size_t)flat * 64; }
inline int lane() { return S().cur->flat % WAVE; }
inline int wave_base() { return S().cur->flat - S().cur->flat % WAVE; }

inline void trampoline() {
    State& s = S();
    Fiber* f = s.cur;
    s.body(s.body_arg);
    f->land();
    f->done = true;
    --s.live;
    --s.wave_live[f->flat / WAVE];
    // whoever is still waiting must not wait for this fiber any more
    if (s.bar_count > 0 && s.bar_count == s.live) { s.bar_count = 0; ++s.bar_gen; }
    const int w = f->flat / WAVE;
    if (s.wave_count[w] > 0 && s.wave_count[w] == s.wave_live[w]) { s.wave_count[w] = 0; ++s.wave_gen[w]; }
    swapcontext(&f->ctx, &s.sched);
}

template <typename F>
inline void launch(dim3 grid, dim3 block, F&& fn) {
    State& s = S();
    const int n = (int)(block.x * block.y * block.z);
    if ((int)s.fibers.size() < n) {
        const size_t old = s.fibers.size();
        s.fibers.resize(n);
        for (size_t i = old; i < s.fibers.size(); ++i) s.fibers[i].stack = (char*)malloc(STACK);
    }
    const int nw = (n + WAVE - 1) / WAVE;
    s.wave_live.assign(nw, 0); s.wave_count.assign(nw, 0); s.wave_gen.assign(nw, 0);
    s.xchg.assign((size_t)n * 64, 0);
    s.bdim = block; s.gdim = grid;
    auto thunk = [](void* p) { (*static_cast<F*>(p))(); };
    s.body = thunk; s.body_arg = (void*)&fn;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.bidx = dim3(bx, by, bz);
                s.live = n; s.bar_count = 0;
                for (int w = 0; w < nw; ++w) { s.wave_live[w] = 0; s.wave_count[w] = 0; }
                for (int t = 0; t < n; ++t) {
                    Fiber& f = s.fibers[t];
                    f.flat = t;
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.done = false;
                    ++s.wave_live[t / WAVE];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = STACK;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, (void (*)())trampoline, 0);
                }
                // Visiting order of the fibers within a scheduling round: HIPCPU_SCHED = forward (default) | reverse |
                // random.  A kernel whose result depends on it has a missing barrier (a data race on the GPU).
                static const int mode = [] {
                    const char* e = getenv("HIPCPU_SCHED");
                    return (e && e[0] == 'r' && e[1] == 'e') ? 1 : (e && e[0] == 'r' && e[1] == 'a') ? 2 : 0;
                }();
                static unsigned long long lcg = 0x9E3779B97F4A7C15ull;
                std::vector<int> order(n);
                for (int t = 0; t < n; ++t) order[t] = (mode == 1) ? n - 1 - t : t;
                int remaining = n;
                while (remaining > 0) {
                    remaining = 0;
                    if (mode == 2)
                        for (int t = n - 1; t > 0; --t) {
                            lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                            const int j = (int)((lcg >> 33) % (unsigned)(t + 1));
                            const int tmp = order[t]; order[t] = order[j]; order[j] = tmp;
                        }
                    for (int k = 0; k < n; ++k) {
                        const int t = order[k];
                        if (s.fibers[t].done) continue;
                        s.cur = &s.fibers[t];
                        swapcontext(&s.sched, &s.fibers[t].ctx);
                        if (!s.fibers[t].done) ++remaining;
                    }
                }
            }
    s.cur = nullptr;
}

}  // namespace hipcpu

#define threadIdx (hipcpu::S().cur->tid)
#define blockIdx (hipcpu::S().bidx)
#define blockDim (hipcpu::S().bdim)
#define gridDim (hipcpu::S().gdim)

static inline void __syncthreads() { hipcpu::S().cur->land(); hipcpu::block_barrier(); }

// blocks run one after the other and a block's fibers never run in parallel: a fence is a no-op, an atomic a plain update
static inline void __threadfence() {}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { const unsigned o = *p; if (v > o) *p = v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    static_assert(sizeof(T) <= 64, "exchange slot too small");
    (void)width;
    std::memcpy(hipcpu::xchg_slot(hipcpu::S().cur->flat), &v, sizeof(T));
    hipcpu::wave_barrier();
    T r;
    std::memcpy(&r, hipcpu::xchg_slot(hipcpu::wave_base() + (hipcpu::lane() ^ mask)), sizeof(T));
    hipcpu::wave_barrier();
    return r;
}
// DPP row_newbcast:K of csrc/rih_mano.hip's skinning block: every lane reads lane K of its own row of 16.  One exchange per
// block of 48 products (the three registers of the block published together), not one per product.
static inline void hipcpu_skin_group(float* T, float g0, float g1, float g2, float w0, float w1, float w2, float w3) {
    const float mine[3] = {g0, g1, g2}, w_[4] = {w0, w1, w2, w3};
    std::memcpy(hipcpu::xchg_slot(hipcpu::S().cur->flat), mine, sizeof(mine));
    hipcpu::wave_barrier();
    const int row0 = hipcpu::wave_base() + (hipcpu::lane() & ~15);
    for (int jj = 0; jj < 4; ++jj)
        for (int c = 0; c < 12; ++c) {
            const int f = jj * 12 + c;
            float g[3];
            std::memcpy(g, hipcpu::xchg_slot(row0 + (f & 15)), sizeof(g));
            T[c] = __builtin_fmaf(g[f >> 4], w_[jj], T[c]);
        }
    hipcpu::wave_barrier();
}
#define RIH_SKIN_GROUP(T, G0, G1, G2, W0, W1, W2, W3) hipcpu_skin_group(T, G0, G1, G2, W0, W1, W2, W3)
static inline int __builtin_amdgcn_readfirstlane_emul(int v) {
    std::memcpy(hipcpu::xchg_slot(hipcpu::S().cur->flat), &v, sizeof(int));
    hipcpu::wave_barrier();
    int r;
    std::memcpy(&r, hipcpu::xchg_slot(hipcpu::wave_base()), sizeof(int));
    hipcpu::wave_barrier();
    return r;
}
#define __builtin_amdgcn_readfirstlane(x) __builtin_amdgcn_readfirstlane_emul(x)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipcpu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
static inline hipError_t hipGetLastError() { return hipSuccess; }
// device query of the persistent kernels (csrc/rih_conv3.hip rih_panel sizes its grid by the CU count): a "device" of HIPCPU_CUS
// compute units (default 6, not a divisor-friendly number on purpose), so that workgroups walk over several tiles on small tests
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    const char* e = std::getenv("HIPCPU_CUS");
    p->multiProcessorCount = e ? std::atoi(e) : 6;
    return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {
    const char* e = std::getenv("HIPCPU_CUS");
    *v = e ? std::atoi(e) : 6;
    return hipSuccess;
}
#include "../hipcpu_gfx950.h"

// shader-clock read of the phase-stamp development aid (csrc/rih_mano.hip): no meaning on the host
static inline long long clock64() { return 0; }
