// rih_hash.h -- the counter-based dropout RNG shared by every kernel that draws or re-draws a mask (rih_add_dropout,
// rih_dropout_bwd, rih_softmax_fwd / _bwd, the fused attention kernels): element `idx` of the mask stream `seed` is KEPT iff
//     rih_hash(seed, idx) >= p * 2^32            (kept values are scaled by 1 / (1 - p)).
// 32 uniform bits = a 32-bit avalanche finalizer (two 32-bit multiplies, "lowbias32") of the element index, keyed TWICE by the
// 64-bit seed: key word A is xor-ed into the index in front of the first multiply, key word B is added between the two multiplies
// (round 4: with word A alone every mask stream was an xor-re-indexing of ONE fixed 2^32-entry sequence, round-3 advisor
// finding; the addition between the multiplies does not commute with the xor-shifts, so two seeds give unrelated sequences,
// and both seed words enter both key words).  The high index word (tensors of >= 2^32 elements) enters through one more
// multiply, so that the 32-bit and the 64-bit form agree wherever both apply.  Cheap on purpose: the attention kernels draw 16
// values per lane and key tile, and a 64-bit-multiply hash (three 64 x 64 products = a dozen quarter-rate 32-bit multiplies per
// value) cost them more than their matrix products.  numpy mirrors: tests/abi_emulator.py::hash_np, tests/test_gpu_ops.py::_hash_np.
#pragma once
#include <stdint.h>

#ifndef RIH_HASH_FN
#define RIH_HASH_FN __device__ __forceinline__
#endif

RIH_HASH_FN uint32_t rih_mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// the key of a mask stream: word A in the low half, word B in the high half
RIH_HASH_FN uint64_t rih_seed_key(uint64_t seed) {
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const uint32_t ka = rih_mix32(lo) ^ rih_mix32(hi ^ 0x9E3779B9u);
    const uint32_t kb = rih_mix32(lo ^ 0x85EBCA6Bu) + rih_mix32(hi + 0xC2B2AE35u);
    return (uint64_t)ka | ((uint64_t)kb << 32);
}
RIH_HASH_FN uint32_t rih_mix32k(uint32_t x, uint32_t kb) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x += kb;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RIH_HASH_FN uint32_t rih_hash_k32(uint64_t key, uint32_t idx) { return rih_mix32k(idx ^ (uint32_t)key, (uint32_t)(key >> 32)); }
RIH_HASH_FN uint32_t rih_hash_k64(uint64_t key, uint64_t idx) {
    return rih_mix32k((uint32_t)idx ^ (uint32_t)key ^ ((uint32_t)(idx >> 32) * 0x85EBCA6Bu), (uint32_t)(key >> 32));
}
RIH_HASH_FN uint32_t rih_hash(uint64_t seed, uint64_t idx) { return rih_hash_k64(rih_seed_key(seed), idx); }
