// nq_common.h -- shared definitions of the gfx950 NIQKI engine (device + host
// side of libniqki_hip.so).  Integer arithmetic of the hot path as specified
// by SURVEY.md appendix C; reference lines are cited per function.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NQ_HD __host__ __device__ __forceinline__

namespace nq {

constexpr uint32_t kWave = 64;             // gfx950 wavefront
constexpr uint32_t kEmpty32 = 0xFFFFFFFFu; // int32 -1: empty sketch cell
constexpr uint16_t kEmpty16 = 0xFFFFu;     // empty / invalid cell in the u16 sketch store

constexpr uint64_t kRevMul = 0xD6E8FEB86659FD93ULL;   // src/niqki_index.cpp:292-293
constexpr uint64_t kUnrevMul = 0xCFEE444D8B59A89BULL; // src/niqki_index.cpp:301-302

// Derived constants of one index (constructor body, src/niqki_index.cpp:16-29)
struct Derived {
  uint32_t K, S, W, H, M;
  uint32_t F;          // 1<<S
  uint32_t R;          // 1<<W fingerprint range
  uint32_t mask_m;     // (1<<M)-1
  uint32_t max_rem;    // (1<<H)-1
  uint32_t min_score;
  uint32_t slot_begin, slot_end; // shard's slot range
  uint64_t kmer_mask;  // 4^K - 1
};

NQ_HD uint64_t mix64(uint64_t x, uint64_t c) {
  x = ((x >> 32) ^ x) * c;
  x = ((x >> 32) ^ x) * c;
  return (x >> 32) ^ x;
}
// revhash64 / unrevhash64: src/niqki_index.cpp:291-296, :300-305
NQ_HD uint64_t rev64(uint64_t x) { return mix64(x, kRevMul); }
NQ_HD uint64_t unrev64(uint64_t x) { return mix64(x, kUnrevMul); }

NQ_HD uint32_t clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return x ? (uint32_t)__clzll((long long)x) : 64u;
#else
  return x ? (uint32_t)__builtin_clzll(x) : 64u;
#endif
}

// get_fingerprint: src/niqki_index.cpp:277-287; h == 0 gives 0 (bsr(0) is UB
// in the reference, observed 0).
NQ_HD uint32_t fingerprint(uint64_t h, uint32_t M, uint32_t mask_m, uint32_t max_rem) {
  uint32_t lz = clz64(h);
  uint32_t rem = lz < max_rem ? max_rem - lz : 0u;
  return ((uint32_t)h & mask_m) + (rem << M);
}

// Sketch slot of a canonical k-mer: src/niqki_index.cpp:347.  Only the top
// half of the second product is needed: the final xor leaves the top 32 bits
// unchanged and S <= 15.
NQ_HD uint32_t slot_of(uint64_t canon, uint32_t S) {
  uint64_t x = ((canon >> 32) ^ canon) * kUnrevMul;
  x = ((x >> 32) ^ x) * kUnrevMul;
  return (uint32_t)(x >> (64 - S));
}

}  // namespace nq
