// nq_synth.h -- deterministic synthetic genome generator (counter based), the
// same integer function on host and device.  Measurement / parity input only
// (SURVEY.md 8d); not part of the reference's path.
//
// Genome (family f, member m, rate r/16384) of length L:
//   ancestor base at p  = 2 bits of mix(key_anc(seed,f) + (p>>5)) at (p&31)
//   substitution at p   : 16 bits u of mix(key_mut(seed,f,m) + (p>>2)) at (p&3);
//                         if (u & 0x3FFF) < r the base becomes
//                         (base + 1 + (u>>14)%3) & 3
//   ASCII = "ACGT"[base]
// rate14 == 0 gives the ancestor itself.
#pragma once
#include "nq_common.h"

namespace nq {

NQ_HD uint64_t smix(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

NQ_HD uint64_t synth_key_anc(uint64_t seed, uint32_t family) {
  return smix(smix(seed) ^ (0xA5A5A5A500000000ULL + family));
}
NQ_HD uint64_t synth_key_mut(uint64_t seed, uint32_t family, uint32_t member) {
  return smix(smix(seed + 1) ^ (((uint64_t)family << 32) | member));
}

// 32 bases starting at block*32, as 2-bit codes little-endian in a u64
// (base p at bits 2*(p&31)).
NQ_HD uint64_t synth_block(uint64_t key_anc, uint64_t key_mut, uint32_t rate14, uint64_t block) {
  uint64_t codes = smix(key_anc + block);
  if (rate14) {
    for (uint32_t q = 0; q < 8; ++q) {
      uint64_t r = smix(key_mut + block * 8 + q);
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t u = (uint32_t)(r >> (16 * j)) & 0xFFFFu;
        if ((u & 0x3FFFu) < rate14) {
          uint32_t pos = q * 4 + j;
          uint64_t b = (codes >> (2 * pos)) & 3u;
          b = (b + 1 + ((u >> 14) % 3u)) & 3u;
          codes = (codes & ~(3ULL << (2 * pos))) | (b << (2 * pos));
        }
      }
    }
  }
  return codes;
}

// second layer of substitutions on a block (a read sampled from a genome: the genome's own
// substitutions first, then the read's, keyed by key_mut2)
NQ_HD uint64_t synth_block2(uint64_t key_anc, uint64_t key_mut, uint32_t rate14, uint64_t key_mut2, uint32_t rate14_2,
                            uint64_t block) {
  uint64_t codes = synth_block(key_anc, key_mut, rate14, block);
  if (rate14_2) {
    for (uint32_t q = 0; q < 8; ++q) {
      uint64_t r = smix(key_mut2 + block * 8 + q);
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t u = (uint32_t)(r >> (16 * j)) & 0xFFFFu;
        if ((u & 0x3FFFu) < rate14_2) {
          uint32_t pos = q * 4 + j;
          uint64_t b = (codes >> (2 * pos)) & 3u;
          b = (b + 1 + ((u >> 14) % 3u)) & 3u;
          codes = (codes & ~(3ULL << (2 * pos))) | (b << (2 * pos));
        }
      }
    }
  }
  return codes;
}
NQ_HD uint64_t synth_key_read(uint64_t seed, uint32_t family, uint32_t read_id) {
  return smix(smix(seed + 2) ^ (((uint64_t)family << 32) | read_id));
}

NQ_HD uint8_t synth_ascii(uint32_t code) {
  // A C G T = 0x41 0x43 0x47 0x54
  return (uint8_t)((0x54474341u >> (8 * code)) & 0xFFu);
}

}  // namespace nq
