// nq_query.hip -- kernel #4: gather-histogram query over the inverted index,
// and the threshold / compaction / ordering of the hits, for gfx950.
//
// Replaces Index::query_sketch (src/niqki_index.cpp:633-687): the counting
// loop (:652-661) is gather_kernel, the threshold (:662-666) and the
// descending (count, gid) order (:685) are the hits_* kernels.
//
// gather_kernel: one workgroup per query, the genome tiles walked one after
// another.  The tile's per-genome hit counters live in LDS as packed u16 pairs (a
// count never exceeds F <= 2^15, so the two halves of a word cannot carry into
// each other) and are bumped with ds_add_u32.  Each wave takes 64 sketch slots at
// a time: every lane looks up the bucket of its slot for ALL tiles with one random
// table access (the other tiles' entries are parked in a coalesced stash), then
// the wave walks the 64 buckets one after another, all lanes reading consecutive
// u16 genome ids of one 128-byte aligned bucket.  HBM-bound by design:
// algorithmic bytes per query = 4T + 20F.  On large indexes the
// queries of a launch are first put into a locality order (probe_kernel,
// order_kernel): similar queries then run on the same XCD at the same time and
// share their table and bucket lines through its L2.  For real batches the table
// look-ups are taken out of the kernel altogether by a pre-pass (lookup_rows_kernel
// / lookup_kernel) that walks the table slot block by slot block for all queries of
// the launch; the gather kernel then reads one packed word per (tile, slot).
#include "nq_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace nq {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

__device__ __forceinline__ void bump(uint32_t *cnt, uint32_t g) {
  atomicAdd(&cnt[g >> 1], 1u << ((g & 1u) * 16u));  // ds_add_u32, result unused
}
// Branch-free form for the bucket walks: lanes past the end of the bucket add 0 to a
// word of their own (no exec-mask juggling, no same-address serialisation).
__device__ __forceinline__ void bump_if(uint32_t *cnt, uint32_t g, bool on, uint32_t lane) {
  const uint32_t gg = on ? g : 2u * lane;
  const uint32_t inc = on ? (1u << ((g << 4) & 31u)) : 0u;   // 1 << 16 * (g & 1): the shifter takes the low 5 bits
  atomicAdd(&cnt[gg >> 1], inc);
}

// A bucket chunk: up to 64 consecutive ids at unit `pos` (1 << align_log2 ids per
// unit, counted from the tile's base).
struct Item {
  uint32_t pos, len;
};
constexpr uint32_t kQueue = 256;  // items per wave-private LDS work queue

// Walks 64 chunks held one per lane (pos, len <= 64): all lanes read consecutive
// u16 ids of one chunk, UNROLL chunks per round trip, two rounds in flight (the
// loads of round r+1 are issued before the LDS atomics of round r; unconditional
// loads, lanes past a chunk's end read what follows it and are masked).
// The form of indexes that are not padded (and of the measurement modes); padded ones take PairWalk below.
// n (wave uniform, PARTIAL only): how many of the 64 lanes hold a chunk -- the last batch of a tile's walk; a short
// read's whole walk is such a batch of ~10 chunks per wave, and walking all 64 places cost it half of its instructions.
template <int UNROLL, int MODE, bool PARTIAL = false>
__device__ __forceinline__ void walk64(const uint16_t *gl, uint32_t a, uint32_t pos, uint32_t len,
                                       uint32_t lane, uint32_t *cnt, uint32_t &sink, uint32_t n = 64) {
  uint32_t ga[UNROLL], gb[UNROLL];
  auto fetch = [&](uint32_t j0, uint32_t (&g)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t b = __builtin_amdgcn_readlane(pos, j0 + u);
      g[u] = (gl + ((uint64_t)b << a))[lane];
    }
  };
  auto apply = [&](uint32_t j0, uint32_t (&g)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t l = __builtin_amdgcn_readlane(len, j0 + u);
      if (MODE == 1) { if (lane < l) sink ^= g[u]; }
      else bump_if(cnt, g[u], lane < l, lane);
    }
  };
  fetch(0, ga);
  if constexpr (PARTIAL) {
#pragma unroll
    for (uint32_t j0 = 0; j0 < 64; j0 += 2 * UNROLL) {
      const bool more1 = j0 + UNROLL < n;   // (wave uniform)
      if (more1) fetch(j0 + UNROLL, gb);
      apply(j0, ga);
      if (!more1) break;
      const bool more2 = j0 + 2 * UNROLL < 64 && j0 + 2 * UNROLL < n;
      if (more2) fetch(j0 + 2 * UNROLL, ga);
      apply(j0 + UNROLL, gb);
      if (!more2) break;
    }
    return;
  }
#pragma unroll
  for (uint32_t j0 = 0; j0 < 64; j0 += 2 * UNROLL) {
    fetch(j0 + UNROLL, gb);
    apply(j0, ga);
    if (j0 + 2 * UNROLL < 64) fetch(j0 + 2 * UNROLL, ga);
    apply(j0 + UNROLL, gb);
  }
}

// The padded walk's form (PAD, 64-id lines): ONE load instruction fetches TWO lines -- lanes 0..31 read
// the 32 dwords of one chunk's line, lanes 32..63 those of the next chunk's -- and every lane counts the two
// ids of its dword.  The CU's texture-address unit takes a wave-wide load every ~7.7 cycles whatever its
// width (tools/ubench_lines.hip: 271 loads per microsecond and CU from L2, ushort or dword), so at one
// line per instruction the walk was bound by load issue (31 SIMD cycles per line) before HBM; the lanes
// take their chunk's position straight from the wave's LDS queue (one ds_read_b32 per load, its queue
// slot in the instruction's offset field) instead of a v_readlane + s_lshl per line.  Per line now:
// 0.5 load, 0.5 LDS read, ~5 vector ALU ops, 1 LDS atomic per lane and id (2 per load).
template <int UNROLL>
struct PairWalk {
  static_assert(UNROLL == 16 || UNROLL == 32, "a round is UNROLL lines = UNROLL / 2 loads; two or four rounds per batch");
  static constexpr int L = UNROLL / 2;   // loads per round
  const uint8_t *lane_base;              // tile's ids + this lane's dword within a line
  uint32_t half;                         // lane >> 5: which chunk of a pair this lane reads
  uint32_t ga[L], gb[L];

  __device__ __forceinline__ void init(const uint16_t *gl, uint32_t lane) {
    lane_base = (const uint8_t *)gl + (lane & 31u) * 4u;
    half = lane >> 5;
  }
  // loads of the chunks [j0, j0 + UNROLL) of the batch whose items start at `items` (LDS, this wave's queue)
  __device__ __forceinline__ void fetch(uint32_t (&g)[L], const Item *items, uint32_t j0) {
    const Item *mine = items + half;
    uint32_t pos[L];   // line index within the tile (align_log2 = 6); all LDS reads first, one wait for them
#pragma unroll
    for (int k = 0; k < L; ++k) pos[k] = mine[j0 + 2 * k].pos;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < L; ++k) {
      uint64_t addr;   // lane_base + 128 * pos in one vector op
      asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(addr) : "v"(pos[k]), "s"(128u), "v"((uint64_t)lane_base) : "vcc");
      g[k] = *(const __attribute__((address_space(1))) uint32_t *)addr;   // (global_load, not flat)
    }
  }
  __device__ __forceinline__ void apply(uint32_t (&g)[L]) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
      // per id: word g >> 1, increment 1 << 16 * (g & 1); the counters start at LDS address 0, so the word's
      // byte offset IS its LDS address.  and / add issue in 2.8 cycles on gfx950, the mad in 4.4
      // (profiles/r03_opcode_costs.txt).
      uint32_t even, addr, odd, inc, hi;
      asm("v_and_b32 %0, 0xfffe, %1" : "=v"(even) : "v"(g[k]));
      asm("v_add_u32 %0, %1, %1" : "=v"(addr) : "v"(even));
      asm("v_and_b32 %0, 1, %1" : "=v"(odd) : "v"(g[k]));
      asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(odd), "s"(0xFFFFu));
      __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(g[k]));
      asm("v_and_b32 %0, 0xfffe, %1" : "=v"(even) : "v"(hi));
      asm("v_add_u32 %0, %1, %1" : "=v"(addr) : "v"(even));
      asm("v_and_b32 %0, 1, %1" : "=v"(odd) : "v"(hi));
      asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(odd), "s"(0xFFFFu));
      __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
#ifdef NQ_ABLATION
  // COST MODEL of a one-tile walk (measurement only, wrong counters): per id what byte counters for
  // 100 000 genomes in one tile would cost on top of the same lines -- a half-select by the id's position in its line
  // (ids stay 16 bits: a bucket as [ids below the split | ids above]), the byte's increment 1 << 8 (id & 3) as a
  // RETURNING ds_add, and the test whether the byte just wrapped (its old value 0xFF: the carry has to be logged).
  uint32_t sa[L], sb[L];   // the chunks' split positions (stand-in: their lengths, fetched with the positions)
  __device__ __forceinline__ void fetch_model(uint32_t (&g)[L], uint32_t (&sp)[L], const Item *items, uint32_t j0) {
    const Item *mine = items + half;
    uint32_t pos[L];
#pragma unroll
    for (int k = 0; k < L; ++k) { const Item it = mine[j0 + 2 * k]; pos[k] = it.pos; sp[k] = it.len; }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < L; ++k) {
      uint64_t addr;
      asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(addr) : "v"(pos[k]), "s"(128u), "v"((uint64_t)lane_base) : "vcc");
      g[k] = *(const __attribute__((address_space(1))) uint32_t *)addr;
    }
  }
  // PARTS 4 / 5 / 6 take the SHIPPED walk apart instead: 4 = its loads and vector work without the LDS atomics (the ids
  // are folded into a register), 5 = one of its two atomics per dword, 6 = the chunks' positions by v_readlane from one
  // queue read per batch instead of a ds_read_b32 per load.
  __device__ __forceinline__ bool apply_part(uint32_t (&g)[L], int part) {
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      uint32_t even, addr, odd, inc, hi;
      asm("v_and_b32 %0, 0xfffe, %1" : "=v"(even) : "v"(g[k]));
      asm("v_add_u32 %0, %1, %1" : "=v"(addr) : "v"(even));
      asm("v_and_b32 %0, 1, %1" : "=v"(odd) : "v"(g[k]));
      asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(odd), "s"(0xFFFFu));
      if (part == 4) acc ^= addr + inc;
      else __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm("v_lshrrev_b32 %0, 16, %1" : "=v"(hi) : "v"(g[k]));
      asm("v_and_b32 %0, 0xfffe, %1" : "=v"(even) : "v"(hi));
      asm("v_add_u32 %0, %1, %1" : "=v"(addr) : "v"(even));
      asm("v_and_b32 %0, 1, %1" : "=v"(odd) : "v"(hi));
      asm("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(odd), "s"(0xFFFFu));
      if (part == 4 || part == 5) acc ^= addr + inc;
      else __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return acc == 0x12345u;
  }
  __device__ __forceinline__ void fetch_readlane(uint32_t (&g)[L], uint32_t posreg, uint32_t j0) {
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const uint32_t pa = __builtin_amdgcn_readlane(posreg, j0 + 2 * k), pb = __builtin_amdgcn_readlane(posreg, j0 + 2 * k + 1);
      const uint32_t pos = half ? pb : pa;
      uint64_t addr;
      asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(addr) : "v"(pos), "s"(128u), "v"((uint64_t)lane_base) : "vcc");
      g[k] = *(const __attribute__((address_space(1))) uint32_t *)addr;
    }
  }
  template <int PART>
  __device__ __forceinline__ bool batch_part(const Item *items, uint32_t lane) {
    constexpr int R = 64 / UNROLL;
    bool x = false;
    if (PART == 6) {
      const uint32_t posreg = items[lane].pos;
      fetch_readlane(ga, posreg, 0);
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        fetch_readlane(gb, posreg, (r + 1) * UNROLL);
        apply(ga);
        if (r + 2 < R) fetch_readlane(ga, posreg, (r + 2) * UNROLL);
        apply(gb);
      }
      return false;
    }
    fetch(ga, items, 0);
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      fetch(gb, items, (r + 1) * UNROLL);
      x |= apply_part(ga, PART);
      if (r + 2 < R) fetch(ga, items, (r + 2) * UNROLL);
      x |= apply_part(gb, PART);
    }
    return x;
  }
  template <int PARTS>   // 3: all of it; 1: returning add + wrap test only; 2: half-select only (plain ds_add on the byte)
  __device__ __forceinline__ bool apply_model(uint32_t (&g)[L], uint32_t (&sp)[L], uint32_t p0) {
    bool carry = false;
#pragma unroll
    for (int k = 0; k < L; ++k) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t id = h ? g[k] >> 16 : g[k];                          // (the low id needs no mask: the and below drops the rest)
        const uint32_t off = (PARTS & 2) ? ((p0 + (uint32_t)h >= sp[k]) ? 0x8000u : 0u) : 0u;   // which half of the genomes (model: an offset that stays inside the counters)
        const uint32_t addr = (id & 0xFFFCu) | off;
        const uint32_t sh = (id << 3) & 31u;
        if (PARTS & 1) {
          const uint32_t old = __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          carry |= ((old >> sh) & 0xFFu) == 0xFFu;
        } else {
          __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    return carry;
  }
  template <int PARTS>
  __device__ __forceinline__ bool batch_model(const Item *items, uint32_t p0) {
    constexpr int R = 64 / UNROLL;
    bool carry = false;
    fetch_model(ga, sa, items, 0);
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      fetch_model(gb, sb, items, (r + 1) * UNROLL);
      carry |= apply_model<PARTS>(ga, sa, p0);
      if (r + 2 < R) fetch_model(ga, sa, items, (r + 2) * UNROLL);
      carry |= apply_model<PARTS>(gb, sb, p0);
    }
    return carry;
  }
#endif
  // 64 chunks: items[0 .. 64) of the wave's queue; all their lines in flight before the first id is counted
  // (keeping a round in flight across batches, over the cutting of the next buckets, gained nothing: measured)
  __device__ __forceinline__ void batch(const Item *items) {
    constexpr int R = 64 / UNROLL;
    fetch(ga, items, 0);
#pragma unroll
    for (int r = 0; r < R; r += 2) {
      fetch(gb, items, (r + 1) * UNROLL);
      apply(ga);
      if (r + 2 < R) fetch(ga, items, (r + 2) * UNROLL);
      apply(gb);
    }
  }
};

// One pass of a workgroup over all slots of one tile.
//   STASH_OUT: the lookup fetched the entries of NT tiles at once; tile 0 is
//              walked now, the others are parked in `stash` (coalesced) for the
//              later passes, so every slot costs ONE random table line per query.
//   STASH_IN : entries come from the stash (coalesced 8-byte loads).
// Each wave takes 64 slots per iteration; lookups run one iteration ahead,
// fingerprints two.  Buckets are cut into chunks of <= 64 ids that go through a
// wave-private LDS queue, so the walk always runs on full batches of 64 chunks
// whatever the bucket lengths are.
// MODE is a measurement aid (results are wrong for MODE != 0): 1 = no LDS
// atomics, 6 = lookups only, 7 = counters not written back, 8 = no walk (zero + write-back only).  MODE != 0 is instantiated only in -DNQ_ABLATION builds;
// the shipped library cannot be switched into it.
//   PRE      : entries come from the slot-major look-up pre-pass (lookup_kernel below): one
//              packed word per (query, tile, slot) = bucket start relative to the slot's first
//              unit << 16 | length, read coalesced; no table access in this kernel at all.
//   PAD      : padded index (whole-line chunks): batches go through PairWalk (two lines per load), and the four
//              waves of a SIMD take turns at the top issue priority; otherwise walk64 with explicit lengths.
//   ahead    : (PRE) this wave's first two look-ups of the tile, issued a tile earlier by the caller / this function.
template <int BLOCK, int UNROLL, int NT, bool STASH_OUT, bool STASH_IN, int MODE, bool PRE = false, bool PAD = false>
__device__ __forceinline__ void walk_tile(const IndexView &v, const int32_t *sk, uint32_t q, uint32_t t,
                                          uint32_t *cnt, Item *queue, Entry *stash, uint32_t &sink,
                                          uint32_t it_lo, uint32_t n_it, const uint32_t *pre = nullptr,
                                          Entry *ahead = nullptr) {
  // slots [64 * it_lo, min(64 * n_it, f_local)): one pass of the kernel
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t NW = BLOCK / 64;
  constexpr int NE = STASH_OUT ? NT : 1;  // entries fetched per lookup
  const uint32_t R = v.d.R, a = v.align_log2;
  const uint16_t *gl = v.gids + v.tile_base[t];
  Entry *my_stash = PRE ? nullptr : stash + (uint64_t)q * (v.n_tiles - 1) * v.f_local;
  const uint32_t *my_pre = PRE ? pre + ((uint64_t)q * v.n_tiles + t) * v.f_local : nullptr;
  const uint32_t *my_units = v.slot_units + (uint64_t)t * (v.f_local + 1);
  Item *wq = queue + wave * kQueue;
  uint32_t q_head = 0, q_count = 0;  // wave-uniform
  constexpr bool PAIR = PAD && (MODE == 0 || (MODE >= 9 && MODE <= 14)) && (UNROLL == 16 || UNROLL == 32);
  uint32_t prio_turn = wave >> 2;   // the four waves of a SIMD take turns at the issue arbiter's top priority (PAIR)
  PairWalk<PAIR ? UNROLL : 32> pw;
  pw.init(gl, lane);

  struct Look { Entry e[NE]; };
  // HM: a single-tile index with its per-slot class mask (IndexView::hmask): a fingerprint whose class holds no
  // bucket in its slot is never looked up, its table line never requested (the masked loop below).
  constexpr bool HM_FORM = NE == 1 && NT == 1 && !STASH_IN && !PRE;
  const uint16_t *hm = HM_FORM ? v.hmask : nullptr;
  auto slot_ok = [&](uint32_t it) -> bool { return it < n_it && it * 64 + lane < v.f_local; };
  auto load_fp = [&](uint32_t it) -> int32_t {
    if (STASH_IN || PRE) return 0;
    const uint32_t s = it * 64 + lane;
    return sk[s < v.f_local ? s : v.f_local - 1];
  };
  auto valid_of = [&](uint32_t it, int32_t fp) -> bool {
    if (STASH_IN || PRE) return slot_ok(it);
    return slot_ok(it) && fp >= 0 && (uint32_t)fp < R;  // src/niqki_index.cpp:654
  };
  // unconditional loads from clamped addresses: they stay in flight across the walk
  auto lookup = [&](uint32_t it, int32_t fp) -> Look {
    Look L;
    uint32_t s = it * 64 + lane;
    if (s >= v.f_local) s = v.f_local - 1;
    if (PRE) {
      const uint32_t w = my_pre[s];
      L.e[0] = Entry{my_units[s] + (w >> 16), w & 0xFFFFu};
    } else if (STASH_IN) {
      L.e[0] = my_stash[(uint64_t)(t - 1) * v.f_local + s];
    } else {
      const bool ok = fp >= 0 && (uint32_t)fp < R;
      const Entry *p = v.entries + ((uint64_t)s * R + (ok ? (uint32_t)fp : 0u)) * v.n_tiles + (STASH_OUT ? 0u : t);
#pragma unroll
      for (int k = 0; k < NE; ++k) L.e[k] = p[k];
    }
    return L;
  };
  auto drain = [&]() {
    while (q_count >= 64) {
      if constexpr (PAIR) {
        // the issue arbiter serves the oldest wave of a SIMD first: left alone, the four waves of a SIMD finish
        // a tile 3 us apart (18 / 20 / 23 / 26 us) and the youngest runs the last stretch alone
        prio_turn = (prio_turn + 1) & 3;
        if (prio_turn == 0) __builtin_amdgcn_s_setprio(0);
        else if (prio_turn == 1) __builtin_amdgcn_s_setprio(1);
        else if (prio_turn == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
#ifdef NQ_ABLATION
        if constexpr (MODE == 12 || MODE == 13 || MODE == 14) {
          if (__any(pw.template batch_part<MODE == 12 ? 4 : MODE == 13 ? 5 : 6>(wq + q_head, lane))) sink += 1u;
        } else if constexpr (MODE == 9 || MODE == 10 || MODE == 11) {
          // (a wrapped byte would be logged: one append to a per-query list in global memory, by the lanes that saw one)
          if (__any(pw.template batch_model<MODE == 9 ? 3 : MODE == 10 ? 1 : 2>(wq + q_head, 2u * (lane & 31u)))) sink += 1u;
        } else
#endif
        pw.batch(wq + q_head);             // (q_head is a multiple of 64: a batch never wraps; a wave's LDS traffic is in order)
        q_head = (q_head + 64) & (kQueue - 1);
        q_count -= 64;
        continue;
      }
      const Item x = wq[(q_head + lane) & (kQueue - 1)];
      q_head = (q_head + 64) & (kQueue - 1);
      q_count -= 64;
      walk64<UNROLL, MODE>(gl, a, x.pos, x.len, lane, cnt, sink);
    }
  };

  // cut the wave's 64 buckets (start `pos`, `rem` ids left; rem = 0: none) into chunks of <= 64 ids, at most 3 per
  // lane and round (q_count < 64 on entry, so at most 63 + 192 items are ever queued), and walk full batches
  auto cut_and_walk = [&](uint32_t pos, uint32_t rem) {
    const uint32_t step = 64u >> a;  // units per 64 ids (align_log2 <= 6)
    do {
      uint32_t nch = (rem + 63) >> 6;
      if (nch > 3) nch = 3;
      // exclusive prefix of nch (two bits) over the lanes from two ballots: bit counts below the lane (v_mbcnt) instead
      // of a six-step shuffle scan through the LDS crossbar
      const uint64_t b0 = __ballot((nch & 1u) != 0), b1 = __ballot((nch & 2u) != 0);
      const uint32_t excl = __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                            2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
      const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1);
      uint32_t slot = q_head + q_count + excl;
#pragma unroll
      for (uint32_t k = 0; k < 3; ++k)
        if (k < nch) {
          const uint32_t left = rem - 64 * k;
          wq[(slot + k) & (kQueue - 1)] = Item{pos + step * k, left < 64 ? left : 64u};
        }
      q_count += total;
      pos += step * nch;
      rem -= rem < 192 ? rem : 192u;
      drain();
    } while (__any(rem != 0));
  };

  if (HM_FORM && hm) {
    // Masked form (single-tile index with its per-slot class mask): most of a query's fingerprints fall into
    // classes that hold no bucket in their slot -- for a short read against a genome index, ~99 % -- so the loop
    // is bound by the latency of its own input.  All loads of kDeep iterations are issued before any is used:
    // fingerprints and mask words first, then (only for lanes that still hold a fingerprint, only in iterations
    // where some lane does) the table entries.
    constexpr int kDeep = 8;
    for (uint32_t it0 = it_lo + wave; it0 < n_it; it0 += kDeep * NW) {
      int32_t fpv[kDeep];
      uint32_t mw[kDeep];
#pragma unroll
      for (int k = 0; k < kDeep; ++k) {
        uint32_t s = (it0 + k * NW) * 64 + lane;
        s = s < v.f_local ? s : v.f_local - 1;
        fpv[k] = sk[s];
        mw[k] = hm[s];
      }
      Entry en[kDeep];
      bool live[kDeep];
#pragma unroll
      for (int k = 0; k < kDeep; ++k) {
        const uint32_t itk = it0 + k * NW;
        const int32_t fp = fpv[k];
        const bool ok = slot_ok(itk) && fp >= 0 && (uint32_t)fp < R && ((mw[k] >> ((uint32_t)fp >> v.hmask_shift)) & 1u);   // :654
        live[k] = __any(ok);
        en[k] = Entry{0u, 0u};
        if (ok) {
          uint32_t s = itk * 64 + lane;
          en[k] = v.entries[((uint64_t)s * R + (uint32_t)fp) * v.n_tiles + t];
        }
      }
#pragma unroll
      for (int k = 0; k < kDeep; ++k)
        if (live[k]) {   // (wave-uniform)
          if (MODE == 6) { sink += en[k].start ^ en[k].len; continue; }
          cut_and_walk(en[k].start, en[k].len);
        }
    }
    if (q_count) {  // the last partial batch; "no chunk" = the tile's spare line of padding ids
      if constexpr (PAIR) {
        if (lane >= q_count) wq[q_head + lane] = Item{my_units[v.f_local], 0u};
        pw.batch(wq + q_head);
      } else {
        Item x = wq[(q_head + lane) & (kQueue - 1)];
        if (lane >= q_count) x = Item{PAD ? my_units[v.f_local] : 0u, 0u};
        walk64<(UNROLL > 8 ? 8 : UNROLL), MODE, true>(gl, a, x.pos, x.len, lane, cnt, sink, q_count);
      }
    }
    return;
  }

  // software pipeline: table look-ups run two iterations ahead of the walk, fingerprints
  // three (a pass over short buckets has little walk work to hide a look-up behind)
  uint32_t it = it_lo + wave;
  int32_t fp0 = load_fp(it);
  int32_t fp1 = load_fp(it + NW);
  int32_t fp2 = load_fp(it + 2 * NW);
  Look cur, nxt;
  if (PRE && ahead) {
    // `ahead`: this wave's first two look-ups of the tile, issued during the walk of the tile before
    // (a slot shard has only a few iterations per tile: nothing else hides their latency); the ones of
    // the next tile are put on their way now
    cur.e[0] = ahead[0];
    nxt.e[0] = ahead[1];
    if (t + 1 < v.n_tiles) {
      const uint32_t *np = my_pre + v.f_local, *nu = my_units + (v.f_local + 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        uint32_t s = (it + k * NW) * 64 + lane;
        if (s >= v.f_local) s = v.f_local - 1;
        const uint32_t w = np[s];
        ahead[k] = Entry{nu[s] + (w >> 16), w & 0xFFFFu};
      }
    }
  } else {
    cur = lookup(it, fp0);
    nxt = lookup(it + NW, fp1);
  }
  bool cur_ok = valid_of(it, fp0);
  bool nxt_ok = valid_of(it + NW, fp1);

  for (; it < n_it; it += NW) {
    const Look nxt2 = lookup(it + 2 * NW, fp2);
    const bool nxt2_ok = valid_of(it + 2 * NW, fp2);
    fp2 = load_fp(it + 3 * NW);
    uint32_t pos = cur.e[0].start, rem = cur_ok ? cur.e[0].len : 0u;
    if (STASH_OUT) {
      const uint32_t s = it * 64 + lane;
      if (s < v.f_local) {
#pragma unroll
        for (int k = 1; k < NE; ++k)
          my_stash[(uint64_t)(k - 1) * v.f_local + s] = Entry{cur.e[k].start, cur_ok ? cur.e[k].len : 0u};
      }
    }
    if (MODE == 6) { sink += pos ^ rem; cur = nxt; cur_ok = nxt_ok; nxt = nxt2; nxt_ok = nxt2_ok; continue; }
    cut_and_walk(pos, rem);
    cur = nxt;
    cur_ok = nxt_ok;
    nxt = nxt2;
    nxt_ok = nxt2_ok;
  }
  if (q_count) {  // the last partial batch; "no chunk" = the tile's spare line of padding ids
    if constexpr (PAIR) {
      if (lane >= q_count) wq[q_head + lane] = Item{my_units[v.f_local], 0u};
      pw.batch(wq + q_head);
    } else {
      Item x = wq[(q_head + lane) & (kQueue - 1)];
      if (lane >= q_count) x = Item{PAD ? my_units[v.f_local] : 0u, 0u};
      walk64<(UNROLL > 8 ? 8 : UNROLL), MODE, true>(gl, a, x.pos, x.len, lane, cnt, sink, q_count);
    }
  }
}

// ---- slot-major look-up pre-pass ---------------------------------------------------------
// In gather_kernel's own look-up every (query, slot) costs one random 128-byte table line of
// which 8 * n_tiles bytes are used: 4.2 MB of a 14 MB query at the north-star shape.  For a real
// batch the table is better walked slot block by slot block for all queries of the launch:
//   lookup_kernel  one workgroup = PS consecutive slots x 256 queries (one per thread).  A thread
//                  reads its query's PS fingerprints (one whole 128-byte line of the sketch for
//                  PS = 32), looks each one up with one 8 * n_tiles byte load, 8 slots in flight
//                  at a time, packs the results and stores, per tile, PS words = one whole line
//                  of pre[q][t][s].
// All workgroups of one slot block (nq / 256 of them) are neighbours in one XCD's dispatch order
// (block ids x, x+8, ...): they run at the same time on that XCD and walk the block's rows in
// the same order, so a table line is fetched from HBM once and served from the XCD's L2 to the
// other queries that need it (4096 queries into 512 lines per slot).
// Packed word: (bucket start - first unit of its slot) << 16 | length; launch_lookup_usable()
// says whether both halves fit 16 bits for the index at hand.
// HBM traffic per launch: the table once + 4 bytes per (query, slot) in and 4 * n_tiles out,
// instead of 128 bytes per (query, slot).
constexpr uint32_t kXcds = 8;
constexpr uint32_t kPreBlock = 256;      // queries per lookup workgroup, one per thread
constexpr uint32_t kPreFlight = 8;       // look-ups in flight per thread

// NT tiles from tile t0 on per launch: an index of more than 4 tiles takes several launches.
template <int NT, int PS>
__global__ __launch_bounds__(kPreBlock) void lookup_kernel(IndexView v, const int32_t *sketches, uint32_t nq,
                                                           uint32_t n_qchunk, uint32_t *pre, uint32_t t0) {
  const uint32_t tid = threadIdx.x;
  const uint32_t n_sb = v.f_local / PS;
  const uint32_t x = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
  const uint32_t sb = (k / n_qchunk) * kXcds + x;
  if (sb >= n_sb) return;   // padding block (uniform)
  const uint32_t q = (k % n_qchunk) * kPreBlock + tid;
  if (q >= nq) return;      // no barrier below
  const uint32_t R = v.d.R;
  // this thread's fingerprints (src/niqki_index.cpp:654: anything outside [0, R) has no bucket)
  int32_t fp[PS];
  {
    const int4 *src = (const int4 *)(sketches + (uint64_t)q * v.q_stride + v.q_off + (uint64_t)sb * PS);
#pragma unroll
    for (int u = 0; u < PS / 4; ++u) {
      const int4 a = src[u];
      fp[4 * u] = a.x; fp[4 * u + 1] = a.y; fp[4 * u + 2] = a.z; fp[4 * u + 3] = a.w;
    }
  }
  uint32_t res[NT][PS];
#pragma unroll
  for (int i0 = 0; i0 < PS; i0 += kPreFlight) {
    Entry en[kPreFlight][NT];
#pragma unroll
    for (int j = 0; j < (int)kPreFlight; ++j) {   // all loads of the group first
      const int i = i0 + j;
      const bool ok = fp[i] >= 0 && (uint32_t)fp[i] < R;
      const Entry *e = v.entries + ((uint64_t)(sb * PS + i) * R + (ok ? (uint32_t)fp[i] : 0u)) * v.n_tiles + t0;
#pragma unroll
      for (int t = 0; t < NT; ++t) en[j][t] = e[t];
    }
#pragma unroll
    for (int j = 0; j < (int)kPreFlight; ++j) {
      const int i = i0 + j;
      const bool ok = fp[i] >= 0 && (uint32_t)fp[i] < R;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const uint32_t base = v.slot_units[(uint64_t)(t0 + t) * (v.f_local + 1) + sb * PS + i];
        res[t][i] = ok ? (((en[j][t].start - base) << 16) | en[j][t].len) : 0u;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    uint4 *dst = (uint4 *)(pre + ((uint64_t)q * v.n_tiles + t0 + t) * v.f_local + (uint64_t)sb * PS);
#pragma unroll
    for (int u = 0; u < PS / 4; ++u) dst[u] = make_uint4(res[t][4 * u], res[t][4 * u + 1], res[t][4 * u + 2], res[t][4 * u + 3]);
  }
}

// The same pre-pass with the table rows staged in LDS (1 or 2 tiles, R * tiles * 4 = 16 or 32 KB),
// from the PACKED copy of the table (IndexView::ptab: the 4-byte word of every entry): one workgroup
// = 32 consecutive slots x 1024 queries (one per thread).  Slot by slot the 1024 threads copy the
// slot's whole row (coalesced 16-byte loads straight into LDS: the table is streamed, not hit at
// random), three rows ahead, and every thread picks its query's word(s) with one LDS read.  At the end the 32 words per (query, tile) go through LDS once more so that
// eight neighbouring lanes store one 128-byte line of pre[q][t][s] together.  The workgroups of one
// slot block (nq / 1024) are neighbours in one XCD's dispatch order: the row comes from HBM once.
constexpr uint32_t kRowSlots = 32, kRowBlock = 1024;
// The packed table (IndexView::ptab, one word per entry) keeps the tiles in GROUPS of two: the words of tiles
// t0, t0 + 1 (t0 even) of all slots lie together as [f_local][R][2] from this word on ([f_local][R][1] for a last
// odd tile) -- a launch over one group streams exactly its own rows.  With <= 2 tiles this is the whole table.
__host__ __device__ inline uint64_t packed_table_offset(const IndexView &v, uint32_t t0) { return (uint64_t)v.f_local * v.d.R * t0; }
__device__ __forceinline__ void wave_lds_fence() {   // this wave's LDS traffic so far has completed
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_barrier() {   // a workgroup barrier that waits for LDS traffic only:
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the row loads in flight stay in flight
}
// NT tiles from tile t0 on per launch (a tile GROUP: the packed table keeps the rows of a group together, see
// packed_table_offset): an index of more than 2 tiles takes ceil(n_tiles / 2) launches, each streaming its group's
// part of the table once.
template <int NT, int PER>   // PER: 16-byte pieces of a packed row per thread (R * NT / 4096)
__global__ __launch_bounds__(kRowBlock) void lookup_rows_kernel(IndexView v, const int32_t *sketches, uint32_t nq,
                                                                uint32_t n_qchunk, uint32_t *pre, uint32_t t0) {
  extern __shared__ __align__(16) uint32_t rows[];   // 4 buffers of R * NT words; at the end 1024 x 17 words
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t n_sb = v.f_local / kRowSlots;
  const uint32_t x = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
  const uint32_t sb = (k / n_qchunk) * kXcds + x;
  if (sb >= n_sb) return;   // padding block (uniform)
  const uint32_t q0 = (k % n_qchunk) * kRowBlock, q = q0 + tid;
  const bool live = q < nq;   // the others still help to stage the rows
  const uint32_t R = v.d.R, RW = R * NT;
  int32_t fp[kRowSlots];
  {
    const int4 *src = (const int4 *)(sketches + (uint64_t)(live ? q : 0u) * v.q_stride + v.q_off + (uint64_t)sb * kRowSlots);
#pragma unroll
    for (int u = 0; u < (int)kRowSlots / 4; ++u) {
      const int4 a = src[u];
      fp[4 * u] = a.x; fp[4 * u + 1] = a.y; fp[4 * u + 2] = a.z; fp[4 * u + 3] = a.w;
    }
  }
  // Rows travel memory -> LDS directly (global_load_lds_dwordx4: a wave's 64 lanes fill 1 KB of LDS from
  // the wave-uniform base on, no registers, no ds_write).  The compiler would wait for ALL such loads in
  // flight before any LDS read it can see, so the pipeline's waits and the look-up read are written by
  // hand: loads of one kind retire in issue order (vmcnt), and a wave passes the row's barrier only when
  // its own pieces of that row have landed.
  const uint4 *row0 = (const uint4 *)(v.ptab + packed_table_offset(v, t0) + (uint64_t)sb * kRowSlots * RW) + tid;
  const uint32_t row_u4 = RW / 4;   // 16-byte pieces per row
  const uint32_t wbase = tid & ~63u;
  auto request = [&](int i) {
    uint32_t *dst = rows + (uint32_t)(i & 3) * RW;
#pragma unroll
    for (int j = 0; j < PER; ++j)
      __builtin_amdgcn_global_load_lds((glb_cvoid *)(row0 + (uint64_t)i * row_u4 + (uint32_t)j * kRowBlock),
                                       (lds_void *)(dst + ((uint32_t)j * kRowBlock + wbase) * 4u), 16, 0, 0);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)rows;   // LDS byte address of the row buffers
  uint32_t res[NT][kRowSlots];
  request(0);
  request(1);
  request(2);
  // step I: own pieces of row I landed (only the 2 newer rows' loads may be outstanding) -> barrier: the
  // whole row is in LDS and everybody is done with row I - 1, whose buffer row I + 3 may now overwrite
#define NQ_ROW_STEP(I)                                                                                \
  {                                                                                                   \
    if ((I) + 2 < (int)kRowSlots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");      \
    else if ((I) + 1 < (int)kRowSlots) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");     \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    lds_barrier();                                                                                    \
    if ((I) + 3 < (int)kRowSlots) request((I) + 3);                                                   \
    const bool ok = fp[(I)] >= 0 && (uint32_t)fp[(I)] < R; /* src/niqki_index.cpp:654 */              \
    const uint32_t at = lds0 + ((uint32_t)((I) & 3) * RW + (ok ? (uint32_t)fp[(I)] : 0u) * NT) * 4u;   \
    uint32_t w0, w1 = 0;                                                                              \
    if (NT == 2) {                                                                                    \
      uint2 w;                                                                                        \
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(at) : "memory");      \
      w0 = w.x; w1 = w.y;                                                                             \
    } else {                                                                                          \
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w0) : "v"(at) : "memory");     \
    }                                                                                                 \
    res[0][(I)] = ok ? w0 : 0u;                                                                       \
    res[NT - 1][(I)] = ok ? (NT == 2 ? w1 : w0) : 0u;                                                 \
  }
#define NQ_ROW_STEP8(B) NQ_ROW_STEP((B)) NQ_ROW_STEP((B) + 1) NQ_ROW_STEP((B) + 2) NQ_ROW_STEP((B) + 3) \
                        NQ_ROW_STEP((B) + 4) NQ_ROW_STEP((B) + 5) NQ_ROW_STEP((B) + 6) NQ_ROW_STEP((B) + 7)
  NQ_ROW_STEP8(0) NQ_ROW_STEP8(8) NQ_ROW_STEP8(16) NQ_ROW_STEP8(24)
#undef NQ_ROW_STEP8
#undef NQ_ROW_STEP
  static_assert(kRowSlots == 32, "the steps above are written out");
  // Out: thread q holds 32 words per tile = one 128-byte line of pre[q][t][s]; lanes 8j .. 8j + 7 of a
  // wave store the eight 16-byte pieces of query (wave base + 8 r + j)'s line, r = 0 .. 7, so one store
  // instruction writes 8 whole lines.  The transposition goes through LDS (33-word rows: no bank conflicts).
  lds_barrier();   // all look-ups of the last row are done: the row buffers are free
  uint32_t *mine = rows + tid * (kRowSlots + 1u);
#define NQ_STORE_TILE(T)                                                                                             \
  {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < (int)kRowSlots; ++i) mine[i] = res[(T)][i];                               \
    wave_lds_fence(); /* (one wave reads what its own lanes wrote) */                                                \
    _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                                  \
      const uint32_t src_t = (tid & ~63u) + 8u * (uint32_t)r + (lane >> 3); /* whose line this lane helps to store */ \
      const uint32_t piece = lane & 7u;                                                                              \
      const uint32_t *sp = rows + src_t * (kRowSlots + 1u) + piece * 4u;                                             \
      const uint4 w = make_uint4(sp[0], sp[1], sp[2], sp[3]);                                                        \
      const uint32_t qq = q0 + src_t;                                                                                \
      if (qq < nq) *(uint4 *)(pre + ((uint64_t)qq * v.n_tiles + t0 + (T)) * v.f_local + (uint64_t)sb * kRowSlots + piece * 4u) = w; \
    }                                                                                                                \
    wave_lds_fence(); /* before the next tile overwrites the wave's words */                                         \
  }
  NQ_STORE_TILE(0)
  if (NT == 2) NQ_STORE_TILE(NT - 1)
#undef NQ_STORE_TILE
}
// Packed rows that the LDS buffers hold, dealt to 1024 threads in whole 16-byte loads: tiles are taken two at a
// time (rows of 2 R words = 32 KB at W = 12), a last odd tile alone.
bool lookup_wants_packed(const IndexView &v) {
  if (!launch_lookup_usable(v) || v.f_local % kRowSlots) return false;
  auto row_ok = [&](uint32_t nt) {   // a group's packed row: within 32 KB, whole 16-byte pieces per thread, 1 or 2 of them
    const uint32_t rw = v.d.R * nt;
    if (rw * 4u > 32768u || rw % (4u * kRowBlock)) return false;
    const uint32_t per = rw / (4u * kRowBlock);
    return per == 1 || per == 2;
  };
  if (v.n_tiles >= 2 && !row_ok(2)) return false;
  if ((v.n_tiles & 1u) && !row_ok(1)) return false;
  return true;
}

__global__ __launch_bounds__(256) void pack_entries_kernel(IndexView v, uint32_t *ptab) {
  const uint64_t n = (uint64_t)v.f_local * v.d.R * v.n_tiles, step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    const uint32_t t = (uint32_t)(i % v.n_tiles);
    const uint64_t sf = i / v.n_tiles;   // slot * R + fp
    const uint32_t s = (uint32_t)(sf / v.d.R);
    const Entry e = v.entries[i];
    const uint32_t t0 = t & ~1u, nt = v.n_tiles - t0 >= 2 ? 2u : 1u;   // the tile's group
    ptab[packed_table_offset(v, t0) + sf * nt + (t - t0)] = ((e.start - v.slot_units[(uint64_t)t * (v.f_local + 1) + s]) << 16) | e.len;
  }
}
hipError_t launch_pack_entries(const IndexView &v, uint32_t *ptab, hipStream_t stream) {
  const uint64_t n = (uint64_t)v.f_local * v.d.R * v.n_tiles;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_entries_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 65536)), dim3(256), 0, stream, v, ptab);
  return hipGetLastError();
}

static uint32_t lookup_slots(const IndexView &v) { return v.n_tiles <= 2 ? 32u : 16u; }

// Can the pre-pass serve this index?  Whole slot blocks, and both halves of the packed word
// within 16 bits (bucket lengths <= tile, starts relative to the slot <= tile / unit + R units).
bool launch_lookup_usable(const IndexView &v) {
  if (v.n_tiles < 1 || v.f_local % lookup_slots(v)) return false;
  if (v.tile > 65535u) return false;
  return (uint64_t)(v.tile >> v.align_log2) + v.d.R + 1 <= 65535u;
}

size_t lookup_pre_bytes(const IndexView &v, uint32_t nq) { return (size_t)v.f_local * nq * v.n_tiles * 4; }

hipError_t launch_lookup(const IndexView &v, const int32_t *sketches, uint32_t nq, uint32_t *pre, hipStream_t stream) {
  if (nq == 0 || !launch_lookup_usable(v)) return hipErrorInvalidValue;
  if (v.ptab && lookup_wants_packed(v) && nq >= kRowBlock / 2) {
    const uint32_t n_sb = v.f_local / kRowSlots, n_qchunk = (nq + kRowBlock - 1) / kRowBlock;
    const uint64_t grid = (uint64_t)((n_sb + kXcds - 1) / kXcds * kXcds) * n_qchunk;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
#define NQ_LAUNCH_ROWS(NT, PER)                                                                                        \
  do {                                                                                                                 \
    e = hipFuncSetAttribute((const void *)lookup_rows_kernel<NT, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                     \
    hipLaunchKernelGGL((lookup_rows_kernel<NT, PER>), dim3((uint32_t)grid), dim3(kRowBlock), lds, stream, v, sketches, nq, n_qchunk, pre, t0); \
  } while (0)
    for (uint32_t t0 = 0; t0 < v.n_tiles; t0 += 2) {   // tile groups (packed_table_offset)
      const uint32_t nt = v.n_tiles - t0 >= 2 ? 2u : 1u;
      const size_t lds = std::max<size_t>((size_t)v.d.R * nt * 4 * 4, (size_t)kRowBlock * (kRowSlots + 1) * 4);
      const uint32_t per = v.d.R * nt / (4u * kRowBlock);
      if (nt == 1) { if (per == 1) NQ_LAUNCH_ROWS(1, 1); else NQ_LAUNCH_ROWS(1, 2); }
      else { if (per == 1) NQ_LAUNCH_ROWS(2, 1); else NQ_LAUNCH_ROWS(2, 2); }
    }
#undef NQ_LAUNCH_ROWS
    return hipGetLastError();
  }
  const uint32_t ps = lookup_slots(v);
  const uint32_t n_sb = v.f_local / ps, n_qchunk = (nq + kPreBlock - 1) / kPreBlock;
  const uint64_t grid = (uint64_t)((n_sb + kXcds - 1) / kXcds * kXcds) * n_qchunk;
  if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
#define NQ_LAUNCH_LOOKUP(NT, PS) \
  hipLaunchKernelGGL((lookup_kernel<NT, PS>), dim3((uint32_t)grid), dim3(kPreBlock), 0, stream, v, sketches, nq, n_qchunk, pre, t0)
  for (uint32_t t0 = 0; t0 < v.n_tiles; t0 += 4) {   // (blocks of 16 slots whenever there are more than 2 tiles)
    const uint32_t nt = v.n_tiles - t0 < 4 ? v.n_tiles - t0 : 4;
    if (ps == 32) { if (nt == 1) NQ_LAUNCH_LOOKUP(1, 32); else NQ_LAUNCH_LOOKUP(2, 32); }
    else if (nt == 1) NQ_LAUNCH_LOOKUP(1, 16);
    else if (nt == 2) NQ_LAUNCH_LOOKUP(2, 16);
    else if (nt == 3) NQ_LAUNCH_LOOKUP(3, 16);
    else NQ_LAUNCH_LOOKUP(4, 16);
  }
#undef NQ_LAUNCH_LOOKUP
  return hipGetLastError();
}

// ---- locality order of a query batch -----------------------------------------------------
// Queries that hit the same genomes read the same table and bucket lines.  When they run on
// the same XCD at the same time those lines are fetched from HBM once (measured: 13 % off the
// launch when the 4 queries of a family sit 8 blocks apart).  So a batch is ordered by its
// best probable hit: probe_kernel counts the first kProbeSlots slots only and takes the genome
// with the most hits as the query's key, order_kernel sorts (key, query), and gather_kernel
// maps sorted neighbours to one XCD.  Only the order of the work changes, never a result.
constexpr uint32_t kOrderGroup = 8;
constexpr uint32_t kProbeSlots = 16;
constexpr uint32_t kOrderMax = 4096;   // queries per launch (12 index bits next to a 20-bit key)

// Counters per block of kProbeBlock consecutive genomes (a key only has to bring the queries of one
// family together): a few KB of LDS instead of the gather kernel's whole tile, so many probes share
// a CU and hide each other's three dependent memory round trips (sketch -> entry -> ids).
constexpr uint32_t kProbeBlock = 64;
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe_kernel(IndexView v, const int32_t *sketches, uint32_t *keys) {
  extern __shared__ __align__(16) uint32_t cnt[];
  __shared__ uint32_t s_best;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const int32_t *sk = sketches + (uint64_t)q * v.q_stride + v.q_off;
  const uint32_t n_probe = v.f_local < kProbeSlots ? v.f_local : kProbeSlots;
  const uint32_t n_blocks = (v.n_genomes + kProbeBlock - 1) / kProbeBlock;
  if (tid == 0) s_best = 0;
  for (uint32_t i = tid; i < n_blocks; i += BLOCK) cnt[i] = 0;
  __syncthreads();
  for (uint32_t w = wave; w < n_probe * v.n_tiles; w += BLOCK / 64) {
    const uint32_t s = w / v.n_tiles, t = w % v.n_tiles;
    const int32_t fp = sk[s];
    if (fp < 0 || (uint32_t)fp >= v.d.R) continue;  // wave uniform
    const Entry e = v.entries[((uint64_t)s * v.d.R + (uint32_t)fp) * v.n_tiles + t];
    const uint16_t *b = v.gids + v.tile_base[t] + ((uint64_t)e.start << v.align_log2);
    for (uint32_t o = lane; o < e.len; o += 64) atomicAdd(&cnt[tile_gid(v, t, b[o]) / kProbeBlock], 1u);
  }
  __syncthreads();
  uint32_t best = 0;  // count << 20 | block
  for (uint32_t i = tid; i < n_blocks; i += BLOCK) {
    const uint32_t c = cnt[i] < 4095u ? cnt[i] : 4095u;
    const uint32_t a = (c << 20) | i;
    best = best > a ? best : a;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t y = __shfl_xor(best, o, 64);
    best = best > y ? best : y;
  }
  if (lane == 0) atomicMax(&s_best, best);
  __syncthreads();
  if (tid == 0) {
    const uint32_t c = s_best >> 20;
    const uint32_t key = c >= 2 ? (s_best & 0xFFFFFu) : 0xFFFFFu;  // unrelated queries: one group at the end
    keys[q] = (key << 12) | q;
  }
}

// keys[0..nq) -> order[0..nq): ascending (key, query index); one workgroup, bitonic in LDS
__global__ __launch_bounds__(1024) void order_kernel(const uint32_t *keys, uint32_t nq, uint32_t *order) {
  __shared__ uint32_t a[kOrderMax];
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < kOrderMax; i += 1024) a[i] = i < nq ? keys[i] : 0xFFFFFFFFu;
  __syncthreads();
  for (uint32_t k = 2; k <= kOrderMax; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < kOrderMax; i += 1024) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const uint32_t x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < nq; i += 1024) order[i] = a[i] & 0xFFFu;
}

#ifdef NQ_GATHER_CLOCK   // measurement builds only (tools/gather_clock.py): per-workgroup phase times
__device__ unsigned long long *g_gclk;
extern "C" int nq_debug_gather_clock(unsigned long long *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_gclk), &buf, sizeof(buf)); }
#define NQ_GCLK(k) do { if (threadIdx.x == 0 && g_gclk) g_gclk[(uint64_t)blockIdx.x * 64 + (k)] = wall_clock64(); } while (0)
#define NQ_GCLK_WAVE(k) do { if ((threadIdx.x & 63u) == 0 && g_gclk) g_gclk[(uint64_t)blockIdx.x * 64 + (k) + (threadIdx.x >> 6)] = wall_clock64(); } while (0)
#else
#define NQ_GCLK(k)
#define NQ_GCLK_WAVE(k)
#endif

// One workgroup per query; the genome tiles are walked one after another with
// the tile's hit counters (packed u16 pairs) in LDS.
// Its barriers order LDS traffic only (lds_barrier): threads never read each other's global writes here,
// and the look-ups already on their way for the next tile must not be waited for.
// NT = -1: every tile's entries come from the look-up pre-pass (`stash` then holds its packed words)
template <int BLOCK, int UNROLL, int NT, int MODE = 0, bool PAD = false>
__global__ __launch_bounds__(BLOCK) void gather_kernel(IndexView v, const int32_t *sketches,
                                                       uint16_t *counts, uint16_t *counts2, uint64_t stride, Entry *stash,
                                                       const uint32_t *order, uint32_t nq, CandOut co) {
  extern __shared__ __align__(16) uint32_t cnt[];
  uint32_t q = blockIdx.x;
  if (order) {
    // Locality order (see order_queries below): block b runs on XCD b % 8 as the (b / 8)-th
    // workgroup of that XCD; kOrderGroup consecutive entries of the sorted order share an XCD
    // and sit next to each other in its dispatch queue.
    const uint32_t x = blockIdx.x % kXcds, k = blockIdx.x / kXcds;
    const uint32_t i = ((k / kOrderGroup) * kXcds + x) * kOrderGroup + k % kOrderGroup;
    if (i >= nq) return;  // padding block (uniform)
    q = order[i];
  }
  const uint32_t tid = threadIdx.x;
  NQ_GCLK(0);
#ifdef NQ_GATHER_CLOCK
  if (tid == 0 && g_gclk) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gclk[(uint64_t)blockIdx.x * 64 + 15] = ((uint64_t)xcc << 32) | hw;
  }
#endif
  // (PAD: the padded walk addresses the counters from LDS address 0 -- launch_gather checks that this
  // instantiation has no static LDS in front of its dynamic block and otherwise launches the form without PAD)
  const int32_t *sk = NT < 0 ? nullptr : sketches + (uint64_t)q * v.q_stride + v.q_off;
  Item *queue = (Item *)(cnt + (v.tile + 1) / 2 + (PAD ? kPadWords : 0u));  // behind the counters: kQueue items per wave
  uint32_t sink = 0;
  const bool want_cand = co.cand != nullptr;
  const bool want_surv = co.surv != nullptr;
  const uint32_t emit_thr = want_surv ? co.surv_thr : co.thr;   // surv_thr <= thr
  // List lengths of this query: only this workgroup appends to them during the launch, so they are counted
  // in LDS (same-address global atomics serialise in L2: ~45 entries per tile cost 5 us) and written back once.
  // The two words are wave 0's queue head, idle between two walks (the LDS is full: kPadMaxTile): thread 0
  // parks the running totals there before the barrier that ends a walk and takes them back after the scan.
  uint32_t *lds_n = (uint32_t *)queue;
  uint32_t n_surv = 0, n_cand = 0;
  if (co.hl_over && blockIdx.x == 0 && tid == 0) co.hl_over[0] = 0u;   // the scan that follows counts the overflowing lists here
  if (want_cand && tid == 0) {
    n_cand = (uint32_t)co.n[q];
    if (want_surv) n_surv = (uint32_t)co.surv_n[q];
  }
  auto emit = [&](uint32_t c, uint32_t col) {   // col: column of the row = genome id - g_base
    if (c >= emit_thr) {
      if (want_surv) {
        const uint32_t i = atomicAdd(&lds_n[0], 1u);
        if (i < co.surv_cap) co.surv[(uint64_t)q * co.surv_cap + i] = make_int2((int)(v.g_base + col), (int)c);
      }
      if (c >= co.thr) {
        const uint32_t i = atomicAdd(&lds_n[1], 1u);
        if (i < co.cap) co.cand[(uint64_t)q * co.cap + i] = (int32_t)(v.g_base + col);
      }
    }
  };
  uint32_t t_emit = 0;   // the tile being scanned / written out
  // The eight counters of a quad (words 4k .. 4k+3 of tile t) at once: ONE LDS atomic per list claims the places
  // of all its entries, then the lane stores them.  (Survivors come in runs -- the genomes of a family sit next
  // to each other -- and one atomic per entry made a lane with eight of them wait eight LDS round trips.)
  auto emit_quad = [&](const uint4 &c, uint32_t k, uint32_t n_t) {
    const uint32_t o = c.x | c.y | c.z | c.w;
    if ((o & 0xFFFFu) < emit_thr && (o >> 16) < emit_thr) return;   // no half of the quad can reach the threshold
    const uint32_t cw[4] = {c.x, c.y, c.z, c.w};
    uint32_t ms = 0, mc = 0;   // halves that go to the survivor / candidate list
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = (i & 1) ? cw[i >> 1] >> 16 : cw[i >> 1] & 0xFFFFu;
      const bool in = 8 * k + i < n_t;
      ms |= (uint32_t)(in && h >= emit_thr) << i;
      mc |= (uint32_t)(in && h >= co.thr) << i;
    }
    uint32_t bs = 0, bc = 0;
    if (want_surv && ms) bs = atomicAdd(&lds_n[0], (uint32_t)__builtin_popcount(ms));
    if (mc) bc = atomicAdd(&lds_n[1], (uint32_t)__builtin_popcount(mc));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = (i & 1) ? cw[i >> 1] >> 16 : cw[i >> 1] & 0xFFFFu;
      const uint32_t gid = v.g_base + tile_gid(v, t_emit, 8 * k + i);
      if (want_surv && (ms >> i & 1u)) {
        if (bs < co.surv_cap) co.surv[(uint64_t)q * co.surv_cap + bs] = make_int2((int)gid, (int)h);
        ++bs;
      }
      if (mc >> i & 1u) {
        if (bc < co.cap) co.cand[(uint64_t)q * co.cap + bc] = (int32_t)gid;
        ++bc;
      }
    }
  };
  const uint32_t n_it_all = (v.f_local + 63) / 64, it_pass = kPassSlots / 64;
  for (uint32_t it_lo = 0; it_lo < n_it_all; it_lo += it_pass) {   // one pass unless f_local > 2^15 (S = 16)
  const uint32_t n_it = it_lo + it_pass < n_it_all ? it_lo + it_pass : n_it_all;
  uint16_t *plane = it_lo ? counts2 : counts;
  Entry ahead[2];
  if constexpr (NT < 0) {   // the first tile's first look-ups (walk_tile keeps `ahead` one tile ahead from here on)
    const uint32_t *p0 = (const uint32_t *)stash + (uint64_t)q * v.n_tiles * v.f_local;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      uint32_t s = (it_lo + (tid >> 6) + k * (BLOCK / 64)) * 64 + (tid & 63u);
      if (s >= v.f_local) s = v.f_local - 1;
      const uint32_t w = p0[s];
      ahead[k] = Entry{v.slot_units[s] + (w >> 16), w & 0xFFFFu};
    }
  }
  bool zeroed = false;   // the scan of the tile before left this tile's counters at zero
  for (uint32_t t = 0; t < v.n_tiles; ++t) {
    const uint32_t n_t = tile_count(v, t);
    const uint32_t n_words = (n_t + 1) / 2;
    t_emit = t;
    if (!zeroed) {
      uint4 *c4 = (uint4 *)cnt;   // (the dynamic LDS block is 16-byte aligned)
      for (uint32_t i = tid; i < n_words / 4; i += BLOCK) c4[i] = make_uint4(0, 0, 0, 0);
      for (uint32_t i = (n_words & ~3u) + tid; i < n_words; i += BLOCK) cnt[i] = 0;
      if (co.hl)
        for (uint32_t i = tid; i < co.hl_cap; i += BLOCK) ((uint32_t *)(queue + (BLOCK / 64) * kQueue))[i] = 0u;
      lds_barrier();
    }
    zeroed = false;
    NQ_GCLK(1 + 4 * (t & 1));
    if constexpr (MODE == 8) {
    } else if constexpr (NT < 0) {
      walk_tile<BLOCK, UNROLL, 1, false, false, MODE, true, PAD>(v, sk, q, t, cnt, queue, nullptr, sink, it_lo, n_it, (const uint32_t *)stash, ahead);
    } else if constexpr (NT >= 2) {
      if (t == 0) walk_tile<BLOCK, UNROLL, NT, true, false, MODE, false, PAD>(v, sk, q, t, cnt, queue, stash, sink, it_lo, n_it);
      else walk_tile<BLOCK, UNROLL, NT, false, true, MODE, false, PAD>(v, sk, q, t, cnt, queue, stash, sink, it_lo, n_it);
    } else {
      walk_tile<BLOCK, UNROLL, 1, false, false, MODE, false, PAD>(v, sk, q, t, cnt, queue, stash, sink, it_lo, n_it);
    }
    if (MODE != 0) cnt[tid % n_words] ^= sink;
    if (want_cand && tid == 0) { lds_n[0] = n_surv; lds_n[1] = n_cand; }
    if (co.hl && tid == 0) lds_n[0] = 0;
    NQ_GCLK_WAVE(16 + 16 * (t & 1));
    NQ_GCLK(2 + 4 * (t & 1));
    lds_barrier();
    NQ_GCLK(3 + 4 * (t & 1));
    if (MODE == 7) continue;
    if (plane == nullptr) {
      // no counter row (survivor output only): the tile's counters are scanned where they are
      // ... and left at zero for the next tile (whose counters are no more than this one's)
      zeroed = t + 1 < v.n_tiles && tile_count(v, t + 1) <= n_t;
      // 16 bytes per LDS access, four accesses in flight per thread; a quad whose OR-ed halves stay under
      // the threshold holds no survivor
      uint4 *c4 = (uint4 *)cnt;
      const uint32_t n_quads = n_words / 4;
      auto word = [&](uint32_t c, uint32_t w) {
        if ((c & 0xFFFFu) >= emit_thr) emit(c & 0xFFFFu, tile_gid(v, t, 2 * w));
        if (2 * w + 1 < n_t && (c >> 16) >= emit_thr) emit(c >> 16, tile_gid(v, t, 2 * w + 1));
      };
      for (uint32_t k0 = tid; k0 < n_quads; k0 += 4 * BLOCK) {
        uint4 c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t k = k0 + j * BLOCK;
          c[j] = k < n_quads ? c4[k] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t k = k0 + j * BLOCK;
          if (zeroed && k < n_quads) c4[k] = make_uint4(0, 0, 0, 0);
          emit_quad(c[j], k, n_t);
        }
      }
      for (uint32_t w = 4 * n_quads + tid; w < n_words; w += BLOCK) {
        const uint32_t c = cnt[w];
        if (zeroed) cnt[w] = 0;
        word(c, w);
      }
      lds_barrier();
      if (tid == 0) { n_surv = lds_n[0]; n_cand = lds_n[1]; }
      NQ_GCLK(4 + 4 * (t & 1));
      continue;
    }
    if (co.hl) {
      // Hit lists (launch_gather: one tile, one plane, nothing to add to): the query's hits are picked while its
      // counters are in LDS -- count >= min_score, src/niqki_index.cpp:662-666 -- and ordered here, by rank: the keys
      // count << 32 | gid are distinct, so an entry's place under greater<pair<count, gid>> (:685) is the number of
      // entries with a larger key.  A short read against a genome index has a few dozen hits: no 2N-byte counter row
      // is written for it, none read again.  A query with more than hl_cap hits (min_score 0: every genome) leaves
      // through its counter row as before; launch_hitlist_emit orders it.
      uint32_t *list = (uint32_t *)(queue + (BLOCK / 64) * kQueue);   // hl_cap packed entries: count << 16 | gid
      const uint32_t cap = co.hl_cap, ms = co.hl_min;
      // (hits are few: an LDS atomic each; ballot ranking was slower.  The counters are read eight at a time -- the
      // workgroup's clock showed 2.7 of a read's 14.6 us in this scan when it took them word by word -- and a quad
      // whose OR-ed halves stay under the threshold holds no hit)
      auto pick = [&](uint32_t c, uint32_t i) {
        const uint32_t lo = c & 0xFFFFu, hi = c >> 16;
        if (lo >= ms) {
          const uint32_t at = atomicAdd(&lds_n[0], 1u);
          if (at < cap) list[at] = (lo << 16) | (v.g_base + 2 * i);
        }
        if (2 * i + 1 < n_t && hi >= ms) {
          const uint32_t at = atomicAdd(&lds_n[0], 1u);
          if (at < cap) list[at] = (hi << 16) | (v.g_base + 2 * i + 1);
        }
      };
      const uint32_t n_quads = n_words / 4;
      const uint4 *c4 = (const uint4 *)cnt;   // (the dynamic LDS block is 16-byte aligned)
      for (uint32_t k = tid; k < n_quads; k += BLOCK) {
        const uint4 c = c4[k];
        const uint32_t o = c.x | c.y | c.z | c.w;
        if ((o & 0xFFFFu) < ms && (o >> 16) < ms) continue;
        pick(c.x, 4 * k); pick(c.y, 4 * k + 1); pick(c.z, 4 * k + 2); pick(c.w, 4 * k + 3);
      }
      for (uint32_t i = 4 * n_quads + tid; i < n_words; i += BLOCK) pick(cnt[i], i);
      lds_barrier();
      NQ_GCLK(4);
      const uint32_t total = lds_n[0];
      if (total <= cap) {
        // (entries are read four at a time; the list was zeroed with the counters, and key 0 -- the places behind
        // the last entry -- is larger than no key: a real key 0 is count 0 of genome 0, the smallest there is)
        const uint32_t t4 = (total + 3u) & ~3u;
        uint32_t *out = co.hl + (uint64_t)q * cap;
        for (uint32_t i = tid; i < total; i += BLOCK) {
          const uint32_t key = list[i];
          uint32_t rank = 0;
          for (uint32_t j = 0; j < t4; j += 4) {   // (all lanes read the same four entries: a broadcast)
            const uint4 o = *(const uint4 *)(list + j);
            rank += (o.x > key) + (o.y > key) + (o.z > key) + (o.w > key);
          }
          out[rank] = key;
        }
      } else {
        uint16_t *row = plane + (uint64_t)q * stride + v.g_base;
        if (((uintptr_t)row & 3u) == 0) {
          uint32_t *out = (uint32_t *)row;
          for (uint32_t i = tid; i < n_t / 2; i += BLOCK) out[i] = cnt[i];
          if ((n_t & 1u) && tid == 0) row[n_t - 1] = (uint16_t)(cnt[n_t / 2] & 0xFFFFu);
        } else {
          for (uint32_t i = tid; i < n_t; i += BLOCK) row[i] = (uint16_t)(cnt[i >> 1] >> ((i & 1u) * 16u));
        }
      }
      if (tid == 0) co.hl_n[q] = total;
      NQ_GCLK(5);
      continue;   // (one tile: nothing of this workgroup follows that touches LDS)
    }
    uint16_t *row = plane + (uint64_t)q * stride + v.g_base;
    if (v.stripe > 1 && v.n_tiles > 1 && ((uintptr_t)row & 3u) == 0) {
      // tiles striped in blocks of an even number of genomes: local ids 2w, 2w + 1 are neighbours in
      // the row, a wave writes whole blocks (64-byte blocks are what HBM takes without a
      // read-modify-write, tools/ubench_partial_write.hip)
      // blocks of >= 8 genomes on 16-byte aligned rows: 8 counters (4 words) per lane and store
      const uint32_t quads = (v.stripe % 8u == 0 && ((uintptr_t)row & 15u) == 0 && !v.accumulate) ? n_t / 8 : 0u;
      for (uint32_t k = tid; k < quads; k += BLOCK) {
        const uint32_t col = tile_gid(v, t, 8 * k);
        const uint4 c = *(const uint4 *)(cnt + 4 * k);
        *(uint4 *)(row + col) = c;
        if (want_cand) emit_quad(c, k, n_t);
      }
      for (uint32_t w = 4 * quads + tid; w < n_words; w += BLOCK) {
        const uint32_t col = tile_gid(v, t, 2 * w), c = cnt[w];
        uint32_t *dst = (uint32_t *)(row + col);
        if (2 * w + 1 < n_t) *dst = v.accumulate ? *dst + c : c;
        else { uint16_t *d = (uint16_t *)dst; *d = v.accumulate ? (uint16_t)(*d + c) : (uint16_t)c; }
        if (want_cand) {
          emit(c & 0xFFFFu, col);
          if (2 * w + 1 < n_t) emit(c >> 16, col + 1);
        }
      }
    } else if (v.stripe && v.n_tiles > 1) {
      // striped tiles: the tile's i-th counter belongs to genome tile_gid(t, i)
      for (uint32_t i = tid; i < n_t; i += BLOCK) {
        const uint16_t c = (uint16_t)(cnt[i >> 1] >> ((i & 1u) * 16u));
        uint16_t *dst = row + tile_gid(v, t, i);
        *dst = v.accumulate ? (uint16_t)(*dst + c) : c;
        if (want_cand) emit(c, tile_gid(v, t, i));
      }
    } else {
      // dense counter row of this tile: u16 counts[q*stride + g0 + i], written as the packed words
      const uint32_t g0 = t * v.tile;
      // (packed words need an even first column: odd g_base or g0 -> element-wise)
      const bool packed = ((v.g_base + g0) & 1u) == 0;
      uint32_t *out = (uint32_t *)(row + g0);
      const uint32_t full = packed ? n_t / 2 : 0;
      if (!packed)
        for (uint32_t i = tid; i < n_t; i += BLOCK) {
          const uint16_t c = (uint16_t)(cnt[i >> 1] >> ((i & 1u) * 16u));
          row[g0 + i] = v.accumulate ? (uint16_t)(row[g0 + i] + c) : c;
          if (want_cand) emit(c, g0 + i);
        }
      // (accumulating: sums stay <= F <= 2^15 per half, so the packed add cannot carry)
      for (uint32_t i = tid; i < full; i += BLOCK) {
        const uint32_t c = cnt[i];
        out[i] = v.accumulate ? out[i] + c : c;
        if (want_cand) { emit(c & 0xFFFFu, g0 + 2 * i); emit(c >> 16, g0 + 2 * i + 1); }
      }
      if (packed && (n_t & 1u) && tid == 0) {
        const uint16_t c = (uint16_t)(cnt[full] & 0xFFFFu);
        row[g0 + n_t - 1] = v.accumulate ? (uint16_t)(row[g0 + n_t - 1] + c) : c;
        if (want_cand) emit(c, g0 + n_t - 1);
      }
    }
    lds_barrier();
    if (want_cand && tid == 0) { n_surv = lds_n[0]; n_cand = lds_n[1]; }
    NQ_GCLK(4 + 4 * (t & 1));
  }
  }
  if (want_cand && tid == 0) {
    co.n[q] = (int32_t)n_cand;
    if (want_surv) co.surv_n[q] = (int32_t)n_surv;
  }
}

hipError_t launch_order(const IndexView &v, const int32_t *sketches, uint32_t nq, uint32_t *keys,
                        uint32_t *order, hipStream_t stream) {
  if (nq == 0 || nq > kOrderMax || v.n_genomes >= (1u << 20) - 1) return hipErrorInvalidValue;
  const size_t lds = (size_t)((v.n_genomes + kProbeBlock - 1) / kProbeBlock) * 4;   // <= 64 KB (n_genomes < 2^20)
  hipError_t e = hipFuncSetAttribute((const void *)probe_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(probe_kernel<256>, dim3(nq), dim3(256), lds, stream, v, sketches, keys);
  hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, stream, keys, nq, order);
  return hipGetLastError();
}

bool gather_variant_valid(int variant) {
#ifdef NQ_ABLATION
  if (variant == 11 || variant == 12 || variant == 16 || variant == 17 || variant == 18 || (variant >= 19 && variant <= 24)) return true;
#endif
  return variant >= 0 && variant <= 5;
}

hipError_t launch_gather(const IndexView &v, const int32_t *sketches, uint32_t nq, uint16_t *counts, uint16_t *counts2,
                         uint64_t stride, Entry *stash, const uint32_t *order, int variant, bool pre,
                         hipStream_t stream, const CandOut &co) {
  if (nq == 0 || v.n_tiles == 0) return hipSuccess;
  if (v.f_local > kPassSlots && (!counts2 || v.accumulate)) return hipErrorInvalidValue;
  if (co.cand && (v.accumulate || v.f_local > kPassSlots || !co.n)) return hipErrorInvalidValue;
  if (co.surv && (!co.cand || !co.surv_n || co.surv_thr > co.thr || !co.surv_cap)) return hipErrorInvalidValue;
  if (!counts && !co.surv) return hipErrorInvalidValue;
  if (co.hl && (co.cand || !co.hl_n || !counts || !co.hl_cap || (co.hl_cap & 3u) || co.hl_cap > kHitListMaxCap || v.g_base + v.n_genomes > 65536u || v.n_tiles != 1 || v.accumulate ||
                v.f_local > kPassSlots || v.tile > kHitListMaxTile))
    return hipErrorInvalidValue;
#define NQ_GATHER_LDS(B) ((size_t)((v.tile + 1) / 2 + (v.padded ? kPadWords : 0u)) * 4 + (size_t)(B / 64) * kQueue * sizeof(Item) + (size_t)co.hl_cap * (co.hl ? 4 : 0))
  // with a locality order the grid is padded to whole groups on every XCD
  const uint32_t per_round = kXcds * kOrderGroup;
  dim3 grid(order ? (nq + per_round - 1) / per_round * per_round : nq);
  hipError_t e;
#define NQ_LAUNCH_GATHER(B, U, NT, ...)                                                          \
  do {                                                                                           \
    auto k = gather_kernel<B, U, NT, ##__VA_ARGS__>;                                             \
    const size_t lds = NQ_GATHER_LDS(B);                                                         \
    e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                               \
    hipLaunchKernelGGL(k, grid, dim3(B), lds, stream, v, sketches, counts, counts2, stride, stash, order, nq, co); \
  } while (0)
  // The padded walk (PAD = true) takes a counter's LDS address from the genome id alone: the kernel's dynamic
  // LDS must start at LDS address 0, i.e. the instantiation must have no static LDS in front of it.  Asked of
  // the runtime once per process; if a toolchain ever puts something there, the form with explicit lengths
  // (any LDS base, any index layout) is launched instead.
  static const bool pad_at_zero = [] {
    hipFuncAttributes fa{};
    return hipFuncGetAttributes(&fa, (const void *)gather_kernel<1024, 32, -1, 0, true>) == hipSuccess && fa.sharedSizeBytes == 0 &&
           hipFuncGetAttributes(&fa, (const void *)gather_kernel<1024, 16, 2, 0, true>) == hipSuccess && fa.sharedSizeBytes == 0;
  }();
  const bool padded = v.padded && pad_at_zero;
#define NQ_BY_TILES(B, U, ...)                                                                   \
  do {                                                                                           \
    if (pre) NQ_LAUNCH_GATHER(B, U, -1, ##__VA_ARGS__);                                           \
    else if (v.n_tiles == 2) NQ_LAUNCH_GATHER(B, U, 2, ##__VA_ARGS__);                           \
    else if (v.n_tiles == 3) NQ_LAUNCH_GATHER(B, U, 3, ##__VA_ARGS__);                           \
    else if (v.n_tiles == 4) NQ_LAUNCH_GATHER(B, U, 4, ##__VA_ARGS__);                           \
    else NQ_LAUNCH_GATHER(B, U, 1, ##__VA_ARGS__);                                               \
  } while (0)
  switch (variant) {
    case 1: if (padded) NQ_BY_TILES(1024, 8, 0, true); else NQ_BY_TILES(1024, 8); break;
    case 2: if (padded) NQ_BY_TILES(1024, 32, 0, true); else NQ_BY_TILES(1024, 32); break;
    case 3: NQ_BY_TILES(512, 16); break;
    case 4: NQ_BY_TILES(256, 16); break;
    case 5: NQ_BY_TILES(128, 16); break;
#ifdef NQ_ABLATION  // measurement builds only (make ABLATION=1): these variants return wrong counters
    case 11: NQ_BY_TILES(1024, 16, 1); break;
    case 12: NQ_BY_TILES(1024, 32, 1, true); break;
    case 16: NQ_BY_TILES(1024, 16, 6); break;
    case 17: NQ_BY_TILES(1024, 16, 7, true); break;
    case 18: NQ_BY_TILES(1024, 16, 8, true); break;
    case 19: NQ_BY_TILES(1024, 32, 9, true); break;   // the one-tile walk's per-id cost on the same lines (PairWalk::apply_model)
    case 20: NQ_BY_TILES(1024, 32, 10, true); break;  // ... its returning byte add + wrap test alone
    case 21: NQ_BY_TILES(1024, 32, 11, true); break;  // ... its half-select alone (plain ds_add on the byte)
    case 22: NQ_BY_TILES(1024, 32, 12, true); break;  // the shipped walk without its LDS atomics
    case 23: NQ_BY_TILES(1024, 32, 13, true); break;  // ... with one of the two atomics per dword
    case 24: NQ_BY_TILES(1024, 32, 14, true); break;  // ... with the chunks' positions by v_readlane (no ds_read per load); counters exact
#endif
    default:
      // small tiles (short-read indexes): counters of <= 24 KB leave room for several
      // workgroups per CU, and 4 waves per query then beat 16 (tools/bench_reads.py)
      if (v.tile <= 12288) NQ_BY_TILES(256, 16);
      else if (padded && pre) NQ_BY_TILES(1024, 32, 0, true);   // without look-ups of its own the walk gains from 32 lines per round trip (8.42 against 8.6 ms)
      else if (padded) NQ_BY_TILES(1024, 16, 0, true);   // padded index: mask-free bucket walk
      else NQ_BY_TILES(1024, 16);
      break;
  }
#undef NQ_BY_TILES
#undef NQ_GATHER_LDS
#undef NQ_LAUNCH_GATHER
  return hipGetLastError();
}

// Sum of touched bucket lengths per query (T of the roofline formula).
__global__ __launch_bounds__(256) void gathered_kernel(IndexView v, const int32_t *sketches,
                                                      unsigned long long *per_query) {
  const uint32_t q = blockIdx.x;
  const uint32_t R = v.d.R;
  const int32_t *sk = sketches + (uint64_t)q * v.q_stride + v.q_off;
  unsigned long long sum = 0;
  for (uint32_t s = threadIdx.x; s < v.f_local; s += blockDim.x) {
    int32_t fp = sk[s];
    if (fp >= 0 && (uint32_t)fp < R) {
      const Entry *p = v.entries + ((uint64_t)s * R + (uint32_t)fp) * v.n_tiles;
      for (uint32_t t = 0; t < v.n_tiles; ++t) sum += p[t].len;
    }
  }
  atomicAdd(&per_query[q], sum);
}

hipError_t launch_gathered(const IndexView &v, const int32_t *sketches, uint32_t nq,
                           unsigned long long *per_query, hipStream_t stream) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(gathered_kernel, dim3(nq), dim3(256), 0, stream, v, sketches, per_query);
  return hipGetLastError();
}

// ---- hits: threshold, compaction in descending gid order, stable sort on count ----

// blk_counts[q][b] = number of genomes of block b with count >= min_score, and the query's total
// in hit_off[q] (scanned into offsets by hits_scan_kernel).  One workgroup per query: wave w takes
// blocks w, w+16, ...; a lane reads 8 counters (16 bytes) at a time.
// WIDE: the 16 waves of a workgroup share one query (large indexes).  !WIDE (fewer than 8 blocks,
// i.e. < 32 768 genomes: the short-read indexes): a wave per query, 16 queries per workgroup.
template <bool WIDE>
__global__ __launch_bounds__(1024) void hits_count_kernel(HitsArgs a) {
  __shared__ uint32_t s_sum;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t q = WIDE ? blockIdx.x : blockIdx.x * 16u + wave;
  if (WIDE) {
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
  } else if (q >= a.nq) {
    return;   // (no barrier on this path)
  }
  const uint16_t *row = a.counts + (uint64_t)q * a.stride + a.gid_begin;
  const uint16_t *row2 = a.counts2 ? a.counts2 + (uint64_t)q * a.stride + a.gid_begin : nullptr;
  const bool vec = (((uintptr_t)row) & 15) == 0 && !row2;   // uniform per wave
  uint32_t mine = 0;
  for (uint32_t b = WIDE ? wave : 0u; b < a.n_blk; b += WIDE ? 16u : 1u) {
    const uint32_t lo = b * kHitsBlk;
    const uint32_t hi = (lo + kHitsBlk < a.n_gids) ? lo + kHitsBlk : a.n_gids;
    uint32_t c = 0;
    if (vec) {
      for (uint32_t i = lo + lane * 8; i < hi; i += 512) {
        if (i + 8 <= hi) {
          const uint4 w = *(const uint4 *)(row + i);
          const uint32_t x[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) c += ((x[k] & 0xFFFFu) >= a.min_score) + ((x[k] >> 16) >= a.min_score);
        } else {
          for (uint32_t j = i; j < hi; ++j) c += (row[j] >= a.min_score);
        }
      }
    } else {
      for (uint32_t i = lo + lane; i < hi; i += 64) c += ((uint32_t)row[i] + (row2 ? (uint32_t)row2[i] : 0u) >= a.min_score);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane == 0) { a.blk_counts[(uint64_t)q * a.n_blk + b] = c; mine += c; }
  }
  if (!WIDE) {
    if (lane == 0) a.hit_off[q] = mine;
    return;
  }
  if (lane == 0 && mine) atomicAdd(&s_sum, mine);
  __syncthreads();
  if (threadIdx.x == 0) a.hit_off[q] = s_sum;
}

// hit_off[0..nq): per-query totals -> exclusive prefix, hit_off[nq] = grand total.  One workgroup walks
// the totals 4096 at a time (four consecutive queries per thread, coalesced 32-byte pieces): thread sums, wave
// scan, the 16 wave totals through LDS, a running base.
__global__ __launch_bounds__(1024) void hits_scan_kernel(HitsArgs a) {
  __shared__ unsigned long long wave_tot[2][16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  unsigned long long base = 0;
  uint32_t flip = 0;
  for (uint32_t q0 = 0; q0 < a.nq; q0 += 4096, flip ^= 1u) {
    const uint32_t q = q0 + 4 * tid;
    unsigned long long x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = q + j < a.nq ? a.hit_off[q + j] : 0ull;
    const unsigned long long mine = x[0] + x[1] + x[2] + x[3];
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned long long y = __shfl_up(incl, o, 64);
      if (lane >= (uint32_t)o) incl += y;
    }
    if (lane == 63) wave_tot[flip][wave] = incl;
    __syncthreads();   // (the other half of wave_tot is what the previous round may still be reading)
    unsigned long long before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; ++w) {
      const unsigned long long t = wave_tot[flip][w];
      if (w < wave) before += t;
      total += t;
    }
    unsigned long long run = base + before + incl - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (q + j < a.nq) a.hit_off[q + j] = run;
      run += x[j];
    }
    base += total;
  }
  if (tid == 0) a.hit_off[a.nq] = base;
}

// Each (query, block) writes its hits at the mirrored position so that a
// query's segment ends up in DESCENDING gid order; the global block prefix is
// recomputed from hit_off and the per-block counts.
__global__ __launch_bounds__(256) void hits_compact_kernel(HitsArgs a) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_before;
  const uint32_t q = blockIdx.x / a.n_blk, b = blockIdx.x % a.n_blk;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (a.blk_counts[blockIdx.x] == 0) return;  // nothing above the threshold in this block (uniform)
  if (tid == 0) {
    uint32_t before = 0;
    for (uint32_t i = 0; i < b; ++i) before += a.blk_counts[(uint64_t)q * a.n_blk + i];
    s_before = before;
  }
  const unsigned long long seg0 = a.hit_off[q], seg1 = a.hit_off[q + 1];
  const uint16_t *row = a.counts + (uint64_t)q * a.stride + a.gid_begin;
  const uint16_t *row2 = a.counts2 ? a.counts2 + (uint64_t)q * a.stride + a.gid_begin : nullptr;
  const uint32_t lo = b * kHitsBlk;
  const uint32_t hi = (lo + kHitsBlk < a.n_gids) ? lo + kHitsBlk : a.n_gids;
  __syncthreads();
  uint32_t run = s_before;  // hits of this query with smaller gid, so far
  for (uint32_t base = lo; base < hi; base += 256) {
    uint32_t i = base + tid;
    uint32_t c = (i < hi) ? (uint32_t)row[i] + (row2 ? (uint32_t)row2[i] : 0u) : 0u;
    bool hit = (i < hi) && c >= a.min_score;
    uint64_t bal = __ballot(hit);
    uint32_t rank = __popcll(bal & ((1ULL << lane) - 1ULL));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    uint32_t pre = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) { uint32_t x = wsum[w]; if (w < wave) pre += x; tot += x; }
    if (hit) {
      unsigned long long asc = run + pre + rank;            // rank in ascending gid order
      unsigned long long pos = seg1 - 1 - asc;              // mirrored: descending gid
      if (pos >= seg0 && pos < a.capacity) {
        a.hit_counts[pos] = c;
        a.hit_gids[pos] = a.gid_begin + i;
      }
    }
    run += tot;
    __syncthreads();
  }
}

// One wave per query: stable LSD radix sort (2 x 8 bits) of the segment on the
// count, descending, in place in hit_* (descending gid on entry) through tmp_*.  Equal counts
// keep descending gid: greater<pair<count,gid>>, src/niqki_index.cpp:685.
__device__ void radix_pass_desc(const uint32_t *in_c, const uint32_t *in_g, uint32_t *out_c,
                                uint32_t *out_g, unsigned long long n, uint32_t shift,
                                uint32_t *cur, uint32_t lane) {
  for (uint32_t i = lane; i < 256; i += 64) cur[i] = 0;
  for (unsigned long long i = lane; i < n; i += 64) atomicAdd(&cur[(in_c[i] >> shift) & 0xFFu], 1u);
  // exclusive scan from digit 255 downwards
  uint32_t running = 0;
  for (int c = 192; c >= 0; c -= 64) {
    uint32_t d = (uint32_t)c + 63u - lane;  // lane 0 holds the largest digit of the chunk
    uint32_t x = cur[d];
    uint32_t incl = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t y = __shfl_up(incl, o, 64);
      if (lane >= (uint32_t)o) incl += y;
    }
    cur[d] = running + incl - x;
    running += __shfl(incl, 63, 64);
  }
  const uint64_t lt_mask = (1ULL << lane) - 1ULL;
  for (unsigned long long base = 0; base < n; base += 64) {
    unsigned long long i = base + lane;
    bool valid = i < n;
    uint32_t c = valid ? in_c[i] : 0u, g = valid ? in_g[i] : 0u;
    uint32_t dgt = (c >> shift) & 0xFFu;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
      bool bit = (dgt >> b) & 1u;
      uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    if (valid) {
      uint32_t rank = __popcll(peers & lt_mask), cntp = __popcll(peers);
      uint32_t p = cur[dgt];
      out_c[p + rank] = c;
      out_g[p + rank] = g;
      if (rank == cntp - 1) cur[dgt] = p + cntp;
    }
  }
}

__global__ __launch_bounds__(256) void hits_sort_kernel(HitsArgs a) {
  __shared__ uint32_t curs[4][256];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x * 4 + wave;
  if (q >= a.nq) return;  // wave-private below
  unsigned long long seg0 = a.hit_off[q], seg1 = a.hit_off[q + 1];
  if (seg1 > a.capacity) seg1 = a.capacity;
  if (seg0 >= seg1) return;
  const unsigned long long n = seg1 - seg0;
  uint32_t *tc = a.tmp_counts + seg0, *tg = a.tmp_gids + seg0;
  uint32_t *hc = a.hit_counts + seg0, *hg = a.hit_gids + seg0;
  radix_pass_desc(hc, hg, tc, tg, n, 0, curs[wave], lane);
  __threadfence_block();
  radix_pass_desc(tc, tg, hc, hg, n, 8, curs[wave], lane);
  if (a.counts2) {  // S = 16: a count can be 2^16, 17 bits (two more passes bring the result back into hit_*)
    __threadfence_block();
    radix_pass_desc(hc, hg, tc, tg, n, 16, curs[wave], lane);
    __threadfence_block();
    radix_pass_desc(tc, tg, hc, hg, n, 24, curs[wave], lane);
  }
}

// Candidate genomes of every query: ids with counts[q][g] >= thr, at most `cap`
// per query (unordered), cand[q*cap + i], n[q] = how many there are (may exceed
// cap: the caller must then fall back to the dense exchange).  Used by the
// multi-GPU path: a genome whose summed count reaches min_score has a partial
// count >= ceil(min_score / shards) on at least one shard.
__global__ __launch_bounds__(256) void candidates_kernel(const uint16_t *counts, uint64_t stride, uint32_t n_gids,
                                                        uint32_t thr, uint32_t cap, int32_t *cand, int32_t *n) {
  __shared__ uint32_t s_n;
  const uint32_t q = blockIdx.x;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const uint16_t *row = counts + (uint64_t)q * stride;
  for (uint32_t g = threadIdx.x; g < n_gids; g += 256) {
    if (row[g] >= thr) {
      const uint32_t i = atomicAdd(&s_n, 1u);
      if (i < cap) cand[(uint64_t)q * cap + i] = (int32_t)g;
    }
  }
  __syncthreads();
  const uint32_t tot = s_n;
  for (uint32_t i = tot + threadIdx.x; i < cap; i += 256) cand[(uint64_t)q * cap + i] = -1;
  if (threadIdx.x == 0) n[q] = (int32_t)tot;
}

hipError_t launch_candidates(const uint16_t *counts, uint64_t stride, uint32_t nq, uint32_t n_gids, uint32_t thr,
                             uint32_t cap, int32_t *cand, int32_t *n, hipStream_t stream) {
  if (nq == 0) return hipSuccess;
  hipLaunchKernelGGL(candidates_kernel, dim3(nq), dim3(256), 0, stream, counts, stride, n_gids, thr, cap, cand, n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void plane_add16_kernel(uint16_t *a, const uint16_t *b, uint64_t n) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) a[i] = (uint16_t)(a[i] + b[i]);
}
__global__ __launch_bounds__(256) void plane_sum32_kernel(const uint16_t *a, const uint16_t *b, uint32_t *out, uint64_t n) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) out[i] = (uint32_t)a[i] + (b ? (uint32_t)b[i] : 0u);
}
hipError_t launch_plane_add16(uint16_t *a, const uint16_t *b, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(plane_add16_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 16384)), dim3(256), 0, stream, a, b, n);
  return hipGetLastError();
}
hipError_t launch_plane_sum32(const uint16_t *a, const uint16_t *b, uint32_t *out, uint64_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(plane_sum32_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 16384)), dim3(256), 0, stream, a, b, out, n);
  return hipGetLastError();
}

// After a gather launch with hit lists (CandOut::hl).
// hitlist_scan_kernel: hit_off[0..nq] = exclusive prefix of n[0..nq), 4096 queries per workgroup -- a workgroup first
// adds up what lies before its block (at most a few hundred KB of u32, from L2), then scans its own 4096 -- and the
// queries whose lists overflowed (n > hl_cap) are collected in over[1 ..], over[0] = how many (preset to 0).
__global__ __launch_bounds__(1024) void hitlist_scan_kernel(const uint32_t *n, uint32_t nq, uint32_t hl_cap, unsigned long long *hit_off,
                                                            uint32_t *over) {
  __shared__ unsigned long long wave_tot[16];
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_over, s_over_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t q0 = blockIdx.x * 4096u;
  if (tid == 0) s_over = 0;
  unsigned long long part = 0;
  {   // (q0 is a multiple of 4096: whole 16-byte pieces, four independent loads in flight per thread)
    const uint4 *n4 = (const uint4 *)n;
    const uint32_t m = q0 / 4;
    uint32_t i = tid;
    for (; i + 3 * 1024 < m; i += 4 * 1024) {
      const uint4 a0 = n4[i], a1 = n4[i + 1024], a2 = n4[i + 2048], a3 = n4[i + 3072];
      part += (unsigned long long)a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w;
      part += (unsigned long long)a2.x + a2.y + a2.z + a2.w + a3.x + a3.y + a3.z + a3.w;
    }
    for (; i < m; i += 1024) {
      const uint4 a0 = n4[i];
      part += (unsigned long long)a0.x + a0.y + a0.z + a0.w;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  if (lane == 0) wave_tot[wave] = part;
  __syncthreads();
  if (tid == 0) {
    unsigned long long b = 0;
    for (uint32_t w = 0; w < 16; ++w) b += wave_tot[w];
    s_base = b;
  }
  __syncthreads();
  const unsigned long long base = s_base;
  const uint32_t q = q0 + 4 * tid;
  unsigned long long x[4];
  uint32_t n_over = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x[j] = q + j < nq ? n[q + j] : 0ull;
    n_over += x[j] > hl_cap ? 1u : 0u;
  }
  uint32_t my_over = n_over ? atomicAdd(&s_over, n_over) : 0u;
  const unsigned long long mine = x[0] + x[1] + x[2] + x[3];
  unsigned long long incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long y = __shfl_up(incl, o, 64);
    if (lane >= (uint32_t)o) incl += y;
  }
  __syncthreads();   // (wave_tot is reused; s_over is complete)
  if (lane == 63) wave_tot[wave] = incl;
  if (tid == 0 && s_over) s_over_base = atomicAdd(&over[0], s_over);
  __syncthreads();
  unsigned long long before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < 16; ++w) {
    const unsigned long long t = wave_tot[w];
    if (w < wave) before += t;
    total += t;
  }
  unsigned long long run = base + before + incl - mine;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (q + j < nq) hit_off[q + j] = run;
    run += x[j];
    if (x[j] > hl_cap) over[1 + s_over_base + my_over++] = q + j;
  }
  if (tid == 0 && q0 + 4096u >= nq) hit_off[nq] = base + total;
}

// 256 * ITEMS keys (count << 16 | gid, zero = none) from LDS, ordered descending by a bitonic network over the 256
// threads of a workgroup, the first n_out unpacked to hc / hg.  Element i = r * 256 + tid lives in register k[r]:
// a stage whose partner i ^ j lies in the same thread (j >= 256) or the same wave (j < 64) needs no LDS and no
// barrier; only the strides 64 and 128 go through LDS (9 of the 66 stages of 2048 keys).
template <int ITEMS>
__device__ __forceinline__ void bitonic_desc_256(uint32_t *keys, uint32_t tid, uint32_t n_out, uint32_t *hc, uint32_t *hg) {
  uint32_t k[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) k[r] = keys[r * 256 + tid];
  constexpr uint32_t P = 256u * ITEMS;
#pragma unroll
  for (uint32_t kk = 2; kk <= P; kk <<= 1) {
#pragma unroll
    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
      if (j >= 256) {   // partner in this thread
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
          const int r2 = r ^ (int)(j >> 8);
          if (r2 > r) {
            const bool desc = (((uint32_t)r * 256u) & kk) == 0;   // (kk > j >= 256: bit kk of i is a bit of r)
            const uint32_t x = k[r], y = k[r2];
            const uint32_t hi = x > y ? x : y, lo = x > y ? y : x;
            k[r] = desc ? hi : lo;
            k[r2] = desc ? lo : hi;
          }
        }
      } else {
        if (j >= 64) {   // partner in another wave: through LDS
          __syncthreads();
#pragma unroll
          for (int r = 0; r < ITEMS; ++r) keys[r * 256 + tid] = k[r];
          __syncthreads();
        }
        const bool lower = (tid & j) == 0;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
          const uint32_t i = (uint32_t)r * 256u + tid;
          const bool desc = (i & kk) == 0;
          const uint32_t x = k[r];
          const uint32_t y = j >= 64 ? keys[r * 256 + (tid ^ j)] : (uint32_t)__shfl_xor((int)x, (int)j, 64);
          const uint32_t hi = x > y ? x : y, lo = x > y ? y : x;
          k[r] = (desc == lower) ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t i = (uint32_t)r * 256u + tid;
    if (i < n_out) { hc[i] = k[r] >> 16; hg[i] = k[r] & 0xFFFFu; }
  }
}

// hitlist_emit_kernel, two parts.  First, one wave per query (4 per workgroup): a query whose list fits -- the usual
// case -- is a copy of its ordered entries to [hit_off[q], hit_off[q+1]).  Then the queries whose lists overflowed, one
// workgroup each (grid-stride over the list the scan made): the hits are thresholded from the query's counter row
// (src/niqki_index.cpp:662-666) into LDS as count << 16 | gid -- distinct keys whose descending order is
// greater<pair<count, gid>> (:685) -- and ordered there by a bitonic network.  keys: P words of LDS (P >= 256).
__global__ __launch_bounds__(256) void hitlist_emit_kernel(HitsArgs a, const uint32_t *hl, uint32_t hl_cap, const uint32_t *over, uint32_t P) {
  extern __shared__ __align__(16) uint32_t keys[];
  __shared__ uint32_t s_n;
  const uint32_t tid = threadIdx.x;
  {
    const uint32_t lane = tid & 63u, q = blockIdx.x * 4 + (tid >> 6);
    if (q < a.nq) {
      const unsigned long long seg0 = a.hit_off[q], n_all = a.hit_off[q + 1] - seg0;
      if (n_all && n_all <= hl_cap) {
        const uint32_t *src = hl + (uint64_t)q * hl_cap;
        for (uint32_t i = lane; i < (uint32_t)n_all; i += 64) {
          const unsigned long long pos = seg0 + i;
          if (pos < a.capacity) {
            const uint32_t e = src[i];
            a.hit_counts[pos] = e >> 16;
            a.hit_gids[pos] = e & 0xFFFFu;
          }
        }
      }
    }
  }
  const uint32_t n_over = over[0];
  for (uint32_t k = blockIdx.x; k < n_over; k += gridDim.x) {
    const uint32_t q = over[1 + k];
    const unsigned long long seg0 = a.hit_off[q], n_all = a.hit_off[q + 1] - seg0;
    const uint16_t *row = a.counts + (uint64_t)q * a.stride + a.gid_begin;
    if (n_all > P) {
      // more hits than the network holds (a threshold that lets a sixth of the index through): one wave thresholds the
      // row in descending gid and orders it with the stable radix passes of hits_sort_kernel, through tmp_*
      if (tid < 64) {
        uint32_t *cur = keys;   // 256 words
        const uint32_t lane = tid;
        const uint64_t lt_mask = (1ULL << lane) - 1ULL;
        unsigned long long run = 0;
        for (uint32_t top = (a.n_gids + 63u) & ~63u; top > 0; top -= 64) {
          const uint32_t i = top - 1 - lane;
          const uint32_t c = i < a.n_gids ? (uint32_t)row[i] : 0u;
          const bool hit = i < a.n_gids && c >= a.min_score;
          const uint64_t bal = __ballot(hit);
          if (hit) {
            const unsigned long long pos = seg0 + run + __popcll(bal & lt_mask);
            if (pos < a.capacity) { a.hit_counts[pos] = c; a.hit_gids[pos] = a.gid_begin + i; }
          }
          run += __popcll(bal);
        }
        unsigned long long seg1 = seg0 + n_all;
        if (seg1 > a.capacity) seg1 = a.capacity;
        if (seg0 < seg1) {
          const unsigned long long n = seg1 - seg0;
          uint32_t *tc = a.tmp_counts + seg0, *tg = a.tmp_gids + seg0;
          uint32_t *hc = a.hit_counts + seg0, *hg = a.hit_gids + seg0;
          __threadfence_block();
          radix_pass_desc(hc, hg, tc, tg, n, 0, cur, lane);
          __threadfence_block();
          radix_pass_desc(tc, tg, hc, hg, n, 8, cur, lane);
        }
      }
      __syncthreads();
      continue;
    }
    if (tid == 0) s_n = 0;
    // the network's size for this query
    uint32_t Pq = 256;
    while (Pq < n_all) Pq <<= 1;   // (n_all <= P here, and P >= 256 is a power of two)
    for (uint32_t i = tid; i < Pq; i += 256) keys[i] = 0u;
    __syncthreads();
    // 8 counters per lane and load (rows start on 128-byte lines: NIQKI_ROW_STRIDE)
    const bool vec = (((uintptr_t)row) & 15u) == 0;
    for (uint32_t i0 = tid * 8; i0 < a.n_gids; i0 += 256 * 8) {
      uint32_t c[8];
      if (vec && i0 + 8 <= a.n_gids) {
        const uint4 w = *(const uint4 *)(row + i0);
        c[0] = w.x & 0xFFFFu; c[1] = w.x >> 16; c[2] = w.y & 0xFFFFu; c[3] = w.y >> 16;
        c[4] = w.z & 0xFFFFu; c[5] = w.z >> 16; c[6] = w.w & 0xFFFFu; c[7] = w.w >> 16;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = i0 + j < a.n_gids ? (uint32_t)row[i0 + j] : 0u;
      }
      uint32_t m = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) m |= (uint32_t)(i0 + j < a.n_gids && c[j] >= a.min_score) << j;
      if (m) {
        uint32_t at = atomicAdd(&s_n, (uint32_t)__builtin_popcount(m));
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (m >> j & 1u) keys[at++] = (c[j] << 16) | (a.gid_begin + i0 + j);
      }
    }
    __syncthreads();
    // bitonic sort, descending (the zero keys behind the n_all real ones end up last: a real key 0 -- count 0 of
    // genome 0 at min_score 0 -- is the smallest key and belongs there too), in registers: see bitonic_desc_256
    const unsigned long long room = seg0 < a.capacity ? a.capacity - seg0 : 0ull;
    const uint32_t n_out = (uint32_t)(n_all < room ? n_all : room);
    if (Pq <= 256) bitonic_desc_256<1>(keys, tid, n_out, a.hit_counts + seg0, a.hit_gids + seg0);
    else if (Pq == 512) bitonic_desc_256<2>(keys, tid, n_out, a.hit_counts + seg0, a.hit_gids + seg0);
    else if (Pq == 1024) bitonic_desc_256<4>(keys, tid, n_out, a.hit_counts + seg0, a.hit_gids + seg0);
    else bitonic_desc_256<8>(keys, tid, n_out, a.hit_counts + seg0, a.hit_gids + seg0);
    __syncthreads();
  }
}

hipError_t launch_hitlist_scan(const uint32_t *n, const HitsArgs &a, uint32_t hl_cap, uint32_t *over, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;   // (over[0] = 0: the gather launch that made n has done it, CandOut::hl_over)
  hipLaunchKernelGGL(hitlist_scan_kernel, dim3((a.nq + 4095u) / 4096u), dim3(1024), 0, stream, n, a.nq, hl_cap, a.hit_off, over);
  return hipGetLastError();
}

hipError_t launch_hitlist_emit(const HitsArgs &a, const uint32_t *hl, uint32_t hl_cap, const uint32_t *over, hipStream_t stream) {
  if (a.nq == 0) return hipSuccess;
  // (the network: 2048 keys = 8 KB of LDS, eight workgroups per CU; a query with more hits takes the wave path)
  uint32_t P = 256;
  while (P < a.n_gids && P < 2048u) P <<= 1;
  hipLaunchKernelGGL(hitlist_emit_kernel, dim3((a.nq + 3) / 4), dim3(256), (size_t)P * 4, stream, a, hl, hl_cap, over, P);
  return hipGetLastError();
}

hipError_t launch_hits_count(const HitsArgs &a, hipStream_t stream) {
  if (a.nq == 0 || a.n_blk == 0) return hipSuccess;
  if (a.n_blk >= 8) hipLaunchKernelGGL(hits_count_kernel<true>, dim3(a.nq), dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL(hits_count_kernel<false>, dim3((a.nq + 15) / 16), dim3(1024), 0, stream, a);
  hipLaunchKernelGGL(hits_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_hits_emit(const HitsArgs &a, hipStream_t stream) {
  if (a.nq == 0 || a.n_blk == 0) return hipSuccess;
  hipLaunchKernelGGL(hits_compact_kernel, dim3(a.nq * a.n_blk), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(hits_sort_kernel, dim3((a.nq + 3) / 4), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace nq
