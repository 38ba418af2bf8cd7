// nq_sketch.hip -- kernel #1/#2: canonical k-mer rolling hash, per-slot
// HyperMinHash min reduction in LDS, and densification, for gfx950.
//
// Replaces Index::compute_sketch (src/niqki_index.cpp:335-358, with
// update_kmer :225-229, update_kmer_RC :233-236, str2numstrand :255-273,
// rcb :240-250, revhash64/unrevhash64 :291-305, get_fingerprint :277-287) and
// Index::sketch_densification (:313-331).
//
// Shape: one workgroup per sketch (times `splits` for long single records).
// The F = 2^S sketch cells live in LDS as u32 with 0xFFFFFFFF (= int32 -1)
// for "empty", so the per-slot minimum is one unsigned ds_min_u32.  A record
// is cut into chunks of CHUNK k-mers; each lane rolls one chunk: K-1 cheap
// warm-up steps rebuild the forward / reverse-complement words (a k-mer only
// depends on its own K bases, see DESIGN.md "positional codes"), then CHUNK
// hash steps.  The kernel is integer-ALU bound (4 64-bit multiplies per
// k-mer), HBM traffic is 1 byte per base.
#include "nq_kernels.h"
#include "../../include/niqki_hip.h"

#include <cstdlib>

namespace nq {

// Per-byte code table, built in LDS by the first 256 threads.
//   bits 1:0  forward code of the rolling update   (:114-123  A0 C1 G2 T3, else 0)
//   bits 3:2  reverse-complement code of the update (:211-221 A3 C2 G1, else 0)
//   bits 5:4  case-insensitive digit of the K-1 prefix (:255-273)
//   bit  6    byte is a legal prefix character (ACGTacgt)
__device__ __forceinline__ uint8_t code_entry(uint32_t c) {
  uint32_t fwd = c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u;
  uint32_t rc = c == 'A' ? 3u : c == 'C' ? 2u : c == 'G' ? 1u : 0u;
  uint32_t u = c & 0xDFu;  // fold lower case onto upper case
  uint32_t ok = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? 1u : 0u;
  uint32_t pd = u == 'C' ? 1u : u == 'G' ? 2u : u == 'T' ? 3u : 0u;
  return (uint8_t)(fwd | (rc << 2) | ((ok ? pd : 0u) << 4) | (ok << 6));
}

// 16 bytes of a byte stream that starts at an arbitrary address: the stream is
// fetched as dword-aligned uint4 loads and re-aligned with v_alignbyte.  Two loads are
// in flight: the one next16() issues is consumed by the call after it, so a caller that
// does a group's worth of work between calls never waits for memory.  Reads at most 35
// bytes past the last byte handed out (NIQKI_SEQ_PAD covers the end of the buffer).
struct ByteStream {
  const uint32_t *q;  // dword-aligned cursor
  uint32_t sh;        // byte phase 0..3
  uint4 cur, nxt;
  __device__ __forceinline__ void open(const uint8_t *p) {
    uintptr_t a = (uintptr_t)p;
    sh = (uint32_t)(a & 3u);
    q = (const uint32_t *)(a & ~(uintptr_t)3);
    cur = *(const uint4 *)q;  // dword aligned 16-byte loads
    nxt = *(const uint4 *)(q + 4);
    q += 8;
  }
  // returns the next 16 stream bytes as 4 dwords (little endian); loads never go past qmax (a lane
  // that steps through more groups than its own chunk holds, beside lanes with longer chunks)
  __device__ __forceinline__ uint4 next16(const uint32_t *qmax = nullptr) {
    const uint4 nn = *(const uint4 *)((qmax && q > qmax) ? qmax : q);
    q += 4;
    uint4 r;
    r.x = __builtin_amdgcn_alignbyte(cur.y, cur.x, sh);
    r.y = __builtin_amdgcn_alignbyte(cur.z, cur.y, sh);
    r.z = __builtin_amdgcn_alignbyte(cur.w, cur.z, sh);
    r.w = __builtin_amdgcn_alignbyte(nxt.x, cur.w, sh);
    cur = nxt;
    nxt = nn;
    return r;
  }
};

__device__ __forceinline__ uint32_t dword_of(const uint4 &v, int i) {
  return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// In-LDS densification, pass-parallel restatement of the serial loop at
// src/niqki_index.cpp:313-331 (equivalence: DESIGN.md "densification").
// Within one pass every cell occupied at pass start proposes itself to its
// target cell with atomicMin of (1<<31 | source index << W | value); real
// values are < 2^W so occupied targets are never changed, and among several
// proposals the smallest source index wins, which is the serial loop's
// first-writer rule.  Returns with cells still empty only where the reference
// would loop forever (F consecutive passes without a fill prove a fixpoint).
template <int BLOCK>
__device__ void densify_lds(uint32_t *sk, const Derived &d, uint32_t *s_flag) {
  const uint32_t F = d.F;
  const uint32_t tid = threadIdx.x;
  // count empties
  uint32_t local = 0;
  for (uint32_t i = tid; i < F; i += BLOCK) local += (sk[i] == kEmpty32);
  if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
  __syncthreads();
  if (local) atomicAdd(&s_flag[0], local);
  __syncthreads();
  uint32_t empty = s_flag[0];
  if (empty == 0 || empty == F) return;
  uint32_t step = 0, idle = 0;
  while (true) {
    // propose: an empty cell remembers the smallest source cell that points at it (the
    // first writer of the reference's ascending loop); occupied cells hold values below
    // 2^31 and are left alone by the min.  No assumption on the value range: after
    // niqki_select_best_H cells may exceed 2^W.
    for (uint32_t i = tid; i < F; i += BLOCK) {
      uint32_t v = sk[i];
      if (v < 0x80000000u) {
        // hash_family(v, step) % F, src/niqki_index.cpp:308-310,:319 (low bits only)
        uint32_t t = ((uint32_t)unrev64(v) + step * (uint32_t)rev64(v)) & (F - 1u);
        atomicMin(&sk[t], 0x80000000u | i);
      }
    }
    __syncthreads();
    // resolve: copy from the winning source (occupied since before this pass, so no
    // other thread writes it now)
    uint32_t filled = 0;
    for (uint32_t i = tid; i < F; i += BLOCK) {
      uint32_t v = sk[i];
      if (v >= 0x80000000u && v != kEmpty32) { sk[i] = sk[v & 0x7FFFFFFFu]; ++filled; }
    }
    if (filled) atomicAdd(&s_flag[1], filled);
    __syncthreads();
    uint32_t tot = s_flag[1];
    __syncthreads();
    if (tid == 0) s_flag[1] = 0;
    empty -= tot;
    ++step;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
    __syncthreads();
  }
}

// Densification over DISTINCT VALUES (short-read path).  A fill copies a value,
// so every copy of a value proposes the same target in a pass and only the copy
// with the smallest index can win: it is enough to track, per fingerprint value v,
// mi[v] = smallest cell index holding v (at pass start), and the two hash words of
// v, which never change.  A pass is then one proposal per distinct value (~120 for a
// 150-base read) instead of one per occupied cell (up to F), with the same result
// as densify_lds.  aux: 3*R words of LDS (mi, A = low word of unrev(v), B = low word
// of rev(v)).
template <int BLOCK>
__device__ void densify_lds_distinct(uint32_t *sk, const Derived &d, uint32_t *s_flag, uint32_t *aux) {
  const uint32_t F = d.F, R = d.R, W = d.W;
  const uint32_t tid = threadIdx.x;
  uint32_t *mi = aux, *ha = aux + R, *hb = aux + 2 * R;
  uint32_t local = 0;
  for (uint32_t i = tid; i < F; i += BLOCK) local += (sk[i] == kEmpty32);
  for (uint32_t v = tid; v < R; v += BLOCK) mi[v] = kEmpty32;
  if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
  __syncthreads();
  if (local) atomicAdd(&s_flag[0], local);
  for (uint32_t i = tid; i < F; i += BLOCK) {
    uint32_t v = sk[i];
    if (v != kEmpty32) atomicMin(&mi[v], i);
  }
  __syncthreads();
  uint32_t empty = s_flag[0];
  if (empty == 0 || empty == F) return;
  for (uint32_t v = tid; v < R; v += BLOCK)
    if (mi[v] != kEmpty32) { ha[v] = (uint32_t)unrev64(v); hb[v] = (uint32_t)rev64(v); }
  uint32_t step = 0, idle = 0;
  while (true) {
    for (uint32_t v = tid; v < R; v += BLOCK) {
      const uint32_t m = mi[v];
      if (m != kEmpty32) {
        const uint32_t t = (ha[v] + step * hb[v]) & (F - 1u);  // hash_family(v, step) % F, :308-310,:319
        if (sk[t] >= 0x80000000u) atomicMin(&sk[t], 0x80000000u | (m << W) | v);
      }
    }
    __syncthreads();
    uint32_t filled = 0;
    for (uint32_t v = tid; v < R; v += BLOCK) {
      const uint32_t m = mi[v];
      if (m != kEmpty32) {
        const uint32_t t = (ha[v] + step * hb[v]) & (F - 1u);
        if (sk[t] == (0x80000000u | (m << W) | v)) {
          sk[t] = v;
          if (t < m) mi[v] = t;
          ++filled;
        }
      }
    }
    if (filled) atomicAdd(&s_flag[1], filled);
    __syncthreads();
    const uint32_t tot = s_flag[1];  // every proposed-to cell now holds its winner's value
    __syncthreads();
    if (tid == 0) s_flag[1] = 0;
    empty -= tot;
    ++step;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
    __syncthreads();
  }
}

// Rolling update of the forward / reverse-complement words with the code-table
// entry `e` of the incoming base (src/niqki_index.cpp:225-236) and the canonical
// k-mer (:345).  KFIX != 0 fixes K at compile time (K = 31: constant shifts).
template <int KFIX>
__device__ __forceinline__ uint64_t roll_step(uint32_t e, uint64_t &fw, uint64_t &rc, const Derived &d,
                                              uint32_t rc_shift) {
  if (KFIX) {
    constexpr uint64_t mask = (1ULL << (2 * KFIX)) - 1ULL;
    fw = ((fw << 2) | (uint64_t)(e & 3u)) & mask;
    rc = (rc >> 2) | ((uint64_t)((e >> 2) & 3u) << (2 * KFIX - 2));
  } else {
    fw = ((fw << 2) | (uint64_t)(e & 3u)) & d.kmer_mask;
    rc = (rc >> 2) | ((uint64_t)((e >> 2) & 3u) << rc_shift);
  }
  return fw < rc ? fw : rc;
}

// slot + fingerprint of a canonical k-mer and the per-slot min (:346-355)
// hsel: 0 = sk holds all F cells; 1 / 2 = sk holds the lower / upper half of the slots (S = 16: the
// cells of a whole sketch do not fit LDS) and k-mers of the other half change nothing
__device__ __forceinline__ void sketch_update(uint64_t canon, const Derived &d, uint32_t *sk, bool live, uint32_t hsel = 0) {
  uint32_t slot = slot_of(canon, d.S);
  uint32_t fp = fingerprint(rev64(canon), d.M, d.mask_m, d.max_rem);
  if (hsel) {  // uniform
    live = live && (slot >> (d.S - 1)) == hsel - 1u;
    slot &= (d.F >> 1) - 1u;
  }
  fp = live ? fp : kEmpty32;  // a min with "empty" changes nothing
  atomicMin(&sk[slot], fp);
}

// ---- the filtered long-record path ------------------------------------------------------------
// Issue costs on gfx950 (tools/ubench_opcodes.hip, profiles/r03_opcode_costs.txt): v_add / v_sub /
// v_and / v_or / v_xor / v_mov / v_lshrrev_b32 take ~2.4 SIMD cycles per wave instruction, EVERY other
// vector opcode ~4.3 -- 32-bit multiplies, v_mad_u64_u32 (4.4), 64-bit shifts and compares included.
// So the step below is written for the fewest instructions, multiplies are not what to avoid:
//   * the code table for K = 31 holds 8-byte entries {forward code, rc code << 28}: both rolling
//     updates are one 64-bit shift plus an `or` (5 instructions instead of 7);
//   * the 64 x 64 -> 64 multiplies run as chains of v_mad_u64_u32 (the addend carries the cross terms);
//   * candidates are stored under the exec mask (compare, 2 x mbcnt, 1 address instruction) into a
//     wave-private LIFO stack whose top lives in a scalar register: no ring wrap, no scratch slot.

typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;

// a * b (32 x 32 -> 64) and a * b + c as ONE v_mad_u64_u32 each: the empty asm makes the whole 64-bit
// result "used", so the compiler cannot narrow the expression back into v_mul_lo_u32 + adds
__device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) {
  uint64_t r = (uint64_t)a * (uint64_t)b;
  asm("" : "+v"(r));
  return r;
}
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t r = (uint64_t)a * (uint64_t)b + c;
  asm("" : "+v"(r));
  return r;
}

// High word of rev64(canon): enough to bound the fingerprint from below.
__device__ __forceinline__ uint32_t rev64_hi(uint64_t canon) {
  uint64_t x = ((canon >> 32) ^ canon) * kRevMul;
  x = ((x >> 32) ^ x) * kRevMul;
  return (uint32_t)(x >> 32);
}

// One xorshift-multiply round of mix64 (src/niqki_index.cpp:292,:301): x = ((x >> 32) ^ x) * c with c
// = chi:clo.  The low 32 bits of the product of the high words' cross terms ride in the addend.
__device__ __forceinline__ void mix_round(uint32_t &lo, uint32_t &hi, uint32_t clo, uint32_t chi) {
  const uint32_t y = lo ^ hi;
  const uint64_t t = mul64(hi, clo);
  const uint64_t p = mad64(y, chi, t);   // low word: y * chi + hi * clo
  const uint64_t q = mul64(y, clo);
  lo = (uint32_t)q;
  hi = (uint32_t)p + (uint32_t)(q >> 32);
}
// the same round when only the high word of the product is wanted
__device__ __forceinline__ uint32_t mix_round_hi(uint32_t lo, uint32_t hi, uint32_t clo, uint32_t chi) {
  const uint32_t y = lo ^ hi;
  const uint64_t t = mul64(hi, clo);
  const uint64_t p = mad64(y, chi, t);
  return (uint32_t)p + __umulhi(y, clo);
}
__device__ __forceinline__ uint32_t rev64_hi_mad(uint64_t canon) {
  uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
  mix_round(lo, hi, (uint32_t)kRevMul, (uint32_t)(kRevMul >> 32));
  return mix_round_hi(lo, hi, (uint32_t)kRevMul, (uint32_t)(kRevMul >> 32));
}

constexpr uint32_t kFastKMin = 17;   // smallest K of the fast filtered path: 2K - 2 >= 32
constexpr uint32_t kStack = 192;  // candidate k-mers per wave-private stack: < 64 left by a drain + two steps of <= 64

// Candidate store of one step: lanes whose hash word hh is below thr push canon onto the wave's stack
// (LDS byte address of its top in `top`, wave-uniform, advanced here).  gfx950 wants two wait states
// between a vector compare and a vector instruction that reads its mask as an operand (the mbcnt): the two
// scalar instructions in between are that.
__device__ __forceinline__ void push_candidates(uint32_t hh, uint32_t thr, uint64_t canon, uint32_t &top) {
  uint32_t n, r;
  uint64_t save;
  asm volatile(
      "v_cmp_gt_u32 vcc, %[thr], %[hh]\n\t"
      "s_and_saveexec_b64 %[save], vcc\n\t"
      "s_bcnt1_i32_b64 %[n], vcc\n\t"
      "v_mbcnt_lo_u32_b32 %[r], vcc_lo, 0\n\t"
      "v_mbcnt_hi_u32_b32 %[r], vcc_hi, %[r]\n\t"
      "v_lshl_add_u32 %[r], %[r], 3, %[top]\n\t"
      "ds_write_b64 %[r], %[canon]\n\t"
      "s_mov_b64 exec, %[save]\n\t"
      "s_lshl3_add_u32 %[top], %[n], %[top]"
      : [n] "=&s"(n), [r] "=&v"(r), [save] "=&s"(save), [top] "+s"(top)
      : [thr] "s"(thr), [hh] "v"(hh), [canon] "v"(canon)
      : "vcc", "scc", "memory");
}

// LDS layout of sketch_kernel (byte offsets from the start of its dynamic LDS, which is LDS address 0: the
// kernel has no static LDS): the K = 31 code table first, so that a look-up address is the byte value
// times 8 (one SDWA shift), and the sketch cells at a fixed offset (an instruction immediate).
constexpr uint32_t kLut64Off = 0;       // 256 x {forward code, rc code << 28}
constexpr uint32_t kLutOff = 2048;      // 256 code_entry bytes
constexpr uint32_t kFlagOff = 2304;     // 4 words
constexpr uint32_t kCellOff = 2320;     // the sketch cells; behind them the distinct-value tables or the candidate stacks

// byte B of w, times 8: the offset of its 8-byte code table entry
template <int B>
__device__ __forceinline__ uint32_t byte_x8(uint32_t w) {
  uint32_t r;
  if (B == 0) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(w));
  if (B == 1) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(w));
  if (B == 2) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(w));
  if (B == 3) asm("v_lshlrev_b32_sdwa %0, 3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(w));
  return r;
}
// the 16 code table entries of 16 stream bytes
__device__ __forceinline__ void lut64_16(const uint4 &v, uint32_t lds0, uint64_t (&e)[16]) {
#define NQ_E(J, W, B) e[J] = *(lds_u64_t *)(uintptr_t)(lds0 + kLut64Off + byte_x8<B>(W));
  NQ_E(0, v.x, 0) NQ_E(1, v.x, 1) NQ_E(2, v.x, 2) NQ_E(3, v.x, 3)
  NQ_E(4, v.y, 0) NQ_E(5, v.y, 1) NQ_E(6, v.y, 2) NQ_E(7, v.y, 3)
  NQ_E(8, v.z, 0) NQ_E(9, v.z, 1) NQ_E(10, v.z, 2) NQ_E(11, v.z, 3)
  NQ_E(12, v.w, 0) NQ_E(13, v.w, 1) NQ_E(14, v.w, 2) NQ_E(15, v.w, 3)
#undef NQ_E
}
// ... of the first (HALF = 0) or last eight of them, into e[0..8) or e[8..16)
template <int HALF>
__device__ __forceinline__ void lut64_8(const uint4 &v, uint32_t lds0, uint64_t (&e)[16]) {
#define NQ_E(J, W, B) e[J] = *(lds_u64_t *)(uintptr_t)(lds0 + kLut64Off + byte_x8<B>(W));
  if (HALF == 0) {
    NQ_E(0, v.x, 0) NQ_E(1, v.x, 1) NQ_E(2, v.x, 2) NQ_E(3, v.x, 3)
    NQ_E(4, v.y, 0) NQ_E(5, v.y, 1) NQ_E(6, v.y, 2) NQ_E(7, v.y, 3)
  } else {
    NQ_E(8, v.z, 0) NQ_E(9, v.z, 1) NQ_E(10, v.z, 2) NQ_E(11, v.z, 3)
    NQ_E(12, v.w, 0) NQ_E(13, v.w, 1) NQ_E(14, v.w, 2) NQ_E(15, v.w, 3)
  }
#undef NQ_E
}
// 64-bit shifts the compiler cannot look into (it otherwise re-derives halves of the result with extra
// instructions)
__device__ __forceinline__ uint64_t shl2_64(uint64_t x) {
  uint64_t r;
  asm("v_lshlrev_b64 %0, 2, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ uint64_t shr2_64(uint64_t x) {
  uint64_t r;
  asm("v_lshrrev_b64 %0, 2, %1" : "=v"(r) : "v"(x));
  return r;
}

// slot + fingerprint + per-slot min of up to 64 candidates (:346-355), whole sketch in LDS at kCellOff
// upwards (S <= 15).  The hash of the filter is recomputed: passing its words through the stack costs
// more register traffic than the 5 instructions saved.
__device__ __forceinline__ void candidate_update(uint64_t canon, const Derived &d, uint32_t lds0, bool live) {
  uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
  uint32_t ulo = lo, uhi = hi;
  mix_round(ulo, uhi, (uint32_t)kUnrevMul, (uint32_t)(kUnrevMul >> 32));
  const uint32_t uh = mix_round_hi(ulo, uhi, (uint32_t)kUnrevMul, (uint32_t)(kUnrevMul >> 32));
  mix_round(lo, hi, (uint32_t)kRevMul, (uint32_t)(kRevMul >> 32));
  mix_round(lo, hi, (uint32_t)kRevMul, (uint32_t)(kRevMul >> 32));
  // get_fingerprint (:277-287) of h = hi : lo ^ hi.  A candidate's high word is below 2^29 and, but for one
  // hash in 2^29, not 0: its leading zeros are those of the high word (one v_ffbh); the rare zero high word
  // takes the general form for the whole wave.
  const uint32_t hl = lo ^ hi;
  uint32_t lz = (uint32_t)__builtin_clz(hi);
  if (__builtin_expect(__any(hi == 0u), 0)) lz = clz64(((uint64_t)hi << 32) | hl);
  const uint32_t rem = lz < d.max_rem ? d.max_rem - lz : 0u;
  uint32_t fp = (hl & d.mask_m) + (rem << d.M);
  fp = live ? fp : kEmpty32;  // a min with "empty" changes nothing
  const uint32_t cell = (uh >> (30u - d.S)) & ~3u;   // byte offset of the slot's cell
  __hip_atomic_fetch_min((lds_u32_t *)(uintptr_t)(lds0 + kCellOff + cell), fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// All records of one sketch, this workgroup's share of the chunks.
// FILTER (long inputs only): a k-mer whose hash has fewer than T leading zeros
// (hi word >= thr) has a larger fingerprint than any k-mer with at least T, so it
// can only matter for a slot that no such k-mer reaches.  Those k-mers (7 of 8 at
// T = 3) skip the slot hash and the LDS min; the others are pushed onto a
// wave-private stack and finished 64 at a time.  The caller re-runs the records
// unfiltered if any slot is still empty afterwards, so the result is exact.
// GROUPS: most 16-base groups per chunk; a record's chunks are sized so that its last round of
// chunks over the workgroup's lanes is nearly full (5 Mbp over 1024 lanes: 10 rounds of 496 k-mers).
// ZERO: the kernel's LDS layout starts at LDS address 0 (checked at kernel entry), so the fast path
// addresses the code table and the cells with instruction immediates.
template <int BLOCK, int GROUPS, int KFIX, bool FILTER, bool HALF, bool ZERO>
__device__ void roll_records(const SketchArgs &a, uint32_t entry, uint32_t part, uint32_t *sk, const uint8_t *lut,
                             uint64_t *stack_base, uint32_t thr, uint32_t hsel) {
  const Derived &d = a.d;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t Km1 = d.K - 1u;
  const uint32_t rc_shift = 2u * d.K - 2u;
  // 8-byte code table, mad chains, S <= 15.  K = 31 is a compile-time specialisation; the KFIX = 0 kernel takes
  // the same path for any K in 17..31 (kFastKMin: the rc code of a table entry must sit in its high word): the
  // shifts by 2 are immediates whatever K, only the table's contents, the mask of the forward word's high half
  // and the position the warm-up starts at depend on K.
  constexpr bool FAST = FILTER && !HALF && ZERO && (KFIX == 31 || KFIX == 0);
  const bool fast_k = KFIX == 31 || (d.K >= kFastKMin && d.K <= 31u);
  const uint32_t mask_hi = KFIX == 31 ? 0x3FFFFFFFu : __builtin_amdgcn_readfirstlane((uint32_t)(d.kmer_mask >> 32));
  const uint64_t fw_mask = ((uint64_t)mask_hi << 32) | 0xFFFFFFFFull;
  const uint32_t warm_back = KFIX == 31 ? 0u : 31u - d.K;   // the 30 warm-up steps start this many bases before the chunk
  constexpr uint32_t lds0 = 0;
  uint64_t *stack = stack_base + (tid >> 6) * kStack;
  // LDS byte address of the stack's bottom / top (wave-uniform: scalar registers)
  const uint32_t bottom = FILTER ? __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_u64_t *)stack) : 0u;
  uint32_t top = bottom;
  const uint32_t lane8 = lane * 8u;
  auto drain64 = [&]() {   // the 64 youngest candidates
    const uint64_t c = *(lds_u64_t *)(uintptr_t)(top - 512u + lane8);
    if (FAST) candidate_update(c, d, lds0, true);
    else sketch_update(c, d, sk, true, hsel);
    top -= 512u;
  };
  auto drain_rest = [&]() {   // fewer than 64 left
    const uint32_t n = (top - bottom) >> 3;
    const bool live = lane < n;
    const uint64_t c = live ? stack[lane] : 0ull;
    if (FAST) candidate_update(c, d, lds0, live);
    else sketch_update(c, d, sk, live, hsel);
    top = bottom;
  };

  uint32_t r0 = a.entry_rec ? a.entry_rec[entry] : entry;
  uint32_t r1 = a.entry_rec ? a.entry_rec[entry + 1] : entry + 1;
  for (uint32_t rec = r0; rec < r1; ++rec) {
    const uint64_t b0 = a.rec_off[rec], b1 = a.rec_off[rec + 1];
    const uint64_t len = b1 - b0;
    if (len <= d.K) continue;              // src/niqki_index.cpp:395,:450
    const uint64_t n_kmers = len - d.K;    // last k-mer skipped, :342
    // chunk length: the fewest rounds of GROUPS-group chunks, then the shortest chunks that still fit them
    uint32_t groups = 1;
    if (GROUPS > 1) {
      const uint64_t share = n_kmers / a.splits + 1u;
      const uint64_t rounds = (share + (uint64_t)BLOCK * 16u * GROUPS - 1u) / ((uint64_t)BLOCK * 16u * GROUPS);
      groups = (uint32_t)((share + rounds * BLOCK * 16u - 1u) / (rounds * BLOCK * 16u));
      groups = groups < 1u ? 1u : groups > (uint32_t)GROUPS ? (uint32_t)GROUPS : groups;
      groups = __builtin_amdgcn_readfirstlane(groups);
    }
    const uint32_t CHUNK = 16u * groups;
    const uint64_t n_chunks = (n_kmers + CHUNK - 1) / CHUNK;
    const uint64_t c_lo = n_chunks * part / a.splits;
    const uint64_t c_hi = n_chunks * (part + 1) / a.splits;
    const uint8_t *base = a.seqs + b0;
    // Whole waves step together (uniform loop control: the filtered path's stack
    // is wave-collective and its top lives in a scalar register).
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(tid & ~63u);
    for (uint64_t cb = c_lo + wave0; cb < c_hi; cb += BLOCK) {
      const uint64_t c = cb + lane;
      const bool alive = c < c_hi;
      const uint64_t i0 = (alive ? c : c_lo) * CHUNK;
      const uint64_t left = n_kmers - i0;
      const uint32_t cnt = alive ? (left < CHUNK ? (uint32_t)left : CHUNK) : 0u;
      // ---- warm-up: K-1 rolling updates from zero rebuild both words ----
      // Positions < K-1 of a record carry the str2numstrand digits
      // (case-insensitive; any other byte among the first K-1 zeroes all of
      // them, :255-273) and their complements (rcb, :240-250); every later
      // position carries the codes of the rolling tables.
      uint64_t fw = 0, rc = 0;
      if (FAST && fast_k && __all(i0 >= Km1 + warm_back)) {
        // no lane of the wave starts inside a record's first K-1 positions (all chunks but a record's
        // first): the 30 steps are plain rolling updates from the 8-byte code table, 4 instructions each.
        // (K < 31: 30 steps all the same, from 31 - K bases further back -- the forward word is masked below,
        // the reverse word's older codes leave at the bottom.)
        ByteStream ws;
        ws.open(base + i0 - warm_back);
        uint64_t ew[32];
        {
          uint64_t e16[16];
          lut64_16(ws.next16(), lds0, e16);
#pragma unroll
          for (int j = 0; j < 16; ++j) ew[j] = e16[j];
          lut64_16(ws.next16(), lds0, e16);
#pragma unroll
          for (int j = 0; j < 16; ++j) ew[16 + j] = e16[j];
        }
#pragma unroll
        for (int j = 0; j < 30; ++j) {
          fw = shl2_64(fw) | (uint32_t)ew[j];
          rc = shr2_64(rc) | (ew[j] & 0xFFFFFFFF00000000ULL);
        }
        if (KFIX != 31) fw &= fw_mask;   // (K = 31: 30 steps from zero stay below 2^60)
      } else {
        uint32_t ok = 1;
        if (i0 < Km1) {
          ByteStream ps;
          ps.open(base);
          uint4 p0 = ps.next16(), p1 = ps.next16();
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            uint32_t w = dword_of(j < 16 ? p0 : p1, (j & 15) >> 2);
            uint32_t e = lut[(w >> (8 * (j & 3))) & 0xFFu];
            if ((uint32_t)j < Km1) ok &= (e >> 6) & 1u;
          }
        }
        ByteStream ws;
        ws.open(base + i0);
        uint4 g0 = ws.next16(), g1 = ws.next16();
        uint32_t ew[32];  // all table look-ups first, then the dependent updates
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          uint32_t w = dword_of(j < 16 ? g0 : g1, (j & 15) >> 2);
          ew[j] = lut[(w >> (8 * (j & 3))) & 0xFFu];
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          if ((uint32_t)j < Km1) {
            const uint32_t e = ew[j];
            const bool pfx = i0 + (uint32_t)j < Km1;
            uint32_t dgt = ok ? ((e >> 4) & 3u) : 0u;
            uint32_t cf = pfx ? dgt : (e & 3u);
            uint32_t cr = pfx ? (3u - dgt) : ((e >> 2) & 3u);
            fw = (fw << 2) | cf;
            rc = (rc >> 2) | ((uint64_t)cr << rc_shift);
          }
        }
      }
      // ---- CHUNK hash steps, bases i0+K-1 .. ----
      // Per 16-byte group the 16 table look-ups are issued together and one group
      // ahead of their use (LDS answers in order, so they return before the LDS
      // traffic of the group in between).
      if (FAST && fast_k && __all(cnt == CHUNK)) {
        // Every lane of the wave has a full chunk: no per-step liveness at all.  Group g's 16 bytes sit
        // in the five dwords at qa + 4g at byte phase sh; they are loaded a group ahead, and the eight
        // table entries of a half group are reloaded into their own registers as soon as the half is
        // used up (eight steps before they are needed again), so nothing is copied at the loop's end.
        // The loop reads at most 40 bytes past the chunk's last base (NIQKI_SEQ_PAD).
        const uintptr_t a0 = (uintptr_t)(base + i0 + Km1);
        const uint32_t sh = (uint32_t)(a0 & 3u);
        const uint32_t *qa = (const uint32_t *)(a0 & ~(uintptr_t)3);
        uint64_t e[16];
        {
          const uint4 A = *(const uint4 *)qa;
          const uint32_t B = qa[4];
          uint4 w;
          w.x = __builtin_amdgcn_alignbyte(A.y, A.x, sh);
          w.y = __builtin_amdgcn_alignbyte(A.z, A.y, sh);
          w.z = __builtin_amdgcn_alignbyte(A.w, A.z, sh);
          w.w = __builtin_amdgcn_alignbyte(B, A.w, sh);
          lut64_16(w, lds0, e);
        }
        auto step = [&](uint64_t ent, bool check) {
          // :225-229 and :233-236 with the entry's pre-placed codes
          fw = shl2_64(fw);
          fw = (fw | (uint32_t)ent) & fw_mask;   // (only the high word's `and` is an instruction)
          rc = shr2_64(rc) | (ent & 0xFFFFFFFF00000000ULL);
          const uint64_t canon = fw < rc ? fw : rc;   // :345
          push_candidates(rev64_hi_mad(canon), thr, canon, top);
          // two steps add at most 128 to fewer than 64.  Unlikely: the drain is laid out behind the loop,
          // the common path has no taken branch
          if (check && __builtin_expect(top >= bottom + 512u, 0)) {
            drain64();
            if (top >= bottom + 512u) drain64();
          }
        };
        for (uint32_t g = 0; g < groups; ++g) {
          qa += 4;
          const uint4 A = *(const uint4 *)qa;   // the next group's bytes
          const uint32_t B = qa[4];
#pragma unroll
          for (int j = 0; j < 8; ++j) step(e[j], (j & 1) != 0);
          uint4 w;
          w.x = __builtin_amdgcn_alignbyte(A.y, A.x, sh);
          w.y = __builtin_amdgcn_alignbyte(A.z, A.y, sh);
          w.z = __builtin_amdgcn_alignbyte(A.w, A.z, sh);
          w.w = __builtin_amdgcn_alignbyte(B, A.w, sh);
          lut64_8<0>(w, lds0, e);
#pragma unroll
          for (int j = 8; j < 16; ++j) step(e[j], (j & 1) != 0);
          lut64_8<1>(w, lds0, e);
        }
        continue;
      }
      ByteStream bs;
      bs.open(base + i0 + Km1);
      // last dword-aligned address a 16-byte load may start at: the record's end plus the pad
      const uint32_t *qmax = (const uint32_t *)(((uintptr_t)(base + len) + NIQKI_SEQ_PAD - 16) & ~(uintptr_t)3);
      uint32_t en[16];
      {
        const uint4 v = bs.next16(qmax);
#pragma unroll
        for (int j = 0; j < 16; ++j) en[j] = lut[(dword_of(v, j >> 2) >> (8 * (j & 3))) & 0xFFu];
      }
      // filtered: all lanes of the wave run the groups its longest chunk needs (the stack is wave-collective);
      // no lane reads more than two groups past its own bases
      uint32_t cnt_wave = cnt;
      if (FILTER) {
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
          const uint32_t o = (uint32_t)__shfl_xor((int)cnt_wave, dd, 64);
          cnt_wave = o > cnt_wave ? o : cnt_wave;
        }
        cnt_wave = __builtin_amdgcn_readfirstlane(cnt_wave);
      }
      for (uint32_t g = 0; g < groups; ++g) {
        if (g * 16u >= cnt_wave) break;
        uint32_t e[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = en[j];
        if ((g + 1) * 16u < cnt_wave) {
          const uint4 v = bs.next16(qmax);
#pragma unroll
          for (int j = 0; j < 16; ++j) en[j] = lut[(dword_of(v, j >> 2) >> (8 * (j & 3))) & 0xFFu];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const uint64_t canon = roll_step<KFIX>(e[j], fw, rc, d, rc_shift);
          const bool live = g * 16u + (uint32_t)j < cnt;
          if (!FILTER) {
            sketch_update(canon, d, sk, live, hsel);
          } else {
            // dead steps get an all-ones hash word and never pass
            push_candidates(live ? rev64_hi(canon) : 0xFFFFFFFFu, thr, canon, top);
            if (top >= bottom + 512u) drain64();
          }
        }
      }
    }
  }
  if (FILTER && top != bottom) drain_rest();
}

// GROUPS 16-byte groups of hash steps per chunk: CHUNK = 16*GROUPS k-mers.
#ifdef NQ_SKETCH_CLOCK
__device__ unsigned long long nq_sketch_clk[2];   // shader cycles / 100 MHz ticks spent by one workgroup (tools/ubench_sketch.hip)
__device__ unsigned long long nq_sketch_trace[3 * 8192];   // per workgroup: start, end (100 MHz ticks), hardware id
#endif

template <int BLOCK, int GROUPS, int KFIX>
__global__ __launch_bounds__(BLOCK) void sketch_kernel(SketchArgs a) {
  extern __shared__ __align__(16) uint32_t smem[];
#ifdef NQ_SKETCH_CLOCK
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
#endif
  const Derived &d = a.d;
  const uint32_t Fc = d.F / a.halves;               // cells this workgroup keeps
  uint2 *lut64 = (uint2 *)(smem + kLut64Off / 4);   // 256 x {forward code, rc code << 28} (K = 31 filtered path)
  uint8_t *lut = (uint8_t *)(smem + kLutOff / 4);   // 256 bytes
  uint32_t *s_flag = smem + kFlagOff / 4;           // 4 words
  uint32_t *sk = smem + kCellOff / 4;               // Fc cells
  uint32_t *aux = sk + Fc;                          // distinct-value tables or the candidate stacks
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32_t *)smem;   // LDS address of the layout's start
  const uint32_t tid = threadIdx.x;
  const uint32_t part = blockIdx.x % a.splits;
  const uint32_t half = (blockIdx.x / a.splits) % a.halves;
  const uint32_t entry = blockIdx.x / (a.splits * a.halves);
  const uint32_t hsel = a.halves > 1 ? half + 1u : 0u;
  if (a.redo_only && a.redo[entry] != a.redo_only) return;   // (uniform) not this launch's sketch

  for (uint32_t i = tid; i < 256; i += BLOCK) {
    const uint32_t e = code_entry(i);
    lut[i] = (uint8_t)e;
    // rc code at bit 2K - 2 of the reverse word = bit 2K - 34 of the entry's high word (K >= 17; else unused)
    lut64[i] = make_uint2(e & 3u, d.K >= kFastKMin ? ((e >> 2) & 3u) << (2u * d.K - 34u) : 0u);
  }
  if (a.accumulate) {
    const uint32_t *src = (const uint32_t *)a.sketches + (uint64_t)entry * d.F + (uint64_t)half * Fc;
    for (uint32_t i = tid; i < Fc; i += BLOCK) sk[i] = src[i];
  } else {
    for (uint32_t i = tid; i < Fc; i += BLOCK) sk[i] = kEmpty32;
  }
  __syncthreads();

  if (a.seqs != nullptr) {  // nullptr = densify-only launch
    // Filter strength from the k-mers per slot: expected undecided slots
    // F*(1-2^-T)^(n/F) must be negligible (a miss only costs the exact re-run below).
    uint32_t thr = 0;
    if (a.filter) {
      uint64_t n = 0;
      const uint32_t r0 = a.entry_rec ? a.entry_rec[entry] : entry;
      const uint32_t r1 = a.entry_rec ? a.entry_rec[entry + 1] : entry + 1;
      for (uint32_t rec = r0; rec < r1; ++rec) {
        const uint64_t len = a.rec_off[rec + 1] - a.rec_off[rec];
        if (len > d.K) n += len - d.K;
      }
      // every part of a split record must be exact on its own share
      const uint64_t per_slot = (n / a.splits) >> d.S;
      uint32_t T = per_slot >= 240 ? 4u : per_slot >= 115 ? 3u : per_slot >= 55 ? 2u : 0u;
      if (a.filter >= 2) T = a.filter - 1;  // forced strength (tests)
      // the ordering argument needs a non-saturated HyperLogLog part: 2^H - 1 >= T
      if (T > d.max_rem || T > 16) T = 0;
      thr = T ? (1u << (32 - T)) : 0u;
    }
    thr = __builtin_amdgcn_readfirstlane(thr);
    if (thr) {
      // the generic filtered form also serves a layout that does not start at LDS address 0
      if (hsel || lds0 != 0) roll_records<BLOCK, GROUPS, KFIX, true, true, false>(a, entry, part, sk, lut, (uint64_t *)aux, thr, hsel);
      else roll_records<BLOCK, GROUPS, KFIX, true, false, true>(a, entry, part, sk, lut, (uint64_t *)aux, thr, 0);
      __syncthreads();
      uint32_t local = 0;
      for (uint32_t i = tid; i < Fc; i += BLOCK) local += (sk[i] == kEmpty32);
      if (tid == 0) s_flag[2] = 0;
      __syncthreads();
      if (local) atomicAdd(&s_flag[2], local);
      __syncthreads();
      // a slot without a candidate may still have skipped k-mers: exact re-run.
      // (With splits > 1 another part may hold the candidates, so every part re-runs.)
      if (s_flag[2] != 0) roll_records<BLOCK, GROUPS, KFIX, false, true, false>(a, entry, part, sk, lut, nullptr, 0, hsel);
    } else {
      roll_records<BLOCK, GROUPS, KFIX, false, true, false>(a, entry, part, sk, lut, nullptr, 0, hsel);
    }
  }
  __syncthreads();

  uint32_t *out = (uint32_t *)a.sketches + (uint64_t)entry * d.F + (uint64_t)half * Fc;
  if (a.splits > 1 || a.halves > 1) {
    // partial sketch (a part of a split record, or one half of the slots): merged in global
    // memory, densified by a second launch once all workgroups are in
    for (uint32_t i = tid; i < Fc; i += BLOCK) {
      uint32_t v = sk[i];
      if (v != kEmpty32) atomicMin(&out[i], v);
    }
    return;
  }
  if (a.densify) {
    if (a.distinct) densify_lds_distinct<BLOCK>(sk, d, s_flag, aux);
    else densify_lds<BLOCK>(sk, d, s_flag);
  }
  __syncthreads();
  for (uint32_t i = tid; i < d.F; i += BLOCK) out[i] = sk[i];
#ifdef NQ_SKETCH_CLOCK
  if (blockIdx.x == gridDim.x / 2 && tid == 0) {
    nq_sketch_clk[0] = __builtin_readcyclecounter() - clk_c0;
    nq_sketch_clk[1] = wall_clock64() - clk_r0;
  }
  if (tid == 0 && blockIdx.x < 8192) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    nq_sketch_trace[3 * blockIdx.x] = clk_r0;
    nq_sketch_trace[3 * blockIdx.x + 1] = wall_clock64();
    nq_sketch_trace[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}


// ---- short records: one wavefront per sketch -----------------------------------------
// A 150-base read holds ~120 k-mers but its densification takes ~350 passes
// (SURVEY.md 8a a8), so the short-record kernel is built around the passes: one wave
// owns one sketch, nothing is synchronised across waves, and a pass costs one LDS round
// trip.  Each occupied cell of the sketch becomes an ENTRY held in registers
// (up to kReadEntries per lane): T = running low word of hash_family(v, step) (the target
// is T mod F and T += B per pass, B = low word of revhash64(v), src/niqki_index.cpp:308-310),
// mi = smallest cell index known to hold v.  A pass: every entry proposes
// 2^31 | mi to its target cell with an LDS min (occupied cells hold values below 2^31
// and stay as they are, the smallest source index wins = the first writer of the
// reference's ascending loop, :313-331); then every entry reads its target back, and the
// one that finds its own proposal writes the value and lowers its mi.  Copies of a value
// propose the same target, so the entries present at the start are all that ever
// matter; two entries with the same value are harmless (the lower index always wins).
constexpr uint32_t kReadEntries = 6;                  // entries per lane (384: reads up to ~400 bases)
constexpr uint32_t kReadMaxEntries = 64 * kReadEntries;
constexpr uint32_t kReadTile = 512;                   // positions coded per tile

// entry list capacity by the records' average length: a 150-base read fills ~120 cells, and with 192 entries a
// sketch of 2^12 cells leaves room for nine wavefronts per CU instead of eight (a record with more occupied cells
// than the list holds takes the plain pass over all cells)
static uint32_t sketch_reads_entries(uint64_t avg_len) { return avg_len <= 200 ? 192u : kReadMaxEntries; }
static bool sketch_reads_shape(const Derived &d, uint64_t avg_len, uint32_t splits, uint32_t halves);
static size_t sketch_reads_lds_bytes(const Derived &d, uint32_t entries = kReadMaxEntries) {
  // sketch cells + one region that holds the position codes and the code table while the k-mers are hashed, the
  // entry list (cell, value) afterwards, the closed-form tail's two lists at the end
  const size_t region = std::max<size_t>((size_t)entries * 8, kReadTile + 256 + 64);
  return (size_t)d.F * 4 + region;
}

// plain pass over all cells (more occupied cells than register entries: long records,
// few passes)
__device__ void densify_wave_cells(uint32_t *sk, const Derived &d, uint32_t empty) {
  const uint32_t F = d.F, lane = threadIdx.x;
  uint32_t step = 0, idle = 0;
  for (;;) {
    for (uint32_t i = lane; i < F; i += 64) {
      const uint32_t v = sk[i];
      if (v < 0x80000000u) {
        const uint32_t t = ((uint32_t)unrev64(v) + step * (uint32_t)rev64(v)) & (F - 1u);
        atomicMin(&sk[t], 0x80000000u | i);
      }
    }
    __syncthreads();
    uint32_t filled = 0;
    for (uint32_t i = lane; i < F; i += 64) {
      const uint32_t v = sk[i];
      if (v >= 0x80000000u && v != kEmpty32) { sk[i] = sk[v & 0x7FFFFFFFu]; ++filled; }
    }
    __syncthreads();
    uint32_t tot = filled;
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) tot += __shfl_xor(tot, dd, 64);
    empty -= tot;
    ++step;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
  }
}


// all LDS traffic of the wave issued so far has completed, and the compiler keeps the
// accesses on either side apart (one wave: the LDS serves its instructions in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Orders the wave's LDS instructions on either side in the instruction stream without waiting for
// them: the LDS serves one wave's instructions in issue order, so a later instruction of ANY lane
// sees the effect of an earlier one of any lane.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Sum of a per-lane count over the wave (the usual DPP ladder: shifts inside the rows of 16 lanes, then the rows' last
// lanes broadcast onward; lane 63 ends up with the total).  No LDS instruction, unlike a __shfl reduction.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, true);    // row_shr:4, lanes 4..15 of a row
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, true);    // row_shr:8, lanes 8..15 of a row
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Minimum of a per-lane word over the wave (the same DPP ladder as wave_sum_u32; lanes the ladder leaves out bring
// the identity 2^32 - 1).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#define NQ_DPP_MIN(CTRL, ROWS, BANKS)                                                                      \
  {                                                                                                        \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROWS, BANKS, false);        \
    v = o < v ? o : v;                                                                                     \
  }
  NQ_DPP_MIN(0x111, 0xf, 0xf)   // row_shr:1
  NQ_DPP_MIN(0x112, 0xf, 0xf)   // row_shr:2
  NQ_DPP_MIN(0x114, 0xf, 0xe)   // row_shr:4, lanes 4..15 of a row
  NQ_DPP_MIN(0x118, 0xf, 0xc)   // row_shr:8, lanes 8..15 of a row
  NQ_DPP_MIN(0x142, 0xa, 0xf)   // row_bcast:15 into rows 1 and 3
  NQ_DPP_MIN(0x143, 0xc, 0xf)   // row_bcast:31 into rows 2 and 3
#undef NQ_DPP_MIN
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- the last cells of a short record's densification in closed form --------------------------------------------
// Going from E empty cells to E - 1 takes F / (n E) passes (coupon collector; more where only few orbits reach a cell),
// so more than half of a 150-base read's ~420 passes fill its last 16 cells, and a pass costs its LDS instructions
// whatever it fills.  Once at most kTailCells cells are empty the passes stop.  An entry's target is affine in the
// pass (src/niqki_index.cpp:308-310): T_k + s B_k = c (mod F) with B_k = 2^j odd has the solution
//     s_k(c) = ((c - T_k) >> j) odd^-1  mod (F >> j)      if 2^j divides c - T_k, none otherwise
// (odd^-1 by Newton steps, once per entry).  Cell c is filled in pass s* = min_k s_k(c) by the entry that reaches it
// then -- among several, by the one whose value sits at the smallest cell index at the start of that pass (:313-331:
// the first writer of the ascending loop).  The lanes keep their entries; the cells are taken one after another:
// every lane solves for its own entries, one wave-wide minimum of (s << 15 | index at the start of the tail) names
// pass and winner.  An index can still drop during the tail (an entry wins a cell below its own), which matters only
// to a cell that several entries reach in its pass s*, and only if one of the LOSING ones has won some cell in an
// earlier pass of the tail: that case (a few reads in a hundred) is detected and handed back to the passes, which
// are exact for anything -- nothing is written before the check.  Cells no orbit reaches stay empty, as after the F
// fruitless passes that end the loop (documented divergences).  Pinned by the goldens, tests/test_gpu_fuzz.py,
// tests/test_gpu_reads_config.py.
constexpr uint32_t kTailCells = 16;   // (8 .. 24 measure the same; 48: 3 % slower)

// T: every entry's target of the NEXT pass, B its stride, mk = 2^31 | the smallest index that holds its value, V its
// value (lanes without an entry: an occupied cell as T, B = 0 -- they reach no empty cell).  scratch: 2 x 64 words.
// true: every reachable empty cell is filled; false: nothing was touched, the passes go on.
template <int R>
__device__ __forceinline__ bool densify_tail(uint32_t *sk, uint32_t *scratch, uint32_t F, const uint32_t (&T)[R],
                                             const uint32_t (&B)[R], const uint32_t (&mk)[R], const uint32_t (&V)[R]) {
  const uint32_t lane = threadIdx.x, Fm = F - 1u;
  uint32_t *list = scratch, *vals = scratch + 64;
  // ---- the empty cells (F is a multiple of 256 on this path) ----
  uint32_t n_e = 0;
  vals[lane] = kEmpty32;
  for (uint32_t c0 = 0; c0 < F; c0 += 256) {
    const uint4 v = *(const uint4 *)(sk + c0 + 4 * lane);
    const uint32_t m01 = v.x > v.y ? v.x : v.y, m23 = v.z > v.w ? v.z : v.w;
    if (__any((m01 > m23 ? m01 : m23) == kEmpty32)) {   // (wave uniform, rare: the sketch is nearly full)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool e = w[i] == kEmpty32;
        const uint64_t bal = __builtin_amdgcn_ballot_w64(e);
        const uint32_t at = n_e + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (e && at < 64u) list[at] = c0 + 4 * lane + (uint32_t)i;
        n_e += (uint32_t)__builtin_popcountll(bal);
      }
    }
  }
  wave_lds_sync();
  if (n_e > 64u) return false;   // (cannot happen: the caller counted)
  const uint32_t cells = lane < n_e ? list[lane] : 0u;
  // ---- the entries' orbits: stride = 2^j odd; B = 0 (mod F): the orbit is the one cell T (j = log2 F, s = 0) ----
  uint32_t low[R], jsh[R], inv[R], win_min[R], lose_max[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t j = (uint32_t)__builtin_ctz((B[k] & Fm) | F);
    const uint32_t odd = (B[k] >> j) | 1u;
    uint32_t x = odd;                       // correct to 3 bits; every step doubles them
#pragma unroll
    for (int i = 0; i < 3; ++i) x *= 2u - odd * x;
    low[k] = (1u << j) - 1u;
    jsh[k] = j;
    inv[k] = x;
    win_min[k] = 0xFFFFFFFFu;
    lose_max[k] = 0u;
  }
  // one cell: its keys per entry (s << 15 | index; all ones: the orbit never reaches it) and the lane's smallest
  auto solve = [&](uint32_t c, uint32_t (&key)[R]) -> uint32_t {
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint32_t x = c - T[k];
      const uint32_t s = ((x * inv[k]) & Fm) >> jsh[k];   // valid when 2^j divides x
      key[k] = (x & low[k]) == 0u ? ((s << 15) | (mk[k] & 0x7FFFu)) : 0xFFFFFFFFu;
      best = key[k] < best ? key[k] : best;
    }
    return best;
  };
  // ... and what follows from the wave's smallest key: the winner's value, the passes of wins and of lost ties
  auto settle = [&](uint32_t ci, uint32_t wkey, const uint32_t (&key)[R]) {
    if (wkey == 0xFFFFFFFFu) return;        // no orbit reaches the cell: it stays empty
    const uint32_t s_star = wkey >> 15, top = wkey | 0x7FFFu;
    uint32_t n_tied = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      n_tied += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(key[k] <= top));
      if (key[k] == wkey) {
        vals[ci] = V[k];
        win_min[k] = s_star < win_min[k] ? s_star : win_min[k];
      }
    }
    if (n_tied > 1u) {   // (wave uniform, rare) the entries that reach the cell in its pass and lose it
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (key[k] <= top && key[k] != wkey) lose_max[k] = s_star > lose_max[k] ? s_star : lose_max[k];
    }
  };
  for (uint32_t ci = 0; ci < n_e; ++ci) {   // (wave uniform; two cells per round are no faster)
    uint32_t key[R];
    const uint32_t best = solve((uint32_t)__builtin_amdgcn_readlane((int)cells, (int)ci), key);
    settle(ci, wave_min_u32(best), key);
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < R; ++k) bad |= win_min[k] < lose_max[k];
  if (__any(bad)) return false;
  wave_lds_sync();
  if (lane < n_e) {
    const uint32_t v = vals[lane];
    if (v != kEmpty32) sk[cells] = v;
  }
  wave_lds_sync();
  return true;
}

// The passes over R register entries per lane (entry e = k*64 + lane of elist), targets read a window ahead.
//
// A pass of the reference's loop (src/niqki_index.cpp:313-331) changes a cell only where an entry's target is EMPTY,
// and in the coupon-collector tail (half of a read's ~420 passes run with fewer than 256 of the 4096 cells empty)
// almost no target is.  So the targets of the next U passes are read first -- U x R plain LDS reads per lane in ONE
// round trip -- and a pass then costs LDS traffic only for the entries whose target was empty in that window read:
// they alone propose (ds_min), read back and, where they won, write; a pass without such an entry anywhere in the
// wave costs one ballot.  The window read may be stale for the later passes of its window in one direction only: a
// cell it saw empty may have been filled by an earlier pass of the window (cells never go back to empty); the
// proposal that follows is then a ds_min on an occupied cell, which changes nothing, and its read-back is not the
// proposer's own word.  Exactly the passes of the plain loop below; 1.05 x its speed on 150- and 300-base reads
// (profiles/r06_densify_forms.txt, form a: the passes are paid in LDS instructions, and this form issues fewer only
// where no lane proposes).
template <int R, int U>
__device__ __forceinline__ void densify_wave_entries_window(uint32_t *sk, const uint32_t *elist, uint32_t n_ent,
                                                            uint32_t F, uint32_t empty, uint32_t *scratch, bool tail) {
  const uint32_t lane = threadIdx.x, Fm = F - 1u;
  uint32_t T[R], B[R], mk[R], V[R];
  // A lane without an entry in round k watches a cell that is occupied from the start (entry 0's own) and never
  // moves on (B = 0): its window reads never show "empty", so it never proposes and the passes need no validity test.
  const uint32_t cell0 = elist[0];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t e = (uint32_t)k * 64u + lane;
    const bool valid = e < n_ent;
    const uint32_t v = valid ? elist[2 * e + 1] : 0u;
    V[k] = v;
    mk[k] = valid ? (0x80000000u | elist[2 * e]) : kEmpty32;
    T[k] = valid ? (uint32_t)unrev64(v) : cell0;
    B[k] = valid ? (uint32_t)rev64(v) : 0u;
  }
  uint32_t idle = 0;
  for (;;) {
    uint32_t pre[U][R];
    {
      uint32_t t[R];
#pragma unroll
      for (int k = 0; k < R; ++k) t[k] = T[k];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < R; ++k) {
          pre[u][k] = sk[t[k] & Fm];
          t[k] += B[k];
        }
    }
    wave_lds_order();
    // The exit tests run once per window: a pass behind the one that filled the last cell finds no empty target
    // (or a stale "empty", whose proposal changes nothing), and fruitless passes beyond the F that prove a fixpoint
    // change nothing either -- the cells are those of the loop that stops at the pass itself.
    uint32_t wins = 0;   // per lane; summed over the wave once per window
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bool prop[R], any = false;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        prop[k] = pre[u][k] == kEmpty32;
        any |= prop[k];
      }
      if (__any(any)) {   // wave uniform
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (prop[k]) atomicMin(&sk[T[k] & Fm], mk[k]);
        wave_lds_order();
        // all proposals of the wave are issued before any read-back: a read sees the surviving proposal of its cell
        // (which names exactly one entry: the markers of two entries never agree), or a value -- another pass's or,
        // behind its winner's write, this pass's.  (Read by all lanes: the LDS is paid per instruction, and a mask
        // around it costs two scalar instructions.)
        uint32_t back[R];
#pragma unroll
        for (int k = 0; k < R; ++k) back[k] = sk[T[k] & Fm];
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const uint32_t t = T[k] & Fm;
          const bool won = prop[k] && back[k] == mk[k];
          if (won) {
            sk[t] = V[k];
            const uint32_t m = 0x80000000u | t;
            mk[k] = m < mk[k] ? m : mk[k];
            ++wins;
          }
        }
        wave_lds_order();
      }
#pragma unroll
      for (int k = 0; k < R; ++k) T[k] += B[k];
    }
    const uint32_t tot_w = wave_sum_u32(wins);
    empty -= tot_w;
    idle = tot_w ? 0u : idle + (uint32_t)U;
    if (empty == 0 || idle >= F) return;
    if (tail && empty <= kTailCells) {   // (wave uniform) the last cells in closed form; once
      if (densify_tail<R>(sk, scratch, F, T, B, mk, V)) return;
      tail = false;
    }
  }
}

// The plain form (NIQKI_DENSIFY_WINDOW=0: the A/B figure in profiles/): every entry proposes in every pass.
template <int R>
__device__ __forceinline__ void densify_wave_entries(uint32_t *sk, const uint32_t *elist, uint32_t n_ent,
                                                     uint32_t F, uint32_t empty) {
  const uint32_t lane = threadIdx.x, Fm = F - 1u;
  uint32_t T[R], B[R], mk[R], V[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t e = (uint32_t)k * 64u + lane;
    const bool valid = e < n_ent;
    const uint32_t v = valid ? elist[2 * e + 1] : 0u;
    V[k] = v;
    mk[k] = valid ? (0x80000000u | elist[2 * e]) : kEmpty32;  // a min with "empty" changes nothing
    T[k] = (uint32_t)unrev64(v);
    B[k] = (uint32_t)rev64(v);
  }
  uint32_t idle = 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (mk[k] != kEmpty32) atomicMin(&sk[T[k] & Fm], mk[k]);   // (lanes without an entry stay out of the LDS)
    wave_lds_order();
    // all proposals of the wave are issued before any read-back: a read sees the surviving proposal
    // of its cell (which names exactly one entry), or, behind another entry's winner write of this
    // pass, that winner's value -- not its own proposal either way.  One LDS round trip per pass:
    // only the read-back is waited for; the winner writes and the next pass's proposals follow in order.
    uint32_t back[R];
#pragma unroll
    for (int k = 0; k < R; ++k) back[k] = sk[T[k] & Fm];   // (unconditional: masking these reads costs 30 %)
    wave_lds_order();
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const uint32_t t = T[k] & Fm;
      const bool won = back[k] == mk[k] && mk[k] != kEmpty32;
      if (won) {
        sk[t] = V[k];
        const uint32_t m = 0x80000000u | t;
        mk[k] = m < mk[k] ? m : mk[k];
      }
      tot += (uint32_t)__popcll(__ballot(won));
      T[k] += B[k];
    }
    wave_lds_order();
    empty -= tot;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
  }
}

__global__ __launch_bounds__(64) void sketch_reads_kernel(SketchArgs a) {
  extern __shared__ __align__(16) uint32_t smem[];
  const Derived &d = a.d;
  const uint32_t F = d.F, lane = threadIdx.x;
  uint32_t *sk = smem;                                       // F cells
  // one region behind the cells: position codes + code table while the k-mers are hashed, then the entry list
  uint32_t *elist = smem + F;                                // a.read_entries x {cell, value}
  uint8_t *codes = (uint8_t *)elist;                         // kReadTile position codes
  uint8_t *lut = codes + kReadTile;                          // 256-byte code table
  const uint32_t cap = a.read_entries;
  const uint32_t entry = blockIdx.x;
  const uint32_t K = d.K, Km1 = d.K - 1u;
  if (a.redo_only && a.redo[entry] != a.redo_only) return;   // (one wave: uniform) not this launch's sketch

  for (uint32_t i = lane; i < 256; i += 64) lut[i] = code_entry(i);
  // (the kernel's time is its LDS instructions: the passes over all cells move 16 bytes per lane; F is a multiple
  // of 256 on this path -- S >= 8 -- or the loops below fall back to single cells)
  const bool quads = (F & 255u) == 0u && ((uintptr_t)a.sketches & 15u) == 0u;   // (the caller's rows: 16-byte aligned as a rule)
  if (a.accumulate) {
    const uint32_t *src = (const uint32_t *)a.sketches + (uint64_t)entry * F;
    if (quads) for (uint32_t i = 4 * lane; i < F; i += 256) *(uint4 *)(sk + i) = *(const uint4 *)(src + i);
    else for (uint32_t i = lane; i < F; i += 64) sk[i] = src[i];
  } else {
    if (quads) for (uint32_t i = 4 * lane; i < F; i += 256) *(uint4 *)(sk + i) = make_uint4(kEmpty32, kEmpty32, kEmpty32, kEmpty32);
    else for (uint32_t i = lane; i < F; i += 64) sk[i] = kEmpty32;
  }
  __syncthreads();

  // ---- k-mers: positions are coded tile by tile (forward code | rc code << 2 per position,
  // "positional codes" of DESIGN.md), every lane then assembles whole k-mers from K codes ----
  if (a.seqs != nullptr) {
    const uint32_t r0 = a.entry_rec ? a.entry_rec[entry] : entry;
    const uint32_t r1 = a.entry_rec ? a.entry_rec[entry + 1] : entry + 1;
    for (uint32_t rec = r0; rec < r1; ++rec) {
      const uint64_t b0 = a.rec_off[rec], b1 = a.rec_off[rec + 1];
      const uint64_t len = b1 - b0;
      if (len <= K) continue;                 // src/niqki_index.cpp:395,:450
      const uint64_t n_kmers = len - K;       // last k-mer skipped, :342
      const uint8_t *base = a.seqs + b0;
      // the K-1 prefix digits are zeroed together when one of them is no base (:255-273)
      uint32_t ok = 1;
      if (lane < Km1) ok = (lut[base[lane]] >> 6) & 1u;
      ok = __all(ok) ? 1u : 0u;
      const uint32_t per_tile = kReadTile - Km1;   // k-mers served by one tile of positions
      for (uint64_t k0 = 0; k0 < n_kmers; k0 += per_tile) {
        __syncthreads();  // the previous tile's codes are no longer read
        for (uint32_t j = lane; j < kReadTile; j += 64) {
          const uint64_t pos = k0 + j;
          uint32_t c = 0;
          if (pos < len) {
            const uint32_t e = lut[base[pos]];
            if (pos < Km1) {
              const uint32_t dgt = ok ? ((e >> 4) & 3u) : 0u;
              c = dgt | ((3u - dgt) << 2);      // digit and its complement (rcb, :240-250)
            } else {
              c = e & 15u;                       // rolling tables (:114-123, :211-221)
            }
          }
          codes[j] = (uint8_t)c;
        }
        __syncthreads();
        const uint64_t left = n_kmers - k0;
        const uint32_t n_here = left < per_tile ? (uint32_t)left : per_tile;
        for (uint32_t q = lane; q < ((n_here + 63u) & ~63u); q += 64) {
          const bool live = q < n_here;
          uint64_t fw = 0, rc = 0;
          if (live) {
            for (uint32_t j = 0; j < K; ++j) {
              const uint32_t c = codes[q + j];
              fw = (fw << 2) | (c & 3u);
              rc |= (uint64_t)(c >> 2) << (2u * j);
            }
          }
          const uint64_t canon = fw < rc ? fw : rc;   // :345
          sketch_update(canon, d, sk, live);
        }
      }
    }
  }
  __syncthreads();

  uint32_t *out = (uint32_t *)a.sketches + (uint64_t)entry * F;
  if (a.densify) {
    // ---- entries: the occupied cells, dealt to the lanes (any order: an entry only knows its own cell and value) ----
    uint32_t n_ent = 0;
    if (quads) {
      for (uint32_t c0 = 0; c0 < F; c0 += 256) {
        const uint4 q4 = *(const uint4 *)(sk + c0 + 4 * lane);
        const uint32_t w[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint64_t bal = __ballot(w[i] != kEmpty32);
          if (bal) {   // (wave uniform: a read's sketch is nearly empty here)
            const uint32_t at = n_ent + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (w[i] != kEmpty32 && at < cap) { elist[2 * at] = c0 + 4 * lane + (uint32_t)i; elist[2 * at + 1] = w[i]; }
            n_ent += (uint32_t)__popcll(bal);
          }
        }
      }
    } else {
      for (uint32_t c0 = 0; c0 < F; c0 += 64) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < F ? sk[c] : kEmpty32;
        const uint64_t bal = __ballot(v != kEmpty32);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        const uint32_t at = n_ent + rank;
        if (v != kEmpty32 && at < cap) { elist[2 * at] = c; elist[2 * at + 1] = v; }
        n_ent += (uint32_t)__popcll(bal);
      }
    }
    __syncthreads();
    const uint32_t empty0 = F - n_ent;
    if (empty0 != 0 && n_ent != 0) {
      if (n_ent > cap && a.redo_mark) {
        // more occupied cells than the entry list holds (a record far longer than the batch's average): a later
        // launch -- the long list, then the workgroup kernel -- sketches it again; the plain pass over all cells
        // would take fifty times a read's time.  Nothing of this sketch is stored here.
        if (lane == 0) a.redo[entry] = a.redo_mark;
        return;
      }
      if (n_ent > cap) {
        densify_wave_cells(sk, d, empty0);
      } else {
        const uint32_t rounds = (n_ent + 63u) >> 6;   // wave uniform
        if (a.window) {
          // (the entries are in registers by then: their list's space holds the tail's two lists)
          uint32_t *scratch = elist;
          static_assert(192 * 8 >= 2 * 64 * 4, "the tail's lists live in the entry list's space");
          const bool tail = a.window >= 2 && quads;
          if (rounds <= 1) densify_wave_entries_window<1, 8>(sk, elist, n_ent, F, empty0, scratch, tail);
          else if (rounds <= 2) densify_wave_entries_window<2, 8>(sk, elist, n_ent, F, empty0, scratch, tail);
          else if (rounds <= 3) densify_wave_entries_window<3, 4>(sk, elist, n_ent, F, empty0, scratch, tail);
          else if (rounds <= 4) densify_wave_entries_window<4, 4>(sk, elist, n_ent, F, empty0, scratch, tail);
          else densify_wave_entries_window<6, 4>(sk, elist, n_ent, F, empty0, scratch, tail);
        }
        else if (rounds <= 1) densify_wave_entries<1>(sk, elist, n_ent, F, empty0);
        else if (rounds <= 2) densify_wave_entries<2>(sk, elist, n_ent, F, empty0);
        else if (rounds <= 3) densify_wave_entries<3>(sk, elist, n_ent, F, empty0);
        else if (rounds <= 4) densify_wave_entries<4>(sk, elist, n_ent, F, empty0);
        else densify_wave_entries<6>(sk, elist, n_ent, F, empty0);
        static_assert(kReadEntries == 6, "dispatch above covers 1..6 rounds");
      }
    }
    __syncthreads();
  }
  if (quads) for (uint32_t i = 4 * lane; i < F; i += 256) *(uint4 *)(out + i) = *(const uint4 *)(sk + i);
  else for (uint32_t i = lane; i < F; i += 64) out[i] = sk[i];
}

static size_t sketch_lds_bytes(const Derived &d, bool distinct, uint32_t ring_waves, uint32_t halves = 1) {
  return (size_t)(d.F / halves) * 4 + 16 + 256 + 2048 + (distinct ? (size_t)d.R * 12 : 0) + (size_t)ring_waves * kStack * 8;
}
constexpr size_t kLdsLimit = 160 * 1024;

bool sketch_needs_merge(const Derived &d) { return sketch_lds_bytes(d, false, 16) > kLdsLimit; }

// Densification of sketches whose F cells do not fit LDS (S = 16): the same pass-parallel algorithm as
// Distinct-value densification as a launch of its own, for sketches whose cells leave no room for the three tables
// of densify_lds_distinct beside them (the reference's defaults, S = 15 W = 12: 128 KB of cells, 48 KB of tables): a
// thread's values are always the same ones (v = tid + k 1024, k < 4: 2^W <= 4096), so their two hash words live in its
// registers and the LDS holds the cells and mi[v] alone.  The sketch kernel stores the cells as they are (densify = 0),
// this kernel takes them from there.  Records of 0.6 .. 15 kbp at S = 15 W = 12: see profiles/r06_densify_lean_and_tail.txt.
constexpr int kLateValues = 4;
__global__ __launch_bounds__(1024) void densify_distinct_kernel(SketchArgs a) {
  extern __shared__ __align__(16) uint32_t smem[];
  const Derived &d = a.d;
  const uint32_t F = d.F, R = d.R, W = d.W, tid = threadIdx.x;
  const uint32_t entry = blockIdx.x;
  if (a.redo_only && a.redo[entry] != a.redo_only) return;   // (uniform) not this launch's sketch
  uint32_t *sk = smem, *mi = smem + F, *s_flag = mi + R;
  uint32_t *row = (uint32_t *)a.sketches + (uint64_t)entry * F;
  uint32_t local = 0;
  for (uint32_t i = tid; i < F; i += 1024) {
    const uint32_t v = row[i];
    sk[i] = v;
    local += (v == kEmpty32);
  }
  for (uint32_t v = tid; v < R; v += 1024) mi[v] = kEmpty32;
  if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
  __syncthreads();
  if (local) atomicAdd(&s_flag[0], local);
  for (uint32_t i = tid; i < F; i += 1024) {
    const uint32_t v = sk[i];
    if (v != kEmpty32) atomicMin(&mi[v], i);
  }
  __syncthreads();
  uint32_t empty = s_flag[0];
  if (empty == 0 || empty == F) return;   // (nothing to fill, or nothing to fill from: the row stays as it is)
  uint32_t ra[kLateValues], rb[kLateValues];
#pragma unroll
  for (int k = 0; k < kLateValues; ++k) {
    const uint32_t v = tid + (uint32_t)k * 1024u;
    ra[k] = (uint32_t)unrev64(v);
    rb[k] = (uint32_t)rev64(v);
  }
  uint32_t step = 0, idle = 0;
  while (true) {
    uint32_t mk[kLateValues], filled = 0;   // this pass's proposals (all ones: none)
#pragma unroll
    for (int k = 0; k < kLateValues; ++k) {
      const uint32_t v = tid + (uint32_t)k * 1024u;
      const uint32_t m = v < R ? mi[v] : kEmpty32;
      const uint32_t t = (ra[k] + step * rb[k]) & (F - 1u);  // hash_family(v, step) % F, src/niqki_index.cpp:308-310,:319
      const bool prop = m != kEmpty32 && sk[t] >= 0x80000000u;
      mk[k] = prop ? (0x80000000u | (m << W) | v) : kEmpty32;
      if (prop) atomicMin(&sk[t], mk[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kLateValues; ++k) {
      const uint32_t v = tid + (uint32_t)k * 1024u;
      const uint32_t t = (ra[k] + step * rb[k]) & (F - 1u);
      const bool won = mk[k] != kEmpty32 && sk[t] == mk[k];
      if (won) {
        sk[t] = v;
        if (t < ((mk[k] & 0x7FFFFFFFu) >> W)) mi[v] = t;
      }
      filled += won ? 1u : 0u;
    }
    if (filled) atomicAdd(&s_flag[1], filled);
    __syncthreads();
    const uint32_t tot = s_flag[1];  // every proposed-to cell now holds its winner's value
    __syncthreads();
    if (tid == 0) s_flag[1] = 0;
    empty -= tot;
    ++step;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t i = tid; i < F; i += 1024) row[i] = sk[i];
}

// densify_lds on the cells in global memory.  One workgroup per sketch; the cells are read and
// written with agent-scope atomics only (a plain load could see a stale L1 line behind another
// wave's atomicMin).  Rare path: a 5 Mbp genome has no empty cell at S = 16 and leaves after the count.
__global__ __launch_bounds__(1024) void densify_global_kernel(SketchArgs a) {
  __shared__ uint32_t s_flag[2];
  const Derived &d = a.d;
  const uint32_t F = d.F, tid = threadIdx.x;
  uint32_t *sk = (uint32_t *)a.sketches + (uint64_t)blockIdx.x * F;
  auto ld = [&](uint32_t i) { return __hip_atomic_load(&sk[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st = [&](uint32_t i, uint32_t v) { __hip_atomic_store(&sk[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  uint32_t local = 0;
  for (uint32_t i = tid; i < F; i += 1024) local += (ld(i) == kEmpty32);
  if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
  __syncthreads();
  if (local) atomicAdd(&s_flag[0], local);
  __syncthreads();
  uint32_t empty = s_flag[0];
  if (empty == 0 || empty == F) return;
  uint32_t step = 0, idle = 0;
  while (true) {
    for (uint32_t i = tid; i < F; i += 1024) {
      const uint32_t v = ld(i);
      if (v < 0x80000000u) {
        const uint32_t t = ((uint32_t)unrev64(v) + step * (uint32_t)rev64(v)) & (F - 1u);  // :308-310, :319
        atomicMin(&sk[t], 0x80000000u | i);
      }
    }
    __threadfence();
    __syncthreads();
    uint32_t filled = 0;
    for (uint32_t i = tid; i < F; i += 1024) {
      const uint32_t v = ld(i);
      if (v >= 0x80000000u && v != kEmpty32) { st(i, ld(v & 0x7FFFFFFFu)); ++filled; }
    }
    if (filled) atomicAdd(&s_flag[1], filled);
    __threadfence();
    __syncthreads();
    const uint32_t tot = s_flag[1];
    __syncthreads();
    if (tid == 0) s_flag[1] = 0;
    empty -= tot;
    ++step;
    idle = tot ? 0u : idle + 1u;
    if (empty == 0 || idle >= F) break;
    __syncthreads();
  }
}

// reads: one wavefront per sketch when the sketch and its entry list leave room for several waves per CU
// (NIQKI_SKETCH_WAVE=0 switches this shape off)
static bool sketch_reads_shape(const Derived &d, uint64_t avg_len, uint32_t splits, uint32_t halves) {
  const char *wv = std::getenv("NIQKI_SKETCH_WAVE");
  // (records of up to 384 + K bases: their occupied cells fit the 384-entry list.  Longer ones would all take the plain
  // pass over all cells -- 21 ms instead of the workgroup kernel's 4.9 per 16 384 records of 500 bases)
  return avg_len <= 415 && splits == 1 && halves == 1 && sketch_reads_lds_bytes(d, sketch_reads_entries(avg_len)) <= 160 * 1024 &&
         !(wv && std::atoi(wv) == 0);
}
uint32_t sketch_read_list(const Derived &d, uint64_t avg_len) {
  return sketch_reads_shape(d, avg_len, 1, sketch_needs_merge(d) ? 2u : 1u) ? sketch_reads_entries(avg_len) : 0u;
}

hipError_t launch_sketch(const SketchArgs &a_in, uint32_t n_entry, uint64_t avg_len,
                         hipStream_t stream) {
  if (n_entry == 0) return hipSuccess;
  SketchArgs a = a_in;
  a.halves = sketch_needs_merge(a.d) ? 2u : 1u;
  if (a.halves > 1) {
    if (a.seqs == nullptr) {  // densify-only launch
      hipLaunchKernelGGL(densify_global_kernel, dim3(n_entry), dim3(1024), 0, stream, a);
      return hipGetLastError();
    }
    if (a.densify) return hipErrorInvalidValue;  // see sketch_needs_merge
  }
  // Launch shape by the average input length of a sketch: a 256-thread workgroup for
  // reads, else 1024 threads with chunks of 32 / 128 / 512 k-mers per lane (long chunks
  // amortise the K-1 warm-up steps, short ones keep all lanes busy on short records).
  // reads: one wavefront per sketch when the sketch and its entry list leave room for
  // several waves per CU (NIQKI_SKETCH_WAVE=0 switches this shape off)
  {
    a.read_entries = sketch_reads_entries(avg_len);
    const size_t wl = sketch_reads_lds_bytes(a.d, a.read_entries);
    if (sketch_reads_shape(a.d, avg_len, a.splits, a.halves)) {
      const char *dw = std::getenv("NIQKI_DENSIFY_WINDOW");   // 0: every entry proposes in every pass (measurement)
      const char *dt = std::getenv("NIQKI_DENSIFY_TAIL");     // 0: passes to the end, no closed-form tail (measurement)
      a.window = (dw && std::atoi(dw) == 0) ? 0u : (dt && std::atoi(dt) == 0) ? 1u : 2u;
      hipError_t e = hipFuncSetAttribute((const void *)sketch_reads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(sketch_reads_kernel, dim3(n_entry), dim3(64), wl, stream, a);
      return hipGetLastError();
    }
  }
  const bool short_records = avg_len < 16384;
  // distinct-value densification
  // after niqki_select_best_H the fingerprint parts may overlap and leave [0, 2^W): the
  // value-indexed tables and the filter's ordering argument need the regular form
  const bool regular = a.d.mask_m == (1u << a.d.M) - 1u && a.d.max_rem == (1u << a.d.H) - 1u;
  // ... and cells this launch produced itself: a caller's sketch (niqki_densify, accumulate) may hold
  // any value, the value-indexed tables only values below 2^W
  const bool own_cells = a.seqs != nullptr && !a.accumulate;
  // ... and where the values are few against the cells (a pass walks 2^W values instead of 2^S cells, but pays three table
  // words per value: at 2^W = 2^S the plain passes are 1.6 x faster, at 2^S = 4 x 2^W the distinct ones 1.5 x, at 32 x
  // seventeen times -- profiles/r06_densify_lean_and_tail.txt 7) and the tables fit the CU's LDS beside the cells
  // -- inside the kernel for sketches of up to 2^13 cells; larger ones take the launch of its own below, which is faster
  // there even where the tables would fit (S=14 W=12: 28 -> 9 ms per 4096 records of 600 bases)
  a.distinct = (regular && own_cells && short_records && a.halves == 1 && 4u * a.d.R <= a.d.F && a.d.S <= 13 &&
                sketch_lds_bytes(a.d, true, 0) <= 150 * 1024) ? 1u : 0u;
  // candidate filter: long records only (the kernel picks its strength per sketch);
  // NIQKI_SKETCH_FILTER=0 switches it off
  // NIQKI_SKETCH_FILTER: 0 = off, unset/1 = automatic, n >= 2 = force n-1 leading zeros (tests)
  const char *fv = std::getenv("NIQKI_SKETCH_FILTER");
  const uint32_t fmode = fv ? (uint32_t)std::atoi(fv) : 1u;
  a.filter = (regular && !short_records && a.seqs != nullptr) ? fmode : 0u;
  // ... or afterwards, as a launch of its own with a thread's hash words in registers, where the three tables do not fit
  // beside the cells (densify_distinct_kernel)
  const size_t late_lds = (size_t)a.d.F * 4 + (size_t)a.d.R * 4 + 64;
  // (records of up to 2^19 bases: longer ones leave no cell empty at these sketch sizes, and a 5 Mbp genome's sketch must not
  // pay a launch that finds nothing to do)
  const bool late_distinct = a.densify && !a.distinct && regular && own_cells && avg_len < (1u << 19) && a.halves == 1 && a.splits == 1 &&
                             4u * a.d.R <= a.d.F && a.d.R <= (uint32_t)kLateValues * 1024u && late_lds <= kLdsLimit;
  if (late_distinct) a.densify = 0;
  const size_t lds = sketch_lds_bytes(a.d, a.distinct != 0, a.filter ? (short_records ? 4 : 16) : 0, a.halves);
  dim3 grid(n_entry * a.splits * a.halves);
#define NQ_LAUNCH_SKETCH(B, G, KF)                                                               \
  do {                                                                                           \
    auto k = sketch_kernel<B, G, KF>;                                                            \
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                               \
    hipLaunchKernelGGL(k, grid, dim3(B), lds, stream, a);                                        \
  } while (0)
  if (short_records) {
    if (a.d.K == 31) NQ_LAUNCH_SKETCH(256, 1, 31); else NQ_LAUNCH_SKETCH(256, 1, 0);
  } else if (avg_len < (1u << 18)) {
    if (a.d.K == 31) NQ_LAUNCH_SKETCH(1024, 2, 31); else NQ_LAUNCH_SKETCH(1024, 2, 0);
  } else if (avg_len < (1u << 21)) {
    if (a.d.K == 31) NQ_LAUNCH_SKETCH(1024, 8, 31); else NQ_LAUNCH_SKETCH(1024, 8, 0);
  } else {
    if (a.d.K == 31) NQ_LAUNCH_SKETCH(1024, 32, 31); else NQ_LAUNCH_SKETCH(1024, 32, 0);
  }
#undef NQ_LAUNCH_SKETCH
  if (late_distinct) {
    hipError_t e = hipFuncSetAttribute((const void *)densify_distinct_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)late_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(densify_distinct_kernel, dim3(n_entry), dim3(1024), late_lds, stream, a);
  }
  return hipGetLastError();
}

// ---- integer-ALU ceiling probes ---------------------------------------------------------------
template <int WHAT>
__global__ __launch_bounds__(1024) void alu_probe_kernel(uint32_t iters, uint32_t *sink) {
  const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
  if (WHAT == 2) {
    // the long-record path's arithmetic per k-mer as the kernel does it (K = 31): rolling update of both
    // words from pre-placed codes, canonical choice, high word of revhash64 as v_mad_u64_u32 chains and
    // the filter compare; 16 k-mers per round
    uint64_t fw = t * 0x9E3779B97F4A7C15ULL, rc = ~fw;
    fw &= (1ULL << 62) - 1; rc &= (1ULL << 62) - 1;
    uint32_t e = t * 2654435761u, pass = 0;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t c = (e >> (2 * j)) & 3u;
        fw = shl2_64(fw);
        fw = (fw | c) & ((1ULL << 62) - 1ULL);
        rc = shr2_64(rc) | ((uint64_t)((3u - c) << 28) << 32);
        const uint64_t canon = fw < rc ? fw : rc;
        pass += rev64_hi_mad(canon) < (1u << 29);
      }
      e = e * 1664525u + 1013904223u;
    }
    if (pass == 0xFFFFFFFFu) sink[0] = pass;
  } else if (WHAT == 3) {
    // a three-operand integer instruction (v_lshl_add_u32): the issue rate of every vector opcode outside the
    // add / sub / and / or / xor / mov / shift-right class on gfx950 (profiles/r03_opcode_costs.txt)
    uint32_t a0 = t, a1 = t * 3 + 1, a2 = t ^ 0x1234567, a3 = t + 77, a4 = t * 5, a5 = ~t, a6 = t + 9, a7 = t * 7;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#define NQ_LA(x, y) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(y))
        NQ_LA(a0, a1); NQ_LA(a1, a2); NQ_LA(a2, a3); NQ_LA(a3, a4); NQ_LA(a4, a5); NQ_LA(a5, a6); NQ_LA(a6, a7); NQ_LA(a7, a0);
#undef NQ_LA
      }
    }
    const uint32_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (x == 0x12345u) sink[0] = x;
  } else {
    uint32_t a0 = t, a1 = t * 3 + 1, a2 = t ^ 0x1234567, a3 = t + 77, a4 = t * 5, a5 = ~t, a6 = t + 9, a7 = t * 7;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (WHAT == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0; }
        else { a0 *= a1; a1 *= a2; a2 *= a3; a3 *= a4; a4 *= a5; a5 *= a6; a6 *= a7; a7 *= (a0 | 1u); }  // data dependent: nothing folds
      }
    }
    const uint32_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (x == 0x12345u) sink[0] = x;
  }
}

// streaming copy: the HBM rate a kernel reaches on this device (the practical ceiling beside the spec peak).
// Four 16-byte nontemporal loads in flight per thread, one workgroup per 16 KB: the fastest of the forms in
// tools/ubench_copy.hip (6.2 TB/s; a plain one-load loop 4.8, hipMemcpyAsync 4.7).
typedef unsigned int copy_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_probe_kernel(const copy_u4 *src, copy_u4 *dst, uint64_t n) {
  const uint64_t step = (uint64_t)gridDim.x * 1024;
  for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += step) {
    copy_u4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i + u * 256 < n) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
}
hipError_t launch_copy_probe(const void *src, void *dst, uint64_t bytes, hipStream_t stream) {
  hipLaunchKernelGGL(copy_probe_kernel, dim3(32768), dim3(256), 0, stream, (const copy_u4 *)src, (copy_u4 *)dst, bytes / 16);
  return hipGetLastError();
}

// The short-read kernel's densification passes with nothing but their LDS traffic and exit test: one wavefront per
// workgroup with the 150-base kernel's LDS footprint (the F = 4096 cells + entry list + code tile: 8 workgroups per
// CU), two register entries per lane (a 150-base read has ~120 occupied cells), per pass two ds_min_u32 proposals
// to pseudo-random cells, their read-backs behind them in issue order, a ballot + popcount and the advance of the
// targets -- densify_wave_entries<2> without winner writes.  The rate of these passes is the ceiling the passes of
// sketch_reads_kernel are measured against (LDS round-trip latency at 8 waves per CU).
__global__ __launch_bounds__(64) void lds_pass_probe_kernel(uint32_t iters, uint32_t *sink) {
  extern __shared__ __align__(16) uint32_t smem[];
  const uint32_t F = 4096, Fm = F - 1u, lane = threadIdx.x;
  for (uint32_t i = lane; i < F; i += 64) smem[i] = kEmpty32;
  __syncthreads();
  uint32_t T[2], B[2], mk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t v = (blockIdx.x * 131u + lane * 2u + (uint32_t)k) * 2654435761u;
    T[k] = (uint32_t)unrev64(v);
    B[k] = (uint32_t)rev64(v) | 1u;
    mk[k] = 0x80000000u | ((lane * 2u + (uint32_t)k) & Fm);
  }
  uint32_t tot = 0;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 2; ++k) atomicMin(&smem[T[k] & Fm], mk[k]);
    wave_lds_order();
    uint32_t back[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) back[k] = smem[T[k] & Fm];
    wave_lds_order();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      tot += (uint32_t)__popcll(__ballot(back[k] == mk[k]));
      T[k] += B[k];
    }
    if (tot == 0xFFFFFFFFu) break;   // (the exit test of a pass; never taken)
  }
  if (tot == 0x12345u) sink[0] = tot;
}

hipError_t launch_alu_probe(int what, uint32_t iters, uint32_t *sink, uint64_t *units, hipStream_t stream) {
  if (what == 5) {
    const uint32_t blocks = 256 * 9 * 4;   // four rounds of 9 one-wave workgroups per CU
    const size_t lds = 4096 * 4 + 192 * 8;   // sketch_reads_lds_bytes at S = 12 for 150-base reads
    hipLaunchKernelGGL(lds_pass_probe_kernel, dim3(blocks), dim3(64), lds, stream, iters * 16, sink);
    *units = (uint64_t)blocks * iters * 16;
    return hipGetLastError();
  }
  const uint32_t blocks = 256 * 4;  // four 1024-thread workgroups per CU: 16 waves per SIMD-quartet, as the sketch kernel
  const uint64_t threads = (uint64_t)blocks * 1024;
  if (what == 0) { hipLaunchKernelGGL(alu_probe_kernel<0>, dim3(blocks), dim3(1024), 0, stream, iters, sink); *units = threads * iters * 64; }
  else if (what == 1) { hipLaunchKernelGGL(alu_probe_kernel<1>, dim3(blocks), dim3(1024), 0, stream, iters, sink); *units = threads * iters * 64; }
  else if (what == 2) { hipLaunchKernelGGL(alu_probe_kernel<2>, dim3(blocks), dim3(1024), 0, stream, iters, sink); *units = threads * iters * 16; }
  else if (what == 3) { hipLaunchKernelGGL(alu_probe_kernel<3>, dim3(blocks), dim3(1024), 0, stream, iters, sink); *units = threads * iters * 64; }
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

__global__ void fill_u32_kernel(uint32_t *p, uint64_t n, uint32_t v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

hipError_t launch_fill_u32(uint32_t *p, uint64_t n, uint32_t v, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, p, n, v);
  return hipGetLastError();
}

}  // namespace nq
