// nq_inflate.hip -- gzip members inflated on the GPU, one wavefront per file (or per size-tagged member).
//
// The reference opens every input through zstr::ifstream (src/zstr.hpp:190-203, :236-239: gzip is detected by its
// magic and inflated by zlib, member after member) before Index::Biogetline frames the lines
// (src/niqki_index.cpp:890-941); its example data (resources/*.fa.gz) and the genome collections it is run on are
// gzip'd FASTA.  With the framing on the device (nq_ingest.hip) the inflate was the last per-byte stage left on the
// host's cores; here a gzip file crosses PCIe as it is and this kernel writes its bytes where niqki_stage_raw's
// framing kernels expect them.
//
// DEFLATE (RFC 1951) decodes one symbol after the other, so a job -- a gzip file, or one member of a file whose members
// say how long they are (BGZF: the caller cuts such a file up) -- is serial: ONE 64-lane wavefront per job.  The
// parallelism is over the jobs of a launch and, inside the wavefront, over bit offsets and bytes:
//   a round   lane i decodes the token that WOULD start at bit pos + i of the input (literal, or length + distance with
//             their extra bits: two table gathers), whatever the real token boundaries are; the chain of real tokens is
//             walked through the lanes' results by v_readlane; the bytes of all tokens of the round, up to 64, are
//             written in one LDS gather + scatter.  A token that is not plain (end of block, a code behind the tables,
//             a match overlapping the round's own bytes) is taken serially from the wave-uniform reader.
//   ring      the window in LDS (all 32 KB of it, or its last 8 KB with far matches read back from the job's own
//             flushed output: template parameter); a byte leaves for HBM when its 4 KB page is complete (16-byte
//             stores, CRC-32 of the page on the way: 64 lanes x 64 bytes, folded with the x^(8n) mod P operators of xtab)
//   lit / dist tables   10 / 8 bits per look-up (longer codes: the canonical walk), built by the wave per block
//   inbuf     the two 256-byte input blocks the lanes' windows reach into (the block behind them is on its way)
// The decoder state (bit position, output position) is wave-uniform and lives in scalar registers.  A wavefront alone
// on its SIMD issues a dependent instruction every other turn (~10 cycles each here): LDS decides how many run.
//
// What is not a plain, intact gzip file of the announced size is not decided here: any irregularity ends the file's
// job with a status != 0 and the caller (niqki_stage_raw) reports the file, which the host then reads through zlib
// as before -- the kernel is at least as strict as zlib's inflate (RFC 1951 + zlib's inflate_table rules), so a file
// it accepts has exactly zlib's bytes.  Every loop consumes input bits or ends; every store lies inside the job's
// [dst, dst + cap) and every load inside the wire buffer, whatever the bytes say.
#include "nq_kernels.h"

namespace nq {

namespace {

constexpr uint32_t kLitP = 10, kDistP = 8, kClP = 7;
constexpr uint32_t kPage = 4096;
constexpr uint32_t kKindLit = 0, kKindLen = 1, kKindEob = 2, kKindSlow = 3;
constexpr uint32_t kSlow = kKindSlow << 5;

__constant__ uint8_t c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Table entries (0 = no such code):
//   literal/length  bits 0-4 code length, 5-6 kind; literal: 8-15 the byte; length: 8-16 base, 17-19 extra bits
//   distance        bits 0-4 code length, 5-6 kind (0 or slow); 8-22 base, 24-27 extra bits
//   code lengths    bits 0-4 code length, 8-12 the symbol
__device__ __forceinline__ uint32_t lit_entry(uint32_t sym, uint32_t l) {
  if (sym < 256u) return l | (kKindLit << 5) | (sym << 8);
  if (sym == 256u) return l | (kKindEob << 5);
  if (sym < 286u) {
    // RFC 1951 3.2.5 in closed form: codes 257..264 are lengths 3..10; then four codes per extra bit; 285 is 258
    const uint32_t i = sym - 257u;
    const uint32_t x = i < 8u || i == 28u ? 0u : (i >> 2) - 1u;
    const uint32_t base = i < 8u ? 3u + i : i == 28u ? 258u : 3u + ((4u + (i & 3u)) << x);
    return l | (kKindLen << 5) | (base << 8) | (x << 17);
  }
  return 0u;   // 286, 287: in the fixed code, never valid
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t l) {
  if (sym < 30u) {
    // codes 0..3 are distances 1..4; then two codes per extra bit
    const uint32_t x = sym < 4u ? 0u : (sym >> 1) - 1u;
    const uint32_t base = sym < 4u ? 1u + sym : 1u + ((2u + (sym & 1u)) << x);
    return l | (base << 8) | (x << 24);
  }
  return 0u;   // 30, 31
}
template <int WHICH>
__device__ __forceinline__ uint32_t make_entry(uint32_t sym, uint32_t l) {
  return WHICH == 0 ? lit_entry(sym, l) : WHICH == 1 ? dist_entry(sym, l) : (l | (sym << 8));
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

#ifdef NQ_INFLATE_CLOCK
#define NQ_CLK(i) do { const uint64_t t_ = __builtin_readcyclecounter(); res.clk[i] += t_ - clk_last; clk_last = t_; } while (0)
#else
#define NQ_CLK(i) do {} while (0)
#endif

// all LDS traffic of the wave so far is done and the compiler keeps the accesses on either side apart (one wave
// per workgroup: the LDS serves its instructions in order)
__device__ __forceinline__ void lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a(x) * b(x) mod P in the reflected representation of CRC-32 (bit 31 = x^0), as zlib's multmodp
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
#pragma unroll 4
  for (int i = 0; i < 32; ++i) {
    p ^= (a & 0x80000000u) ? b : 0u;
    a <<= 1;
    b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
  }
  return p;
}

// Code lengths L[0..n) -> canonical-Huffman decoding data: cnt / first / offs per length, the symbols ordered by
// (length, symbol) in `sorted`, and the table of 2^P entries indexed by the next P stream bits.  false: the lengths
// are over-subscribed, or incomplete in a way zlib's inflate_table refuses.
template <int WHICH, uint32_t P>
__device__ bool build_table(const uint8_t *L, uint32_t n, uint32_t *tab, uint16_t *sorted, uint32_t *cnt, uint32_t *first,
                            uint32_t *offs, uint32_t *run, uint32_t lane) {
  if (lane < 16u) cnt[lane] = 0u;
  lds_fence();
  for (uint32_t s = lane; s < n; s += 64u) {
    const uint32_t l = L[s];
    if (l) atomicAdd(&cnt[l], 1u);
  }
  lds_fence();
  // (every lane walks the 15 lengths: the same values everywhere)
  uint32_t code = 0, at = 0, maxl = 0;
  int32_t left = 1;
  bool over = false;
  for (uint32_t l = 1; l <= 15u; ++l) {
    const uint32_t c = cnt[l];
    code <<= 1;
    left <<= 1;
    left -= (int32_t)c;
    over |= left < 0;
    if (lane == 0u) { first[l] = code; offs[l] = at; run[l] = at; }
    code += c;
    at += c;
    if (c) maxl = l;
  }
  if (over) return false;
  if (left > 0 && (WHICH == 2 || maxl != 1u)) return false;   // zlib: an incomplete set only as one 1-bit code
  if (WHICH == 2 && maxl == 0u) return false;
  lds_fence();
  // symbols in (length, symbol) order: rank among the symbols of the same length by ballots
  for (uint32_t s0 = 0; s0 < n; s0 += 64u) {
    const uint32_t s = s0 + lane;
    const uint32_t l = s < n ? L[s] : 0u;
    uint64_t pending = __ballot(l != 0u);
    while (pending) {
      const uint32_t src = (uint32_t)__builtin_ctzll(pending);
      const uint32_t ll = (uint32_t)__builtin_amdgcn_readlane((int)l, (int)src);
      const uint64_t m = __ballot(l == ll);
      const uint32_t base = run[ll];
      if (l == ll) sorted[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)s;
      lds_fence();
      if (lane == 0u) run[ll] = base + (uint32_t)__popcll(m);
      lds_fence();
      pending &= ~m;
    }
  }
  // the table, one entry per lane and round: the canonical walk over the entry's own index bits
  for (uint32_t i = lane; i < (1u << P); i += 64u) {
    uint32_t c2 = 0, at_l = 0, at_i = 0;
    for (uint32_t l = 1; l <= P; ++l) {
      c2 = (c2 << 1) | ((i >> (l - 1u)) & 1u);
      const uint32_t idx = c2 - first[l];
      if (at_l == 0u && idx < cnt[l]) { at_l = l; at_i = offs[l] + idx; }
    }
    tab[i] = at_l ? make_entry<WHICH>(sorted[at_i], at_l) : kSlow;
  }
  lds_fence();
  return true;
}

}  // namespace

// kRing: bytes of the window kept in LDS.  32768: all of it (40.9 KB per file: four wavefronts per CU).  8192: the
// last 8 KB (16.3 KB per file: nine fit a CU, eight -- two per SIMD -- is what a batch of 2048 uses); a match that
// reaches further back (kNear) reads its bytes from the file's own output in HBM, which the flushes have written by
// then: a byte is flushed at most 4 KB + one match + one round after it is made.  A wavefront alone on its SIMD issues
// a dependent instruction every other turn, so two per SIMD run at 0.88 of the speed each (1024 files: 147 ms whole
// window, 160 ms this form; 2048 files: 180 ms).  (A 9-bit literal table would make it ten per CU: 2560 files in
// 206 ms, + 9 % -- not worth batches of 2560.)
template <uint32_t kRing>
__global__ __launch_bounds__(64) void inflate_kernel(const InflateJob *jobs, const uint8_t *wire, uint64_t wire_bytes,
                                                     uint8_t *raw, const uint32_t *xtab, InflateOut *outs) {
  constexpr uint32_t kRingMask = kRing - 1u;
  constexpr bool kFar = kRing < 32768u;
  constexpr uint32_t kNear = kFar ? kRing - 512u : 0xFFFFFFFFu;   // distances up to here are served by the ring
  __shared__ __align__(16) uint8_t ring[kRing];
  __shared__ uint32_t lit_tab[1u << kLitP];
  __shared__ uint32_t dist_tab[1u << kDistP];
  __shared__ uint32_t crc_tab[256];
  __shared__ uint16_t lit_sorted[288];
  __shared__ uint16_t dist_sorted[32];
  __shared__ uint8_t lens[320];
  __shared__ uint8_t cl_lens[32];
  __shared__ uint32_t hc[2][3][16];   // [lit, dist][cnt, first, offs]
  __shared__ uint32_t run[16];
  __shared__ uint32_t inbuf[132];   // the input blocks bblk and bblk + 1 (slot = block & 1), dwords 0..3 once more behind them

  const uint32_t lane = threadIdx.x;
  const InflateJob job = jobs[blockIdx.x];
  InflateOut res;
  res.status = 0; res.members = 0; res.produced = 0; res.consumed = 0;
  res.rounds = 0; res.round_bytes = 0; res.serial_tokens = 0; res.blocks = 0;
#ifdef NQ_INFLATE_CLOCK
  for (int i = 0; i < 8; ++i) res.clk[i] = 0;
  uint64_t clk_last = __builtin_readcyclecounter();
#endif

  // CRC-32 byte table (reflected, polynomial 0xEDB88320)
  for (uint32_t i = lane; i < 256u; i += 64u) {
    uint32_t c = i;
#pragma unroll
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0xEDB88320u : 0u);
    crc_tab[i] = c;
  }
  lds_fence();

  // ---- input: dwords from the wire buffer, 64 at a time in a vector register; the next 160 bits in five scalars ----
  const uint64_t in_addr = job.src;                           // byte offset of the file in `wire`
  const uint32_t skip = (uint32_t)(in_addr & 3u);             // bytes of the first dword that precede the file
  const uint32_t *in_base = (const uint32_t *)(wire + (in_addr - skip));
  const uint64_t last_dword64 = ((wire_bytes + 3u) >> 2) - 1u - ((in_addr - skip) >> 2);   // last loadable dword, from in_base
  const uint32_t last_dword = last_dword64 > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)last_dword64;
  const uint32_t n_words = (uint32_t)((skip + job.src_len + 3u) >> 2);   // dwords that hold bytes of the file (< 2^29)
  const uint64_t total_bits = (skip + job.src_len) * 8u;
  uint32_t err = 0;
  uint32_t cur, nxt, nn;  // lane i: dword 64 * bblk + i of the input / the blocks behind it (nn: on its way)
  uint32_t bblk = 0;      // which 64-dword block `cur` holds
  uint32_t bitpos = 0;    // the stream's next bit, counted from the start of `cur`; < 2048 between tokens
  auto load_blk = [&](uint32_t b) -> uint32_t {
    uint32_t w = b * 64u + lane;
    w = w > last_dword ? last_dword : w;
    return in_base[w];
  };
  // the lanes read the window of a round from LDS: block b lies in slot b & 1, and the first dwords of slot 0 are
  // repeated behind slot 1, so that three consecutive dwords never wrap
  auto stage_blk = [&](uint32_t b, uint32_t v) {
    const uint32_t slot = (b & 1u) << 6;
    inbuf[slot + lane] = v;
    if (slot == 0u && lane < 4u) inbuf[128u + lane] = v;
  };
  auto s_dword = [&](uint32_t i) -> uint32_t {   // dword i (< 128) behind the start of `cur`, wave-uniform
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(i & 63u));
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)(i & 63u));
    return i < 64u ? a : b;
  };
  auto drop = [&](uint32_t n) {   // n < 2048
    bitpos += n;
    if (bitpos >= 2048u) {
      cur = nxt;
      nxt = nn;              // (asked for a whole block of input ago)
      ++bblk;
      stage_blk(bblk + 1u, nxt);
      nn = load_blk(bblk + 2u);
      bitpos -= 2048u;
      if (bblk * 64u > n_words + 64u) err = err ? err : 8u;   // far past the file: truncated
    }
  };
  auto peek64 = [&]() -> uint64_t {
    const uint32_t i = bitpos >> 5, sh = bitpos & 31u;
    const uint32_t d0 = s_dword(i), d1 = s_dword(i + 1u), d2 = s_dword(i + 2u);
    const uint64_t lo = (((uint64_t)d1 << 32) | d0) >> sh;
    const uint64_t hi = ((uint64_t)d2 << 32) << (32u - sh);   // (sh = 0: shifted out entirely)
    return lo | (sh ? hi : 0ull);
  };
  auto getbits = [&](uint32_t n) -> uint32_t {   // n <= 32
    const uint32_t v = (uint32_t)(peek64() & ((1ull << n) - 1ull));
    drop(n);
    return v;
  };
  auto consumed_bits = [&]() -> uint64_t { return (uint64_t)bblk * 2048u + bitpos; };
  auto seek_byte = [&](uint64_t byte_from_base) {   // (re)start the reader at a byte position counted from in_base
    bblk = (uint32_t)(byte_from_base >> 8);
    cur = load_blk(bblk);
    nxt = load_blk(bblk + 1u);
    nn = load_blk(bblk + 2u);
    lds_fence();
    stage_blk(bblk, cur);
    stage_blk(bblk + 1u, nxt);
    lds_fence();
    bitpos = 8u * (uint32_t)(byte_from_base & 255u);
  };

  // ---- output: positions count from pos0 = (dst & 15), so that ring index and HBM address agree modulo 16 ----
  const uint32_t pos0 = (uint32_t)(job.dst & 15u);
  uint8_t *out_al = raw + (job.dst - pos0);
  const uint32_t cap_end = pos0 + (uint32_t)job.cap;
  uint32_t pos = pos0, flushed = pos0, member_start = pos0;
  uint32_t crc = 0;   // CRC-32 of the member's bytes below `flushed`

  // a byte of the file's own output that has left the ring (agent scope: past this CU's L1, from the L2 the flushes
  // wrote through to)
  auto load_far = [&](uint32_t p) -> uint32_t {
    return __hip_atomic_load(out_al + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // bytes [lo, hi) of one 4 KB page of positions leave the ring: HBM stores, and the member's CRC moves on
  auto flush = [&](uint32_t lo, uint32_t hi) {
    lds_fence();
    const uint32_t win = lo & ~(kPage - 1u);
    const uint32_t b0 = win + lane * 64u;
    const uint32_t a = lo > b0 ? lo : b0, b = hi < b0 + 64u ? hi : b0 + 64u;
    uint32_t c = 0;
    if (a < b) {
      c = 0xFFFFFFFFu;
      if (b - a == 64u) {
        const uint4 *rp = (const uint4 *)(ring + (b0 & kRingMask));
        uint4 *gp = (uint4 *)(out_al + b0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint4 v = rp[k];
          gp[k] = v;
          const uint32_t w4v[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c ^= w4v[j];
#pragma unroll
            for (int t = 0; t < 4; ++t) c = crc_tab[c & 0xFFu] ^ (c >> 8);
          }
        }
      } else {
        for (uint32_t p = a; p < b; ++p) {
          const uint32_t v = ring[p & kRingMask];
          out_al[p] = (uint8_t)v;
          c = crc_tab[(c ^ v) & 0xFFu] ^ (c >> 8);
        }
      }
      c = ~c;
    }
    // crc(A || B) = x^(8|B|) * crc(A) + crc(B): the lanes' pieces, then the member's running value
    const uint32_t first_blk = (lo - win) >> 6, last_blk = (hi - 1u - win) >> 6;
    const uint32_t tail = hi - (win + last_blk * 64u);   // bytes of the last piece, 1..64
    uint32_t t = 0;
    if (lane >= first_blk && lane < last_blk) t = mulmod(xtab[65u + (last_blk - lane - 1u)], c);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) t ^= (uint32_t)__shfl_xor((int)t, d, 64);
    const uint32_t c_last = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)last_blk);
    const uint32_t chunk = mulmod(xtab[tail], uni(t)) ^ c_last;
    const uint32_t n = hi - lo;
    crc = uni(mulmod(mulmod(xtab[65u + (n >> 6)], xtab[n & 63u]), crc) ^ chunk);
    if (kFar) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the page's stores are done before any later load_far
  };
  auto flush_pages = [&]() {   // after `pos` moved: every completed page
    while ((flushed ^ pos) & ~(kPage - 1u)) {
      const uint32_t hi = (flushed & ~(kPage - 1u)) + kPage;
      flush(flushed, hi);
      flushed = hi;
    }
  };

  seek_byte(skip);
  if (job.src_len < 18u || job.src_len > 0x7FFF0000ull || job.cap > 0x7FFF0000ull) err = 1u;

  // ---- members ----
  while (!err) {
    // header (RFC 1952)
    const uint32_t id = getbits(16), cm = getbits(8), flg = getbits(8);
    (void)getbits(32);          // MTIME
    (void)getbits(16);          // XFL, OS
    if (id != 0x8B1Fu || cm != 8u || (flg & 0xE0u)) { err = err ? err : 1u; break; }
    if (flg & 4u) {             // FEXTRA
      uint32_t xlen = getbits(16);
      while (xlen && !err) { (void)getbits(8); --xlen; }
    }
    if (flg & 8u) while (!err && getbits(8) != 0u) {}    // FNAME
    if (flg & 16u) while (!err && getbits(8) != 0u) {}   // FCOMMENT
    if (flg & 2u) (void)getbits(16);                     // FHCRC (not verified, as zlib's gzread)
    if (err) break;
    crc = 0;
    member_start = pos;

    // ---- blocks (RFC 1951) ----
    uint32_t bfinal = 0;
    while (!err && !bfinal) {
      NQ_CLK(0);
      bfinal = getbits(1);
      const uint32_t btype = getbits(2);
      ++res.blocks;
      if (btype == 0u) {
        // stored: LEN bytes behind the next byte boundary
        drop((8u - (bitpos & 7u)) & 7u);
        const uint32_t len = getbits(16), nlen = getbits(16);
        if ((len ^ nlen) != 0xFFFFu) { err = err ? err : 3u; break; }
        if (err) break;
        uint64_t at = consumed_bits() >> 3;   // byte position from in_base
        if (at + len > skip + job.src_len) { err = 8u; break; }
        if (pos + len > cap_end) { err = 7u; break; }
        const uint8_t *sb = (const uint8_t *)in_base;
        uint32_t left = len;
        while (left) {
          const uint32_t room = kPage - (pos & (kPage - 1u));
          const uint32_t n = left < room ? left : room;
          for (uint32_t i = lane; i < n; i += 64u) ring[(pos + i) & kRingMask] = sb[at + i];
          pos += n; at += n; left -= n;
          flush_pages();
        }
        seek_byte(at);
        continue;
      }
      if (btype == 3u) { err = 2u; break; }
      if (btype == 1u) {
        // fixed code: lengths 8 / 9 / 7 / 8 for the 288 literal/length symbols, 5 for the 32 distances
        for (uint32_t s = lane; s < 288u; s += 64u) lens[s] = (uint8_t)(s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u);
        if (lane < 32u) lens[288u + lane] = 5u;
        lds_fence();
        build_table<0, kLitP>(lens, 288u, lit_tab, lit_sorted, hc[0][0], hc[0][1], hc[0][2], run, lane);
        build_table<1, kDistP>(lens + 288, 32u, dist_tab, dist_sorted, hc[1][0], hc[1][1], hc[1][2], run, lane);
      } else {
        const uint32_t hlit = getbits(5) + 257u, hdist = getbits(5) + 1u, hclen = getbits(4) + 4u;
        if (hlit > 286u || hdist > 30u) { err = err ? err : 4u; break; }
        if (lane < 19u) cl_lens[lane] = 0u;
        lds_fence();
        for (uint32_t i = 0; i < hclen; ++i) {
          const uint32_t v = getbits(3);
          cl_lens[c_cl_order[i]] = (uint8_t)v;   // (all lanes: the same byte)
        }
        lds_fence();
        if (err) break;
        // the code-length code's table and order share the literal table's memory, which is built after its last use
        if (!build_table<2, kClP>(cl_lens, 19u, lit_tab, lit_sorted, hc[0][0], hc[0][1], hc[0][2], run, lane)) { err = 4u; break; }
        const uint32_t total = hlit + hdist;
        uint32_t i = 0, prev = 0;
        while (i < total && !err) {
          const uint32_t e = uni(lit_tab[(uint32_t)peek64() & ((1u << kClP) - 1u)]);
          const uint32_t l = e & 31u;
          if (l == 0u || (e & kSlow)) { err = 4u; break; }   // (code lengths are at most 7 bits: never slow)
          drop(l);
          const uint32_t sym = e >> 8;
          if (sym < 16u) {
            lens[i++] = (uint8_t)sym;
            prev = sym;
            continue;
          }
          uint32_t rep, val = 0;
          if (sym == 16u) {
            if (i == 0u) { err = 4u; break; }
            rep = 3u + getbits(2);
            val = prev;
          } else if (sym == 17u) {
            rep = 3u + getbits(3);
          } else {
            rep = 11u + getbits(7);
          }
          if (i + rep > total) { err = 4u; break; }
          for (uint32_t k = lane; k < rep; k += 64u) lens[i + k] = (uint8_t)val;
          i += rep;
          prev = val;
        }
        lds_fence();
        if (err) break;
        if (lens[256] == 0u) { err = 4u; break; }   // zlib: "missing end-of-block"
        if (!build_table<0, kLitP>(lens, hlit, lit_tab, lit_sorted, hc[0][0], hc[0][1], hc[0][2], run, lane)) { err = 4u; break; }
        if (!build_table<1, kDistP>(lens + hlit, hdist, dist_tab, dist_sorted, hc[1][0], hc[1][1], hc[1][2], run, lane)) { err = 4u; break; }
      }

      NQ_CLK(6);   // block header and tables
      // ---- symbols ----
      for (;;) {
        if (err) break;
        // A ROUND: lane i decodes the token that would start at bit i of the window -- literal, or length + distance
        // with their extra bits -- as if it were the next one; the chain of real tokens (0, its length, ...) is then
        // walked through the lanes' results, and the bytes of all tokens of the round (up to 64) are written in one
        // LDS gather + scatter.  The round ends before a token that is not plain (end of block, a code behind the
        // tables, a match that overlaps the round's own bytes, more than 64 bytes): the serial step below takes that one.
        NQ_CLK(0);   // everything outside the rounds: headers, tables, serial tokens
        uint32_t v_info, v_tokv;
        {
          // the 64 bits behind bit `bitpos + lane`: three dwords of the staged input
          const uint32_t q = bitpos + lane;
          const uint32_t *wp = inbuf + ((bblk & 1u) << 6) + (q >> 5);
          const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2];
          const uint32_t sh = q & 31u;
          const uint32_t x0 = __builtin_amdgcn_alignbit(d1, d0, sh), x1 = __builtin_amdgcn_alignbit(d2, d1, sh);
          const uint32_t e = lit_tab[x0 & ((1u << kLitP) - 1u)];
          const uint32_t l = e & 31u, kind = (e >> 5) & 3u;
          const uint32_t xl = (e >> 17) & 7u;
          const uint32_t t0 = __builtin_amdgcn_alignbit(x1, x0, l);
          const uint32_t mlen = ((e >> 8) & 511u) + (t0 & ((1u << xl) - 1u));
          const uint32_t a = l + xl;
          const uint32_t y = __builtin_amdgcn_alignbit(x1, x0, a);
          const uint32_t d = dist_tab[y & ((1u << kDistP) - 1u)];
          const uint32_t dl = d & 31u, xd = (d >> 24) & 15u;
          const uint32_t dist = ((d >> 8) & 0x7FFFu) + ((y >> dl) & ((1u << xd) - 1u));
          const bool lit = kind == kKindLit, mat = kind == kKindLen;
          // a token the round may take: a literal, or a match through both tables that reaches back no further than
          // the member's start (as seen from the round's start; a little strict near the start of a member)
          const bool plain = l != 0u && (lit || (mat && dl != 0u && (d & 0x60u) == 0u && dist <= pos - member_start));
          const uint32_t tot = lit ? l : a + dl + xd;
          const uint32_t olen = lit ? 1u : mlen;
          // where the round's output may end with this token in it: 64 bytes, the output's capacity, and -- a match
          // copies from before the round's first byte -- its distance
          uint32_t lim = cap_end - pos;
          lim = lim < 64u ? lim : 64u;
          lim = (lit || dist >= lim) ? lim : dist;
          v_info = plain ? (tot | (olen << 6) | (lim << 15)) : (1u << 6);   // tot 6 bits, olen 9, lim 7; no token here: a byte that never fits
          v_tokv = lit ? (0x10000u | ((e >> 8) & 0xFFu)) : dist;
        }
        uint32_t p, outb, v_tok;
#ifdef NQ_INFLATE_CLOCK
        asm volatile("s_nop 0" :: "v"(v_info), "v"(v_tokv));
        NQ_CLK(1);   // the lanes' decode
#endif
        // The chain of tokens through the lanes' results: the token at bit p takes the round's bytes [o, end) when
        // end <= its lim, and the lanes of those bytes remember its tokv.  Written out by hand (the compiler turns the
        // loop's exits into flag arithmetic and two taken branches per token): two tokens per trip, 14 instructions
        // each, the running end alternating between two registers.  Every adjacent producer / consumer pair below
        // also occurs in compiler-emitted code for gfx950; the s_nop covers the lanes' last vector writes of v_info /
        // v_tokv ahead of the first v_readlane.
        {
          uint32_t s_info, s_tv, s_o0, s_o1, s_lim, v_tmp;
          asm volatile(
              "s_nop 1\n\t"
              "s_mov_b32 %[p], 0\n\t"
              "s_mov_b32 %[o0], 0\n\t"
              "v_mov_b32 %[vtok], 0\n"
              ".Lnq_walk_%=:\n\t"
              "v_readlane_b32 %[info], %[vinfo], %[p]\n\t"
              "v_readlane_b32 %[tv], %[vtokv], %[p]\n\t"
              "s_bfe_u32 %[o1], %[info], 0x90006\n\t"
              "s_add_u32 %[o1], %[o1], %[o0]\n\t"
              "s_lshr_b32 %[lim], %[info], 15\n\t"
              "s_cmp_gt_u32 %[o1], %[lim]\n\t"
              "s_cbranch_scc1 .Lnq_walk_end0_%=\n\t"
              "v_mov_b32 %[vtmp], %[tv]\n\t"
              "v_cmp_gt_u32 vcc, %[o0], %[lane]\n\t"
              "s_and_b32 %[info], %[info], 63\n\t"
              "s_add_u32 %[p], %[p], %[info]\n\t"
              "v_cndmask_b32 %[vtok], %[vtmp], %[vtok], vcc\n\t"
              "s_cmp_lt_u32 %[p], 64\n\t"
              "s_cbranch_scc0 .Lnq_walk_end1_%=\n\t"
              "v_readlane_b32 %[info], %[vinfo], %[p]\n\t"
              "v_readlane_b32 %[tv], %[vtokv], %[p]\n\t"
              "s_bfe_u32 %[o0], %[info], 0x90006\n\t"
              "s_add_u32 %[o0], %[o0], %[o1]\n\t"
              "s_lshr_b32 %[lim], %[info], 15\n\t"
              "s_cmp_gt_u32 %[o0], %[lim]\n\t"
              "s_cbranch_scc1 .Lnq_walk_end1_%=\n\t"
              "v_mov_b32 %[vtmp], %[tv]\n\t"
              "v_cmp_gt_u32 vcc, %[o1], %[lane]\n\t"
              "s_and_b32 %[info], %[info], 63\n\t"
              "s_add_u32 %[p], %[p], %[info]\n\t"
              "v_cndmask_b32 %[vtok], %[vtmp], %[vtok], vcc\n\t"
              "s_cmp_lt_u32 %[p], 64\n\t"
              "s_cbranch_scc1 .Lnq_walk_%=\n\t"
              "s_mov_b32 %[outb], %[o0]\n\t"
              "s_branch .Lnq_walk_done_%=\n"
              ".Lnq_walk_end0_%=:\n\t"
              "s_mov_b32 %[outb], %[o0]\n\t"
              "s_branch .Lnq_walk_done_%=\n"
              ".Lnq_walk_end1_%=:\n\t"
              "s_mov_b32 %[outb], %[o1]\n"
              ".Lnq_walk_done_%=:\n\t"
              "s_nop 0"
              : [p] "=&s"(p), [outb] "=&s"(outb), [vtok] "=&v"(v_tok), [vtmp] "=&v"(v_tmp), [info] "=&s"(s_info), [tv] "=&s"(s_tv),
                [o0] "=&s"(s_o0), [o1] "=&s"(s_o1), [lim] "=&s"(s_lim)
              : [vinfo] "v"(v_info), [vtokv] "v"(v_tokv), [lane] "v"(lane)
              : "vcc", "scc");
        }
        NQ_CLK(2);   // the walk
        if (outb) {
          if (lane < outb) {
            const uint32_t dist = v_tok & 0xFFFFu, src = pos + lane - dist;
            uint32_t b;
            if (kFar && !(v_tok >> 16) && dist > kNear) b = load_far(src);
            else b = ring[src & kRingMask];
            ring[(pos + lane) & kRingMask] = (uint8_t)((v_tok >> 16) ? v_tok : b);
          }
          const uint32_t before = pos;
          pos += outb;
          ++res.rounds;
          res.round_bytes += outb;
          NQ_CLK(3);   // the copy
          drop(p);
          NQ_CLK(4);   // the reader moving on
          if ((before ^ pos) & ~(kPage - 1u)) { flush_pages(); NQ_CLK(5); }
          continue;
        }
        // ---- one token, serially ----
        NQ_CLK(0);
        ++res.serial_tokens;
        uint64_t bb = peek64();
        uint32_t e = uni(lit_tab[(uint32_t)bb & ((1u << kLitP) - 1u)]);
        if ((e & 0x60u) == kSlow) {
          // a code of more than kLitP bits (or none at all): the canonical walk, bit by bit
          uint32_t c2 = 0, found = 0;
          for (uint32_t l = 1; l <= 15u; ++l) {
            c2 = (c2 << 1) | ((uint32_t)(bb >> (l - 1u)) & 1u);
            const uint32_t idx = c2 - uni(hc[0][1][l]);
            if (idx < uni(hc[0][0][l])) {
              e = lit_entry(uni((uint32_t)lit_sorted[uni(hc[0][2][l]) + idx]), l);
              found = 1;
              break;
            }
          }
          if (!found) e = 0;
        }
        const uint32_t l = e & 31u;
        if (l == 0u) { err = 5u; break; }
        bb >>= l;
        uint32_t used = l;
        const uint32_t kind = (e >> 5) & 3u;
        if (kind == kKindLit) {
          if (pos >= cap_end) { err = 7u; break; }
          ring[pos & kRingMask] = (uint8_t)(e >> 8);
          ++pos;
          drop(used);
          if ((pos & (kPage - 1u)) == 0u) flush_pages();
          continue;
        }
        if (kind == kKindEob) { drop(used); break; }
        // a match: length, then distance (at most 15 + 5 + 15 + 13 = 48 bits in all)
        const uint32_t xl = (e >> 17) & 7u;
        const uint32_t mlen = ((e >> 8) & 511u) + ((uint32_t)bb & ((1u << xl) - 1u));
        bb >>= xl;
        used += xl;
        uint32_t d = uni(dist_tab[(uint32_t)bb & ((1u << kDistP) - 1u)]);
        if ((d & 0x60u) == kSlow) {
          uint32_t c2 = 0, found = 0;
          for (uint32_t dl = 1; dl <= 15u; ++dl) {
            c2 = (c2 << 1) | ((uint32_t)(bb >> (dl - 1u)) & 1u);
            const uint32_t idx = c2 - uni(hc[1][1][dl]);
            if (idx < uni(hc[1][0][dl])) {
              d = dist_entry(uni((uint32_t)dist_sorted[uni(hc[1][2][dl]) + idx]), dl);
              found = 1;
              break;
            }
          }
          if (!found) d = 0;
        }
        const uint32_t dl = d & 31u;
        if (dl == 0u) { err = 5u; break; }
        bb >>= dl;
        const uint32_t xd = (d >> 24) & 15u;
        const uint32_t dist = ((d >> 8) & 0x7FFFu) + ((uint32_t)bb & ((1u << xd) - 1u));
        used += dl + xd;
        drop(used);
        if (dist > pos - member_start) { err = 6u; break; }   // zlib: "invalid distance too far back"
        if (pos + mlen > cap_end) { err = 7u; break; }
        const uint32_t src0 = pos - dist;
        if (kFar && dist > kNear) {
          // the whole source has left the ring (it ends kNear - 258 bytes back or more)
          for (uint32_t o = 0; o < mlen; o += 64u) {
            const uint32_t i = o + lane;
            if (i < mlen) ring[(pos + i) & kRingMask] = (uint8_t)load_far(src0 + i);
          }
        } else if (dist >= 64u || dist >= mlen) {
          // 64 bytes at a time: a chunk's sources lie before the chunk, and the LDS serves the wave in order
          for (uint32_t o = 0; o < mlen; o += 64u) {
            const uint32_t i = o + lane;
            if (i < mlen) ring[(pos + i) & kRingMask] = ring[(src0 + i) & kRingMask];
          }
        } else {
          // a pattern of `dist` bytes repeated: every source byte was written before the match
          for (uint32_t o = 0; o < mlen; o += 64u) {
            const uint32_t i = o + lane;
            if (i < mlen) ring[(pos + i) & kRingMask] = ring[(src0 + i % dist) & kRingMask];
          }
        }
        const uint32_t before = pos;
        pos += mlen;
        if ((before ^ pos) & ~(kPage - 1u)) flush_pages();
        NQ_CLK(7);   // a token taken serially
      }
    }
    if (err) break;
    // trailer: CRC-32 and ISIZE behind the next byte boundary
    drop((8u - (bitpos & 7u)) & 7u);
    const uint32_t want_crc = getbits(32), want_size = getbits(32);
    if (err) break;
    if (pos != flushed) { flush(flushed, pos); flushed = pos; }
    if (crc != want_crc) { err = 9u; break; }
    if (want_size != pos - member_start) { err = 10u; break; }
    ++res.members;
    const uint64_t used = (consumed_bits() >> 3) - skip;   // bytes of the file consumed
    if (used > job.src_len) { err = 8u; break; }
    if (used == job.src_len) break;
    if (job.src_len - used < 18u) { err = 11u; break; }
    // (the next member's magic is checked by its header parse; anything else ends with status 1 -> the host decides)
  }
  if (!err) {
    if (consumed_bits() > total_bits) err = 8u;
    else if (pos != cap_end) err = 12u;
  }
  res.status = err;
  res.produced = pos - pos0;
  res.consumed = (consumed_bits() >> 3) >= skip ? (consumed_bits() >> 3) - skip : 0u;
  if (lane == 0u) outs[blockIdx.x] = res;
}

// x^(8k) and x^(512k) mod P, k = 0..64 (host side of the kernel's CRC folding): 130 words
void inflate_xtab(uint32_t *t) {
  auto mul = [](uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
      if (a & 0x80000000u) p ^= b;
      a <<= 1;
      b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
  };
  uint32_t x8 = 0x80000000u;
  for (int i = 0; i < 8; ++i) x8 = mul(x8, 0x40000000u);   // x^8
  t[0] = 0x80000000u;
  for (int k = 1; k <= 64; ++k) t[k] = mul(t[k - 1], x8);
  t[65] = 0x80000000u;
  for (int k = 1; k <= 64; ++k) t[65 + k] = mul(t[65 + k - 1], t[64]);
}

uint32_t inflate_resident_files(bool small_ring) {
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const hipError_t e = small_ring ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, inflate_kernel<8192u>, 64, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, inflate_kernel<32768u>, 64, 0);
  return e == hipSuccess ? (uint32_t)(per * cus) : 0u;
}

hipError_t launch_inflate(const InflateJob *jobs, uint32_t n_jobs, const uint8_t *wire, uint64_t wire_bytes, uint8_t *raw,
                          const uint32_t *xtab, InflateOut *outs, hipStream_t stream, int small_ring) {
  if (n_jobs == 0) return hipSuccess;
  // more files than the whole window in LDS lets run at once (four per CU): the small ring, eight per CU
  static const uint32_t big_form_files = inflate_resident_files(false);
  if (small_ring < 0 ? n_jobs > (big_form_files ? big_form_files : 1024u) : small_ring != 0)
    hipLaunchKernelGGL(inflate_kernel<8192u>, dim3(n_jobs), dim3(64), 0, stream, jobs, wire, wire_bytes, raw, xtab, outs);
  else
    hipLaunchKernelGGL(inflate_kernel<32768u>, dim3(n_jobs), dim3(64), 0, stream, jobs, wire, wire_bytes, raw, xtab, outs);
  return hipGetLastError();
}

}  // namespace nq
