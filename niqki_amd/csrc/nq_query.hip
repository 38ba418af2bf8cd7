// nq_query.hip -- kernel #4: gather-histogram query over the inverted index,
// and the threshold / compaction / ordering of the hits, for gfx950.
//
// Replaces Index::query_sketch (src/niqki_index.cpp:633-687): the counting
// loop (:652-661) is gather_kernel, the threshold (:662-666) and the
// descending (count, gid) order (:685) are the hits_* kernels.
//
// gather_kernel: one workgroup per (query, genome tile).  The tile's per-genome
// hit counters live in LDS as packed u16 pairs (a count never exceeds F <= 2^15,
// so the two halves of a word cannot carry into each other) and are bumped with
// ds_add_u32.  Each wave takes 64 sketch slots at a time: every lane looks up
// the bucket of its slot (fp -> two adjacent CSR offsets), then the wave walks
// the 64 buckets one after another, all lanes reading consecutive u16 genome
// ids of one bucket (coalesced, <=128 B per bucket chunk).  HBM-bound by design:
// algorithmic bytes per query = 4T + 20F.
#include "nq_kernels.h"

namespace nq {

__device__ __forceinline__ void bump(uint32_t *cnt, uint32_t g) {
  atomicAdd(&cnt[g >> 1], 1u << ((g & 1u) * 16u));  // ds_add_u32, result unused
}

struct __attribute__((packed, aligned(4))) OffPair { uint32_t lo, hi; };

// UNROLL buckets are fetched per round trip; two rounds are kept in flight
// (the gid loads of round r+1 are issued before the LDS atomics of round r), and
// the CSR lookups run two 64-slot iterations ahead of the bucket walk, so the
// wave never waits on a load it has just issued.
// MODE is a measurement aid (bench ablations only, results are wrong for MODE != 0):
//   1 = no LDS atomics, 2 = no gid loads (synthetic ids), 3 = lookups only.
template <int BLOCK, int UNROLL, int MODE = 0, int ROT = 0>
__global__ __launch_bounds__(BLOCK) void gather_kernel(IndexView v, const int32_t *sketches,
                                                       uint16_t *counts, uint64_t stride) {
  extern __shared__ __align__(16) uint32_t cnt[];
  const uint32_t q = blockIdx.x / v.n_tiles, t = blockIdx.x % v.n_tiles;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t NW = BLOCK / 64;
  const uint32_t R = v.d.R;
  const uint32_t g0 = t * v.tile;
  const uint32_t n_t = (v.n_genomes - g0) < v.tile ? (v.n_genomes - g0) : v.tile;
  const uint32_t n_words = (n_t + 1) / 2;

  for (uint32_t i = tid; i < n_words; i += BLOCK) cnt[i] = 0;
  __syncthreads();

  const int32_t *sk = sketches + (uint64_t)q * v.d.F + v.d.slot_begin;
  const uint32_t *off = v.offsets + (uint64_t)t * v.f_local * (R + 1);
  const uint16_t *gl = v.gids + (uint64_t)t * v.f_local * v.tile;
  const uint32_t n_it = (v.f_local + 63) / 64;

  // Both lookups are unconditional loads from clamped addresses (validity is
  // applied when the values are used), so that they stay in flight across the
  // bucket walk instead of being waited for where they are issued.
  // ROT: every workgroup starts its walk over the slots at a different place, so
  // that the workgroups in flight do not all hit the same window of the table.
  const uint32_t rot = (ROT && n_it % NW == 0) ? (uint32_t)(((uint64_t)blockIdx.x * 2654435761u) >> 12) % n_it : 0u;
  auto phys = [&](uint32_t it) -> uint32_t { uint32_t e = it + rot; return e >= n_it && it < n_it ? e - n_it : e; };
  auto load_fp = [&](uint32_t it) -> int32_t {
    const uint32_t s = phys(it) * 64 + lane;
    return sk[s < v.f_local ? s : v.f_local - 1];
  };
  auto slot_ok = [&](uint32_t it) -> bool { return it < n_it && phys(it) * 64 + lane < v.f_local; };
  auto load_off = [&](uint32_t it, int32_t fp, bool ok) -> OffPair {
    ok = ok && fp >= 0 && (uint32_t)fp < R;  // src/niqki_index.cpp:654
    const uint64_t key = ok ? (uint64_t)(phys(it) * 64 + lane) * (R + 1) + (uint32_t)fp : 0;
    return *(const OffPair *)(off + key);
  };
  auto valid_of = [&](uint32_t it, int32_t fp) -> bool {
    return slot_ok(it) && fp >= 0 && (uint32_t)fp < R;
  };

  uint32_t sink = 0;  // keeps the loads alive in the ablation modes
  // pipeline prologue
  uint32_t it = wave;
  int32_t fp0 = load_fp(it);
  int32_t fp1 = load_fp(it + NW);                         // fingerprints one iteration ahead
  OffPair cur = load_off(it, fp0, slot_ok(it));           // bucket extents of the current iteration
  bool cur_ok = valid_of(it, fp0);

  for (; it < n_it; it += NW) {
    // lookups for the next iteration, fingerprints for the one after
    const OffPair nxt = load_off(it + NW, fp1, slot_ok(it + NW));
    const bool nxt_ok = valid_of(it + NW, fp1);
    fp1 = load_fp(it + 2 * NW);
    const uint32_t o0 = cur.lo, len = cur_ok ? cur.hi - cur.lo : 0u;

    uint32_t ga[UNROLL], gb[UNROLL];
    // Unconditional loads (lanes past the end re-read the bucket's last id, an
    // empty bucket reads one id at its offset -- the gid array is padded), so
    // that the compiler can count them in vmcnt and keep both rounds in flight.
    auto fetch = [&](uint32_t j0, uint32_t (&g)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t b = __builtin_amdgcn_readlane(o0, j0 + u);
        const uint32_t l = __builtin_amdgcn_readlane(len, j0 + u);
        const uint32_t last = l ? l - 1 : 0;
        if (MODE >= 2) g[u] = ((b * 2654435761u + lane * 40503u) >> 17) & 0x7FFFu;
        else g[u] = gl[b + (lane < last ? lane : last)];
      }
    };
    auto apply = [&](uint32_t j0, uint32_t (&g)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint32_t l = __builtin_amdgcn_readlane(len, j0 + u);
        if (MODE == 1 || MODE == 3) { if (lane < l) sink ^= g[u]; }
        else if (lane < l) bump(cnt, g[u]);
      }
    };
    fetch(0, ga);
#pragma unroll
    for (uint32_t j0 = 0; j0 < 64; j0 += 2 * UNROLL) {
      fetch(j0 + UNROLL, gb);
      apply(j0, ga);
      if (j0 + 2 * UNROLL < 64) fetch(j0 + 2 * UNROLL, ga);
      apply(j0 + UNROLL, gb);
    }
    // buckets longer than one wave: the rest, 64 ids at a time
    if (__any(len > 64)) {
      for (uint32_t j = 0; j < 64; ++j) {
        const uint32_t l = __builtin_amdgcn_readlane(len, j);
        if (l > 64) {
          const uint32_t b = __builtin_amdgcn_readlane(o0, j);
          for (uint32_t e = 64 + lane; e < l; e += 64) bump(cnt, gl[b + e]);
        }
      }
    }
    cur = nxt;
    cur_ok = nxt_ok;
  }
  if (MODE != 0) cnt[tid % n_words] ^= sink;
  __syncthreads();

  // dense counter row of this tile: u16 counts[q*stride + g0 + i], written as the packed words
  uint32_t *out = (uint32_t *)(counts + (uint64_t)q * stride + g0);
  const uint32_t full = n_t / 2;
  for (uint32_t i = tid; i < full; i += BLOCK) out[i] = cnt[i];
  if ((n_t & 1u) && tid == 0) counts[(uint64_t)q * stride + g0 + n_t - 1] = (uint16_t)(cnt[full] & 0xFFFFu);
}

hipError_t launch_gather(const IndexView &v, const int32_t *sketches, uint32_t nq, uint16_t *counts,
                         uint64_t stride, int variant, hipStream_t stream) {
  if (nq == 0 || v.n_tiles == 0) return hipSuccess;
  size_t lds = (size_t)((v.tile + 1) / 2) * 4;
  dim3 grid(nq * v.n_tiles);
  hipError_t e;
#define NQ_LAUNCH_GATHER(B, U, ...)                                                              \
  do {                                                                                           \
    auto k = gather_kernel<B, U, ##__VA_ARGS__>;                                                                \
    e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                               \
    hipLaunchKernelGGL(k, grid, dim3(B), lds, stream, v, sketches, counts, stride);              \
  } while (0)
  switch (variant) {
    case 1: NQ_LAUNCH_GATHER(1024, 8); break;
    case 2: NQ_LAUNCH_GATHER(1024, 32); break;
    case 3: NQ_LAUNCH_GATHER(512, 16); break;
    case 4: NQ_LAUNCH_GATHER(512, 32); break;
    case 5: NQ_LAUNCH_GATHER(1024, 16, 0, 1); break;  // rotated slot walk
    case 6: NQ_LAUNCH_GATHER(1024, 8, 0, 1); break;
    case 15: NQ_LAUNCH_GATHER(1024, 16, 3, 1); break;
    case 11: NQ_LAUNCH_GATHER(1024, 16, 1); break;  // ablations, see MODE
    case 12: NQ_LAUNCH_GATHER(1024, 16, 2); break;
    case 13: NQ_LAUNCH_GATHER(1024, 16, 3); break;
    default: NQ_LAUNCH_GATHER(1024, 16); break;
  }
#undef NQ_LAUNCH_GATHER
  return hipGetLastError();
}

// Sum of touched bucket lengths per query (T of the roofline formula).
__global__ __launch_bounds__(256) void gathered_kernel(IndexView v, const int32_t *sketches,
                                                      unsigned long long *per_query) {
  const uint32_t q = blockIdx.x;
  const uint32_t R = v.d.R;
  const int32_t *sk = sketches + (uint64_t)q * v.d.F + v.d.slot_begin;
  unsigned long long sum = 0;
  for (uint32_t s = threadIdx.x; s < v.f_local; s += blockDim.x) {
    int32_t fp = sk[s];
    if (fp >= 0 && (uint32_t)fp < R)
      for (uint32_t t = 0; t < v.n_tiles; ++t) {
        const uint32_t *p = v.offsets + ((uint64_t)t * v.f_local + s) * (R + 1) + (uint32_t)fp;
        sum += p[1] - p[0];
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

// blk_counts[q][b] = number of genomes of block b with count >= min_score
__global__ __launch_bounds__(256) void hits_count_kernel(HitsArgs a) {
  __shared__ uint32_t s_sum;
  const uint32_t q = blockIdx.x / a.n_blk, b = blockIdx.x % a.n_blk;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const uint16_t *row = a.counts + (uint64_t)q * a.stride + a.gid_begin;
  const uint32_t lo = b * kHitsBlk;
  const uint32_t hi = (lo + kHitsBlk < a.n_gids) ? lo + kHitsBlk : a.n_gids;
  uint32_t c = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) c += (row[i] >= a.min_score);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63u) == 0 && c) atomicAdd(&s_sum, c);
  __syncthreads();
  if (threadIdx.x == 0) a.blk_counts[blockIdx.x] = s_sum;
}

// exclusive scan of blk_counts (nq*n_blk entries) in place + hit_off[q]; single workgroup
__global__ __launch_bounds__(1024) void hits_scan_kernel(HitsArgs a) {
  __shared__ unsigned long long part[1024];
  const uint64_t n = (uint64_t)a.nq * a.n_blk;
  const uint32_t tid = threadIdx.x;
  const uint64_t per = (n + 1023) / 1024;
  const uint64_t lo = tid * per, hi = (lo + per < n) ? lo + per : n;
  unsigned long long sum = 0;
  for (uint64_t i = lo; i < hi; ++i) sum += a.blk_counts[i];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    unsigned long long run = 0;
    for (uint32_t i = 0; i < 1024; ++i) { unsigned long long x = part[i]; part[i] = run; run += x; }
    a.hit_off[a.nq] = run;
  }
  __syncthreads();
  unsigned long long run = part[tid];
  for (uint64_t i = lo; i < hi; ++i) {
    if (i % a.n_blk == 0) a.hit_off[i / a.n_blk] = run;
    run += a.blk_counts[i];
  }
}

// Each (query, block) writes its hits at the mirrored position so that a
// query's segment ends up in DESCENDING gid order; the global block prefix is
// recomputed from hit_off and the per-block counts.
__global__ __launch_bounds__(256) void hits_compact_kernel(HitsArgs a) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_before;
  const uint32_t q = blockIdx.x / a.n_blk, b = blockIdx.x % a.n_blk;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) {
    uint32_t before = 0;
    for (uint32_t i = 0; i < b; ++i) before += a.blk_counts[(uint64_t)q * a.n_blk + i];
    s_before = before;
  }
  const unsigned long long seg0 = a.hit_off[q], seg1 = a.hit_off[q + 1];
  const uint16_t *row = a.counts + (uint64_t)q * a.stride + a.gid_begin;
  const uint32_t lo = b * kHitsBlk;
  const uint32_t hi = (lo + kHitsBlk < a.n_gids) ? lo + kHitsBlk : a.n_gids;
  __syncthreads();
  uint32_t run = s_before;  // hits of this query with smaller gid, so far
  for (uint32_t base = lo; base < hi; base += 256) {
    uint32_t i = base + tid;
    uint32_t c = (i < hi) ? (uint32_t)row[i] : 0u;
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
}

hipError_t launch_hits_count(const HitsArgs &a, hipStream_t stream) {
  if (a.nq == 0 || a.n_blk == 0) return hipSuccess;
  hipLaunchKernelGGL(hits_count_kernel, dim3(a.nq * a.n_blk), dim3(256), 0, stream, a);
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
