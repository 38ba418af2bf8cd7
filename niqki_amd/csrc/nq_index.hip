// nq_index.hip -- kernel #3: sketch store and inverted-index build for gfx950.
//
// Replaces the reference's vector<gid> Buckets[2^W * F] with push_back under
// striped locks (src/niqki_index.h:55, src/niqki_index.cpp:27,362-370) and the
// bucket walks of dump_index_disk / the loading constructor (:42-55, :63-90).
//
// Layout: inserted sketches are kept slot-major as u16
// [F][cap]; the index is rebuilt from that store by a stable counting sort of each
// (tile, slot) row on the fingerprint: entries {start,len} for every (slot, fp,
// tile) and the tile-local u16 id lists, optionally padded so that every bucket
// starts on a 128-byte line.  Buckets come out ascending in genome id, the order
// the reference produces single-threaded.
#include "nq_kernels.h"

#include <algorithm>

namespace nq {

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, uint32_t lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(x, o, 64);
    if (lane >= (uint32_t)o) x += y;
  }
  return x;
}

// ---- int32 [n][F] sketches -> u16 [F_local][cap] store (64x64 tiles via LDS) ----
__global__ __launch_bounds__(256) void store_insert_kernel(Derived d, const int32_t *sk, uint32_t sk_stride,
                                                          uint32_t sk_off, uint32_t n, uint16_t *store,
                                                          uint64_t cap, uint32_t first_gid) {
  __shared__ uint16_t tile[64][66];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t f_local = d.slot_end - d.slot_begin;
  const uint32_t s0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  for (uint32_t r = wave; r < 64; r += 4) {
    uint32_t i = i0 + r, s = s0 + lane;
    uint16_t v = kEmpty16;
    if (i < n && s < f_local) {
      int32_t x = sk[(uint64_t)i * sk_stride + sk_off + s];
      if (x >= 0 && (uint32_t)x < d.R) v = (uint16_t)x;  // src/niqki_index.cpp:364
    }
    tile[r][lane] = v;
  }
  __syncthreads();
  for (uint32_t r = wave; r < 64; r += 4) {
    uint32_t s = s0 + r, i = i0 + lane;
    if (s < f_local && i < n) store[(uint64_t)s * cap + first_gid + i] = tile[lane][r];
  }
}

hipError_t launch_store_insert(const Derived &d, const int32_t *sketches, uint32_t sk_stride, uint32_t sk_off,
                               uint32_t n, uint16_t *store, uint64_t cap, uint32_t first_gid,
                               hipStream_t stream) {
  if (n == 0) return hipSuccess;
  uint32_t f_local = d.slot_end - d.slot_begin;
  dim3 grid((f_local + 63) / 64, (n + 63) / 64);
  hipLaunchKernelGGL(store_insert_kernel, grid, dim3(256), 0, stream, d, sketches, sk_stride, sk_off, n, store,
                     cap, first_gid);
  return hipGetLastError();
}

// ---- store -> int32 sketches of genomes [begin, begin+n) ----
__global__ __launch_bounds__(256) void store_read_kernel(Derived d, const uint16_t *store,
                                                        uint64_t cap, uint32_t begin, uint32_t n,
                                                        int32_t *sk) {
  __shared__ uint16_t tile[64][66];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t s0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  for (uint32_t r = wave; r < 64; r += 4) {
    uint32_t s = s0 + r, i = i0 + lane;  // s is a global slot number here
    uint16_t v = kEmpty16;
    if (s >= d.slot_begin && s < d.slot_end && i < n)
      v = store[(uint64_t)(s - d.slot_begin) * cap + begin + i];
    tile[r][lane] = v;
  }
  __syncthreads();
  for (uint32_t r = wave; r < 64; r += 4) {
    uint32_t i = i0 + r, s = s0 + lane;
    if (i < n && s < d.F) {
      uint16_t v = tile[lane][r];
      sk[(uint64_t)i * d.F + s] = v == kEmpty16 ? -1 : (int32_t)v;
    }
  }
}

hipError_t launch_store_read(const Derived &d, const uint16_t *store, uint64_t cap, uint32_t begin,
                             uint32_t n, int32_t *sketches, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  dim3 grid((d.F + 63) / 64, (n + 63) / 64);
  hipLaunchKernelGGL(store_read_kernel, grid, dim3(256), 0, stream, d, store, cap, begin, n,
                     sketches);
  return hipGetLastError();
}

// ---- index build: one wave per (tile, slot) row, stable counting sort on fp ----
// FILL = false: units (1 << align_log2 ids) the row needs -> slot_units[t][s]
// FILL = true : entries {start, len} of every fingerprint + the ascending id lists
template <bool FILL>
__global__ void build_kernel(IndexView v, uint32_t *slot_units, Entry *entries, uint16_t *gids, uint32_t wpb) {
  extern __shared__ __align__(16) uint32_t smem[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint64_t task = (uint64_t)blockIdx.x * wpb + wave;
  if (task >= (uint64_t)v.n_tiles * v.f_local) return;  // wave-private work, no block barrier below
  const uint32_t t = (uint32_t)(task / v.f_local), s = (uint32_t)(task % v.f_local);
  const uint32_t R = v.d.R, W = v.d.W, a = v.align_log2;
  const uint32_t round = (1u << a) - 1u;
  uint32_t *cur = smem + (size_t)wave * R;
  const uint32_t n_t = tile_count(v, t);
  const uint16_t *srow = v.store + (uint64_t)s * v.cap;
  auto row_at = [&](uint32_t i) -> uint32_t { return srow[v.g_base + tile_gid(v, t, i)]; };  // i-th genome of the tile
  auto units_of = [&](uint32_t h) -> uint32_t { return (h + round) >> a; };  // units a bucket of h ids needs

  for (uint32_t i = lane; i < R; i += 64) cur[i] = 0;
  for (uint32_t i = lane; i < n_t; i += 64) {
    uint32_t fp = row_at(i);
    if (fp != kEmpty16) atomicAdd(&cur[fp], 1u);
  }
  if (!FILL) {
    uint32_t units = 0;
    for (uint32_t c = lane; c < R; c += 64) units += units_of(cur[c]);
    for (int o = 32; o > 0; o >>= 1) units += __shfl_down(units, o, 64);
    if (lane == 0) slot_units[(uint64_t)t * (v.f_local + 1) + s] = units;
    return;
  }
  const uint32_t base_units = slot_units[(uint64_t)t * (v.f_local + 1) + s];
  uint32_t running = 0;  // units used so far inside this row
  for (uint32_t c = 0; c < R; c += 64) {
    const uint32_t fp = c + lane;
    uint32_t h = fp < R ? cur[fp] : 0u;
    uint32_t x = units_of(h);
    uint32_t incl = wave_incl_scan(x, lane);
    uint32_t excl = running + incl - x;
    if (fp < R) {
      cur[fp] = excl << a;  // cursor, in ids, relative to the row's first unit
      entries[((uint64_t)s * R + fp) * v.n_tiles + t] = Entry{base_units + excl, h};
    }
    running += __shfl(incl, 63, 64);
  }
  uint16_t *gl = gids + v.tile_base[t] + ((uint64_t)base_units << a);
  const uint64_t lt_mask = (1ULL << lane) - 1ULL;
  for (uint32_t base = 0; base < n_t; base += 64) {
    uint32_t i = base + lane;
    uint32_t fp = (i < n_t) ? row_at(i) : (uint32_t)kEmpty16;
    bool valid = fp != kEmpty16;
    uint64_t peers = __ballot(valid);
    for (uint32_t b = 0; b < W; ++b) {
      bool bit = (fp >> b) & 1u;
      uint64_t bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    if (valid) {
      uint32_t rank = __popcll(peers & lt_mask);
      uint32_t cnt = __popcll(peers);
      uint32_t p = cur[fp];
      gl[p + rank] = (uint16_t)i;
      if (rank == cnt - 1) cur[fp] = p + cnt;
    }
  }
}

// per tile: exclusive prefix of slot_units over the slots; total -> totals[t] (in ids)
__global__ __launch_bounds__(1024) void slot_scan_kernel(IndexView v, uint32_t *slot_units, uint64_t *totals) {
  __shared__ uint64_t part[1024];
  const uint32_t t = blockIdx.x, tid = threadIdx.x;
  const uint32_t per = (v.f_local + 1023) / 1024;
  const uint32_t lo = tid * per, hi = (lo + per < v.f_local) ? lo + per : v.f_local;
  uint32_t *u = slot_units + (uint64_t)t * (v.f_local + 1);
  uint64_t sum = 0;
  for (uint32_t s = lo; s < hi; ++s) sum += u[s];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    uint64_t run = 0;
    for (uint32_t i = 0; i < 1024; ++i) { uint64_t x = part[i]; part[i] = run; run += x; }
    u[v.f_local] = (uint32_t)run;   // padded layout: also the unit of the tile's spare padding line
    totals[t + 1] = (run + (v.padded ? 1u : 0u)) << v.align_log2;
  }
  __syncthreads();
  uint64_t run = part[tid];
  for (uint32_t s = lo; s < hi; ++s) { uint32_t x = u[s]; u[s] = (uint32_t)run; run += x; }
}

__global__ void tile_base_kernel(uint64_t *tile_base, uint32_t n_tiles) {
  if (threadIdx.x || blockIdx.x) return;
  uint64_t run = 0;
  tile_base[0] = 0;
  for (uint32_t t = 0; t < n_tiles; ++t) { run += tile_base[t + 1]; tile_base[t + 1] = run; }
}

static void build_shape(const IndexView &v, uint32_t &wpb, size_t &lds, uint64_t &blocks) {
  size_t per_wave = (size_t)v.d.R * 4;
  wpb = (uint32_t)(65536 / per_wave);
  if (wpb > 4) wpb = 4;
  if (wpb < 1) wpb = 1;
  lds = per_wave * wpb;
  uint64_t tasks = (uint64_t)v.n_tiles * v.f_local;
  blocks = (tasks + wpb - 1) / wpb;
}

hipError_t launch_build_sizes(const IndexView &v, uint32_t *slot_units, uint64_t *tile_base,
                              hipStream_t stream) {
  if ((uint64_t)v.n_tiles * v.f_local == 0) return hipSuccess;
  uint32_t wpb; size_t lds; uint64_t blocks;
  build_shape(v, wpb, lds, blocks);
  hipError_t e = hipFuncSetAttribute((const void *)build_kernel<false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_kernel<false>, dim3((uint32_t)blocks), dim3(64 * wpb), lds, stream, v, slot_units,
                     (Entry *)nullptr, (uint16_t *)nullptr, wpb);
  hipLaunchKernelGGL(slot_scan_kernel, dim3(v.n_tiles), dim3(1024), 0, stream, v, slot_units, tile_base);
  hipLaunchKernelGGL(tile_base_kernel, dim3(1), dim3(64), 0, stream, tile_base, v.n_tiles);
  return hipGetLastError();
}

hipError_t launch_build_fill(const IndexView &v, Entry *entries, uint16_t *gids, hipStream_t stream) {
  if ((uint64_t)v.n_tiles * v.f_local == 0) return hipSuccess;
  uint32_t wpb; size_t lds; uint64_t blocks;
  build_shape(v, wpb, lds, blocks);
  hipError_t e = hipFuncSetAttribute((const void *)build_kernel<true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(build_kernel<true>, dim3((uint32_t)blocks), dim3(64 * wpb), lds, stream, v,
                     (uint32_t *)v.slot_units, entries, gids, wpb);
  return hipGetLastError();
}

// hmask[s] (IndexView::hmask) from the finished table: one wave per slot ORs the classes of its non-empty buckets
__global__ __launch_bounds__(256) void hmask_kernel(IndexView v, uint16_t *hmask) {
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  if (s >= v.f_local) return;
  const uint32_t R = v.d.R;
  uint32_t m = 0;
  for (uint32_t fp = lane; fp < R; fp += 64) {
    const Entry *e = v.entries + ((uint64_t)s * R + fp) * v.n_tiles;
    uint32_t len = 0;
    for (uint32_t t = 0; t < v.n_tiles; ++t) len |= e[t].len;
    if (len) m |= 1u << (fp >> v.hmask_shift);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o, 64);
  if (lane == 0) hmask[s] = (uint16_t)m;
}
hipError_t launch_hmask(const IndexView &v, uint16_t *hmask, hipStream_t stream) {
  if (v.f_local == 0) return hipSuccess;
  hipLaunchKernelGGL(hmask_kernel, dim3((v.f_local + 3) / 4), dim3(256), 0, stream, v, hmask);
  return hipGetLastError();
}

// padded layout: position p of every 128-byte line holds id tile + 2p until the fill overwrites it
__global__ __launch_bounds__(256) void pad_fill_kernel(uint4 *gids, uint64_t n_vec, uint32_t tile) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    const uint32_t p = (uint32_t)(i & 7u) * 8u;   // first of this vector's 8 positions in its line
    auto pair = [&](uint32_t k) { return (tile + 2u * (p + k)) | ((tile + 2u * (p + k + 1)) << 16); };
    gids[i] = make_uint4(pair(0), pair(2), pair(4), pair(6));
  }
}

hipError_t launch_pad_fill(uint16_t *gids, uint64_t n_ids, uint32_t tile, hipStream_t stream) {
  if (n_ids == 0) return hipSuccess;
  const uint64_t n_vec = n_ids / 8;
  const uint64_t blocks = std::min<uint64_t>((n_vec + 255) / 256, 16384);
  hipLaunchKernelGGL(pad_fill_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, (uint4 *)gids, n_vec, tile);
  return hipGetLastError();
}

// ---- dump stream export (src/niqki_index.cpp:42-55) ----
// slot_word[s] = word position of bucket (s, 0): s*R buckets' size words + the ids before it.
__device__ __forceinline__ uint32_t bucket_len(const IndexView &v, uint32_t t, uint32_t s, uint32_t fp) {
  return v.entries[((uint64_t)s * v.d.R + fp) * v.n_tiles + t].len;
}
__device__ __forceinline__ void bucket_copy(const IndexView &v, uint32_t t, uint32_t s, uint32_t fp, uint32_t *out,
                                            unsigned long long &pos) {
  const Entry e = v.entries[((uint64_t)s * v.d.R + fp) * v.n_tiles + t];
  const uint16_t *gl = v.gids + v.tile_base[t] + ((uint64_t)e.start << v.align_log2);
  for (uint32_t j = 0; j < e.len; ++j) out[pos++] = t * v.tile + gl[j];
}
// striped tiles: the bucket's ids in ascending GLOBAL order are a merge of the tiles' lists
// (each ascending in its local ids)
constexpr uint32_t kMaxTilesMerge = 64;
__device__ __forceinline__ void bucket_merge(const IndexView &v, uint32_t s, uint32_t fp, uint32_t *out,
                                             unsigned long long &pos) {
  const Entry *e = v.entries + ((uint64_t)s * v.d.R + fp) * v.n_tiles;
  uint32_t at[kMaxTilesMerge];
  for (uint32_t t = 0; t < v.n_tiles; ++t) at[t] = 0;
  for (;;) {
    uint32_t best = 0xFFFFFFFFu, bt = 0;
    for (uint32_t t = 0; t < v.n_tiles; ++t) {
      if (at[t] < e[t].len) {
        const uint16_t *gl = v.gids + v.tile_base[t] + ((uint64_t)e[t].start << v.align_log2);
        const uint32_t g = tile_gid(v, t, gl[at[t]]);
        if (g < best) { best = g; bt = t; }
      }
    }
    if (best == 0xFFFFFFFFu) break;
    out[pos++] = best;
    ++at[bt];
  }
}

__global__ __launch_bounds__(256) void export_slot_ids_kernel(IndexView v, unsigned long long *slot_word) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= v.f_local) return;
  const uint64_t n = (uint64_t)v.d.R * v.n_tiles;
  unsigned long long sum = 0;
  for (uint64_t i = lane; i < n; i += 64) sum += bucket_len(v, (uint32_t)(i % v.n_tiles), s, (uint32_t)(i / v.n_tiles));
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
  if (lane == 0) slot_word[s + 1] = sum;  // raw; prefixed below
}

__global__ void export_slot_scan_kernel(unsigned long long *slot_word, uint32_t f_local, uint32_t R) {
  if (threadIdx.x || blockIdx.x) return;
  unsigned long long run = 0;
  slot_word[0] = 0;
  for (uint32_t s = 0; s < f_local; ++s) { run += slot_word[s + 1] + R; slot_word[s + 1] = run; }
}

// slots [s0, s1) into `out`, whose word 0 is the first word of slot s0
__global__ __launch_bounds__(256) void export_kernel(IndexView v, const unsigned long long *slot_word, uint32_t *out,
                                                    uint32_t s0, uint32_t s1) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t s = s0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= s1) return;
  const uint32_t R = v.d.R;
  unsigned long long running = slot_word[s] - slot_word[s0];
  for (uint32_t c = 0; c < R; c += 64) {
    const uint32_t fp = c + lane;
    uint32_t size = 0;
    if (fp < R)
      for (uint32_t t = 0; t < v.n_tiles; ++t) size += bucket_len(v, t, s, fp);
    // inclusive scan of (1 + size) words
    unsigned long long x = fp < R ? 1ull + size : 0ull, incl = x;
    for (int o = 1; o < 64; o <<= 1) {
      unsigned long long y = __shfl_up(incl, o, 64);
      if (lane >= (uint32_t)o) incl += y;
    }
    unsigned long long pos = running + incl - x;
    if (fp < R) {
      out[pos++] = size;
      if (v.stripe && v.n_tiles > 1) bucket_merge(v, s, fp, out, pos);
      else for (uint32_t t = 0; t < v.n_tiles; ++t) bucket_copy(v, t, s, fp, out, pos);
    }
    running += __shfl(incl, 63, 64);
  }
}

hipError_t launch_export_layout(const IndexView &v, unsigned long long *slot_word, hipStream_t stream) {
  if (v.f_local == 0) return hipSuccess;
  dim3 grid((v.f_local + 3) / 4);
  hipLaunchKernelGGL(export_slot_ids_kernel, grid, dim3(256), 0, stream, v, slot_word);
  hipLaunchKernelGGL(export_slot_scan_kernel, dim3(1), dim3(64), 0, stream, slot_word, v.f_local, v.d.R);
  return hipGetLastError();
}

hipError_t launch_export(const IndexView &v, const unsigned long long *slot_word, uint32_t *out,
                         uint32_t s0, uint32_t s1, hipStream_t stream) {
  if (s1 <= s0) return hipSuccess;
  dim3 grid((s1 - s0 + 3) / 4);
  hipLaunchKernelGGL(export_kernel, grid, dim3(256), 0, stream, v, slot_word, out, s0, s1);
  return hipGetLastError();
}

// ---- dump stream import (src/niqki_index.cpp:78-85): one wave per slot ----
// slots [s0, s0+n_slots): slot_word[i] = word position of slot s0+i inside `words`
__global__ __launch_bounds__(256) void import_kernel(Derived d, const uint32_t *words,
                                                    const uint64_t *slot_word, uint16_t *store,
                                                    uint64_t cap, uint32_t n_genomes, uint32_t *bad,
                                                    uint32_t s0, uint32_t n_slots) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_slots) return;
  const uint32_t s = s0 + i;
  uint64_t p = slot_word[i];
  const uint64_t end = slot_word[i + 1];
  for (uint32_t fp = 0; fp < d.R; ++fp) {
    if (p >= end) { if (lane == 0) atomicAdd(bad, 1u); return; }
    uint32_t size = words[p];
    for (uint32_t j = lane; j < size; j += 64) {
      uint32_t g = words[p + 1 + j];
      if (g < n_genomes) store[(uint64_t)s * cap + g] = (uint16_t)fp;
      else atomicAdd(bad, 1u);
    }
    p += 1 + (uint64_t)size;
  }
  if (p != end && lane == 0) atomicAdd(bad, 1u);
}

hipError_t launch_import(const Derived &d, const uint32_t *words, const uint64_t *slot_word,
                         uint16_t *store, uint64_t cap, uint32_t n_genomes, uint32_t *bad,
                         uint32_t s0, uint32_t n_slots, hipStream_t stream) {
  if (n_slots == 0) return hipSuccess;
  hipLaunchKernelGGL(import_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, stream, d, words, slot_word,
                     store, cap, n_genomes, bad, s0, n_slots);
  return hipGetLastError();
}

}  // namespace nq
