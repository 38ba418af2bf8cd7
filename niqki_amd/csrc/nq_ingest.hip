// nq_ingest.hip -- FASTA / FASTQ record framing on the GPU.
//
// Replaces the host-side line reader of the reference (Index::Biogetline,
// src/niqki_index.cpp:890-941, and the read loops around it, :383-456, :505-519)
// for callers that hand over the raw bytes of their files: the bytes of a batch of
// files go to the device once, and three passes over 8 KB chunks turn them into the
// (seqs, rec_off, entry_rec) arrays the sketch kernel reads.
//
// Framing rules restated (the sketch kernel applies the "longer than K" test itself,
// a record of at most K bases contributes no k-mer):
//   FASTA 'A'  line 0 of a file and every line whose first byte is '>' (or 0xFF) is a header and
//              starts a record; all other lines, without their '\n', are the record's
//              sequence (getline + peek loop, :904-909).  Nothing is trimmed or upper-cased
//              ('\r' stays in the sequence as a non-ACGT byte, like there).
//   FASTQ 'Q'  line 4r is the header of record r, line 4r+1 its sequence, lines 4r+2 and
//              4r+3 are dropped (:897-901).
//
// Passes:  scan   per chunk: newline / line-start census that does not depend on what
//                 precedes the chunk (the dependence is carried symbolically)
//          files  one wavefront per file chains its chunks (state in, bases), one
//                 workgroup turns per-file totals into global bases
//          emit   per chunk: kept bytes compacted through LDS into `seqs` with aligned
//                 dword stores; record starts write rec_off / hdr_pos
//          entries (lines mode) records longer than K -> entries, capped at max_entries
#include "nq_kernels.h"

namespace nq {

namespace {

constexpr uint32_t kIB = kIngestBlock;       // threads per chunk
constexpr uint32_t kBPT = kIngestChunk / kIB;  // bytes per thread
static_assert(kBPT == 32, "one 32-bit mask per thread");
constexpr uint32_t kHdr = 0, kSeq = 1;       // FASTA line classes

// ---- block-wide scans over kIB threads ------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if ((threadIdx.x & 63) >= (uint32_t)d) v += o;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t o = __shfl_up(v, d, 64);
    if ((threadIdx.x & 63) >= (uint32_t)d) v = max(v, o);
  }
  return v;
}
// exclusive prefix sum of v over the block, total in *total; tmp: kIB/64 + 1 words of LDS
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t *tmp, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_add(v);
  __syncthreads();
  if (lane == 63) tmp[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t i = 0; i < kIB / 64; ++i) {
    const uint32_t t = tmp[i];
    if (i < w) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}
// exclusive prefix max (0 when nothing precedes)
__device__ __forceinline__ uint32_t block_excl_max(uint32_t v, uint32_t *tmp, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_max(v);
  uint32_t prev = __shfl_up(inc, 1, 64);
  if (lane == 0) prev = 0;
  __syncthreads();
  if (lane == 63) tmp[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (uint32_t i = 0; i < kIB / 64; ++i) {
    const uint32_t t = tmp[i];
    if (i < w) base = max(base, t);
    tot = max(tot, t);
  }
  *total = tot;
  return max(base, prev);
}

__device__ __forceinline__ uint32_t low_mask(uint32_t n) {  // bits [0, n), n <= 32
  return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
}

// The 32 bytes of one thread and what the framing needs to know about them.
struct Lane {
  uint32_t w[8];     // the bytes, little endian
  uint32_t valid;    // bit i: byte i lies inside the file
  uint32_t nl;       // valid byte i is '\n'
  uint32_t gt;       // valid byte i is '>' (or 0xFF, see load_lane)
  uint32_t ls;       // valid byte i is the first byte of a line
};

struct ChunkPos {
  uint32_t file;
  uint64_t file_begin, file_end, begin;  // raw offsets; begin = first byte of the chunk
  uint8_t type;
};

__device__ __forceinline__ ChunkPos locate(const IngestArgs &a, uint32_t chunk) {
  // last file whose first chunk is <= chunk (uniform per workgroup)
  uint32_t lo = 0, hi = a.n_files;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.chunk_first[mid] <= chunk) lo = mid; else hi = mid;
  }
  ChunkPos p;
  p.file = lo;
  p.file_begin = a.file_off[lo];
  p.file_end = a.file_off[lo + 1];
  p.begin = p.file_begin + (uint64_t)(chunk - a.chunk_first[lo]) * kIngestChunk;
  p.type = a.file_type[lo];
  return p;
}

__device__ __forceinline__ void load_lane(const IngestArgs &a, const ChunkPos &p, Lane &L) {
  const uint64_t at = p.begin + (uint64_t)threadIdx.x * kBPT;
  // aligned dwords around the 32 bytes, funnel-shifted into place (raw has >= 64
  // readable bytes after its end and starts 16-byte aligned)
  const uint32_t *src = (const uint32_t *)(a.raw + (at & ~(uint64_t)3));
  const uint32_t sh = (uint32_t)(at & 3);
  uint32_t d[9];
  if (at < p.file_end) {
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = src[i];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) L.w[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
  const uint64_t left = at < p.file_end ? p.file_end - at : 0;
  L.valid = low_mask(left >= 32 ? 32u : (uint32_t)left);
  uint32_t nl = 0, gt = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const uint32_t b = (L.w[i >> 2] >> ((i & 3) * 8)) & 0xFFu;
    nl |= (b == (uint32_t)'\n') << i;
    // `char c = in->peek(); while (c != '>' and c != EOF)` (:904-905): char is signed there,
    // so a line starting with byte 0xFF ends the record exactly like one starting with '>'
    gt |= (uint32_t)(b == (uint32_t)'>' || b == 0xFFu) << i;
  }
  L.nl = nl & L.valid;
  L.gt = gt & L.valid;
  // line starts: the byte after a newline, and the first byte of the file
  uint32_t prev_nl;
  const uint32_t last = __shfl_up(L.nl >> 31, 1, 64);
  if ((threadIdx.x & 63) != 0) {
    prev_nl = last;
  } else if (threadIdx.x == 0) {
    prev_nl = (p.begin == p.file_begin) ? 1u : (uint32_t)(a.raw[p.begin - 1] == '\n');
  } else {
    prev_nl = (uint32_t)(a.raw[at - 1] == '\n');  // first lane of waves 1..3: the neighbour wave's byte
  }
  L.ls = ((L.nl << 1) | prev_nl) & L.valid;
}

// FASTA: which bytes of the thread are sequence.  carry = class of the line in
// progress at the thread's first byte.  hdr_ls: line starts that are headers.
__device__ __forceinline__ uint32_t fasta_keep(const Lane &L, uint32_t hdr_ls, uint32_t carry) {
  uint32_t keep = 0, m = L.ls, start = 0, cls = carry;
  for (;;) {
    const uint32_t end = m ? (uint32_t)__builtin_ctz(m) : 32u;
    if (cls == kSeq) keep |= low_mask(end) & ~low_mask(start);
    if (!m) break;
    cls = ((hdr_ls >> end) & 1u) ? kHdr : kSeq;
    start = end;
    m &= m - 1;
  }
  return keep & L.valid & ~L.nl;
}

// FASTQ: bytes of lines with index % 4 == 1.  line0 = index of the line in progress
// at the thread's first byte (= newlines of the file before it).
__device__ __forceinline__ uint32_t fastq_keep(const Lane &L, uint32_t line0) {
  uint32_t keep = 0, m = L.nl, start = 0, line = line0;
  for (;;) {
    const uint32_t end = m ? (uint32_t)__builtin_ctz(m) : 32u;
    if ((line & 3u) == 1u) keep |= low_mask(end) & ~low_mask(start);
    if (!m) break;
    ++line;
    start = end + 1;
    m &= m - 1;
  }
  return keep & L.valid & ~L.nl;
}

// ---- pass 1 ------------------------------------------------------------------------
// summ[c] (5 words)   FASTA: {has_ls, class of the last line start, kept if the chunk
//                              starts inside a sequence line, sequence-candidate bytes
//                              before the first line start, header line starts}
//                     FASTQ: {newlines, bytes by (chunk-local line index % 4) x 4}
__global__ __launch_bounds__(kIB) void ingest_scan_kernel(IngestArgs a) {
  __shared__ uint32_t tmp[kIB / 64 + 1];
  __shared__ uint32_t acc[8];
  const uint32_t chunk = blockIdx.x;
  const ChunkPos p = locate(a, chunk);
  Lane L;
  load_lane(a, p, L);
  if (threadIdx.x < 8) acc[threadIdx.x] = 0;
  uint32_t *out = a.summ + (uint64_t)chunk * 5;
  if (p.type == 'Q') {
    uint32_t tot;
    const uint32_t nl_before = block_excl_add((uint32_t)__builtin_popcount(L.nl), tmp, &tot);
    // bytes per local line phase
    uint32_t cnt[4] = {0, 0, 0, 0};
    uint32_t m = L.nl, start = 0, line = nl_before;
    const uint32_t body = L.valid & ~L.nl;
    for (;;) {
      const uint32_t end = m ? (uint32_t)__builtin_ctz(m) : 32u;
      const uint32_t n = (uint32_t)__builtin_popcount(body & low_mask(end) & ~low_mask(start));
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) cnt[j] += ((line & 3u) == j) ? n : 0u;
      if (!m) break;
      ++line;
      start = end + 1;
      m &= m - 1;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      uint32_t v = cnt[j];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[j], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      out[0] = tot;
      out[1] = acc[0]; out[2] = acc[1]; out[3] = acc[2]; out[4] = acc[3];
    }
    return;
  }
  // FASTA
  const bool at_file_start = (p.begin == p.file_begin) && threadIdx.x == 0;
  const uint32_t hdr_ls = L.ls & (L.gt | (at_file_start ? 1u : 0u));
  // class of the line in progress at this thread's first byte: the last line start of
  // an earlier thread, else unknown (taken as sequence here, corrected by `pre`)
  uint32_t key = 0;
  if (L.ls) {
    const uint32_t top = 31u - (uint32_t)__builtin_clz(L.ls);
    key = ((threadIdx.x + 1) << 1) | (((hdr_ls >> top) & 1u) ? 0u : 1u);
  }
  uint32_t last_key;
  const uint32_t before = block_excl_max(key, tmp, &last_key);
  const uint32_t carry = before ? (before & 1u ? kSeq : kHdr) : kSeq;
  const uint32_t keep = fasta_keep(L, hdr_ls, carry);
  uint32_t pre = 0;
  if (!before) {  // bytes before the first line start of the chunk
    const uint32_t first = L.ls ? (uint32_t)__builtin_ctz(L.ls) : 32u;
    pre = (uint32_t)__builtin_popcount(L.valid & ~L.nl & low_mask(first));
  }
  uint32_t v0 = (uint32_t)__builtin_popcount(keep), v1 = pre, v2 = (uint32_t)__builtin_popcount(hdr_ls);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    v0 += __shfl_xor(v0, d, 64);
    v1 += __shfl_xor(v1, d, 64);
    v2 += __shfl_xor(v2, d, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    if (v0) atomicAdd(&acc[0], v0);
    if (v1) atomicAdd(&acc[1], v1);
    if (v2) atomicAdd(&acc[2], v2);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = last_key ? 1u : 0u;
    out[1] = (last_key & 1u) ? kSeq : kHdr;
    out[2] = acc[0];
    out[3] = acc[1];
    out[4] = acc[2];
  }
}

// ---- pass 2a: one wavefront per file ----------------------------------------------
// chunk_out[c] = {kept bytes of the file before the chunk (u64), record base (FASTA:
// header line starts before the chunk; FASTQ: newlines before it), FASTA class at
// the chunk's first byte}.  file_kept / file_nrec: totals of the file.
__global__ __launch_bounds__(64) void ingest_files_kernel(IngestArgs a) {
  const uint32_t f = blockIdx.x, lane = threadIdx.x;
  const uint32_t c0 = a.chunk_first[f], c1 = a.chunk_first[f + 1];
  const bool fastq = a.file_type[f] == 'Q';
  uint64_t kept_base = 0;
  uint32_t rec_base = 0, state = kHdr;
  for (uint32_t g = c0; g < c1; g += 64) {
    const uint32_t c = g + lane;
    const bool on = c < c1;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    if (on) {
      const uint32_t *s = a.summ + (uint64_t)c * 5;
      s0 = s[0]; s1 = s[1]; s2 = s[2]; s3 = s[3]; s4 = s[4];
    }
    uint32_t kept, recs, st_in = 0;
    uint32_t rec_before;
    if (fastq) {
      const uint32_t inc = wave_incl_add(s0);
      rec_before = rec_base + inc - s0;  // newlines before the chunk = index of its first line
      const uint32_t sel = (1u - rec_before) & 3u;
      kept = sel == 0 ? s1 : sel == 1 ? s2 : sel == 2 ? s3 : s4;
      recs = s0;
    } else {
      // class at the chunk's first byte: the last line start of an earlier chunk
      const uint64_t has = __ballot(on && s0 != 0);
      const uint64_t below = has & ((1ull << lane) - 1ull);
      uint32_t src = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
      const uint32_t from = __shfl(s1, src, 64);
      st_in = below ? from : state;
      kept = s2 - (st_in == kHdr ? s3 : 0u);
      recs = s4;
      const uint32_t inc = wave_incl_add(recs);
      rec_before = rec_base + inc - recs;
      // state after this group
      const uint32_t top = has ? 63u - (uint32_t)__builtin_clzll(has) : 0u;
      const uint32_t last = __shfl(s1, top, 64);
      if (has) state = last;
    }
    // 64-bit prefix of kept
    uint64_t kinc = kept;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t o = __shfl_up(kinc, d, 64);
      if (lane >= (uint32_t)d) kinc += o;
    }
    if (on) {
      uint32_t *o = a.chunk_out + (uint64_t)c * 4;
      const uint64_t kb = kept_base + kinc - kept;
      o[0] = (uint32_t)kb;
      o[1] = (uint32_t)(kb >> 32);
      o[2] = rec_before;
      o[3] = st_in;
    }
    kept_base += __shfl(kinc, 63, 64);
    const uint32_t rsum = wave_incl_add(recs);
    rec_base += __shfl(rsum, 63, 64);
  }
  if (lane == 0) {
    uint32_t nrec = rec_base;  // FASTA: header line starts
    if (fastq) {
      const uint64_t b = a.file_off[f], e = a.file_off[f + 1];
      const uint32_t lines = rec_base + ((e > b && a.raw[e - 1] != '\n') ? 1u : 0u);
      nrec = (lines + 3) >> 2;
    }
    a.file_kept[f] = kept_base;
    a.file_nrec[f] = nrec;
  }
}

// ---- pass 2b: exclusive scan over files (one workgroup) ---------------------------
// file_kept / file_nrec become bases (n_files+1 entries each, the last one the total);
// totals[0] = records, totals[1] = kept bytes.
__global__ __launch_bounds__(1024) void ingest_bases_kernel(IngestArgs a) {
  __shared__ uint64_t sk[1024];
  __shared__ uint32_t sr[1024];
  __shared__ uint64_t carry_k;
  __shared__ uint32_t carry_r;
  const uint32_t t = threadIdx.x;
  if (t == 0) { carry_k = 0; carry_r = 0; }
  __syncthreads();
  for (uint32_t f0 = 0; f0 < a.n_files; f0 += 1024) {
    const uint32_t f = f0 + t;
    const uint64_t k = f < a.n_files ? a.file_kept[f] : 0;
    const uint32_t r = f < a.n_files ? a.file_nrec[f] : 0;
    sk[t] = k;
    sr[t] = r;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint64_t ok = t >= d ? sk[t - d] : 0;
      const uint32_t orr = t >= d ? sr[t - d] : 0;
      __syncthreads();
      sk[t] += ok;
      sr[t] += orr;
      __syncthreads();
    }
    if (f < a.n_files) {
      a.file_kept[f] = carry_k + sk[t] - k;
      a.file_nrec[f] = carry_r + sr[t] - r;
    }
    __syncthreads();
    if (t == 1023) { carry_k += sk[t]; carry_r += sr[t]; }
    __syncthreads();
  }
  if (t == 0) {
    a.file_kept[a.n_files] = carry_k;
    a.file_nrec[a.n_files] = carry_r;
    a.totals[0] = carry_r;
    a.totals[1] = carry_k;
  }
}

// ---- pass 3 -------------------------------------------------------------------------
__global__ __launch_bounds__(kIB) void ingest_emit_kernel(IngestArgs a) {
  __shared__ uint32_t tmp[kIB / 64 + 1];
  __shared__ __attribute__((aligned(16))) uint8_t stage[kIngestChunk + 16];
  const uint32_t chunk = blockIdx.x;
  const ChunkPos p = locate(a, chunk);
  Lane L;
  load_lane(a, p, L);
  const uint32_t *co = a.chunk_out + (uint64_t)chunk * 4;
  const uint64_t out_base = a.file_kept[p.file] + (((uint64_t)co[1] << 32) | co[0]);
  const uint32_t rec_in = co[2], st_in = co[3];
  const uint32_t rec_file = a.file_nrec[p.file];

  uint32_t keep, rec_ls;  // rec_ls: line starts that begin a record
  uint32_t line0 = 0;
  if (p.type == 'Q') {
    uint32_t tot;
    line0 = rec_in + block_excl_add((uint32_t)__builtin_popcount(L.nl), tmp, &tot);
    keep = fastq_keep(L, line0);
    // line starts whose index is a multiple of 4
    rec_ls = 0;
    uint32_t m = L.ls;
    while (m) {
      const uint32_t i = (uint32_t)__builtin_ctz(m);
      const uint32_t idx = line0 + (uint32_t)__builtin_popcount(L.nl & low_mask(i));
      if ((idx & 3u) == 0) rec_ls |= 1u << i;
      m &= m - 1;
    }
  } else {
    const bool at_file_start = (p.begin == p.file_begin) && threadIdx.x == 0;
    const uint32_t hdr_ls = L.ls & (L.gt | (at_file_start ? 1u : 0u));
    uint32_t key = 0;
    if (L.ls) {
      const uint32_t top = 31u - (uint32_t)__builtin_clz(L.ls);
      key = ((threadIdx.x + 1) << 1) | (((hdr_ls >> top) & 1u) ? 0u : 1u);
    }
    uint32_t last_key;
    const uint32_t before = block_excl_max(key, tmp, &last_key);
    const uint32_t carry = before ? (before & 1u ? kSeq : kHdr) : st_in;
    keep = fasta_keep(L, hdr_ls, carry);
    rec_ls = hdr_ls;
  }
  uint32_t n_keep;
  const uint32_t my_out = block_excl_add((uint32_t)__builtin_popcount(keep), tmp, &n_keep);
  uint32_t n_rec_chunk;
  const uint32_t my_rec = block_excl_add((uint32_t)__builtin_popcount(rec_ls), tmp, &n_rec_chunk);

  // record starts
  {
    uint32_t m = rec_ls, j = 0;
    while (m) {
      const uint32_t i = (uint32_t)__builtin_ctz(m);
      uint32_t r;
      if (p.type == 'Q') r = (line0 + (uint32_t)__builtin_popcount(L.nl & low_mask(i))) >> 2;
      else r = rec_in + my_rec + j;
      r += rec_file;
      a.rec_off[r] = out_base + my_out + (uint32_t)__builtin_popcount(keep & low_mask(i));
      a.hdr_pos[r] = p.begin + (uint64_t)threadIdx.x * kBPT + i;
      ++j;
      m &= m - 1;
    }
  }
  // kept bytes -> LDS at the destination's dword phase, then aligned dword stores
  uint8_t *dst = a.seqs + out_base;
  const uint32_t mis = (uint32_t)((uintptr_t)dst & 3u);
  {
    uint32_t m = keep, at = mis + my_out;
    if (keep == 0xFFFFFFFFu && (at & 3u) == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *(uint32_t *)(stage + at + 4 * i) = L.w[i];
    } else {
      while (m) {
        const uint32_t i = (uint32_t)__builtin_ctz(m);
        stage[at++] = (uint8_t)(L.w[i >> 2] >> ((i & 3) * 8));
        m &= m - 1;
      }
    }
  }
  __syncthreads();
  const uint32_t span_end = mis + n_keep;           // staged bytes are [mis, span_end)
  const uint32_t d0 = (mis + 3) >> 2;               // first whole dword
  const uint32_t d1 = span_end >> 2;                // one past the last whole dword
  uint8_t *dbase = dst - mis;                       // dword aligned
  if (d1 > d0) {
    for (uint32_t d = d0 + threadIdx.x; d < d1; d += kIB) ((uint32_t *)dbase)[d] = ((const uint32_t *)stage)[d];
  }
  // edge bytes
  const uint32_t head_end = min(d1 > d0 ? d0 * 4 : span_end, span_end);
  if (threadIdx.x < 8) {  // at most 6 bytes when the span holds no whole dword
    const uint32_t k = mis + threadIdx.x;
    if (k < head_end) dbase[k] = stage[k];
  } else if (threadIdx.x >= 64 && threadIdx.x < 68 && d1 > d0) {
    const uint32_t k = d1 * 4 + (threadIdx.x - 64);
    if (k < span_end) dbase[k] = stage[k];
  }
}

// ---- pass 4: entries of lines mode --------------------------------------------------
// Records [0, n_use) longer than K become entries (at most max_entries).  entry e spans
// records [entry_rec[e], entry_rec[e+1]) -- the records in between are too short to
// hold a k-mer.  result: {n_entry, stop_rec}: stop_rec = first record NOT consumed.
__global__ __launch_bounds__(1024) void ingest_entries_kernel(const uint64_t *rec_off, const uint64_t *hdr_pos,
                                                             uint32_t n_use, uint32_t K, uint32_t max_entries,
                                                             uint32_t *entry_rec, uint64_t *entry_hdr,
                                                             uint32_t *result) {
  __shared__ uint32_t s[1024];
  __shared__ uint32_t carry, stop;
  const uint32_t t = threadIdx.x;
  if (t == 0) { carry = 0; stop = n_use; }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < n_use; r0 += 1024) {
    const uint32_t r = r0 + t;
    const uint32_t flag = (r < n_use && rec_off[r + 1] - rec_off[r] > K) ? 1u : 0u;
    s[t] = flag;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
      const uint32_t o = t >= d ? s[t - d] : 0;
      __syncthreads();
      s[t] += o;
      __syncthreads();
    }
    const uint32_t idx = carry + s[t] - flag;
    if (flag) {
      if (idx < max_entries) {
        entry_rec[idx] = r;
        entry_hdr[idx] = hdr_pos[r];
      } else if (idx == max_entries) {
        stop = r;
      }
    }
    __syncthreads();
    if (t == 1023) carry += s[t];
    __syncthreads();
    if (carry > max_entries) break;  // uniform
  }
  if (t == 0) {
    const uint32_t n_entry = min(carry, max_entries);
    entry_rec[n_entry] = stop;
    result[0] = n_entry;
    result[1] = stop;
  }
}

}  // namespace

// ---- packed FASTA back to the file's bytes (nq_pack.h) ----------------------------------------
// One workgroup per kUnpackChunk raw bytes of a segment.  A periodic segment's byte r is '\n' at the end of
// every line of `width` bases and otherwise the letter of a 2-bit code; a raw segment is a copy.  Whole dwords
// are stored where the destination allows, the (at most 3 + 3) bytes around them singly: neighbouring chunks
// and segments never share a store.
__global__ __launch_bounds__(256) void unpack_kernel(const UnpackSeg *segs, uint32_t n_seg, const uint8_t *wire, uint8_t *raw) {
  uint32_t lo = 0, hi = n_seg;   // last segment whose first block is <= blockIdx.x (uniform)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (segs[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
  }
  const UnpackSeg s = segs[lo];
  const uint32_t tid = threadIdx.x;
  const uint64_t seg_len = s.width ? (uint64_t)s.count * (s.width + 1ull) : (uint64_t)s.count;
  const uint64_t r0 = (uint64_t)(blockIdx.x - s.first_block) * kUnpackChunk;
  if (r0 >= seg_len) return;
  const uint32_t nbytes = (uint32_t)(seg_len - r0 < kUnpackChunk ? seg_len - r0 : kUnpackChunk);
  uint8_t *dst = raw + s.dst + r0;
  const uint32_t head = (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) < nbytes ? (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u) : nbytes;
  const uint32_t n_words = (nbytes - head) / 4, tail0 = head + 4 * n_words;
  if (!s.width) {
    const uint8_t *src = wire + s.src + r0;
    if (tid < head) dst[tid] = src[tid];
    for (uint32_t i = tid; i < n_words; i += 256) {
      const uint8_t *p = src + head + 4 * i;
      *(uint32_t *)(dst + head + 4 * i) = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    }
    if (tid < nbytes - tail0) dst[tail0 + tid] = src[tail0 + tid];
    return;
  }
  const uint32_t W1 = s.width + 1u, wb = (s.width + 3u) / 4u;
  const uint64_t line0 = r0 / W1;
  const uint32_t col0 = (uint32_t)(r0 - line0 * W1);
  const uint8_t *src = wire + s.src + line0 * wb;
  constexpr uint32_t kLetters = 0x47544341u;   // 'A' 'C' 'T' 'G' by code
  auto byte_at = [&](uint32_t l, uint32_t c) -> uint32_t {
    return c == s.width ? (uint32_t)'\n' : (kLetters >> (8u * ((src[(uint64_t)l * wb + (c >> 2)] >> (2u * (c & 3u))) & 3u))) & 0xFFu;
  };
  if (tid < head) { const uint32_t rr = col0 + tid, l = rr / W1; dst[tid] = (uint8_t)byte_at(l, rr - l * W1); }
  for (uint32_t i = tid; i < n_words; i += 256) {
    const uint32_t rr = col0 + head + 4 * i;
    uint32_t l = rr / W1, c = rr - l * W1, w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      w |= byte_at(l, c) << (8 * k);
      if (++c == W1) { c = 0; ++l; }
    }
    *(uint32_t *)(dst + head + 4 * i) = w;
  }
  if (tid < nbytes - tail0) { const uint32_t rr = col0 + tail0 + tid, l = rr / W1; dst[tail0 + tid] = (uint8_t)byte_at(l, rr - l * W1); }
}

hipError_t launch_unpack(const UnpackSeg *segs, uint32_t n_seg, uint32_t n_blocks, const uint8_t *wire, uint8_t *raw, hipStream_t stream) {
  if (n_seg == 0 || n_blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(unpack_kernel, dim3(n_blocks), dim3(256), 0, stream, segs, n_seg, wire, raw);
  return hipGetLastError();
}

hipError_t launch_ingest_scan(const IngestArgs &a, hipStream_t stream) {
  if (a.n_chunks) hipLaunchKernelGGL(ingest_scan_kernel, dim3(a.n_chunks), dim3(kIB), 0, stream, a);
  if (a.n_files) {
    hipLaunchKernelGGL(ingest_files_kernel, dim3(a.n_files), dim3(64), 0, stream, a);
  }
  hipLaunchKernelGGL(ingest_bases_kernel, dim3(1), dim3(1024), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ingest_emit(const IngestArgs &a, hipStream_t stream) {
  if (a.n_chunks) hipLaunchKernelGGL(ingest_emit_kernel, dim3(a.n_chunks), dim3(kIB), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_ingest_entries(const uint64_t *rec_off, const uint64_t *hdr_pos, uint32_t n_use, uint32_t K,
                                 uint32_t max_entries, uint32_t *entry_rec, uint64_t *entry_hdr,
                                 uint32_t *result, hipStream_t stream) {
  hipLaunchKernelGGL(ingest_entries_kernel, dim3(1), dim3(1024), 0, stream, rec_off, hdr_pos, n_use, K,
                     max_entries, entry_rec, entry_hdr, result);
  return hipGetLastError();
}

}  // namespace nq
