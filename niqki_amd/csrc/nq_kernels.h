// nq_kernels.h -- launch interface between the C ABI (nq_api.hip) and the
// gfx950 kernels.  All pointers are device pointers unless noted.
#pragma once
#include "nq_common.h"

namespace nq {

// ---- sketch (nq_sketch.hip) -------------------------------------------------
struct SketchArgs {
  Derived d;
  const uint8_t *seqs;        // record bytes; nullptr = densify-only launch
  const uint64_t *rec_off;    // n_rec+1
  const uint32_t *entry_rec;  // n_entry+1 or nullptr (one record per sketch)
  int32_t *sketches;          // n_entry x F
  uint32_t splits;            // workgroups per sketch (>1: partial mins merged in global memory)
  uint32_t halves;            // set by launch_sketch: 2 when the F cells do not fit LDS (S = 16): every workgroup
                              // keeps one half of the slots and sees all k-mers; results merge in global memory
  uint32_t accumulate;        // start from the sketches already in `sketches`
  uint32_t densify;           // run densification before the store (splits == 1 only)
  uint32_t distinct;          // set by launch_sketch: densify over distinct values (short-read path)
  uint32_t filter;            // set by launch_sketch: candidate filter for long inputs
  uint32_t read_entries;      // set by launch_sketch: capacity of the short-read kernel's entry list (192 or 384)
  uint32_t *redo;             // short records: one flag per sketch, or nullptr (then a sketch with more occupied cells than the
                              // one-wavefront kernel's entry list takes the plain pass over all cells)
  uint32_t redo_only;         // != 0: this launch works only on the sketches whose flag has this value
  uint32_t redo_mark;         // != 0 (one-wavefront kernel): a sketch that exceeds the entry list gets this flag and is left
                              // to a later launch -- nothing of it is stored
  uint32_t window;            // set by launch_sketch: the short-read kernel's passes read their targets a window ahead
};
// avg_len: average input bytes per sketch (picks the launch shape)
hipError_t launch_sketch(const SketchArgs &a, uint32_t n_entry, uint64_t avg_len,
                         hipStream_t stream);
hipError_t launch_fill_u32(uint32_t *p, uint64_t n, uint32_t v, hipStream_t stream);
// 0: launch_sketch would not take the one-wavefront short-record kernel for this average length; else the capacity of
// the entry list it would take (192 or 384)
uint32_t sketch_read_list(const Derived &d, uint64_t avg_len);
constexpr uint64_t kSketchWorkgroupLen = 4097;   // an average length that makes launch_sketch take the workgroup kernel
// true: launch_sketch cannot densify inside the kernel for these parameters -- the caller fills the
// output with "empty" first, launches with densify = 0, then a densify-only launch (seqs = nullptr)
bool sketch_needs_merge(const Derived &d);

// ---- sketch store + inverted index (nq_index.hip) ---------------------------
// Sketch store: u16 [F_local][cap], slot major; 0xFFFF = empty/invalid cell.
// Inverted index over genome tiles (tile t = genomes [t*T, min((t+1)*T, N)), or, striped,
// the genomes g with g % n_tiles == t in ascending order):
//   entries  Entry [F_local][R][n_tiles]   bucket (slot, fp) of every tile side by
//                                          side, so ONE lookup serves all tiles
//   gids     u16, tile t at gids + tile_base[t]; a bucket starts at
//            (entry.start << align_log2) inside its tile and holds entry.len
//            tile-local genome ids, ascending.  With align_log2 = 6 every bucket
//            starts on a 128-byte line.
struct Entry {
  uint32_t start;  // in units of (1 << align_log2) ids
  uint32_t len;
};
struct IndexView {
  Derived d;
  uint32_t n_genomes;   // genomes of this segment
  uint32_t g_base;      // ... which are genomes [g_base, g_base + n_genomes) of the store / of the counter rows
                        // (0 for the main index; the delta segment of genomes inserted after the last full build)
  uint32_t tile;     // T, genomes per tile (multiple of 64, <= 65536)
  uint32_t n_tiles;
  uint32_t f_local;  // slot_end - slot_begin
  uint32_t align_log2;
  uint32_t padded;   // 128-byte aligned buckets whose unused tail positions hold padding ids: position p of
                     // a line holds id tile + 2p, which the gather kernel counts into 64 dummy words behind
                     // the tile's counters -- its bucket walk then needs no per-lane length test.  Every
                     // tile's id array ends with one line of padding only (the walk's "no chunk").
  uint32_t stripe;   // 0: tiles are ranges.  B = 1, 2, 4 .. 64: blocks of B consecutive genomes are dealt to the
                     // tiles round-robin (tile = (gid / B) % n_tiles): a run of related genomes is spread over
                     // all tiles, which keeps their buckets short in every tile
  uint64_t cap;      // row stride of the sketch store (genomes)
  const uint16_t *store;
  // query sketches handed to the query kernels: row q starts at sketches + q * q_stride, this
  // shard's first slot at + q_off (whole sketches: F and slot_begin; slot slices: f_local and 0)
  uint32_t q_stride, q_off;
  uint32_t accumulate;   // gather: add to the counter rows instead of overwriting them (slot pages after the first)
  const Entry *entries;
  const uint16_t *gids;
  const uint64_t *tile_base;   // n_tiles+1 (in ids), device
  const uint32_t *slot_units;  // n_tiles x (f_local+1): units before slot s of tile t
  const uint32_t *ptab = nullptr;   // [F_local][R][n_tiles] (start - first unit of the slot) << 16 | len, or nullptr
  // Single-tile indexes: bit c of hmask[s] = some bucket (s, fp) with fp >> hmask_shift == c is not empty
  // (hmask_shift = max(W - 4, 0): sixteen classes of the fingerprint range; with the regular H = 4 form a class
  // is a HyperLogLog part).  The gather kernel's own look-up tests it before it touches the table: a short
  // read's fingerprints (few leading zeros) fall into classes no indexed genome's minimum ever has, so a read
  // against a genome index skips ~99 % of its 2^S random table lines.  nullptr = no mask (look everything up).
  const uint16_t *hmask = nullptr;
  uint32_t hmask_shift = 0;
};
hipError_t launch_hmask(const IndexView &v, uint16_t *hmask, hipStream_t stream);

// genomes of tile t / global id of its i-th genome
NQ_HD uint32_t tile_count(const IndexView &v, uint32_t t) {
  if (v.stripe) {  // blocks of `stripe` genomes dealt round-robin
    const uint32_t nb = v.n_genomes / v.stripe, r = v.n_genomes % v.stripe;
    return (nb / v.n_tiles) * v.stripe + (t < nb % v.n_tiles ? v.stripe : 0u) + (t == nb % v.n_tiles ? r : 0u);
  }
  const uint32_t g0 = t * v.tile;
  return (v.n_genomes - g0) < v.tile ? (v.n_genomes - g0) : v.tile;
}
NQ_HD uint32_t tile_gid(const IndexView &v, uint32_t t, uint32_t i) {
  if (!v.stripe) return t * v.tile + i;
  const uint32_t sh = (uint32_t)__builtin_ctz(v.stripe);   // a power of two
  return (((i >> sh) * v.n_tiles + t) << sh) + (i & (v.stripe - 1u));
}

// sk_stride / sk_off: as q_stride / q_off of IndexView
hipError_t launch_store_insert(const Derived &d, const int32_t *sketches, uint32_t sk_stride, uint32_t sk_off,
                               uint32_t n, uint16_t *store, uint64_t cap, uint32_t first_gid,
                               hipStream_t stream);
hipError_t launch_store_read(const Derived &d, const uint16_t *store, uint64_t cap,
                             uint32_t begin, uint32_t n, int32_t *sketches, hipStream_t stream);
// Build, phase 1: units per (tile, slot) -> exclusive prefix per tile in
// slot_units, tile totals as a prefix (in ids) in tile_base.
hipError_t launch_build_sizes(const IndexView &v, uint32_t *slot_units, uint64_t *tile_base,
                              hipStream_t stream);
// Build, phase 2: entries + gids (gids sized from tile_base[n_tiles]).
hipError_t launch_build_fill(const IndexView &v, Entry *entries, uint16_t *gids, hipStream_t stream);
// padded layout: the padding pattern over the whole id array (n_ids a multiple of 64), before the fill
hipError_t launch_pad_fill(uint16_t *gids, uint64_t n_ids, uint32_t tile, hipStream_t stream);
constexpr uint32_t kPadWords = 64;        // dummy counter words of a padded tile
constexpr uint32_t kPadMaxTile = 65408;   // tile + 2 * 63 must fit 16 bits
// dump stream (src/niqki_index.cpp:42-55) of a whole-range index.
// layout: slot_word[s] (F+1 entries) = word position of bucket (s, 0) in the stream
// (header excluded); export: the words of slots [s0, s1) into `out` (word 0 = first
// word of slot s0).
hipError_t launch_export_layout(const IndexView &v, unsigned long long *slot_word, hipStream_t stream);
hipError_t launch_export(const IndexView &v, const unsigned long long *slot_word, uint32_t *out,
                         uint32_t s0, uint32_t s1, hipStream_t stream);
// inverse: walk the dump words of slots [s0, s0+n_slots) and write the sketch
// store.  slot_word: n_slots+1 word positions of the slots inside `words`.
hipError_t launch_import(const Derived &d, const uint32_t *words, const uint64_t *slot_word,
                         uint16_t *store, uint64_t cap, uint32_t n_genomes, uint32_t *bad,
                         uint32_t s0, uint32_t n_slots, hipStream_t stream);

// ---- query (nq_query.hip) ----------------------------------------------------
// gather-histogram: counts[q*stride + g] for all genomes (u16), one workgroup
// per (query, tile).
// stash: nq x (n_tiles-1) x f_local Entry scratch (unused for n_tiles == 1)
// order: nullptr, or the locality order of the batch from launch_order (nq <= 4096)
// pre: `stash` holds the packed words of launch_lookup for ALL tiles (sketches are then unused)
// counts2: second counter plane, needed (and used) when the view has more than kPassSlots slots
// (S = 16 on a whole-range handle): a count can then reach 2^16, so the slots are walked in two
// passes of <= 2^15 and plane p holds pass p's counters; the true count is the 32-bit sum.
constexpr uint32_t kPassSlots = 32768;
// Candidate output of a gather launch (multi-GPU path): while a query's counters leave LDS, the genomes
// with a count >= thr are appended to cand[q*cap ..] (unordered, at most cap kept) and n[q] grows by how
// many qualified.  The caller presets n to 0 and cand to -1; launches over further segments of the same
// index append.  Not with accumulate (slot pages) or a second plane.
struct CandOut {
  int32_t *cand = nullptr;
  int32_t *n = nullptr;
  uint32_t thr = 0, cap = 0;
  // Survivors (the sparse exchange without counter rows): every genome whose count reaches surv_thr
  // (<= thr) is appended as {genome id, count} to surv[q*surv_cap ..] (unordered, at most surv_cap kept)
  // and surv_n[q] grows by how many qualified.  With survivors the launch may run WITHOUT a counter row
  // (counts = nullptr): what another rank may ask about a genome that is not among them is answered from
  // the sketch store instead (nq_group.hip).
  int2 *surv = nullptr;
  int32_t *surv_n = nullptr;
  uint32_t surv_thr = 0, surv_cap = 0;
  // Hit lists (single-tile, single-segment indexes with small tiles: the short-read shape): the thresholded hits of
  // a query -- Index::query_sketch's result, src/niqki_index.cpp:662-666,:685 -- leave the kernel while its
  // counters are still in LDS, already ordered by descending (count, gid): hl[q*hl_cap + i] = count << 16 | gid
  // (a count <= 2^15, a genome id < 2^16 on such an index: the packed words order like the pairs) and hl_n[q] = how
  // many there are.  Only a query with more than hl_cap hits writes its 2N-byte counter row (to `counts`, which must
  // be given) and leaves hl_n[q] > hl_cap: launch_hitlist_emit thresholds and orders that row.
  // Not together with cand / surv.  hl_cap: a multiple of 4.
  uint32_t *hl = nullptr;
  uint32_t *hl_n = nullptr;
  uint32_t *hl_over = nullptr;   // nq + 1 words: [0] is zeroed by the launch (launch_hitlist_scan lists the overflowing queries there)
  uint32_t hl_cap = 0, hl_min = 0;
};
// the largest tile the hit-list form takes (the 256-thread launch shape: counters + queues + list within a CU's LDS
// several times over)
constexpr uint32_t kHitListMaxTile = 12288;
constexpr uint32_t kHitListMaxCap = 2048;
hipError_t launch_gather(const IndexView &v, const int32_t *sketches, uint32_t nq,
                         uint16_t *counts, uint16_t *counts2, uint64_t stride, Entry *stash, const uint32_t *order,
                         int variant, bool pre, hipStream_t stream, const CandOut &co = CandOut());
// out[i] = a[i] + b[i]: as u16 with wrap-around (the reference's uint16 matrix counters, src/niqki_index.cpp:572)
// or as u32
hipError_t launch_plane_add16(uint16_t *a, const uint16_t *b, uint64_t n, hipStream_t stream);
hipError_t launch_plane_sum32(const uint16_t *a, const uint16_t *b, uint32_t *out, uint64_t n, hipStream_t stream);
// Slot-major look-up pre-pass for the queries of a launch (nq_query.hip): fills
// pre[q][tile][slot] (lookup_pre_bytes of scratch) from the table streamed once.
bool launch_lookup_usable(const IndexView &v);
// The pre-pass reads a packed copy of the table (4 bytes per entry) where its row-staging kernel applies
bool lookup_wants_packed(const IndexView &v);
hipError_t launch_pack_entries(const IndexView &v, uint32_t *ptab, hipStream_t stream);
size_t lookup_pre_bytes(const IndexView &v, uint32_t nq);
hipError_t launch_lookup(const IndexView &v, const int32_t *sketches, uint32_t nq, uint32_t *pre, hipStream_t stream);
// launch shapes selectable through the "gather_variant" option (0 = choose)
bool gather_variant_valid(int variant);
// keys / order: nq words of scratch each
hipError_t launch_order(const IndexView &v, const int32_t *sketches, uint32_t nq, uint32_t *keys,
                        uint32_t *order, hipStream_t stream);
hipError_t launch_gathered(const IndexView &v, const int32_t *sketches, uint32_t nq,
                           unsigned long long *per_query, hipStream_t stream);

// threshold + compaction + order.  blk_counts: nq x n_blk scratch;
// hit_off nq+1 (u64); tmp_*: capacity-sized scratch for the sort.
struct HitsArgs {
  const uint16_t *counts;
  const uint16_t *counts2;   // second counter plane (S = 16) or nullptr: the count is the 32-bit sum
  uint64_t stride;
  uint32_t nq;
  uint32_t gid_begin, n_gids;
  uint32_t min_score;
  uint32_t *blk_counts;   // nq * n_blk
  uint32_t n_blk;
  unsigned long long *hit_off;  // nq+1
  uint32_t *hit_counts, *hit_gids;
  uint32_t *tmp_counts, *tmp_gids;
  uint64_t capacity;
};
constexpr uint32_t kHitsBlk = 4096;  // genomes per compaction block
hipError_t launch_candidates(const uint16_t *counts, uint64_t stride, uint32_t nq, uint32_t n_gids, uint32_t thr,
                             uint32_t cap, int32_t *cand, int32_t *n, hipStream_t stream);
hipError_t launch_hits_count(const HitsArgs &a, hipStream_t stream);
// After a gather launch with hit lists (CandOut::hl): launch_hitlist_scan turns the queries' hit counts n[0..nq) into
// the offsets a.hit_off (hit_off[nq] = total); launch_hitlist_emit copies every query's ordered list to its place in
// hit_counts / hit_gids, and thresholds + orders the counter row (a.counts) of a query whose list overflowed.
// over: nq + 1 words of scratch (the queries whose lists overflowed, made by the scan, taken by the emit launch)
hipError_t launch_hitlist_scan(const uint32_t *n, const HitsArgs &a, uint32_t hl_cap, uint32_t *over, hipStream_t stream);
hipError_t launch_hitlist_emit(const HitsArgs &a, const uint32_t *hl, uint32_t hl_cap, const uint32_t *over, hipStream_t stream);
hipError_t launch_hits_emit(const HitsArgs &a, hipStream_t stream);

// ---- FASTA / FASTQ framing (nq_ingest.hip) --------------------------------------
constexpr uint32_t kIngestBlock = 256;
constexpr uint32_t kIngestChunk = 8192;  // bytes per workgroup; chunks never span two files
struct IngestArgs {
  const uint8_t *raw;           // bytes of all files (16-byte aligned, >= 64 readable bytes after the end)
  const uint64_t *file_off;     // n_files+1 offsets into raw
  const uint8_t *file_type;     // n_files: 'A' (FASTA) or 'Q' (FASTQ)
  const uint32_t *chunk_first;  // n_files+1: first chunk of each file
  uint32_t n_files, n_chunks;
  uint32_t *summ;               // n_chunks x 5, pass 1 -> 2
  uint32_t *chunk_out;          // n_chunks x 4, pass 2 -> 3
  uint64_t *file_kept;          // n_files+1: sequence bytes before each file (after launch_ingest_scan)
  uint32_t *file_nrec;          // n_files+1: records before each file   (  "  )
  uint64_t *totals;             // {records, sequence bytes}
  uint8_t *seqs;                // out: record sequences back to back
  uint64_t *rec_off;            // out: n_rec+1 (entry n_rec is written by the caller)
  uint64_t *hdr_pos;            // out: raw offset of each record's header line
};
// passes 1+2 (census, per-file chaining, bases); then the caller sizes rec_off/hdr_pos
// from totals[0] and runs pass 3
hipError_t launch_ingest_scan(const IngestArgs &a, hipStream_t stream);
hipError_t launch_ingest_emit(const IngestArgs &a, hipStream_t stream);
// lines mode: records [0, n_use) longer than K -> entries (<= max_entries);
// result = {n_entry, first record not consumed}; entry_rec gets n_entry+1 values
hipError_t launch_ingest_entries(const uint64_t *rec_off, const uint64_t *hdr_pos, uint32_t n_use, uint32_t K,
                                 uint32_t max_entries, uint32_t *entry_rec, uint64_t *entry_hdr,
                                 uint32_t *result, hipStream_t stream);

// Packed FASTA (nq_pack.h) back to the files' bytes: segment k writes raw[dst ..) from wire[src ..); count / width as
// nqp::PackSeg (width 0: `count` bytes copied; else `count` lines of `width` bases + '\n' from 2-bit codes);
// first_block: exclusive prefix of ceil(raw length / kUnpackChunk) over the segments (n_blocks = its total).
constexpr uint32_t kUnpackChunk = 16384;
struct UnpackSeg {
  uint64_t dst, src;
  uint32_t count, width;
  uint32_t first_block, pad_;
};
hipError_t launch_unpack(const UnpackSeg *segs, uint32_t n_seg, uint32_t n_blocks, const uint8_t *wire, uint8_t *raw, hipStream_t stream);

// ---- gzip members inflated on the device (nq_inflate.hip) ------------------------------------------
// One job per gzip file: its bytes wire[src, src + src_len) become raw[dst, dst + cap), cap = the size the file's
// trailer announces.  status 0: every member decoded, CRC-32 and sizes agree, exactly cap bytes were written.  Any
// other status (1 header, 2 block type, 3 stored length, 4 code lengths, 5 symbol, 6 distance too far back, 7 more
// than cap bytes, 8 input ends early, 9 CRC, 10 member size, 11 bytes behind the last member, 12 fewer than cap
// bytes): the file is to be read some other way -- nothing was written outside [dst, dst + cap).
struct InflateJob {
  uint64_t src, src_len, dst, cap;
};
struct InflateOut {
  uint32_t status, members;
  uint64_t produced, consumed;
  uint32_t rounds, round_bytes, serial_tokens, blocks;   // how the file was decoded (nq_inflate.hip)
#ifdef NQ_INFLATE_CLOCK
  uint64_t clk[8];   // shader cycles by phase (diagnosis build only)
#endif
};
constexpr uint32_t kInflateXtabWords = 130;
uint32_t inflate_resident_files(bool small_ring);   // workgroups of a form of the kernel the current device keeps resident
void inflate_xtab(uint32_t *t);   // host: the CRC folding constants the kernel reads (kInflateXtabWords words)
// wire_bytes: readable bytes of `wire` (the kernel loads whole dwords up to there); raw: 16-byte aligned
// small_ring: -1 = choose by the number of files (the whole 32 KB window in LDS, four files per CU at a time, or its
// last 8 KB with far matches read back from the output, ten per CU), 0 / 1 = force (tests)
hipError_t launch_inflate(const InflateJob *jobs, uint32_t n_jobs, const uint8_t *wire, uint64_t wire_bytes, uint8_t *raw,
                          const uint32_t *xtab, InflateOut *outs, hipStream_t stream, int small_ring = -1);

// ---- synthetic genomes (nq_synth.hip) ----------------------------------------
hipError_t launch_synth(uint64_t seed, const uint32_t *family, const uint32_t *member,
                        const uint32_t *rate14, uint32_t n, uint64_t len, uint64_t stride,
                        uint8_t *out, hipStream_t stream);

hipError_t launch_synth_reads(uint64_t seed, const uint32_t *family, const uint32_t *member, const uint32_t *rate14,
                              const uint64_t *offset, const uint32_t *read_id, uint32_t read_rate14, uint32_t n,
                              uint32_t len, uint64_t stride, uint8_t *out, hipStream_t stream);

// ---- integer-ALU ceiling probes (nq_sketch.hip; measurement support) ----------------------------
// what = 0: independent 32-bit adds; 1: 32-bit multiplies (v_mul_lo_u32); 2: the sketch kernel's
// per-k-mer arithmetic alone (roll, canonical choice, high word of the filter hash, K = 31) with no
// LDS, memory or compaction; 3: v_lshl_add_u32 (the issue class of most vector opcodes).
// *units = adds / multiplies / k-mers / instructions executed; timed by the caller.
hipError_t launch_alu_probe(int what, uint32_t iters, uint32_t *sink, uint64_t *units, hipStream_t stream);
// dst[0, bytes) = src[0, bytes) by a plain grid-stride kernel (bytes a multiple of 16)
hipError_t launch_copy_probe(const void *src, void *dst, uint64_t bytes, hipStream_t stream);

}  // namespace nq
