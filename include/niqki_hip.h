/*
 * niqki_hip.h -- C ABI of the MI355X-native NIQKI sketch/query engine
 * (libniqki_hip.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * has no FFI layer: its boundary is the C++ class Index
 * (/root/reference/src/niqki_index.h:35-213) whose inner operators
 *
 *     void         compute_sketch(const string&, vector<int32_t>&) const;   niqki_index.h:103
 *     void         insert_sketch(const vector<int32_t>&, uint32_t gid);      niqki_index.h:108
 *     query_output query_sketch(const vector<int32_t>&) const;               niqki_index.h:142
 *
 * are what the file drivers call per record.  Each entry point below names the
 * reference interface (file:line under /root/reference) it replaces.  All
 * signatures are plain C: opaque handle, pointers and sizes, int status.
 * No exceptions cross the boundary.  Calls on one handle must be serialised by
 * the caller (one handle per GPU; the reference's const query methods map to
 * batched calls here, not to concurrent ones) -- except the niqki_*_shared entry
 * points, which any number of host threads may call on one handle at the same time
 * (the reference's `omp parallel` record loops, kept as they are).
 *
 * Memory spaces: every array argument of a call lives in the space named by
 * the call's `mem` argument -- NIQKI_MEM_HOST (pageable or pinned host memory;
 * the library stages it) or NIQKI_MEM_DEVICE (device memory of the handle's
 * GPU; nothing is copied, work is enqueued on the handle's stream and the call
 * returns without synchronising unless stated otherwise).
 *
 * Sketch layout: F = 2^S int32 per sketch, row major, -1 = empty cell, exactly
 * the reference's vector<int32_t>. *
 * Not here: measurement, diagnosis and test support (synthetic inputs, per-kernel timers, the steps of the sparse
 * multi-GPU exchange one by one ...) -- include/niqki_hip_bench.h.
 *
 * Environment variables the LIBRARY reads (none changes a result; options of niqki_set_option are the way to set
 * anything per handle):
 *   NIQKI_GROUP_TRANSPORT   rccl | ipc | local: transport of niqki_group_create / niqki_group_new_id (see "one index
 *                           over several GPUs" below); default: RCCL, device copies for shards that share a device
 *   NIQKI_IPC_WORDS         host | coarse: where an ipc rank keeps its sequence words (default: fine-grained device
 *                           memory, the shared host block where that cannot be exported); NIQKI_IPC_ARENA=coarse:
 *                           plain device memory for its exchange buffers.  Test switches.
 *   NIQKI_LOOKUP_PREPASS    -1 | 0 | 1: initial value of option "lookup_prepass" of every handle (the test suite runs
 *                           under both look-up paths this way)
 *   NIQKI_TILE_STRIPE       overrides option "tile_stripe" at every index build (tests of the tile dealing)
 *   NIQKI_HMASK=0           builds single-tile indexes without their per-slot class mask (stat "class_mask")
 *   NIQKI_SKETCH_WAVE=0     short records take the workgroup sketch kernel instead of the one-wavefront one;
 *   NIQKI_SKETCH_FILTER     0 = long records are sketched without the candidate filter, n >= 2 = with a fixed filter
 *                           strength (default: chosen per sketch).  Test switches of nq_sketch.hip's launch shapes.
 *   NIQKI_DENSIFY_WINDOW=0  the one-wavefront kernel's densification passes propose from every entry in every pass
 *                           instead of reading their targets a window ahead (the A/B figure in profiles/)
 *   NIQKI_DENSIFY_TAIL=0    ... and run to the end instead of filling their last 16 empty cells in closed form
 * The `niqki` host program reads NIQKI_HOST_THREADS (reader threads; default: the CPUs the process may use),
 * NIQKI_HOST_TIMING (phase times on stderr), NIQKI_HOST_NO_PACK (plain FASTA files travel as their bytes),
 * NIQKI_HOST_NO_GPU_INFLATE (gzip files are always inflated by the reader threads), NIQKI_HOST_GPU_INFLATE_MIN (how many
 * gzip files a list must hold for the device to inflate them; default 32 per reader thread),
 * NIQKI_HOST_ZLIB_ONLY (no libdeflate), NIQKI_SHARDS_ON_ONE_DEVICE (--gpus N on one device: tests).
 */
#ifndef NIQKI_HIP_H
#define NIQKI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIQKI_ABI_VERSION 2   /* 2: niqki_raw_batch.file_status, NIQKI_E_GZIP, niqki_sketch_ahead / niqki_query_ahead */

/* Bytes the caller must keep readable after the last sequence byte of a
 * NIQKI_MEM_DEVICE sequence buffer (the sketch kernel reads whole dwords). */
#define NIQKI_SEQ_PAD 64

enum niqki_status {
  NIQKI_OK = 0,
  NIQKI_E_INVALID = 1,     /* bad argument / unsupported parameter combination */
  NIQKI_E_NOMEM = 2,       /* host or device allocation failed */
  NIQKI_E_HIP = 3,         /* a HIP runtime call failed; see niqki_last_error */
  NIQKI_E_CAPACITY = 4,    /* caller buffer too small; sizes were still reported */
  NIQKI_E_STATE = 5,       /* call not valid in the handle's current state */
  NIQKI_E_NODEVICE = 6,    /* no usable gfx950 device */
  NIQKI_E_GZIP = 7         /* niqki_stage_raw: a file flagged NIQKI_FILE_GZIP is not a plain intact gzip file of the
                              size its trailer states (see niqki_raw_batch.file_status); nothing was staged */
};

enum niqki_mem { NIQKI_MEM_HOST = 0, NIQKI_MEM_DEVICE = 1 };

typedef struct niqki_index niqki_index; /* opaque */

/* Constructor arguments: Index(lF,K,W,H,filename,min_fract),
 * src/niqki_index.cpp:13-38 (output-file handling stays in the host program).
 * Supported: 1<=K<=31 (K=32 is UB in the reference, :28-29), 1<=S<=16,
 * H<=W<=15, S+W<=30.  S = 16 is the reference's lF>15 branch (uint32 counters, :668-682): a
 * count can reach 2^16, so the u16 counter calls (niqki_query_counts, niqki_hits_from_counts)
 * refuse it on a whole-range handle; niqki_query* / niqki_staged_query / niqki_query_counts32 are
 * exact (also on a paged handle, whose pages add up per half of the slots), and so are groups of
 * two or more shards (a shard counts at most 2^15 slots, the cross-shard sums travel as u32)
 * and niqki_matrix_range wraps like the reference's uint16 matrix counters (:572). */
typedef struct niqki_params {
  uint32_t K;          /* k-mer length */
  uint32_t S;          /* lF: log2 of the number of sketch slots */
  uint32_t W;          /* fingerprint bits */
  uint32_t H;          /* HyperLogLog bits inside the fingerprint */
  uint32_t min_score;  /* hits need count >= min_score (niqki_min_score) */
  uint32_t slot_begin; /* this shard owns sketch slots [slot_begin, slot_end); */
  uint32_t slot_end;   /*   0,0 means the whole range [0, 2^S)                 */
  int32_t device;      /* HIP device ordinal; -1 = current device */
  uint32_t tile_genomes; /* genomes per counter tile (0 = choose); see DESIGN.md */
  uint32_t resident_mib; /* > 0: paged index, as option "resident_bytes" (in MiB) from the start --
                            the way to load a dump into a paged handle */
  uint32_t reserved[2];
} niqki_params;

int niqki_abi_version(void);
const char *niqki_status_string(int status);

/* min_score = (uint32)(min_fract * 2^S): src/niqki_index.cpp:21-22 */
uint32_t niqki_min_score(double min_fract, uint32_t S);

/* Index::Index / Index::~Index: src/niqki_index.cpp:13-38, :106-109.
 * Fails with NIQKI_E_NODEVICE when there is no GPU: there is no CPU path. */
int niqki_create(const niqki_params *params, niqki_index **out);
void niqki_destroy(niqki_index *ix);
/* Index::select_best_H(genome_size) (the CLI's -G): src/niqki_index.cpp:126-164.
 * Picks H in 2..6 from the expected genome size and replaces H and M = W-H on
 * the handle.  As in the reference, the fingerprint's low-part mask and
 * saturation constant keep the values the constructor derived from the
 * ORIGINAL H (:24-25 are not recomputed), so the parts of a fingerprint may
 * overlap or exceed W bits; cells outside [0, 2^W) are sketched but never
 * indexed or queried (:364, :654).  May be called at any time, like the
 * reference's member; *H_out (may be NULL) receives the chosen H.
 * NIQKI_E_INVALID when the chosen H exceeds W (the reference's unsigned M
 * wraps there). */
int niqki_select_best_H(niqki_index *ix, double genome_size, uint32_t *H_out);
/* Text of the last error on the handle; with ix == NULL, why the last
 * niqki_create / niqki_import_dump of the calling thread failed. */
const char *niqki_last_error(const niqki_index *ix);
int niqki_get_params(const niqki_index *ix, niqki_params *out);

/* Work is enqueued on this hipStream_t (default: a stream the handle owns). */
int niqki_set_stream(niqki_index *ix, void *hip_stream);
void *niqki_get_stream(const niqki_index *ix);
int niqki_synchronize(niqki_index *ix);

/* Tuning knobs (no reference counterpart; none of them changes a result):
 * "gather_variant" (launch shape of the gather kernel, 0 = choose .. 5), "query_batch",
 * "tile_genomes" (multiple of 64, <= 65536; takes effect at the next build),
 * "bucket_align_log2" (-1 = choose, 0..6: buckets start on multiples of 2^a ids),
 * "tile_stripe" (B = 1, 2, 4 .. 64, default 32: with several tiles, blocks of B consecutive
 * genomes are dealt to the tiles round-robin, so a run of related genomes is spread over all
 * tiles; blocks of 32 are 64 bytes of a counter row, the smallest piece HBM writes without
 * a read-modify-write; 0 = tiles are ranges of genome ids),
 * "min_score", "record_len_hint" (average bytes per sketch of NIQKI_MEM_DEVICE
 * batches, so that niqki_sketch need not read rec_off back to pick a launch
 * shape; 0 = read it back), "query_order" (1 = default: the queries of a launch
 * are processed in an order that puts similar ones on the same XCD where that pays --
 * >= 16384 genomes and >= 8192 slots on the handle; 2 = on every index; results are
 * unaffected; 0 = input order), "lookup_prepass" (1 = the index table is walked
 * once per launch, slot block by slot block, for all its queries instead of one random
 * table line per query and slot inside the gather kernel, wherever the index shape
 * allows: 16 % less HBM traffic.  The default (-1) uses it where it is also faster: batches of
 * >= 1024 queries on indexes of 1 or 2 counter tiles with W <= 12 (whole table rows are then
 * streamed through LDS from a packed copy of the table, +4 bytes per table entry: 8 % faster), and
 * indexes of more than 4 tiles, > 261 632 genomes (25 % faster); 0 = never),
 * "incremental_build" (1 = default: genomes inserted after a build get a delta index of their own
 * -- a query walks both -- until they pass an eighth of the main index, then everything is rebuilt;
 * 0 = every insert after a query rebuilds the whole index at the next query),
 * "resident_bytes" (indexes beyond the memory one wants to give them -- or beyond HBM: with a
 * value > 0, set before the first insert, the sketch store (2 bytes per genome and slot) lives
 * in page-locked host memory and the inverted index is built for one PAGE of slots at a time,
 * sized so that the page's store rows + index stay within the value; a query batch walks the
 * pages and the gather kernel accumulates the hit counters, which are sums over slots.  Same
 * answers as a resident index.  Insert, the dump import (niqki_params.resident_mib), every query
 * call, niqki_get_sketches, niqki_matrix_range (the stored sketches are read from the host store)
 * and the dump export (page after page) work on a paged handle; niqki_query_gathered, the
 * candidate / survivor calls and groups do not (NIQKI_E_STATE / NIQKI_E_INVALID)),
 * "stream_priority" (1 / 0 / -1: the handle gets a stream of its own, made at the top / default / bottom of
 * the device's stream priority range -- also after niqki_set_stream, whose stream stays the caller's --, so
 * that e.g. a query handle's short kernels are dispatched ahead of another handle's long sketch kernel;
 * niqki_get_stream returns it),
 * "sketch_lane_cus" (0 = default: all; n: the lane of niqki_sketch_ahead is made anew on the TOP n compute units of the
 * device's numbering (hipExtStreamCreateWithCUMask) -- for a caller that fences the two lanes, with its own stream
 * limited to the other units; on one MI355X the split is no faster than the hardware's own dispatch),
 * "hit_lists" (1 = default: on an index of ONE counter tile of at most 12 288 genomes -- the short-read shape of
 * src/niqki_index.cpp:412-430 -- niqki_query* / niqki_staged_query take a query's thresholded hits out of the gather
 * kernel while its counters are in LDS, already ordered, instead of writing a 2N-byte counter row per query and reading
 * it again; 0 = always through counter rows), "hit_list_cap" (1..2048, default 256: hits per query such a list holds; a
 * query with more -- min_score 0: every genome -- leaves through its counter row and is ordered from there),
 * "inflate_window" (-1 = default: a launch of the device inflate, NIQKI_FILE_GZIP below, keeps each file's whole 32 KB
 * window in LDS -- four files per CU at a time -- unless it holds more files than that runs at once; then only the
 * window's last 8 KB, ten files per CU, and matches that reach further back read the file's own flushed output;
 * 0 / 1 = always the first / second form.  Same bytes either way). */
int niqki_set_option(niqki_index *ix, const char *key, int64_t value);

/* Pre-sizes the sketch store for n_genomes (optional; the store grows). */
int niqki_reserve(niqki_index *ix, uint32_t n_genomes);

/* Index::compute_sketch (src/niqki_index.cpp:335-358) including
 * sketch_densification (:313-331), batched.
 *   seqs        concatenated record bytes (ASCII, no newlines)
 *   rec_off     n_rec+1 byte offsets into seqs
 *   entry_rec   n_entry+1 record indices: sketch e accumulates records
 *               [entry_rec[e], entry_rec[e+1]) (whole-file mode,
 *               src/niqki_index.cpp:442-456); NULL = one record per sketch
 *               (lines mode, :383-408) and then n_entry must equal n_rec
 *   sketches    n_entry x F int32 out
 * Records with length <= K contribute nothing (:395,:450).  Densification
 * runs once per sketch after all its records (the reference's per-record
 * densification hangs on multi-record files).  Where
 * the reference would never terminate (B.2) the remaining cells stay -1. */
int niqki_sketch(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off,
                 uint32_t n_rec, const uint32_t *entry_rec, uint32_t n_entry,
                 int32_t *sketches, int mem);

/* Index::sketch_densification alone (src/niqki_index.cpp:313-331), in place. */
int niqki_densify(niqki_index *ix, int32_t *sketches, uint32_t n, int mem);

/* Index::insert_sketch (src/niqki_index.cpp:362-370) for n sketches; they get
 * genome ids genome_count .. genome_count+n-1 in order, like the id counter of
 * :396-401 / :479-490 does single-threaded. */
int niqki_insert(niqki_index *ix, const int32_t *sketches, uint32_t n, int mem);

/* Number of inserted genomes: Index::getNbGenomes, src/niqki_index.h:138-140 */
uint32_t niqki_genome_count(const niqki_index *ix);

/* Builds the device-resident inverted index (CSR per genome tile) from
 * everything inserted so far.  Replaces the reference's incremental
 * vector<gid> Buckets[] appends (src/niqki_index.h:55, :365-367).  Idempotent;
 * the query calls run it when inserts are pending. */
int niqki_build(niqki_index *ix);

/* Counting half of Index::query_sketch (src/niqki_index.cpp:652-661): for
 * query q, counts[q*stride + g] = number of this shard's slots whose bucket
 * holds genome g.  uint16 counters like the reference's lF<=15 branch.
 * stride >= genome_count (in elements), even; a NIQKI_MEM_DEVICE `counts` must be
 * 4-byte aligned (rows are written as packed u16 pairs).  Fastest when every row
 * starts on a 128-byte line: NIQKI_ROW_STRIDE(genome_count) and a 128-byte aligned
 * buffer (a partial 64-byte block costs HBM a read-modify-write).  This is the
 * per-genome hit vector the multi-GPU path sums across slot shards. */
#define NIQKI_ROW_STRIDE(n) ((((uint64_t)(n)) + 63u) & ~(uint64_t)63u)
int niqki_query_counts(niqki_index *ix, const int32_t *sketches, uint32_t nq,
                       uint16_t *counts, uint64_t stride, int mem);

/* The same with uint32 counters, exact for every S (the reference's lF>15 branch,
 * src/niqki_index.cpp:668-677). */
int niqki_query_counts32(niqki_index *ix, const int32_t *sketches, uint32_t nq,
                         uint32_t *counts, uint64_t stride, int mem);

/* Threshold + order half of Index::query_sketch (src/niqki_index.cpp:662-666,
 * :685): from (possibly cross-shard summed) counters of genomes
 * [gid_begin, gid_begin+n_gids) produce, per query, the (count,gid) pairs with
 * count >= min_score sorted by descending (count, gid).
 *   hit_off     nq+1 exclusive prefix offsets (always exact, even on
 *               NIQKI_E_CAPACITY)
 *   hit_counts, hit_gids   capacity entries each
 * Synchronises the stream when mem == NIQKI_MEM_HOST. */
int niqki_hits_from_counts(niqki_index *ix, const uint16_t *counts, uint32_t nq,
                           uint64_t stride, uint32_t gid_begin, uint32_t n_gids,
                           uint64_t *hit_off, uint32_t *hit_counts,
                           uint32_t *hit_gids, uint64_t capacity, int mem);

/* ---- many host threads on one handle ------------------------------------------------------
 * A handle is single-caller (see the top of this file).  The reference's drivers, however, call
 * compute_sketch / insert_sketch / query_sketch from every thread of an `omp parallel` region on
 * one Index (src/niqki_index.cpp:391-401, :415-428, :479-490, :525-538).  These four entry points
 * keep that contract: any number of host threads may call them on one handle at the same time
 * (and ONLY them, while such calls are in flight).  Concurrent callers are combined into batches --
 * the first thread to arrive leads a batch, threads that arrive while it is on the GPU form the
 * next one -- and each batch runs through niqki_sketch / niqki_insert / niqki_query
 * (NIQKI_MEM_HOST): same results, about one launch per `threads` records instead of one per record.
 *   niqki_sketch_shared           Index::compute_sketch of one record (len bytes) -> sketch[2^S]
 *   niqki_insert_shared           Index::insert_sketch; *genome_id (may be NULL) = the id it got:
 *                                 ids are handed out in arrival order, as the reference's critical
 *                                 section does with several threads (:396-401, :486-490)
 *   niqki_query_shared            Index::query_sketch: *n_hits = number of hits, the first
 *                                 min(*n_hits, capacity) written in the reference's order
 *   niqki_query_sequence_shared   Index::query_sequence (:691-695)
 * (niqki_shared_stats, niqki_hip_bench.h: batches run so far, requests served, the largest batch.) */
int niqki_sketch_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, int32_t *sketch);
int niqki_insert_shared(niqki_index *ix, const int32_t *sketch, uint32_t *genome_id);
int niqki_query_shared(niqki_index *ix, const int32_t *sketch, uint64_t *n_hits, uint32_t *hit_counts,
                       uint32_t *hit_gids, uint64_t capacity);
int niqki_query_sequence_shared(niqki_index *ix, const uint8_t *seq, uint64_t len, uint64_t *n_hits,
                                uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity);

/* Index::query_sketch (src/niqki_index.cpp:633-687), batched: both halves. */
int niqki_query(niqki_index *ix, const int32_t *sketches, uint32_t nq,
                uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
                uint64_t capacity, int mem);

/* Index::query_sequence (src/niqki_index.cpp:691-695), batched: sketch +
 * query.  Arguments as niqki_sketch / niqki_query. */
int niqki_query_sequences(niqki_index *ix, const uint8_t *seqs,
                          const uint64_t *rec_off, uint32_t n_rec,
                          const uint32_t *entry_rec, uint32_t n_entry,
                          uint64_t *hit_off, uint32_t *hit_counts,
                          uint32_t *hit_gids, uint64_t capacity, int mem);

/* The same in two halves, so that consecutive batches overlap on the device: niqki_sketch_ahead starts the sketch
 * kernel of a batch on a lane of the handle's own (a side stream) and returns; niqki_query_ahead runs the query of the
 * OLDEST batch sketched ahead on the handle's stream, behind that batch's sketch kernel.  A caller that keeps one batch
 * ahead --
 *     niqki_sketch_ahead(batch 0);
 *     for i: niqki_sketch_ahead(batch i + 1); niqki_query_ahead(hits of batch i);
 * -- has batch i + 1's sketch kernel (bound by vector instruction issue) running beside batch i's gather and hit
 * kernels (bound by HBM): 164 k against 158 k query genomes/s at the 100 000-genome shape.  At most two
 * batches are ahead at a time (NIQKI_E_STATE beyond).  Device memory only, and the one call whose device inputs are NOT
 * taken in the order of the handle's stream: `seqs`, `rec_off` and `entry_rec` are read on the sketch lane, so they must
 * be complete when niqki_sketch_ahead is called and stay untouched until the niqki_query_ahead that takes the batch
 * has been called.  The results of niqki_query_ahead are those of niqki_query_sequences on the same records
 * and, as there, in the handle's stream order for NIQKI_MEM_DEVICE outputs.  `n_entry` (may be NULL) receives the batch's
 * entry count, `sketches` (may be NULL; n_entry x 2^S cells in `mem`) a copy of its sketches.  NIQKI_E_CAPACITY (host
 * outputs) leaves the batch the oldest one: the same call again with larger arrays.  niqki_synchronize waits for the
 * sketch lane too. */
int niqki_sketch_ahead(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec,
                       const uint32_t *entry_rec, uint32_t n_entry, int mem);
int niqki_query_ahead(niqki_index *ix, uint32_t *n_entry, uint64_t *hit_off, uint32_t *hit_counts,
                      uint32_t *hit_gids, uint64_t capacity, int32_t *sketches, int mem);

/* ---- raw file bytes in: FASTA / FASTQ framing on the GPU -------------------
 * Index::Biogetline (src/niqki_index.cpp:890-941) and the read loops of
 * insert_file_whole / query_file_whole (:442-456, :505-519) and
 * insert_file_lines / query_file_lines (:383-430): the caller hands over the
 * bytes of its files (gzip files as they are, see NIQKI_FILE_GZIP below, or inflated), the records are framed on the device and stay
 * there as the "staged batch" of the handle; niqki_staged_* then sketch, insert
 * or query it.  Framing is the reference's: FASTA = the first line of a file and
 * every line starting with '>' is a header, all other lines are concatenated
 * (nothing trimmed or upper-cased); FASTQ = 4-line records.  A record of at most
 * K bases is skipped (:395,:423,:450,:512).  The staged batch stays usable until the
 * next niqki_stage_raw or the next HOST-memory niqki_sketch / niqki_query_sequences
 * on the handle (they share its staging buffers; niqki_staged_* then report
 * NIQKI_E_STATE); every other call leaves it alone. */
typedef struct niqki_raw_batch {
  const uint8_t *raw;        /* bytes of n_files files back to back, in the memory space of the
                                call (device: 4-byte aligned, NIQKI_SEQ_PAD readable bytes after) */
  const uint8_t *const *file_ptr; /* optional (host memory space only): HOST array of n_files host
                                pointers, file f's bytes start at file_ptr[f]; raw is then ignored */
  const uint64_t *file_off;  /* HOST array, n_files+1 offsets into raw (their differences are the
                                file sizes when file_ptr is used) */
  const uint8_t *file_type;  /* HOST array, n_files: 'A' FASTA / 'Q' FASTQ (get_data_type, :944-952); 'a' = a FASTA
                                file handed over as the container niqki_pack_fasta made of it (host memory space, the
                                file_ptr form, whole mode): 2 bits per base across PCIe, the device writes the file's
                                own bytes back before it frames them -- same results as 'A' on the file itself;
                                'A' | NIQKI_FILE_GZIP, 'Q' | NIQKI_FILE_GZIP = the file as it lies on disk, gzip'd (host
                                memory space, the file_ptr form, whole mode): inflated on the device, see below */
  uint32_t n_files;
  uint32_t lines;            /* 0: one sketch per file (whole mode); 1: one sketch per record longer
                                than K (lines mode, n_files must be 1) */
  uint32_t final;            /* lines mode: raw reaches the end of the file; otherwise the last
                                (possibly incomplete) record is left for the next call */
  uint32_t max_entries;      /* lines mode: stop after this many sketches */
  uint8_t *file_status;      /* optional HOST array, n_files, written when the call returns NIQKI_E_GZIP: 0, or why the
                                device would not inflate gzip file f (1..13, nq_kernels.h InflateJob) */
} niqki_raw_batch;

/* Gzip files inflated on the device.  The reference reads every input through zstr::ifstream (src/zstr.hpp:190-203,
 * :236-239: gzip by magic, zlib member after member); a file flagged NIQKI_FILE_GZIP crosses PCIe as it lies on disk
 * (a quarter of its FASTA bytes) and one wavefront per file writes its bytes where the framing expects them.  The
 * device takes what is plainly a gzip file: members that inflate without any irregularity, CRC-32 and ISIZE of every
 * member right, and either ONE member as long as the file's last four bytes say (the usual case), or a file whose
 * members all carry their size -- BGZF (bgzip / htslib: the 'B' 'C' extra subfield) or this project's 'N' 'Q' tag --,
 * which is cut into its members and inflated by one wavefront per member.  For anything else -- damaged or truncated
 * streams, plainly concatenated members, bytes behind the last member, sizes of 2 GiB and more -- niqki_stage_raw stages nothing,
 * returns NIQKI_E_GZIP and names the files in file_status; the caller inflates those itself (zlib decides what they
 * yield, as before) and hands them over as 'A' / 'Q'.  The kernel is at least as strict as zlib's inflate, so a file
 * it accepts has zlib's bytes. */
#define NIQKI_FILE_GZIP 0x80

typedef struct niqki_stage_info {
  uint32_t n_entry;    /* sketches the staged batch produces */
  uint32_t n_rec;      /* records framed, short ones included */
  uint64_t consumed;   /* raw bytes covered by the entries: a lines-mode stream resumes here
                          (always the first byte of a header line, or the end) */
  uint64_t seq_bytes;  /* sequence bytes staged */
} niqki_stage_info;

/* entry_hdr (HOST array of max_entries, lines mode, may be NULL): raw offset of the
 * header line of each entry (names are the header lines, :395-400). */
int niqki_stage_raw(niqki_index *ix, const niqki_raw_batch *batch, int mem_space,
                    niqki_stage_info *info, uint64_t *entry_hdr);
/* Starts the host-to-device copy of the NEXT batch's file bytes (file_ptr form, whole mode) on a
 * copy stream of the handle and returns without waiting: the copy runs beside the kernels of the
 * batch staged now (the read loops of :461-500 / :523-540 with the transfer of file i+1 under the
 * work on file i).  A niqki_stage_raw(NIQKI_MEM_HOST) whose batch names the same file_ptr[] and
 * file_off[] takes these bytes instead of copying.  Two prefetches may be on their way at a time (the
 * batch about to be staged and the one behind it: a third takes the older one's place); staging
 * one of them leaves the other alone, staging any other batch drops both.  The
 * caller keeps the host bytes valid until that call returns (page-locked memory, or the copy is
 * not asynchronous). */
int niqki_stage_raw_prefetch(niqki_index *ix, const niqki_raw_batch *batch);
/* compute_sketch of every staged entry (n_entry x 2^S int32). */
int niqki_staged_sketch(niqki_index *ix, int32_t *sketches, int mem_space);
/* ... + insert_sketch: ids follow the entries (:396-401, :479-490). */
int niqki_staged_insert(niqki_index *ix);
/* ... + query_sketch per entry; outputs as niqki_query. */
int niqki_staged_query(niqki_index *ix, uint64_t *hit_off, uint32_t *hit_counts,
                       uint32_t *hit_gids, uint64_t capacity, int mem_space);
/* Packed FASTA: what a host-to-device copy of whole genomes costs is 1 byte per base; a FASTA file is almost
 * entirely full lines of one width holding A, C, G, T only.  niqki_pack_fasta turns the bytes of a file into a
 * container in which every run of such lines travels as 2 bits per base and everything else -- header lines, lines
 * with any other byte, a last line without its newline -- verbatim; staged with file_type 'a', the device restores
 * the file's exact bytes (so Index::Biogetline's framing, src/niqki_index.cpp:890-941, sees what it would have seen).
 * Host code, no device needed.  niqki_pack_fasta returns the container's size, or 0 when the file is not worth
 * packing (reads, FASTQ, CRLF lines: hand it over raw); capacity >= niqki_pack_bound(n).  niqki_unpack_fasta is the
 * host restatement of the device pass (raw == NULL: only *raw_len). */
size_t niqki_pack_bound(size_t n);
size_t niqki_pack_fasta(const uint8_t *raw, size_t n, uint8_t *out, size_t capacity);
int niqki_unpack_fasta(const uint8_t *container, size_t len, uint8_t *raw, size_t capacity, size_t *raw_len);
/* Page-locked host memory for raw batches (plain malloc'ed memory works too, slower). */
void *niqki_host_alloc(size_t bytes);
void niqki_host_free(void *p);

/* Counting loop of Index::query_range (src/niqki_index.cpp:570-597): for
 * target genomes t in [begin,end), counts[(t-begin)*stride + a] = number of
 * buckets holding both a and t (the reference's counts[a*batch + t-begin];
 * the matrix is symmetric).  uint16 like :572. */
int niqki_matrix_range(niqki_index *ix, uint32_t begin, uint32_t end,
                       uint16_t *counts, uint64_t stride, int mem);

/* dump_index_disk payload (src/niqki_index.cpp:42-55), before gzip and
 * without the trailing names: 6 x u32 header {lF,K,H,W,min_score,N} then per
 * bucket u32 size + size x u32 gid, buckets in fp + slot*2^W order, gids
 * ascending.  Call with buf == NULL to get the size.  Host memory only;
 * whole-range handles only. */
int niqki_export_dump(niqki_index *ix, uint8_t *buf, uint64_t capacity,
                      uint64_t *size);

/* Loading constructor (src/niqki_index.cpp:63-90) from the gunzipped bytes of
 * a dump (names excluded; *consumed = offset of the first name byte).  The
 * parameters stored in the dump override those given (like :67-72), except
 * device / tile_genomes / slot range, which are taken from `params`: a handle
 * created with a slot range keeps only its own slots of the dump (one shard of
 * a multi-GPU index), the bytes of the other slots are walked and skipped. */
int niqki_import_dump(const niqki_params *params, const uint8_t *buf,
                      uint64_t len, uint64_t *consumed, niqki_index **out);

/* Streaming forms of the two calls above, for dumps that should not sit in one
 * buffer (13.6 GB at 100k genomes): the 24-byte header, the byte position of
 * every slot's first bucket in the payload (2^S + 1 entries, header excluded),
 * and the payload of a range of whole slots. */
/* On a slot-range handle (one shard of a multi-GPU index) the layout and the slot numbers of
 * niqki_export_dump_slots are relative to the shard's first slot (2^S -> slot_end - slot_begin):
 * the shards' payloads in rank order are the dump of the whole index. */
int niqki_export_dump_header(niqki_index *ix, uint8_t header[24]);
int niqki_export_dump_layout(niqki_index *ix, uint64_t *slot_bytes);
int niqki_export_dump_slots(niqki_index *ix, uint32_t slot_begin, uint32_t slot_end,
                            uint8_t *buf, uint64_t capacity, uint64_t *size);
/* niqki_import_begin creates the handle from the header (slot range from `params`,
 * as above); niqki_import_slots takes the payload of whole slots [slot_begin,
 * slot_end) in order (*consumed = bytes used; slots outside the handle's range are
 * validated and dropped); after the last slot the handle is a normal index (built
 * on first use). */
int niqki_import_begin(const niqki_params *params, const uint8_t header[24], niqki_index **out);
int niqki_import_slots(niqki_index *ix, uint32_t slot_begin, uint32_t slot_end,
                       const uint8_t *buf, uint64_t len, uint64_t *consumed);

/* Reads back the stored sketches of genomes [begin, begin+n) as n x F int32
 * (slots outside the shard's range read -1). */
int niqki_get_sketches(niqki_index *ix, uint32_t begin, uint32_t n,
                       int32_t *sketches, int mem);

/* ---- one index over several GPUs: slot-range shards -----------------------------
 * No reference counterpart (the reference is one process on one host); this is how
 * Index::insert_sketch / Index::query_sketch (src/niqki_index.cpp:362-370, :633-687)
 * run when the F x 2^W inverted index is split by sketch-slot range over the GPUs
 * of a node.  Rank r of `world` owns slots niqki_group_slot_range(r) of EVERY
 * genome: a handle created with that slot range is one shard.  The hit count of a
 * genome is a sum over slots, so a batch needs one exchange of partial results,
 * done with RCCL over xGMI inside the library (librccl is loaded when the first
 * group is made).  Results are exactly those of one whole-range handle.
 * NIQKI_GROUP_TRANSPORT=ipc (environment, read by niqki_group_create and
 * niqki_group_new_id; one rank per process) replaces RCCL by direct peer access: every
 * rank maps its peers' exchange buffers through HIP IPC handles and pulls its part with
 * copy / summing kernels, ordered by sequence words in device memory -- a direct
 * all-to-all over all xGMI links, no host synchronisation inside a step, and ranks may
 * share a device (which RCCL refuses).
 *
 * A niqki_group is the set of ranks that live in the calling process:
 *   - one rank per process (n_local = 1, first_rank = this process' rank; `id` made
 *     once by niqki_group_new_id and carried to every process by the caller), or
 *   - all ranks in one process (n_local = world, first_rank = 0, id may be NULL).
 *     Shards of such a group may even share a device (tests; emulating G shards on
 *     one GPU): device-to-device copies then stand in for the collectives.
 * The per-rank arrays of the calls below have n_local entries, rank first_rank + i at
 * index i; sketch buffers are device memory of that rank's GPU, work is enqueued on
 * the shard handles' streams.  Calls are collective: every process of the group makes
 * the same calls in the same order. */
typedef struct niqki_group niqki_group; /* opaque */
#define NIQKI_GROUP_ID_BYTES 128

void niqki_group_slot_range(uint32_t rank, uint32_t world, uint32_t S, uint32_t *slot_begin,
                            uint32_t *slot_end);
int niqki_group_new_id(uint8_t id[NIQKI_GROUP_ID_BYTES]);
/* The shards must agree in K, S, W, min_score and genome count and own the slot ranges
 * of their ranks (S = 16: at least two shards).  On failure niqki_last_error(shards[0]) says why. */
int niqki_group_create(niqki_index *const *shards, uint32_t n_local, uint32_t first_rank,
                       uint32_t world, const uint8_t *id, niqki_group **out);
void niqki_group_destroy(niqki_group *g); /* the shard handles stay the caller's */
const char *niqki_group_last_error(const niqki_group *g);
/* "exchange": 0 = choose (sparse when min_score >= 4 * world), 1 = sparse (candidate
 * genomes only), 2 = dense (reduce-scatter of whole hit vectors); "cand_cap": candidate
 * ids per query and rank of the sparse form, even, default 256 (a step whose lists
 * overflow is redone densely, never answered wrongly); "surv_cap": survivors (genomes with a
 * partial count of at least half the candidate threshold) a shard keeps per query, default 1024,
 * at most 4096. */
int niqki_group_set_option(niqki_group *g, const char *key, int64_t value);
/* "overflows" (sparse steps redone densely so far), "rccl" (1 = RCCL transport),
 * "transport" (0 = device copies inside one process, 1 = RCCL, 2 = ipc),
 * "sparse" (1 = the sparse exchange is selected), "ipc_words_kind" (ipc transport: where this rank's
 * sequence words live -- 1 = fine-grained device memory mapped by the peers, 2 = the processes' shared block
 * page-locked and mapped into every device, 0 = plain device memory), "ipc_arena_fine" (1 = its exchange
 * buffers are fine-grained device memory), "ranks_seen" (how many ranks the transport itself knows of: the
 * communicator's size as RCCL reports it, the peers whose sequence words this process has mapped (ipc), the shards of
 * the process (local) -- equal to `world` when the group really spans what the caller thinks it spans). */
int niqki_group_get_stat(const niqki_group *g, const char *key, uint64_t *value);
/* Index::insert_sketch for a batch of world * per sketches of which rank r holds rows
 * [r*per, (r+1)*per) (local_sketches[i]: per x 2^S int32, device memory); the first
 * n_total rows of the batch (rank major) get the next genome ids, the rest is padding. */
int niqki_group_insert(niqki_group *g, const int32_t *const *local_sketches, uint32_t per,
                       uint32_t n_total);
/* Index::query_sketch for such a batch: rank r receives the hits of ITS rows against the
 * whole index -- hit_off[i] per+1 offsets, hit_counts[i] / hit_gids[i] capacity entries
 * each, in the memory space `mem` (as niqki_query). */
int niqki_group_query(niqki_group *g, const int32_t *const *local_sketches, uint32_t per,
                      uint64_t *const *hit_off, uint32_t *const *hit_counts,
                      uint32_t *const *hit_gids, uint64_t capacity, int mem);
/* The same in two halves.  _begin enqueues the whole batch on the shards' streams (slice
 * exchange, gather, sparse or dense sum, threshold + order into the caller's device buffers)
 * and returns without waiting for the device: the caller can enqueue other work -- the sketch
 * kernel of the next batch on another handle's stream -- that then runs beside the exchange.
 * _end is the one place the host waits: it checks the batch's candidate-overflow word (a
 * sparse batch whose lists overflowed is redone densely here), copies host results out
 * (NIQKI_MEM_HOST) and reports a peer that never answered (ipc transport).  One batch in
 * flight per group; the argument arrays of _begin may go away after it returns, the buffers
 * they point to must stay until _end.  niqki_group_query = _begin + _end. */
int niqki_group_query_begin(niqki_group *g, const int32_t *const *local_sketches, uint32_t per,
                            uint64_t *const *hit_off, uint32_t *const *hit_counts,
                            uint32_t *const *hit_gids, uint64_t capacity, int mem);
int niqki_group_query_end(niqki_group *g);

/* The two calls above on the shards' STAGED batches (niqki_stage_raw on each shard handle --
 * how the `niqki` host program feeds a multi-GPU index from files; all ranks in one process):
 * rank r contributes the first n_entry[r] sketches of its staged batch (0: none) as rows
 * [r*per, r*per + n_entry[r]) of the batch, the rest of its rows is padding.  An insert gives
 * the entries their ids rank after rank (rank 0's entries first), like the id counter of
 * src/niqki_index.cpp:396-401 does in file order. */
int niqki_group_staged_insert(niqki_group *g, uint32_t per, const uint32_t *n_entry);
int niqki_group_staged_query(niqki_group *g, uint32_t per, const uint32_t *n_entry, uint64_t *const *hit_off,
                             uint32_t *const *hit_counts, uint32_t *const *hit_gids, uint64_t capacity,
                             int mem);

/* Sizes of the handle's state: "store_bytes" (sketch store), "index_bytes" (table + id lists of
 * the built index or resident page), "tiles", "pages" / "page_slots" (pages a query walks and
 * slots per page; 1 / all slots unless the index is paged), "delta_genomes" (genomes indexed by
 * the delta segment, see option "incremental_build"), "last_gather_form" (launch form the last
 * counter call used: bit 0 look-up pre-pass, bit 1 its streamed-rows kernel, bit 2 locality
 * order), "last_hits_form" (1 = the last niqki_query* call took the hit-list form, option "hit_lists"), "class_mask" (1 = the built index carries its per-slot class mask: single-tile
 * indexes; the gather kernel then looks up only fingerprints whose sixteenth of the
 * fingerprint range holds a bucket in their slot -- a short read against a genome index
 * skips nearly all of its 2^S table look-ups; results are unaffected; NIQKI_HMASK=0 in the
 * environment builds indexes without it). */
int niqki_get_stat(const niqki_index *ix, const char *key, uint64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* NIQKI_HIP_H */
