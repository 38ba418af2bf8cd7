/*
 * niqki_hip_bench.h -- measurement, diagnosis and test support of libniqki_hip.so.
 *
 * NOT part of the drop-in boundary (include/niqki_hip.h): nothing here has a counterpart in the reference, a host
 * program that replaces the reference's hot path needs none of it.  bench.py, tools/ and tests/ do: deterministic
 * synthetic inputs, per-kernel-class device times, the roofline's T, integer-ALU / copy ceilings
 * measured live, the steps of the sparse multi-GPU exchange one by one, the staged framing read back, the pure
 * arithmetic of a group batch, the counters of the *_shared entry points.  Same library, same handle.
 */
#ifndef NIQKI_HIP_BENCH_H
#define NIQKI_HIP_BENCH_H

#include "niqki_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- diagnosis ------------------------------------------------------------------------------ */

/* The *_shared entry points' counters: batches run so far, requests served, the largest batch (any may be NULL).
 * Callable beside *_shared calls. */
int niqki_shared_stats(const niqki_index *ix, uint64_t *batches, uint64_t *requests,
                       uint64_t *largest_batch);

/* Read the staged framing back (HOST arrays, any may be NULL): rec_off n_rec+1,
 * seqs seq_bytes, entry_rec n_entry+1, hdr_pos n_rec.  For tests and diagnosis. */
int niqki_staged_records(niqki_index *ix, uint64_t *rec_off, uint8_t *seqs,
                         uint32_t *entry_rec, uint64_t *hdr_pos);

/* Sum over the shard's slots of the bucket length each query touches (the T
 * of the roofline formula).  Host out. Synchronises. */
int niqki_query_gathered(niqki_index *ix, const int32_t *sketches, uint32_t nq,
                         uint64_t *gathered_per_query, int mem);

/* The sparse exchange WITHOUT counter rows, step by step (NIQKI_MEM_DEVICE only; what a slot shard of
 * a niqki_group runs -- exposed for tests and for timing one shard's compute):
 *   niqki_query_survivors       gather over the handle's slots; per query the candidates (count >=
 *                               cand_threshold: ids, as niqki_candidates_from_counts) and the survivors
 *                               (count >= surv_threshold <= cand_threshold: surv[q*surv_cap + i] =
 *                               {genome id, count} as two int32, n_surv[q] how many qualified).  No
 *                               2N-byte counter row per query is written.
 *   niqki_survivor_counts       counts[q*m + i] = the handle's count of genome ids[q*m + i] (-1: 0) for
 *                               query q: from the survivor list, or -- an id that is not among them --
 *                               counted exactly as the slots where the genome's stored sketch equals the
 *                               query's (sketches: the same rows as given to niqki_query_survivors).
 *   niqki_hits_from_candidates  the hits of nq queries from m candidate ids each and their (cross-shard)
 *                               sums: distinct ids with a sum >= min_score, ordered as niqki_query's.
 * m <= 4096, surv_cap <= 4096. */
int niqki_query_survivors(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint32_t cand_threshold,
                          uint32_t surv_threshold, uint32_t cand_cap, uint32_t surv_cap, int32_t *cand,
                          int32_t *n_cand, int32_t *surv, int32_t *n_surv, int mem);
int niqki_survivor_counts(niqki_index *ix, const int32_t *sketches, uint32_t nq, const int32_t *ids, uint32_t m,
                          const int32_t *surv, const int32_t *n_surv, uint32_t surv_cap, uint16_t *counts,
                          int mem);
int niqki_hits_from_candidates(niqki_index *ix, const int32_t *ids, const uint16_t *totals, uint32_t nq,
                               uint32_t m, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
                               uint64_t capacity, int mem);

/* Multi-GPU helper (no reference counterpart; NIQKI_MEM_DEVICE only): for every
 * query the genomes whose counter in this shard's hit vector is >= threshold, at
 * most cap per query, unordered: cand[q*cap + i] (padded with -1), n_cand[q] =
 * how many qualified (may exceed cap).  A genome whose cross-shard sum reaches
 * min_score has a partial count >= ceil(min_score / shards) in some shard, so the
 * shards only need to exchange and sum these candidates. */
int niqki_candidates_from_counts(niqki_index *ix, const uint16_t *counts, uint32_t nq,
                                 uint64_t stride, uint32_t n_gids, uint32_t threshold,
                                 uint32_t cap, int32_t *cand, int32_t *n_cand, int mem);

/* niqki_query_counts and niqki_candidates_from_counts in one pass (NIQKI_MEM_DEVICE
 * only): the candidates are picked while a query's counters leave the gather kernel's
 * LDS, so the hit vectors are not read again (6.5 GB per 32 768 queries at 100 000
 * genomes).  Same outputs, the candidates of a query in another order.  Not on paged
 * or whole-range S = 16 handles.  What a slot shard of a niqki_group runs. */
int niqki_query_counts_candidates(niqki_index *ix, const int32_t *sketches, uint32_t nq,
                                  uint16_t *counts, uint64_t stride, uint32_t threshold,
                                  uint32_t cap, int32_t *cand, int32_t *n_cand, int mem);

/* The arithmetic of one query batch of a group, as niqki_group_query decides it -- a pure function (no device, no
 * handle: callable on a host without a GPU; the CPU test of the exchange protocol over gloo takes its slot
 * ranges, thresholds and buffer shapes from here and from niqki_group_slot_range, tests/test_dist_cpu.py).
 *   world, S, min_score     the group's shape (Index::query_sketch's threshold, src/niqki_index.cpp:646-650)
 *   exchange_option         option "exchange": 0 = choose, 1 = sparse, 2 = dense
 *   per, n_genomes          queries per rank in the batch, genomes indexed
 *   cand_cap                option "cand_cap" */
typedef struct niqki_group_plan {
  uint32_t sparse;          /* 1 = sparse candidate exchange, 0 = dense reduce-scatter of the counter rows */
  uint32_t cand_threshold;  /* ceil(min_score / world): a partial count from which a genome is a candidate */
  uint32_t surv_threshold;  /* max(1, cand_threshold / 2): ... from which a shard keeps it as a survivor */
  uint32_t slice_slots;     /* ceil(2^S / world): cells per query and peer in the slice exchange (int16 each) */
  uint64_t slice_bytes;     /* bytes of one peer's part of the slice exchange (per queries, 16-byte rounded) */
  uint64_t cand_blob_bytes; /* bytes of a rank's all-gathered candidate blob: world*per lists of cand_cap ids + 2 sizes each */
  uint64_t sum_words;       /* u32 words a rank receives from the reduce-scatter: candidates' packed u16 counts, or dense rows */
  uint64_t row_stride;      /* u16 cells per counter row (NIQKI_ROW_STRIDE(n_genomes)) */
} niqki_group_plan;
int niqki_group_plan_batch(uint32_t world, uint32_t S, uint32_t min_score, int exchange_option, uint32_t per,
                           uint32_t n_genomes, uint32_t cand_cap, niqki_group_plan *out);

/* ---- measurement support -------------------------------------------------- */

enum niqki_kernel_class {
  NIQKI_KC_SKETCH = 0,   /* rolling hash + per-slot min */
  NIQKI_KC_DENSIFY = 1,
  NIQKI_KC_GATHER = 2,   /* gather-histogram over the inverted index */
  NIQKI_KC_HITS = 3,     /* threshold + compaction + sort */
  NIQKI_KC_BUILD = 4,    /* insert transpose + CSR build */
  NIQKI_KC_INGEST = 5,   /* FASTA / FASTQ framing */
  NIQKI_KC_EXCHANGE = 6, /* slot-shard exchange: slice packing, candidate kernels, collectives */
  NIQKI_KC_INFLATE = 7,  /* gzip members inflated on the device */
  NIQKI_KC_COUNT = 8
};

/* When enabled, every launch of the classes above is bracketed by HIP events
 * on the handle's stream; niqki_profile_read synchronises and returns the
 * accumulated device time and launch count since the last reset. */
int niqki_profile_enable(niqki_index *ix, int on);
int niqki_profile_reset(niqki_index *ix);
int niqki_profile_read(niqki_index *ix, int kernel_class, double *ms,
                       uint64_t *launches);

/* Deterministic synthetic genomes (bench / parity inputs):
 * genome i is the ancestor of family[i] with per-base substitutions drawn for
 * (family[i], member[i]) at rate rate14[i]/16384.  Output: n records of len
 * bytes each, record i at out + i*stride.  Upper-case ACGT. The host and
 * device generators produce identical bytes. */
int niqki_synth_genomes(niqki_index *ix, uint64_t seed, const uint32_t *family,
                        const uint32_t *member, const uint32_t *rate14,
                        uint32_t n, uint64_t len, uint64_t stride, uint8_t *out,
                        int mem);
void niqki_synth_genome_host(uint64_t seed, uint32_t family, uint32_t member,
                             uint32_t rate14, uint64_t len, uint8_t *out);
/* Deterministic synthetic reads (BASELINE.json's short-sequence config): read i is bases
 * [offset[i], offset[i]+len) of the genome (family[i], member[i], rate14[i]) above, with the
 * read's own substitutions at rate read_rate14/16384 keyed by read_id[i] on top.  Record i at
 * out + i*stride.  Host and device produce identical bytes. */
int niqki_synth_reads(niqki_index *ix, uint64_t seed, const uint32_t *family, const uint32_t *member,
                      const uint32_t *rate14, const uint64_t *offset, const uint32_t *read_id,
                      uint32_t read_rate14, uint32_t n, uint32_t len, uint64_t stride, uint8_t *out,
                      int mem);
/* Integer-ALU ceilings of the device, measured live (SURVEY.md 8d asks for the sketch kernel as a
 * fraction of one): what = 0 independent 32-bit adds, 1 = 32-bit multiplies, 2 = the sketch
 * kernel's per-k-mer arithmetic alone (K = 31 roll + canonical choice + filter hash; no LDS,
 * memory or compaction), 3 = three-operand integer instructions (v_lshl_add_u32: the issue class of
 * every gfx950 vector opcode except add / sub / and / or / xor / mov / shift-right), 4 = a streaming
 * copy of 1 GiB (bytes read + written per second: the HBM rate a plain kernel reaches), 5 = the
 * densification passes of the short-read sketch kernel with nothing but their LDS traffic and exit
 * test (one wavefront per sketch, 9 per CU, two proposals + two read-backs per lane and pass: the
 * LDS round-trip ceiling of src/niqki_index.cpp:313-331 on this device).
 * *rate = adds / multiplies / k-mers / instructions / bytes / passes per second over ~ms milliseconds. */
int niqki_measure_alu(niqki_index *ix, int what, double ms, double *rate);

/* The device inflate alone (what niqki_stage_raw does with NIQKI_FILE_GZIP files; nq_inflate.hip), for tests and
 * rates: gzip file f = HOST bytes gz[gz_off[f], gz_off[f+1]); its bytes are expected to be exactly
 * raw_off[f+1] - raw_off[f] long and come back in HOST raw[raw_off[f] ..) (raw == NULL: they stay on the device).
 * status[f] = 0 or why the file was not taken (1..12, nq_kernels.h InflateJob), produced[f] = bytes written (never
 * more than expected), members[f] = members completed; the three may be NULL.  *outside (may be NULL) = bytes of the
 * output buffer outside the files' own ranges that differ from the pattern it was filled with (must be 0). */
int niqki_gunzip(niqki_index *ix, const uint8_t *gz, const uint64_t *gz_off, uint32_t n_files, const uint64_t *raw_off,
                 uint8_t *raw, uint32_t *status, uint64_t *produced, uint32_t *members, uint64_t *outside);
/* how the files of the last niqki_gunzip / niqki_stage_raw were decoded, summed over them: out[0] speculative
 * rounds, [1] bytes they produced, [2] tokens taken one by one, [3] DEFLATE blocks */
int niqki_gunzip_stats(niqki_index *ix, uint64_t out[4]);


#ifdef __cplusplus
}
#endif
#endif /* NIQKI_HIP_BENCH_H */
