// nq_handle.h -- the handle behind the C ABI (niqki_index) and the library-internal helpers
// shared by nq_api.hip (single-GPU entry points) and nq_group.hip (slot-sharded groups).
// Private to libniqki_hip.so.
#pragma once
#include "../../include/niqki_hip.h"
#include "../../include/niqki_hip_bench.h"
#include "nq_kernels.h"

#include <string>
#include <vector>

namespace nqi {

struct Buf {
  void *p = nullptr;
  size_t n = 0;
};

struct ProfSpan {
  int kc;
  hipEvent_t a, b;
};

void shared_free(struct ::niqki_index *ix);   // nq_shared.hip

}  // namespace nqi

struct niqki_index {
  niqki_params p{};
  nq::Derived d{};
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool stream_prio_set = false;   // option "stream_priority": the own stream (and the side stream) were made with stream_prio
  int stream_prio = 0;
  std::string err;

  // sketch store, u16 [f_local][cap]
  uint16_t *store = nullptr;
  uint64_t cap = 0;
  uint32_t n_genomes = 0;

  // Paged index (option "resident_bytes"): the sketch store lives in page-locked host memory and
  // the inverted index exists for one page of slots at a time; a query batch walks the pages and
  // the gather kernel accumulates the counters (hit counts are sums over slots).  While a page is
  // resident, d.slot_begin/slot_end, store and cap describe THAT page, so every kernel and launch
  // sequence of the resident case is reused unchanged.
  uint64_t resident_bytes = 0;          // 0 = everything resident (default)
  uint16_t *host_store = nullptr;       // u16 [full_end - full_begin][host_cap], page-locked
  uint64_t host_cap = 0;
  uint32_t full_begin = 0, full_end = 0;   // the handle's real slot range
  uint32_t page_begin = 0, page_end = 0;   // slots (relative to full_begin) of the resident page; equal = none
  uint32_t page_n = 0;                     // genomes the resident page was built for
  nqi::Buf pg_store, pg_stage;
  std::vector<uint64_t> pg_layout;      // dump word positions of all slots (export of a paged index), for pg_layout_n genomes
  uint32_t pg_layout_n = 0xFFFFFFFFu;

  // inverted index
  uint32_t tile = 0, n_tiles = 0, built_n = 0, align_log2 = 0, padded = 0;
  nq::Entry *entries = nullptr;
  uint16_t *gids = nullptr;
  uint64_t *tile_base = nullptr;   // n_tiles+1, device
  uint32_t *slot_units = nullptr;  // n_tiles x (f_local+1), device
  size_t entries_bytes = 0, gids_bytes = 0, tile_base_bytes = 0, slot_units_bytes = 0;
  uint32_t *ptab = nullptr;        // packed copy of `entries` for the look-up pre-pass (made at its first launch)
  size_t ptab_bytes = 0;
  bool ptab_ok = false;
  uint16_t *hmask = nullptr;       // per slot: which sixteenths of the fingerprint range hold a bucket (IndexView::hmask)
  size_t hmask_bytes = 0;
  bool hmask_ok = false;
  uint32_t stripe = 0;             // tiles are dealt round-robin
  int stripe_opt = 32;             // option: block size of the stripes when there are several tiles (0 = ranges)
  int bucket_align = -1;           // option: log2 ids per bucket alignment unit, -1 = choose
  bool built = false;
  // Delta segment: genomes inserted after the last full build get an index of their own (same
  // layout, over store columns [g_base, g_base + seg_n)) instead of a rebuild of everything; a query
  // walks both, the counter columns are disjoint.  The flat members above describe the CURRENT
  // segment -- the main one except while swap_segment() has put the delta there (its build, its
  // gather launch); `alt` keeps the other one's state.
  struct Seg {
    nq::Entry *entries = nullptr;
    uint16_t *gids = nullptr;
    uint64_t *tile_base = nullptr;
    uint32_t *slot_units = nullptr;
    uint32_t *ptab = nullptr;   // packed copy of `entries` for the look-up pre-pass (made at its first launch)
    uint16_t *hmask = nullptr;
    size_t entries_bytes = 0, gids_bytes = 0, tile_base_bytes = 0, slot_units_bytes = 0, ptab_bytes = 0, hmask_bytes = 0;
    uint32_t tile = 0, n_tiles = 0, seg_n = 0, g_base = 0, align_log2 = 0, padded = 0, stripe = 0;
    bool ptab_ok = false, hmask_ok = false;
  } alt;
  uint32_t seg_n = 0;      // genomes of the current segment
  uint32_t g_base = 0;     // its first genome
  uint32_t delta_n = 0;    // genomes the delta segment covers (0 = there is none)
  int incremental = 1;     // option "incremental_build"

  void *shared_state = nullptr;   // nq_shared.hip: the combiner of the *_shared (many host threads) entry points

  int gather_variant = 0;
  uint32_t last_form = 0;        // stat "last_gather_form": 1 = look-up pre-pass, 2 = its streamed-rows form, 4 = locality order
  uint64_t record_len_hint = 0;  // avg bytes per sketch for device-side batches (0 = read it back)
  uint32_t query_batch = 1024;
  int query_order = 1;           // option: order the queries of a launch for cache locality (1 = where it pays, 2 = wherever possible)
  int lookup_prepass = -1;       // option: slot-major table look-up pre-pass: -1 = when it pays, 0 = never, 1 = whenever usable
  int hit_lists = 1;             // option: queries of a single small tile leave the gather kernel as ordered hit lists (no counter rows)
  uint32_t hit_list_cap = 256;   // option: hits per query such a list holds; a query with more goes through its counter row
  uint32_t last_hits_form = 0;   // stat "last_hits_form": 1 = the last query call took the hit-list form

  nqi::Buf ws_seq, ws_recoff, ws_entry, ws_sk, ws_counts, ws_blk, ws_hitoff, ws_hc, ws_hg, ws_tc, ws_tg,
      ws_misc, ws_stash, ws_hl;
  // staged batch (niqki_stage_raw): framing results live in ws_seq / ws_recoff / ws_entry
  nqi::Buf ws_raw, ws_fmeta, ws_summ, ws_chunk, ws_fkept, ws_fnrec, ws_hdrpos, ws_ehdr, ws_stsk, ws_order, ws_pre, ws_useg, ws_ijob, ws_xtab;
  uint64_t inflate_stats[4] = {0, 0, 0, 0};   // niqki_gunzip_stats
  int inflate_window = -1;   // option "inflate_window": which form of the inflate kernel a launch takes
  bool xtab_ok = false;   // ws_xtab holds the CRC folding constants of the inflate kernel
  struct {
    bool valid = false, sketched = false;
    uint32_t n_entry = 0, n_rec = 0;
    uint64_t seq_bytes = 0;
    uint64_t entry_bytes = 0;   // sequence bytes of the records the entries cover (lines mode may stop before the last framed record)
    const uint32_t *entry_rec = nullptr;  // device, n_entry+1
  } staged;
  // niqki_stage_raw_prefetch: file bytes of coming batches on their way into ws_wire[slot] on copy_stream.  Two
  // slots, taken in turn: the bytes of batch i + 1 may cross while batch i -- whose own (prefetched) bytes are still
  // being inflated / unpacked out of the other slot -- is staged.
  nqi::Buf ws_wire[2];
  nqi::Buf ws_redo[2];   // sketch_dev: flags of short-record sketches left to the second launch ([1]: calls on the sketch lane)
  hipStream_t copy_stream = nullptr;
  struct {
    bool valid = false;
    std::vector<const uint8_t *> ptr;
    std::vector<uint64_t> off;
    hipEvent_t ev = nullptr;
  } pre[2];
  uint32_t pre_next = 0;   // the slot the next prefetch takes

  // niqki_sketch_ahead / niqki_query_ahead: the sketches of up to two coming query batches, made on the sketch lane (a
  // side stream of the handle) beside whatever the handle's stream runs -- the gather and hit kernels of the batch before
  hipStream_t sk_stream = nullptr;
  uint32_t sk_lane_cus = 0;      // option "sketch_lane_cus": the sketch lane's compute units (0 = all)
  struct Ahead {
    nqi::Buf sk;                 // n_entry x F cells
    uint32_t n_entry = 0;
    hipEvent_t done = nullptr;   // its sketch kernel has finished (recorded on sk_stream)
    hipEvent_t used = nullptr;   // the query that took it has read it (recorded on the handle's stream)
    bool used_set = false;
  } ahead[2];
  uint32_t ahead_head = 0, ahead_n = 0;   // the oldest slot in flight, how many are

  bool prof = false;
  double prof_ms[NIQKI_KC_COUNT] = {0};
  uint64_t prof_n[NIQKI_KC_COUNT] = {0};
  std::vector<nqi::ProfSpan> spans;
  std::vector<hipEvent_t> ev_pool;
  // side stream: the locality probe + order of a launch run beside its look-up pre-pass
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};


namespace nqi {

// ---- nq_api.hip: the handle ----
void *host_alloc(size_t bytes);   // page-locked host memory (niqki_host_alloc): slabs on transparent huge pages
void host_free(void *p);
int fail(niqki_index *ix, int code, const std::string &msg);
int ensure(niqki_index *ix, Buf &b, size_t bytes);   // device scratch of at least `bytes`
nq::IndexView view(const niqki_index *ix);
std::string &create_error();   // thread-local: why the last niqki_create / niqki_import_* of the calling thread failed
int derive(const niqki_params &p, nq::Derived &d, std::string &why);
int collect_spans(niqki_index *ix);
// more than 2^15 slots on the handle (whole-range S = 16): counts reach 2^16, two counter planes (nq_kernels.h, kPassSlots)
bool two_planes(const niqki_index *ix);
// first slot of the handle in a whole sketch row (while a page is resident d.slot_begin is the page's)
uint32_t first_slot(const niqki_index *ix);
// ---- nq_api_build.hip: sketch store, sketching, index segments ----
int reserve_store(niqki_index *ix, uint64_t want);
int sketch_dev(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec, const uint32_t *entry_rec,
               uint32_t n_entry, int32_t *sketches, uint64_t total_bytes);
void swap_segment(niqki_index *ix);   // flat index members <-> alt ("Delta segment" above)
int build_range(niqki_index *ix, uint32_t g_base, uint32_t N);
int build_if_needed(niqki_index *ix);
int build_single(niqki_index *ix);    // ONE index over all genomes (dump export, per-bucket statistics)
// ---- nq_api_query.hip: counters and hits ----
uint32_t page_slots(const niqki_index *ix);
int load_page(niqki_index *ix, uint32_t s0, uint32_t s1);
int counts_resident(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq, uint16_t *counts,
                    uint64_t stride, bool accumulate, uint16_t *counts2 = nullptr, const nq::CandOut *co = nullptr);
int query_hits_dev(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint16_t *c1, uint16_t *c2, uint64_t stride,
                   unsigned long long *hit_off, uint32_t *hc, uint32_t *hg, uint64_t capacity, bool check_capacity,
                   uint64_t *total_out);
int query_to_host(niqki_index *ix, const int32_t *sketches, bool sk_dev, uint32_t nq, uint64_t *hit_off, uint32_t *hit_counts,
                  uint32_t *hit_gids, uint64_t capacity);
// hit counters of nq device-resident sketches (rows q_stride apart, this shard's slots at q_off)
// counts2: the second counter plane of a whole-range S = 16 handle (nq_kernels.h, kPassSlots), else nullptr
// co: also the candidate lists of the rows (nq_kernels.h CandOut; presets them itself), not on paged or S = 16 handles
int counts_dev(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq,
               uint16_t *counts, uint64_t stride, uint16_t *counts2 = nullptr, const nq::CandOut *co = nullptr);
int hits_dev(niqki_index *ix, const uint16_t *counts, uint32_t nq, uint64_t stride, uint32_t gid_begin,
             uint32_t n_gids, unsigned long long *hit_off, uint32_t *hc, uint32_t *hg, uint64_t capacity,
             bool check_capacity, uint64_t *total_out, const uint16_t *counts2 = nullptr);
// sketches of the handle's staged batch (niqki_stage_raw) into ix->ws_stsk, once per batch
int staged_sketch_ws(niqki_index *ix);
// appends n device-resident sketches (same addressing as counts_dev) to the sketch store
int insert_dev(niqki_index *ix, const int32_t *sketches, uint32_t sk_stride, uint32_t sk_off, uint32_t n);

// HIP events around a kernel class while profiling is on (niqki_profile_*)
struct Span {
  niqki_index *ix;
  int kc;
  hipEvent_t a = nullptr, b = nullptr;
  Span(niqki_index *ix_, int kc_);
  ~Span();
};

}  // namespace nqi

#define NQ_HIP(ix, call)                                                                    \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return nqi::fail(ix, e_ == hipErrorOutOfMemory ? NIQKI_E_NOMEM : NIQKI_E_HIP,         \
                       std::string(#call) + ": " + hipGetErrorString(e_));                  \
  } while (0)
