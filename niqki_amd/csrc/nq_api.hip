// nq_api.hip -- the C ABI of libniqki_hip.so (include/niqki_hip.h): handle,
// device memory, staging of host buffers, and the launch sequences.  No
// compute happens on the host here and there is no CPU fallback: without a
// gfx950 device niqki_create fails.
#include "nq_handle.h"
#include "nq_pack.h"
#include "nq_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {
thread_local std::string g_create_err;  // why the last niqki_create / niqki_import_dump on this thread failed
}  // namespace

namespace nqi {

int fail(niqki_index *ix, int code, const std::string &msg) {
  if (ix) ix->err = msg;
  return code;
}

int ensure(niqki_index *ix, Buf &b, size_t bytes) {
  if (bytes <= b.n && b.p) return NIQKI_OK;
  if (b.p) NQ_HIP(ix, hipFree(b.p));
  b.p = nullptr;
  b.n = 0;
  size_t want = std::max<size_t>(bytes + bytes / 4, 256);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    want = std::max<size_t>(bytes, 256);
    NQ_HIP(ix, hipMalloc(&b.p, want));
  }
  b.n = want;
  return NIQKI_OK;
}

hipEvent_t get_event(niqki_index *ix) {
  if (!ix->ev_pool.empty()) {
    hipEvent_t e = ix->ev_pool.back();
    ix->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();   // a span that cannot be timed is dropped, the error must not stick
    return nullptr;
  }
  return e;
}

// The events belong to the handle's device: a caller that walks several shards (nq_group.hip) may
// have another one current.
Span::Span(niqki_index *ix_, int kc_) : ix(ix_), kc(kc_) {
  if (!ix->prof) return;
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || (cur != ix->device && hipSetDevice(ix->device) != hipSuccess)) {
    (void)hipGetLastError();
    return;
  }
  a = get_event(ix);
  b = get_event(ix);
  if (!a || !b || hipEventRecord(a, ix->stream) != hipSuccess) {
    (void)hipGetLastError();
    if (a) ix->ev_pool.push_back(a);
    if (b) ix->ev_pool.push_back(b);
    a = b = nullptr;
  }
  if (cur != ix->device) (void)hipSetDevice(cur);
}
Span::~Span() {
  if (!ix->prof || !a || !b) return;
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (cur != ix->device) (void)hipSetDevice(ix->device);
  if (hipEventRecord(b, ix->stream) == hipSuccess) {
    ix->spans.push_back({kc, a, b});
  } else {
    (void)hipGetLastError();
    ix->ev_pool.push_back(a);
    ix->ev_pool.push_back(b);
  }
  if (cur != ix->device && cur >= 0) (void)hipSetDevice(cur);
}

int collect_spans(niqki_index *ix) {
  if (ix->spans.empty()) return NIQKI_OK;
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  for (auto &s : ix->spans) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      ix->prof_ms[s.kc] += ms;
      ix->prof_n[s.kc] += 1;
    }
    ix->ev_pool.push_back(s.a);
    ix->ev_pool.push_back(s.b);
  }
  ix->spans.clear();
  return NIQKI_OK;
}

int derive(const niqki_params &p, nq::Derived &d, std::string &why) {
  if (p.K < 1 || p.K > 31) { why = "K must be in 1..31"; return NIQKI_E_INVALID; }
  if (p.S < 1 || p.S > 16) { why = "S must be in 1..16"; return NIQKI_E_INVALID; }
  if (p.W < 1 || p.W > 15 || p.H > p.W) { why = "need H <= W <= 15"; return NIQKI_E_INVALID; }
  if (p.S + p.W > 30) { why = "S+W must be <= 30"; return NIQKI_E_INVALID; }
  d.K = p.K; d.S = p.S; d.W = p.W; d.H = p.H; d.M = p.W - p.H;
  d.F = 1u << p.S;
  d.R = 1u << p.W;
  d.mask_m = (1u << d.M) - 1u;
  d.max_rem = (1u << p.H) - 1u;
  d.min_score = p.min_score;
  d.slot_begin = p.slot_begin;
  d.slot_end = p.slot_end;
  if (d.slot_begin == 0 && d.slot_end == 0) d.slot_end = d.F;
  if (d.slot_begin >= d.slot_end || d.slot_end > d.F) { why = "bad slot range"; return NIQKI_E_INVALID; }
  d.kmer_mask = (1ULL << (2 * p.K)) - 1ULL;
  return NIQKI_OK;
}

nq::IndexView view(const niqki_index *ix) {
  nq::IndexView v;
  v.d = ix->d;
  v.n_genomes = ix->seg_n;
  v.g_base = ix->g_base;
  v.tile = ix->tile;
  v.n_tiles = ix->n_tiles;
  v.f_local = ix->d.slot_end - ix->d.slot_begin;
  v.align_log2 = ix->align_log2;
  v.padded = ix->padded;
  v.stripe = ix->stripe;
  v.cap = ix->cap;
  v.store = ix->store;
  v.q_stride = ix->d.F;
  v.q_off = ix->d.slot_begin;
  v.accumulate = 0;
  v.entries = ix->entries;
  v.gids = ix->gids;
  v.tile_base = ix->tile_base;
  v.slot_units = ix->slot_units;
  v.ptab = ix->ptab_ok ? ix->ptab : nullptr;
  v.hmask = ix->hmask_ok ? ix->hmask : nullptr;
  v.hmask_shift = ix->d.W > 4 ? ix->d.W - 4 : 0;
  return v;
}

// paged index: the store is page-locked host memory
int reserve_host_store(niqki_index *ix, uint64_t want) {
  if (want <= ix->host_cap) return NIQKI_OK;
  uint64_t cap = std::max<uint64_t>(want, ix->host_cap * 2);
  cap = (cap + 63) / 64 * 64;
  const uint32_t f_all = ix->full_end - ix->full_begin;
  uint16_t *ns = nullptr;
  if (hipHostMalloc((void **)&ns, (size_t)f_all * cap * 2, hipHostMallocDefault) != hipSuccess)
    return fail(ix, NIQKI_E_NOMEM, "page-locked sketch store allocation failed");
  if (ix->host_store && ix->n_genomes) {
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    for (uint32_t s = 0; s < f_all; ++s)
      std::memcpy(ns + (size_t)s * cap, ix->host_store + (size_t)s * ix->host_cap, (size_t)ix->n_genomes * 2);
  }
  if (ix->host_store) (void)hipHostFree(ix->host_store);
  ix->host_store = ns;
  ix->host_cap = cap;
  return NIQKI_OK;
}

int reserve_store(niqki_index *ix, uint64_t want) {
  if (ix->resident_bytes) return reserve_host_store(ix, want);
  if (want <= ix->cap) return NIQKI_OK;
  uint64_t cap = std::max<uint64_t>(want, ix->cap * 2);
  cap = (cap + 63) / 64 * 64;
  const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
  uint16_t *ns = nullptr;
  hipError_t e = hipMalloc((void **)&ns, (size_t)f_local * cap * 2);
  if (e != hipSuccess && cap > (want + 63) / 64 * 64) {
    cap = (want + 63) / 64 * 64;
    e = hipMalloc((void **)&ns, (size_t)f_local * cap * 2);
  }
  if (e != hipSuccess) return fail(ix, NIQKI_E_NOMEM, "sketch store allocation failed");
  if (ix->store && ix->n_genomes) {
    NQ_HIP(ix, hipMemcpy2DAsync(ns, cap * 2, ix->store, ix->cap * 2, (size_t)ix->n_genomes * 2, f_local,
                                hipMemcpyDeviceToDevice, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  }
  if (ix->store) NQ_HIP(ix, hipFree(ix->store));
  ix->store = ns;
  ix->cap = cap;
  return NIQKI_OK;
}

// Launches the sketch kernel(s) on device-resident inputs.
int sketch_dev(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec,
               const uint32_t *entry_rec, uint32_t n_entry, int32_t *sketches, uint64_t total_bytes) {
  if (n_entry == 0) return NIQKI_OK;
  nq::SketchArgs a;
  a.d = ix->d;
  a.seqs = seqs;
  a.rec_off = rec_off;
  a.entry_rec = entry_rec;
  a.sketches = sketches;
  a.accumulate = 0;
  a.densify = 1;
  a.splits = 1;
  const uint64_t avg = total_bytes / n_entry;
  if (avg >= 16384 && !entry_rec && n_entry < 128 && avg >= (1u << 20))
    a.splits = std::min<uint32_t>(32, 512 / n_entry);
  if (a.splits > 1 || nq::sketch_needs_merge(ix->d)) {  // partial sketches merged in global memory, then densified
    {
      Span sp(ix, NIQKI_KC_SKETCH);
      NQ_HIP(ix, nq::launch_fill_u32((uint32_t *)sketches, (uint64_t)n_entry * ix->d.F, nq::kEmpty32,
                                     ix->stream));
      a.densify = 0;
      NQ_HIP(ix, nq::launch_sketch(a, n_entry, avg / a.splits, ix->stream));
    }
    Span sp(ix, NIQKI_KC_DENSIFY);
    nq::SketchArgs b = a;
    b.seqs = nullptr;
    b.splits = 1;
    b.accumulate = 1;
    b.densify = 1;
    NQ_HIP(ix, nq::launch_sketch(b, n_entry, (uint64_t)1 << 22, ix->stream));
  } else {
    Span sp(ix, NIQKI_KC_SKETCH);
    NQ_HIP(ix, nq::launch_sketch(a, n_entry, avg, ix->stream));
  }
  (void)n_rec;
  return NIQKI_OK;
}

// a whole-range S = 16 handle counts in two planes of <= 2^15 slots each (nq_kernels.h, kPassSlots)
// more than 2^15 slots on the handle (whole-range S = 16): counts reach 2^16, the slots are walked in two halves
// into two counter planes (a paged handle: its pages never straddle the halves)
bool two_planes(const niqki_index *ix) {
  return (ix->resident_bytes ? ix->full_end - ix->full_begin : ix->d.slot_end - ix->d.slot_begin) > nq::kPassSlots;
}

// first slot of the handle in a whole sketch row (while a page is resident d.slot_begin is the page's)
uint32_t first_slot(const niqki_index *ix) { return ix->resident_bytes ? ix->full_begin : ix->d.slot_begin; }

// flat index members <-> alt (nq_handle.h, "Delta segment")
void swap_segment(niqki_index *ix) {
  auto &a = ix->alt;
  std::swap(ix->entries, a.entries); std::swap(ix->gids, a.gids);
  std::swap(ix->tile_base, a.tile_base); std::swap(ix->slot_units, a.slot_units);
  std::swap(ix->entries_bytes, a.entries_bytes); std::swap(ix->gids_bytes, a.gids_bytes);
  std::swap(ix->tile_base_bytes, a.tile_base_bytes); std::swap(ix->slot_units_bytes, a.slot_units_bytes);
  std::swap(ix->tile, a.tile); std::swap(ix->n_tiles, a.n_tiles); std::swap(ix->seg_n, a.seg_n);
  std::swap(ix->g_base, a.g_base); std::swap(ix->align_log2, a.align_log2);
  std::swap(ix->padded, a.padded); std::swap(ix->stripe, a.stripe);
  std::swap(ix->ptab, a.ptab); std::swap(ix->ptab_bytes, a.ptab_bytes); std::swap(ix->ptab_ok, a.ptab_ok);
  std::swap(ix->hmask, a.hmask); std::swap(ix->hmask_bytes, a.hmask_bytes); std::swap(ix->hmask_ok, a.hmask_ok);
}

int build_range(niqki_index *ix, uint32_t g_base, uint32_t N);

int build_if_needed(niqki_index *ix) {
  if (ix->resident_bytes) {   // paged: pages are built while a query walks them
    ix->built_n = ix->n_genomes;
    return NIQKI_OK;
  }
  if (ix->built && ix->built_n == ix->n_genomes) return NIQKI_OK;
  // Genomes inserted after a build: a delta segment for them while they are few (the fixed part of a
  // build -- one table row per slot -- is ~10 ms at the north-star shape, a full rebuild of 100 000
  // genomes 65 ms), a full rebuild once the delta would pass an eighth of the main index.
  const uint32_t main_n = ix->seg_n;
  if (ix->incremental && main_n >= 4096 && ix->n_genomes > main_n &&
      ix->n_genomes - main_n <= std::min<uint32_t>(main_n / 8, nq::kPadMaxTile)) {
    swap_segment(ix);
    int rc = build_range(ix, main_n, ix->n_genomes - main_n);
    swap_segment(ix);
    if (rc) { ix->built = false; return rc; }
    ix->delta_n = ix->n_genomes - main_n;
    ix->built_n = ix->n_genomes;
    ix->built = true;
    return NIQKI_OK;
  }
  return niqki_build(ix);
}

// before anything that needs ONE index over all genomes (dump export, per-bucket statistics)
int build_single(niqki_index *ix) {
  if (ix->resident_bytes || (ix->built && ix->built_n == ix->n_genomes && ix->delta_n == 0)) return build_if_needed(ix);
  return niqki_build(ix);
}

// ---- paged index ---------------------------------------------------------------------------
// slots per page so that a page's sketch-store rows and its inverted index stay within the budget
uint32_t page_slots(const niqki_index *ix) {
  const uint64_t N = std::max<uint64_t>(ix->n_genomes, 1), R = ix->d.R;
  const uint64_t nt = (N + nq::kPadMaxTile - 1) / nq::kPadMaxTile;
  const uint64_t per_slot = (N + 63) / 64 * 64 * 2      // store row
                            + R * nt * sizeof(nq::Entry)  // table row
                            + (N + R * nt * 32) * 2;      // id lists with their alignment padding (estimate)
  uint64_t f = ix->resident_bytes / per_slot / 32 * 32;
  const uint32_t f_all = ix->full_end - ix->full_begin;
  if (f < 32) f = 32;
  return (uint32_t)std::min<uint64_t>({f, (uint64_t)f_all, (uint64_t)nq::kPassSlots});
}

// Makes slots [s0, s1) (relative to the handle's first slot) the resident page: store rows from host
// memory, then the normal index build on them.
int load_page(niqki_index *ix, uint32_t s0, uint32_t s1) {
  if (ix->page_begin == s0 && ix->page_end == s1 && ix->page_n == ix->n_genomes && ix->built) return NIQKI_OK;
  const uint32_t N = ix->n_genomes;
  const uint64_t cap = ((uint64_t)N + 63) / 64 * 64;
  int rc = ensure(ix, ix->pg_store, std::max<size_t>((size_t)(s1 - s0) * cap * 2, 4));
  if (rc) return rc;
  if (N)
    NQ_HIP(ix, hipMemcpy2DAsync(ix->pg_store.p, cap * 2, ix->host_store + (size_t)s0 * ix->host_cap, ix->host_cap * 2,
                                (size_t)N * 2, s1 - s0, hipMemcpyHostToDevice, ix->stream));
  ix->d.slot_begin = ix->full_begin + s0;
  ix->d.slot_end = ix->full_begin + s1;
  ix->store = (uint16_t *)ix->pg_store.p;
  ix->cap = cap;
  ix->page_begin = s0;
  ix->page_end = s1;
  ix->page_n = N;
  ix->built = false;
  return niqki_build(ix);
}

int counts_resident(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq,
                    uint16_t *counts, uint64_t stride, bool accumulate, uint16_t *counts2 = nullptr,
                    const nq::CandOut *co = nullptr);

// counts over a paged index: page after page, the gather kernel adding to the rows from the second on
int counts_paged(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq, uint16_t *counts,
                 uint64_t stride, uint16_t *counts2) {
  const uint32_t f_all = ix->full_end - ix->full_begin, f_page = page_slots(ix);
  if (stride < ix->n_genomes || (stride & 1)) return fail(ix, NIQKI_E_INVALID, "stride must be even and >= genome count");
  if (f_all > nq::kPassSlots && !counts2) return fail(ix, NIQKI_E_INVALID, "S = 16: counts reach 2^16, use niqki_query_counts32 (or the hit calls)");
  ix->built_n = ix->n_genomes;
  if (nq == 0 || ix->n_genomes == 0) return NIQKI_OK;
  // the pages of slots [0, 2^15) add up in `counts`, those of the slots behind (S = 16 only) in `counts2`:
  // either sum stays <= 2^15
  for (uint32_t h0 = 0; h0 < f_all; h0 += nq::kPassSlots) {
    const uint32_t h1 = std::min(f_all, h0 + nq::kPassSlots);
    for (uint32_t s0 = h0; s0 < h1; s0 += f_page) {
      const uint32_t s1 = std::min(h1, s0 + f_page);
      int rc = load_page(ix, s0, s1);
      if (rc) return rc;
      // q_off addresses the handle's first slot in a sketch row; the page starts s0 slots further
      if ((rc = counts_resident(ix, sketches, q_stride, q_off + s0, nq, h0 ? counts2 : counts, stride, s0 != h0))) return rc;
    }
  }
  return NIQKI_OK;
}

// counts for nq device-resident sketches into a device buffer
int counts_dev(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq,
               uint16_t *counts, uint64_t stride, uint16_t *counts2, const nq::CandOut *co) {
  if (co && co->hl) {   // hit lists (query_hits_dev has checked the index shape)
    if (ix->resident_bytes || two_planes(ix) || co->cand || !counts) return fail(ix, NIQKI_E_INVALID, "hit lists: resident single-plane handles, with counter rows to fall back on");
    return counts_resident(ix, sketches, q_stride, q_off, nq, counts, stride, false, nullptr, co);
  }
  if (co && (ix->resident_bytes || two_planes(ix) || !co->cand || !co->n || !co->cap))
    return fail(ix, NIQKI_E_INVALID, "candidate lists: not on a paged or whole-range S = 16 handle; cand, n_cand and cap > 0 needed");
  if (ix->resident_bytes) return counts_paged(ix, sketches, q_stride, q_off, nq, counts, stride, counts2);
  if (two_planes(ix) && !counts2) return fail(ix, NIQKI_E_INVALID, "S = 16: counts reach 2^16, use niqki_query_counts32 (or the hit calls)");
  if (!counts && !(co && co->surv)) return fail(ix, NIQKI_E_INVALID, "no counter rows: only together with survivor lists");
  if (co && co->surv && (!co->surv_n || !co->surv_cap || co->surv_thr > co->thr))
    return fail(ix, NIQKI_E_INVALID, "survivor lists need surv_n, surv_cap > 0 and surv_thr <= thr");
  if (co && nq) {
    NQ_HIP(ix, hipMemsetAsync(co->n, 0, (size_t)nq * 4, ix->stream));
    NQ_HIP(ix, hipMemsetAsync(co->cand, 0xFF, (size_t)nq * co->cap * 4, ix->stream));
    if (co->surv) NQ_HIP(ix, hipMemsetAsync(co->surv_n, 0, (size_t)nq * 4, ix->stream));
  }
  return counts_resident(ix, sketches, q_stride, q_off, nq, counts, stride, false, counts2, co);
}

int counts_resident(niqki_index *ix, const int32_t *sketches, uint32_t q_stride, uint32_t q_off, uint32_t nq,
                    uint16_t *counts, uint64_t stride, bool accumulate, uint16_t *counts2, const nq::CandOut *co) {
  int rc = ix->resident_bytes ? NIQKI_OK : build_if_needed(ix);
  if (rc) return rc;
  if (nq == 0) return NIQKI_OK;
  if (counts && (stride < ix->built_n || (stride & 1))) return fail(ix, NIQKI_E_INVALID, "stride must be even and >= genome count");
  if ((uintptr_t)counts & 3) return fail(ix, NIQKI_E_INVALID, "counts must be 4-byte aligned (rows are written as packed u16 pairs)");
  if (ix->built_n == 0) return NIQKI_OK;
  if (ix->delta_n && !ix->resident_bytes) {  // the delta segment first (its columns are its own), then the main index below
    const uint32_t dn = ix->delta_n;
    ix->delta_n = 0;
    swap_segment(ix);
    rc = counts_resident(ix, sketches, q_stride, q_off, nq, counts, stride, accumulate, counts2, co);
    swap_segment(ix);
    ix->delta_n = dn;
    if (rc) return rc;
  }
  // launches of at most `chunk` queries bound the per-query stash (one Entry per
  // slot and extra tile) whatever the caller's batch size is
  const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
  nq::IndexView v = view(ix);
  v.q_stride = q_stride;
  v.q_off = q_off;
  v.accumulate = accumulate ? 1u : 0u;
  // Table look-ups: inside the gather kernel (one random table line per query and slot), or
  // by the slot-major pre-pass, which walks the table once per launch for all its queries.
  bool pre = ix->lookup_prepass != 0 && nq::launch_lookup_usable(v) && (((uintptr_t)(sketches + q_off)) & 15) == 0 && (q_stride & 3) == 0;
  // Measured at the north-star shape (profiles/r02_*): with random 16-byte look-ups (lookup_kernel) the
  // pre-pass takes 16 % of the HBM traffic off a launch but not its time -- both forms are bound by the
  // number of random line requests a CU keeps in flight, and inside the gather kernel the look-ups
  // overlap with the bucket walk.  The default (-1) therefore takes the pre-pass only where it wins:
  // An index of more than 4 tiles (> 261 632 genomes) is different: inside the kernel only 4 tiles'
  // entries can be parked per look-up, so every further tile would cost its own random table line
  // per query and slot; there the pre-pass is the default for real batches.
  // Up to 2 tiles of W <= 12 the pre-pass has a form that streams whole table rows through LDS from a
  // packed copy of the table (lookup_rows_kernel): 0.75 ms per 4096 queries, the launch 8 % faster than
  // with the look-ups inside the gather kernel -- the default for batches of >= 1024 queries.
  // (not on a paged index: the packed copy would be made again for every page, outside its memory budget)
  if (ix->lookup_prepass < 0)
    pre = pre && ((ix->n_tiles > 4 && nq >= 256) || (nq::lookup_wants_packed(v) && nq >= 1024 && !ix->resident_bytes && ix->n_tiles <= 2));
  // locality order of each launch: worth its probe on large indexes and real batches.  It takes ~8 % off
  // the gather kernel and costs 0.1 ms per 4096 queries at 100 000 genomes whatever the slot count:
  // measured even on a slot shard of 4096 slots (1.56 against 1.57 ms per 4096 queries), +4 % at 8192.
  const bool ordered = ix->query_order && nq >= 64 && ix->seg_n < (1u << 20) - 1 &&
                       (ix->query_order >= 2 || (ix->seg_n >= 16384 && f_local >= 8192));
  // launches of at most `chunk` queries: the order kernel sorts <= 4096, and the per-query scratch
  // (stash or pre-pass words) stays <= 128 MiB whatever the caller's batch size is (bigger launches are
  // no faster: 32 768 query shards in one launch take 8 x the time of 4096)
  const size_t per_query = pre ? nq::lookup_pre_bytes(v, 1) : (size_t)(ix->n_tiles - 1) * f_local * sizeof(nq::Entry);
  uint32_t chunk = nq;
  if (per_query) chunk = (uint32_t)std::max<size_t>(4096, ((size_t)128 << 20) / per_query);
  if (ordered || pre) chunk = 4096;
  // (the pre-pass words of a launch: at most 8 GiB -- a launch of 4096 queries on up to 16 tiles at S = 15; the streamed
  // form reads the table once per launch, so fewer, larger launches halve its traffic on a 500 000-genome index)
  if (pre && per_query * chunk > ((size_t)8 << 30)) chunk = std::max<uint32_t>(256, (uint32_t)((((size_t)8 << 30) / per_query) & ~(size_t)255));
  if (pre) {
    if (nq::lookup_wants_packed(v) && !ix->ptab_ok) {   // packed copy of the table, once per build
      const size_t want = (size_t)f_local * ix->d.R * ix->n_tiles * 4;
      if (want > ix->ptab_bytes) {
        if (ix->ptab) NQ_HIP(ix, hipFree(ix->ptab));
        ix->ptab = nullptr; ix->ptab_bytes = 0;
        NQ_HIP(ix, hipMalloc((void **)&ix->ptab, want));
        ix->ptab_bytes = want;
      }
      NQ_HIP(ix, nq::launch_pack_entries(v, ix->ptab, ix->stream));
      ix->ptab_ok = true;
      v.ptab = ix->ptab;
    }
    if ((rc = ensure(ix, ix->ws_pre, nq::lookup_pre_bytes(v, std::min(nq, chunk))))) return rc;
  } else if (ix->n_tiles > 1) {
    rc = ensure(ix, ix->ws_stash, (size_t)std::min(nq, chunk) * (ix->n_tiles - 1) * f_local * sizeof(nq::Entry));
    if (rc) return rc;
  }
  if (ordered && (rc = ensure(ix, ix->ws_order, (size_t)chunk * 8))) return rc;
  ix->last_form = (pre ? 1u : 0u) | (pre && nq::lookup_wants_packed(v) ? 2u : 0u) | (ordered ? 4u : 0u);
  for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t n = std::min(chunk, nq - q0);
    Span sp(ix, NIQKI_KC_GATHER);
    const uint32_t *order = nullptr;
    const bool fork = ordered && n >= 64 && pre;   // probe + order beside the pre-pass (both only read the sketches)
    if (fork) {
      if (!ix->aux_stream) {
        if (ix->stream_prio_set) NQ_HIP(ix, hipStreamCreateWithPriority(&ix->aux_stream, hipStreamNonBlocking, ix->stream_prio));
        else NQ_HIP(ix, hipStreamCreateWithFlags(&ix->aux_stream, hipStreamNonBlocking));
        NQ_HIP(ix, hipEventCreateWithFlags(&ix->ev_fork, hipEventDisableTiming));
        NQ_HIP(ix, hipEventCreateWithFlags(&ix->ev_join, hipEventDisableTiming));
      }
      NQ_HIP(ix, hipEventRecord(ix->ev_fork, ix->stream));
      NQ_HIP(ix, hipStreamWaitEvent(ix->aux_stream, ix->ev_fork, 0));
    }
    if (ordered && n >= 64) {
      uint32_t *keys = (uint32_t *)ix->ws_order.p;
      NQ_HIP(ix, nq::launch_order(v, sketches + (size_t)q0 * q_stride, n, keys, keys + chunk, fork ? ix->aux_stream : ix->stream));
      order = keys + chunk;
      if (fork) NQ_HIP(ix, hipEventRecord(ix->ev_join, ix->aux_stream));
    }
    if (pre)
      NQ_HIP(ix, nq::launch_lookup(v, sketches + (size_t)q0 * q_stride, n, (uint32_t *)ix->ws_pre.p, ix->stream));
    if (fork) NQ_HIP(ix, hipStreamWaitEvent(ix->stream, ix->ev_join, 0));
    nq::CandOut c;
    if (co) {
      c = *co;
      if (co->cand) { c.cand += (size_t)q0 * co->cap; c.n += q0; }
      if (co->surv) { c.surv += (size_t)q0 * co->surv_cap; c.surv_n += q0; }
      if (co->hl) { c.hl += (size_t)q0 * co->hl_cap; c.hl_n += q0; }
    }
    NQ_HIP(ix, nq::launch_gather(v, sketches + (size_t)q0 * q_stride, n, counts ? counts + (size_t)q0 * stride : nullptr,
                                 counts2 ? counts2 + (size_t)q0 * stride : nullptr, stride,
                                 pre ? (nq::Entry *)ix->ws_pre.p : (nq::Entry *)ix->ws_stash.p, order, ix->gather_variant,
                                 pre, ix->stream, c));
  }
  return NIQKI_OK;
}

// hits from device-resident counters into device buffers; hit_off device (nq+1)
int hits_dev(niqki_index *ix, const uint16_t *counts, uint32_t nq, uint64_t stride, uint32_t gid_begin,
             uint32_t n_gids, unsigned long long *hit_off, uint32_t *hc, uint32_t *hg, uint64_t capacity,
             bool check_capacity, uint64_t *total_out, const uint16_t *counts2) {
  nq::HitsArgs a;
  a.counts = counts;
  a.counts2 = counts2;
  a.stride = stride;
  a.nq = nq;
  a.gid_begin = gid_begin;
  a.n_gids = n_gids;
  a.min_score = ix->d.min_score;
  a.n_blk = (n_gids + nq::kHitsBlk - 1) / nq::kHitsBlk;
  a.hit_off = hit_off;
  a.hit_counts = hc;
  a.hit_gids = hg;
  a.capacity = capacity;
  if (nq == 0 || a.n_blk == 0) {
    NQ_HIP(ix, hipMemsetAsync(hit_off, 0, (size_t)(nq + 1) * 8, ix->stream));
    if (total_out) *total_out = 0;
    return NIQKI_OK;
  }
  int rc = ensure(ix, ix->ws_blk, (size_t)nq * a.n_blk * 4);
  if (rc) return rc;
  if ((rc = ensure(ix, ix->ws_tc, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;
  if ((rc = ensure(ix, ix->ws_tg, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;
  a.blk_counts = (uint32_t *)ix->ws_blk.p;
  a.tmp_counts = (uint32_t *)ix->ws_tc.p;
  a.tmp_gids = (uint32_t *)ix->ws_tg.p;
  Span sp(ix, NIQKI_KC_HITS);
  NQ_HIP(ix, nq::launch_hits_count(a, ix->stream));
  if (check_capacity) {
    unsigned long long total = 0;
    NQ_HIP(ix, hipMemcpyAsync(&total, hit_off + nq, 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    if (total_out) *total_out = total;
    if (total > capacity) return NIQKI_E_CAPACITY;
  }
  NQ_HIP(ix, nq::launch_hits_emit(a, ix->stream));
  return NIQKI_OK;
}

// Index::query_sketch (src/niqki_index.cpp:633-687) for nq device-resident whole sketches into device buffers: counters,
// threshold, order.  c1 / c2: counter planes of nq rows (c2 only on a two-plane handle).  On a single-tile, single-
// segment index with a small tile -- the short-read shape -- the hits leave the gather kernel as ordered lists and no
// counter row is written or read again, except for a query with more than hit_list_cap hits.
int query_hits_dev(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint16_t *c1, uint16_t *c2, uint64_t stride,
                   unsigned long long *hit_off, uint32_t *hc, uint32_t *hg, uint64_t capacity, bool check_capacity,
                   uint64_t *total_out) {
  int rc = build_if_needed(ix);
  if (rc) return rc;
  const uint32_t N = ix->built_n;
  const bool lists = ix->hit_lists && nq && N && !ix->resident_bytes && !two_planes(ix) && ix->n_tiles == 1 && ix->delta_n == 0 &&
                     ix->tile <= nq::kHitListMaxTile && ix->g_base == 0 && N <= 65536u;
  ix->last_hits_form = lists ? 1u : 0u;
  if (!lists) {
    if ((rc = counts_dev(ix, sketches, ix->d.F, first_slot(ix), nq, c1, stride, c2))) return rc;
    return hits_dev(ix, c1, nq, stride, 0, N, hit_off, hc, hg, capacity, check_capacity, total_out, c2);
  }
  const uint32_t cap = std::min<uint32_t>((std::max<uint32_t>(ix->hit_list_cap, 1) + 3u) & ~3u, nq::kHitListMaxCap);
  if ((rc = ensure(ix, ix->ws_hl, (size_t)nq * cap * 4))) return rc;
  if ((rc = ensure(ix, ix->ws_blk, ((size_t)nq * 2 + 1) * 4))) return rc;   // the lists' sizes, then the overflowing queries
  nq::CandOut co;
  co.hl = (uint32_t *)ix->ws_hl.p;
  co.hl_n = (uint32_t *)ix->ws_blk.p;
  co.hl_over = (uint32_t *)ix->ws_blk.p + nq;
  if ((rc = ensure(ix, ix->ws_tc, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;   // (lists of > 2048 hits)
  if ((rc = ensure(ix, ix->ws_tg, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;
  co.hl_cap = cap;
  co.hl_min = ix->d.min_score;
  if ((rc = counts_dev(ix, sketches, ix->d.F, first_slot(ix), nq, c1, stride, nullptr, &co))) return rc;
  nq::HitsArgs a{};
  a.counts = c1;
  a.counts2 = nullptr;
  a.stride = stride;
  a.nq = nq;
  a.gid_begin = 0;
  a.n_gids = N;
  a.min_score = ix->d.min_score;
  a.hit_off = hit_off;
  a.hit_counts = hc;
  a.hit_gids = hg;
  a.tmp_counts = (uint32_t *)ix->ws_tc.p;
  a.tmp_gids = (uint32_t *)ix->ws_tg.p;
  a.capacity = capacity;
  uint32_t *over = (uint32_t *)ix->ws_blk.p + nq;
  Span sp(ix, NIQKI_KC_HITS);
  NQ_HIP(ix, nq::launch_hitlist_scan((const uint32_t *)ix->ws_blk.p, a, cap, over, ix->stream));
  if (check_capacity) {
    unsigned long long total = 0;
    NQ_HIP(ix, hipMemcpyAsync(&total, hit_off + nq, 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    if (total_out) *total_out = total;
    if (total > capacity) return NIQKI_E_CAPACITY;
  }
  NQ_HIP(ix, nq::launch_hitlist_emit(a, (const uint32_t *)ix->ws_hl.p, cap, over, ix->stream));
  return NIQKI_OK;
}

// Hits of nq sketches (host memory, or device-resident when sk_dev) into HOST arrays:
// batches of query_batch sketches, hits appended in query order.
int query_to_host(niqki_index *ix, const int32_t *sketches, bool sk_dev, uint32_t nq, uint64_t *hit_off,
                  uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity) {
  int rc;
  const uint32_t N = ix->built_n;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  uint64_t base = 0;
  bool overflow = false;
  hit_off[0] = 0;
  const uint32_t qb = ix->query_batch;
  const size_t planes = two_planes(ix) ? 2 : 1;
  std::vector<unsigned long long> off(qb + 1);
  for (uint32_t q0 = 0; q0 < nq; q0 += qb) {
    const uint32_t n = std::min(qb, nq - q0);
    if (!sk_dev && (rc = ensure(ix, ix->ws_sk, (size_t)n * ix->d.F * 4))) return rc;
    const size_t plane = std::max<size_t>((size_t)n * stride * 2, 2);
    if ((rc = ensure(ix, ix->ws_counts, plane * planes))) return rc;
    uint16_t *c1 = (uint16_t *)ix->ws_counts.p, *c2 = planes == 2 ? (uint16_t *)((char *)ix->ws_counts.p + plane) : nullptr;
    if ((rc = ensure(ix, ix->ws_hitoff, (size_t)(n + 1) * 8))) return rc;
    const uint64_t room = overflow || base > capacity ? 0 : capacity - base;
    if ((rc = ensure(ix, ix->ws_hc, (size_t)std::max<uint64_t>(room, 1) * 4))) return rc;
    if ((rc = ensure(ix, ix->ws_hg, (size_t)std::max<uint64_t>(room, 1) * 4))) return rc;
    const int32_t *d_sk = sketches + (size_t)q0 * ix->d.F;
    if (!sk_dev) {
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, d_sk, (size_t)n * ix->d.F * 4, hipMemcpyHostToDevice, ix->stream));
      d_sk = (const int32_t *)ix->ws_sk.p;
    }
    uint64_t total = 0;
    rc = query_hits_dev(ix, d_sk, n, c1, c2, stride, (unsigned long long *)ix->ws_hitoff.p, (uint32_t *)ix->ws_hc.p,
                        (uint32_t *)ix->ws_hg.p, room, true, &total);
    if (rc && rc != NIQKI_E_CAPACITY) return rc;
    NQ_HIP(ix, hipMemcpyAsync(off.data(), ix->ws_hitoff.p, (size_t)(n + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
    if (rc == NIQKI_OK && total) {
      NQ_HIP(ix, hipMemcpyAsync(hit_counts + base, ix->ws_hc.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipMemcpyAsync(hit_gids + base, ix->ws_hg.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
    }
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    if (rc == NIQKI_E_CAPACITY) overflow = true;
    for (uint32_t i = 0; i < n; ++i) hit_off[q0 + i + 1] = base + off[i + 1];
    base += off[n];
  }
  return overflow ? NIQKI_E_CAPACITY : NIQKI_OK;
}

int insert_dev(niqki_index *ix, const int32_t *sketches, uint32_t sk_stride, uint32_t sk_off, uint32_t n) {
  if (n == 0) return NIQKI_OK;
  if ((uint64_t)ix->n_genomes + n > 0xFFFFFFFFull) return fail(ix, NIQKI_E_INVALID, "too many genomes");
  int rc = reserve_store(ix, (uint64_t)ix->n_genomes + n);
  if (rc) return rc;
  if (ix->resident_bytes) {
    // paged: transpose into a device staging block of all the handle's slots, then rows to the host store
    const uint32_t f_all = ix->full_end - ix->full_begin;
    const uint64_t n_pad = ((uint64_t)n + 63) / 64 * 64;
    if ((rc = ensure(ix, ix->pg_stage, (size_t)f_all * n_pad * 2))) return rc;
    nq::Derived d = ix->d;
    d.slot_begin = ix->full_begin;
    d.slot_end = ix->full_end;
    {
      Span sp(ix, NIQKI_KC_BUILD);
      NQ_HIP(ix, nq::launch_store_insert(d, sketches, sk_stride, sk_off, n, (uint16_t *)ix->pg_stage.p, n_pad, 0, ix->stream));
    }
    NQ_HIP(ix, hipMemcpy2DAsync(ix->host_store + ix->n_genomes, ix->host_cap * 2, ix->pg_stage.p, n_pad * 2, (size_t)n * 2, f_all,
                                hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    ix->n_genomes += n;
    ix->built = false;
    return NIQKI_OK;
  }
  {
    Span sp(ix, NIQKI_KC_BUILD);
    NQ_HIP(ix, nq::launch_store_insert(ix->d, sketches, sk_stride, sk_off, n, ix->store, ix->cap, ix->n_genomes, ix->stream));
  }
  ix->n_genomes += n;
  ix->built = false;
  return NIQKI_OK;
}

}  // namespace nqi

using namespace nqi;

extern "C" {

int niqki_abi_version(void) { return NIQKI_ABI_VERSION; }

const char *niqki_status_string(int s) {
  switch (s) {
    case NIQKI_OK: return "ok";
    case NIQKI_E_INVALID: return "invalid argument";
    case NIQKI_E_NOMEM: return "out of memory";
    case NIQKI_E_HIP: return "HIP runtime error";
    case NIQKI_E_CAPACITY: return "output capacity too small";
    case NIQKI_E_STATE: return "invalid state";
    case NIQKI_E_NODEVICE: return "no gfx950 device";
    default: return "unknown status";
  }
}

uint32_t niqki_min_score(double min_fract, uint32_t S) {
  double f = (double)(1u << S);
  return (uint32_t)(min_fract * f);
}

namespace {
// Width of the fingerprint interval a candidate H spans between the 2 % and 98 %
// quantiles of a slot's minimum hash when x k-mers fall into the slot
// (score_H, src/niqki_index.cpp:142-164).  q -> position of hash quantile u = q * 2^64
// on the fingerprint axis: below the HyperLogLog range the fingerprint is the hash
// scaled down, inside it a saturating exponent plus the mantissa share.
double fingerprint_axis(double u, double h_range, double m_bits) {
  if (u < std::pow(2, 64 - h_range + 1)) return u * std::pow(2, h_range - 64 - m_bits - 1);
  const double i = std::log2(u) + h_range - 64;
  const double j = u * std::pow(2, m_bits - 64 - i + h_range);
  return i * std::pow(2, m_bits) + j;
}
double interval_width(double x, int try_h, uint32_t W) {
  const double eps = 0.02;
  // W - try_h is unsigned in the reference (wraps when try_h > W)
  const double h_range = std::pow(2, try_h), m_bits = (double)(uint32_t)(W - (uint32_t)try_h);
  const double lo = ((double)1 - std::pow(1 - eps, 1 / x)) * std::pow(2, 64);
  const double hi = ((double)1 - std::pow(eps, 1 / x)) * std::pow(2, 64);
  return fingerprint_axis(hi, h_range, m_bits) - fingerprint_axis(lo, h_range, m_bits);
}
}  // namespace

int niqki_select_best_H(niqki_index *ix, double genome_size, uint32_t *H_out) {
  if (!ix) return NIQKI_E_INVALID;
  const double x = genome_size / (double)ix->d.F;
  uint32_t H = ix->d.H;
  double best = 0;
  for (int try_h = 2; try_h < 7; ++try_h) {  // src/niqki_index.cpp:129-135
    const double w = interval_width(x, try_h, ix->d.W);
    if (w > best) { best = w; H = (uint32_t)try_h; }
  }
  if (H > ix->d.W) return fail(ix, NIQKI_E_INVALID, "select_best_H chose H > W");
  // :136 -- H and M only; mask_m / max_rem stay as the constructor left them
  ix->d.H = H;
  ix->d.M = ix->d.W - H;
  ix->p.H = H;
  if (H_out) *H_out = H;
  return NIQKI_OK;
}

int niqki_create(const niqki_params *params, niqki_index **out) {
  if (!params || !out) return NIQKI_E_INVALID;
  *out = nullptr;
  niqki_index *ix = new (std::nothrow) niqki_index();
  if (!ix) return NIQKI_E_NOMEM;
  ix->p = *params;
  auto bail = [&](int code, const std::string &why) { g_create_err = why; delete ix; return code; };
  std::string why;
  int rc = derive(*params, ix->d, why);
  if (rc) return bail(rc, why);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return bail(NIQKI_E_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e) + ", devices " + std::to_string(ndev));
  int dev = params->device;
  if (dev < 0 && (e = hipGetDevice(&dev)) != hipSuccess) return bail(NIQKI_E_NODEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
  if (dev >= ndev) return bail(NIQKI_E_NODEVICE, "device ordinal " + std::to_string(dev) + " >= device count " + std::to_string(ndev));
  if ((e = hipSetDevice(dev)) != hipSuccess) return bail(NIQKI_E_NODEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return bail(NIQKI_E_NODEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return bail(NIQKI_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is gfx950 only");
  ix->device = dev;
  ix->resident_bytes = (uint64_t)params->resident_mib << 20;
  ix->full_begin = ix->d.slot_begin;
  ix->full_end = ix->d.slot_end;
  if ((e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking)) != hipSuccess) return bail(NIQKI_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  ix->own_stream = true;
  if (const char *v = std::getenv("NIQKI_GATHER_VARIANT"))
    if (nq::gather_variant_valid(std::atoi(v))) ix->gather_variant = std::atoi(v);
  if (const char *v = std::getenv("NIQKI_QUERY_ORDER")) ix->query_order = std::min(std::max(std::atoi(v), 0), 2);
  if (const char *v = std::getenv("NIQKI_LOOKUP_PREPASS")) ix->lookup_prepass = std::atoi(v);
  *out = ix;
  return NIQKI_OK;
}

void niqki_destroy(niqki_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  (void)hipStreamSynchronize(ix->stream);
  for (Buf *b : {&ix->ws_seq, &ix->ws_recoff, &ix->ws_entry, &ix->ws_sk, &ix->ws_counts, &ix->ws_blk,
                 &ix->ws_hitoff, &ix->ws_hc, &ix->ws_hg, &ix->ws_tc, &ix->ws_tg, &ix->ws_misc, &ix->ws_stash,
                 &ix->ws_raw, &ix->ws_raw2, &ix->ws_fmeta, &ix->ws_summ, &ix->ws_chunk, &ix->ws_fkept, &ix->ws_fnrec,
                 &ix->ws_hdrpos, &ix->ws_ehdr, &ix->ws_stsk, &ix->ws_order, &ix->ws_pre, &ix->ws_hl, &ix->ws_useg})
    if (b->p) (void)hipFree(b->p);
  for (Buf *b : {&ix->pg_store, &ix->pg_stage})
    if (b->p) (void)hipFree(b->p);
  if (ix->store && !ix->resident_bytes) (void)hipFree(ix->store);
  if (ix->host_store) (void)hipHostFree(ix->host_store);
  for (void *p : {(void *)ix->alt.entries, (void *)ix->alt.gids, (void *)ix->alt.tile_base, (void *)ix->alt.slot_units,
                  (void *)ix->alt.ptab, (void *)ix->ptab, (void *)ix->alt.hmask, (void *)ix->hmask})
    if (p) (void)hipFree(p);
  if (ix->entries) (void)hipFree(ix->entries);
  if (ix->gids) (void)hipFree(ix->gids);
  if (ix->tile_base) (void)hipFree(ix->tile_base);
  if (ix->slot_units) (void)hipFree(ix->slot_units);
  for (auto &s : ix->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto e : ix->ev_pool) (void)hipEventDestroy(e);
  if (ix->aux_stream) { (void)hipStreamSynchronize(ix->aux_stream); (void)hipStreamDestroy(ix->aux_stream); }
  if (ix->copy_stream) { (void)hipStreamSynchronize(ix->copy_stream); (void)hipStreamDestroy(ix->copy_stream); }
  if (ix->ev_copy) (void)hipEventDestroy(ix->ev_copy);
  if (ix->ev_fork) (void)hipEventDestroy(ix->ev_fork);
  if (ix->ev_join) (void)hipEventDestroy(ix->ev_join);
  if (ix->own_stream) (void)hipStreamDestroy(ix->stream);
  nqi::shared_free(ix);
  delete ix;
}

const char *niqki_last_error(const niqki_index *ix) { return ix ? ix->err.c_str() : g_create_err.c_str(); }

int niqki_get_params(const niqki_index *ix, niqki_params *out) {
  if (!ix || !out) return NIQKI_E_INVALID;
  *out = ix->p;
  out->slot_begin = ix->resident_bytes ? ix->full_begin : ix->d.slot_begin;
  out->slot_end = ix->resident_bytes ? ix->full_end : ix->d.slot_end;
  out->device = ix->device;
  out->tile_genomes = ix->tile ? ix->tile : ix->p.tile_genomes;
  return NIQKI_OK;
}

int niqki_set_stream(niqki_index *ix, void *s) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  if (ix->own_stream) { (void)hipStreamDestroy(ix->stream); ix->own_stream = false; }
  ix->stream = (hipStream_t)s;
  return NIQKI_OK;
}

void *niqki_get_stream(const niqki_index *ix) { return ix ? (void *)ix->stream : nullptr; }

int niqki_synchronize(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_set_option(niqki_index *ix, const char *key, int64_t value) {
  if (!ix || !key) return NIQKI_E_INVALID;
  if (!std::strcmp(key, "gather_variant")) {
    // launch shapes 0 (choose) .. 5; the measurement-only variants exist in ABLATION builds alone
    if (!nq::gather_variant_valid((int)value)) return fail(ix, NIQKI_E_INVALID, "unknown gather_variant");
    ix->gather_variant = (int)value;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "tile_stripe")) { ix->stripe_opt = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 64); ix->built = false; ix->seg_n = 0; return NIQKI_OK; }
  if (!std::strcmp(key, "query_order")) { ix->query_order = (int)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return NIQKI_OK; }
  if (!std::strcmp(key, "incremental_build")) { ix->incremental = value != 0; return NIQKI_OK; }
  if (!std::strcmp(key, "lookup_prepass")) {
    if (value < -1 || value > 1) return fail(ix, NIQKI_E_INVALID, "lookup_prepass: -1 = when it pays, 0 = never, 1 = whenever usable");
    ix->lookup_prepass = (int)value;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "hit_lists")) { ix->hit_lists = value != 0; return NIQKI_OK; }
  if (!std::strcmp(key, "hit_list_cap")) {
    if (value < 1 || value > (int64_t)nq::kHitListMaxCap) return fail(ix, NIQKI_E_INVALID, "hit_list_cap must be in 1..2048");
    ix->hit_list_cap = (uint32_t)value;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "query_batch")) { if (value < 1) return NIQKI_E_INVALID; ix->query_batch = (uint32_t)value; return NIQKI_OK; }
  if (!std::strcmp(key, "tile_genomes")) {
    if (value < 0 || value > 65536 || (value & 63)) return fail(ix, NIQKI_E_INVALID, "tile_genomes must be a multiple of 64, <= 65536");
    ix->p.tile_genomes = (uint32_t)value;
    ix->built = false;
    ix->seg_n = 0;   // the next build is a full one
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "bucket_align_log2")) {
    if (value < -1 || value > 6) return fail(ix, NIQKI_E_INVALID, "bucket_align_log2 must be -1 (choose) .. 6");
    ix->bucket_align = (int)value;
    ix->built = false;
    ix->seg_n = 0;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "resident_bytes")) {
    if (value < 0) return fail(ix, NIQKI_E_INVALID, "resident_bytes must be >= 0");
    if (ix->n_genomes || ix->store) return fail(ix, NIQKI_E_STATE, "resident_bytes must be set before the first insert");
    ix->resident_bytes = (uint64_t)value;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "record_len_hint")) { ix->record_len_hint = (uint64_t)value; return NIQKI_OK; }
  if (!std::strcmp(key, "stream_priority")) {
    // a stream of the handle's own (and its side stream) made anew, at the top (1), default (0) or bottom (-1) of the
    // device's priority range: the hardware scheduler hands free CUs to the queue of higher priority first
    if (value < -1 || value > 1) return fail(ix, NIQKI_E_INVALID, "stream_priority: 1 = high, 0 = default, -1 = low");
    NQ_HIP(ix, hipSetDevice(ix->device));
    int least = 0, greatest = 0;   // (numerically: greatest priority = the smallest value)
    NQ_HIP(ix, hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = value > 0 ? greatest : (value < 0 ? least : 0);
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    hipStream_t s = nullptr;
    NQ_HIP(ix, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio));
    if (ix->own_stream) (void)hipStreamDestroy(ix->stream);   // (a caller's stream, niqki_set_stream, is the caller's to keep)
    ix->stream = s;
    ix->own_stream = true;
    if (ix->aux_stream) {
      NQ_HIP(ix, hipStreamSynchronize(ix->aux_stream));
      (void)hipStreamDestroy(ix->aux_stream);
      ix->aux_stream = nullptr;
    }
    ix->stream_prio = prio;
    ix->stream_prio_set = true;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "min_score")) { ix->p.min_score = ix->d.min_score = (uint32_t)value; return NIQKI_OK; }
  return fail(ix, NIQKI_E_INVALID, std::string("unknown option ") + key);
}

int niqki_reserve(niqki_index *ix, uint32_t n_genomes) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  return reserve_store(ix, n_genomes);
}

int niqki_sketch(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec,
                 const uint32_t *entry_rec, uint32_t n_entry, int32_t *sketches, int mem) {
  if (!ix || (!seqs && n_rec) || !rec_off || (!sketches && n_entry)) return NIQKI_E_INVALID;
  if (!entry_rec && n_entry != n_rec) return fail(ix, NIQKI_E_INVALID, "n_entry must equal n_rec without entry_rec");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (n_entry == 0) return NIQKI_OK;
  const size_t sk_bytes = (size_t)n_entry * ix->d.F * 4;
  if (mem == NIQKI_MEM_DEVICE) {
    // total size is only needed to pick the launch shape: read the last offset
    uint64_t total = ix->record_len_hint * n_entry;
    if (total == 0) {
      NQ_HIP(ix, hipMemcpyAsync(&total, rec_off + n_rec, 8, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    }
    return sketch_dev(ix, seqs, rec_off, n_rec, entry_rec, n_entry, sketches, total);
  }
  const uint64_t total = rec_off[n_rec];
  int rc;
  ix->staged.valid = false;  // the staging buffers are shared with niqki_stage_raw
  if ((rc = ensure(ix, ix->ws_seq, (size_t)total + NIQKI_SEQ_PAD))) return rc;
  if ((rc = ensure(ix, ix->ws_recoff, (size_t)(n_rec + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_sk, sk_bytes))) return rc;
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_seq.p, seqs, total, hipMemcpyHostToDevice, ix->stream));
  NQ_HIP(ix, hipMemsetAsync((uint8_t *)ix->ws_seq.p + total, 0, NIQKI_SEQ_PAD, ix->stream));
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_recoff.p, rec_off, (size_t)(n_rec + 1) * 8, hipMemcpyHostToDevice, ix->stream));
  const uint32_t *d_entry = nullptr;
  if (entry_rec) {
    if ((rc = ensure(ix, ix->ws_entry, (size_t)(n_entry + 1) * 4))) return rc;
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_entry.p, entry_rec, (size_t)(n_entry + 1) * 4, hipMemcpyHostToDevice, ix->stream));
    d_entry = (const uint32_t *)ix->ws_entry.p;
  }
  rc = sketch_dev(ix, (const uint8_t *)ix->ws_seq.p, (const uint64_t *)ix->ws_recoff.p, n_rec, d_entry,
                  n_entry, (int32_t *)ix->ws_sk.p, total);
  if (rc) return rc;
  NQ_HIP(ix, hipMemcpyAsync(sketches, ix->ws_sk.p, sk_bytes, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_densify(niqki_index *ix, int32_t *sketches, uint32_t n, int mem) {
  if (!ix || (!sketches && n)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (n == 0) return NIQKI_OK;
  int32_t *d_sk = sketches;
  const size_t bytes = (size_t)n * ix->d.F * 4;
  if (mem == NIQKI_MEM_HOST) {
    int rc = ensure(ix, ix->ws_sk, bytes);
    if (rc) return rc;
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, sketches, bytes, hipMemcpyHostToDevice, ix->stream));
    d_sk = (int32_t *)ix->ws_sk.p;
  }
  nq::SketchArgs a;
  a.d = ix->d;
  a.seqs = nullptr;
  a.rec_off = nullptr;
  a.entry_rec = nullptr;
  a.sketches = d_sk;
  a.splits = 1;
  a.accumulate = 1;
  a.densify = 1;
  {
    Span sp(ix, NIQKI_KC_DENSIFY);
    NQ_HIP(ix, nq::launch_sketch(a, n, ix->d.F <= 4096 ? 150 : ((uint64_t)1 << 22), ix->stream));
  }
  if (mem == NIQKI_MEM_HOST) {
    NQ_HIP(ix, hipMemcpyAsync(sketches, d_sk, bytes, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  }
  return NIQKI_OK;
}

int niqki_insert(niqki_index *ix, const int32_t *sketches, uint32_t n, int mem) {
  if (!ix || (!sketches && n)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (n == 0) return NIQKI_OK;
  int rc;
  const int32_t *d_sk = sketches;
  if (mem == NIQKI_MEM_HOST) {
    const size_t bytes = (size_t)n * ix->d.F * 4;
    if ((rc = ensure(ix, ix->ws_sk, bytes))) return rc;
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, sketches, bytes, hipMemcpyHostToDevice, ix->stream));
    d_sk = (const int32_t *)ix->ws_sk.p;
  }
  if ((rc = insert_dev(ix, d_sk, ix->d.F, first_slot(ix), n))) return rc;
  if (mem == NIQKI_MEM_HOST) NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

uint32_t niqki_genome_count(const niqki_index *ix) { return ix ? ix->n_genomes : 0; }

int niqki_build(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (ix->resident_bytes && ix->page_begin == ix->page_end) {  // paged, no page chosen: queries build their pages
    ix->built_n = ix->n_genomes;
    return NIQKI_OK;
  }
  ix->delta_n = 0;   // one index over everything inserted so far
  int rc = build_range(ix, 0, ix->n_genomes);
  if (rc == NIQKI_OK) {
    ix->built_n = ix->n_genomes;
    // the delta segment's buffers are not needed until genomes arrive again: give their memory back
    auto &a = ix->alt;
    for (void *p : {(void *)a.entries, (void *)a.gids, (void *)a.tile_base, (void *)a.slot_units, (void *)a.ptab, (void *)a.hmask})
      if (p) (void)hipFree(p);
    a = niqki_index::Seg();
  }
  return rc;
}

}  // extern "C"

namespace nqi {
// the index of store columns [g_base, g_base + N) into the current segment's buffers
int build_range(niqki_index *ix, uint32_t g_base, uint32_t N) {
  const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
  uint32_t tile = ix->p.tile_genomes;
  if (const char *v = std::getenv("NIQKI_TILE_GENOMES")) tile = (uint32_t)std::atoi(v);
  // genomes are dealt to the tiles round-robin in blocks (option "tile_stripe": 0 = ranges, B = block size)
  int stripe = ix->stripe_opt;
  if (const char *v = std::getenv("NIQKI_TILE_STRIPE")) stripe = std::atoi(v);
  uint32_t B = 1;
  while (stripe > 0 && B * 2 <= (uint32_t)stripe && B < 64) B *= 2;
  if (tile == 0 || tile > 65536 || (tile & 63)) {
    // as few tiles as the 16-bit tile-local ids (padding ids included) and the LDS counter array allow
    uint32_t nt = std::max<uint32_t>(1, (N + nq::kPadMaxTile - 1) / nq::kPadMaxTile);
    tile = ((N + nt - 1) / nt + 63) / 64 * 64;
    // room for the fullest tile of a block-striped index, if that does not cost a tile
    const uint32_t want = (((N + B - 1) / B + nt - 1) / nt * B + 63) / 64 * 64;
    if (stripe > 0 && nt > 1 && want <= nq::kPadMaxTile && (N + want - 1) / want == nt) tile = std::max(tile, want);
    if (tile == 0) tile = 64;
  }
  const uint32_t n_tiles = (N + tile - 1) / tile;
  // 128-byte aligned buckets pay off once buckets are long (big tiles); for small tiles the
  // padding would dominate the id array.
  int al = ix->bucket_align;
  if (const char *v = std::getenv("NIQKI_BUCKET_ALIGN_LOG2")) al = std::atoi(v);
  if (al < 0 || al > 6) al = tile >= 16384 ? 6 : (tile >= 2048 ? 3 : 0);
  auto grow = [&](void **p, size_t &have, size_t want) -> int {
    want = std::max<size_t>(want, 256);
    if (want <= have) return NIQKI_OK;
    if (*p) NQ_HIP(ix, hipFree(*p));
    *p = nullptr; have = 0;
    NQ_HIP(ix, hipMalloc(p, want));
    have = want;
    return NIQKI_OK;
  };
  // Nothing of the segment counts as built until the fill has gone through: an error on the way (out of
  // memory: grow() has freed the old buffer by then) must not leave seg_n naming ids that do not exist --
  // build_if_needed would take such a segment for a main index and put a delta on top of it.
  struct Uncommitted {
    niqki_index *ix;
    bool ok = false;
    ~Uncommitted() {
      if (!ok) { ix->seg_n = 0; ix->built = false; }
    }
  } commit{ix};
  ix->built = false;
  int rc;
  if ((rc = grow((void **)&ix->entries, ix->entries_bytes, (size_t)f_local * ix->d.R * n_tiles * sizeof(nq::Entry)))) return rc;
  if ((rc = grow((void **)&ix->slot_units, ix->slot_units_bytes, (size_t)n_tiles * (f_local + 1) * 4))) return rc;
  if ((rc = grow((void **)&ix->tile_base, ix->tile_base_bytes, (size_t)(n_tiles + 1) * 8))) return rc;
  ix->tile = tile;
  ix->n_tiles = n_tiles;
  ix->ptab_ok = false;
  ix->hmask_ok = false;
  ix->seg_n = N;
  ix->g_base = g_base;
  ix->align_log2 = (uint32_t)al;
  // line-aligned buckets carry padding ids behind their last id (see IndexView::padded)
  ix->padded = (al == 6 && tile <= nq::kPadMaxTile) ? 1u : 0u;
  ix->stripe = 0;
  if (stripe > 0 && n_tiles > 1 && n_tiles <= 64) {
    // the fullest tile must fit the tile size (it always does for B = 1)
    while (B > 1 && ((N + B - 1) / B + n_tiles - 1) / n_tiles * B > tile) B /= 2;
    ix->stripe = B;
  }
  if (n_tiles == 0) { commit.ok = true; ix->built = true; return NIQKI_OK; }
  {
    Span sp(ix, NIQKI_KC_BUILD);
    NQ_HIP(ix, nq::launch_build_sizes(view(ix), ix->slot_units, ix->tile_base, ix->stream));
  }
  std::vector<uint64_t> tb(n_tiles + 1);
  NQ_HIP(ix, hipMemcpyAsync(tb.data(), ix->tile_base, (size_t)(n_tiles + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  for (uint32_t t = 0; t < n_tiles; ++t)
    if (((tb[t + 1] - tb[t]) >> al) >= (1ull << 32))  // bucket starts are 32-bit unit counts
      return fail(ix, NIQKI_E_INVALID, "tile id array too large for 32-bit bucket starts");
  const uint64_t total_ids = tb[n_tiles];
  // + pad: the gather kernel reads up to 64 ids from a bucket's start whatever its length
  if ((rc = grow((void **)&ix->gids, ix->gids_bytes, (size_t)total_ids * 2 + 512))) return rc;
  if (ix->padded) {
    Span sp(ix, NIQKI_KC_BUILD);
    NQ_HIP(ix, nq::launch_pad_fill(ix->gids, total_ids, tile, ix->stream));
  }
  {
    Span sp(ix, NIQKI_KC_BUILD);
    NQ_HIP(ix, nq::launch_build_fill(view(ix), ix->entries, ix->gids, ix->stream));
  }
  // single-tile indexes: the per-slot class mask the gather kernel's own look-ups test first (IndexView::hmask;
  // NIQKI_HMASK=0 leaves it out).  One more pass over the table: 32 MB at S = 12 W = 10.
  {
    const char *hv = std::getenv("NIQKI_HMASK");
    if (n_tiles == 1 && !(hv && hv[0] == '0')) {
      if ((rc = grow((void **)&ix->hmask, ix->hmask_bytes, (size_t)f_local * 2))) return rc;
      Span sp(ix, NIQKI_KC_BUILD);
      NQ_HIP(ix, nq::launch_hmask(view(ix), ix->hmask, ix->stream));
      ix->hmask_ok = true;
    }
  }
  commit.ok = true;
  ix->built = true;
  return NIQKI_OK;
}
}  // namespace nqi

extern "C" {

int niqki_query_counts(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint16_t *counts,
                       uint64_t stride, int mem) {
  if (!ix || (!sketches && nq) || (!counts && nq)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (mem == NIQKI_MEM_DEVICE) return counts_dev(ix, sketches, ix->d.F, first_slot(ix), nq, counts, stride);
  if (stride < ix->n_genomes || (stride & 1)) return fail(ix, NIQKI_E_INVALID, "stride must be even and >= genome count");
  const uint32_t qb = ix->query_batch;
  for (uint32_t q0 = 0; q0 < nq; q0 += qb) {
    const uint32_t n = std::min(qb, nq - q0);
    int rc;
    if ((rc = ensure(ix, ix->ws_sk, (size_t)n * ix->d.F * 4))) return rc;
    if ((rc = ensure(ix, ix->ws_counts, (size_t)n * stride * 2))) return rc;
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, sketches + (size_t)q0 * ix->d.F, (size_t)n * ix->d.F * 4, hipMemcpyHostToDevice, ix->stream));
    NQ_HIP(ix, hipMemsetAsync(ix->ws_counts.p, 0, (size_t)n * stride * 2, ix->stream));
    if ((rc = counts_dev(ix, (const int32_t *)ix->ws_sk.p, ix->d.F, first_slot(ix), n, (uint16_t *)ix->ws_counts.p, stride))) return rc;
    NQ_HIP(ix, hipMemcpyAsync(counts + (size_t)q0 * stride, ix->ws_counts.p, (size_t)n * stride * 2, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  }
  return NIQKI_OK;
}

int niqki_query_counts32(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint32_t *counts, uint64_t stride, int mem) {
  if (!ix || (!sketches && nq) || (!counts && nq)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = build_if_needed(ix);
  if (rc) return rc;
  if (stride < ix->n_genomes || (stride & 1)) return fail(ix, NIQKI_E_INVALID, "stride must be even and >= genome count");
  const uint32_t qb = mem == NIQKI_MEM_DEVICE ? std::min<uint32_t>(nq, 4096) : ix->query_batch;
  const size_t planes = two_planes(ix) ? 2 : 1;
  for (uint32_t q0 = 0; q0 < nq; q0 += qb) {
    const uint32_t n = std::min(qb, nq - q0);
    const size_t plane = (size_t)n * stride * 2;
    if ((rc = ensure(ix, ix->ws_counts, std::max<size_t>(plane * planes, 4)))) return rc;
    uint16_t *c1 = (uint16_t *)ix->ws_counts.p, *c2 = planes == 2 ? (uint16_t *)((char *)ix->ws_counts.p + plane) : nullptr;
    NQ_HIP(ix, hipMemsetAsync(ix->ws_counts.p, 0, std::max<size_t>(plane * planes, 4), ix->stream));
    const int32_t *d_sk = sketches + (size_t)q0 * ix->d.F;
    uint32_t *d_out = counts + (size_t)q0 * stride;
    if (mem == NIQKI_MEM_HOST) {
      if ((rc = ensure(ix, ix->ws_sk, (size_t)n * ix->d.F * 4))) return rc;
      if ((rc = ensure(ix, ix->ws_misc, (size_t)n * stride * 4))) return rc;
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, d_sk, (size_t)n * ix->d.F * 4, hipMemcpyHostToDevice, ix->stream));
      d_sk = (const int32_t *)ix->ws_sk.p;
      d_out = (uint32_t *)ix->ws_misc.p;
    }
    if ((rc = counts_dev(ix, d_sk, ix->d.F, first_slot(ix), n, c1, stride, c2))) return rc;
    NQ_HIP(ix, nq::launch_plane_sum32(c1, c2, d_out, (uint64_t)n * stride, ix->stream));
    if (mem == NIQKI_MEM_HOST) {
      NQ_HIP(ix, hipMemcpyAsync(counts + (size_t)q0 * stride, d_out, (size_t)n * stride * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    }
  }
  return NIQKI_OK;
}

int niqki_hits_from_counts(niqki_index *ix, const uint16_t *counts, uint32_t nq, uint64_t stride,
                           uint32_t gid_begin, uint32_t n_gids, uint64_t *hit_off, uint32_t *hit_counts,
                           uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix || !hit_off || (!counts && nq)) return NIQKI_E_INVALID;
  if ((uint64_t)gid_begin + n_gids > stride) return fail(ix, NIQKI_E_INVALID, "gid range exceeds stride");
  if (ix->d.S > 15) return fail(ix, NIQKI_E_INVALID, "S = 16: u16 counters cannot hold a count of 2^16; use niqki_query");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (mem == NIQKI_MEM_DEVICE)
    return hits_dev(ix, counts, nq, stride, gid_begin, n_gids, (unsigned long long *)hit_off, hit_counts,
                    hit_gids, capacity, false, nullptr);
  int rc;
  if ((rc = ensure(ix, ix->ws_counts, std::max<size_t>((size_t)nq * stride * 2, 2)))) return rc;
  if ((rc = ensure(ix, ix->ws_hitoff, (size_t)(nq + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_hc, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;
  if ((rc = ensure(ix, ix->ws_hg, (size_t)std::max<uint64_t>(capacity, 1) * 4))) return rc;
  if (nq) NQ_HIP(ix, hipMemcpyAsync(ix->ws_counts.p, counts, (size_t)nq * stride * 2, hipMemcpyHostToDevice, ix->stream));
  uint64_t total = 0;
  rc = hits_dev(ix, (const uint16_t *)ix->ws_counts.p, nq, stride, gid_begin, n_gids,
                (unsigned long long *)ix->ws_hitoff.p, (uint32_t *)ix->ws_hc.p, (uint32_t *)ix->ws_hg.p,
                capacity, true, &total);
  if (rc && rc != NIQKI_E_CAPACITY) return rc;
  NQ_HIP(ix, hipMemcpyAsync(hit_off, ix->ws_hitoff.p, (size_t)(nq + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
  if (rc == NIQKI_OK && total) {
    NQ_HIP(ix, hipMemcpyAsync(hit_counts, ix->ws_hc.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipMemcpyAsync(hit_gids, ix->ws_hg.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
  }
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return rc;
}

int niqki_candidates_from_counts(niqki_index *ix, const uint16_t *counts, uint32_t nq, uint64_t stride,
                                 uint32_t n_gids, uint32_t threshold, uint32_t cap, int32_t *cand, int32_t *n_cand,
                                 int mem) {
  if (!ix || (nq && (!counts || !cand || !n_cand)) || n_gids > stride || cap == 0) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return fail(ix, NIQKI_E_INVALID, "niqki_candidates_from_counts is device-memory only");
  NQ_HIP(ix, hipSetDevice(ix->device));
  Span sp(ix, NIQKI_KC_HITS);
  NQ_HIP(ix, nq::launch_candidates(counts, stride, nq, n_gids, threshold, cap, cand, n_cand, ix->stream));
  return NIQKI_OK;
}

int niqki_query_counts_candidates(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint16_t *counts, uint64_t stride,
                                  uint32_t threshold, uint32_t cap, int32_t *cand, int32_t *n_cand, int mem) {
  if (!ix || (nq && (!sketches || !counts || !cand || !n_cand)) || cap == 0) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return fail(ix, NIQKI_E_INVALID, "niqki_query_counts_candidates is device-memory only");
  NQ_HIP(ix, hipSetDevice(ix->device));
  nq::CandOut co;
  co.cand = cand;
  co.n = n_cand;
  co.thr = threshold;
  co.cap = cap;
  return counts_dev(ix, sketches, ix->d.F, first_slot(ix), nq, counts, stride, nullptr, &co);
}

int niqki_query(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint64_t *hit_off,
                uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix || !hit_off || (!sketches && nq)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = build_if_needed(ix);
  if (rc) return rc;
  const uint32_t N = ix->built_n;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  if (mem == NIQKI_MEM_DEVICE) {
    const size_t plane = std::max<size_t>((size_t)nq * stride * 2, 2);
    if ((rc = ensure(ix, ix->ws_counts, plane * (two_planes(ix) ? 2 : 1)))) return rc;
    uint16_t *c1 = (uint16_t *)ix->ws_counts.p, *c2 = two_planes(ix) ? (uint16_t *)((char *)ix->ws_counts.p + plane) : nullptr;
    return query_hits_dev(ix, sketches, nq, c1, c2, stride, (unsigned long long *)hit_off, hit_counts, hit_gids, capacity, false, nullptr);
  }
  return query_to_host(ix, sketches, false, nq, hit_off, hit_counts, hit_gids, capacity);
}

int niqki_query_sequences(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec,
                          const uint32_t *entry_rec, uint32_t n_entry, uint64_t *hit_off,
                          uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (mem == NIQKI_MEM_DEVICE) {
    int rc = ensure(ix, ix->ws_sk, std::max<size_t>((size_t)n_entry * ix->d.F * 4, 4));
    if (rc) return rc;
    if ((rc = niqki_sketch(ix, seqs, rec_off, n_rec, entry_rec, n_entry, (int32_t *)ix->ws_sk.p, NIQKI_MEM_DEVICE))) return rc;
    return niqki_query(ix, (const int32_t *)ix->ws_sk.p, n_entry, hit_off, hit_counts, hit_gids, capacity, NIQKI_MEM_DEVICE);
  }
  std::vector<int32_t> sk((size_t)n_entry * ix->d.F);
  int rc = niqki_sketch(ix, seqs, rec_off, n_rec, entry_rec, n_entry, sk.data(), NIQKI_MEM_HOST);
  if (rc) return rc;
  return niqki_query(ix, sk.data(), n_entry, hit_off, hit_counts, hit_gids, capacity, NIQKI_MEM_HOST);
}

int niqki_get_sketches(niqki_index *ix, uint32_t begin, uint32_t n, int32_t *sketches, int mem) {
  if (!ix || (!sketches && n)) return NIQKI_E_INVALID;
  if ((uint64_t)begin + n > ix->n_genomes) return fail(ix, NIQKI_E_INVALID, "genome range out of bounds");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (n == 0) return NIQKI_OK;
  int32_t *d_sk = sketches;
  const size_t bytes = (size_t)n * ix->d.F * 4;
  if (mem == NIQKI_MEM_HOST) {
    int rc = ensure(ix, ix->ws_sk, bytes);
    if (rc) return rc;
    d_sk = (int32_t *)ix->ws_sk.p;
  }
  if (ix->resident_bytes) {
    // paged: the genomes' columns of every slot row, host -> device, then the usual transpose
    const uint32_t f_all = ix->full_end - ix->full_begin;
    const uint64_t n_pad = ((uint64_t)n + 63) / 64 * 64;
    int rc = ensure(ix, ix->pg_stage, (size_t)f_all * n_pad * 2);
    if (rc) return rc;
    NQ_HIP(ix, hipMemcpy2DAsync(ix->pg_stage.p, n_pad * 2, ix->host_store + begin, ix->host_cap * 2, (size_t)n * 2, f_all,
                                hipMemcpyHostToDevice, ix->stream));
    nq::Derived d = ix->d;
    d.slot_begin = ix->full_begin;
    d.slot_end = ix->full_end;
    NQ_HIP(ix, nq::launch_store_read(d, (const uint16_t *)ix->pg_stage.p, n_pad, 0, n, d_sk, ix->stream));
  } else
  NQ_HIP(ix, nq::launch_store_read(ix->d, ix->store, ix->cap, begin, n, d_sk, ix->stream));
  if (mem == NIQKI_MEM_HOST) {
    NQ_HIP(ix, hipMemcpyAsync(sketches, d_sk, bytes, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  }
  return NIQKI_OK;
}

int niqki_stage_raw_prefetch(niqki_index *ix, const niqki_raw_batch *b) {
  if (!ix || !b) return NIQKI_E_INVALID;
  if (!b->file_ptr || !b->file_off) return fail(ix, NIQKI_E_INVALID, "a prefetch takes the file_ptr form of a host batch");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (!ix->copy_stream) {
    NQ_HIP(ix, hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
    NQ_HIP(ix, hipEventCreateWithFlags(&ix->ev_copy, hipEventDisableTiming));
  }
  if (ix->pre.valid) NQ_HIP(ix, hipStreamSynchronize(ix->copy_stream));  // an unused one: its host bytes may go away now
  ix->pre.valid = false;
  const uint32_t nf = b->n_files;
  if (nf == 0) return NIQKI_OK;
  for (uint32_t f = 0; f < nf; ++f)
    if (b->file_off[f + 1] < b->file_off[f]) return fail(ix, NIQKI_E_INVALID, "file_off must be non-decreasing");
  const uint64_t T = b->file_off[nf];
  int rc = ensure(ix, ix->ws_raw2, (size_t)T + 2 * NIQKI_SEQ_PAD);
  if (rc) return rc;
  for (uint32_t f = 0; f < nf; ++f) {
    const uint64_t n = b->file_off[f + 1] - b->file_off[f];
    if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_raw2.p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->copy_stream));
  }
  NQ_HIP(ix, hipEventRecord(ix->ev_copy, ix->copy_stream));
  ix->pre.ptr.assign(b->file_ptr, b->file_ptr + nf);
  ix->pre.off.assign(b->file_off, b->file_off + nf + 1);
  ix->pre.valid = true;
  return NIQKI_OK;
}

int niqki_stage_raw(niqki_index *ix, const niqki_raw_batch *b, int mem, niqki_stage_info *info,
                    uint64_t *entry_hdr) {
  if (!ix || !b || !info) return NIQKI_E_INVALID;
  if (b->n_files && (!b->file_off || !b->file_type)) return NIQKI_E_INVALID;
  if (b->lines && b->n_files > 1) return fail(ix, NIQKI_E_INVALID, "lines mode frames one file per call");
  if (b->lines && b->max_entries == 0) return fail(ix, NIQKI_E_INVALID, "max_entries must be > 0");
  NQ_HIP(ix, hipSetDevice(ix->device));
  ix->staged.valid = false;
  ix->staged.sketched = false;
  *info = niqki_stage_info{0, 0, 0, 0};
  const uint32_t nf = b->n_files;
  const uint64_t T = nf ? b->file_off[nf] : 0;   // bytes handed over ("wire" bytes: packed files count as their containers)
  if (nf && !b->raw && !b->file_ptr && T) return NIQKI_E_INVALID;
  if (b->file_ptr && mem != NIQKI_MEM_HOST) return fail(ix, NIQKI_E_INVALID, "file_ptr needs the host memory space");
  // Packed FASTA files (file_type 'a': a container of niqki_pack_fasta): the device writes the file's own bytes back
  // first (nq::unpack_kernel), so everything from here on sees raw files at their raw offsets.
  bool any_packed = false;
  for (uint32_t f = 0; f < nf; ++f) any_packed |= b->file_type[f] == 'a';
  if (any_packed && (mem != NIQKI_MEM_HOST || !b->file_ptr || b->lines))
    return fail(ix, NIQKI_E_INVALID, "packed files (type 'a'): host memory, the file_ptr form, whole-file mode");
  std::vector<uint64_t> roff((size_t)nf + 1, 0);   // raw offsets of the files
  std::vector<nq::UnpackSeg> segs;
  uint64_t unpack_blocks = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    if (b->file_off[f + 1] < b->file_off[f]) return fail(ix, NIQKI_E_INVALID, "file_off must be non-decreasing");
    const uint64_t wire_len = b->file_off[f + 1] - b->file_off[f];
    uint64_t raw_len = wire_len;
    if (b->file_type[f] == 'a') {
      const uint8_t *c = b->file_ptr[f];
      if (!c || !nqp::valid(c, wire_len)) return fail(ix, NIQKI_E_INVALID, "file " + std::to_string(f) + " is not a well-formed packed container");
      nqp::PackHeader h;
      std::memcpy(&h, c, sizeof h);
      raw_len = h.raw_len;
      for (uint32_t k = 0; k < h.n_seg; ++k) {
        nqp::PackSeg ps;
        std::memcpy(&ps, c + sizeof(nqp::PackHeader) + (size_t)k * sizeof(nqp::PackSeg), sizeof ps);
        segs.push_back(nq::UnpackSeg{roff[f] + ps.raw_off, b->file_off[f] + h.payload_off + ps.pk_off, ps.count, ps.width, (uint32_t)unpack_blocks, 0u});
        unpack_blocks += (nqp::seg_raw_len(ps) + nq::kUnpackChunk - 1) / nq::kUnpackChunk;
      }
    } else if (any_packed && wire_len) {   // a raw file in a batch with packed ones: one raw segment
      if (wire_len > 0xFFFFFFFFull) return fail(ix, NIQKI_E_INVALID, "a raw file of 4 GiB or more cannot share a batch with packed files");
      segs.push_back(nq::UnpackSeg{roff[f], b->file_off[f], (uint32_t)wire_len, 0u, (uint32_t)unpack_blocks, 0u});
      unpack_blocks += (wire_len + nq::kUnpackChunk - 1) / nq::kUnpackChunk;
    }
    roff[f + 1] = roff[f] + raw_len;
  }
  if (unpack_blocks > 0x7FFFFFFFull) return fail(ix, NIQKI_E_INVALID, "raw batch too large");
  const uint64_t T_raw = roff[nf];
  // chunk table: chunks never span two files
  std::vector<uint8_t> meta((size_t)(nf + 1) * 12 + nf + 16);
  uint64_t *h_off = (uint64_t *)meta.data();
  uint32_t *h_first = (uint32_t *)(meta.data() + (size_t)(nf + 1) * 8);
  uint8_t *h_type = meta.data() + (size_t)(nf + 1) * 12;
  uint64_t chunks = 0;
  for (uint32_t f = 0; f < nf; ++f) {
    const uint8_t ty = b->file_type[f] == 'a' ? (uint8_t)'A' : b->file_type[f];
    if (ty != 'A' && ty != 'Q') return fail(ix, NIQKI_E_INVALID, "file_type must be 'A', 'Q' or 'a' (packed FASTA)");
    h_off[f] = roff[f];
    h_first[f] = (uint32_t)chunks;
    h_type[f] = ty;
    chunks += (roff[f + 1] - roff[f] + nq::kIngestChunk - 1) / nq::kIngestChunk;
  }
  if (chunks > 0x7FFFFFFFull) return fail(ix, NIQKI_E_INVALID, "raw batch too large");
  h_off[nf] = T_raw;
  h_first[nf] = (uint32_t)chunks;
  int rc;
  const uint8_t *d_raw = b->raw;
  bool prefetched = false;
  if (ix->pre.valid) {  // bytes a niqki_stage_raw_prefetch put on their way: this batch's, or dropped
    prefetched = mem == NIQKI_MEM_HOST && b->file_ptr && nf == ix->pre.ptr.size() &&
                 std::equal(ix->pre.ptr.begin(), ix->pre.ptr.end(), b->file_ptr) &&
                 std::equal(ix->pre.off.begin(), ix->pre.off.end(), b->file_off);
    ix->pre.valid = false;
    if (prefetched) {
      if (!any_packed) std::swap(ix->ws_raw, ix->ws_raw2);   // (packed: ws_raw2 stays the wire buffer, unpacked below)
      NQ_HIP(ix, hipStreamWaitEvent(ix->stream, ix->ev_copy, 0));
      d_raw = (const uint8_t *)ix->ws_raw.p;
    } else {
      NQ_HIP(ix, hipStreamSynchronize(ix->copy_stream));
    }
  }
  if (any_packed) {
    if (!prefetched) {   // the containers (and raw files) as they are, into the wire buffer
      if ((rc = ensure(ix, ix->ws_raw2, (size_t)T + 2 * NIQKI_SEQ_PAD))) return rc;
      for (uint32_t f = 0; f < nf; ++f) {
        const uint64_t n = b->file_off[f + 1] - b->file_off[f];
        if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_raw2.p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->stream));
      }
    }
    if ((rc = ensure(ix, ix->ws_raw, (size_t)T_raw + 2 * NIQKI_SEQ_PAD))) return rc;
    if ((rc = ensure(ix, ix->ws_useg, std::max<size_t>(segs.size() * sizeof(nq::UnpackSeg), 32)))) return rc;
    if (!segs.empty()) {
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_useg.p, segs.data(), segs.size() * sizeof(nq::UnpackSeg), hipMemcpyHostToDevice, ix->stream));
      Span sp(ix, NIQKI_KC_INGEST);
      NQ_HIP(ix, nq::launch_unpack((const nq::UnpackSeg *)ix->ws_useg.p, (uint32_t)segs.size(), (uint32_t)unpack_blocks,
                                   (const uint8_t *)ix->ws_raw2.p, (uint8_t *)ix->ws_raw.p, ix->stream));
    }
    d_raw = (const uint8_t *)ix->ws_raw.p;
  } else if (prefetched) {
  } else if (mem == NIQKI_MEM_HOST) {
    if ((rc = ensure(ix, ix->ws_raw, (size_t)T + 2 * NIQKI_SEQ_PAD))) return rc;
    if (b->file_ptr) {
      for (uint32_t f = 0; f < nf; ++f) {
        const uint64_t n = b->file_off[f + 1] - b->file_off[f];
        if (n) NQ_HIP(ix, hipMemcpyAsync((uint8_t *)ix->ws_raw.p + b->file_off[f], b->file_ptr[f], n, hipMemcpyHostToDevice, ix->stream));
      }
    } else if (T) {
      NQ_HIP(ix, hipMemcpyAsync(ix->ws_raw.p, b->raw, T, hipMemcpyHostToDevice, ix->stream));
    }
    d_raw = (const uint8_t *)ix->ws_raw.p;
  } else if ((uintptr_t)d_raw & 3) {
    return fail(ix, NIQKI_E_INVALID, "device raw bytes must be 4-byte aligned");
  }
  if ((rc = ensure(ix, ix->ws_fmeta, meta.size()))) return rc;
  if ((rc = ensure(ix, ix->ws_summ, std::max<size_t>((size_t)chunks * 20, 4)))) return rc;
  if ((rc = ensure(ix, ix->ws_chunk, std::max<size_t>((size_t)chunks * 16, 4)))) return rc;
  if ((rc = ensure(ix, ix->ws_fkept, (size_t)(nf + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_fnrec, (size_t)(nf + 1) * 4))) return rc;
  if ((rc = ensure(ix, ix->ws_misc, 256))) return rc;
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_fmeta.p, meta.data(), meta.size(), hipMemcpyHostToDevice, ix->stream));
  nq::IngestArgs a;
  a.raw = d_raw;
  a.file_off = (const uint64_t *)ix->ws_fmeta.p;
  a.chunk_first = (const uint32_t *)((uint8_t *)ix->ws_fmeta.p + (size_t)(nf + 1) * 8);
  a.file_type = (const uint8_t *)ix->ws_fmeta.p + (size_t)(nf + 1) * 12;
  a.n_files = nf;
  a.n_chunks = (uint32_t)chunks;
  a.summ = (uint32_t *)ix->ws_summ.p;
  a.chunk_out = (uint32_t *)ix->ws_chunk.p;
  a.file_kept = (uint64_t *)ix->ws_fkept.p;
  a.file_nrec = (uint32_t *)ix->ws_fnrec.p;
  a.totals = (uint64_t *)ix->ws_misc.p;
  a.seqs = nullptr;
  a.rec_off = nullptr;
  a.hdr_pos = nullptr;
  uint64_t totals[2] = {0, 0};
  {
    Span sp(ix, NIQKI_KC_INGEST);
    NQ_HIP(ix, nq::launch_ingest_scan(a, ix->stream));
  }
  NQ_HIP(ix, hipMemcpyAsync(totals, a.totals, 16, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));  // also: `meta` and the caller's raw bytes are consumed
  if (totals[0] > 0xFFFFFFF0ull) return fail(ix, NIQKI_E_INVALID, "too many records in one batch");
  const uint32_t n_rec = (uint32_t)totals[0];
  const uint64_t kept = totals[1];
  if ((rc = ensure(ix, ix->ws_recoff, (size_t)(n_rec + 1) * 8))) return rc;
  if ((rc = ensure(ix, ix->ws_hdrpos, std::max<size_t>((size_t)n_rec * 8, 8)))) return rc;
  if ((rc = ensure(ix, ix->ws_seq, (size_t)kept + 2 * NIQKI_SEQ_PAD))) return rc;
  a.seqs = (uint8_t *)ix->ws_seq.p;
  a.rec_off = (uint64_t *)ix->ws_recoff.p;
  a.hdr_pos = (uint64_t *)ix->ws_hdrpos.p;
  {
    Span sp(ix, NIQKI_KC_INGEST);
    NQ_HIP(ix, nq::launch_ingest_emit(a, ix->stream));
  }
  NQ_HIP(ix, hipMemcpyAsync(a.rec_off + n_rec, a.totals + 1, 8, hipMemcpyDeviceToDevice, ix->stream));
  NQ_HIP(ix, hipMemsetAsync(a.seqs + kept, 0, NIQKI_SEQ_PAD, ix->stream));
  uint32_t n_entry = nf;
  uint64_t consumed = T_raw;
  const uint32_t *d_entry = a.file_nrec;  // whole mode: entry f = the records of file f
  if (b->lines) {
    const uint32_t n_use = b->final ? n_rec : (n_rec ? n_rec - 1 : 0);
    if ((rc = ensure(ix, ix->ws_entry, (size_t)(b->max_entries + 1) * 4))) return rc;
    if ((rc = ensure(ix, ix->ws_ehdr, (size_t)b->max_entries * 8))) return rc;
    uint32_t *d_res = (uint32_t *)((uint8_t *)ix->ws_misc.p + 64);
    NQ_HIP(ix, nq::launch_ingest_entries(a.rec_off, a.hdr_pos, n_use, ix->d.K, b->max_entries,
                                         (uint32_t *)ix->ws_entry.p, (uint64_t *)ix->ws_ehdr.p, d_res, ix->stream));
    uint32_t res[2] = {0, 0};
    NQ_HIP(ix, hipMemcpyAsync(res, d_res, 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    n_entry = res[0];
    if (res[1] < n_rec)
      NQ_HIP(ix, hipMemcpyAsync(&consumed, a.hdr_pos + res[1], 8, hipMemcpyDeviceToHost, ix->stream));
    if (entry_hdr && n_entry)
      NQ_HIP(ix, hipMemcpyAsync(entry_hdr, ix->ws_ehdr.p, (size_t)n_entry * 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    d_entry = (const uint32_t *)ix->ws_entry.p;
  }
  ix->staged.valid = true;
  ix->staged.n_entry = n_entry;
  ix->staged.n_rec = n_rec;
  ix->staged.seq_bytes = kept;
  ix->staged.entry_rec = d_entry;
  info->n_entry = n_entry;
  info->n_rec = n_rec;
  info->consumed = consumed;
  info->seq_bytes = kept;
  return NIQKI_OK;
}

extern "C++" {
namespace nqi {
// sketches of the staged entries into their own buffer (once per staged batch; the other
// entry points keep using ws_sk, so they cannot disturb a staged batch)
int staged_sketch_ws(niqki_index *ix) {
  if (!ix->staged.valid) return fail(ix, NIQKI_E_STATE, "no staged batch (niqki_stage_raw first)");
  if (ix->staged.sketched) return NIQKI_OK;
  const uint32_t n = ix->staged.n_entry;
  int rc = ensure(ix, ix->ws_stsk, std::max<size_t>((size_t)n * ix->d.F * 4, 4));
  if (rc) return rc;
  rc = sketch_dev(ix, (const uint8_t *)ix->ws_seq.p, (const uint64_t *)ix->ws_recoff.p, ix->staged.n_rec,
                  ix->staged.entry_rec, n, (int32_t *)ix->ws_stsk.p, ix->staged.seq_bytes);
  if (rc) return rc;
  ix->staged.sketched = true;
  return NIQKI_OK;
}
}  // namespace nqi
}  // extern "C++"

int niqki_staged_sketch(niqki_index *ix, int32_t *sketches, int mem) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  const size_t bytes = (size_t)ix->staged.n_entry * ix->d.F * 4;
  if (!bytes) return NIQKI_OK;
  if (!sketches) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipMemcpyAsync(sketches, ix->ws_stsk.p, bytes,
                            mem == NIQKI_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ix->stream));
  if (mem == NIQKI_MEM_HOST) NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_staged_insert(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  return niqki_insert(ix, (const int32_t *)ix->ws_stsk.p, ix->staged.n_entry, NIQKI_MEM_DEVICE);
}

int niqki_staged_query(niqki_index *ix, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
                       uint64_t capacity, int mem) {
  if (!ix || !hit_off) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = staged_sketch_ws(ix);
  if (rc) return rc;
  if (mem == NIQKI_MEM_DEVICE)
    return niqki_query(ix, (const int32_t *)ix->ws_stsk.p, ix->staged.n_entry, hit_off, hit_counts, hit_gids,
                       capacity, NIQKI_MEM_DEVICE);
  if ((rc = build_if_needed(ix))) return rc;
  return query_to_host(ix, (const int32_t *)ix->ws_stsk.p, true, ix->staged.n_entry, hit_off, hit_counts,
                       hit_gids, capacity);
}

int niqki_staged_records(niqki_index *ix, uint64_t *rec_off, uint8_t *seqs, uint32_t *entry_rec,
                         uint64_t *hdr_pos) {
  if (!ix) return NIQKI_E_INVALID;
  if (!ix->staged.valid) return fail(ix, NIQKI_E_STATE, "no staged batch (niqki_stage_raw first)");
  NQ_HIP(ix, hipSetDevice(ix->device));
  const auto &st = ix->staged;
  if (rec_off) NQ_HIP(ix, hipMemcpyAsync(rec_off, ix->ws_recoff.p, (size_t)(st.n_rec + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
  if (seqs && st.seq_bytes) NQ_HIP(ix, hipMemcpyAsync(seqs, ix->ws_seq.p, st.seq_bytes, hipMemcpyDeviceToHost, ix->stream));
  if (entry_rec) NQ_HIP(ix, hipMemcpyAsync(entry_rec, st.entry_rec, (size_t)(st.n_entry + 1) * 4, hipMemcpyDeviceToHost, ix->stream));
  if (hdr_pos && st.n_rec) NQ_HIP(ix, hipMemcpyAsync(hdr_pos, ix->ws_hdrpos.p, (size_t)st.n_rec * 8, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

// ---- packed FASTA (nq_pack.h): host code, no device needed ----
size_t niqki_pack_bound(size_t n) { return nqp::pack_bound(n); }
size_t niqki_pack_fasta(const uint8_t *raw, size_t n, uint8_t *out, size_t capacity) {
  if (!raw || !out) return 0;
  return nqp::pack(raw, n, out, capacity);
}
int niqki_unpack_fasta(const uint8_t *container, size_t len, uint8_t *raw, size_t capacity, size_t *raw_len) {
  if (!container || !nqp::valid(container, len)) return NIQKI_E_INVALID;
  nqp::PackHeader h;
  std::memcpy(&h, container, sizeof h);
  if (raw_len) *raw_len = (size_t)h.raw_len;
  if (!raw) return NIQKI_OK;
  if (h.raw_len > capacity) return NIQKI_E_CAPACITY;
  return nqp::unpack(container, len, raw, capacity) ? NIQKI_OK : NIQKI_E_INVALID;
}

void *niqki_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void niqki_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int niqki_matrix_range(niqki_index *ix, uint32_t begin, uint32_t end, uint16_t *counts, uint64_t stride,
                       int mem) {
  if (!ix || begin > end || end > ix->n_genomes || (!counts && end > begin)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  // A paged index keeps its sketch store in page-locked host memory: the stored sketches of a batch are read
  // from there by the device (zero-copy, 2 bytes per cell), the counters then come from the paged walk.
  int rc = ix->resident_bytes ? NIQKI_OK : build_if_needed(ix);
  if (rc) return rc;
  const uint32_t n_all = ix->resident_bytes ? ix->n_genomes : ix->built_n;
  if (stride < n_all || (stride & 1)) return fail(ix, NIQKI_E_INVALID, "stride must be even and >= genome count");
  nq::Derived d_full = ix->d;
  const uint16_t *store_dev = ix->store;
  uint64_t store_cap = ix->cap;
  if (ix->resident_bytes) {
    d_full.slot_begin = ix->full_begin;
    d_full.slot_end = ix->full_end;
    void *dp = nullptr;
    NQ_HIP(ix, hipHostGetDevicePointer(&dp, ix->host_store, 0));
    store_dev = (const uint16_t *)dp;
    store_cap = ix->host_cap;
  }
  // The bucket co-occurrence count of (a, t) equals the hit count of genome a
  // for the stored sketch of t: both count the slots where the two sketches
  // hold the same valid fingerprint.  So the range is answered by the gather
  // kernel on the stored sketches of [begin, end).
  const uint32_t qb = std::min<uint32_t>(ix->query_batch, 256);
  for (uint32_t t0 = begin; t0 < end; t0 += qb) {
    const uint32_t n = std::min(qb, end - t0);
    if ((rc = ensure(ix, ix->ws_misc, (size_t)n * ix->d.F * 4))) return rc;
    NQ_HIP(ix, nq::launch_store_read(d_full, store_dev, store_cap, t0, n, (int32_t *)ix->ws_misc.p, ix->stream));
    uint16_t *dst = counts + (size_t)(t0 - begin) * stride;
    if (mem == NIQKI_MEM_DEVICE) {
      uint16_t *c2 = nullptr;
      if (two_planes(ix)) {
        if ((rc = ensure(ix, ix->ws_counts, (size_t)n * stride * 2))) return rc;
        c2 = (uint16_t *)ix->ws_counts.p;
      }
      if ((rc = counts_dev(ix, (const int32_t *)ix->ws_misc.p, ix->d.F, first_slot(ix), n, dst, stride, c2))) return rc;
      // uint16 counters whatever S (src/niqki_index.cpp:572): at S = 16 a count of 2^16 reads 0, as in the reference
      if (c2) NQ_HIP(ix, nq::launch_plane_add16(dst, c2, (uint64_t)n * stride, ix->stream));
    } else {
      const size_t plane = (size_t)n * stride * 2;
      if ((rc = ensure(ix, ix->ws_counts, plane * (two_planes(ix) ? 2 : 1)))) return rc;
      NQ_HIP(ix, hipMemsetAsync(ix->ws_counts.p, 0, plane * (two_planes(ix) ? 2 : 1), ix->stream));
      uint16_t *c2 = two_planes(ix) ? (uint16_t *)((char *)ix->ws_counts.p + plane) : nullptr;
      if ((rc = counts_dev(ix, (const int32_t *)ix->ws_misc.p, ix->d.F, first_slot(ix), n, (uint16_t *)ix->ws_counts.p, stride, c2))) return rc;
      if (c2) NQ_HIP(ix, nq::launch_plane_add16((uint16_t *)ix->ws_counts.p, c2, (uint64_t)n * stride, ix->stream));
      NQ_HIP(ix, hipMemcpyAsync(dst, ix->ws_counts.p, (size_t)n * stride * 2, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipStreamSynchronize(ix->stream));
    }
  }
  return NIQKI_OK;
}

// ---- dump / load ---------------------------------------------------------------

namespace {

// the page of a paged index that holds slot s (relative to the handle's first slot): pages never straddle 2^15
void page_of(const niqki_index *ix, uint32_t s, uint32_t &pb, uint32_t &pe) {
  const uint32_t f_all = ix->full_end - ix->full_begin, f_page = page_slots(ix);
  const uint32_t h0 = s / nq::kPassSlots * nq::kPassSlots, h1 = std::min(f_all, h0 + nq::kPassSlots);
  pb = h0 + (s - h0) / f_page * f_page;
  pe = std::min(h1, pb + f_page);
}

// slot_word (F+1 word positions, header excluded) computed on the device, copied to the host.
// Paged index: page after page (each page's index is built for it), the positions chained on the host.
int export_layout(niqki_index *ix, std::vector<uint64_t> &slot_word) {
  if (ix->resident_bytes) {
    const uint32_t f_all = ix->full_end - ix->full_begin;
    if (ix->pg_layout_n == ix->n_genomes && ix->pg_layout.size() == (size_t)f_all + 1) {   // (a dump asks slot group by slot group)
      slot_word = ix->pg_layout;
      return NIQKI_OK;
    }
    slot_word.assign((size_t)f_all + 1, 0);
    if (ix->n_genomes == 0) {
      for (uint32_t s = 0; s <= f_all; ++s) slot_word[s] = (uint64_t)s * ix->d.R;
      return NIQKI_OK;
    }
    uint64_t base = 0;
    std::vector<uint64_t> local;
    for (uint32_t pb = 0, pe = 0; pb < f_all; pb = pe) {
      page_of(ix, pb, pb, pe);
      int rc = load_page(ix, pb, pe);
      if (rc) return rc;
      nq::IndexView v = view(ix);
      local.assign((size_t)v.f_local + 1, 0);
      if ((rc = ensure(ix, ix->ws_misc, (size_t)(v.f_local + 1) * 8))) return rc;
      NQ_HIP(ix, nq::launch_export_layout(v, (unsigned long long *)ix->ws_misc.p, ix->stream));
      NQ_HIP(ix, hipMemcpyAsync(local.data(), ix->ws_misc.p, (size_t)(v.f_local + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipStreamSynchronize(ix->stream));
      for (uint32_t i = 0; i <= v.f_local; ++i) slot_word[pb + i] = base + local[i];
      base += local[v.f_local];
    }
    ix->pg_layout = slot_word;
    ix->pg_layout_n = ix->n_genomes;
    return NIQKI_OK;
  }
  int rc = build_single(ix);
  if (rc) return rc;
  nq::IndexView v = view(ix);
  slot_word.assign((size_t)v.f_local + 1, 0);
  if (v.n_tiles == 0) {  // empty index: one size word per bucket
    for (uint32_t s = 0; s <= v.f_local; ++s) slot_word[s] = (uint64_t)s * v.d.R;
    return NIQKI_OK;
  }
  if ((rc = ensure(ix, ix->ws_misc, (size_t)(v.f_local + 1) * 8))) return rc;
  NQ_HIP(ix, nq::launch_export_layout(v, (unsigned long long *)ix->ws_misc.p, ix->stream));
  NQ_HIP(ix, hipMemcpyAsync(slot_word.data(), ix->ws_misc.p, (size_t)(v.f_local + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

// payload of slots [s0, s1) to host memory; slot_word device copy is in ws_misc (export_layout ran)
int export_slots(niqki_index *ix, const std::vector<uint64_t> &slot_word, uint32_t s0, uint32_t s1, uint8_t *dst) {
  const uint64_t words = slot_word[s1] - slot_word[s0];
  if (words == 0) return NIQKI_OK;
  if (ix->resident_bytes) {
    if (ix->n_genomes == 0) { std::memset(dst, 0, words * 4); return NIQKI_OK; }
    // piece by piece of the pages that hold the slots; a page's word positions are made again when it comes in
    for (uint32_t a = s0; a < s1;) {
      uint32_t pb, pe;
      page_of(ix, a, pb, pe);
      const uint32_t b = std::min(s1, pe);
      int rc = load_page(ix, pb, pe);
      if (rc) return rc;
      nq::IndexView v = view(ix);
      if ((rc = ensure(ix, ix->ws_misc, (size_t)(v.f_local + 1) * 8))) return rc;
      NQ_HIP(ix, nq::launch_export_layout(v, (unsigned long long *)ix->ws_misc.p, ix->stream));
      const uint64_t w = slot_word[b] - slot_word[a];
      if ((rc = ensure(ix, ix->ws_counts, std::max<uint64_t>(w, 1) * 4))) return rc;
      NQ_HIP(ix, nq::launch_export(v, (const unsigned long long *)ix->ws_misc.p, (uint32_t *)ix->ws_counts.p, a - pb, b - pb, ix->stream));
      NQ_HIP(ix, hipMemcpyAsync(dst + (slot_word[a] - slot_word[s0]) * 4, ix->ws_counts.p, w * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_HIP(ix, hipStreamSynchronize(ix->stream));
      a = b;
    }
    return NIQKI_OK;
  }
  nq::IndexView v = view(ix);
  if (v.n_tiles == 0) { std::memset(dst, 0, words * 4); return NIQKI_OK; }
  int rc = ensure(ix, ix->ws_counts, words * 4);
  if (rc) return rc;
  NQ_HIP(ix, nq::launch_export(v, (const unsigned long long *)ix->ws_misc.p, (uint32_t *)ix->ws_counts.p, s0, s1, ix->stream));
  NQ_HIP(ix, hipMemcpyAsync(dst, ix->ws_counts.p, words * 4, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

}  // namespace

int niqki_export_dump_header(niqki_index *ix, uint8_t header[24]) {
  if (!ix || !header) return NIQKI_E_INVALID;
  uint32_t hdr[6] = {ix->d.S, ix->d.K, ix->d.H, ix->d.W, ix->d.min_score, ix->n_genomes};
  std::memcpy(header, hdr, 24);
  return NIQKI_OK;
}

int niqki_export_dump_layout(niqki_index *ix, uint64_t *slot_bytes) {
  if (!ix || !slot_bytes) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  std::vector<uint64_t> sw;
  int rc = export_layout(ix, sw);
  if (rc) return rc;
  for (size_t i = 0; i < sw.size(); ++i) slot_bytes[i] = sw[i] * 4;
  return NIQKI_OK;
}

int niqki_export_dump_slots(niqki_index *ix, uint32_t slot_begin, uint32_t slot_end, uint8_t *buf,
                            uint64_t capacity, uint64_t *size) {
  if (!ix || !size || slot_begin > slot_end) return NIQKI_E_INVALID;
  if (slot_end > (ix->resident_bytes ? ix->full_end - ix->full_begin : ix->d.slot_end - ix->d.slot_begin)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  std::vector<uint64_t> sw;
  int rc = export_layout(ix, sw);
  if (rc) return rc;
  *size = (sw[slot_end] - sw[slot_begin]) * 4;
  if (!buf) return NIQKI_OK;
  if (capacity < *size) return NIQKI_E_CAPACITY;
  return export_slots(ix, sw, slot_begin, slot_end, buf);
}

int niqki_export_dump(niqki_index *ix, uint8_t *buf, uint64_t capacity, uint64_t *size) {
  if (!ix || !size) return NIQKI_E_INVALID;
  if (first_slot(ix) != 0 || (ix->resident_bytes ? ix->full_end : ix->d.slot_end) != ix->d.F)
    return fail(ix, NIQKI_E_STATE, "export needs a whole-range handle");
  NQ_HIP(ix, hipSetDevice(ix->device));
  std::vector<uint64_t> sw;
  int rc = export_layout(ix, sw);
  if (rc) return rc;
  const uint32_t F = ix->d.F;
  *size = 24 + sw[F] * 4;
  if (!buf) return NIQKI_OK;
  if (capacity < *size) return NIQKI_E_CAPACITY;
  niqki_export_dump_header(ix, buf);
  // chunks of whole slots, at most ~256 MiB of device staging each
  uint32_t s0 = 0;
  while (s0 < F) {
    uint32_t s1 = s0 + 1;
    while (s1 < F && (sw[s1 + 1] - sw[s0]) * 4 <= (256ull << 20)) ++s1;
    if ((rc = export_slots(ix, sw, s0, s1, buf + 24 + sw[s0] * 4))) return rc;
    s0 = s1;
  }
  return NIQKI_OK;
}

int niqki_import_begin(const niqki_params *params, const uint8_t header[24], niqki_index **out) {
  if (!params || !header || !out) return NIQKI_E_INVALID;
  uint32_t hdr[6];
  std::memcpy(hdr, header, 24);
  niqki_params p = *params;
  p.S = hdr[0]; p.K = hdr[1]; p.H = hdr[2]; p.W = hdr[3]; p.min_score = hdr[4];
  // slot_begin / slot_end stay the caller's: a slot shard loads only its own slots of the dump
  niqki_index *ix = nullptr;
  int rc = niqki_create(&p, &ix);
  if (rc) return rc;
  const uint32_t N = hdr[5];
  rc = reserve_store(ix, std::max<uint32_t>(N, 1));
  hipError_t e = hipSuccess;
  if (!rc && ix->resident_bytes) std::memset(ix->host_store, 0xFF, (size_t)(ix->full_end - ix->full_begin) * ix->host_cap * 2);
  else if (!rc) e = hipMemsetAsync(ix->store, 0xFF, (size_t)(ix->d.slot_end - ix->d.slot_begin) * ix->cap * 2, ix->stream);
  if (rc || e != hipSuccess) {
    g_create_err = rc ? ix->err : std::string(hipGetErrorString(e));
    niqki_destroy(ix);
    return rc ? rc : NIQKI_E_HIP;
  }
  ix->n_genomes = N;  // ids are validated against this while the slots arrive
  ix->built = false;
  *out = ix;
  return NIQKI_OK;
}

int niqki_import_slots(niqki_index *ix, uint32_t slot_begin, uint32_t slot_end, const uint8_t *buf, uint64_t len,
                       uint64_t *consumed) {
  if (!ix || !buf || slot_begin > slot_end || slot_end > ix->d.F) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  const uint32_t n_slots = slot_end - slot_begin;
  const uint64_t R = ix->d.R, n_words = len / 4;
  // sequential walk of the bucket sizes (they chain), recording where each slot starts
  std::vector<uint64_t> slot_word((size_t)n_slots + 1);
  uint64_t w = 0;
  for (uint32_t i = 0; i < n_slots; ++i) {
    slot_word[i] = w;
    for (uint64_t fp = 0; fp < R; ++fp) {
      if (w >= n_words) return fail(ix, NIQKI_E_INVALID, "dump payload ends inside a slot");
      uint32_t sz;
      std::memcpy(&sz, buf + w * 4, 4);
      w += 1 + (uint64_t)sz;
    }
  }
  if (w > n_words) return fail(ix, NIQKI_E_INVALID, "dump payload ends inside a bucket");
  slot_word[n_slots] = w;
  if (consumed) *consumed = w * 4;
  // the part of [slot_begin, slot_end) this shard owns (all of it for a whole-range handle)
  const uint32_t my0 = ix->resident_bytes ? ix->full_begin : ix->d.slot_begin, my1 = ix->resident_bytes ? ix->full_end : ix->d.slot_end;
  const uint32_t own0 = std::max(slot_begin, my0), own1 = std::min(slot_end, my1);
  if (own0 >= own1) return NIQKI_OK;
  const uint32_t n_own = own1 - own0;
  const uint64_t w0 = slot_word[own0 - slot_begin], w1 = slot_word[own1 - slot_begin];
  std::vector<uint64_t> own_word(slot_word.begin() + (own0 - slot_begin), slot_word.begin() + (own1 - slot_begin) + 1);
  for (auto &x : own_word) x -= w0;
  int rc;
  if ((rc = ensure(ix, ix->ws_counts, std::max<uint64_t>((w1 - w0) * 4, 4)))) return rc;
  if ((rc = ensure(ix, ix->ws_misc, (size_t)(n_own + 1) * 8 + 8))) return rc;
  uint8_t *d_slot = (uint8_t *)ix->ws_misc.p;
  uint32_t *d_bad = (uint32_t *)(d_slot + (size_t)(n_own + 1) * 8);
  NQ_HIP(ix, hipMemcpyAsync(ix->ws_counts.p, buf + w0 * 4, (w1 - w0) * 4, hipMemcpyHostToDevice, ix->stream));
  NQ_HIP(ix, hipMemcpyAsync(d_slot, own_word.data(), (size_t)(n_own + 1) * 8, hipMemcpyHostToDevice, ix->stream));
  NQ_HIP(ix, hipMemsetAsync(d_bad, 0, 4, ix->stream));
  if (ix->resident_bytes) {
    // paged: the slots' rows are made in a device block and copied to the host store
    const uint64_t cap2 = ((uint64_t)std::max<uint32_t>(ix->n_genomes, 1) + 63) / 64 * 64;
    if ((rc = ensure(ix, ix->pg_stage, (size_t)n_own * cap2 * 2))) return rc;
    NQ_HIP(ix, hipMemsetAsync(ix->pg_stage.p, 0xFF, (size_t)n_own * cap2 * 2, ix->stream));
    NQ_HIP(ix, nq::launch_import(ix->d, (const uint32_t *)ix->ws_counts.p, (const uint64_t *)d_slot, (uint16_t *)ix->pg_stage.p, cap2,
                                 ix->n_genomes, d_bad, 0, n_own, ix->stream));
    if (ix->n_genomes)
      NQ_HIP(ix, hipMemcpy2DAsync(ix->host_store + (size_t)(own0 - my0) * ix->host_cap, ix->host_cap * 2, ix->pg_stage.p, cap2 * 2,
                                  (size_t)ix->n_genomes * 2, n_own, hipMemcpyDeviceToHost, ix->stream));
  } else
  // rows of the store are shard-local slots
  NQ_HIP(ix, nq::launch_import(ix->d, (const uint32_t *)ix->ws_counts.p, (const uint64_t *)d_slot, ix->store, ix->cap,
                               ix->n_genomes, d_bad, own0 - ix->d.slot_begin, n_own, ix->stream));
  uint32_t bad = 0;
  NQ_HIP(ix, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  if (bad) return fail(ix, NIQKI_E_INVALID, "dump holds genome ids >= genome count");
  return NIQKI_OK;
}

int niqki_import_dump(const niqki_params *params, const uint8_t *buf, uint64_t len, uint64_t *consumed,
                      niqki_index **out) {
  if (!params || !buf || !out || len < 24) return NIQKI_E_INVALID;
  niqki_index *ix = nullptr;
  int rc = niqki_import_begin(params, buf, &ix);
  if (rc) return rc;
  // groups of whole slots, ~256 MiB of payload each
  const uint32_t F = ix->d.F;
  const uint64_t R = ix->d.R;
  uint64_t pos = 24;
  uint32_t s0 = 0;
  while (s0 < F) {
    // find how many slots fit: walk sizes (cheap; import_slots walks them again for the device)
    uint64_t p = pos;
    uint32_t s1 = s0;
    while (s1 < F && (p - pos) <= (256ull << 20)) {
      for (uint64_t fp = 0; fp < R; ++fp) {
        if (p + 4 > len) { g_create_err = "dump payload is truncated"; niqki_destroy(ix); return NIQKI_E_INVALID; }
        uint32_t sz;
        std::memcpy(&sz, buf + p, 4);
        p += 4 + (uint64_t)sz * 4;
      }
      ++s1;
    }
    if (p > len) { g_create_err = "dump payload is truncated"; niqki_destroy(ix); return NIQKI_E_INVALID; }
    uint64_t used = 0;
    rc = niqki_import_slots(ix, s0, s1, buf + pos, p - pos, &used);
    if (rc) { g_create_err = ix->err; niqki_destroy(ix); return rc; }
    pos += used;
    s0 = s1;
  }
  if (consumed) *consumed = pos;
  *out = ix;
  return NIQKI_OK;
}

int niqki_query_gathered(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint64_t *gathered, int mem) {
  if (!ix || (!sketches && nq) || (!gathered && nq)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (ix->resident_bytes) return fail(ix, NIQKI_E_STATE, "niqki_query_gathered is not available on a paged index (resident_bytes)");
  int rc = build_single(ix);
  if (rc) return rc;
  if (nq == 0) return NIQKI_OK;
  const int32_t *d_sk = sketches;
  if (mem == NIQKI_MEM_HOST) {
    if ((rc = ensure(ix, ix->ws_sk, (size_t)nq * ix->d.F * 4))) return rc;
    NQ_HIP(ix, hipMemcpyAsync(ix->ws_sk.p, sketches, (size_t)nq * ix->d.F * 4, hipMemcpyHostToDevice, ix->stream));
    d_sk = (const int32_t *)ix->ws_sk.p;
  }
  if ((rc = ensure(ix, ix->ws_misc, (size_t)nq * 8))) return rc;
  NQ_HIP(ix, hipMemsetAsync(ix->ws_misc.p, 0, (size_t)nq * 8, ix->stream));
  if (ix->built_n) NQ_HIP(ix, nq::launch_gathered(view(ix), d_sk, nq, (unsigned long long *)ix->ws_misc.p, ix->stream));
  NQ_HIP(ix, hipMemcpyAsync(gathered, ix->ws_misc.p, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  return NIQKI_OK;
}

int niqki_get_stat(const niqki_index *ix, const char *key, uint64_t *value) {
  if (!ix || !key || !value) return NIQKI_E_INVALID;
  const uint32_t f_all = ix->resident_bytes ? ix->full_end - ix->full_begin : ix->d.slot_end - ix->d.slot_begin;
  if (!std::strcmp(key, "store_bytes")) { *value = (uint64_t)f_all * (ix->resident_bytes ? ix->host_cap : ix->cap) * 2; return NIQKI_OK; }
  if (!std::strcmp(key, "index_bytes")) { *value = ix->built ? (uint64_t)ix->entries_bytes + ix->gids_bytes + ix->ptab_bytes + ix->hmask_bytes + ix->alt.entries_bytes + ix->alt.gids_bytes + ix->alt.ptab_bytes + ix->alt.hmask_bytes : 0; return NIQKI_OK; }
  if (!std::strcmp(key, "delta_genomes")) { *value = ix->delta_n; return NIQKI_OK; }
  if (!std::strcmp(key, "tiles")) { *value = ix->n_tiles; return NIQKI_OK; }
  if (!std::strcmp(key, "class_mask")) { *value = (ix->built && ix->hmask_ok) ? 1 : 0; return NIQKI_OK; }
  if (!std::strcmp(key, "last_gather_form")) { *value = ix->last_form; return NIQKI_OK; }
  if (!std::strcmp(key, "last_hits_form")) { *value = ix->last_hits_form; return NIQKI_OK; }
  if (!std::strcmp(key, "page_slots")) { *value = ix->resident_bytes ? page_slots(ix) : f_all; return NIQKI_OK; }
  if (!std::strcmp(key, "pages")) {
    const uint32_t ps = ix->resident_bytes ? page_slots(ix) : f_all;
    uint64_t n = 0;   // (a paged handle's pages do not straddle the halves of 2^15 slots)
    for (uint32_t h0 = 0; ps && h0 < f_all; h0 += ix->resident_bytes ? nq::kPassSlots : f_all)
      n += (std::min(f_all - h0, ix->resident_bytes ? nq::kPassSlots : f_all) + ps - 1) / ps;
    *value = n;
    return NIQKI_OK;
  }
  return NIQKI_E_INVALID;
}

int niqki_profile_enable(niqki_index *ix, int on) {
  if (!ix) return NIQKI_E_INVALID;
  int rc = collect_spans(ix);
  ix->prof = on != 0;
  return rc;
}

int niqki_profile_reset(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  int rc = collect_spans(ix);
  for (int i = 0; i < NIQKI_KC_COUNT; ++i) { ix->prof_ms[i] = 0; ix->prof_n[i] = 0; }
  return rc;
}

int niqki_profile_read(niqki_index *ix, int kc, double *ms, uint64_t *launches) {
  if (!ix || kc < 0 || kc >= NIQKI_KC_COUNT) return NIQKI_E_INVALID;
  int rc = collect_spans(ix);
  if (ms) *ms = ix->prof_ms[kc];
  if (launches) *launches = ix->prof_n[kc];
  return rc;
}

void niqki_synth_genome_host(uint64_t seed, uint32_t family, uint32_t member, uint32_t rate14,
                             uint64_t len, uint8_t *out) {
  const uint64_t ka = nq::synth_key_anc(seed, family), km = nq::synth_key_mut(seed, family, member);
  for (uint64_t blk = 0; blk * 32 < len; ++blk) {
    uint64_t codes = nq::synth_block(ka, km, rate14, blk);
    for (uint32_t j = 0; j < 32 && blk * 32 + j < len; ++j)
      out[blk * 32 + j] = nq::synth_ascii((uint32_t)(codes >> (2 * j)) & 3u);
  }
}

int niqki_synth_reads(niqki_index *ix, uint64_t seed, const uint32_t *family, const uint32_t *member, const uint32_t *rate14,
                      const uint64_t *offset, const uint32_t *read_id, uint32_t read_rate14, uint32_t n, uint32_t len,
                      uint64_t stride, uint8_t *out, int mem) {
  if (!ix || (n && (!family || !member || !rate14 || !offset || !read_id || !out)) || stride < len) return NIQKI_E_INVALID;
  if (mem == NIQKI_MEM_HOST) {
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t ka = nq::synth_key_anc(seed, family[i]), km = nq::synth_key_mut(seed, family[i], member[i]);
      const uint64_t kr = nq::synth_key_read(seed, family[i], read_id[i]);
      uint64_t blk = ~0ull, codes = 0;
      for (uint32_t j = 0; j < len; ++j) {
        const uint64_t p = offset[i] + j;
        if ((p >> 5) != blk) { blk = p >> 5; codes = nq::synth_block2(ka, km, rate14[i], kr, read_rate14, blk); }
        out[(uint64_t)i * stride + j] = nq::synth_ascii((uint32_t)(codes >> (2 * (p & 31))) & 3u);
      }
    }
    return NIQKI_OK;
  }
  NQ_HIP(ix, hipSetDevice(ix->device));
  NQ_HIP(ix, nq::launch_synth_reads(seed, family, member, rate14, offset, read_id, read_rate14, n, len, stride, out, ix->stream));
  return NIQKI_OK;
}

int niqki_measure_alu(niqki_index *ix, int what, double ms, double *rate) {
  if (!ix || !rate || what < 0 || what > 5 || !(ms > 0)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (what == 4) {   // streaming copy of 1 GiB: bytes read + bytes written per second
    const uint64_t bytes = 1ull << 30;
    void *a_ = nullptr, *b_ = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a_, bytes);
    if (e == hipSuccess) e = hipMalloc(&b_, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a_, 1, bytes, ix->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = nq::launch_copy_probe(a_, b_, bytes, ix->stream);
    const int reps = 8;
    if (e == hipSuccess) e = hipEventRecord(e0, ix->stream);
    for (int r = 0; r < reps && e == hipSuccess; ++r) e = nq::launch_copy_probe(a_, b_, bytes, ix->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ix->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, e0, e1);
    if (a_) (void)hipFree(a_);
    if (b_) (void)hipFree(b_);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(ix, e == hipErrorOutOfMemory ? NIQKI_E_NOMEM : NIQKI_E_HIP, std::string("copy probe: ") + hipGetErrorString(e));
    *rate = f > 0 ? 2.0 * reps * (double)bytes / (f * 1e-3) : 0.0;
    return NIQKI_OK;
  }
  int rc = ensure(ix, ix->ws_misc, 256);
  if (rc) return rc;
  hipEvent_t a = nullptr, b = nullptr;
  struct Events {   // destroyed on every way out
    hipEvent_t &a, &b;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } guard{a, b};
  NQ_HIP(ix, hipEventCreate(&a));
  NQ_HIP(ix, hipEventCreate(&b));
  auto run = [&](uint32_t iters, double &t_ms, uint64_t &units) -> hipError_t {
    hipError_t e = hipEventRecord(a, ix->stream);
    if (e == hipSuccess) e = nq::launch_alu_probe(what, iters, (uint32_t *)ix->ws_misc.p, &units, ix->stream);
    if (e == hipSuccess) e = hipEventRecord(b, ix->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, a, b);
    t_ms = f;
    return e;
  };
  double t = 0;
  uint64_t units = 0;
  uint32_t iters = 256;
  NQ_HIP(ix, run(iters, t, units));                       // warm-up + calibration
  iters = (uint32_t)std::min<double>(1e7, std::max<double>(256, iters * ms / std::max(t, 1e-3)));
  NQ_HIP(ix, run(iters, t, units));
  *rate = t > 0 ? (double)units / (t * 1e-3) : 0.0;
  return NIQKI_OK;
}

int niqki_synth_genomes(niqki_index *ix, uint64_t seed, const uint32_t *family, const uint32_t *member,
                        const uint32_t *rate14, uint32_t n, uint64_t len, uint64_t stride, uint8_t *out,
                        int mem) {
  if (!ix || (n && (!family || !member || !rate14 || !out)) || stride < len) return NIQKI_E_INVALID;
  if (mem == NIQKI_MEM_HOST) {
    for (uint32_t i = 0; i < n; ++i)
      niqki_synth_genome_host(seed, family[i], member[i], rate14[i], len, out + (uint64_t)i * stride);
    return NIQKI_OK;
  }
  NQ_HIP(ix, hipSetDevice(ix->device));
  NQ_HIP(ix, nq::launch_synth(seed, family, member, rate14, n, len, stride, out, ix->stream));
  return NIQKI_OK;
}

}  // extern "C"
