// nq_api_query.hip -- Index::query_sketch / query_range behind the C ABI (src/niqki_index.cpp:570-687): counter
// launches (resident, delta segment, paged walks), threshold + order (counter rows or hit lists), host staging,
// the matrix path, and the counter / candidate calls of niqki_hip_bench.h.
#include "nq_handle.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace nqi {

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

// the built index takes the hit-list form of query_hits_dev (one small tile, one segment, resident, one plane)
static bool hit_lists_apply(const niqki_index *ix) {
  const uint32_t N = ix->built_n;
  return ix->hit_lists && N && !ix->resident_bytes && !two_planes(ix) && ix->n_tiles == 1 && ix->delta_n == 0 &&
         ix->tile <= nq::kHitListMaxTile && ix->g_base == 0 && N <= 65536u;
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
  const bool lists = nq && hit_lists_apply(ix);
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
  // queries per round of launches: option "query_batch" bounds the counter rows (2N bytes per query); an index that
  // takes the hit-list form writes rows only for the rare overflowing query and is small (<= 12 288 genomes), so a
  // whole staged batch of short reads goes through in one round (64 rounds of 1024 cost the lines-mode host path
  // twice its kernels' time)
  const uint32_t qb = hit_lists_apply(ix) ? std::max<uint32_t>(ix->query_batch, 65536u) : ix->query_batch;
  const size_t planes = two_planes(ix) ? 2 : 1;
  std::vector<unsigned long long> off(std::min(qb, nq) + 1);
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

}  // namespace nqi

using namespace nqi;

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

// ---- sketches made ahead: batch i + 1 is sketched beside batch i's gather and hit kernels -------------------------
// The sketch kernel is bound by vector instruction issue, the gather launch by random HBM lines; a step of the query
// path is the one after the other (18.9 + 7.3 ms per 4096 genomes).  With the coming batch's sketch kernel on a lane of
// its own the two overlap where either leaves CUs idle (the tails of both launches): + 5 % genomes/s.
int niqki_sketch_ahead(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec, const uint32_t *entry_rec,
                       uint32_t n_entry, int mem) {
  if (!ix || (!seqs && n_rec) || !rec_off) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return fail(ix, NIQKI_E_INVALID, "niqki_sketch_ahead is device-memory only (host records: niqki_query_sequences)");
  if (!entry_rec && n_entry != n_rec) return fail(ix, NIQKI_E_INVALID, "n_entry must equal n_rec without entry_rec");
  if (ix->ahead_n >= 2) return fail(ix, NIQKI_E_STATE, "two batches are sketched ahead already: niqki_query_ahead takes the older one");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (!ix->sk_stream) {
    // the sketch lane never outranks the handle's stream: what a query is waiting for runs there
    int cus = 0;
    NQ_HIP(ix, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device));
    if (ix->sk_lane_cus && (int)ix->sk_lane_cus < cus) {
      std::vector<uint32_t> mask(((size_t)cus + 31) / 32, 0u);
      for (int c = cus - (int)ix->sk_lane_cus; c < cus; ++c) mask[(size_t)c / 32] |= 1u << (c % 32);
      NQ_HIP(ix, hipExtStreamCreateWithCUMask(&ix->sk_stream, (uint32_t)mask.size(), mask.data()));
    } else if (ix->stream_prio_set) {
      int least = 0, greatest = 0;
      NQ_HIP(ix, hipDeviceGetStreamPriorityRange(&least, &greatest));
      NQ_HIP(ix, hipStreamCreateWithPriority(&ix->sk_stream, hipStreamNonBlocking, least));
    } else {
      NQ_HIP(ix, hipStreamCreateWithFlags(&ix->sk_stream, hipStreamNonBlocking));
    }
  }
  auto &a = ix->ahead[(ix->ahead_head + ix->ahead_n) & 1u];
  if (!a.done) {
    NQ_HIP(ix, hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
    NQ_HIP(ix, hipEventCreateWithFlags(&a.used, hipEventDisableTiming));
  }
  const size_t bytes = std::max<size_t>((size_t)n_entry * ix->d.F * 4, 4);
  if (bytes > a.sk.n && a.used_set) NQ_HIP(ix, hipEventSynchronize(a.used));   // (growing the slot frees what its last query read)
  int rc = ensure(ix, a.sk, bytes);
  if (rc) return rc;
  if (a.used_set) NQ_HIP(ix, hipStreamWaitEvent(ix->sk_stream, a.used, 0));
  uint64_t total = ix->record_len_hint * n_entry;
  if (total == 0 && n_entry) {   // only to pick the launch shape: the last offset
    NQ_HIP(ix, hipMemcpyAsync(&total, rec_off + n_rec, 8, hipMemcpyDeviceToHost, ix->sk_stream));
    NQ_HIP(ix, hipStreamSynchronize(ix->sk_stream));
  }
  hipStream_t main = ix->stream;
  ix->stream = ix->sk_stream;      // (sketch_dev and its profiling spans enqueue on the handle's current stream)
  rc = sketch_dev(ix, seqs, rec_off, n_rec, entry_rec, n_entry, (int32_t *)a.sk.p, total);
  ix->stream = main;
  if (rc) return rc;
  NQ_HIP(ix, hipEventRecord(a.done, ix->sk_stream));
  a.n_entry = n_entry;
  ix->ahead_n += 1;
  return NIQKI_OK;
}

int niqki_query_ahead(niqki_index *ix, uint32_t *n_entry, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity,
                      int32_t *sketches, int mem) {
  if (!ix || !hit_off) return NIQKI_E_INVALID;
  if (ix->ahead_n == 0) return fail(ix, NIQKI_E_STATE, "no batch sketched ahead (niqki_sketch_ahead first)");
  NQ_HIP(ix, hipSetDevice(ix->device));
  auto &a = ix->ahead[ix->ahead_head];
  if (n_entry) *n_entry = a.n_entry;
  NQ_HIP(ix, hipStreamWaitEvent(ix->stream, a.done, 0));
  int rc;
  if (mem == NIQKI_MEM_DEVICE) {
    rc = niqki_query(ix, (const int32_t *)a.sk.p, a.n_entry, hit_off, hit_counts, hit_gids, capacity, NIQKI_MEM_DEVICE);
  } else {
    if ((rc = build_if_needed(ix))) return rc;
    rc = query_to_host(ix, (const int32_t *)a.sk.p, true, a.n_entry, hit_off, hit_counts, hit_gids, capacity);
  }
  if (rc == NIQKI_E_CAPACITY) return rc;   // the batch stays the oldest one: the same call again with larger arrays
  if (!rc && sketches && a.n_entry)
    NQ_HIP(ix, hipMemcpyAsync(sketches, a.sk.p, (size_t)a.n_entry * ix->d.F * 4,
                              mem == NIQKI_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ix->stream));
  NQ_HIP(ix, hipEventRecord(a.used, ix->stream));
  if (!rc && sketches && mem == NIQKI_MEM_HOST) NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  a.used_set = true;
  ix->ahead_head ^= 1u;
  ix->ahead_n -= 1;
  return rc;
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

}  // extern "C"
