// nq_api_build.hip -- sketching, the sketch store and the inverted index behind the C ABI: niqki_sketch / _densify /
// _insert / _build / _get_sketches (Index::compute_sketch, sketch_densification, insert_sketch,
// src/niqki_index.cpp:313-370) and the internal build: store growth, index segments (main + delta), tile shapes.
#include "nq_handle.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace nqi {

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
  a.redo = nullptr;
  a.redo_only = a.redo_mark = 0;
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
  } else if (const uint32_t list = nq::sketch_read_list(ix->d, avg)) {
    // Short records: the one-wavefront kernel, with its 192-entry list (nine sketches per CU) where the average record
    // has at most 200 bases.  A record with more occupied cells than the list holds -- far longer than the batch's
    // average -- is flagged instead of taking the plain pass over all cells (fifty reads' time) and sketched by the
    // next launch: the 384-entry list, then the workgroup kernel; a launch's other workgroups leave at once.
    nqi::Buf &flags = ix->ws_redo[ix->sk_stream && ix->stream == ix->sk_stream ? 1 : 0];
    int rc = ensure(ix, flags, (size_t)n_entry * 4);
    if (rc) return rc;
    NQ_HIP(ix, hipMemsetAsync(flags.p, 0, (size_t)n_entry * 4, ix->stream));
    Span sp(ix, NIQKI_KC_SKETCH);
    a.redo = (uint32_t *)flags.p;
    if (list < 384) {
      a.redo_only = 0; a.redo_mark = 1;
      NQ_HIP(ix, nq::launch_sketch(a, n_entry, avg, ix->stream));
      a.redo_only = 1; a.redo_mark = 2;
      NQ_HIP(ix, nq::launch_sketch(a, n_entry, 400, ix->stream));   // (an average that takes the long list)
    } else {
      a.redo_only = 0; a.redo_mark = 2;
      NQ_HIP(ix, nq::launch_sketch(a, n_entry, avg, ix->stream));
    }
    a.redo_only = 2; a.redo_mark = 0;
    NQ_HIP(ix, nq::launch_sketch(a, n_entry, nq::kSketchWorkgroupLen, ix->stream));
  } else {
    Span sp(ix, NIQKI_KC_SKETCH);
    NQ_HIP(ix, nq::launch_sketch(a, n_entry, avg, ix->stream));
  }
  (void)n_rec;
  return NIQKI_OK;
}

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

// the index of store columns [g_base, g_base + N) into the current segment's buffers
int build_range(niqki_index *ix, uint32_t g_base, uint32_t N) {
  const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
  uint32_t tile = ix->p.tile_genomes;
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

using namespace nqi;

extern "C" {

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
  a.redo = nullptr;
  a.redo_only = a.redo_mark = 0;
  {
    Span sp(ix, NIQKI_KC_DENSIFY);
    // (an average that picks the one-wavefront kernel with its long entry list)
    NQ_HIP(ix, nq::launch_sketch(a, n, ix->d.F <= 4096 ? 256 : ((uint64_t)1 << 22), ix->stream));
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

}  // extern "C"
