// nq_api_dump.hip -- dump_index_disk / the loading constructor behind the C ABI (src/niqki_index.cpp:42-102): the
// reference's bucket stream exported from / imported into the sketch store, whole or by groups of slots, resident
// or paged, whole-range or one slot shard.
#include "nq_handle.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

using namespace nqi;

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

extern "C" {

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
    nqi::create_error() = rc ? ix->err : std::string(hipGetErrorString(e));
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
        if (p + 4 > len) { nqi::create_error() = "dump payload is truncated"; niqki_destroy(ix); return NIQKI_E_INVALID; }
        uint32_t sz;
        std::memcpy(&sz, buf + p, 4);
        p += 4 + (uint64_t)sz * 4;
      }
      ++s1;
    }
    if (p > len) { nqi::create_error() = "dump payload is truncated"; niqki_destroy(ix); return NIQKI_E_INVALID; }
    uint64_t used = 0;
    rc = niqki_import_slots(ix, s0, s1, buf + pos, p - pos, &used);
    if (rc) { nqi::create_error() = ix->err; niqki_destroy(ix); return rc; }
    pos += used;
    s0 = s1;
  }
  if (consumed) *consumed = pos;
  *out = ix;
  return NIQKI_OK;
}

}  // extern "C"
