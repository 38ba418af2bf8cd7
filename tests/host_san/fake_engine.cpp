// tests/host_san/fake_engine.cpp -- TEST INFRASTRUCTURE ONLY: the part of the C ABI (include/niqki_hip.h) that the
// `niqki` host program calls, answered on the CPU by the parity oracle (oracle/niqki_oracle.c), so that the host
// program -- reader threads, the two-deep batch pipeline of Index::for_each_batch, the lines-mode reader / writer
// threads, the multi-member gzip writer and reader of gzio.h, the option parser, the packer -- can be compiled
// WITHOUT HIP and run under ThreadSanitizer and AddressSanitizer + UBSan on a machine without a GPU
// (tests/test_host_sanitizers.py; never on the GPU box).  Its outputs are compared with the reference CLI's goldens,
// so the host logic is checked here too.  This is not a CPU path of the product: nothing under niqki_amd/ or
// include/ knows of it, libniqki_hip.so has no such fallback, and the binaries land in tests/host_san/bin/.
// One handle = one whole index; groups (--gpus > 1) are refused.
//
// Record framing restates the reference's reader loop (Index::Biogetline, src/niqki_index.cpp:890-941, as the GPU's
// nq_ingest.hip does): FASTA -- line 0 of a file and every line whose first byte is '>' (or 0xFF) is a header and
// starts a record, all other lines without their '\n' are its sequence; FASTQ -- line 4r is a header, line 4r + 1
// the sequence.  Whole-file mode: one sketch per file over all its records longer than K, densified once.
#include "../../include/niqki_hip.h"
#include "../../niqki_amd/csrc/nq_pack.h"
#include "../../oracle/niqki_oracle.h"

#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct niqki_index {
  niqki_params p{};
  nqo_params op{};
  uint32_t F = 0, R = 0;
  std::vector<int32_t> sk;        // n x F, the inserted sketches
  uint32_t n = 0;
  nqo_index *ix = nullptr;        // built lazily from sk
  uint32_t ix_n = 0;
  std::vector<int32_t> staged;    // sketches of the staged batch
  uint32_t staged_n = 0;
  bool staged_ok = false;
  uint64_t gz_seen = 0;   // NIQKI_FILE_GZIP files shown so far
  std::vector<uint8_t> dump;      // export cache / import accumulation
  uint32_t dump_n = 0xFFFFFFFFu;
  std::vector<uint64_t> dump_slot;   // byte position of every slot's first bucket (header excluded), F + 1
  uint32_t import_n = 0;
  bool importing = false;
  std::string err;
};
struct niqki_group {};

namespace {

thread_local std::string g_err;

int fail(niqki_index *ix, int code, const std::string &m) {
  if (ix) ix->err = m; else g_err = m;
  return code;
}

void drop_index(niqki_index *ix) {
  if (ix->ix) nqo_index_free(ix->ix);
  ix->ix = nullptr;
}

const nqo_index *index_of(niqki_index *ix) {
  if (!ix->ix || ix->ix_n != ix->n) {
    drop_index(ix);
    ix->ix = nqo_index_build(&ix->op, ix->sk.data(), ix->n);
    ix->ix_n = ix->n;
  }
  return ix->ix;
}

struct Rec { uint64_t hdr; std::vector<uint8_t> seq; };

// the records of one file (all of them, short ones included)
void frame(const uint8_t *d, uint64_t n, char type, std::vector<Rec> &out) {
  uint64_t pos = 0;
  uint64_t line = 0;
  while (pos < n) {
    const uint8_t *nl = (const uint8_t *)memchr(d + pos, '\n', n - pos);
    const uint64_t end = nl ? (uint64_t)(nl - d) : n;
    if (type == 'Q') {
      if ((line & 3) == 0) out.push_back(Rec{pos, {}});
      else if ((line & 3) == 1) out.back().seq.assign(d + pos, d + end);
    } else {
      const bool hdr = line == 0 || d[pos] == '>' || d[pos] == 0xFF;
      if (hdr) out.push_back(Rec{pos, {}});
      else out.back().seq.insert(out.back().seq.end(), d + pos, d + end);
    }
    ++line;
    pos = nl ? end + 1 : n;
  }
}

void sketch_records(const niqki_index *ix, const std::vector<Rec> &recs, size_t a, size_t b, int32_t *sk) {
  std::fill(sk, sk + ix->F, -1);
  for (size_t r = a; r < b; ++r)
    if (recs[r].seq.size() > ix->p.K) nqo_sketch_accumulate(&ix->op, recs[r].seq.data(), recs[r].seq.size(), sk);
  nqo_densify(&ix->op, sk);
}

int hits_of(niqki_index *ix, const int32_t *sk, uint32_t nq, uint64_t *off, uint32_t *hc, uint32_t *hg, uint64_t cap) {
  const nqo_index *x = index_of(ix);
  std::vector<uint32_t> counts(std::max<uint32_t>(ix->n, 1)), c(std::max<uint32_t>(ix->n, 1)), g(std::max<uint32_t>(ix->n, 1));
  uint64_t at = 0;
  off[0] = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    uint32_t k = 0;
    if (ix->n) {
      nqo_query_counts(x, sk + (size_t)q * ix->F, counts.data());
      k = nqo_hits_from_counts(counts.data(), ix->n, ix->op.min_score, c.data(), g.data(), ix->n);
    }
    for (uint32_t j = 0; j < k; ++j)
      if (at + j < cap) { hc[at + j] = c[j]; hg[at + j] = g[j]; }
    at += k;
    off[q + 1] = at;
  }
  return at > cap ? NIQKI_E_CAPACITY : NIQKI_OK;
}

int make_dump(niqki_index *ix) {
  if (ix->dump_n == ix->n && !ix->dump.empty()) return NIQKI_OK;
  const nqo_index *x = index_of(ix);
  const uint64_t size = nqo_dump_bytes(x, nullptr, 0);
  ix->dump.resize(size);
  nqo_dump_bytes(x, ix->dump.data(), size);
  ix->dump_slot.assign((size_t)ix->F + 1, 0);
  uint64_t w = 24;
  for (uint32_t s = 0; s < ix->F; ++s) {
    ix->dump_slot[s] = w - 24;
    for (uint32_t fp = 0; fp < ix->R; ++fp) {
      uint32_t sz;
      memcpy(&sz, ix->dump.data() + w, 4);
      w += 4 + (uint64_t)sz * 4;
    }
  }
  ix->dump_slot[ix->F] = w - 24;
  ix->dump_n = ix->n;
  return NIQKI_OK;
}

}  // namespace

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
    case NIQKI_E_GZIP: return "gzip file not taken by the device inflate";
    default: return "unknown status";
  }
}

uint32_t niqki_min_score(double min_fract, uint32_t S) { return nqo_min_score(min_fract, S); }

int niqki_create(const niqki_params *p, niqki_index **out) {
  if (!p || !out) return NIQKI_E_INVALID;
  if (p->K < 1 || p->K > 31 || p->S < 1 || p->S > 16 || p->W < 1 || p->W > 15 || p->H > p->W || p->S + p->W > 30)
    return fail(nullptr, NIQKI_E_INVALID, "unsupported parameters");
  if ((p->slot_begin || p->slot_end) && !(p->slot_begin == 0 && p->slot_end == (1u << p->S)))
    return fail(nullptr, NIQKI_E_INVALID, "fake engine: whole-range handles only");
  niqki_index *ix = new niqki_index();
  ix->p = *p;
  ix->op = nqo_params{p->K, p->S, p->W, p->H, p->min_score, 0};
  ix->F = 1u << p->S;
  ix->R = 1u << p->W;
  *out = ix;
  return NIQKI_OK;
}

void niqki_destroy(niqki_index *ix) {
  if (!ix) return;
  drop_index(ix);
  delete ix;
}

const char *niqki_last_error(const niqki_index *ix) { return ix ? ix->err.c_str() : g_err.c_str(); }

int niqki_get_params(const niqki_index *ix, niqki_params *out) {
  if (!ix || !out) return NIQKI_E_INVALID;
  *out = ix->p;
  out->slot_begin = 0;
  out->slot_end = ix->F;
  return NIQKI_OK;
}

int niqki_select_best_H(niqki_index *ix, double genome_size, uint32_t *H_out) {
  if (!ix) return NIQKI_E_INVALID;
  const uint32_t h0 = ix->op.H0p1 ? ix->op.H0p1 - 1 : ix->op.H;
  const uint32_t H = nqo_select_best_H(genome_size, ix->op.S, ix->op.W, ix->op.H);
  if (H > ix->op.W) return fail(ix, NIQKI_E_INVALID, "select_best_H chose H > W");
  ix->op.H0p1 = h0 + 1;
  ix->op.H = H;
  ix->p.H = H;
  if (H_out) *H_out = H;
  return NIQKI_OK;
}

uint32_t niqki_genome_count(const niqki_index *ix) { return ix ? (ix->importing ? ix->import_n : ix->n) : 0; }

int niqki_sketch(niqki_index *ix, const uint8_t *seqs, const uint64_t *rec_off, uint32_t n_rec, const uint32_t *entry_rec,
                 uint32_t n_entry, int32_t *sketches, int mem) {
  if (!ix || !rec_off || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  for (uint32_t e = 0; e < n_entry; ++e) {
    int32_t *sk = sketches + (size_t)e * ix->F;
    std::fill(sk, sk + ix->F, -1);
    const uint32_t a = entry_rec ? entry_rec[e] : e, b = entry_rec ? entry_rec[e + 1] : e + 1;
    for (uint32_t r = a; r < b && r < n_rec; ++r)
      if (rec_off[r + 1] - rec_off[r] > ix->p.K) nqo_sketch_accumulate(&ix->op, seqs + rec_off[r], rec_off[r + 1] - rec_off[r], sk);
    nqo_densify(&ix->op, sk);
  }
  return NIQKI_OK;
}

int niqki_insert(niqki_index *ix, const int32_t *sketches, uint32_t n, int mem) {
  if (!ix || (!sketches && n) || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  ix->sk.insert(ix->sk.end(), sketches, sketches + (size_t)n * ix->F);
  ix->n += n;
  return NIQKI_OK;
}

int niqki_query(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids,
                uint64_t capacity, int mem) {
  if (!ix || !hit_off || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  return hits_of(ix, sketches, nq, hit_off, hit_counts, hit_gids, capacity);
}

int niqki_query_counts(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint16_t *counts, uint64_t stride, int mem) {
  if (!ix || mem != NIQKI_MEM_HOST || stride < ix->n) return NIQKI_E_INVALID;
  const nqo_index *x = index_of(ix);
  std::vector<uint32_t> c(std::max<uint32_t>(ix->n, 1));
  for (uint32_t q = 0; q < nq; ++q) {
    nqo_query_counts(x, sketches + (size_t)q * ix->F, c.data());
    for (uint32_t g = 0; g < ix->n; ++g) counts[(size_t)q * stride + g] = (uint16_t)c[g];
  }
  return NIQKI_OK;
}

int niqki_hits_from_counts(niqki_index *ix, const uint16_t *counts, uint32_t nq, uint64_t stride, uint32_t gid_begin, uint32_t n_gids,
                           uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  std::vector<uint32_t> c(std::max<uint32_t>(n_gids, 1)), hc(std::max<uint32_t>(n_gids, 1)), hg(std::max<uint32_t>(n_gids, 1));
  uint64_t at = 0;
  hit_off[0] = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    for (uint32_t g = 0; g < n_gids; ++g) c[g] = counts[(size_t)q * stride + gid_begin + g];
    const uint32_t k = nqo_hits_from_counts(c.data(), n_gids, ix->op.min_score, hc.data(), hg.data(), n_gids);
    for (uint32_t j = 0; j < k; ++j)
      if (at + j < capacity) { hit_counts[at + j] = hc[j]; hit_gids[at + j] = gid_begin + hg[j]; }
    at += k;
    hit_off[q + 1] = at;
  }
  return at > capacity ? NIQKI_E_CAPACITY : NIQKI_OK;
}

int niqki_stage_raw_prefetch(niqki_index *ix, const niqki_raw_batch *b) { return (ix && b) ? NIQKI_OK : NIQKI_E_INVALID; }

int niqki_stage_raw(niqki_index *ix, const niqki_raw_batch *b, int mem, niqki_stage_info *info, uint64_t *entry_hdr) {
  if (!ix || !b || !info || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  ix->staged_ok = false;
  *info = niqki_stage_info{0, 0, 0, 0};
  if (b->lines && b->n_files > 1) return fail(ix, NIQKI_E_INVALID, "lines mode frames one file per call");
  std::vector<uint8_t> unpacked;
  // NIQKI_FILE_GZIP files: this stand-in refuses every third one it is shown (so that the host program's way back
  // through zlib runs under the sanitizers too) and inflates the others with zlib, as strict about them as the device
  bool refused = false;
  for (uint32_t f = 0; f < b->n_files; ++f) {
    if (b->file_status) b->file_status[f] = 0;
    if ((b->file_type[f] & NIQKI_FILE_GZIP) && (!b->file_ptr || b->lines)) return fail(ix, NIQKI_E_INVALID, "gzip files: file_ptr form, whole mode");
    if ((b->file_type[f] & NIQKI_FILE_GZIP) && (ix->gz_seen++ % 3) == 2) {
      if (b->file_status) b->file_status[f] = 5;
      refused = true;
    }
  }
  if (refused) return fail(ix, NIQKI_E_GZIP, "gzip files refused (file_status)");
  std::vector<Rec> recs;
  std::vector<size_t> file_first(b->n_files + 1, 0);
  uint64_t raw_total = 0;
  std::vector<uint64_t> file_raw_off(b->n_files + 1, 0);
  for (uint32_t f = 0; f < b->n_files; ++f) {
    const uint64_t wire = b->file_off[f + 1] - b->file_off[f];
    const uint8_t *d = b->file_ptr ? b->file_ptr[f] : b->raw + b->file_off[f];
    uint64_t n = wire;
    char type = (char)b->file_type[f];
    if (type == 'a') {
      if (!b->file_ptr || b->lines || !nqp::valid(d, wire)) return fail(ix, NIQKI_E_INVALID, "not a well-formed packed container");
      nqp::PackHeader h;
      memcpy(&h, d, sizeof h);
      unpacked.resize(h.raw_len);
      if (!nqp::unpack(d, wire, unpacked.data(), unpacked.size())) return fail(ix, NIQKI_E_INVALID, "unpack failed");
      d = unpacked.data();
      n = h.raw_len;
      type = 'A';
    } else if (b->file_type[f] & NIQKI_FILE_GZIP) {
      type = (char)(b->file_type[f] & ~NIQKI_FILE_GZIP);
      if (type != 'A' && type != 'Q') return fail(ix, NIQKI_E_INVALID, "bad gzip file type");
      uint64_t isize = 0;
      if (wire >= 18) isize = (uint64_t)d[wire - 4] | (uint64_t)d[wire - 3] << 8 | (uint64_t)d[wire - 2] << 16 | (uint64_t)d[wire - 1] << 24;
      unpacked.assign((size_t)isize + 1, 0);
      z_stream z{};
      bool ok = wire >= 18 && inflateInit2(&z, 15 + 16) == Z_OK;
      if (ok) {
        z.next_in = const_cast<Bytef *>(d);
        z.avail_in = (uInt)wire;
        z.next_out = unpacked.data();
        z.avail_out = (uInt)unpacked.size();
        const int r = inflate(&z, Z_FINISH);
        ok = r == Z_STREAM_END && z.avail_in == 0 && z.total_out == isize;
        inflateEnd(&z);
      }
      if (!ok) {
        if (b->file_status) b->file_status[f] = 12;
        return fail(ix, NIQKI_E_GZIP, "gzip file refused (file_status)");
      }
      d = unpacked.data();
      n = isize;
    } else if (type != 'A' && type != 'Q') {
      return fail(ix, NIQKI_E_INVALID, "file_type must be 'A', 'Q' or 'a'");
    }
    file_first[f] = recs.size();
    const size_t before = recs.size();
    frame(d, n, type, recs);
    for (size_t r = before; r < recs.size(); ++r) recs[r].hdr += raw_total;
    raw_total += n;
    file_raw_off[f + 1] = raw_total;
  }
  file_first[b->n_files] = recs.size();
  info->n_rec = (uint32_t)recs.size();
  for (auto &r : recs) info->seq_bytes += r.seq.size();
  if (!b->lines) {
    ix->staged.assign((size_t)b->n_files * ix->F, -1);
    for (uint32_t f = 0; f < b->n_files; ++f) sketch_records(ix, recs, file_first[f], file_first[f + 1], ix->staged.data() + (size_t)f * ix->F);
    ix->staged_n = b->n_files;
    info->n_entry = b->n_files;
    info->consumed = raw_total;
  } else {
    if (b->max_entries == 0) return fail(ix, NIQKI_E_INVALID, "max_entries must be > 0");
    const size_t n_use = b->final ? recs.size() : (recs.empty() ? 0 : recs.size() - 1);
    ix->staged.clear();
    uint32_t n_entry = 0;
    size_t stop = n_use;
    for (size_t r = 0; r < n_use; ++r) {
      if (recs[r].seq.size() <= ix->p.K) continue;
      if (n_entry == b->max_entries) { stop = r; break; }
      ix->staged.resize((size_t)(n_entry + 1) * ix->F);
      sketch_records(ix, recs, r, r + 1, ix->staged.data() + (size_t)n_entry * ix->F);
      if (entry_hdr) entry_hdr[n_entry] = recs[r].hdr;
      ++n_entry;
    }
    ix->staged_n = n_entry;
    info->n_entry = n_entry;
    info->consumed = stop < recs.size() ? recs[stop].hdr : raw_total;
  }
  ix->staged_ok = true;
  return NIQKI_OK;
}

int niqki_staged_insert(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  if (!ix->staged_ok) return fail(ix, NIQKI_E_STATE, "no staged batch");
  return niqki_insert(ix, ix->staged.data(), ix->staged_n, NIQKI_MEM_HOST);
}

int niqki_staged_query(niqki_index *ix, uint64_t *hit_off, uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix || !hit_off || mem != NIQKI_MEM_HOST) return NIQKI_E_INVALID;
  if (!ix->staged_ok) return fail(ix, NIQKI_E_STATE, "no staged batch");
  return hits_of(ix, ix->staged.data(), ix->staged_n, hit_off, hit_counts, hit_gids, capacity);
}

void *niqki_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void niqki_host_free(void *p) { std::free(p); }

int niqki_matrix_range(niqki_index *ix, uint32_t begin, uint32_t end, uint16_t *counts, uint64_t stride, int mem) {
  if (!ix || begin > end || end > ix->n || mem != NIQKI_MEM_HOST || stride < ix->n) return NIQKI_E_INVALID;
  const nqo_index *x = index_of(ix);
  const uint32_t nb = end - begin;
  std::vector<uint16_t> m((size_t)std::max<uint32_t>(ix->n, 1) * std::max<uint32_t>(nb, 1));
  if (nb && ix->n) nqo_matrix_range(x, begin, end, m.data());
  for (uint32_t t = 0; t < nb; ++t)
    for (uint32_t a = 0; a < ix->n; ++a) counts[(size_t)t * stride + a] = m[(size_t)a * nb + t];
  return NIQKI_OK;
}

int niqki_export_dump_header(niqki_index *ix, uint8_t header[24]) {
  if (!ix || !header) return NIQKI_E_INVALID;
  uint32_t hdr[6] = {ix->op.S, ix->op.K, ix->op.H, ix->op.W, ix->op.min_score, ix->n};
  memcpy(header, hdr, 24);
  return NIQKI_OK;
}

int niqki_export_dump_layout(niqki_index *ix, uint64_t *slot_bytes) {
  if (!ix || !slot_bytes) return NIQKI_E_INVALID;
  make_dump(ix);
  memcpy(slot_bytes, ix->dump_slot.data(), ((size_t)ix->F + 1) * 8);
  return NIQKI_OK;
}

int niqki_export_dump_slots(niqki_index *ix, uint32_t s0, uint32_t s1, uint8_t *buf, uint64_t capacity, uint64_t *size) {
  if (!ix || !size || s0 > s1 || s1 > ix->F) return NIQKI_E_INVALID;
  make_dump(ix);
  *size = ix->dump_slot[s1] - ix->dump_slot[s0];
  if (!buf) return NIQKI_OK;
  if (capacity < *size) return NIQKI_E_CAPACITY;
  memcpy(buf, ix->dump.data() + 24 + ix->dump_slot[s0], *size);
  return NIQKI_OK;
}

int niqki_import_begin(const niqki_params *params, const uint8_t header[24], niqki_index **out) {
  if (!params || !header || !out) return NIQKI_E_INVALID;
  uint32_t hdr[6];
  memcpy(hdr, header, 24);
  niqki_params p = *params;
  p.S = hdr[0]; p.K = hdr[1]; p.H = hdr[2]; p.W = hdr[3]; p.min_score = hdr[4];
  niqki_index *ix = nullptr;
  const int rc = niqki_create(&p, &ix);
  if (rc) return rc;
  ix->importing = true;
  ix->import_n = hdr[5];
  ix->dump.assign(header, header + 24);
  *out = ix;
  return NIQKI_OK;
}

int niqki_import_slots(niqki_index *ix, uint32_t s0, uint32_t s1, const uint8_t *buf, uint64_t len, uint64_t *consumed) {
  if (!ix || !ix->importing || !buf || s0 > s1 || s1 > ix->F) return NIQKI_E_INVALID;
  uint64_t w = 0;
  for (uint32_t s = s0; s < s1; ++s)
    for (uint32_t fp = 0; fp < ix->R; ++fp) {
      if (w + 4 > len) return fail(ix, NIQKI_E_INVALID, "dump payload ends inside a slot");
      uint32_t sz;
      memcpy(&sz, buf + w, 4);
      w += 4 + (uint64_t)sz * 4;
    }
  if (w > len) return fail(ix, NIQKI_E_INVALID, "dump payload ends inside a bucket");
  if (consumed) *consumed = w;
  ix->dump.insert(ix->dump.end(), buf, buf + w);
  if (s1 == ix->F) {   // the whole stream: the sketches back from the buckets (bucket fp + slot * R holds the genomes whose slot cell is fp)
    ix->sk.assign((size_t)ix->import_n * ix->F, -1);
    uint64_t at = 24;
    for (uint32_t s = 0; s < ix->F; ++s)
      for (uint32_t fp = 0; fp < ix->R; ++fp) {
        uint32_t sz;
        memcpy(&sz, ix->dump.data() + at, 4);
        at += 4;
        for (uint32_t k = 0; k < sz; ++k, at += 4) {
          uint32_t g;
          memcpy(&g, ix->dump.data() + at, 4);
          if (g >= ix->import_n) return fail(ix, NIQKI_E_INVALID, "dump holds genome ids >= genome count");
          ix->sk[(size_t)g * ix->F + s] = (int32_t)fp;
        }
      }
    ix->n = ix->import_n;
    ix->importing = false;
    ix->dump.clear();
    ix->dump_n = 0xFFFFFFFFu;
  }
  return NIQKI_OK;
}

// ---- groups: one whole-range handle only ----
void niqki_group_slot_range(uint32_t rank, uint32_t world, uint32_t S, uint32_t *b, uint32_t *e) {
  const uint64_t F = 1ull << S;
  if (b) *b = (uint32_t)(F * rank / world);
  if (e) *e = (uint32_t)(F * (rank + 1) / world);
}
int niqki_group_create(niqki_index *const *shards, uint32_t, uint32_t, uint32_t, const uint8_t *, niqki_group **) {
  if (shards && shards[0]) shards[0]->err = "the fake engine of the sanitizer build has no groups (--gpus 1 only)";
  return NIQKI_E_STATE;
}
void niqki_group_destroy(niqki_group *) {}
const char *niqki_group_last_error(const niqki_group *) { return "the fake engine has no groups"; }
int niqki_group_staged_insert(niqki_group *, uint32_t, const uint32_t *) { return NIQKI_E_STATE; }
int niqki_group_staged_query(niqki_group *, uint32_t, const uint32_t *, uint64_t *const *, uint32_t *const *, uint32_t *const *, uint64_t, int) {
  return NIQKI_E_STATE;
}

}  // extern "C"
