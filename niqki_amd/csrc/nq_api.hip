// nq_api.hip -- the C ABI of libniqki_hip.so (include/niqki_hip.h): the handle -- creation, options, statistics,
// per-class timers, device scratch -- and the helpers every other part of the boundary uses.  No compute happens on
// the host here and there is no CPU fallback: without a gfx950 device niqki_create fails.
// The rest of the boundary by concern:  nq_api_build.hip  sketch, insert, index build, the sketch store
//                                       nq_api_query.hip  counters, hits, matrix, paged walks
//                                       nq_api_stage.hip  raw file bytes in: framing, packed FASTA, staged batches
//                                       nq_api_dump.hip   dump export / import
//                                       nq_api_bench.hip  synthetic inputs and ALU / copy probes (niqki_hip_bench.h)
//                                       nq_shared.hip     many host threads on one handle;  nq_group.hip  slot-range shards
#include "nq_handle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>

#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <new>
#include <string>
#include <vector>

namespace nqi {

std::string &create_error() {   // why the last niqki_create / niqki_import_* on this thread failed
  thread_local std::string err;
  return err;
}

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
  if (ix->sk_stream) NQ_HIP(ix, hipStreamSynchronize(ix->sk_stream));   // (spans of niqki_sketch_ahead lie on the sketch lane)
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

// a whole-range S = 16 handle counts in two planes of <= 2^15 slots each (nq_kernels.h, kPassSlots)
// more than 2^15 slots on the handle (whole-range S = 16): counts reach 2^16, the slots are walked in two halves
// into two counter planes (a paged handle: its pages never straddle the halves)
bool two_planes(const niqki_index *ix) {
  return (ix->resident_bytes ? ix->full_end - ix->full_begin : ix->d.slot_end - ix->d.slot_begin) > nq::kPassSlots;
}

// first slot of the handle in a whole sketch row (while a page is resident d.slot_begin is the page's)
uint32_t first_slot(const niqki_index *ix) { return ix->resident_bytes ? ix->full_begin : ix->d.slot_begin; }

}  // namespace nqi

using namespace nqi;


// ---- page-locked host memory (niqki_host_alloc) ----------------------------------------------------------------
// hipHostMalloc page-locks at ~5 GB/s on this platform whatever the size or the number of calling threads
// (profiles/r05_ubench_pin.txt: the time goes into faulting 4 KB pages in one by one), and a run's reader buffers are
// gigabytes: its first phase waited for them.  Buffers of 256 KB and more are therefore cut from 256 MB slabs of
// anonymous memory that asks for transparent huge pages (madvise) and is registered with the runtime as a whole
// (hipHostRegister: 25 GB/s where huge pages are to be had, the old rate where not); freed pieces are kept by size
// and handed out again, slabs are never returned (a process has a few, for its lifetime).  Anything that fails falls
// back to hipHostMalloc.
namespace nqi {
namespace {
constexpr size_t kSlabBytes = size_t(256) << 20, kHuge = size_t(2) << 20, kPieceGran = size_t(64) << 10, kSlabMin = size_t(256) << 10;
struct Slab {
  uint8_t *base;
  size_t size, used;
};
std::mutex g_host_mu;
std::vector<Slab> g_slabs;
std::multimap<size_t, void *> g_free_pieces;      // size -> piece
std::unordered_map<void *, size_t> g_piece_size;  // every piece ever cut: its size

uint8_t *new_slab(size_t size) {
  void *raw = mmap(nullptr, size + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == MAP_FAILED) return nullptr;
  uint8_t *base = (uint8_t *)(((uintptr_t)raw + kHuge - 1) & ~(uintptr_t)(kHuge - 1));
  (void)madvise(base, size, MADV_HUGEPAGE);
  if (hipHostRegister(base, size, hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    munmap(raw, size + kHuge);
    return nullptr;
  }
  return base;
}
}  // namespace

// Piece sizes come in classes (2^k and 1.5 x 2^k, in units of kPieceGran): a reader's buffers grow with their files,
// and with exact sizes a freed piece was all but never asked for again -- page-locked memory that is registered once and
// never returned piled up to several times the live buffers on a long list.
static size_t piece_class(size_t bytes) {
  const size_t n = (bytes + kPieceGran - 1) & ~(kPieceGran - 1);
  size_t p = kPieceGran;
  while (p * 2 <= n) p *= 2;
  if (n == p) return p;
  const size_t mid = (p + p / 2 + kPieceGran - 1) & ~(kPieceGran - 1);
  return n <= mid ? mid : 2 * p;
}

void *host_alloc(size_t bytes) {
  if (bytes >= kSlabMin) {
    const size_t n = piece_class(bytes);
    std::lock_guard<std::mutex> g(g_host_mu);
    // the smallest free piece that holds the request, if it is not more than twice its class
    auto it = g_free_pieces.lower_bound(n);
    if (it != g_free_pieces.end() && it->first <= 2 * n) {
      void *p = it->second;
      g_free_pieces.erase(it);
      return p;
    }
    if (g_slabs.empty() || g_slabs.back().size - g_slabs.back().used < n) {
      if (!g_slabs.empty()) {   // what is left of the slab before becomes a free piece instead of being abandoned
        Slab &old = g_slabs.back();
        const size_t tail = (old.size - old.used) & ~(kPieceGran - 1);
        if (tail >= kSlabMin) {
          void *tp = old.base + old.used;
          old.used += tail;
          g_piece_size[tp] = tail;
          g_free_pieces.emplace(tail, tp);
        }
      }
      // (32 MB, 64, 128, then 256 MB slabs: a small run does not lock a quarter of a gigabyte)
      const size_t size = std::max(std::min(kSlabBytes, (size_t(32) << 20) << std::min<size_t>(g_slabs.size(), 3)), (n + kHuge - 1) & ~(kHuge - 1));
      uint8_t *base = new_slab(size);
      if (base) g_slabs.push_back(Slab{base, size, 0});
    }
    if (!g_slabs.empty() && g_slabs.back().size - g_slabs.back().used >= n) {
      Slab &sl = g_slabs.back();
      void *p = sl.base + sl.used;
      sl.used += n;
      g_piece_size[p] = n;
      return p;
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void host_free(void *p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> g(g_host_mu);
    auto it = g_piece_size.find(p);
    if (it != g_piece_size.end()) {
      g_free_pieces.emplace(it->second, p);
      return;
    }
  }
  (void)hipHostFree(p);
}
}  // namespace nqi

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
  auto bail = [&](int code, const std::string &why) { nqi::create_error() = why; delete ix; return code; };
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
  if (const char *v = std::getenv("NIQKI_LOOKUP_PREPASS")) ix->lookup_prepass = std::min(std::max(std::atoi(v), -1), 1);   // (niqki_hip.h: the environment table)
  *out = ix;
  return NIQKI_OK;
}

void niqki_destroy(niqki_index *ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->sk_stream) (void)hipStreamSynchronize(ix->sk_stream);
  (void)hipStreamSynchronize(ix->stream);
  for (Buf *b : {&ix->ws_seq, &ix->ws_recoff, &ix->ws_entry, &ix->ws_sk, &ix->ws_counts, &ix->ws_blk,
                 &ix->ws_hitoff, &ix->ws_hc, &ix->ws_hg, &ix->ws_tc, &ix->ws_tg, &ix->ws_misc, &ix->ws_stash,
                 &ix->ws_raw, &ix->ws_wire[0], &ix->ws_wire[1], &ix->ws_redo[0], &ix->ws_redo[1], &ix->ws_fmeta, &ix->ws_summ, &ix->ws_chunk, &ix->ws_fkept, &ix->ws_fnrec,
                 &ix->ws_hdrpos, &ix->ws_ehdr, &ix->ws_stsk, &ix->ws_order, &ix->ws_pre, &ix->ws_hl, &ix->ws_useg, &ix->ws_ijob, &ix->ws_xtab})
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
  if (ix->sk_stream) { (void)hipStreamSynchronize(ix->sk_stream); (void)hipStreamDestroy(ix->sk_stream); }
  for (auto &a : ix->ahead) {
    if (a.sk.p) (void)hipFree(a.sk.p);
    if (a.done) (void)hipEventDestroy(a.done);
    if (a.used) (void)hipEventDestroy(a.used);
  }
  for (auto &pr : ix->pre) if (pr.ev) (void)hipEventDestroy(pr.ev);
  if (ix->ev_fork) (void)hipEventDestroy(ix->ev_fork);
  if (ix->ev_join) (void)hipEventDestroy(ix->ev_join);
  if (ix->own_stream) (void)hipStreamDestroy(ix->stream);
  nqi::shared_free(ix);
  delete ix;
}

const char *niqki_last_error(const niqki_index *ix) { return ix ? ix->err.c_str() : nqi::create_error().c_str(); }

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
  if (ix->sk_stream) NQ_HIP(ix, hipStreamSynchronize(ix->sk_stream));
  NQ_HIP(ix, hipStreamSynchronize(ix->stream));
  if (ix->own_stream) { (void)hipStreamDestroy(ix->stream); ix->own_stream = false; }
  ix->stream = (hipStream_t)s;
  return NIQKI_OK;
}

void *niqki_get_stream(const niqki_index *ix) { return ix ? (void *)ix->stream : nullptr; }

int niqki_synchronize(niqki_index *ix) {
  if (!ix) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (ix->sk_stream) NQ_HIP(ix, hipStreamSynchronize(ix->sk_stream));   // sketches made ahead (niqki_sketch_ahead)
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
  if (!std::strcmp(key, "inflate_window")) {
    if (value < -1 || value > 1) return fail(ix, NIQKI_E_INVALID, "inflate_window: -1 = by the number of files, 0 = whole in LDS, 1 = its last 8 KB in LDS");
    ix->inflate_window = (int)value;
    return NIQKI_OK;
  }
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
  if (!std::strcmp(key, "sketch_lane_cus")) {
    // the lane of niqki_sketch_ahead limited to the TOP `value` compute units of the device's numbering (0 = all); a
    // caller that queries on a stream limited to the others (hipExtStreamCreateWithCUMask) splits the device in two
    NQ_HIP(ix, hipSetDevice(ix->device));
    int cus = 0;
    NQ_HIP(ix, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ix->device));
    if (value < 0 || value > cus) return fail(ix, NIQKI_E_INVALID, "sketch_lane_cus must be 0 (all) .. the device's compute units");
    if (ix->sk_stream) {
      NQ_HIP(ix, hipStreamSynchronize(ix->sk_stream));
      (void)hipStreamDestroy(ix->sk_stream);
      ix->sk_stream = nullptr;
    }
    ix->sk_lane_cus = (uint32_t)value;
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
  if (!std::strcmp(key, "inflate_files_in_flight") || !std::strcmp(key, "inflate_files_in_flight_8k")) {
    // how many files a launch of the device inflate runs at once (workgroups the device keeps resident), by kernel form
    (void)hipSetDevice(ix->device);
    *value = nq::inflate_resident_files(key[23] != 0);
    return NIQKI_OK;
  }
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

void *niqki_host_alloc(size_t bytes) { return nqi::host_alloc(bytes); }

void niqki_host_free(void *p) { nqi::host_free(p); }

}  // extern "C"
