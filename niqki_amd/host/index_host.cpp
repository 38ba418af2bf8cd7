// index_host.cpp -- file-level drivers of the `niqki` host program.  Moves the
// bytes of the input files into page-locked memory and calls the gfx950 engine
// through the C ABI (include/niqki_hip.h): record framing (the reference's
// Biogetline), sketching, counting and sorting all happen on the GPU.
#include "index_host.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <future>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../csrc/nq_pack.h"
#include "file_reader.h"
#include "seqio.h"

namespace nqhost {

namespace {
bool exists_test(const std::string &name) {  // src/niqki_index.h:150-153
  struct stat st;
  return stat(name.c_str(), &st) == 0;
}
// default ostream << double: 6 significant digits, %g style (src/niqki_index.cpp:550,:758)
std::string fmt_double(double v) {
  char buf[64];
  snprintf(buf, sizeof buf, "%g", v);
  return buf;
}
constexpr size_t kBatchBytes = size_t(1) << 30;   // dump import: bytes per GPU call

constexpr size_t kWholeBatchFiles = 64;                // files per GPU call (whole-file mode)
constexpr size_t kWholeBatchBytes = size_t(3) << 29;   // ... or 1.5 GB
constexpr size_t kReaderBufs = 2 * kWholeBatchFiles + 32;
// Gzip files inflated on the device: a file is one wavefront's serial job there and the device runs up to 2048 of them
// at once (eight per CU with the window's last 8 KB in LDS; 1024 with all of it), so a batch takes as long as its
// longest file whatever it holds -- batches of up to 2048 files (their bytes are a third of the plain ones'), 4 GB
// of file bytes or 12 GB of inflated ones.
constexpr size_t kGzBatchFiles = 2048;
constexpr size_t kGzBatchWireBytes = size_t(4) << 30;
constexpr size_t kGzBatchRawBytes = size_t(12) << 30;
constexpr size_t kGzReaderBufs = 2 * kGzBatchFiles + 64;
}  // namespace

// Files of one GPU call (whole-file mode): one sketch per file.
struct Index::Batch {
  std::vector<OrderedFileReader::File *> files;
  std::vector<std::string> names;
  size_t bytes = 0;       // bytes that cross PCIe
  size_t raw_bytes = 0;   // bytes the device holds once packed / gzip'd files are their own bytes again (gzip: as announced)
  size_t n_gz = 0;
};

void Index::check(int rc, const char *what) const {
  if (rc == NIQKI_OK) return;
  throw std::runtime_error(std::string(what) + ": " + niqki_status_string(rc) + " (" + niqki_last_error(h_) + ")");
}

void Index::check_group(int rc, const char *what) const {
  if (rc == NIQKI_OK) return;
  throw std::runtime_error(std::string(what) + ": " + niqki_status_string(rc) + " (" + niqki_group_last_error(grp_) + ")");
}

// One handle per GPU: with n_gpus > 1 shard r owns the slots of rank r (niqki_group_slot_range) on device
// device + r.  dump_header != nullptr: the handles are made for a streamed dump import.
void Index::make_shards(const niqki_params &p0, int device, int n_gpus, const uint8_t *dump_header) {
  if (n_gpus < 1 || n_gpus > 64) throw std::runtime_error("--gpus must be in 1..64");
  for (int r = 0; r < n_gpus; ++r) {
    niqki_params p = p0;
    p.device = n_gpus > 1 ? (device < 0 ? 0 : device) + r : device;
    // test hook: all shards on one device (a 1-GPU box emulating N; the library then moves the
    // exchanged data with device-to-device copies instead of RCCL)
    if (n_gpus > 1 && std::getenv("NIQKI_SHARDS_ON_ONE_DEVICE")) p.device = device;
    if (n_gpus > 1) niqki_group_slot_range((uint32_t)r, (uint32_t)n_gpus, dump_header ? ((const uint32_t *)dump_header)[0] : p.S, &p.slot_begin, &p.slot_end);
    niqki_index *h = nullptr;
    const int rc = dump_header ? niqki_import_begin(&p, dump_header, &h) : niqki_create(&p, &h);
    if (rc) throw std::runtime_error(std::string(dump_header ? "niqki_import_begin: " : "niqki_create: ") + niqki_status_string(rc) + " (" + niqki_last_error(nullptr) + ")");
    sh_.push_back(h);
  }
  h_ = sh_[0];
}

Index::Index(uint32_t ilF, uint32_t iK, uint32_t iW, uint32_t iH, const std::string &out_filename,
             double min_fract, int device, int n_gpus, int resident_mib) {
  niqki_params p{};
  p.K = iK; p.S = ilF; p.W = iW; p.H = iH;
  p.resident_mib = resident_mib > 0 ? (uint32_t)resident_mib : 0u;
  p.min_score = niqki_min_score(min_fract, ilF);
  make_shards(p, device, n_gpus, nullptr);
  if (n_gpus > 1) {
    const int rc = niqki_group_create(sh_.data(), (uint32_t)sh_.size(), 0, (uint32_t)sh_.size(), nullptr, &grp_);
    if (rc) throw std::runtime_error(std::string("niqki_group_create: ") + niqki_status_string(rc) + " (" + niqki_last_error(h_) + ")");
  }
  K = iK; W = iW; H = iH; lF = ilF; F = 1u << ilF; min_score = p.min_score;
  outfile.reset(new ParallelTextWriter(out_filename, host_threads()));
}

Index::Index(const std::string &dump_file, bool pretty, const std::string &out_filename, int device, int n_gpus,
             int resident_mib) {
  pretty_printing = pretty;
  // The dump is streamed: header, then the buckets in groups of whole slots (the payload of a 100k-genome index is
  // 13.6 GB), then the names.  A dump this program wrote is a file of size-tagged gzip members (gzio.h): they are
  // inflated side by side by the reader threads, a window ahead; any other gzip file (the reference's dumps are one
  // member) comes through zlib's stream.  Either way the bytes arrive in pieces, and the walk over the buckets' size
  // words -- the one serial thing about the format -- runs over memory, not over a reader call per bucket.
  std::unique_ptr<TaggedGzReader> tagged;
  std::unique_ptr<GzReader> plain;
  if (TaggedGzReader::probe(dump_file)) {
    // (a file that starts like one of ours but does not go on that way -- another gzip file appended, a member re-written
    // by some tool -- is still a gzip file: zlib's stream reads it, as the reference's zstr would)
    try { tagged.reset(new TaggedGzReader(dump_file, host_threads())); } catch (const std::exception &) { tagged.reset(); }
  }
  if (!tagged) plain.reset(new GzReader(dump_file));
  std::vector<uint8_t> buf, piece;
  size_t p = 0;   // parse position in buf
  auto more = [&]() -> bool {   // appends the stream's next piece
    if (tagged) {
      if (!tagged->next(piece)) return false;
    } else {
      piece.resize(size_t(4) << 20);
      piece.resize(plain->read(piece.data(), piece.size()));
      if (piece.empty()) return false;
    }
    buf.insert(buf.end(), piece.begin(), piece.end());
    return true;
  };
  auto need = [&](size_t n) -> bool {
    while (buf.size() - p < n)
      if (!more()) return false;
    return true;
  };
  if (!need(24)) throw std::runtime_error("'" + dump_file + "' is not a niqki dump");
  uint8_t hdr[24];
  std::memcpy(hdr, buf.data(), 24);
  p = 24;
  niqki_params prm{};
  prm.resident_mib = resident_mib > 0 ? (uint32_t)resident_mib : 0u;
  make_shards(prm, device, n_gpus, hdr);   // every shard keeps its own slots of the stream
  niqki_params q{};
  niqki_get_params(h_, &q);
  K = q.K; W = q.W; H = q.H; lF = q.S; F = 1u << q.S; min_score = q.min_score;
  const uint32_t R = 1u << W;
  uint32_t s0 = 0;
  size_t g0 = p;   // where the slots not yet handed over start in buf
  auto flush = [&](uint32_t s1) {
    uint64_t used = 0;
    for (auto *h : sh_) check(niqki_import_slots(h, s0, s1, buf.data() + g0, p - g0, &used), "niqki_import_slots");
    buf.erase(buf.begin(), buf.begin() + (ptrdiff_t)p);   // (what is left is less than one piece)
    p = g0 = 0;
    s0 = s1;
  };
  for (uint32_t s = 0; s < F; ++s) {
    for (uint32_t fp = 0; fp < R; ++fp) {
      if (!need(4)) throw std::runtime_error("'" + dump_file + "' is truncated");
      uint32_t size;
      std::memcpy(&size, buf.data() + p, 4);
      if (!need(4 + (size_t)size * 4)) throw std::runtime_error("'" + dump_file + "' is truncated");
      p += 4 + (size_t)size * 4;
    }
    if (p - g0 >= kBatchBytes / 8) flush(s + 1);
  }
  flush(F);
  // genome names, one per line after the buckets (src/niqki_index.cpp:91-95)
  while (more()) {}
  const uint32_t n = niqki_genome_count(h_);
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t *b0 = buf.data() + p, *e0 = buf.data() + buf.size();
    const uint8_t *nl = p < buf.size() ? (const uint8_t *)memchr(b0, '\n', (size_t)(e0 - b0)) : nullptr;
    const uint8_t *end = nl ? nl : e0;
    filenames.emplace_back(p < buf.size() ? std::string((const char *)b0, (size_t)(end - b0)) : std::string());
    p = nl ? (size_t)(nl + 1 - buf.data()) : buf.size();
  }
  if (n_gpus > 1) {
    const int rc = niqki_group_create(sh_.data(), (uint32_t)sh_.size(), 0, (uint32_t)sh_.size(), nullptr, &grp_);
    if (rc) throw std::runtime_error(std::string("niqki_group_create: ") + niqki_status_string(rc) + " (" + niqki_last_error(h_) + ")");
  }
  outfile.reset(new ParallelTextWriter(out_filename, host_threads()));
}

Index::~Index() {
  if (outfile) outfile->close();
  if (grp_) niqki_group_destroy(grp_);
  for (auto *h : sh_) niqki_destroy(h);
}

// src/niqki_index.cpp:126-138, including its message
void Index::select_best_H(double genome_size) {
  uint32_t chosen = H;
  for (auto *h : sh_) check(niqki_select_best_H(h, genome_size, &chosen), "select_best_H");
  H = chosen;
  std::cout << "I chosed H=" << H << std::endl;
}

void Index::compute_sketch(const std::string &reference, std::vector<int32_t> &sketch) const {
  sketch.assign(F, -1);
  uint64_t off[2] = {0, reference.size()};
  check(niqki_sketch(h_, (const uint8_t *)reference.data(), off, 1, nullptr, 1, sketch.data(), NIQKI_MEM_HOST), "niqki_sketch");
}

void Index::insert_sketch(const std::vector<int32_t> &sketch, uint32_t genome_id) {
  if (genome_id != niqki_genome_count(h_)) throw std::runtime_error("insert_sketch: ids must be consecutive");
  for (auto *h : sh_) check(niqki_insert(h, sketch.data(), 1, NIQKI_MEM_HOST), "niqki_insert");  // each shard keeps its slots
}

query_output Index::query_sketch(const std::vector<int32_t> &sketch) const {
  const uint32_t n = niqki_genome_count(h_);
  std::vector<uint32_t> hc(n ? n : 1), hg(n ? n : 1);
  uint64_t off[2] = {0, 0};
  if (grp_) {  // the shards' partial hit vectors summed on the host (a single sketch: API parity, not a fast path)
    const uint64_t stride = NIQKI_ROW_STRIDE(n);
    std::vector<uint16_t> sum(std::max<uint64_t>(stride, 2), 0), part(std::max<uint64_t>(stride, 2));
    for (auto *h : sh_) {
      check(niqki_query_counts(h, sketch.data(), 1, part.data(), stride, NIQKI_MEM_HOST), "niqki_query_counts");
      for (uint32_t i = 0; i < n; ++i) sum[i] = (uint16_t)(sum[i] + part[i]);
    }
    check(niqki_hits_from_counts(h_, sum.data(), 1, stride, 0, n, off, hc.data(), hg.data(), n, NIQKI_MEM_HOST), "niqki_hits_from_counts");
  } else
  check(niqki_query(h_, sketch.data(), 1, off, hc.data(), hg.data(), n, NIQKI_MEM_HOST), "niqki_query");
  query_output r;
  for (uint64_t i = 0; i < off[1]; ++i) r.push_back({hc[i], hg[i]});
  return r;
}

// ---- insertion / query drivers ---------------------------------------------------

// stage the files of a batch: one entry per file (insert_file_whole / query_file_whole,
// src/niqki_index.cpp:442-456, :505-519)
// multi-GPU: the batch's files are dealt to the shards in list order, `per` to each
static void rank_share(size_t n, size_t per, size_t r, size_t &lo, size_t &hi) {
  lo = std::min(n, r * per);
  hi = std::min(n, lo + per);
}

// prefetch: only start the batch's host-to-device copy (niqki_stage_raw_prefetch); the stage_batch call
// that follows for the same batch takes those bytes
void Index::stage_batch(Batch &b, bool prefetch) {
  auto stage = [&](niqki_index *h, size_t lo, size_t hi) {
    for (;;) {
      std::vector<const uint8_t *> ptr(hi - lo);
      std::vector<uint64_t> off(hi - lo + 1, 0);
      std::vector<uint8_t> type(hi - lo), status(hi - lo, 0);
      for (size_t i = lo; i < hi; ++i) {
        ptr[i - lo] = b.files[i]->buf.p;
        off[i - lo + 1] = off[i - lo] + b.files[i]->buf.size;
        type[i - lo] = b.files[i]->packed ? (uint8_t)'a'
                       : b.files[i]->gz   ? (uint8_t)(data_type(b.names[i]) | NIQKI_FILE_GZIP)
                                          : (uint8_t)data_type(b.names[i]);
      }
      niqki_raw_batch rb{};
      rb.file_ptr = ptr.data();
      rb.file_off = off.data();
      rb.file_type = type.data();
      rb.n_files = (uint32_t)(hi - lo);
      rb.file_status = status.data();
      niqki_stage_info info{};
      const int rc = prefetch ? niqki_stage_raw_prefetch(h, &rb) : niqki_stage_raw(h, &rb, NIQKI_MEM_HOST, &info, nullptr);
      if (rc == NIQKI_E_GZIP) {
        // files the device would not inflate (damaged, several members, ...): through zlib here -- which decides what
        // they yield, or throws, exactly as for a run without the device inflate -- and the batch again (every pass
        // leaves fewer gzip files, so this ends)
        size_t redone = 0;
        for (size_t i = lo; i < hi; ++i)
          if (status[i - lo] && b.files[i]->gz) {
            read_file_bytes(b.names[i], b.files[i]->buf);
            b.files[i]->gz = false;
            ++redone;
          }
        if (redone) continue;
      }
      if (rc) throw std::runtime_error(std::string("niqki_stage_raw: ") + niqki_status_string(rc) + " (" + niqki_last_error(h) + ")");
      return;
    }
  };
  if (!grp_) {
    stage(h_, 0, b.files.size());
    return;
  }
  const size_t n = b.files.size(), per = per_rank(n);
  for (size_t r = 0; r < sh_.size(); ++r) {
    size_t lo, hi;
    rank_share(n, per, r, lo, hi);
    if (lo < hi) stage(sh_[r], lo, hi);
  }
}

namespace {
struct Lap {  // adds the time since construction (or the last lap) to an accumulator
  using clk = std::chrono::steady_clock;
  clk::time_point t = clk::now();
  void to(double &acc) {
    const auto n = clk::now();
    acc += std::chrono::duration<double>(n - t).count();
    t = n;
  }
};
}  // namespace

// the staged batch of b's files: sketched and inserted
void Index::flush_insert(Batch &b) {
  Lap lap;
  if (grp_) {
    const size_t n = b.files.size(), per = per_rank(n);
    std::vector<uint32_t> n_entry(sh_.size());
    for (size_t r = 0; r < sh_.size(); ++r) {
      size_t lo, hi;
      rank_share(n, per, r, lo, hi);
      n_entry[r] = (uint32_t)(hi - lo);
    }
    check_group(niqki_group_staged_insert(grp_, (uint32_t)per, n_entry.data()), "niqki_group_staged_insert");
  } else {
    check(niqki_staged_insert(h_), "niqki_staged_insert");
  }
  for (auto &nm : b.names) filenames.push_back(nm);
  lap.to(t_dev_);
}

// hits of the entries staged on the shards, rank after rank = entry order
void Index::group_query_staged(uint32_t per, const std::vector<uint32_t> &n_entry, Hits &h) {
  const size_t G = sh_.size();
  const uint64_t N = niqki_genome_count(h_);
  uint64_t cap = std::max<uint64_t>(uint64_t(1) << 20, (uint64_t)per * 64);
  std::vector<std::vector<uint64_t>> off(G, std::vector<uint64_t>(per + 1));
  std::vector<std::vector<uint32_t>> hc(G), hg(G);
  for (;;) {
    std::vector<uint64_t *> p_off(G);
    std::vector<uint32_t *> p_hc(G), p_hg(G);
    for (size_t r = 0; r < G; ++r) {
      hc[r].resize(cap);
      hg[r].resize(cap);
      p_off[r] = off[r].data(); p_hc[r] = hc[r].data(); p_hg[r] = hg[r].data();
    }
    const int rc = niqki_group_staged_query(grp_, per, n_entry.data(), p_off.data(), p_hc.data(), p_hg.data(), cap, NIQKI_MEM_HOST);
    if (rc == NIQKI_E_CAPACITY && cap < (uint64_t)per * N) {
      for (size_t r = 0; r < G; ++r) cap = std::max(cap, off[r][per]);
      cap *= 2;
      continue;
    }
    check_group(rc, "niqki_group_staged_query");
    break;
  }
  h.off.assign(1, 0);
  h.hc.clear();
  h.hg.clear();
  for (size_t r = 0; r < G; ++r)
    for (uint32_t i = 0; i < n_entry[r]; ++i) {
      h.hc.insert(h.hc.end(), hc[r].begin() + off[r][i], hc[r].begin() + off[r][i + 1]);
      h.hg.insert(h.hg.end(), hg[r].begin() + off[r][i], hg[r].begin() + off[r][i + 1]);
      h.off.push_back(h.hc.size());
    }
}

// hits of the staged entries, written in entry order
// hits of the staged entries
void Index::query_staged(size_t n, Hits &h) {
  const uint64_t N = niqki_genome_count(h_);
  uint64_t cap = std::max<uint64_t>(uint64_t(1) << 20, n * 64);
  h.off.resize(n + 1);
  for (;;) {
    h.hc.resize(cap);
    h.hg.resize(cap);
    const int rc = niqki_staged_query(h_, h.off.data(), h.hc.data(), h.hg.data(), cap, NIQKI_MEM_HOST);
    if (rc == NIQKI_E_CAPACITY && cap < n * N) { cap = std::max(h.off[n], cap * 2); continue; }
    check(rc, "niqki_staged_query");
    break;
  }
}

// ... written in entry order
void Index::write_hits(const Hits &h) {
  if (!pretty_printing) {
    query_output one;
    for (size_t i = 0; i < h.names.size(); ++i) {
      one.clear();
      for (uint64_t j = h.off[i]; j < h.off[i + 1]; ++j) one.push_back({h.hc[j], h.hg[j]});
      output_query(one, h.names[i]);
    }
    return;
  }
  // the lines of output_query (:546-553), put together a few megabytes at a time: a batch of lines mode is 65 536
  // entries, and a call per entry -- a vector of hits, a string, a write -- was a third of the writer thread's time
  std::string &text = out_text_;
  text.clear();
  char num[64];
  for (size_t i = 0; i < h.names.size(); ++i) {
    text += h.names[i];
    text += ' ';
    for (uint64_t j = h.off[i]; j < h.off[i + 1]; ++j) {
      text += filenames[h.hg[j]];
      text += ':';
      const int n = snprintf(num, sizeof num, "%g", (double)h.hc[j] / F);
      text.append(num, (size_t)n);
      text += ' ';
    }
    text += '\n';
    if (text.size() >= (size_t(4) << 20)) { outfile->write(text); text.clear(); }
  }
  outfile->write(text);
}

// ... sketched, queried and written out
void Index::flush_query(Batch &b) {
  Lap lap;
  if (grp_) {
    const size_t n = b.files.size(), per = per_rank(n);
    std::vector<uint32_t> n_entry(sh_.size());
    for (size_t r = 0; r < sh_.size(); ++r) {
      size_t lo, hi;
      rank_share(n, per, r, lo, hi);
      n_entry[r] = (uint32_t)(hi - lo);
    }
    auto h = std::make_unique<Hits>();
    h->names = b.names;
    group_query_staged((uint32_t)per, n_entry, *h);
    lap.to(t_dev_);
    if (hits_sink_) hits_sink_(std::move(h)); else write_hits(*h);
    lap.to(t_out_);
    return;
  }
  auto h = std::make_unique<Hits>();
  h->names = b.names;
  query_staged(b.names.size(), *h);
  lap.to(t_dev_);
  if (hits_sink_) hits_sink_(std::move(h)); else write_hits(*h);
  lap.to(t_out_);
}

// The files of `paths` in batches: reader threads fill page-locked buffers ahead, the bytes of batch i+1 cross
// to the device (copy stream) while batch i is sketched, inserted or queried and its output written.
void Index::for_each_batch(const std::vector<std::string> &paths, void (Index::*flush)(Batch &)) {
  using clk = std::chrono::steady_clock;
  const bool timing = std::getenv("NIQKI_HOST_TIMING") != nullptr;  // where the wall time goes, on stderr
  double t_wait = 0, t_gpu = 0;
  t_stage_ = t_dev_ = t_out_ = 0;
  const auto t_begin = clk::now();
  // A list of mostly gzip'd files (by their names) is inflated on the device, in the large batches that wants -- if
  // it is long enough: a launch of the device inflate takes one file's time (150 ms for a 5 Mbp genome) however few
  // files it holds, in which a reader thread inflates some 30 files, so below 32 files per reader thread the readers
  // do it (NIQKI_HOST_GPU_INFLATE_MIN sets that number of files; NIQKI_HOST_NO_GPU_INFLATE: always the readers).
  const unsigned n_threads = host_threads();
  size_t n_gz_names = 0, gz_min = size_t(32) * n_threads;
  for (const auto &p : paths) n_gz_names += p.size() > 3 && p.compare(p.size() - 3, 3, ".gz") == 0;
  if (const char *v = std::getenv("NIQKI_HOST_GPU_INFLATE_MIN")) gz_min = (size_t)std::max(0, std::atoi(v));
  const bool gz_list = std::getenv("NIQKI_HOST_NO_GPU_INFLATE") == nullptr && 2 * n_gz_names > paths.size() && n_gz_names >= gz_min;
  OrderedFileReader rd(paths, n_threads, gz_list ? std::min(kGzReaderBufs, 2 * paths.size() + 64) : kReaderBufs,
                       std::getenv("NIQKI_HOST_NO_GPU_INFLATE") ? 0 : gz_list ? 2 : 1);
  size_t i = 0, n_batches = 0;
  auto assemble = [&](Batch &b) {
    const auto t0 = clk::now();
    // the first batches are small (16, 32, 64 files): the GPU and the copy engine start while the reader threads are
    // still page-locking their buffers and filling the pipeline.  (Gzip lists: equal batches of at most 2048 -- a launch
    // of the device inflate takes as long as its longest file however few files it holds, so small batches only cost: on
    // 2048 files 256 + 1024 + 768 gives 3.6 k genomes/s, 1024 + 1024 5.1 k, one batch 5.5 k; on 4096 files
    // 1024 + 2048 + 1024 gives 5.3 k.)
    const size_t gz_batches = (paths.size() + kGzBatchFiles - 1) / kGzBatchFiles;
    const size_t limit = gz_list ? (paths.size() + gz_batches - 1) / std::max<size_t>(gz_batches, 1)
                                 : std::min<size_t>(kWholeBatchFiles, size_t(16) << std::min<size_t>(n_batches++, 8));
    while (b.files.size() < limit && b.bytes < (gz_list ? kGzBatchWireBytes : kWholeBatchBytes) && b.raw_bytes < kGzBatchRawBytes) {
      auto *f = rd.next();
      if (!f) break;
      b.files.push_back(f);
      b.names.push_back(paths[i++]);
      b.bytes += f->buf.size;
      size_t raw = f->buf.size;
      if (f->gz) {   // (the reader checked the trailer: at least 18 bytes, a plausible size)
        const uint8_t *e = f->buf.p + f->buf.size - 4;
        raw = (size_t)e[0] | (size_t)e[1] << 8 | (size_t)e[2] << 16 | (size_t)e[3] << 24;
        if (raw < f->buf.size) raw = 4 * f->buf.size;   // (a file of tagged members: the trailer is its last member's)
        ++b.n_gz;
      }
      b.raw_bytes += raw;
    }
    t_wait += std::chrono::duration<double>(clk::now() - t0).count();
  };
  // The bytes of batch i + 1 cross PCIe while batch i is worked on (the library keeps two prefetch buffers).  Gzip lists
  // (2048 files, gigabytes a batch, readers far ahead of the device): batch i + 1 is put together and sent on its way
  // BEFORE batch i is staged, so the copy runs under batch i's inflate kernel.  Other lists (64 files a batch, readers
  // about as fast as the device): batch i is staged first, so that the readers' buffers are not all spoken for while it
  // runs -- waiting for batch i + 1 up front cost a third of the rate there.
  // Queries: a batch's hit lines are put together and written by a thread of their own while the next batch is on the
  // GPU (at most four batches wait there; output order is batch order).  With thousands of hits per query the text
  // took as long as the GPU calls: 4000 virus-sized genomes against themselves, 0.20 of 0.43 s.
  struct HitsWriter {
    Index *ix;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::unique_ptr<Hits>> q;
    bool closed = false;
    std::string err;
    std::thread th;
    explicit HitsWriter(Index *i) : ix(i), th([this] { run(); }) {}
    void run() {
      for (;;) {
        std::unique_ptr<Hits> h;
        {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return closed || !q.empty(); });
          if (q.empty()) return;
          h = std::move(q.front());
          q.pop_front();
        }
        cv.notify_all();
        try {
          if (err.empty()) ix->write_hits(*h);
        } catch (const std::exception &e) {
          std::lock_guard<std::mutex> g(mu);
          err = e.what();
        }
      }
    }
    void push(std::unique_ptr<Hits> h) {
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return q.size() < 4; });
      q.push_back(std::move(h));
      g.unlock();
      cv.notify_all();
    }
    void finish() {   // everything handed over is written (or the first error kept) when this returns
      {
        std::lock_guard<std::mutex> g(mu);
        closed = true;
      }
      cv.notify_all();
      if (th.joinable()) th.join();
    }
    ~HitsWriter() { finish(); }
  };
  std::unique_ptr<HitsWriter> writer;
  if (flush == &Index::flush_query) {
    writer.reset(new HitsWriter(this));
    hits_sink_ = [&writer](std::unique_ptr<Hits> h) { writer->push(std::move(h)); };
  }
  struct SinkReset {   // (also when a GPU call throws: nothing may hand hits to a writer that is gone)
    std::function<void(std::unique_ptr<Hits>)> &f;
    ~SinkReset() { f = nullptr; }
  } sink_reset{hits_sink_};
  Batch cur, nxt;
  assemble(cur);
  if (!cur.files.empty()) stage_batch(cur, true);
  while (!cur.files.empty()) {
    if (gz_list) {
      assemble(nxt);
      if (!nxt.files.empty()) stage_batch(nxt, true);
    }
    auto t0 = clk::now();
    Lap lap;
    stage_batch(cur, false);
    lap.to(t_stage_);
    t_gpu += std::chrono::duration<double>(clk::now() - t0).count();
    if (!gz_list) {
      assemble(nxt);
      if (!nxt.files.empty()) stage_batch(nxt, true);
    }
    t0 = clk::now();
    (this->*flush)(cur);
    t_gpu += std::chrono::duration<double>(clk::now() - t0).count();
    for (auto *f : cur.files) rd.release(f);
    cur = std::move(nxt);
    nxt = Batch();
  }
  if (writer) {
    Lap lap;
    writer->finish();
    lap.to(t_out_);
    if (!writer->err.empty()) throw std::runtime_error(writer->err);
  }
  if (timing)
    std::cerr << "[niqki timing] " << paths.size() << " files: total "
              << std::chrono::duration<double>(clk::now() - t_begin).count() << " s, waiting for file bytes " << t_wait
              << " s, GPU calls (copy + frame + sketch + insert/query + output) " << t_gpu << " s (copy + frame "
              << t_stage_ << ", sketch + insert/query " << t_dev_ << ", output " << t_out_ << ")" << std::endl;
}

void Index::insert_file_of_file_whole(const std::string &filestr) {
  std::ifstream in(filestr);
  if (!in) {
    std::cout << "Unable to open the file '" << filestr << "'" << std::endl;
    exit(0);  // src/niqki_index.cpp:464-467
  }
  // the list first (:479-490: lines longer than 2 characters naming an existing
  // file; ids follow the list order), then the files, read in parallel
  std::vector<std::string> paths;
  std::string ref;
  while (!in.eof()) {
    std::getline(in, ref);
    if (ref.size() > 2 && exists_test(ref)) paths.push_back(ref);
    ref.clear();
  }
  for_each_batch(paths, &Index::flush_insert);
}

void Index::query_file_of_file_whole(const std::string &filestr) {
  GzReader in(filestr);
  std::vector<std::string> paths;
  std::string ref;
  while (!in.eof()) {
    in.getline(ref);
    if (exists_test(ref)) paths.push_back(ref);  // :534
    ref.clear();
  }
  for_each_batch(paths, &Index::flush_query);
}

// ---- lines mode: reader thread -> GPU calls (caller's thread) -> writer thread ------
namespace {

// End of the last complete record in buf[0, size): the reader cuts its pieces there, so
// every piece holds whole records (the GPU frames them; this is only a safe place to cut).
// FASTA: a line that starts with '>' begins a record.  FASTQ: records are 4 lines, and a
// piece starts at a record, so the cut is after a multiple of 4 newlines.  0 = none yet.
size_t record_cut(const uint8_t *buf, size_t size, char type) {
  if (type == 'Q') {
    size_t lines = 0, cut = 0;
    const uint8_t *p = buf, *end = buf + size;
    while (p < end) {
      const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;
      p = nl + 1;
      if ((++lines & 3) == 0) cut = (size_t)(p - buf);
    }
    return cut;
  }
  size_t hi = size;
  while (hi > 1) {
    const uint8_t *gt = (const uint8_t *)memrchr(buf + 1, '>', hi - 1);
    if (!gt) return 0;
    if (gt[-1] == '\n') return (size_t)(gt - buf);
    hi = (size_t)(gt - buf);
  }
  return 0;
}

// Small blocking queue for handing buffers between the stages.
template <typename T>
class Channel {
 public:
  void push(T v) {
    {
      std::lock_guard<std::mutex> g(mu_);
      q_.push_back(std::move(v));
    }
    cv_.notify_all();
  }
  T pop() {
    std::unique_lock<std::mutex> g(mu_);
    cv_.wait(g, [&] { return !q_.empty(); });
    T v = std::move(q_.front());
    q_.pop_front();
    return v;
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<T> q_;
};

struct Piece {
  PinnedBuf buf;
  bool last = false;
  std::string err;
  std::atomic<int> refs{0};   // the GPU thread while it works on the piece + every batch of hits whose names still lie in it
};

}  // namespace

// One entry per record longer than K, named by its header line (insert_file_lines /
// query_file_lines, :383-430).  A reader thread streams the file into page-locked pieces
// that end at a record boundary, this thread runs the GPU calls, a writer thread formats
// and compresses the hits: the three overlap, output order is input order.
void Index::stream_lines(const std::string &filestr, bool insert) {
  using clk = std::chrono::steady_clock;
  const auto t_begin = clk::now();
  uint64_t n_entries_total = 0;
  const char type = data_type(filestr);
  const uint8_t type_u8 = (uint8_t)type;
  // sketches of one call stay below 2 GB
  const uint32_t max_entries = (uint32_t)std::min<uint64_t>(65536, std::max<uint64_t>(1024, (uint64_t(2) << 30) / ((uint64_t)F * 4)));
  constexpr size_t kPieces = 6;   // one being read, one on the GPU, up to four whose header lines the writer still needs
  std::deque<Piece> pieces(kPieces);
  Channel<Piece *> free_q, ready_q;
  for (auto &pc : pieces) free_q.push(&pc);

  std::thread reader([&] {
    std::vector<uint8_t> carry;
    size_t target = size_t(8) << 20;
    bool eof = false;
    try {
      GzReader in(filestr);
      while (!eof) {
        Piece *pc = free_q.pop();
        if (!pc) return;  // the consumer gave up
        pc->err.clear();
        pc->last = false;
        pc->buf.size = 0;
        pc->buf.reserve(std::max(target, carry.size()) + (size_t(1) << 16));
        if (!carry.empty()) std::memcpy(pc->buf.p, carry.data(), carry.size());   // (an empty vector's data() may be null)
        pc->buf.size = carry.size();
        carry.clear();
        size_t cut = 0;
        for (;;) {
          while (!eof && pc->buf.size < target) {
            const size_t n = in.read(pc->buf.p + pc->buf.size, target - pc->buf.size);
            pc->buf.size += n;
            if (n == 0 || in.eof()) eof = true;
          }
          if (eof) { cut = pc->buf.size; break; }
          cut = record_cut(pc->buf.p, pc->buf.size, type);
          if (cut) break;
          target *= 2;  // not one complete record yet: a longer piece
          pc->buf.reserve(target + (size_t(1) << 16));
        }
        carry.assign(pc->buf.p + cut, pc->buf.p + pc->buf.size);
        pc->buf.size = cut;
        pc->last = eof;
        ready_q.push(pc);
      }
    } catch (const std::exception &e) {
      Piece *pc = free_q.pop();
      if (pc) { pc->err = e.what(); pc->last = true; pc->buf.size = 0; ready_q.push(pc); }
    }
  });

  Channel<Hits *> out_q;
  Channel<Hits *> out_free;
  std::deque<Hits> results(4);
  for (auto &r : results) out_free.push(&r);
  std::string writer_err;
  auto drop_ref = [&](Piece *pc) {
    if (pc->refs.fetch_sub(1) == 1) free_q.push(pc);   // the last user hands the piece back to the reader
  };
  std::thread writer([&] {
    for (;;) {
      Hits *h = out_q.pop();
      if (!h) return;
      try {
        if (h->keep) {   // the names: the header lines of the entries, cut out here instead of on the GPU thread
          h->names.clear();
          h->names.reserve(h->name_at.size());
          for (uint64_t at : h->name_at) {
            const uint8_t *b = h->name_base + at;
            const uint8_t *nl = (const uint8_t *)memchr(b, '\n', h->name_room - at);
            h->names.emplace_back((const char *)b, nl ? (size_t)(nl - b) : (size_t)(h->name_room - at));
          }
        }
        if (writer_err.empty()) write_hits(*h);
      } catch (const std::exception &e) { writer_err = e.what(); }
      if (h->keep) {
        Piece *pc = (Piece *)h->keep;
        h->keep = nullptr;
        drop_ref(pc);
      }
      out_free.push(h);
    }
  });

  std::string err;
  std::vector<uint64_t> hdr(max_entries);
  double t_piece = 0, t_frame = 0, t_names = 0, t_engine = 0, t_slot = 0;   // NIQKI_HOST_TIMING: where this thread's time goes
  try {
    for (bool last = false; !last;) {
      Lap lap;
      Piece *pc = ready_q.pop();
      pc->refs.store(1);
      lap.to(t_piece);
      last = pc->last;
      if (!pc->err.empty()) throw std::runtime_error(pc->err);
      size_t at = 0;
      while (grp_ && at < pc->buf.size) {
        // multi-GPU: the rest of the piece is cut into one run of whole records per GPU; every GPU
        // frames and sketches its run, the entries keep the order of the file
        const size_t G = sh_.size(), left = pc->buf.size - at;
        const uint8_t *base = pc->buf.p + at;
        std::vector<size_t> cutp(G + 1, 0);
        cutp[G] = left;
        for (size_t r = 1; r < G; ++r) cutp[r] = std::max(cutp[r - 1], record_cut(base, left * r / G, type));
        std::vector<niqki_stage_info> info(G);
        std::vector<std::vector<uint64_t>> hdrs(G, std::vector<uint64_t>(max_entries));
        std::vector<uint32_t> n_entry(G, 0);
        size_t done = left;       // bytes of the piece this round covers
        for (size_t r = 0; r < G; ++r) {
          if (cutp[r + 1] == cutp[r]) continue;
          const uint64_t off[2] = {0, cutp[r + 1] - cutp[r]};
          niqki_raw_batch rb{};
          rb.raw = base + cutp[r];
          rb.file_off = off;
          rb.file_type = &type_u8;
          rb.n_files = 1;
          rb.lines = 1;
          rb.final = 1;
          rb.max_entries = max_entries;
          const int rc = niqki_stage_raw(sh_[r], &rb, NIQKI_MEM_HOST, &info[r], hdrs[r].data());
          if (rc) throw std::runtime_error(std::string("niqki_stage_raw: ") + niqki_status_string(rc) + " (" + niqki_last_error(sh_[r]) + ")");
          n_entry[r] = info[r].n_entry;
          if (info[r].consumed < off[1]) {   // more entries than one call takes: the round ends inside this run
            done = cutp[r] + info[r].consumed;
            break;
          }
        }
        uint32_t per = 0, total = 0;
        for (size_t r = 0; r < G; ++r) { per = std::max(per, n_entry[r]); total += n_entry[r]; }
        n_entries_total += total;
        if (total) {
          Hits *h = insert ? nullptr : out_free.pop();
          std::vector<std::string> local;
          std::vector<std::string> &names = h ? h->names : local;
          names.clear();
          for (size_t r = 0; r < G; ++r)
            for (uint32_t e = 0; e < n_entry[r]; ++e) {
              const uint8_t *b = base + cutp[r] + hdrs[r][e];
              const size_t room = left - cutp[r] - hdrs[r][e];
              const uint8_t *nl = (const uint8_t *)memchr(b, '\n', room);
              names.emplace_back((const char *)b, nl ? (size_t)(nl - b) : room);
            }
          if (insert) {
            check_group(niqki_group_staged_insert(grp_, per, n_entry.data()), "niqki_group_staged_insert");
            for (auto &nm : names) filenames.push_back(nm);
          } else {
            try { group_query_staged(per, n_entry, *h); } catch (...) { out_free.push(h); throw; }
            out_q.push(h);
          }
        }
        if (done == 0 && total == 0) break;   // nothing but a header without end
        at += done;
      }
      if (!grp_) do {
        const uint64_t off[2] = {0, pc->buf.size - at};
        niqki_raw_batch rb{};
        rb.raw = pc->buf.p + at;
        rb.file_off = off;
        rb.file_type = &type_u8;
        rb.n_files = 1;
        rb.lines = 1;
        rb.final = 1;  // pieces hold whole records
        rb.max_entries = max_entries;
        niqki_stage_info info{};
        Lap lap2;
        check(niqki_stage_raw(h_, &rb, NIQKI_MEM_HOST, &info, hdr.data()), "niqki_stage_raw");
        lap2.to(t_frame);
        n_entries_total += info.n_entry;
        if (info.n_entry) {
          Hits *h = insert ? nullptr : out_free.pop();
          lap2.to(t_slot);
          const uint8_t *base = pc->buf.p + at;
          const size_t left = pc->buf.size - at;
          if (insert) {
            for (uint32_t e = 0; e < info.n_entry; ++e) {
              const uint8_t *b = base + hdr[e];
              const uint8_t *nl = (const uint8_t *)memchr(b, '\n', left - hdr[e]);
              filenames.emplace_back((const char *)b, nl ? (size_t)(nl - b) : (size_t)(left - hdr[e]));
            }
            lap2.to(t_names);
            check(niqki_staged_insert(h_), "niqki_staged_insert");
          } else {
            // the writer thread cuts the names out of the piece (it stays alive until then: Piece::refs)
            h->names.clear();
            h->name_base = base;
            h->name_room = left;
            h->name_at.assign(hdr.begin(), hdr.begin() + info.n_entry);
            h->keep = pc;
            pc->refs.fetch_add(1);
            lap2.to(t_names);
            try { query_staged(info.n_entry, *h); } catch (...) { h->keep = nullptr; pc->refs.fetch_sub(1); out_free.push(h); throw; }
            out_q.push(h);
          }
          lap2.to(t_engine);
        }
        if (info.consumed == 0 && info.n_entry == 0) break;  // nothing but a header without end
        at += info.consumed;
      } while (at < pc->buf.size);
      drop_ref(pc);
    }
  } catch (const std::exception &e) {
    err = e.what();
    free_q.push(nullptr);  // releases a reader that waits for a buffer
  }
  reader.join();
  out_q.push(nullptr);
  writer.join();
  if (!err.empty()) throw std::runtime_error(err);
  if (!writer_err.empty()) throw std::runtime_error(writer_err);
  if (std::getenv("NIQKI_HOST_TIMING"))
    std::cerr << "[niqki timing] lines mode: " << n_entries_total << " entries in "
              << std::chrono::duration<double>(clk::now() - t_begin).count() << " s (this thread: waiting for file bytes " << t_piece
              << ", copy + frame " << t_frame << ", names " << t_names << ", sketch + insert/query " << t_engine
              << ", waiting for the writer " << t_slot << ")" << std::endl;
}

void Index::insert_file_lines(const std::string &filestr) { stream_lines(filestr, true); }

void Index::query_file_lines(const std::string &filestr) { stream_lines(filestr, false); }

void Index::output_query(const query_output &toprint, const std::string &queryname) {
  if (pretty_printing) {  // :546-553
    std::string line = queryname + " ";
    for (const auto &h : toprint) {
      line += filenames[h.second];
      line += ':';
      line += fmt_double((double)h.first / F);
      line += ' ';
    }
    line += '\n';
    outfile->write(line);
  } else {  // :555-564 (unreachable from the reference CLI, kept for API parity)
    outfile->write(queryname + "\n");
    uint32_t size = (uint32_t)toprint.size();
    outfile->write(&size, 4);
    for (const auto &h : toprint) {
      outfile->write(&h.second, 4);
      outfile->write(&h.first, 4);
    }
  }
}

// ---- matrix ----------------------------------------------------------------------

void Index::output_matrix_row(const uint16_t *counts, const std::string &queryname) {
  // query_range threshold (:600-606) + output_matrix (:747-763)
  std::string line = queryname + "\t";
  const size_t n = filenames.size();
  for (size_t j = 0; j < n; ++j) {
    double v = counts[j] >= min_score ? (double)counts[j] / F : 0.0;
    line += fmt_double(v);
    line += '\t';
  }
  line += '\n';
  outfile->write(line);
}

void Index::query_matrix() {
  std::string head = "##Names\t";  // :615-619
  for (const auto &nm : filenames) { head += nm; head += '\t'; }
  head += '\n';
  outfile->write(head);
  const uint32_t n = (uint32_t)filenames.size();
  const uint64_t stride = NIQKI_ROW_STRIDE(n);
  const uint32_t rows = 256;
  std::vector<uint16_t> counts((size_t)rows * std::max<uint64_t>(stride, 2));
  std::vector<uint16_t> part(grp_ ? counts.size() : 0);
  for (uint32_t t0 = 0; t0 < n; t0 += rows) {
    const uint32_t t1 = std::min(n, t0 + rows);
    check(niqki_matrix_range(h_, t0, t1, counts.data(), stride, NIQKI_MEM_HOST), "niqki_matrix_range");
    for (size_t r = 1; r < sh_.size(); ++r) {   // slot shards: co-occurrence counts add up over the slots
      check(niqki_matrix_range(sh_[r], t0, t1, part.data(), stride, NIQKI_MEM_HOST), "niqki_matrix_range");
      const size_t m = (size_t)(t1 - t0) * stride;
      for (size_t i = 0; i < m; ++i) counts[i] = (uint16_t)(counts[i] + part[i]);
    }
    for (uint32_t t = t0; t < t1; ++t) output_matrix_row(counts.data() + (size_t)(t - t0) * stride, filenames[t]);
  }
}

// ---- dump ------------------------------------------------------------------------

void Index::dump_index_disk(const std::string &filestr) {
  // header + buckets (src/niqki_index.cpp:42-55) exported in groups of whole slots
  // and gzipped in parallel, then the names (:56-58)
  const bool timing = std::getenv("NIQKI_HOST_TIMING") != nullptr;
  Lap lap;
  double t_layout = 0, t_alloc = 0, t_export = 0, t_writer = 0;
  ParallelGzWriter out(filestr, host_threads() + 1);   // (the CPUs this process may use: its cgroup quota counts)
  {
    std::vector<uint8_t> hdr(24);
    check(niqki_export_dump_header(h_, hdr.data()), "niqki_export_dump_header");
    out.add(hdr);
  }
  const uint64_t target = uint64_t(32) << 20;
  uint64_t payload = 0;
  for (size_t r = 0; r < sh_.size(); ++r) {   // the shards' slots in rank order are the whole payload
    uint32_t sb = 0, se = F;
    if (grp_) niqki_group_slot_range((uint32_t)r, (uint32_t)sh_.size(), lF, &sb, &se);
    const uint32_t f_local = se - sb;
    std::vector<uint64_t> slot_bytes((size_t)f_local + 1);
    check(niqki_export_dump_layout(sh_[r], slot_bytes.data()), "niqki_export_dump_layout");
    lap.to(t_layout);
    uint32_t s0 = 0;
    while (s0 < f_local) {
      uint32_t s1 = s0 + 1;
      while (s1 < f_local && slot_bytes[s1 + 1] - slot_bytes[s0] <= target) ++s1;
      const uint64_t want = slot_bytes[s1] - slot_bytes[s0];
      ParallelGzWriter::Block block(want);
      lap.to(t_alloc);
      uint64_t size = 0;
      check(niqki_export_dump_slots(sh_[r], s0, s1, block.data(), want, &size), "niqki_export_dump_slots");
      lap.to(t_export);
      payload += want;
      out.add(std::move(block));
      lap.to(t_writer);
      s0 = s1;
    }
  }
  std::vector<uint8_t> names;
  for (const auto &nm : filenames) {
    names.insert(names.end(), nm.begin(), nm.end());
    names.push_back('\n');
  }
  out.add(names);
  out.finish();
  lap.to(t_writer);
  if (timing)
    std::cerr << "[niqki timing] dump: " << payload / 1e9 << " GB of buckets: layout " << t_layout << " s, buffers " << t_alloc
              << ", export (kernel + copy) " << t_export << ", waiting for the gzip writer " << t_writer << std::endl;
}

}  // namespace nqhost
