// index_host.cpp -- file-level drivers of the `niqki` host program.  Reads
// FASTA/FASTQ exactly like the reference (seqio.h), batches records, and calls
// the gfx950 engine through the C ABI (include/niqki_hip.h).  No sketching,
// counting or sorting happens on the host.
#include "index_host.h"

#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <future>
#include <memory>
#include <stdexcept>
#include <thread>

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
constexpr size_t kBatchBytes = size_t(1) << 30;   // sequence bytes per GPU call
constexpr size_t kBatchEntries = 16384;           // sketches per GPU call

// All records longer than K of one sequence file (insert_file_whole /
// query_file_whole read loop, src/niqki_index.cpp:446-453, :510-515).
std::vector<std::string> read_whole_file(const std::string &path, size_t K) {
  std::vector<std::string> recs;
  const char type = data_type(path);
  GzReader fin(path);
  std::string rec, header;
  while (!fin.eof()) {
    bio_getline(fin, rec, type, header, K);
    if (rec.size() > K) recs.push_back(rec);
  }
  return recs;
}

// Reads (and gunzips) the files of a list on several threads -- the reference
// does this part in its OpenMP region, one file per thread -- and hands them to
// `consume` strictly in list order, so genome ids and output order are those of
// a single-threaded run.
template <typename Consume>
void for_each_file_in_order(const std::vector<std::string> &paths, size_t K, Consume consume) {
  unsigned threads = std::thread::hardware_concurrency();
  threads = threads ? std::min(threads, 32u) : 4u;
  const size_t window = (size_t)threads * 2;
  for (size_t w0 = 0; w0 < paths.size(); w0 += window) {
    const size_t w1 = std::min(paths.size(), w0 + window);
    std::vector<std::vector<std::string>> recs(w1 - w0);
    std::vector<std::string> errs(w1 - w0);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
      pool.emplace_back([&, t] {
        for (size_t i = w0 + t; i < w1; i += threads) {
          try { recs[i - w0] = read_whole_file(paths[i], K); } catch (const std::exception &e) { errs[i - w0] = e.what(); }
        }
      });
    for (auto &th : pool) th.join();
    for (size_t i = w0; i < w1; ++i) {
      if (!errs[i - w0].empty()) throw std::runtime_error(errs[i - w0]);
      consume(paths[i], recs[i - w0]);
    }
  }
}
}  // namespace

// A batch of records on their way to the GPU: entry e = records
// [entry_rec[e], entry_rec[e+1]) of seqs, name[e] its label.
struct Index::Batch {
  std::vector<uint8_t> seqs;
  std::vector<uint64_t> rec_off{0};
  std::vector<uint32_t> entry_rec{0};
  std::vector<std::string> names;
  void add_record(const std::string &s) {
    seqs.insert(seqs.end(), s.begin(), s.end());
    rec_off.push_back(seqs.size());
  }
  void end_entry(const std::string &name) {
    entry_rec.push_back((uint32_t)(rec_off.size() - 1));
    names.push_back(name);
  }
  size_t n_entries() const { return names.size(); }
  bool full() const { return seqs.size() >= kBatchBytes || names.size() >= kBatchEntries; }
  void clear() {
    seqs.clear();
    rec_off.assign(1, 0);
    entry_rec.assign(1, 0);
    names.clear();
  }
};

// Runs the GPU call of a full batch on a helper thread while the caller parses
// the next one; batches are flushed strictly one after another (the handle is
// used by one thread at a time and output order is input order).
class Index::Pipeline {
 public:
  using Fn = void (Index::*)(Batch &);
  Pipeline(Index *ix, Fn fn) : ix_(ix), fn_(fn) {}
  ~Pipeline() { try { wait(); } catch (...) {} }
  void submit(Batch &b) {
    wait();
    auto job = std::make_shared<Batch>(std::move(b));
    b.clear();
    pending_ = std::async(std::launch::async, [this, job] { (ix_->*fn_)(*job); });
  }
  void wait() {
    if (pending_.valid()) pending_.get();  // rethrows a failure of the helper thread
  }

 private:
  Index *ix_;
  Fn fn_;
  std::future<void> pending_;
};

void Index::check(int rc, const char *what) const {
  if (rc == NIQKI_OK) return;
  throw std::runtime_error(std::string(what) + ": " + niqki_status_string(rc) + " (" + niqki_last_error(h_) + ")");
}

Index::Index(uint32_t ilF, uint32_t iK, uint32_t iW, uint32_t iH, const std::string &out_filename,
             double min_fract, int device) {
  niqki_params p{};
  p.K = iK; p.S = ilF; p.W = iW; p.H = iH;
  p.min_score = niqki_min_score(min_fract, ilF);
  p.device = device;
  int rc = niqki_create(&p, &h_);
  if (rc) throw std::runtime_error(std::string("niqki_create: ") + niqki_status_string(rc) + " (" + niqki_last_error(nullptr) + ")");
  K = iK; W = iW; H = iH; lF = ilF; F = 1u << ilF; min_score = p.min_score;
  outfile.reset(new GzWriter(out_filename));
}

Index::Index(const std::string &dump_file, bool pretty, const std::string &out_filename, int device) {
  pretty_printing = pretty;
  // The dump is streamed: header, then the buckets in groups of whole slots (the
  // payload of a 100k-genome index is 13.6 GB), then the names.
  GzReader in(dump_file);
  uint8_t hdr[24];
  if (in.read(hdr, 24) != 24) throw std::runtime_error("'" + dump_file + "' is not a niqki dump");
  niqki_params p{};
  p.device = device;
  int rc = niqki_import_begin(&p, hdr, &h_);
  if (rc) throw std::runtime_error(std::string("niqki_import_begin: ") + niqki_status_string(rc) + " (" + niqki_last_error(nullptr) + ")");
  niqki_params q{};
  niqki_get_params(h_, &q);
  K = q.K; W = q.W; H = q.H; lF = q.S; F = 1u << q.S; min_score = q.min_score;
  const uint32_t R = 1u << W;
  std::vector<uint8_t> chunk;
  uint32_t s0 = 0;
  auto flush = [&](uint32_t s1) {
    uint64_t used = 0;
    check(niqki_import_slots(h_, s0, s1, chunk.data(), chunk.size(), &used), "niqki_import_slots");
    chunk.clear();
    s0 = s1;
  };
  for (uint32_t s = 0; s < F; ++s) {
    for (uint32_t fp = 0; fp < R; ++fp) {
      uint32_t size = 0;
      if (in.read(&size, 4) != 4) throw std::runtime_error("'" + dump_file + "' is truncated");
      const size_t at = chunk.size();
      chunk.resize(at + 4 + (size_t)size * 4);
      std::memcpy(chunk.data() + at, &size, 4);
      if (size && in.read(chunk.data() + at + 4, (size_t)size * 4) != (size_t)size * 4)
        throw std::runtime_error("'" + dump_file + "' is truncated");
    }
    if (chunk.size() >= kBatchBytes / 8) flush(s + 1);
  }
  flush(F);
  // genome names, one per line after the buckets (src/niqki_index.cpp:91-95)
  const uint32_t n = niqki_genome_count(h_);
  std::string name;
  for (uint32_t i = 0; i < n; ++i) {
    in.getline(name);
    filenames.push_back(name);
  }
  outfile.reset(new GzWriter(out_filename));
}

Index::~Index() {
  if (outfile) outfile->close();
  if (h_) niqki_destroy(h_);
}

// src/niqki_index.cpp:126-138, including its message
void Index::select_best_H(double genome_size) {
  uint32_t chosen = H;
  check(niqki_select_best_H(h_, genome_size, &chosen), "select_best_H");
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
  check(niqki_insert(h_, sketch.data(), 1, NIQKI_MEM_HOST), "niqki_insert");
}

query_output Index::query_sketch(const std::vector<int32_t> &sketch) const {
  const uint32_t n = niqki_genome_count(h_);
  std::vector<uint32_t> hc(n ? n : 1), hg(n ? n : 1);
  uint64_t off[2] = {0, 0};
  check(niqki_query(h_, sketch.data(), 1, off, hc.data(), hg.data(), n, NIQKI_MEM_HOST), "niqki_query");
  query_output r;
  for (uint64_t i = 0; i < off[1]; ++i) r.push_back({hc[i], hg[i]});
  return r;
}

// ---- insertion ---------------------------------------------------------------

void Index::flush_insert(Batch &b) {
  const size_t n = b.n_entries();
  if (!n) return;
  std::vector<int32_t> sk(n * (size_t)F);
  check(niqki_sketch(h_, b.seqs.data(), b.rec_off.data(), (uint32_t)(b.rec_off.size() - 1), b.entry_rec.data(),
                     (uint32_t)n, sk.data(), NIQKI_MEM_HOST), "niqki_sketch");
  check(niqki_insert(h_, sk.data(), (uint32_t)n, NIQKI_MEM_HOST), "niqki_insert");
  for (auto &nm : b.names) filenames.push_back(nm);
  b.clear();
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
  Batch b;
  Pipeline pipe(this, &Index::flush_insert);
  for_each_file_in_order(paths, K, [&](const std::string &path, const std::vector<std::string> &recs) {
    // insert_file_whole (:442-456): every record longer than K goes into ONE sketch
    for (const auto &r : recs) b.add_record(r);
    b.end_entry(path);
    if (b.full()) pipe.submit(b);
  });
  pipe.submit(b);
  pipe.wait();
}

void Index::insert_file_lines(const std::string &filestr) {
  const char type = data_type(filestr);
  GzReader in(filestr);
  Batch b;
  Pipeline pipe(this, &Index::flush_insert);
  std::string ref, header;
  while (!in.eof()) {
    bio_getline(in, ref, type, header, K);
    if (ref.size() > K) {  // :395: one entry per record, named by its header line
      b.add_record(ref);
      b.end_entry(header);
      if (b.full()) pipe.submit(b);
    }
  }
  pipe.submit(b);
  pipe.wait();
}

// ---- query ---------------------------------------------------------------------

void Index::flush_query(Batch &b) {
  const size_t n = b.n_entries();
  if (!n) return;
  const uint64_t N = niqki_genome_count(h_);
  uint64_t cap = std::max<uint64_t>(uint64_t(1) << 20, n * 64);
  std::vector<uint64_t> off(n + 1);
  std::vector<uint32_t> hc, hg;
  for (;;) {
    hc.resize(cap);
    hg.resize(cap);
    int rc = niqki_query_sequences(h_, b.seqs.data(), b.rec_off.data(), (uint32_t)(b.rec_off.size() - 1),
                                   b.entry_rec.data(), (uint32_t)n, off.data(), hc.data(), hg.data(), cap,
                                   NIQKI_MEM_HOST);
    if (rc == NIQKI_E_CAPACITY && cap < n * N) { cap = std::max(off[n], cap * 2); continue; }
    check(rc, "niqki_query_sequences");
    break;
  }
  query_output one;
  for (size_t i = 0; i < n; ++i) {
    one.clear();
    for (uint64_t j = off[i]; j < off[i + 1]; ++j) one.push_back({hc[j], hg[j]});
    output_query(one, b.names[i]);
  }
  b.clear();
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
  Batch b;
  Pipeline pipe(this, &Index::flush_query);
  for_each_file_in_order(paths, K, [&](const std::string &path, const std::vector<std::string> &recs) {
    for (const auto &r : recs) b.add_record(r);  // query_file_whole :505-519
    b.end_entry(path);
    if (b.full()) pipe.submit(b);
  });
  pipe.submit(b);
  pipe.wait();
}

void Index::query_file_lines(const std::string &filestr) {
  const char type = data_type(filestr);
  GzReader in(filestr);
  Batch b;
  Pipeline pipe(this, &Index::flush_query);
  std::string ref, head;
  while (!in.eof()) {
    bio_getline(in, ref, type, head, K);
    if (ref.size() > K) {
      b.add_record(ref);
      b.end_entry(head);
      if (b.full()) pipe.submit(b);
    }
  }
  pipe.submit(b);
  pipe.wait();
}

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
  const uint64_t stride = ((uint64_t)n + 1) & ~1ull;
  const uint32_t rows = 256;
  std::vector<uint16_t> counts((size_t)rows * std::max<uint64_t>(stride, 2));
  for (uint32_t t0 = 0; t0 < n; t0 += rows) {
    const uint32_t t1 = std::min(n, t0 + rows);
    check(niqki_matrix_range(h_, t0, t1, counts.data(), stride, NIQKI_MEM_HOST), "niqki_matrix_range");
    for (uint32_t t = t0; t < t1; ++t) output_matrix_row(counts.data() + (size_t)(t - t0) * stride, filenames[t]);
  }
}

// ---- dump ------------------------------------------------------------------------

void Index::dump_index_disk(const std::string &filestr) {
  // header + buckets (src/niqki_index.cpp:42-55) exported in groups of whole slots
  // and gzipped in parallel, then the names (:56-58)
  unsigned threads = std::thread::hardware_concurrency();
  threads = threads ? std::min(threads, 32u) : 4u;
  ParallelGzWriter out(filestr, threads);
  std::vector<uint8_t> block(24);
  check(niqki_export_dump_header(h_, block.data()), "niqki_export_dump_header");
  std::vector<uint64_t> slot_bytes((size_t)F + 1);
  check(niqki_export_dump_layout(h_, slot_bytes.data()), "niqki_export_dump_layout");
  const uint64_t target = uint64_t(32) << 20;
  uint32_t s0 = 0;
  while (s0 < F) {
    uint32_t s1 = s0 + 1;
    while (s1 < F && slot_bytes[s1 + 1] - slot_bytes[s0] <= target) ++s1;
    const size_t at = block.size();
    const uint64_t want = slot_bytes[s1] - slot_bytes[s0];
    block.resize(at + want);
    uint64_t size = 0;
    check(niqki_export_dump_slots(h_, s0, s1, block.data() + at, want, &size), "niqki_export_dump_slots");
    out.add(std::move(block));
    block.clear();
    s0 = s1;
  }
  for (const auto &nm : filenames) {
    block.insert(block.end(), nm.begin(), nm.end());
    block.push_back('\n');
  }
  out.add(std::move(block));
  out.finish();
}

}  // namespace nqhost
