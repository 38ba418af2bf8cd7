// gzio.h -- line-oriented gzip/plain input and gzip(level 1) output on zlib.
// Own replacement for the reference's vendored zstr streams (src/zstr.hpp):
// input auto-detects gzip vs plain like zstr::ifstream (:190-203), output is
// gzip level 1, window 15+16, like zstr::ofstream (:103,:458).
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <future>
#include <memory>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace nqhost {

class GzReader {
 public:
  explicit GzReader(const std::string &path) : buf_(1 << 20) {
    f_ = gzopen(path.c_str(), "rb");
    if (!f_) throw std::runtime_error("cannot open '" + path + "'");
    gzbuffer(f_, 1 << 20);
  }
  ~GzReader() { if (f_) gzclose(f_); }
  GzReader(const GzReader &) = delete;
  GzReader &operator=(const GzReader &) = delete;

  // std::istream::eof() analogue: true once a read hit the end of the data
  bool eof() const { return eof_; }
  // std::istream::peek(): next byte or -1; sets eof at the end like the stream does
  int peek() {
    if (pos_ == len_ && !fill()) { eof_ = true; return -1; }
    return (unsigned char)buf_[pos_];
  }
  // std::getline: reads up to '\n' (dropped); eof is set when the data ends
  // before a newline is seen.  Returns false if nothing at all could be read.
  bool getline(std::string &out) {
    out.clear();
    bool any = false;
    for (;;) {
      if (pos_ == len_ && !fill()) { eof_ = true; return any; }
      any = true;
      const char *b = buf_.data() + pos_;
      const char *e = buf_.data() + len_;
      const char *nl = (const char *)memchr(b, '\n', (size_t)(e - b));
      if (nl) {
        out.append(b, (size_t)(nl - b));
        pos_ += (size_t)(nl - b) + 1;
        return true;
      }
      out.append(b, (size_t)(e - b));
      pos_ = len_;
    }
  }
  // raw read of exactly n bytes (binary dumps); returns bytes read
  size_t read(void *dst, size_t n) {
    size_t got = 0;
    char *d = (char *)dst;
    while (got < n) {
      if (pos_ == len_ && !fill()) { eof_ = true; break; }
      size_t take = std::min(n - got, len_ - pos_);
      memcpy(d + got, buf_.data() + pos_, take);
      pos_ += take;
      got += take;
    }
    return got;
  }
  // everything that is left
  void read_all(std::vector<uint8_t> &out) {
    for (;;) {
      if (pos_ == len_ && !fill()) { eof_ = true; return; }
      out.insert(out.end(), buf_.data() + pos_, buf_.data() + len_);
      pos_ = len_;
    }
  }

 private:
  bool fill() {
    int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
    if (n <= 0) { len_ = pos_ = 0; return false; }
    len_ = (size_t)n;
    pos_ = 0;
    return true;
  }
  gzFile f_ = nullptr;
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0;
  bool eof_ = false;
};

class GzWriter {
 public:
  explicit GzWriter(const std::string &path) {
    f_ = gzopen(path.c_str(), "wb1");
    if (!f_) throw std::runtime_error("cannot open '" + path + "' for writing");
    gzbuffer(f_, 1 << 20);
  }
  ~GzWriter() { close(); }
  GzWriter(const GzWriter &) = delete;
  GzWriter &operator=(const GzWriter &) = delete;
  void write(const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
      unsigned chunk = (unsigned)std::min<size_t>(n, 1u << 30);
      if (gzwrite(f_, c, chunk) != (int)chunk) throw std::runtime_error("gzwrite failed");
      c += chunk;
      n -= chunk;
    }
  }
  void write(const std::string &s) { write(s.data(), s.size()); }
  void close() {
    if (f_) { gzclose(f_); f_ = nullptr; }
  }

 private:
  gzFile f_ = nullptr;
};

// ---- raw DEFLATE of one buffer: libdeflate where the system has it (2-3 x zlib at level 1), else zlib ----
struct LibDeflate {
  void *(*alloc_c)(int) = nullptr;
  void (*free_c)(void *) = nullptr;
  size_t (*deflate)(void *, const void *, size_t, void *, size_t) = nullptr;
  size_t (*bound)(void *, size_t) = nullptr;
  void *(*alloc_d)() = nullptr;
  void (*free_d)(void *) = nullptr;
  int (*inflate)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;   // 0 = ok
  LibDeflate() {
    if (std::getenv("NIQKI_HOST_ZLIB_ONLY")) return;
    void *lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    alloc_c = (void *(*)(int))dlsym(lib, "libdeflate_alloc_compressor");
    free_c = (void (*)(void *))dlsym(lib, "libdeflate_free_compressor");
    deflate = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(lib, "libdeflate_deflate_compress");
    bound = (size_t(*)(void *, size_t))dlsym(lib, "libdeflate_deflate_compress_bound");
    alloc_d = (void *(*)())dlsym(lib, "libdeflate_alloc_decompressor");
    free_d = (void (*)(void *))dlsym(lib, "libdeflate_free_decompressor");
    inflate = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(lib, "libdeflate_deflate_decompress");
    if (!alloc_c || !free_c || !deflate || !bound || !alloc_d || !free_d || !inflate) alloc_c = nullptr;
  }
  bool usable() const { return alloc_c != nullptr; }
  static const LibDeflate &get() {
    static const LibDeflate l;
    return l;
  }
};

// A gzip member that says how long it is: FEXTRA subfield 'N' 'Q' = the member's whole size in bytes (RFC 1952 2.3.1.1;
// every inflater skips it -- zlib, and with it the reference's zstr reader, included), so a reader can find the members
// of a file without inflating them and inflate them side by side.  The payload is a raw DEFLATE stream at level 1 (the
// reference's writer's level, src/zstr.hpp:103).
constexpr size_t kTaggedHeader = 20;   // 10 fixed + XLEN + subfield header + the size
inline void put_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
inline uint32_t get_u32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

inline void tagged_member(const uint8_t *in, size_t n, std::vector<uint8_t> &out) {
  const LibDeflate &ld = LibDeflate::get();
  size_t body = 0;
  if (ld.usable()) {
    thread_local struct C { void *c = nullptr; ~C() { if (c) LibDeflate::get().free_c(c); } } comp;
    if (!comp.c && !(comp.c = ld.alloc_c(1))) throw std::runtime_error("libdeflate_alloc_compressor failed");
    out.resize(kTaggedHeader + ld.bound(comp.c, n) + 8);
    body = ld.deflate(comp.c, in, n, out.data() + kTaggedHeader, out.size() - kTaggedHeader - 8);
    if (!body) throw std::runtime_error("libdeflate_deflate_compress failed");
  } else {
    z_stream z{};
    if (deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    out.resize(kTaggedHeader + deflateBound(&z, (uLong)n) + 64 + 8);
    z.next_in = const_cast<Bytef *>(in);
    z.avail_in = (uInt)n;
    z.next_out = out.data() + kTaggedHeader;
    z.avail_out = (uInt)(out.size() - kTaggedHeader - 8);
    const int ret = deflate(&z, Z_FINISH);
    body = z.total_out;
    deflateEnd(&z);
    if (ret != Z_STREAM_END) throw std::runtime_error("deflate failed");
  }
  const size_t total = kTaggedHeader + body + 8;
  static const uint8_t head[16] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 4, 3, 8, 0, 'N', 'Q', 4, 0};
  std::memcpy(out.data(), head, 16);
  put_u32(out.data() + 16, (uint32_t)total);
  put_u32(out.data() + kTaggedHeader + body, (uint32_t)crc32(0L, in, (uInt)n));
  put_u32(out.data() + kTaggedHeader + body + 4, (uint32_t)n);
  out.resize(total);
}
// the size a tagged member at p announces (0: not one); avail = bytes of the file from p on
inline size_t tagged_size(const uint8_t *p, size_t avail) {
  if (avail < kTaggedHeader + 8 || p[0] != 0x1F || p[1] != 0x8B || p[2] != 8 || p[3] != 4 || p[10] != 8 || p[11] != 0 || p[12] != 'N' ||
      p[13] != 'Q' || p[14] != 4 || p[15] != 0)
    return 0;
  const size_t total = get_u32(p + 16);
  return total >= kTaggedHeader + 8 && total <= avail ? total : 0;
}
inline void tagged_inflate(const uint8_t *p, size_t total, std::vector<uint8_t> &out) {
  const size_t n = get_u32(p + total - 4), body = total - kTaggedHeader - 8;
  // ISIZE comes from the file: before it sizes a buffer it must be what DEFLATE can make of the member's body at all
  // (1032 : 1) and no more than a few of the writer's pieces -- a damaged trailer must not allocate gigabytes per member
  if (n > 1032 * body + 64 || n > (size_t(64) << 20)) throw std::runtime_error("gzip stream is damaged");
  out.resize(n);
  const LibDeflate &ld = LibDeflate::get();
  bool ok;
  if (ld.usable()) {
    thread_local struct D { void *d = nullptr; ~D() { if (d) LibDeflate::get().free_d(d); } } dec;
    if (!dec.d && !(dec.d = ld.alloc_d())) throw std::runtime_error("libdeflate_alloc_decompressor failed");
    size_t made = 0;
    ok = ld.inflate(dec.d, p + kTaggedHeader, body, out.data(), n, &made) == 0 && made == n;
  } else {
    z_stream z{};
    if (inflateInit2(&z, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
    z.next_in = const_cast<Bytef *>(p + kTaggedHeader);
    z.avail_in = (uInt)body;
    z.next_out = out.data();
    z.avail_out = (uInt)n;
    const int ret = inflate(&z, Z_FINISH);
    ok = ret == Z_STREAM_END && z.total_out == n && z.avail_in == 0;
    inflateEnd(&z);
  }
  if (!ok || (uint32_t)crc32(0L, out.data(), (uInt)n) != get_u32(p + total - 8)) throw std::runtime_error("gzip stream is damaged");
}

// Gzip writer that compresses in parallel: what it is given is cut into pieces of 8 MB, every piece becomes a gzip
// member of its own (tagged with its size, above), members are written in order.  Concatenated members are one valid
// gzip file; the reference's reader (zstr::istreambuf, src/zstr.hpp:236-239) restarts its inflater at every member
// end, so it reads these files unchanged (the CPU test suite lets the real reference load such a dump).
class ParallelGzWriter {
 public:
  static constexpr size_t kPiece = size_t(8) << 20;
  // flush_pieces: how many pieces pile up before they are compressed and written (0: four per thread)
  ParallelGzWriter(const std::string &path, unsigned threads, size_t flush_pieces = 0)
      : threads_(threads ? threads : 1), flush_pieces_(flush_pieces ? flush_pieces : 4 * (size_t)(threads ? threads : 1)) {
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd_ < 0) throw std::runtime_error("cannot open '" + path + "' for writing");
  }
  ~ParallelGzWriter() {
    if (busy_.valid()) busy_.wait();
    if (fd_ >= 0) ::close(fd_);
  }
  // a buffer the caller fills and hands over (not zeroed: a dump's blocks are gigabytes in all)
  struct Block {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    explicit Block(size_t bytes = 0) : p(bytes ? new uint8_t[bytes] : nullptr), n(bytes) {}
    uint8_t *data() { return p.get(); }
    const uint8_t *data() const { return p.get(); }
    size_t size() const { return n; }
  };
  void add(Block &&block) {
    if (!block.n) return;
    pieces_ += (block.size() + kPiece - 1) / kPiece;
    pending_.push_back(std::move(block));
    if (pieces_ >= flush_pieces_) flush();
  }
  void add(const std::vector<uint8_t> &bytes) {
    Block b(bytes.size());
    if (!bytes.empty()) std::memcpy(b.data(), bytes.data(), bytes.size());
    add(std::move(b));
  }
  // empty_member: a file that was given nothing still holds one (empty) gzip member, as zlib's writer leaves it
  void finish(bool empty_member = false) {
    flush();
    if (busy_.valid()) busy_.get();
    if (empty_member && offset_ == 0 && fd_ >= 0) {
      std::vector<uint8_t> m;
      const uint8_t none = 0;
      tagged_member(&none, 0, m);
      size_t done = 0;
      while (done < m.size()) {
        const ssize_t w = pwrite(fd_, m.data() + done, m.size() - done, (off_t)done);
        if (w <= 0) throw std::runtime_error("write failed");
        done += (size_t)w;
      }
      offset_ = m.size();
    }
    if (fd_ >= 0) {
      const int rc = ::close(fd_);
      fd_ = -1;
      if (rc != 0) throw std::runtime_error("write failed");
    }
  }

 private:
  // what has piled up is compressed and written by a task of its own, while the caller fetches the next blocks
  // (a dump's come from the device); one such task at a time, so the members stay in order
  void flush() {
    if (busy_.valid()) busy_.get();   // (rethrows what the last batch ran into)
    if (pending_.empty()) return;
    auto batch = std::make_shared<std::vector<Block>>(std::move(pending_));
    pending_.clear();
    pieces_ = 0;
    busy_ = std::async(std::launch::async, [this, batch] { write_batch(*batch); });
  }
  void write_batch(const std::vector<Block> &blocks) {
    struct Piece { const uint8_t *p; size_t n; };
    std::vector<Piece> jobs;
    for (const auto &b : blocks)
      for (size_t at = 0; at < b.size(); at += kPiece) jobs.push_back(Piece{b.data() + at, std::min(kPiece, b.size() - at)});
    std::vector<std::vector<uint8_t>> outs(jobs.size());
    std::vector<std::string> errs(threads_);
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    const unsigned nt = (unsigned)std::min<size_t>(threads_, jobs.size());
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&, t] {
        try {
          for (size_t i; (i = next.fetch_add(1)) < jobs.size();) tagged_member(jobs[i].p, jobs[i].n, outs[i]);
        } catch (const std::exception &e) { errs[t] = e.what(); }
      });
    for (auto &t : pool) t.join();
    for (const auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
    // every member's place in the file is known now: the same threads write them side by side
    std::vector<uint64_t> at(outs.size() + 1, offset_);
    for (size_t i = 0; i < outs.size(); ++i) at[i + 1] = at[i] + outs[i].size();
    pool.clear();
    next = 0;
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&, t] {
        for (size_t i; (i = next.fetch_add(1)) < outs.size();) {
          size_t done = 0;
          while (done < outs[i].size()) {
            const ssize_t w = pwrite(fd_, outs[i].data() + done, outs[i].size() - done, (off_t)(at[i] + done));
            if (w <= 0) { errs[t] = "write failed"; return; }
            done += (size_t)w;
          }
        }
      });
    for (auto &t : pool) t.join();
    for (const auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
    offset_ = at[outs.size()];
  }
  uint64_t offset_ = 0;   // bytes of the file written so far
  std::future<void> busy_;
  int fd_ = -1;
  unsigned threads_;
  size_t flush_pieces_;
  size_t pieces_ = 0;
  std::vector<Block> pending_;
};

// Text output (hit lines, matrix rows) with GzWriter's interface on top of ParallelGzWriter: the text is collected in
// blocks of 8 MB, which become gzip members compressed side by side while the caller formats the next ones (one zlib
// stream at level 1 takes a quarter of a second per 100 MB of hit lines: in lines mode the writer thread was what the
// GPU waited for).  A concatenation of members is one gzip file to every reader.
class ParallelTextWriter {
 public:
  ParallelTextWriter(const std::string &path, unsigned threads) : out_(path, threads, threads ? threads : 1) {}
  ~ParallelTextWriter() {
    try { close(); } catch (...) {}
  }
  ParallelTextWriter(const ParallelTextWriter &) = delete;
  ParallelTextWriter &operator=(const ParallelTextWriter &) = delete;
  void write(const void *p, size_t n) {
    const uint8_t *c = (const uint8_t *)p;
    while (n) {
      if (!cur_.p) { cur_ = ParallelGzWriter::Block(ParallelGzWriter::kPiece); fill_ = 0; }
      const size_t take = std::min(n, ParallelGzWriter::kPiece - fill_);
      std::memcpy(cur_.data() + fill_, c, take);
      fill_ += take;
      c += take;
      n -= take;
      if (fill_ == ParallelGzWriter::kPiece) hand_over();
    }
  }
  void write(const std::string &s) { write(s.data(), s.size()); }
  void close() {
    if (closed_) return;
    closed_ = true;
    if (fill_) hand_over();
    out_.finish(true);
  }

 private:
  void hand_over() {
    cur_.n = fill_;
    out_.add(std::move(cur_));
    cur_ = ParallelGzWriter::Block();
    fill_ = 0;
  }
  ParallelGzWriter out_;
  ParallelGzWriter::Block cur_;
  size_t fill_ = 0;
  bool closed_ = false;
};

// Reader of such a file: the members are found by their tags and inflated by a pool of threads, a window ahead of
// the consumer, which takes their bytes in file order.  probe(): does the file start with a tagged member?
class TaggedGzReader {
 public:
  static bool probe(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint8_t h[kTaggedHeader + 8] = {0};
    const size_t got = fread(h, 1, sizeof h, f);
    fclose(f);
    // (the announced size is checked against the file when the members are walked)
    return got == sizeof h && h[0] == 0x1F && h[1] == 0x8B && h[2] == 8 && h[3] == 4 && h[10] == 8 && h[11] == 0 && h[12] == 'N' &&
           h[13] == 'Q' && h[14] == 4 && h[15] == 0;
  }
  TaggedGzReader(const std::string &path, unsigned threads) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open '" + path + "'");
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size <= 0) { ::close(fd_); throw std::runtime_error("cannot read '" + path + "'"); }
    size_ = (size_t)st.st_size;
    map_ = (const uint8_t *)mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (map_ == MAP_FAILED) { ::close(fd_); throw std::runtime_error("cannot map '" + path + "'"); }
    for (size_t at = 0; at < size_;) {
      const size_t total = tagged_size(map_ + at, size_ - at);
      if (!total) { unmap(); throw std::runtime_error("'" + path + "': gzip stream is damaged or truncated"); }
      members_.push_back({at, total});
      at += total;
    }
    out_.resize(members_.size());
    state_.assign(members_.size(), 0);
    window_ = std::max<size_t>(4 * (size_t)std::max(threads, 1u), 8);
    threads = (unsigned)std::min<size_t>(std::max(threads, 1u), members_.size());
    for (unsigned t = 0; t < threads; ++t) pool_.emplace_back([this] { work(); });
  }
  ~TaggedGzReader() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : pool_) t.join();
    unmap();
  }
  TaggedGzReader(const TaggedGzReader &) = delete;
  TaggedGzReader &operator=(const TaggedGzReader &) = delete;
  // the next member's bytes; false behind the last one
  bool next(std::vector<uint8_t> &raw) {
    std::unique_lock<std::mutex> g(mu_);
    if (taken_ >= members_.size()) return false;
    cv_.wait(g, [&] { return state_[taken_] != 0; });
    if (state_[taken_] == 2) throw std::runtime_error(err_);
    raw = std::move(out_[taken_]);
    ++taken_;
    g.unlock();
    cv_.notify_all();
    return true;
  }

 private:
  void unmap() {
    if (map_ && map_ != MAP_FAILED) munmap((void *)map_, size_);
    map_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }
  void work() {
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_.wait(g, [&] { return stop_ || issued_ >= members_.size() || issued_ < taken_ + window_; });
        if (stop_ || issued_ >= members_.size()) return;
        i = issued_++;
      }
      std::vector<uint8_t> raw;
      std::string err;
      try { tagged_inflate(map_ + members_[i].at, members_[i].total, raw); } catch (const std::exception &e) { err = e.what(); }
      {
        std::lock_guard<std::mutex> g(mu_);
        if (err.empty()) { out_[i] = std::move(raw); state_[i] = 1; } else { err_ = err; state_[i] = 2; }
      }
      cv_.notify_all();
    }
  }
  struct Member { size_t at, total; };
  int fd_ = -1;
  const uint8_t *map_ = nullptr;
  size_t size_ = 0, window_ = 8;
  std::vector<Member> members_;
  std::vector<std::vector<uint8_t>> out_;
  std::vector<uint8_t> state_;   // 0 pending, 1 ready, 2 failed
  std::vector<std::thread> pool_;
  std::mutex mu_;
  std::condition_variable cv_;
  size_t issued_ = 0, taken_ = 0;
  bool stop_ = false;
  std::string err_;
};

}  // namespace nqhost
