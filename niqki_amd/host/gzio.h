// gzio.h -- line-oriented gzip/plain input and gzip(level 1) output on zlib.
// Own replacement for the reference's vendored zstr streams (src/zstr.hpp):
// input auto-detects gzip vs plain like zstr::ifstream (:190-203), output is
// gzip level 1, window 15+16, like zstr::ofstream (:103,:458).
#pragma once
#include <zlib.h>

#include <cstdint>
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

// Gzip writer that compresses blocks in parallel: every block becomes its own
// gzip member (level 1, like the reference's writer), members are written in
// submission order.  Concatenated members are one valid gzip file; the
// reference's reader (zstr::istreambuf, src/zstr.hpp:236-239) restarts its
// inflater at every member end, so it reads these files unchanged.
class ParallelGzWriter {
 public:
  ParallelGzWriter(const std::string &path, unsigned threads) : threads_(threads ? threads : 1) {
    f_ = fopen(path.c_str(), "wb");
    if (!f_) throw std::runtime_error("cannot open '" + path + "' for writing");
  }
  ~ParallelGzWriter() { if (f_) fclose(f_); }
  void add(std::vector<uint8_t> &&block) {
    if (block.empty()) return;
    pending_.push_back(std::move(block));
    if (pending_.size() >= threads_) flush();
  }
  void finish() {
    flush();
    if (f_) { fclose(f_); f_ = nullptr; }
  }

 private:
  static void deflate_member(const std::vector<uint8_t> &in, std::vector<uint8_t> &out) {
    z_stream z{};
    if (deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    out.resize(deflateBound(&z, (uLong)in.size()) + 64);
    size_t ipos = 0, opos = 0;
    int ret = Z_OK;
    while (ret != Z_STREAM_END) {  // avail_in/out are 32-bit: feed in slices
      const size_t ichunk = std::min<size_t>(in.size() - ipos, 1u << 30);
      z.next_in = const_cast<Bytef *>(in.data() + ipos);
      z.avail_in = (uInt)ichunk;
      const size_t ochunk = std::min<size_t>(out.size() - opos, 1u << 30);
      z.next_out = out.data() + opos;
      z.avail_out = (uInt)ochunk;
      ret = deflate(&z, ipos + ichunk == in.size() ? Z_FINISH : Z_NO_FLUSH);
      if (ret != Z_OK && ret != Z_STREAM_END && ret != Z_BUF_ERROR) { deflateEnd(&z); throw std::runtime_error("deflate failed"); }
      ipos += ichunk - z.avail_in;
      opos += ochunk - z.avail_out;
    }
    deflateEnd(&z);
    out.resize(opos);
  }
  void flush() {
    if (pending_.empty()) return;
    std::vector<std::vector<uint8_t>> outs(pending_.size());
    std::vector<std::thread> pool;
    std::vector<std::string> errs(pending_.size());
    for (size_t i = 0; i < pending_.size(); ++i)
      pool.emplace_back([&, i] {
        try { deflate_member(pending_[i], outs[i]); } catch (const std::exception &e) { errs[i] = e.what(); }
      });
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < outs.size(); ++i) {
      if (!errs[i].empty()) throw std::runtime_error(errs[i]);
      if (fwrite(outs[i].data(), 1, outs[i].size(), f_) != outs[i].size()) throw std::runtime_error("write failed");
    }
    pending_.clear();
  }
  FILE *f_ = nullptr;
  unsigned threads_;
  std::vector<std::vector<uint8_t>> pending_;
};

}  // namespace nqhost
