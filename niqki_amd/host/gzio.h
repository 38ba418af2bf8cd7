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

}  // namespace nqhost
