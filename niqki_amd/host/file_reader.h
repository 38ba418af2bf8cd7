// file_reader.h -- the host program's file bytes: page-locked buffers, whole files read (packed, gzip'd as they are, or
// inflated) by a pool of reader threads that hands them out in list order.  Part of index_host.cpp's translation unit
// (the `niqki` program's file drivers); no device code.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/niqki_hip.h"
#include "../csrc/nq_pack.h"
#include "seqio.h"

namespace nqhost {
namespace {

// ---- raw file bytes in page-locked memory --------------------------------------
// The GPU frames the records (niqki_stage_raw), so the host only moves bytes:
// files are read (and gunzipped) straight into page-locked buffers that the
// library copies from by DMA.
struct PinnedBuf {
  uint8_t *p = nullptr;
  size_t cap = 0, size = 0;
  bool locked = false;  // page-locked (else plain memory: the copies still work, slower)
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { release(); }
  void release() {
    if (locked) niqki_host_free(p); else std::free(p);
    p = nullptr;
  }
  void reserve(size_t n) {  // keeps the first `size` bytes
    if (n <= cap) return;
    const size_t want = std::max(n, cap + cap / 2);
    bool q_locked = true;
    uint8_t *q = (uint8_t *)niqki_host_alloc(want);
    if (!q) {  // e.g. a low locked-memory limit
      q = (uint8_t *)std::malloc(want);
      q_locked = false;
      if (!q) throw std::runtime_error("allocation of " + std::to_string(want) + " bytes failed");
    }
    if (size) std::memcpy(q, p, size);
    release();
    p = q;
    locked = q_locked;
    cap = want;
  }
};

// libdeflate (its whole-buffer gzip decoder runs 2-3 x zlib's inflate) when the system has the library;
// loaded once, by name -- no header is needed for the three entry points used.
struct FastInflate {
  void *(*alloc)() = nullptr;
  void (*release)(void *) = nullptr;
  // 0 = ok, 1 = bad data, 2 = short output, 3 = insufficient space
  int (*gzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
  FastInflate() {
    if (std::getenv("NIQKI_HOST_ZLIB_ONLY")) return;
    void *lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    alloc = (void *(*)())dlsym(lib, "libdeflate_alloc_decompressor");
    release = (void (*)(void *))dlsym(lib, "libdeflate_free_decompressor");
    gzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(lib, "libdeflate_gzip_decompress_ex");
    if (!alloc || !release || !gzip_ex) alloc = nullptr;
  }
  bool usable() const { return alloc != nullptr; }
};
const FastInflate &fast_inflate() {
  static const FastInflate f;
  return f;
}

// A whole regular gzip file (all its members) through libdeflate.  false = leave it to zlib (no library, not a
// plain well-formed file: zlib then decides what a damaged or truncated stream yields, as before).
bool gunzip_whole(int fd, size_t file_bytes, PinnedBuf &out) {
  const FastInflate &fi = fast_inflate();
  if (!fi.usable() || file_bytes < 18 || file_bytes > (size_t(1) << 31)) return false;
  thread_local std::vector<uint8_t> in;
  thread_local struct Dec {
    void *d = nullptr;
    ~Dec() { if (d) fast_inflate().release(d); }
  } dec;
  if (!dec.d && !(dec.d = fi.alloc())) return false;
  in.resize(file_bytes);
  size_t got = 0;
  while (got < file_bytes) {
    const ssize_t n = pread(fd, in.data() + got, file_bytes - got, (off_t)got);
    if (n <= 0) return false;
    got += (size_t)n;
  }
  // the last member's size (mod 2^32) closes the file: exact for the usual one-member file
  const uint32_t isize = (uint32_t)in[file_bytes - 4] | (uint32_t)in[file_bytes - 3] << 8 | (uint32_t)in[file_bytes - 2] << 16 |
                         (uint32_t)in[file_bytes - 1] << 24;
  out.size = 0;
  out.reserve(std::max<size_t>((size_t)isize + 64, size_t(1) << 20));
  size_t at = 0;
  while (at + 18 <= file_bytes && in[at] == 0x1F && in[at + 1] == 0x8B) {
    size_t used = 0, made = 0;
    const int r = fi.gzip_ex(dec.d, in.data() + at, file_bytes - at, out.p + out.size, out.cap - out.size, &used, &made);
    if (r == 3) {   // the member needs more room: again from its start
      out.reserve(std::max(out.cap * 2, out.size + 4 * (file_bytes - at)));
      continue;
    }
    if (r != 0 || used == 0) return false;
    at += used;
    out.size += made;
  }
  return at > 0;   // (bytes behind the last member are ignored, as zlib's gzread does)
}

// Whole content of a file, gunzipped when it starts with the gzip magic (the
// reference's zstr::ifstream auto-detects the same way, src/zstr.hpp:190-203).
// want_pack: a plain (not gzipped) regular FASTA file may be handed over as its packed container
// (nq_pack.h, what niqki_pack_fasta makes: 2 bits per base in full A/C/G/T lines, everything else verbatim; the device
// restores the file's exact bytes) -- *packed says whether it was.
// want_gz (2: any, 1: only files whose members carry their size -- BGZF, this project's tag: those the library cuts into
// members, a wavefront each, quick however few files a batch holds): a gzip'd regular file may be handed over as it lies on disk (*gz says whether it was): the device inflates it
// (niqki_stage_raw, NIQKI_FILE_GZIP).  Only what the device will plausibly take -- one member whose trailer states a
// size in keeping with the file's (the library's own test) -- everything else is inflated here, as before.
void read_file_bytes(const std::string &path, PinnedBuf &out, bool want_pack = false, bool *packed = nullptr,
                     int want_gz = 0, bool *gz = nullptr) {
  out.size = 0;
  if (packed) *packed = false;
  if (gz) *gz = false;
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) throw std::runtime_error("cannot open '" + path + "'");
  struct stat st;
  if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("cannot stat '" + path + "'"); }
  unsigned char magic[2] = {0, 0};
  const ssize_t m = pread(fd, magic, 2, 0);
  if (want_pack && packed && S_ISREG(st.st_mode) && st.st_size >= 4096 && !(m == 2 && magic[0] == 0x1F && magic[1] == 0x8B)) {
    // packed while it is read: pieces of 256 KB go through the thread's cache (read() copies them there, the packer
    // reads them from there), a quarter of the bytes is written out -- and later page-locked and sent
    const size_t n = (size_t)st.st_size;
    thread_local std::vector<uint8_t> piece, box;
    const size_t bound = nqp::pack_bound(n);
    if (box.size() < bound) box.resize(bound);
    if (piece.size() < (size_t(256) << 10)) piece.resize(size_t(256) << 10);
    nqp::Packer pk;
    pk.begin(box.data(), box.size(), n);
    size_t have = 0, done = 0;   // bytes in `piece` not yet taken / bytes of the file read so far
    bool io_ok = true;
    while (pk.ok && done < n) {
      if (have == piece.size()) piece.resize(piece.size() * 2);   // one line longer than the piece (an unwrapped genome)
      const ssize_t r = pread(fd, piece.data() + have, std::min(piece.size() - have, n - done), (off_t)done);
      if (r <= 0) { io_ok = false; break; }
      done += (size_t)r;
      have += (size_t)r;
      const size_t used = pk.feed(piece.data(), have, done == n);
      have -= used;
      if (have) std::memmove(piece.data(), piece.data() + used, have);
    }
    const size_t got = io_ok ? pk.finish() : 0;
    if (got) {
      out.reserve(got + 64);
      std::memcpy(out.p, box.data(), got);
      out.size = got;
      *packed = true;
      ::close(fd);
      return;
    }
    // not worth packing (or the file changed under us): its bytes as they are, below
  }
  if (m == 2 && magic[0] == 0x1F && magic[1] == 0x8B) {
    bool tagged_head = false;   // the first member says how long it is
    {
      unsigned char h[64];
      const ssize_t hn = pread(fd, h, sizeof h, 0);
      if (hn >= 28 && (h[3] & 4)) {
        const size_t xlen = (size_t)h[10] | (size_t)h[11] << 8;
        for (size_t x = 12; x + 4 <= 12 + xlen && x + 4 <= (size_t)hn; x += 4 + ((size_t)h[x + 2] | (size_t)h[x + 3] << 8))
          tagged_head |= (h[x] == 'B' && h[x + 1] == 'C') || (h[x] == 'N' && h[x + 1] == 'Q');
      }
    }
    if ((want_gz == 2 || (want_gz == 1 && tagged_head)) && gz && S_ISREG(st.st_mode) && st.st_size >= 18 && (uint64_t)st.st_size <= 0x7FFF0000ull) {
      const size_t n = (size_t)st.st_size;
      out.reserve(n + 64);
      size_t got = 0;
      while (got < n) {
        const ssize_t r = pread(fd, out.p + got, n - got, (off_t)got);
        if (r <= 0) break;
        got += (size_t)r;
      }
      if (got == n) {
        const uint64_t isize = (uint64_t)out.p[n - 4] | (uint64_t)out.p[n - 3] << 8 | (uint64_t)out.p[n - 2] << 16 | (uint64_t)out.p[n - 1] << 24;
        // ... or a file of members that say how long they are: the library cuts it into its members
        if (tagged_head || (isize <= 0x7FFF0000ull && isize <= (uint64_t)n * 64u && isize * 4096u >= (uint64_t)n)) {
          out.size = n;
          *gz = true;
          ::close(fd);
          return;
        }
      }
      out.size = 0;   // (several members, a huge or an empty file, a short read: inflated here)
    }
    if (S_ISREG(st.st_mode) && gunzip_whole(fd, (size_t)st.st_size, out)) { ::close(fd); return; }
    out.size = 0;
    gzFile g = gzdopen(fd, "rb");  // owns fd from here
    if (!g) { ::close(fd); throw std::runtime_error("cannot open '" + path + "'"); }
    gzbuffer(g, 1 << 20);
    out.reserve(std::max<size_t>((size_t)st.st_size * 4, size_t(1) << 20));
    for (;;) {
      if (out.size == out.cap) out.reserve(out.cap * 2);
      const int n = gzread(g, out.p + out.size, (unsigned)std::min<size_t>(out.cap - out.size, 1u << 30));
      if (n < 0) { gzclose(g); throw std::runtime_error("'" + path + "': gzip stream is damaged"); }
      if (n == 0) break;
      out.size += (size_t)n;
    }
    // a stream that ends inside a member gives what it has and then 0: zlib only says so through gzerror
    // (the reference's zstr reader throws on such a file, src/zstr.hpp)
    int zerr = Z_OK;
    (void)gzerror(g, &zerr);
    gzclose(g);
    if (zerr != Z_OK && zerr != Z_STREAM_END) throw std::runtime_error("'" + path + "': gzip stream is damaged or truncated");
    return;
  }
  out.reserve(std::max<size_t>((size_t)st.st_size, 64));
  for (;;) {  // regular files give st_size bytes; pipes and growing files are read to their end
    if (out.size == out.cap) out.reserve(out.cap * 2);
    const ssize_t n = ::read(fd, out.p + out.size, std::min<size_t>(out.cap - out.size, size_t(1) << 30));
    if (n < 0) { ::close(fd); throw std::runtime_error("cannot read '" + path + "'"); }
    if (n == 0) break;
    out.size += (size_t)n;
    if (out.size == (size_t)st.st_size && S_ISREG(st.st_mode)) break;
  }
  ::close(fd);
}

// file-reader threads: NIQKI_HOST_THREADS, else OMP_NUM_THREADS (what sizes the reference's
// reader pool, its files being read inside an OpenMP region), else min(hardware threads, 64)
unsigned host_threads() {
  for (const char *name : {"NIQKI_HOST_THREADS", "OMP_NUM_THREADS"}) {
    if (const char *v = std::getenv(name)) {
      const int n = std::atoi(v);
      if (n > 0) return (unsigned)std::min(n, 256);
    }
  }
  unsigned hw = std::thread::hardware_concurrency();
  hw = hw ? std::min(hw, 64u) : 4u;
  // A container's CPU quota (cgroup v2 cpu.max: "<quota> <period>" in microseconds, or "max"): threads beyond it only
  // get the whole group throttled -- the main thread, which feeds the GPU, with them.  One CPU is left to that thread.
  if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    unsigned long period = 0;
    if (std::fscanf(f, "%31s %lu", q, &period) == 2 && period && std::strcmp(q, "max") != 0) {
      const unsigned long cpus = std::strtoul(q, nullptr, 10) / period;
      if (cpus >= 1) hw = (unsigned)std::min<unsigned long>(hw, std::max<unsigned long>(cpus > 2 ? cpus - 1 : cpus, 2));
    }
    std::fclose(f);
  }
  return hw;
}

// Reads the files of a list on several threads -- the reference does this part in
// its OpenMP region, one file per thread -- and hands them out strictly in list
// order, so genome ids and output order are those of a single-threaded run.  A
// reader takes a buffer BEFORE it takes the next file index, so the oldest
// outstanding file always owns one and the consumer can never starve.
class OrderedFileReader {
 public:
  struct File {
    PinnedBuf buf;
    std::string err;
    bool packed = false;   // buf holds the file's packed container (niqki_pack_fasta), not its bytes
    bool gz = false;       // buf holds the gzip file as it lies on disk: the device inflates it (NIQKI_FILE_GZIP)
  };
  // The page-locked buffers are shared by all readers of the process (index phase, then
  // query phase) and never freed: locking and unlocking 1.5 GB of pages costs more than
  // reading the files, and the process ends right after its last phase.
  static std::deque<File> &pool(size_t n) {
    static std::deque<File> *p = new std::deque<File>();
    while (p->size() < n) p->emplace_back();
    return *p;
  }
  OrderedFileReader(const std::vector<std::string> &paths, unsigned threads, size_t n_bufs, int device_inflate)
      : paths_(paths), bufs_(pool(n_bufs)), ready_(paths.size(), nullptr), pack_(std::getenv("NIQKI_HOST_NO_PACK") == nullptr),
        gz_(device_inflate) {
    for (size_t i = 0; i < n_bufs; ++i) free_.push_back(&bufs_[n_bufs - 1 - i]);  // LIFO: low indices first
    threads = (unsigned)std::min<size_t>(threads, std::max<size_t>(paths.size(), 1));
    for (unsigned t = 0; t < threads; ++t) pool_.emplace_back([this] { work(); });
  }
  ~OrderedFileReader() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_free_.notify_all();
    for (auto &t : pool_) t.join();
  }
  // next file of the list, or nullptr after the last one; give it back with release()
  File *next() {
    std::unique_lock<std::mutex> g(mu_);
    if (taken_ >= paths_.size()) return nullptr;
    cv_ready_.wait(g, [&] { return ready_[taken_] != nullptr; });
    File *f = ready_[taken_++];
    if (!f->err.empty()) throw std::runtime_error(f->err);
    return f;
  }
  void release(File *f) {
    {
      std::lock_guard<std::mutex> g(mu_);
      free_.push_back(f);
    }
    cv_free_.notify_one();
  }

 private:
  void work() {
    for (;;) {
      File *f;
      size_t idx;
      {
        std::unique_lock<std::mutex> g(mu_);
        cv_free_.wait(g, [&] { return stop_ || issued_ >= paths_.size() || !free_.empty(); });
        if (stop_ || issued_ >= paths_.size()) return;
        f = free_.back();
        free_.pop_back();
        idx = issued_++;
      }
      f->err.clear();
      f->packed = false;
      f->gz = false;
      try {
        read_file_bytes(paths_[idx], f->buf, pack_ && data_type(paths_[idx]) == 'A', &f->packed, gz_, &f->gz);
      } catch (const std::exception &e) { f->err = e.what(); f->buf.size = 0; f->packed = false; f->gz = false; }
      {
        std::lock_guard<std::mutex> g(mu_);
        ready_[idx] = f;
      }
      cv_ready_.notify_all();
    }
  }
  const std::vector<std::string> &paths_;
  std::deque<File> &bufs_;
  std::vector<File *> free_, ready_;
  std::vector<std::thread> pool_;
  std::mutex mu_;
  std::condition_variable cv_free_, cv_ready_;
  size_t issued_ = 0, taken_ = 0;
  bool stop_ = false;
  const bool pack_;   // plain FASTA files travel as packed containers (NIQKI_HOST_NO_PACK: as their bytes)
  const int gz_;      // which gzip files travel as they are and are inflated on the device (read_file_bytes' want_gz; for_each_batch decides per list)
};

}  // namespace
}  // namespace nqhost
