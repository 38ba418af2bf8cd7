// The host program's parallel gzip writer and reader (niqki_amd/host/gzio.h: size-tagged members, written and inflated
// side by side) on their own, for ThreadSanitizer / AddressSanitizer (tests/test_host_sanitizers.py):
//   * a stream of blocks of every size around the 8 MB piece -> a file of tagged members -> the same bytes back through
//     TaggedGzReader (4 threads) and through zlib's gzread (what the reference's reader does), twice with another
//     number of threads;
//   * a damaged member, a cut file, a member whose tag lies: an exception, never a wrong byte;
//   * with NIQKI_HOST_ZLIB_ONLY the same through zlib's codec.
// Test infrastructure; prints "ok" and exits 0.
#include "../../niqki_amd/host/gzio.h"

#include <cstdio>
#include <random>

using namespace nqhost;

static std::vector<uint8_t> read_all(const std::string &path) {
  std::vector<uint8_t> v;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return v;
  uint8_t buf[1 << 16];
  for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}
static void write_all(const std::string &path, const std::vector<uint8_t> &v) {
  FILE *f = fopen(path.c_str(), "wb");
  fwrite(v.data(), 1, v.size(), f);
  fclose(f);
}
static bool tagged_read(const std::string &path, unsigned threads, std::vector<uint8_t> &out, std::string &err) {
  out.clear();
  try {
    if (!TaggedGzReader::probe(path)) { err = "not tagged"; return false; }
    TaggedGzReader r(path, threads);
    std::vector<uint8_t> piece;
    while (r.next(piece)) out.insert(out.end(), piece.begin(), piece.end());
    return true;
  } catch (const std::exception &e) { err = e.what(); return false; }
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const bool small = argc > 2 && std::string(argv[2]) == "small";   // fewer bytes (the run with zlib's slower codec)
  const std::string path = dir + "/gzio_stress.gz";
  std::mt19937 rng(7);
  std::vector<uint8_t> all;
  for (unsigned threads : {3u, 7u}) {
    if (small && threads == 7u) break;
    all.clear();
    {
      ParallelGzWriter w(path, threads);
      const size_t sizes[] = {24, 1, ParallelGzWriter::kPiece - 1, ParallelGzWriter::kPiece, ParallelGzWriter::kPiece + 1, 3 * ParallelGzWriter::kPiece + 12345, 0, 700000, 5};
      for (int rep = 0; rep < (small ? 1 : 2); ++rep)
        for (size_t n : sizes) {
          if (small && n > ParallelGzWriter::kPiece + 1) n = 2 * ParallelGzWriter::kPiece + 77;
          ParallelGzWriter::Block b(n);
          uint32_t x = rng();
          for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; b.data()[i] = (uint8_t)((x >> 24) & (rep ? 0x0F : 0xFF)); }
          all.insert(all.end(), b.data(), b.data() + n);
          w.add(std::move(b));
        }
      std::vector<uint8_t> names(1000, 'n');
      all.insert(all.end(), names.begin(), names.end());
      w.add(names);
      w.finish();
    }
    std::vector<uint8_t> back;
    std::string err;
    if (!tagged_read(path, threads + 1, back, err) || back != all) { fprintf(stderr, "tagged read-back differs: %s\n", err.c_str()); return 1; }
    {   // zlib's reader sees one gzip file
      GzReader g(path);
      std::vector<uint8_t> z;
      g.read_all(z);
      if (z != all) { fprintf(stderr, "zlib read-back differs\n"); return 1; }
    }
  }
  // damage: a flipped byte inside a member's payload, a cut file, a tag that overstates its member, a trailer that
  // announces gigabytes (must be refused before anything of that size is allocated)
  const std::vector<uint8_t> good = read_all(path);
  for (int what = 0; what < 4; ++what) {
    std::vector<uint8_t> bad = good;
    if (what == 0) bad[bad.size() / 2] ^= 0x40;
    if (what == 1) bad.resize(bad.size() - 100);
    if (what == 2) bad[16] ^= 0x10;
    if (what == 3) { bad[bad.size() - 1] = 0xF0; bad[bad.size() - 2] = 0xFF; }   // ISIZE of the last member: ~4 GB
    write_all(path, bad);
    std::vector<uint8_t> back;
    std::string err;
    if (tagged_read(path, 4, back, err)) { fprintf(stderr, "damage %d went unnoticed\n", what); return 1; }
  }
  remove(path.c_str());
  puts("ok");
  return 0;
}
