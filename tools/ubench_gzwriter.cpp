// Rate of the dump writer's codec on this host: 8 MB pieces of dump-like words (bucket sizes and ascending ids) through
// tagged_member / tagged_inflate of niqki_amd/host/gzio.h, on 1 and on N threads.
//   g++ -O2 -std=c++17 -pthread tools/ubench_gzwriter.cpp -o /tmp/ubench_gzwriter -lz -ldl && /tmp/ubench_gzwriter 16
#include "../niqki_amd/host/gzio.h"

#include <chrono>
#include <cstdio>
#include <random>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const unsigned nt = argc > 1 ? (unsigned)atoi(argv[1]) : 16;
  const size_t piece = size_t(8) << 20, n_pieces = 64;
  std::vector<std::vector<uint8_t>> in(n_pieces, std::vector<uint8_t>(piece));
  std::mt19937 rng(1);
  for (auto &b : in) {   // buckets of ~25 ascending 17-bit ids behind their size word
    uint32_t *w = (uint32_t *)b.data();
    for (size_t i = 0; i < piece / 4;) {
      const uint32_t len = rng() % 50;
      w[i++] = len;
      uint32_t id = rng() % 4000;
      for (uint32_t k = 0; k < len && i < piece / 4; ++k) { w[i++] = id; id += 1 + rng() % 4000; }
    }
  }
  printf("libdeflate: %s\n", nqhost::LibDeflate::get().usable() ? "yes" : "no (zlib)");
  for (unsigned threads : {1u, nt}) {
    std::vector<std::vector<uint8_t>> out(n_pieces), back(n_pieces);
    std::atomic<size_t> next{0};
    double t0 = now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
      pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n_pieces;) nqhost::tagged_member(in[i].data(), piece, out[i]); });
    for (auto &t : pool) t.join();
    double t1 = now();
    size_t csize = 0;
    for (auto &o : out) csize += o.size();
    pool.clear();
    next = 0;
    for (unsigned t = 0; t < threads; ++t)
      pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n_pieces;) nqhost::tagged_inflate(out[i].data(), out[i].size(), back[i]); });
    for (auto &t : pool) t.join();
    double t2 = now();
    printf("%2u threads: deflate %.2f GB/s (ratio %.3f), inflate %.2f GB/s of raw bytes\n", threads, n_pieces * piece / (t1 - t0) / 1e9,
           (double)csize / (n_pieces * piece), n_pieces * piece / (t2 - t1) / 1e9);
  }
  return 0;
}
