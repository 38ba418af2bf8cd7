// ubench_sketch.hip -- the sketch kernel alone, without Python: N random ACGT genomes of L bases in HBM,
// launch_sketch timed with HIP events; the filtered result is compared with the unfiltered (exact) pass.
// Includes the product source directly, so -D switches of nq_sketch.hip can be tried per binary:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/ubench_sketch.hip -o tools/bin/ubench_sketch
//   tools/bin/ubench_sketch [n_genomes=1024] [len=5000000] [reps=3] [S=15]
#define NQ_SKETCH_CLOCK 1
#include "../niqki_amd/csrc/nq_sketch.hip"

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void fill_acgt(uint8_t *p, uint64_t n, uint64_t seed) {
  uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 32;
  if (i >= n) return;
  uint64_t z = (i / 32 + seed) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  for (int j = 0; j < 32 && i + j < n; ++j) p[i + j] = (uint8_t)((0x54474341u >> (8 * ((z >> (2 * j)) & 3))) & 0xFFu);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 1024;
  const uint64_t L = argc > 2 ? atoll(argv[2]) : 5000000;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  const uint32_t S = argc > 4 ? atoi(argv[4]) : 15;
  nq::Derived d{};
  d.K = 31; d.S = S; d.W = 12; d.H = 4; d.M = 8; d.F = 1u << S; d.R = 1u << 12; d.mask_m = 255; d.max_rem = 15;
  d.min_score = 0; d.slot_begin = 0; d.slot_end = d.F; d.kmer_mask = (1ULL << 62) - 1;
  uint8_t *seq; uint64_t *ro; int32_t *sk, *sk2;
  CK(hipMalloc(&seq, n * L + 4096));
  CK(hipMalloc(&ro, (n + 1) * 8));
  CK(hipMalloc(&sk, (size_t)n * d.F * 4));
  CK(hipMalloc(&sk2, (size_t)n * d.F * 4));
  hipLaunchKernelGGL(fill_acgt, dim3((uint32_t)((n * L / 32 + 255) / 256 + 1)), dim3(256), 0, 0, seq, n * L + 4096, 12345ull);
  std::vector<uint64_t> h(n + 1);
  for (uint32_t i = 0; i <= n; ++i) h[i] = i * L;
  CK(hipMemcpy(ro, h.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  nq::SketchArgs a{};
  a.d = d; a.seqs = seq; a.rec_off = ro; a.entry_rec = nullptr; a.sketches = sk; a.splits = 1; a.halves = 1;
  a.accumulate = 0; a.densify = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(nq::launch_sketch(a, n, L, 0));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) CK(nq::launch_sketch(a, n, L, 0));
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double kmers = (double)reps * n * (L - 31);
  printf("sketch: %u x %llu bp, %.3f ms per launch, %.1f G k-mers/s\n", n, (unsigned long long)L, ms / reps, kmers / (ms * 1e-3) / 1e9);
  {
    unsigned long long clk[2];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(nq::nq_sketch_clk), 16));
    printf("workgroup 0: %llu shader cycles in %.1f us = %.3f GHz\n", clk[0], clk[1] * 0.01, clk[0] / (clk[1] * 10.0));
  }
  if (getenv("SK_TRACE")) {
    std::vector<unsigned long long> tr(3 * 8192);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(nq::nq_sketch_trace), tr.size() * 8));
    unsigned long long t0 = ~0ull, t1 = 0;
    const uint32_t nb = n < 8192 ? n : 8192;
    for (uint32_t b = 0; b < nb; ++b) { if (tr[3 * b] < t0) t0 = tr[3 * b]; if (tr[3 * b + 1] > t1) t1 = tr[3 * b + 1]; }
    printf("last launch: first start to last end %.1f us\n", (t1 - t0) * 0.01);
    FILE *f = fopen(getenv("SK_TRACE"), "w");
    for (uint32_t b = 0; b < nb; ++b) {
      const unsigned long long id = tr[3 * b + 2];
      const uint32_t hw = (uint32_t)id, xcc = (uint32_t)(id >> 32) & 15u;
      // HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
      fprintf(f, "%u %.2f %.2f xcc %u se %u sh %u cu %u\n", b, (tr[3 * b] - t0) * 0.01, (tr[3 * b + 1] - t0) * 0.01, xcc,
              (hw >> 13) & 7u, (hw >> 12) & 1u, (hw >> 8) & 15u);
    }
    fclose(f);
  }
  // exactness: the unfiltered pass on the first genomes
  const uint32_t nc = n < 8 ? n : 8;
  setenv("NIQKI_SKETCH_FILTER", "0", 1);
  a.sketches = sk2;
  CK(nq::launch_sketch(a, nc, L, 0));
  CK(hipDeviceSynchronize());
  std::vector<int32_t> x((size_t)nc * d.F), y((size_t)nc * d.F);
  CK(hipMemcpy(x.data(), sk, x.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y.data(), sk2, y.size() * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < x.size(); ++i) bad += x[i] != y[i];
  printf("filtered vs exact pass on %u genomes: %zu cells differ\n", nc, bad);
  return bad != 0;
}
