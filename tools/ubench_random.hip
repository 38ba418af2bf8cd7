// ubench_random.hip -- random 8-byte load throughput vs table size / locality on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_random.hip -o gpurun_out/ubench_random
// Measurement aid for DESIGN.md ("CSR lookup cost"); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

// region_log2: all 64 lanes of a wave-instruction fall inside one aligned region of this size
// (0 = whole table).  Every lane reads 8 bytes at a 128-byte aligned random offset.
template <int UNROLL>
__global__ __launch_bounds__(1024) void rnd_kernel(const uint64_t *tab, uint64_t n_lines, int region_log2,
                                                   int iters, uint64_t *out) {
  const uint64_t gtid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t wave = gtid >> 6;
  uint64_t acc = 0;
  const uint64_t lines_per_region = region_log2 ? ((1ULL << region_log2) >> 7) : n_lines;
  const uint64_t n_regions = n_lines / lines_per_region;
  for (int i = 0; i < iters; ++i) {
    uint64_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      uint64_t r = mix(wave * 1315423911ULL + (uint64_t)(i * UNROLL + u));
      uint64_t region = r % n_regions;
      uint64_t line = region * lines_per_region + mix(gtid * 77 + i * UNROLL + u) % lines_per_region;
      v[u] = tab[line * 16];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u];
  }
  if (acc == 0x1234567) out[0] = acc;
}

int main() {
  const size_t max_bytes = 8ULL << 30;
  uint64_t *tab, *out;
  hipMalloc(&tab, max_bytes);
  hipMalloc(&out, 8);
  hipMemset(tab, 1, max_bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 2, iters = 64;
  printf("%10s %8s %12s %12s\n", "table_MB", "region", "Gloads/s", "GB/s(128B)");
  for (size_t mb : {16, 64, 256, 1024, 4096, 8192}) {
    for (int reg : {0, 21, 16, 12}) {
      uint64_t n_lines = (mb << 20) >> 7;
      rnd_kernel<4><<<blocks, 1024>>>(tab, n_lines, reg, 4, out);
      hipEventRecord(a);
      rnd_kernel<4><<<blocks, 1024>>>(tab, n_lines, reg, iters, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      double loads = (double)blocks * 1024 * iters * 4;
      printf("%10zu %8d %12.2f %12.1f\n", mb, reg, loads / ms / 1e6, loads * 128 / ms / 1e6);
    }
  }
  return 0;
}
