// ubench_sector.hip -- does a partial read of a random 128-byte line cost less than the whole line?
// Every wave-instruction reads the first `n_ids` u16 of one random 128-byte aligned line (lanes past
// n_ids re-read id n_ids-1), 16 loads per round, two rounds in flight -- the access shape of the
// gather kernel's bucket walk (nq_query.hip walk64).  Reports wave-loads/s; run under
// rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_32B / TCC_EA0_RDREQ for the bytes actually fetched.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_sector.hip -o gpurun_out/ubench_sector
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void sector_kernel(const uint16_t *tab, uint64_t n_lines, uint32_t n_ids,
                                                      uint32_t line_off_ids, int iters, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t idx = line_off_ids + (lane < n_ids ? lane : n_ids - 1);
  uint32_t acc = 0;
  uint32_t ga[UNROLL], gb[UNROLL];
  auto fetch = [&](int r, uint32_t (&g)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t line = mix(wave * 1315423911ULL + (uint64_t)(r * UNROLL + u)) & (n_lines - 1);  // wave-uniform; n_lines = 2^k
      g[u] = (tab + line * 64)[idx];
    }
  };
  fetch(0, ga);
  for (int i = 0; i < iters; i += 2) {
    fetch(i + 1, gb);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += ga[u];
    fetch(i + 2, ga);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += gb[u];
  }
  if (acc == 0x12345) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t bytes = 8ULL << 30;
  uint16_t *tab; uint32_t *out;
  hipMalloc(&tab, bytes + 4096);
  hipMalloc(&out, 8);
  hipMemset(tab, 1, bytes);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = 64;
  const uint64_t n_lines = bytes >> 7;
  printf("%8s %8s %14s %16s\n", "ids", "offset", "Gwaveloads/s", "GB/s if 128B");
  for (uint32_t off : {0u, 32u}) {
    for (uint32_t n_ids : {64u, 48u, 32u, 16u, 8u, 1u}) {
      if (off + n_ids > 64) continue;
      sector_kernel<16><<<blocks, 1024>>>(tab, n_lines, n_ids, off, 4, out);
      hipEventRecord(a);
      sector_kernel<16><<<blocks, 1024>>>(tab, n_lines, n_ids, off, iters, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double loads = (double)blocks * 16 * (iters + 1) * 16;
      printf("%8u %8u %14.2f %16.1f\n", n_ids, off, loads / ms / 1e6, loads * 128 / ms / 1e6);
    }
  }
  return 0;
}
