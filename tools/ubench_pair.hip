// ubench_pair.hip -- do two workgroups that walk the SAME random line sequence, a given number
// of steps apart, share the lines through L2 / Infinity Cache on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_pair.hip -o gpurun_out/ubench_pair
// Measurement aid for DESIGN.md (gather kernel, "paired tiles"); not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

// mode 0: partners are blocks b and b+8 (same XCD when blocks go round-robin over 8 XCDs)
// mode 1: partners are blocks 2p and 2p+1 (neighbouring XCDs)
// mode 2: no partner (every workgroup has its own sequence)
// Every lane reads 2 bytes of its own random 128-byte line (64 lines per wave-load), 16 loads in
// flight per wave: the memory-bound regime of the gather kernel's table look-ups.
__global__ __launch_bounds__(1024) void pair_kernel(const uint16_t *tab, uint64_t n_lines, int iters, int skew,
                                                    int mode, uint64_t *out) {
  extern __shared__ uint32_t lds[];  // only to hold one workgroup per CU, like the gather kernel
  const uint32_t b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t pair, role;
  if (mode == 0) { pair = (b >> 4) * 8 + (b & 7); role = (b >> 3) & 1; }
  else if (mode == 1) { pair = b >> 1; role = b & 1; }
  else { pair = b; role = 0; }
  uint32_t acc = 0;
  const int off = role ? skew : 0;
  for (int i = 0; i < iters; i += 16) {
    uint32_t v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int step = i + u - off;  // the partner reads what the leader read `skew` steps ago
      const uint64_t line = mix(((uint64_t)pair << 44) ^ ((uint64_t)(w * 64 + lane) << 32) ^ (uint32_t)step) % n_lines;
      v[u] = tab[line * 64 + (lane & 7)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  if (acc == 0x12345) out[0] = acc + lds[0];
}

int main() {
  const size_t bytes = 8ULL << 30;
  uint16_t *tab; uint64_t *out;
  hipMalloc(&tab, bytes); hipMalloc(&out, 8);
  hipMemset(tab, 1, bytes);
  hipFuncSetAttribute((const void *)pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, e; hipEventCreate(&a); hipEventCreate(&e);
  const uint64_t n_lines = bytes >> 7;
  const int blocks = 1024, iters = 256;
  printf("%6s %8s %10s %14s\n", "mode", "skew", "ms", "Glines/s");
  for (int mode : {2, 0, 1}) {
    for (int skew : {0, 16, 32, 64, 128, 256}) {
      if (mode == 2 && skew) continue;
      pair_kernel<<<blocks, 1024, 128 * 1024>>>(tab, n_lines, 16, skew, mode, out);
      hipEventRecord(a);
      pair_kernel<<<blocks, 1024, 128 * 1024>>>(tab, n_lines, iters, skew, mode, out);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, a, e);
      const double loads = (double)blocks * 1024 * iters;
      printf("%6d %8d %10.3f %14.2f\n", mode, skew, ms, loads / ms / 1e6);
    }
  }
  return 0;
}
