// ubench_lookup.hip -- what bounds the CSR lookup of the gather kernel?
// Emulates its access pattern: wave w of a 1024-thread workgroup walks 64-slot
// iterations; lane = slot; address = (slot*(R+1) + fp)*4 in a table of F*(R+1) u32,
// 8-byte loads, DEPTH iterations in flight.  Compared with uniformly random lines.
// Measurement aid only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
struct __attribute__((packed, aligned(4))) P { uint32_t a, b; };

template <int DEPTH, int PATTERN>
__global__ __launch_bounds__(1024) void k(const uint32_t *tab, uint32_t F, uint32_t R1, uint32_t n_tiles, uint64_t *out) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_it = F / 64;
  const uint32_t t = blockIdx.x % n_tiles;
  const uint32_t *base = tab + (uint64_t)t * F * R1;
  uint64_t acc = 0;
  P buf[DEPTH];
  auto addr = [&](uint32_t it) -> const P * {
    uint64_t h = mix(((uint64_t)blockIdx.x << 32) + it * 64 + lane);
    uint32_t slot = it * 64 + lane;
    uint64_t idx;
    if (PATTERN == 0) {  // gather-like: hll part concentrated on 7 values, low 8 bits uniform
      uint32_t fp = ((4 + (uint32_t)(h >> 40) % 7) << 8) | (uint32_t)(h & 255);
      idx = (uint64_t)slot * R1 + fp;
    } else if (PATTERN == 1) {  // same rows, uniform fp
      idx = (uint64_t)slot * R1 + (uint32_t)(h % (R1 - 1));
    } else {  // uniformly random 128-byte lines of the whole table
      idx = (h % ((uint64_t)F * R1 / 32)) * 32;
    }
    return (const P *)(base + idx);
  };
  uint32_t it = wave;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = *addr(it + d * 16);
  for (; it < n_it; it += 16) {
    P cur = buf[0];
#pragma unroll
    for (int d = 0; d + 1 < DEPTH; ++d) buf[d] = buf[d + 1];
    uint32_t nx = it + DEPTH * 16;
    buf[DEPTH - 1] = *addr(nx < n_it ? nx : it);
    acc += cur.a ^ cur.b;
  }
  if (acc == 0x12345) out[0] = acc;
}

template <int DEPTH, int PATTERN>
void run(const uint32_t *tab, uint32_t F, uint32_t R1, uint32_t n_tiles, uint64_t *out, int blocks, const char *name) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<DEPTH, PATTERN><<<blocks, 1024>>>(tab, F, R1, n_tiles, out);
  (void)hipEventRecord(a);
  k<DEPTH, PATTERN><<<blocks, 1024>>>(tab, F, R1, n_tiles, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  double n = (double)blocks * F;
  printf("%-28s depth %d  %8.3f ms  %7.2f G lookups/s\n", name, DEPTH, ms, n / ms / 1e6);
}

int main() {
  const uint32_t F = 32768, R1 = 4097, n_tiles = 2;
  uint32_t *tab; uint64_t *out;
  size_t bytes = (size_t)n_tiles * F * R1 * 4;
  (void)hipMalloc(&tab, bytes + 4096); (void)hipMalloc(&out, 8);
  (void)hipMemset(tab, 1, bytes);
  const int blocks = 2000;
  run<1, 0>(tab, F, R1, n_tiles, out, blocks, "gather pattern");
  run<2, 0>(tab, F, R1, n_tiles, out, blocks, "gather pattern");
  run<4, 0>(tab, F, R1, n_tiles, out, blocks, "gather pattern");
  run<8, 0>(tab, F, R1, n_tiles, out, blocks, "gather pattern");
  run<2, 1>(tab, F, R1, n_tiles, out, blocks, "rows, uniform fp");
  run<8, 1>(tab, F, R1, n_tiles, out, blocks, "rows, uniform fp");
  run<2, 2>(tab, F, R1, n_tiles, out, blocks, "uniform random lines");
  run<8, 2>(tab, F, R1, n_tiles, out, blocks, "uniform random lines");
  return 0;
}
