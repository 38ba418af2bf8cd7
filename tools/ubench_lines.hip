// ubench_lines.hip -- how fast can a CU pull whole random 128-byte lines the way nq::walk64 does it
// (one wave-wide load instruction per line, 64 lanes x 2 bytes; 32 lines per round, two rounds in
// flight, 16 waves per CU), and does the instruction's shape matter?
//   form 0: global/buffer_load_ushort, one line per instruction          (the kernel's form)
//   form 1: buffer_load_dword, 64 lanes x 4 bytes = two consecutive lines per instruction
//   form 2: buffer_load_dwordx2, four consecutive lines per instruction
//   form 3: buffer_load_dword on lanes 0..31 only (exec mask), one line per instruction
// over a table that fits L2 (4 MB) and one that does not (2 GB).  Lines per microsecond and CU, GB/s.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lines.hip -o tools/bin/ubench_lines
// Measurement aid for DESIGN.md 4.4; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

template <int FORM, int UNROLL>
__global__ __launch_bounds__(1024) void lines_kernel(const uint8_t *tab, uint32_t n_lines, uint32_t rounds, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_id = (blockIdx.x * 1024u + threadIdx.x) >> 6;
  constexpr uint32_t LPI = FORM == 1 ? 2 : FORM == 2 ? 4 : 1;   // lines per instruction
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, (int)(n_lines * 128u - 1u), 0x00020000);
  uint32_t acc = 0;
  uint32_t pos = mix32(wave_id * 64u + lane) % (n_lines / LPI);   // lane j holds the line group of step j
  uint32_t ga[UNROLL], gb[UNROLL], gc[UNROLL], gd[UNROLL];
  auto fetch = [&](uint32_t (&g)[UNROLL], uint32_t (&h)[UNROLL], uint32_t j0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t off = __builtin_amdgcn_readlane(pos, (j0 + u) & 63) * (128u * LPI);
      if (FORM == 0) g[u] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, lane * 2u, off, 0);
      if (FORM == 1) g[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4u, off, 0);
      if (FORM == 2) {
        auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8u, off, 0);
        g[u] = v[0]; h[u] = v[1];
      }
      if (FORM == 3) g[u] = lane < 32 ? __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4u, off, 0) : 0u;
    }
  };
  auto use = [&](uint32_t (&g)[UNROLL], uint32_t (&h)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc ^= g[u]; if (FORM == 2) acc ^= h[u]; }
  };
  fetch(ga, gc, 0);
  for (uint32_t r = 0; r < rounds; ++r) {
    fetch(gb, gd, UNROLL);
    use(ga, gc);
    pos = mix32(pos + r) % (n_lines / LPI);
    fetch(ga, gc, 0);
    use(gb, gd);
  }
  use(ga, gc);
  if (acc == 0x12345u) out[0] = acc;
}

template <int FORM>
static void run(const uint8_t *tab, uint32_t n_lines, const char *what, uint32_t *out, int waves_per_cu) {
  constexpr int U = 32;
  const uint32_t rounds = 200;
  const int blocks = 256;
  const int threads = waves_per_cu * 64;
  constexpr uint32_t LPI = FORM == 1 ? 2 : FORM == 2 ? 4 : 1;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  lines_kernel<FORM, U><<<blocks, threads>>>(tab, n_lines, 4, out);
  hipEventRecord(a);
  lines_kernel<FORM, U><<<blocks, threads>>>(tab, n_lines, rounds, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double)blocks * waves_per_cu * (2.0 * rounds + 1) * U;
  const double lines = instr * LPI;
  printf("  form %d %-34s waves/CU %2d: %7.1f lines/us/CU  %7.1f load instr/us/CU  %7.0f GB/s\n", FORM, what, waves_per_cu,
         lines / blocks / (ms * 1e3), instr / blocks / (ms * 1e3), lines * 128 / ms / 1e6);
}

// form 4: the padded walk's own form (PairWalk: global_load_dword, two consecutive lines per instruction, 64-bit
// addresses) over tables of 2 to 64 GB: what random 128-byte lines cost once the table outgrows the Infinity Cache
// and the translation caches (a 500 000-genome index keeps 67 GB of ids).
template <int UNROLL>
__global__ __launch_bounds__(1024) void lines_big_kernel(const uint8_t *tab, uint64_t n_pairs, uint32_t rounds, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave_id = (blockIdx.x * 1024u + threadIdx.x) >> 6;
  uint32_t acc = 0;
  uint64_t pos = (((uint64_t)mix32(wave_id * 64u + lane) << 17) ^ mix32(lane * 977u + wave_id)) % n_pairs;
  uint32_t ga[UNROLL], gb[UNROLL];
  auto fetch = [&](uint32_t (&g)[UNROLL], uint32_t j0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)pos, (j0 + u) & 63), hi = __builtin_amdgcn_readlane((uint32_t)(pos >> 32), (j0 + u) & 63);
      const uint64_t off = (((uint64_t)hi << 32) | lo) * 256ull + lane * 4u;
      g[u] = *(const __attribute__((address_space(1))) uint32_t *)(tab + off);
    }
  };
  auto use = [&](uint32_t (&g)[UNROLL]) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= g[u];
  };
  fetch(ga, 0);
  for (uint32_t r = 0; r < rounds; ++r) {
    fetch(gb, UNROLL);
    use(ga);
    pos = (((uint64_t)mix32((uint32_t)pos + r) << 17) ^ mix32((uint32_t)(pos >> 7) + lane)) % n_pairs;
    fetch(ga, 0);
    use(gb);
  }
  use(ga);
  if (acc == 0x12345u) out[0] = acc;
}

static void run_big(const uint8_t *tab, size_t bytes, uint32_t *out) {
  constexpr int U = 16;   // 16 loads = 32 lines per round, two rounds in flight: the kernel's depth
  const uint32_t rounds = 200;
  const int blocks = 256, waves_per_cu = 16;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  lines_big_kernel<U><<<blocks, 1024>>>(tab, bytes / 256, 4, out);
  hipEventRecord(a);
  lines_big_kernel<U><<<blocks, 1024>>>(tab, bytes / 256, rounds, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double lines = (double)blocks * waves_per_cu * (2.0 * rounds + 1) * U * 2;
  printf("  form 4 global dword, 2 lines / instr, table %5zu GB: %7.1f lines/us/CU  %7.0f GB/s\n", bytes >> 30, lines / blocks / (ms * 1e3),
         lines * 128 / ms / 1e6);
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == 'b') {   // ubench_lines big
    uint32_t *out;
    hipMalloc(&out, 64);
    for (size_t gb : {size_t(2), size_t(8), size_t(16), size_t(64)}) {
      uint8_t *t = nullptr;
      if (hipMalloc(&t, gb << 30) != hipSuccess) { printf("  table %zu GB: allocation failed\n", gb); continue; }
      hipMemset(t, 1, gb << 30);
      run_big(t, gb << 30, out);
      hipFree(t);
    }
    return 0;
  }
  uint8_t *tab;
  uint32_t *out;
  const size_t big = 2ull << 30;
  hipMalloc(&tab, big);
  hipMalloc(&out, 64);
  hipMemset(tab, 1, big);
  for (size_t bytes : {size_t(4) << 20, big}) {
    const uint32_t n_lines = (uint32_t)(bytes >> 7);
    printf("table %zu MB\n", bytes >> 20);
    for (int w : {16, 8}) {
      run<0>(tab, n_lines, "ushort, 1 line / instr", out, w);
      run<1>(tab, n_lines, "dword, 2 lines / instr", out, w);
      run<2>(tab, n_lines, "dwordx2, 4 lines / instr", out, w);
      run<3>(tab, n_lines, "dword on 32 lanes, 1 line / instr", out, w);
    }
  }
  return 0;
}
