// ubench_latency.hip -- dependent-issue behaviour on gfx950: N independent chains per wave (1, 2, 4, 8) of one
// opcode, 4 waves per SIMD (one 1024-thread workgroup per CU) and 1 wave per SIMD.  With one chain per wave a
// SIMD has 4 chains in flight: the situation of a kernel whose every instruction depends on the one before it.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_latency.hip -o /tmp/ul && /tmp/ul
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

enum { MAD64, MULLO, XOR, ADD, LSHLADD, CNDMASK, LSHL64, MULHI };

template <int OP, int CH>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint32_t seed, uint64_t *clk) {
  uint32_t t = threadIdx.x + blockIdx.x * 1024u + seed;
  uint32_t a[8];
  uint64_t w[8];
  for (int i = 0; i < 8; ++i) { a[i] = t * (2 * i + 3) + i; w[i] = (uint64_t)a[i] * 0x9E3779B97F4A7C15ULL; }
  uint32_t b = t * 2654435761u | 1u;
  uint32_t s0 = __builtin_amdgcn_readfirstlane(seed * 13u + 0x6659FD93u);
  uint64_t t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
  for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(b), "s"(s0) : "vcc");
        if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "s"(s0));
        if (OP == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "s"(s0));
        if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        if (OP == LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[c]) : "v"(b));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "s"((uint64_t)0x5555555555555555ull));
        if (OP == LSHL64) asm volatile("v_lshlrev_b64 %0, 2, %0" : "+v"(w[c]));
      }
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - r0; }
  if (threadIdx.x == 0) { clk[2 + 2 * blockIdx.x] = r0; clk[3 + 2 * blockIdx.x] = wall_clock64(); }
  uint32_t x = 0;
  for (int i = 0; i < 8; ++i) x ^= a[i] ^ (uint32_t)w[i] ^ (uint32_t)(w[i] >> 32);
  if (x == 0x12345u) out[0] = x;
}

template <int OP, int CH>
void run(const char *name, uint32_t *out, uint64_t *clk, int threads) {
  const int blocks = 256, iters = 1000;
  k<OP, CH><<<blocks, threads>>>(out, 20, 1, clk);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<OP, CH><<<blocks, threads>>>(out, iters, 1, clk);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[2 + 512];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  uint64_t lo = ~0ull, hi = 0, latest_start = 0;
  for (int b = 0; b < blocks; ++b) { if (h[2 + 2 * b] < lo) lo = h[2 + 2 * b]; if (h[3 + 2 * b] > hi) hi = h[3 + 2 * b]; if (h[2 + 2 * b] > latest_start) latest_start = h[2 + 2 * b]; }
  const double waves_per_simd = threads / 256.0;
  const double winst = waves_per_simd * iters * 64.0;
  const double ghz = (double)h[0] / (h[1] * 10.0);
  printf("%-8s chains/wave %d  waves/SIMD %.0f : kernel %7.1f us = %5.2f cycles per wave-instruction per SIMD at %.2f GHz (oldest wave alone: %5.2f)\n", name, CH,
         waves_per_simd, ms * 1e3, ms * 1e6 * ghz / winst, ghz, (double)h[0] / winst);
}

int main() {
  uint32_t *out;
  uint64_t *clk;
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&clk, 8 * 1024);
#define ALL(OP)                                                                                                   \
  run<OP, 1>(#OP, out, clk, 1024); run<OP, 2>(#OP, out, clk, 1024); run<OP, 4>(#OP, out, clk, 1024);              \
  run<OP, 8>(#OP, out, clk, 1024); run<OP, 1>(#OP, out, clk, 256); run<OP, 2>(#OP, out, clk, 256);                \
  run<OP, 8>(#OP, out, clk, 256);
  ALL(ADD) ALL(XOR) ALL(LSHLADD) ALL(MULLO) ALL(MAD64) ALL(LSHL64) ALL(CNDMASK)
  return 0;
}
