// ubench_mix.hip -- do instructions of different kinds (vector ALU, scalar ALU, LDS) from the four waves of a
// SIMD issue side by side on gfx950, or does a SIMD issue one instruction at a time whatever its kind?
// 1024-thread workgroups (4 waves per SIMD), all CUs busy; SIMD cycles per wave-instruction = kernel time x
// clock / (instructions per wave x 4 waves).  Measurement aid for DESIGN.md 4.4; not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o tools/bin/ubench_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define V8(n) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##n) : "v"(b));
#define S8(n) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s##n) : "s"(sb) : "scc");
#define C8(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
#define R8(n) asm volatile("v_readlane_b32 %0, %1, " #n : "=s"(s##n) : "v"(a##n));
#define L8(n) asm volatile("ds_add_u32 %0, %1" : : "v"(la), "v"(a##n) : "memory");
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint64_t *clk) {
  __shared__ uint32_t lds[1024];
  uint32_t t = threadIdx.x + blockIdx.x * 1024u;
  uint32_t a0 = t, a1 = t * 3 + 1, a2 = t ^ 0x1234567u, a3 = t + 77, a4 = t * 5 + 3, a5 = ~t, a6 = t + 9, a7 = t * 7 + 5;
  uint32_t b = t | 1u;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8, sb = __builtin_amdgcn_readfirstlane(blockIdx.x | 1u);
  uint32_t la = (threadIdx.x & 1023u) * 4u;
  lds[threadIdx.x] = 0;
  __syncthreads();
  uint64_t c0 = 0, r0 = 0;
  if (threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
  for (uint32_t i = 0; i < iters; ++i) {
    if (KIND == 0) { REP8(V8) REP8(V8) }                       // 16 full-rate vector ops
    if (KIND == 1) { REP8(S8) REP8(S8) }                       // 16 scalar ops
    if (KIND == 2) { V8(0) S8(0) V8(1) S8(1) V8(2) S8(2) V8(3) S8(3) V8(4) S8(4) V8(5) S8(5) V8(6) S8(6) V8(7) S8(7) }   // 8 + 8
    if (KIND == 3) { REP8(C8) REP8(C8) }                       // 16 cheap-class vector ops
    if (KIND == 4) { REP8(R8) REP8(R8) }                       // 16 v_readlane
    if (KIND == 5) { REP8(L8) }                                // 8 LDS atomics (conflict-free)
    if (KIND == 6) { V8(0) L8(0) V8(1) L8(1) V8(2) L8(2) V8(3) L8(3) V8(4) L8(4) V8(5) L8(5) V8(6) L8(6) V8(7) L8(7) }   // 8 vector + 8 LDS
    if (KIND == 7) { V8(0) S8(0) L8(0) V8(1) S8(1) L8(1) V8(2) S8(2) L8(2) V8(3) S8(3) L8(3) V8(4) S8(4) L8(4) V8(5) S8(5) L8(5) V8(6) S8(6) L8(6) V8(7) S8(7) L8(7) }
    if (KIND == 8) { C8(0) S8(0) C8(1) S8(1) C8(2) S8(2) C8(3) S8(3) C8(4) S8(4) C8(5) S8(5) C8(6) S8(6) C8(7) S8(7) }   // 8 cheap + 8 scalar
  }
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = __builtin_readcyclecounter() - c0; clk[2 * blockIdx.x + 1] = wall_clock64() - r0; }
  out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7 ^ lds[threadIdx.x];
}

template <int KIND>
static void run(const char *what, int n_inst, uint32_t *out, uint64_t *clk) {
  const int blocks = 256, iters = 20000;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k<KIND><<<blocks, 1024>>>(out, 100, clk);
  (void)hipEventRecord(a);
  k<KIND><<<blocks, 1024>>>(out, iters, clk);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms;
  (void)hipEventElapsedTime(&ms, a, b);
  uint64_t h[2];
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);    // shader clock: cycle counter ticks per microsecond
  const double cyc = ms * 1e3 * mhz / ((double)iters * n_inst * 4);
  printf("  %-44s %5.2f SIMD cycles per wave-instruction (%d per wave and iteration, clock %.0f MHz)\n", what, cyc, n_inst, mhz);
}

int main() {
  uint32_t *out;
  uint64_t *clk;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipMalloc(&clk, 256 * 16);
  run<0>("16 x v_lshl_add_u32", 16, out, clk);
  run<3>("16 x v_and_b32", 16, out, clk);
  run<1>("16 x s_add_u32", 16, out, clk);
  run<4>("16 x v_readlane_b32", 16, out, clk);
  run<2>("8 x (v_lshl_add_u32, s_add_u32)", 16, out, clk);
  run<8>("8 x (v_and_b32, s_add_u32)", 16, out, clk);
  run<5>("8 x ds_add_u32 (no conflicts)", 8, out, clk);
  run<6>("8 x (v_lshl_add_u32, ds_add_u32)", 16, out, clk);
  run<7>("8 x (v_lshl_add_u32, s_add_u32, ds_add_u32)", 24, out, clk);
  return 0;
}
