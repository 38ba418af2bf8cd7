// ubench_alu.hip -- issue cost of the integer ops the sketch kernel is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234567, a3 = a0 + 77;
  uint64_t b0 = a0, b1 = a1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) { a0 = a0 + a1; a1 = a1 + a2; a2 = a2 + a3; a3 = a3 + a0; }                 // v_add_u32
      if (OP == 1) { a0 = a0 * 0x9E3779B1u; a1 = a1 * 0x85EBCA77u; a2 = a2 * 0xC2B2AE3Du; a3 = a3 * 0x27D4EB2Fu; }  // v_mul_lo_u32
      if (OP == 2) { a0 = __umulhi(a0, 0x9E3779B1u); a1 = __umulhi(a1, 0x85EBCA77u); a2 = __umulhi(a2, 0xC2B2AE3Du) ; a3 = __umulhi(a3, 0x27D4EB2Fu); a0 |= 0x80000001u; a1 |= 0x80000001u; a2 |= 0x80000001u; a3 |= 0x80000001u; }
      if (OP == 3) { b0 = b0 * 0xD6E8FEB86659FD93ULL; b1 = b1 * 0xCFEE444D8B59A89BULL; }     // 64-bit mul
      if (OP == 4) { b0 = ((b0 >> 32) ^ b0) * 0xD6E8FEB86659FD93ULL; b1 = ((b1 >> 32) ^ b1) * 0xCFEE444D8B59A89BULL; }
      if (OP == 5) { a0 = ((a0 & 0xFFFFFFu) * (a1 & 0xFFFFFFu)) ; a1 = ((a1 & 0xFFFFFFu) * (a2 & 0xFFFFFFu)); a2 = ((a2 & 0xFFFFFFu) * (a3 & 0xFFFFFFu)); a3 = ((a3 & 0xFFFFFFu) * (a0 & 0xFFFFFFu)) | 1; }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)b0 ^ (uint32_t)(b1 >> 32);
}

template <int OP>
void run(const char *name, int ops_per_unroll, uint32_t *out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int blocks = 256 * 2, iters = 2000;
  k<OP><<<blocks, 1024>>>(out, 10, 1);
  (void)hipEventRecord(a);
  k<OP><<<blocks, 1024>>>(out, iters, 1);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  // wave-instructions per SIMD: blocks*16 waves / (256 CUs*4 SIMDs) * iters*16*ops
  double winst = (double)blocks * 16 / 1024.0 * iters * 16.0 * ops_per_unroll;
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.3f ms   %6.2f cycles per wave-op per SIMD (8 waves/SIMD, 2.4 GHz assumed)\n", name, ms, cycles / winst);
}

int main() {
  uint32_t *out; (void)hipMalloc(&out, 256 * 2 * 1024 * 4);
  run<0>("v_add_u32", 4, out);
  run<1>("v_mul_lo_u32", 4, out);
  run<2>("v_mul_hi_u32 (+or)", 8, out);
  run<3>("64-bit mul by const", 2, out);
  run<4>("xorshift + 64-bit mul", 2, out);
  run<5>("v_mul_u32_u24", 4, out);
  return 0;
}
