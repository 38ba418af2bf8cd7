// ubench_lds_atomics.hip -- how many wave-wide LDS atomics does a CU take per cycle?  The bucket walk of
// nq::gather_kernel issues ONE ds_add_u32 per id lane (two per loaded dword: PairWalk::apply), 87.6 k lines x 64 id
// lanes per query at the bench's shape, whatever the lines' fill.  A 1024-thread workgroup per CU (16 waves, as the
// gather kernel), ~100 KB of counters, every lane adds to its own pseudo-random word per instruction:
//   form 0: ds_add_u32, random words of the whole counter array (the walk's pattern for real ids)
//   form 1: ds_add_u32, lane-linear words (no bank conflicts)
//   form 2: ds_add_u32, half of the lanes on random words, half on 32 fixed words of their own (padding ids)
//   form 3: ds_add_rtn_u32, random words (what byte counters with a wrap test would need)
//   form 4: ds_read_b32, random words (for comparison)
//   form 5: ds_add_u32 random + the walk's 5 vector ops per id in between (address and increment from the id)
// Prints SIMD cycles per wave-instruction and CU, and wave-instructions per microsecond and CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_atomics.hip -o tools/bin/ubench_lds_atomics
// Measurement aid for DESIGN.md 4.4; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

constexpr uint32_t kWords = 25024;   // (50048 + 1) / 2 counter words of a tile of the bench's index

template <int FORM>
__global__ __launch_bounds__(1024) void atom_kernel(uint32_t iters, uint32_t *sink) {
  extern __shared__ __align__(16) uint32_t cnt[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  for (uint32_t i = tid; i < kWords + 64; i += 1024) cnt[i] = 0;
  __syncthreads();
  // 16 byte addresses per lane, fixed over the run (the LDS does not care which words): no address arithmetic in the loop
  uint32_t a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    uint32_t w = mix32(blockIdx.x * 7919u + tid * 16u + (uint32_t)k) % kWords;
    if (FORM == 1) w = (tid + 1024u * (uint32_t)k) % kWords;
    if (FORM == 2 && (lane & 1u)) w = kWords + (lane >> 1);
    a[k] = w * 4u;
  }
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (FORM == 3) {
        acc += __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (FORM == 4) {
        acc += *(volatile lds_u32 *)(uintptr_t)a[k];
      } else if (FORM == 5) {
        // the walk's per-id work: word address and increment from a 16-bit id (here: the stored address plays the id)
        uint32_t id = a[k] >> 1, even, addr, odd, inc;
        asm volatile("v_and_b32 %0, 0xfffe, %1" : "=v"(even) : "v"(id));
        asm volatile("v_add_u32 %0, %1, %1" : "=v"(addr) : "v"(even));
        asm volatile("v_and_b32 %0, 1, %1" : "=v"(odd) : "v"(id));
        asm volatile("v_mad_u32_u24 %0, %1, %2, 1" : "=v"(inc) : "v"(odd), "s"(0xFFFFu));
        __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)addr, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  if (acc == 0x12345u || cnt[tid % kWords] == 0xFFFFFFFFu) sink[0] = acc;
}

template <int FORM>
static void run(const char *what, uint32_t *sink) {
  const uint32_t iters = 4096, blocks = 256 * 4;
  const size_t lds = (kWords + 64) * 4;
  (void)hipFuncSetAttribute((const void *)atom_kernel<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((atom_kernel<FORM>), dim3(blocks), dim3(1024), lds, 0, iters / 8, sink);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((atom_kernel<FORM>), dim3(blocks), dim3(1024), lds, 0, iters, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // four workgroups per CU one after another, 16 waves each, 16 instructions per iteration and wave
  const double instr_per_cu = 4.0 * 16.0 * 16.0 * iters;
  const double us = ms * 1e3;
  printf("%-86s %6.2f SIMD cycles per wave-instruction and CU, %7.1f wave-instructions/us/CU\n", what, us * 2400.0 / instr_per_cu, instr_per_cu / us);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
  uint32_t *sink;
  (void)hipMalloc(&sink, 256);
  run<0>("0: ds_add_u32, random words of a 100 KB counter array", sink);
  run<1>("1: ds_add_u32, lane-linear words (no bank conflicts)", sink);
  run<2>("2: ds_add_u32, odd lanes on 32 words of their own (padding ids), even lanes random", sink);
  run<3>("3: ds_add_rtn_u32, random words", sink);
  run<4>("4: ds_read_b32, random words", sink);
  run<5>("5: ds_add_u32 random + the walk's 4 vector ops per id", sink);
  (void)hipFree(sink);
  return 0;
}
