// ubench_densify.hip -- what bounds a densification pass of nq::sketch_reads_kernel (src/niqki_index.cpp:313-331 on
// one wavefront: per pass every entry proposes to a pseudo-random cell with ds_min_u32, reads the cell back, and the
// wave counts the winners)?  One wavefront per workgroup with the kernel's LDS footprint (20 KB: 8 workgroups per CU,
// two waves per SIMD), R entries per lane, cells never filled (every proposal finds an occupied cell: the tail).
//   form 0: the kernel's pass: R ds_min, R ds_read, wait, R compares + ballots        (1 round trip per pass)
//   form 1: R ds_read only, wait, compares + ballots                                   (no atomics)
//   form 2: R ds_min only, nothing waited for                                          (no read-back)
//   form 3: U passes per round trip: U x R ds_min, U x R ds_read, wait, compares      (same LDS work, 1/U of the waits)
//   form 4: as 0, but the addresses are lane-linear (no bank conflicts)
//   form 5: as 0 without the ballots / popcounts (one OR-reduced compare per pass)
//   form 6: 16-bit cells (half the LDS footprint: 16 waves per CU), no atomics: R ds_read_u16 of the targets, and only
//           where a lane found its target empty (never here: the tail) a write / read-back / winner write
//   form 7: the same with a third of the lanes finding an empty target every pass (the dense phase): R ds_read_u16,
//           R masked ds_write_b16, R masked ds_read_u16, compares
// Prints SIMD cycles per pass and wave (wall clock x 2.4 GHz) and passes per microsecond and CU.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_densify.hip -o tools/bin/ubench_densify
// Measurement aid for DESIGN.md 4.2; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int FORM, int R, int U>
__global__ __launch_bounds__(64) void pass_kernel(uint32_t iters, uint32_t *sink) {
  extern __shared__ __align__(16) uint32_t smem[];
  const uint32_t F = 4096, Fm = F - 1u, lane = threadIdx.x;
  for (uint32_t i = lane; i < F; i += 64) smem[i] = mix32(i) & 0x3FFu;   // occupied cells (values below 2^31)
  __syncthreads();
  uint32_t T[R], B[R], mk[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t v = mix32(blockIdx.x * 131u + lane * R + (uint32_t)k);
    T[k] = FORM == 4 ? lane + 64u * (uint32_t)k : v;
    B[k] = FORM == 4 ? 64u * R : (mix32(v) | 1u);
    mk[k] = 0x80000000u | ((lane * R + (uint32_t)k) & Fm);
  }
  uint32_t tot = 0;
  if (FORM == 6 || FORM == 7) {
    uint16_t *c16 = (uint16_t *)smem;
    for (uint32_t it = 0; it < iters; ++it) {
      uint32_t pre[R];
#pragma unroll
      for (int k = 0; k < R; ++k) pre[k] = c16[T[k] & Fm];
      wave_lds_order();
      bool prop[R], any = false;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        prop[k] = FORM == 7 ? ((pre[k] + it + lane) % 3u == 0u) : pre[k] == 0xFFFFu;
        any |= prop[k];
      }
      if (__any(any)) {
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (prop[k]) c16[T[k] & Fm] = (uint16_t)(0x8000u | (mk[k] & 0xFFFu));
        wave_lds_order();
        uint32_t back[R];
#pragma unroll
        for (int k = 0; k < R; ++k) back[k] = prop[k] ? c16[T[k] & Fm] : 0u;
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < R; ++k) {
          const bool won = prop[k] && back[k] == (0x8000u | (mk[k] & 0xFFFu));
          if (won) c16[T[k] & Fm] = (uint16_t)(mk[k] & 0x3FFu);
          tot += (uint32_t)__popcll(__ballot(won));
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) T[k] += B[k];
      if (tot == 0xFFFFFFFFu) break;
    }
  } else if (FORM == 3) {
    for (uint32_t it = 0; it < iters; it += U) {
      uint32_t t[R];
#pragma unroll
      for (int k = 0; k < R; ++k) t[k] = T[k];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < R; ++k) { atomicMin(&smem[t[k] & Fm], mk[k] | ((uint32_t)u << 16)); t[k] += B[k]; }
      wave_lds_order();
      uint32_t back[U][R];
#pragma unroll
      for (int k = 0; k < R; ++k) t[k] = T[k];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < R; ++k) { back[u][k] = smem[t[k] & Fm]; t[k] += B[k]; }
      wave_lds_order();
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < R; ++k) tot += (uint32_t)__popcll(__ballot(back[u][k] == (mk[k] | ((uint32_t)u << 16))));
#pragma unroll
      for (int k = 0; k < R; ++k) T[k] = t[k];
      if (tot == 0xFFFFFFFFu) break;
    }
  } else {
    for (uint32_t it = 0; it < iters; ++it) {
      if (FORM != 1) {
#pragma unroll
        for (int k = 0; k < R; ++k) atomicMin(&smem[T[k] & Fm], mk[k]);
        wave_lds_order();
      }
      if (FORM != 2) {
        uint32_t back[R];
#pragma unroll
        for (int k = 0; k < R; ++k) back[k] = smem[T[k] & Fm];
        wave_lds_order();
        if (FORM == 5) {
          bool any = false;
#pragma unroll
          for (int k = 0; k < R; ++k) any |= back[k] == mk[k];
          tot += __any(any) ? 1u : 0u;
        } else {
#pragma unroll
          for (int k = 0; k < R; ++k) tot += (uint32_t)__popcll(__ballot(back[k] == mk[k]));
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) T[k] += B[k];
      if (tot == 0xFFFFFFFFu) break;   // (the exit test of a pass; never taken)
    }
  }
  if (tot == 0x12345u) sink[0] = tot;
}

template <int FORM, int R, int U>
static void run(const char *what, uint32_t *sink, size_t lds, int per_cu) {
  const uint32_t iters = 8192, blocks = 256 * per_cu * 4;
  hipFuncSetAttribute((const void *)pass_kernel<FORM, R, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((pass_kernel<FORM, R, U>), dim3(blocks), dim3(64), lds, 0, iters / 8, sink);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL((pass_kernel<FORM, R, U>), dim3(blocks), dim3(64), lds, 0, iters, sink);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  // four rounds of per_cu one-wave workgroups on every CU: a wave's pass takes ms / (4 * iters)
  const double us_per_pass = ms * 1e3 / (4.0 * iters);
  printf("%-78s R=%d U=%d %d waves/CU: %7.1f SIMD cycles per pass and wave, %6.2f passes/us/CU\n", what, R, U, per_cu, us_per_pass * 2400.0,
         per_cu / us_per_pass);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  uint32_t *sink;
  hipMalloc(&sink, 256);
  const size_t lds = 20288;     // nq::sketch_reads_lds_bytes at S = 12: 8 workgroups per CU
  const size_t lds4 = 40000;    // 4 per CU
  const size_t lds16 = 10000;   // 16 per CU (what half the footprint would allow)
  run<0, 2, 1>("0: R ds_min, R ds_read, wait, ballots (the kernel's pass)", sink, lds, 8);
  run<5, 2, 1>("5: the same with one any() instead of R ballots + popcounts", sink, lds, 8);
  run<1, 2, 1>("1: R ds_read only", sink, lds, 8);
  run<2, 2, 1>("2: R ds_min only, nothing waited for", sink, lds, 8);
  run<4, 2, 1>("4: as 0 on lane-linear addresses (no bank conflicts)", sink, lds, 8);
  run<3, 2, 2>("3: U passes per round trip", sink, lds, 8);
  run<3, 2, 4>("3: U passes per round trip", sink, lds, 8);
  run<3, 2, 8>("3: U passes per round trip", sink, lds, 8);
  run<0, 2, 1>("0: the kernel's pass at 4 waves per CU", sink, lds4, 4);
  run<0, 2, 1>("0: the kernel's pass at 16 waves per CU", sink, lds16, 16);
  run<3, 2, 8>("3: U = 8 at 16 waves per CU", sink, lds16, 16);
  run<6, 2, 1>("6: 16-bit cells, targets read, nothing empty (the tail), 16 waves per CU", sink, lds16, 16);
  run<7, 2, 1>("7: 16-bit cells, a third of the lanes write / read back every pass, 16 waves per CU", sink, lds16, 16);
  run<6, 2, 1>("6: the same tail pass at 8 waves per CU", sink, lds, 8);
  run<7, 2, 1>("7: the same dense pass at 8 waves per CU", sink, lds, 8);
  run<0, 1, 1>("0: one entry per lane", sink, lds, 8);
  run<0, 4, 1>("0: four entries per lane", sink, lds, 8);
  hipFree(sink);
  return 0;
}
