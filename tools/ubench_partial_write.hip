// ubench_partial_write.hip -- what does a partial-line write cost on MI355X?  The gather kernel's
// striped tiles write every other u16 column of a counter row per tile pass.  Patterns over a 4 GiB
// region (every workgroup owns a 200 KB row, like a counter row of 100 000 genomes):
//   0: whole row, 4-byte stores            1: even u16 columns only (2-byte stores)
//   2: alternate 32-byte segments          3: alternate 64-byte segments
//   4: alternate 128-byte lines            5: both passes of pattern 1 back to back (even, then odd)
//   6: both passes of pattern 2            7: both passes of pattern 3
// Reports useful bytes written, time and the rate.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_partial_write.hip -o /tmp/ubench_partial_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint32_t kRow = 100000;  // u16 columns per row

template <int SEG>  // alternate segments of SEG bytes (SEG >= 4), phase = which half
__device__ void write_segments(uint16_t *row, uint32_t phase, uint32_t val) {
  // word index i over the words of this phase's segments
  constexpr uint32_t WPS = SEG / 4;
  uint32_t *r = (uint32_t *)row;
  const uint32_t n_words = kRow / 2;
  for (uint32_t i = threadIdx.x; i < n_words / 2; i += blockDim.x) {
    const uint32_t seg = i / WPS, off = i % WPS;
    const uint32_t w = (seg * 2 + phase) * WPS + off;
    if (w < n_words) r[w] = val + i;
  }
}

__global__ __launch_bounds__(1024) void k(uint16_t *base, int pattern, uint32_t val) {
  uint16_t *row = base + (uint64_t)blockIdx.x * kRow;
  switch (pattern) {
    case 0: { uint32_t *r = (uint32_t *)row; for (uint32_t i = threadIdx.x; i < kRow / 2; i += 1024) r[i] = val + i; break; }
    case 1: for (uint32_t i = threadIdx.x; i < kRow / 2; i += 1024) row[2 * i] = (uint16_t)(val + i); break;
    case 2: write_segments<32>(row, 0, val); break;
    case 3: write_segments<64>(row, 0, val); break;
    case 4: write_segments<128>(row, 0, val); break;
    case 5: for (uint32_t ph = 0; ph < 2; ++ph) { for (uint32_t i = threadIdx.x; i < kRow / 2; i += 1024) row[2 * i + ph] = (uint16_t)(val + i); __syncthreads(); } break;
    case 6: write_segments<32>(row, 0, val); __syncthreads(); write_segments<32>(row, 1, val); break;
    case 7: write_segments<64>(row, 0, val); __syncthreads(); write_segments<64>(row, 1, val); break;
  }
}

int main() {
  const uint32_t rows = 20480;  // 4.1 GB
  uint16_t *d;
  if (hipMalloc(&d, (uint64_t)rows * kRow * 2) != hipSuccess) return 1;
  hipMemset(d, 0, (uint64_t)rows * kRow * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char *names[] = {"whole row, 4-byte stores", "even u16 columns (2-byte stores)", "alternate 32-byte segments", "alternate 64-byte segments",
                         "alternate 128-byte lines", "even then odd u16 columns", "both phases of 32-byte segments", "both phases of 64-byte segments"};
  for (int p = 0; p < 8; ++p) {
    k<<<rows, 1024>>>(d, p, 1); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 3; ++r) k<<<rows, 1024>>>(d, p, 7 + r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double useful = (double)rows * kRow * 2 * ((p == 0 || p >= 5) ? 1.0 : 0.5);
    printf("pattern %d  %-36s useful %.2f GB  %.3f ms  %.2f TB/s useful\n", p, names[p], useful / 1e9, ms, useful / ms / 1e9);
  }
  return 0;
}
