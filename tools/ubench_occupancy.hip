// ubench_occupancy.hip -- would nq::sketch_kernel's hot loop run faster with more wavefronts per SIMD?  The kernel keeps
// the 2^15 cells of a sketch in LDS as u32 (128 KB): ONE 1024-thread workgroup per CU, 4 wavefronts per SIMD, and its 92
// vector registers allow 5.  The loop of tools/ubench_roll.hip form 1 (roll, canonical choice, filter hash, mbcnt push;
// no drains), here with the LDS footprint and register budget as parameters:
//   A  130 KB of LDS, registers as the compiler likes  (the product: 4 waves per SIMD)
//   B   76 KB of LDS, at most 64 registers             (two workgroups per CU: 8 waves per SIMD)
//   C   76 KB of LDS, 768-thread workgroups, at most 80 registers  (two per CU: 6 waves per SIMD)
//   D  130 KB of LDS, at most 64 registers             (4 waves per SIMD: what the register limit alone costs)
// Prints SIMD cycles per wave step (kernel time x clock x waves per SIMD / steps), i.e. the inverse throughput of a SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/ubench_occupancy.hip -o tools/bin/ubench_occupancy
// (Includes the product source for its helpers; nothing here is part of the product.)
#include "../niqki_amd/csrc/nq_sketch.hip"

#include <cstdio>

using namespace nq;

__device__ unsigned long long ub_clk[2];
constexpr uint32_t kRegion = 32;   // groups of 16 bases a lane's input region holds (re-read: cache resident)

template <int BLOCK, int AREA>
__device__ __forceinline__ void roll_body(const uint8_t *bytes, uint32_t groups, uint32_t thr, uint32_t *sink) {
  extern __shared__ __align__(16) uint32_t smem[];
  const uint32_t tid = threadIdx.x, wave = tid >> 6;
  uint2 *lut64 = (uint2 *)smem;
  for (uint32_t i = tid; i < 256; i += BLOCK) {
    const uint32_t e = code_entry(i);
    lut64[i] = make_uint2(e & 3u, ((e >> 2) & 3u) << 28);
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  const uint32_t bottom = __builtin_amdgcn_readfirstlane(4096u + wave * (uint32_t)AREA);
  uint32_t top = bottom;
  const uint64_t lane_id = (uint64_t)blockIdx.x * BLOCK + tid;
  uint64_t fw = lane_id * 0x9E3779B97F4A7C15ULL & ((1ULL << 62) - 1), rc = ~fw & ((1ULL << 62) - 1);
  uint32_t acc = 0;
  const uint8_t *base = bytes + (uint64_t)tid * (kRegion * 16 + 64) + 1;
  const uintptr_t a0 = (uintptr_t)base;
  const uint32_t sh = (uint32_t)(a0 & 3u);
  const uint32_t *qa0 = (const uint32_t *)(a0 & ~(uintptr_t)3), *qa = qa0;
  uint64_t e[16];
  {
    const uint4 A = *(const uint4 *)qa;
    const uint32_t B = qa[4];
    uint4 w;
    w.x = __builtin_amdgcn_alignbyte(A.y, A.x, sh);
    w.y = __builtin_amdgcn_alignbyte(A.z, A.y, sh);
    w.z = __builtin_amdgcn_alignbyte(A.w, A.z, sh);
    w.w = __builtin_amdgcn_alignbyte(B, A.w, sh);
    lut64_16(w, 0, e);
  }
  auto step = [&](uint64_t ent) {
    fw = shl2_64(fw);
    fw = (fw | (uint32_t)ent) & ((1ULL << 62) - 1ULL);
    rc = shr2_64(rc) | (ent & 0xFFFFFFFF00000000ULL);
    const uint64_t canon = fw < rc ? fw : rc;
    push_candidates(rev64_hi_mad(canon), thr, canon, top);
  };
  for (uint32_t g = 0; g < groups; ++g) {
    qa = qa0 + 4 * ((g + 1) & (kRegion - 1));
    const uint4 A = *(const uint4 *)qa;
    const uint32_t B = qa[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) step(e[j]);
    if (top >= bottom + (uint32_t)AREA / 2u) top = bottom;   // (8 steps push at most 4 KB)
    uint4 w;
    w.x = __builtin_amdgcn_alignbyte(A.y, A.x, sh);
    w.y = __builtin_amdgcn_alignbyte(A.z, A.y, sh);
    w.z = __builtin_amdgcn_alignbyte(A.w, A.z, sh);
    w.w = __builtin_amdgcn_alignbyte(B, A.w, sh);
    lut64_8<0>(w, 0, e);
#pragma unroll
    for (int j = 8; j < 16; ++j) step(e[j]);
    lut64_8<1>(w, 0, e);
    if (top >= bottom + (uint32_t)AREA / 2u) top = bottom;
  }
  acc += (uint32_t)fw ^ (uint32_t)rc ^ top;
  if (acc == 0x12345u) sink[0] = acc;
  if (blockIdx.x == 0 && tid == 0) {
    ub_clk[0] = __builtin_readcyclecounter() - c0;
    ub_clk[1] = wall_clock64() - r0;
  }
}

// 8 KB wave areas as tools/ubench_roll.hip
__global__ __launch_bounds__(1024) void roll_a(const uint8_t *b, uint32_t g, uint32_t t, uint32_t *s) { roll_body<1024, 8192>(b, g, t, s); }
__global__ __launch_bounds__(1024, 8) void roll_b(const uint8_t *b, uint32_t g, uint32_t t, uint32_t *s) { roll_body<1024, 8192>(b, g, t, s); }
__global__ __launch_bounds__(768, 6) void roll_c(const uint8_t *b, uint32_t g, uint32_t t, uint32_t *s) { roll_body<768, 8192>(b, g, t, s); }
__global__ __launch_bounds__(1024, 8) void roll_d(const uint8_t *b, uint32_t g, uint32_t t, uint32_t *s) { roll_body<1024, 8192>(b, g, t, s); }

__global__ void fill_kernel(uint8_t *bytes, uint64_t n_bytes) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  if (i * 16 < n_bytes)
    for (int j = 0; j < 16 && i * 16 + j < n_bytes; ++j) bytes[i * 16 + j] = (uint8_t)((0x54474341u >> (8 * ((z >> (2 * j)) & 3))) & 0xFFu);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <typename KERN>
static int run(const char *what, KERN kern, uint32_t block, size_t lds, uint32_t per_cu, const uint8_t *bytes, uint32_t groups, int reps, uint32_t *sink) {
  CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int resident = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)kern, (int)block, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t grid = 256 * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, bytes, groups, 1u << 29, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, 0, bytes, groups, 1u << 29, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long clk[2];
  CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(ub_clk), 16));
  const double steps_per_lane = (double)groups * 16, ghz = clk[0] / (clk[1] * 10.0);
  const double waves_per_simd = per_cu * block / 64.0 / 4.0;
  const double cyc = ms / reps * 1e-3 * ghz * 1e9 / (steps_per_lane * waves_per_simd);
  printf("%-72s resident %d per CU, %4.1f waves per SIMD: %7.1f SIMD cycles per wave step (%.2f GHz)\n", what, resident, waves_per_simd, cyc, ghz);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 0;
}

int main(int argc, char **argv) {
  const uint32_t groups = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const uint64_t n_bytes = 1024 * (kRegion * 16 + 64) + 4096;
  uint8_t *bytes; uint32_t *sink;
  CK(hipMalloc(&bytes, n_bytes));
  CK(hipMalloc(&sink, 256));
  hipLaunchKernelGGL(fill_kernel, dim3((uint32_t)((n_bytes / 16 + 256) / 256)), dim3(256), 0, 0, bytes, n_bytes);
  CK(hipDeviceSynchronize());
  // LDS: 4 KB of table + the wave areas (16 or 12 x 8 KB), padded up to what leaves room for one or two workgroups per CU
  for (int rep = 0; rep < 2; ++rep) {
    if (run("A  130 KB LDS, compiler's registers (the product's occupancy)", roll_a, 1024, 136 * 1024, 1, bytes, groups, reps, sink)) return 1;
    if (run("B  78 KB LDS, <= 64 registers, two workgroups per CU", roll_b, 1024, 78 * 1024, 2, bytes, groups, reps, sink)) return 1;
    if (run("C  78 KB LDS, 768 threads, <= 80 registers, two workgroups per CU", roll_c, 768, 78 * 1024, 2, bytes, groups, reps, sink)) return 1;
    if (run("D  130 KB LDS, <= 64 registers, one workgroup per CU", roll_d, 1024, 136 * 1024, 1, bytes, groups, reps, sink)) return 1;
  }
  return 0;
}
