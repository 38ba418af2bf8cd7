// ubench_roll.hip -- the hot loop of nq::sketch_kernel<1024, 32, 31> (roll, canonical choice, filter hash, candidate
// push; no drains) in isolation, in the forms VERDICT r3 item 6 asks about:
//   1  the product's form: ASCII bytes -> 8-byte LDS code table (one SDWA shift + ds_read_b64 per base), rolling
//      64-bit shifts, candidates ranked with v_mbcnt under the exec mask
//   2  a 2-bit PACKED resident form of clean (ACGT-only) records: no table, no LDS read -- the forward word is a
//      window of the packed stream (two v_alignbit + one v_and), the reverse-complement word a window of the
//      complemented, digit-reversed stream (made per 16 bases: v_bfrev + 5 more), same hash and push
//   3  form 1 with the candidates ranked by the LDS instead of v_mbcnt: ds_add_rtn_u32 on a wave-private counter
//      returns every passing lane its place, the store follows one step later (software pipelined)
//   0  form 1 without the push (what the push costs)
// Same occupancy as the kernel: 1024-thread workgroups, one per CU (130 KB of dynamic LDS), 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/ubench_roll.hip -o tools/bin/ubench_roll
//   tools/bin/ubench_roll [groups_per_lane=512] [reps=5]
// Prints ns per k-mer step and CU, SIMD cycles per step at the clock measured inside the kernel, and each form
// relative to form 1.  (Includes the product source for its helpers; nothing here is part of the product.)
#include "../niqki_amd/csrc/nq_sketch.hip"

#include <cstdio>
#include <vector>

using namespace nq;

__device__ unsigned long long ub_clk[2];
constexpr uint32_t kRegion = 32;   // groups of 16 bases a lane's input region holds (re-read: cache resident)

template <int VAR>
__global__ __launch_bounds__(1024) void roll_kernel(const uint8_t *bytes, const uint32_t *packed, uint32_t groups, uint32_t thr,
                                                    uint32_t *sink) {
  extern __shared__ __align__(16) uint32_t smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint2 *lut64 = (uint2 *)smem;
  for (uint32_t i = tid; i < 256; i += 1024) {
    const uint32_t e = code_entry(i);
    lut64[i] = make_uint2(e & 3u, ((e >> 2) & 3u) << 28);
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  // wave-private candidate area: 8 KB per wave from LDS byte 4096 on (16 steps x 64 lanes x 8 bytes: never overrun
  // between two resets); form 3 keeps its counter in the word in front of it
  const uint32_t bottom = __builtin_amdgcn_readfirstlane(4096u + wave * 8192u);
  uint32_t top = bottom;
  const uint32_t ctr_addr = 2048u + wave * 4u;   // form 3: LDS address of the wave's counter
  if (VAR == 3 && lane == 0) *(lds_u32_t *)(uintptr_t)ctr_addr = bottom;
  __syncthreads();
  const uint64_t lane_id = (uint64_t)blockIdx.x * 1024 + tid;
  uint64_t fw = lane_id * 0x9E3779B97F4A7C15ULL & ((1ULL << 62) - 1), rc = ~fw & ((1ULL << 62) - 1);
  uint32_t acc = 0;
  if (VAR == 2) {
    // packed stream of this lane: `groups` words (16 bases each, first base in the top bits)
    // (all workgroups read the same 1024 lane regions of 32 groups: the inputs stay in L2, the loop is what is timed)
    const uint32_t *pw = packed + (uint64_t)tid * (kRegion + 4);
    uint32_t w2 = pw[0], w1 = pw[1];                 // the two words before the current one
    auto comp_rev = [](uint32_t w) {                 // complemented bases, digit order reversed
      uint32_t x = __builtin_bitreverse32(w);
      x = ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
      return ~x;
    };
    uint32_t v2 = comp_rev(w2), v1 = comp_rev(w1);
    for (uint32_t g = 0; g < groups; ++g) {
      const uint32_t wc = pw[(g & (kRegion - 1)) + 2];
      const uint32_t vc = comp_rev(wc);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // forward word: the 62 bits of [w2:w1:wc] that end at base i of wc
        const uint32_t flo = __builtin_amdgcn_alignbit(w1, wc, 30 - 2 * i);
        const uint32_t fhi = __builtin_amdgcn_alignbit(w2, w1, 30 - 2 * i) & 0x3FFFFFFFu;
        // reverse-complement word: 62 bits of the complemented little-endian stream from the k-mer's first base on
        uint32_t rlo, rhi;
        if (i <= 13) {
          rlo = __builtin_amdgcn_alignbit(v1, v2, 2 * (i + 2));
          rhi = __builtin_amdgcn_alignbit(vc, v1, 2 * (i + 2)) & 0x3FFFFFFFu;
        } else {
          rlo = __builtin_amdgcn_alignbit(vc, v1, 2 * (i - 14));
          rhi = (vc >> (2 * (i - 14))) & 0x3FFFFFFFu;
        }
        fw = ((uint64_t)fhi << 32) | flo;
        rc = ((uint64_t)rhi << 32) | rlo;
        const uint64_t canon = fw < rc ? fw : rc;
        push_candidates(rev64_hi_mad(canon), thr, canon, top);
      }
      if (top >= bottom + 4096u) top = bottom;
      w2 = w1; w1 = wc; v2 = v1; v1 = vc;
    }
  } else {
    const uint8_t *base = bytes + (uint64_t)tid * (kRegion * 16 + 64) + 1;   // (an odd start: the byte re-alignment is part of the loop)
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
    // form 3: the step before this one's mask, place and candidate (the store trails the atomic by one step)
    uint64_t p_mask = 0, p_canon = 0;
    uint32_t p_at = 0;
    auto step = [&](uint64_t ent) {
      fw = shl2_64(fw);
      fw = (fw | (uint32_t)ent) & ((1ULL << 62) - 1ULL);
      rc = shr2_64(rc) | (ent & 0xFFFFFFFF00000000ULL);
      const uint64_t canon = fw < rc ? fw : rc;
      const uint32_t hh = rev64_hi_mad(canon);
      if (VAR == 1) {
        push_candidates(hh, thr, canon, top);
      } else if (VAR == 3) {
        uint32_t at;
        uint64_t mask;
        const uint32_t eight = 8u;
        asm volatile(
            // the store of the step before, under its mask: its place has had a whole step to come back
            "s_mov_b64 exec, %[pm]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_write_b64 %[pat], %[pc]\n\t"
            "s_mov_b64 exec, -1\n\t"
            // this step: the passing lanes take their places from the wave's counter
            "v_cmp_gt_u32 vcc, %[thr], %[hh]\n\t"
            "s_mov_b64 %[m], vcc\n\t"
            "s_and_saveexec_b64 %[pm], vcc\n\t"
            "ds_add_rtn_u32 %[at], %[ctr], %[eight]\n\t"
            "s_mov_b64 exec, %[pm]\n\t"
            : [at] "=&v"(at), [m] "=&s"(mask), [pm] "+s"(p_mask)
            : [thr] "s"(thr), [hh] "v"(hh), [pat] "v"(p_at), [pc] "v"(p_canon), [ctr] "v"(ctr_addr), [eight] "v"(eight)
            : "vcc", "scc", "memory");
        p_mask = mask;
        p_at = at;
        p_canon = canon;
      } else {
        acc += hh < thr;
      }
    };
    for (uint32_t g = 0; g < groups; ++g) {
      qa = qa0 + 4 * ((g + 1) & (kRegion - 1));
      const uint4 A = *(const uint4 *)qa;
      const uint32_t B = qa[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) step(e[j]);
      uint4 w;
      w.x = __builtin_amdgcn_alignbyte(A.y, A.x, sh);
      w.y = __builtin_amdgcn_alignbyte(A.z, A.y, sh);
      w.z = __builtin_amdgcn_alignbyte(A.w, A.z, sh);
      w.w = __builtin_amdgcn_alignbyte(B, A.w, sh);
      lut64_8<0>(w, 0, e);
#pragma unroll
      for (int j = 8; j < 16; ++j) step(e[j]);
      lut64_8<1>(w, 0, e);
      if (VAR == 1 && top >= bottom + 4096u) top = bottom;
      if (VAR == 3) {   // 16 steps x <= 64 places of 8 bytes: back to the bottom (all lanes write the same word)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        *(lds_u32_t *)(uintptr_t)ctr_addr = bottom;
      }
    }
    acc += (uint32_t)p_canon + p_at;
  }
  acc += (uint32_t)fw ^ (uint32_t)rc ^ top;
  if (acc == 0x12345u) sink[0] = acc;
  if (blockIdx.x == 0 && tid == 0) {
    ub_clk[0] = __builtin_readcyclecounter() - c0;
    ub_clk[1] = wall_clock64() - r0;
  }
}

__global__ void fill_kernel(uint8_t *bytes, uint32_t *packed, uint64_t n_bytes, uint64_t n_words) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  if (i < n_words) packed[i] = (uint32_t)z;
  if (i * 16 < n_bytes)
    for (int j = 0; j < 16 && i * 16 + j < n_bytes; ++j) bytes[i * 16 + j] = (uint8_t)((0x54474341u >> (8 * ((z >> (2 * j)) & 3))) & 0xFFu);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int VAR>
static int run(const uint8_t *bytes, const uint32_t *packed, uint32_t groups, int reps, uint32_t *sink, double *ns_per_step, double *cyc) {
  const size_t lds = 136 * 1024;   // table + counters + 16 wave areas of 8 KB: one workgroup per CU, as the kernel
  CK(hipFuncSetAttribute((const void *)roll_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(roll_kernel<VAR>, dim3(256), dim3(1024), lds, 0, bytes, packed, groups, 1u << 29, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(roll_kernel<VAR>, dim3(256), dim3(1024), lds, 0, bytes, packed, groups, 1u << 29, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long clk[2];
  CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(ub_clk), 16));
  const double steps_per_lane = (double)groups * 16, ghz = clk[0] / (clk[1] * 10.0);
  // one CU: 16 waves, 4 per SIMD: SIMD cycles per wave step = kernel time x clock / (steps per lane x 4 waves)
  *ns_per_step = ms / reps * 1e6 / (steps_per_lane * 1024);
  *cyc = ms / reps * 1e-3 * ghz * 1e9 / (steps_per_lane * 4);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return 0;
}

int main(int argc, char **argv) {
  const uint32_t groups = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const uint64_t lanes = 1024;
  const uint64_t n_bytes = lanes * (kRegion * 16 + 64) + 4096, n_words = lanes * (kRegion + 4) + 64;
  uint8_t *bytes; uint32_t *packed, *sink;
  CK(hipMalloc(&bytes, n_bytes));
  CK(hipMalloc(&packed, n_words * 4));
  CK(hipMalloc(&sink, 256));
  const uint64_t n_fill = n_bytes / 16 + 1 > n_words ? n_bytes / 16 + 1 : n_words;
  hipLaunchKernelGGL(fill_kernel, dim3((uint32_t)((n_fill + 255) / 256)), dim3(256), 0, 0, bytes, packed, n_bytes, n_words);
  CK(hipDeviceSynchronize());
  double ns[4], cy[4];
  if (run<1>(bytes, packed, groups, reps, sink, &ns[1], &cy[1])) return 1;
  if (run<0>(bytes, packed, groups, reps, sink, &ns[0], &cy[0])) return 1;
  if (run<2>(bytes, packed, groups, reps, sink, &ns[2], &cy[2])) return 1;
  if (run<3>(bytes, packed, groups, reps, sink, &ns[3], &cy[3])) return 1;
  if (run<1>(bytes, packed, groups, reps, sink, &ns[1], &cy[1])) return 1;   // (again, after everything is warm)
  const char *name[4] = {"0  byte table, no push", "1  byte table + mbcnt push (product)", "2  2-bit packed windows + mbcnt push",
                         "3  byte table + LDS-ranked push"};
  printf("hot loop of sketch_kernel<1024,32,31> in isolation: %u groups of 16 k-mer steps per lane, 256 workgroups x 1024 threads, no drains\n", groups);
  printf("%-42s %14s %22s %10s\n", "form", "ps/step/lane", "SIMD cycles/wave step", "vs form 1");
  for (int v : {1, 0, 2, 3})
    printf("%-42s %14.2f %22.1f %10.3f\n", name[v], ns[v] * 1e3, cy[v], ns[v] / ns[1]);
  return 0;
}
