// nq_synth.hip -- device side of the synthetic genome generator (nq_synth.h).
// Measurement / parity input only.
#include "nq_kernels.h"
#include "nq_synth.h"

namespace nq {

// one thread per 32-base block
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, const uint32_t *family,
                                                   const uint32_t *member, const uint32_t *rate14,
                                                   uint64_t len, uint64_t stride, uint8_t *out) {
  const uint32_t g = blockIdx.y;
  const uint64_t n_blocks = (len + 31) / 32;
  const uint64_t ka = synth_key_anc(seed, family[g]);
  const uint64_t km = synth_key_mut(seed, family[g], member[g]);
  const uint32_t r = rate14[g];
  uint8_t *dst = out + (uint64_t)g * stride;
  const bool aligned = (((uintptr_t)dst) & 15u) == 0;
  for (uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < n_blocks;
       blk += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t codes = synth_block(ka, km, r, blk);
    uint64_t p0 = blk * 32;
    if (aligned && p0 + 32 <= len) {
      uint32_t w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          x |= (uint32_t)synth_ascii((uint32_t)(codes >> (2 * (4 * k + j))) & 3u) << (8 * j);
        w[k] = x;
      }
      uint4 *o = (uint4 *)(dst + p0);
      o[0] = make_uint4(w[0], w[1], w[2], w[3]);
      o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    } else {
      for (uint32_t j = 0; j < 32 && p0 + j < len; ++j)
        dst[p0 + j] = synth_ascii((uint32_t)(codes >> (2 * j)) & 3u);
    }
  }
}

// one thread per read: bases [offset, offset + len) of genome (family, member, rate14) with the
// read's own substitutions (read_id, read_rate14) on top
__global__ __launch_bounds__(256) void synth_reads_kernel(uint64_t seed, const uint32_t *family, const uint32_t *member,
                                                         const uint32_t *rate14, const uint64_t *offset,
                                                         const uint32_t *read_id, uint32_t read_rate14, uint32_t n,
                                                         uint32_t len, uint64_t stride, uint8_t *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t ka = synth_key_anc(seed, family[i]);
  const uint64_t km = synth_key_mut(seed, family[i], member[i]);
  const uint64_t kr = synth_key_read(seed, family[i], read_id[i]);
  const uint64_t p0 = offset[i];
  uint8_t *dst = out + (uint64_t)i * stride;
  uint64_t blk = ~0ull, codes = 0;
  for (uint32_t j = 0; j < len; ++j) {
    const uint64_t p = p0 + j;
    if ((p >> 5) != blk) { blk = p >> 5; codes = synth_block2(ka, km, rate14[i], kr, read_rate14, blk); }
    dst[j] = synth_ascii((uint32_t)(codes >> (2 * (p & 31))) & 3u);
  }
}

hipError_t launch_synth_reads(uint64_t seed, const uint32_t *family, const uint32_t *member, const uint32_t *rate14,
                              const uint64_t *offset, const uint32_t *read_id, uint32_t read_rate14, uint32_t n,
                              uint32_t len, uint64_t stride, uint8_t *out, hipStream_t stream) {
  if (n == 0 || len == 0) return hipSuccess;
  hipLaunchKernelGGL(synth_reads_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, seed, family, member, rate14, offset,
                     read_id, read_rate14, n, len, stride, out);
  return hipGetLastError();
}

hipError_t launch_synth(uint64_t seed, const uint32_t *family, const uint32_t *member,
                        const uint32_t *rate14, uint32_t n, uint64_t len, uint64_t stride,
                        uint8_t *out, hipStream_t stream) {
  if (n == 0 || len == 0) return hipSuccess;
  uint64_t n_blocks = (len + 31) / 32;
  uint64_t bx = (n_blocks + 255) / 256;
  if (bx > 1024) bx = 1024;
  for (uint32_t g0 = 0; g0 < n; g0 += 65535) {
    uint32_t gy = (n - g0) < 65535 ? (n - g0) : 65535;
    hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)bx, gy), dim3(256), 0, stream, seed, family + g0,
                       member + g0, rate14 + g0, len, stride, out + (uint64_t)g0 * stride);
  }
  return hipGetLastError();
}

}  // namespace nq
