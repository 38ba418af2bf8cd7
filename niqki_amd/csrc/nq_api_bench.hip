// nq_api_bench.hip -- measurement support of include/niqki_hip_bench.h: the deterministic synthetic genomes / reads
// (host and device generators, identical bytes) and the integer-ALU / copy / LDS-pass probes.  Not part of the
// drop-in boundary.
#include "nq_handle.h"
#include "nq_synth.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

using namespace nqi;

extern "C" {

void niqki_synth_genome_host(uint64_t seed, uint32_t family, uint32_t member, uint32_t rate14,
                             uint64_t len, uint8_t *out) {
  const uint64_t ka = nq::synth_key_anc(seed, family), km = nq::synth_key_mut(seed, family, member);
  for (uint64_t blk = 0; blk * 32 < len; ++blk) {
    uint64_t codes = nq::synth_block(ka, km, rate14, blk);
    for (uint32_t j = 0; j < 32 && blk * 32 + j < len; ++j)
      out[blk * 32 + j] = nq::synth_ascii((uint32_t)(codes >> (2 * j)) & 3u);
  }
}

int niqki_synth_reads(niqki_index *ix, uint64_t seed, const uint32_t *family, const uint32_t *member, const uint32_t *rate14,
                      const uint64_t *offset, const uint32_t *read_id, uint32_t read_rate14, uint32_t n, uint32_t len,
                      uint64_t stride, uint8_t *out, int mem) {
  if (!ix || (n && (!family || !member || !rate14 || !offset || !read_id || !out)) || stride < len) return NIQKI_E_INVALID;
  if (mem == NIQKI_MEM_HOST) {
    for (uint32_t i = 0; i < n; ++i) {
      const uint64_t ka = nq::synth_key_anc(seed, family[i]), km = nq::synth_key_mut(seed, family[i], member[i]);
      const uint64_t kr = nq::synth_key_read(seed, family[i], read_id[i]);
      uint64_t blk = ~0ull, codes = 0;
      for (uint32_t j = 0; j < len; ++j) {
        const uint64_t p = offset[i] + j;
        if ((p >> 5) != blk) { blk = p >> 5; codes = nq::synth_block2(ka, km, rate14[i], kr, read_rate14, blk); }
        out[(uint64_t)i * stride + j] = nq::synth_ascii((uint32_t)(codes >> (2 * (p & 31))) & 3u);
      }
    }
    return NIQKI_OK;
  }
  NQ_HIP(ix, hipSetDevice(ix->device));
  NQ_HIP(ix, nq::launch_synth_reads(seed, family, member, rate14, offset, read_id, read_rate14, n, len, stride, out, ix->stream));
  return NIQKI_OK;
}

int niqki_measure_alu(niqki_index *ix, int what, double ms, double *rate) {
  if (!ix || !rate || what < 0 || what > 5 || !(ms > 0)) return NIQKI_E_INVALID;
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (what == 4) {   // streaming copy of 1 GiB: bytes read + bytes written per second
    const uint64_t bytes = 1ull << 30;
    void *a_ = nullptr, *b_ = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a_, bytes);
    if (e == hipSuccess) e = hipMalloc(&b_, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(a_, 1, bytes, ix->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = nq::launch_copy_probe(a_, b_, bytes, ix->stream);
    const int reps = 8;
    if (e == hipSuccess) e = hipEventRecord(e0, ix->stream);
    for (int r = 0; r < reps && e == hipSuccess; ++r) e = nq::launch_copy_probe(a_, b_, bytes, ix->stream);
    if (e == hipSuccess) e = hipEventRecord(e1, ix->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, e0, e1);
    if (a_) (void)hipFree(a_);
    if (b_) (void)hipFree(b_);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return fail(ix, e == hipErrorOutOfMemory ? NIQKI_E_NOMEM : NIQKI_E_HIP, std::string("copy probe: ") + hipGetErrorString(e));
    *rate = f > 0 ? 2.0 * reps * (double)bytes / (f * 1e-3) : 0.0;
    return NIQKI_OK;
  }
  int rc = ensure(ix, ix->ws_misc, 256);
  if (rc) return rc;
  hipEvent_t a = nullptr, b = nullptr;
  struct Events {   // destroyed on every way out
    hipEvent_t &a, &b;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } guard{a, b};
  NQ_HIP(ix, hipEventCreate(&a));
  NQ_HIP(ix, hipEventCreate(&b));
  auto run = [&](uint32_t iters, double &t_ms, uint64_t &units) -> hipError_t {
    hipError_t e = hipEventRecord(a, ix->stream);
    if (e == hipSuccess) e = nq::launch_alu_probe(what, iters, (uint32_t *)ix->ws_misc.p, &units, ix->stream);
    if (e == hipSuccess) e = hipEventRecord(b, ix->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float f = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&f, a, b);
    t_ms = f;
    return e;
  };
  double t = 0;
  uint64_t units = 0;
  uint32_t iters = 256;
  NQ_HIP(ix, run(iters, t, units));                       // warm-up + calibration
  iters = (uint32_t)std::min<double>(1e7, std::max<double>(256, iters * ms / std::max(t, 1e-3)));
  NQ_HIP(ix, run(iters, t, units));
  *rate = t > 0 ? (double)units / (t * 1e-3) : 0.0;
  return NIQKI_OK;
}

int niqki_synth_genomes(niqki_index *ix, uint64_t seed, const uint32_t *family, const uint32_t *member,
                        const uint32_t *rate14, uint32_t n, uint64_t len, uint64_t stride, uint8_t *out,
                        int mem) {
  if (!ix || (n && (!family || !member || !rate14 || !out)) || stride < len) return NIQKI_E_INVALID;
  if (mem == NIQKI_MEM_HOST) {
    for (uint32_t i = 0; i < n; ++i)
      niqki_synth_genome_host(seed, family[i], member[i], rate14[i], len, out + (uint64_t)i * stride);
    return NIQKI_OK;
  }
  NQ_HIP(ix, hipSetDevice(ix->device));
  NQ_HIP(ix, nq::launch_synth(seed, family, member, rate14, n, len, stride, out, ix->stream));
  return NIQKI_OK;
}

}  // extern "C"
