// How fast does this box page-lock host memory?  (the reader buffers of niqki_amd/host are page-locked: the first
// phase of a run pays for it)   hipcc -O2 tools/ubench_pin.cpp -o /tmp/ubench_pin && /tmp/ubench_pin
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipSetDevice(0);
  void *warm;
  hipHostMalloc(&warm, 1 << 20, hipHostMallocDefault);
  const size_t piece = size_t(1600) << 10, n = 1024, total = piece * n;
  {   // (1) one hipHostMalloc per buffer, one thread
    std::vector<void *> p(n);
    const double t0 = now();
    for (size_t i = 0; i < n; ++i) hipHostMalloc(&p[i], piece, hipHostMallocDefault);
    const double t1 = now();
    printf("hipHostMalloc x %zu of %zu KB, 1 thread: %.3f s = %.2f GB/s\n", n, piece >> 10, t1 - t0, total / (t1 - t0) / 1e9);
    for (auto q : p) hipHostFree(q);
  }
  {   // (2) the same from 8 threads
    std::vector<void *> p(n);
    const double t0 = now();
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t) th.emplace_back([&, t] { for (size_t i = t; i < n; i += 8) hipHostMalloc(&p[i], piece, hipHostMallocDefault); });
    for (auto &x : th) x.join();
    const double t1 = now();
    printf("hipHostMalloc x %zu, 8 threads: %.3f s = %.2f GB/s\n", n, t1 - t0, total / (t1 - t0) / 1e9);
    for (auto q : p) hipHostFree(q);
  }
  {   // (3) one slab
    void *p;
    const double t0 = now();
    hipHostMalloc(&p, total, hipHostMallocDefault);
    const double t1 = now();
    printf("hipHostMalloc of one %.2f GB slab: %.3f s = %.2f GB/s\n", total / 1e9, t1 - t0, total / (t1 - t0) / 1e9);
    hipHostFree(p);
  }
  for (int huge = 0; huge < 2; ++huge) {   // (4) mmap (+ MADV_HUGEPAGE), touch, hipHostRegister
    const double t0 = now();
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (huge) madvise(p, total, MADV_HUGEPAGE);
    for (size_t i = 0; i < total; i += 4096) ((volatile char *)p)[i] = 1;
    const double t1 = now();
    const hipError_t e = hipHostRegister(p, total, hipHostRegisterDefault);
    const double t2 = now();
    printf("mmap%s + touch %.3f s, hipHostRegister %.3f s (%s) = %.2f GB/s in all\n", huge ? " + MADV_HUGEPAGE" : "", t1 - t0, t2 - t1,
           hipGetErrorString(e), total / (t2 - t0) / 1e9);
    if (e == hipSuccess) hipHostUnregister(p);
    munmap(p, total);
  }
  FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  if (f) { char b[128] = {0}; if (fgets(b, 127, f)) printf("transparent_hugepage/enabled: %s", b); fclose(f); }
  return 0;
}
