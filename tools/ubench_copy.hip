// ubench_copy.hip -- streaming copy rates on gfx950 (plain / unrolled / nontemporal kernels, hipMemcpyAsync): the practical HBM ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const u4 *src, u4 *dst, uint64_t n) {
  const uint64_t step = (uint64_t)gridDim.x * 256 * U;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; i < n; i += step) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
  }
}
template <int U, bool NT> void run(const char*name, int blocks, void*a, void*b, uint64_t bytes) {
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<U,NT><<<blocks,256>>>((const u4*)a,(u4*)b,bytes/16);
  hipEventRecord(e0);
  for (int r=0;r<8;++r) k<U,NT><<<blocks,256>>>((const u4*)a,(u4*)b,bytes/16);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
  printf("%s U=%d NT=%d blocks=%d: %.0f GB/s\n", name, U, (int)NT, blocks, 16.0*bytes/(ms*1e-3)/1e9);
}
int main(){ uint64_t bytes=1ull<<30; void*a,*b; hipMalloc(&a,bytes); hipMalloc(&b,bytes); hipMemset(a,1,bytes);
 for (int blocks : {2048, 8192, 32768}) { run<1,false>("copy",blocks,a,b,bytes); run<4,false>("copy",blocks,a,b,bytes); run<4,true>("copy",blocks,a,b,bytes); run<8,true>("copy",blocks,a,b,bytes);}
 hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1); hipMemcpyAsync(b,a,bytes,hipMemcpyDeviceToDevice,0); hipEventRecord(e0); for(int r=0;r<8;++r) hipMemcpyAsync(b,a,bytes,hipMemcpyDeviceToDevice,0); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); printf("hipMemcpyAsync D2D: %.0f GB/s\n", 16.0*bytes/(ms*1e-3)/1e9);
 return 0; }
