// How long is one workgroup barrier?  Empty loops of s_barrier for several workgroup sizes, one workgroup per CU.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probe_barrier.hip -o /tmp/pb && /tmp/pb
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(long long* out, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  long long* d; hipMalloc(&d, 8);
  for (int threads : {64, 256, 512, 1024}) {
    for (int blocks : {256, 512}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 10000;
      hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, 10);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
      printf("threads %4d blocks %3d: %.1f ns per barrier (event), %.1f counter ticks per barrier\n", threads, blocks, ms * 1e6 / iters, (double)cyc / iters);
    }
  }
  return 0;
}
