// What makes the GEMM epilogue's stores slow?  One workgroup of 8 waves per CU; every wave: [optionally PRE LDS-DMA loads of fresh
// HBM data, as the kernel's stream has in flight], then 16 x 16-byte-per-lane stores covering 16 rows x 64 B each (the register
// epilogue's pattern), repeated REP times with a workgroup barrier in between.  Prints cycles per store instruction per CU for the
// ISSUE of the stores (stamp right behind the last one) and for their completion (after s_waitcnt vmcnt(0)).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store2.hip -o tools/probe_store2.bin && tools/probe_store2.bin [workgroups=8]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef int i4v __attribute__((ext_vector_type(4)));
template <int PRE, int ASM>
__global__ __launch_bounds__(512) void k(unsigned char* C, const unsigned char* src, long long pitch, int reps, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  unsigned long long ti = 0, tc = 0;
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const u4v v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  for (int r = 0; r < reps; r++) {
    unsigned char* tile = C + ((long long)blockIdx.x * reps + r) * 256 * pitch;   // a fresh 256-row x 512-byte region per repeat
    const unsigned char* s = src + (((long long)blockIdx.x * reps + r) * 8 + wave) * PRE * 1024 + (PRE ? 0 : 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < PRE; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(smem + (wave * 16 + i) * 1024), 16, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned char* base = tile + (long long)((wave >> 2) * 128 + li) * pitch + (wave & 3) * 128 + lq * 16;
    if (ASM) {
      const unsigned long long b = (unsigned long long)tile;
      i4v rs; rs[0] = (int)(unsigned)b; rs[1] = (int)(unsigned)((b >> 32) & 0xffffu); rs[2] = (int)(256 * pitch); rs[3] = 0x00020000;
      const unsigned vo = (unsigned)(((wave >> 2) * 128 + li) * pitch + (wave & 3) * 128 + lq * 16);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int so = (int)((i >> 1) * 16 * pitch);
        if (i & 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:64\n\ts_nop 2" :: "v"(v), "v"(vo), "s"(rs), "s"(so) : "memory");
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:0\n\ts_nop 2" :: "v"(v), "v"(vo), "s"(rs), "s"(so) : "memory");
      }
    } else if (ASM == 2) {   // every instruction its own 16 rows (half lines, the other half never written)
      unsigned char* b2 = tile + (long long)li * pitch + wave * 64 + lq * 16;
#pragma unroll
      for (int i = 0; i < 16; i++) *(u4v*)(b2 + (long long)i * 16 * pitch) = v;
    } else if (ASM == 3) {   // 8 rows x 128 B per instruction (whole lines): lane -> row lane >> 3, 16-byte piece lane & 7
      unsigned char* b3 = tile + (long long)((wave >> 2) * 128 + (lane >> 3)) * pitch + (wave & 3) * 128 + (lane & 7) * 16;
#pragma unroll
      for (int i = 0; i < 16; i++) *(u4v*)(b3 + (long long)i * 8 * pitch) = v;
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) *(u4v*)(base + (long long)(i >> 1) * 16 * pitch + (i & 1) * 64) = v;
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (r) { ti += t2 - t1; tc += t3 - t0; }   // the first repeat warms the instruction cache and the TLB
  }
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = ti; out[(blockIdx.x * 8 + wave) * 2 + 1] = tc; }
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 8;
  const int reps = 8;
  const long long pitch = 4096;
  unsigned char *C, *src; hipMalloc(&C, (size_t)G * reps * 256 * pitch); hipMalloc(&src, (size_t)G * reps * 8 * 16 * 1024 + 4096);
  unsigned long long* out; hipMalloc(&out, G * 8 * 2 * 8);
  std::vector<unsigned long long> h(G * 16);
#define RUN(PRE_, ASM_, name_)                                                                                      \
  do {                                                                                                              \
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<PRE_, ASM_>), dim3(G), dim3(512), 128 * 1024, 0, C, src, pitch, reps, out); \
    hipDeviceSynchronize();                                                                                         \
    hipMemcpy(h.data(), out, G * 16 * 8, hipMemcpyDeviceToHost);                                                    \
    std::vector<double> a, b;                                                                                       \
    for (int i = 0; i < G * 8; i++) { a.push_back((double)h[2 * i] / (reps - 1)); b.push_back((double)h[2 * i + 1] / (reps - 1)); }  \
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());                                                   \
    printf("%-58s issue of a wave's 16 stores: median %6.0f max %6.0f cycles; loads + stores complete: median %6.0f (= %.1f per store instruction per CU)\n", \
           name_, a[a.size() / 2], a.back(), b[b.size() / 2], b[b.size() / 2] / 128.0);                             \
  } while (0)
  hipFuncSetAttribute((const void*)k<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<14, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<14, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipFuncSetAttribute((const void*)k<14, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  printf("%d workgroups\n", G);
  RUN(0, 0, "global_store_dwordx4, nothing in flight");
  RUN(0, 1, "buffer_store_dwordx4 offen + SGPR offset + s_nop 2");
  RUN(2, 1, "the same behind 2 LDS-DMA loads per wave (HBM)");
  RUN(14, 1, "the same behind 14 LDS-DMA loads per wave (HBM)");
  RUN(14, 0, "global_store_dwordx4 behind 14 LDS-DMA loads per wave");
  RUN(0, 2, "global stores, 16 rows x 64 B, every instruction other rows");
  RUN(0, 3, "global stores, 8 rows x 128 B (whole lines) per instruction");
  RUN(14, 3, "whole-line stores behind 14 LDS-DMA loads per wave");
  return 0;
}
