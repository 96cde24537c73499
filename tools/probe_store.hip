// Store-instruction throughput of one CU by access pattern: 8 waves per workgroup, one workgroup per CU, every wave issues R
// buffer-less global_store_dwordx4 (16 B per lane) whose 64 lanes cover rows of `pitch` bytes in segments of 16 * LPR bytes
// (LPR lanes per row: 4 = the eight-phase GEMM's register epilogue, 64 B per row; 64 = 1 KiB contiguous).  Prints cycles per
// store instruction per CU (s_memtime, median over the workgroups) and the chip's write rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store.hip -o /tmp/ps && /tmp/ps [workgroups=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
template <int LPR, int W>   // W = 4: dwordx4, 2: dwordx2
__global__ __launch_bounds__(512) void st_k(unsigned char* C, long long pitch, int R, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane / LPR, seg = lane % LPR;
  constexpr int RPI = 64 / LPR;  // rows per instruction
  // workgroup b owns a 256-row x 512-byte tile region at a distinct place; wave w owns 128 rows (w >> 2) x 128 B columns (w & 3) in
  // the 4-lane case; for longer segments the wave simply walks down its own rows
  unsigned char* base = C + (long long)blockIdx.x * 256 * pitch + (long long)wave * (16 * LPR);
  uint4 v = make_uint4(lane, wave, 3, 4);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < R; i++) {
    unsigned char* p = base + (long long)((i * RPI + row) % 256) * pitch + seg * 16 + (long long)((i * RPI) / 256) * 8 * 16 * LPR;
    if (W == 4) *(uint4*)p = v; else *(uint2*)p = make_uint2(v.x, v.y);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  const long long pitch = 4096;
  unsigned char* C; hipMalloc(&C, (size_t)256 * 256 * pitch + (1 << 20));
  unsigned long long* cyc; hipMalloc(&cyc, G * 8);
  std::vector<unsigned long long> h(G);
  const int R = 64;  // per wave: 64 KiB at dwordx4
#define RUN(LPR_, W_)                                                                                               \
  do {                                                                                                              \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);                                                    \
    hipLaunchKernelGGL((st_k<LPR_, W_>), dim3(G), dim3(512), 0, 0, C, pitch, R, cyc);                               \
    hipEventRecord(e0);                                                                                             \
    for (int it = 0; it < 5; it++) hipLaunchKernelGGL((st_k<LPR_, W_>), dim3(G), dim3(512), 0, 0, C, pitch, R, cyc); \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                                    \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;                                                            \
    hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost);                                                         \
    std::sort(h.begin(), h.end());                                                                                  \
    const double bytes = (double)G * 8 * R * 64 * 4 * W_;                                                          \
    printf("%2d-byte lanes, %4d B per row segment (%2d rows / instruction): %6.1f cycles per store instruction per CU, %5.1f B/clk/CU, "  \
           "chip %.2f TB/s (%d workgroups)\n", 4 * W_, 4 * W_ * LPR_, 64 / LPR_, (double)h[G / 2] / (8.0 * R), 8.0 * R * 64 * 4 * W_ / (double)h[G / 2], \
           bytes / ms * 1e-9, G);                                                                                   \
  } while (0)
  RUN(4, 4); RUN(8, 4); RUN(16, 4); RUN(32, 4); RUN(64, 4);
  RUN(4, 2); RUN(8, 2); RUN(64, 2);
  return 0;
}
