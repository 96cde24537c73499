// HBM write rate by pattern: (a) linear fill, (b) the tile epilogue's pattern (256 x 256 bf16 tiles of a 65536 x N matrix, XCD-aware
// tile order, 16-byte lanes covering 512-byte row segments, 64-row chunks), (c) one 8 KiB row per 256-thread block.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_write.hip -o /tmp/pw && /tmp/pw
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void fill_linear(uint4* p, long long n16) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ __launch_bounds__(1024) void fill_tiles(unsigned short* C, long long M, long long N, int rounds_sleep) {
  const int ntn = (int)(N / 256), nt = (int)(M / 256) * ntn;
  const int bid = blockIdx.x, q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const long long tm0 = (long long)(id / ntn) * 256, tn0 = (long long)(id % ntn) * 256;
  const int tid = threadIdx.x, c8 = (tid & 31) * 8;
  for (int w = 0; w < rounds_sleep; w++) __builtin_amdgcn_s_sleep(127);   // stand-in for the main loop
  for (int qq = 0; qq < 4; qq++) {
    __syncthreads();
    for (int rr = 0; rr < 2; rr++) {
      const long long row = tm0 + qq * 64 + (tid >> 5) + 32 * rr;
      *(uint4*)(C + row * N + tn0 + c8) = make_uint4(1, 2, 3, 4);
    }
  }
}
__global__ void fill_rows(uint4* p, int v16) {  // one row per block
  uint4* r = p + (long long)blockIdx.x * v16;
  for (int c = threadIdx.x; c < v16; c += blockDim.x) r[c] = make_uint4(1, 2, 3, 4);
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int i = 0; i < 5; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const long long M = 65536;
  unsigned short* C; hipMalloc(&C, M * 4096 * 2);
  for (long long N : {512LL, 2048LL, 4096LL}) {
    const double bytes = (double)M * N * 2;
    float a = timeit([&] { hipLaunchKernelGGL(fill_linear, dim3(2048), dim3(256), 0, 0, (uint4*)C, M * N * 2 / 16); });
    float b = timeit([&] { hipLaunchKernelGGL(fill_tiles, dim3((unsigned)(M / 256 * N / 256)), dim3(1024), 0, 0, C, M, N, 0); });
    float b2 = timeit([&] { hipLaunchKernelGGL(fill_tiles, dim3((unsigned)(M / 256 * N / 256)), dim3(1024), 0, 0, C, M, N, 1); });
    float c = timeit([&] { hipLaunchKernelGGL(fill_rows, dim3((unsigned)M), dim3(256), 0, 0, (uint4*)C, (int)(N * 2 / 16)); });
    printf("N %4lld (%4.0f MB): linear %6.1f us %5.2f TB/s | 256x256 tiles %6.1f us %5.2f TB/s | tiles + 3.4 us of sleep per tile %6.1f us | one row per block %6.1f us %5.2f TB/s\n",
           N, bytes / 1e6, a * 1e3, bytes / a / 1e9, b * 1e3, bytes / b / 1e9, b2 * 1e3, c * 1e3, bytes / c / 1e9);
  }
  return 0;
}
