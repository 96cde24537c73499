// Chip-wide store throughput by lane -> address pattern, in the GEMM's tile order: 256 workgroups x 8 waves write 2048 x 8 tiles of 256 rows x 512 B
// (a 524 288 x 2048 bf16 matrix, 2.15 GB) and nothing else.  PAT 0: 16 rows x 64 B per wave-instruction (the epilogue's quads), the two halves of a
// 128-byte line in consecutive instructions; 1: 8 rows x 128 B per instruction (whole lines); 2: as 0 but all first halves, then all second halves.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store4.hip -o tools/probe_store4.bin && tools/probe_store4.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned char* C, long long pitch, int ntn, int nt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3;
  const u4v v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  const int q8 = nt >> 3;
  for (int T = blockIdx.x; T < nt; T += gridDim.x) {
    const int id = (T & 7) * q8 + (T >> 3);
    unsigned char* tile = C + (long long)(id / ntn) * 256 * pitch + (long long)(id % ntn) * 512;
    unsigned char* wbase = tile + (long long)(128 * wr) * pitch + 128 * wc;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      unsigned char* a;
      if (PAT == 0) a = wbase + (long long)(16 * (i >> 1) + (lane >> 2)) * pitch + 64 * (i & 1) + 16 * (lane & 3);
      else if (PAT == 1) a = wbase + (long long)(8 * i + (lane >> 3)) * pitch + 16 * (lane & 7);
      else a = wbase + (long long)(16 * (i & 7) + (lane >> 2)) * pitch + 64 * (i >> 3) + 16 * (lane & 3);
      *(u4v*)a = v;
    }
  }
}
int main() {
  const long long M = 524288, pitch = 4096;
  unsigned char* C; hipMalloc(&C, (size_t)M * pitch);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ntn = 8, nt = (int)(M / 256) * ntn;
#define RUN(P_, G_)                                                                                          \
  do {                                                                                                       \
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k<P_>, dim3(G_), dim3(512), 0, 0, C, pitch, ntn, nt);   \
    hipEventRecord(e0);                                                                                      \
    for (int it = 0; it < 5; it++) hipLaunchKernelGGL(k<P_>, dim3(G_), dim3(512), 0, 0, C, pitch, ntn, nt);   \
    hipEventRecord(e1); hipEventSynchronize(e1);                                                             \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                              \
    printf("pattern %d, %d workgroups: %.0f us per 2.15 GB = %.2f TB/s\n", P_, G_, ms / 5 * 1e3, (double)M * pitch / (ms / 5 * 1e-3) / 1e12); \
  } while (0)
  for (int rep = 0; rep < 2; rep++) { RUN(0, 256); RUN(1, 256); RUN(2, 256); RUN(0, 512); RUN(1, 512); }
  return 0;
}
