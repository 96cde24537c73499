// Pure LDS-DMA stream: how fast can a CU pull a 256-row panel of a row-major bf16 matrix into LDS, as a function of the
// row-segment width per k-step (64 B = BK 32, 128 B = BK 64, 256 B = BK 128), the number of stages in flight and the
// number of workgroups that read the same panel (L2 reuse)?  No compute.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o /tmp/pd && /tmp/pd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int W, int STAGES, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const unsigned char* X, long long ld_bytes, int ksteps, int share, int rows_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int WAVES = THREADS / 64;
  constexpr int LPR = W / 16;                 // lanes per row segment
  constexpr int RPI = 64 / LPR;               // rows per wave instruction
  // the `share` workgroups of a panel run on ONE XCD (blocks b, b+8, ... share an XCD and its L2)
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int panel = (loc / share) * 8 + xcd;
  const unsigned char* base = X + (long long)panel * rows_per_wg * ld_bytes;
  const int stage_bytes = rows_per_wg * W;
  const int instr_per_wave = rows_per_wg / (RPI * WAVES);
  auto issue = [&](int t) {
    unsigned char* dst = smem + (t % STAGES) * stage_bytes;
    for (int i = 0; i < instr_per_wave; i++) {
      const int piece = i * WAVES + wave;
      const int row = piece * RPI + lane / LPR;
      const unsigned char* src = base + (long long)row * ld_bytes + (long long)t * W + (lane % LPR) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
  };
  for (int s = 0; s < STAGES - 1; s++) issue(s);
  for (int t = 0; t < ksteps; t++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // simple: drain, barrier, refill (STAGES - 1 stages were in flight)
    __builtin_amdgcn_s_barrier();
    if (t + STAGES - 1 < ksteps) issue(t + STAGES - 1);
  }
  if (sink && tid == 0) sink[blockIdx.x] = *(unsigned*)smem;
}
template <int W, int STAGES, int THREADS>
void run(const unsigned char* X, long long M, long long Kbytes, int share, unsigned* sink, size_t pad = 0) {
  const int rows = 256;
  const int wgs = (int)(M / rows) * share;
  const int ksteps = (int)(Kbytes / W);
  const size_t lds = (size_t)STAGES * rows * W + pad;
  hipFuncSetAttribute((const void*)k<W, STAGES, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<W, STAGES, THREADS>), dim3(wgs), dim3(THREADS), lds, 0, X, Kbytes, ksteps, share, rows, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<W, STAGES, THREADS>), dim3(wgs), dim3(THREADS), lds, 0, X, Kbytes, ksteps, share, rows, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)M * Kbytes * share;
  printf("W %3d B  stages %d  threads %4d  share %d  lds %3zu KiB: %7.1f us  %6.2f TB/s into LDS (%.0f MB unique)\n", W, STAGES, THREADS, share,
         lds / 1024, ms * 1e3, bytes / ms / 1e9, (double)M * Kbytes / 1e6);
}
int main() {
  const long long M = 65536, Kbytes = 4096;  // 65536 x 2048 bf16 = 268 MB
  unsigned char* X; hipMalloc(&X, M * Kbytes + 4096); hipMemset(X, 1, M * Kbytes + 4096);
  unsigned* sink; hipMalloc(&sink, 4 * 4096);
  for (int share : {2, 4}) {
    printf("-- W 128, 2 stages, two workgroups per CU (64 KiB each) vs one (padded to 96 KiB), by workgroup size\n");
    run<128, 2, 1024>(X, M, Kbytes, share, sink);
    run<128, 2, 1024>(X, M, Kbytes, share, sink, 32768);
    run<128, 2, 512>(X, M, Kbytes, share, sink);
    run<128, 2, 512>(X, M, Kbytes, share, sink, 32768);
    run<128, 2, 256>(X, M, Kbytes, share, sink);
    run<128, 2, 256>(X, M, Kbytes, share, sink, 32768);
    printf("-- W 64, 2 stages (32 KiB): up to 5 workgroups per CU; padded to 64 / 96 KiB: 2 / 1\n");
    run<64, 2, 512>(X, M, Kbytes, share, sink);
    run<64, 2, 512>(X, M, Kbytes, share, sink, 32768);
    run<64, 2, 512>(X, M, Kbytes, share, sink, 65536);
  }
  return 0;
}
