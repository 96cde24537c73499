// How many LDS cycles does ds_bpermute_b32 take for a given source-lane pattern?  8 waves per CU run 256 dependent-free bpermutes each.
//   pattern 0: identity; 1: the GEMM epilogue's lane transpose (dst L <- lane (L >> 2) + 16 * swap2(L & 3)): sources l and l + 32 inside one
//   32-lane destination half; 2: pairs (half 0: (L >> 1) + 16 (L & 1), half 1: 32 + ...): 32 distinct sources mod 32 per half;
//   3: quads with the upper half's rows rotated by 8 (r, r + 16, 32 + (r + 8) % 16, 48 + (r + 8) % 16)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_bperm.hip -o tools/probe_bperm.bin && tools/probe_bperm.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(int pattern, int nw, unsigned long long* out, int* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int src;
  const int er = lane >> 2, ep = lane & 3;
  if (pattern == 0) src = lane;
  else if (pattern == 1) src = er + 16 * (((ep & 1) << 1) | (ep >> 1));
  else if (pattern == 2) src = (lane & 32) + ((lane & 31) >> 1) + 16 * (lane & 1);
  else { const int r = er; src = ep == 0 ? r : ep == 1 ? r + 16 : ep == 2 ? 32 + ((r + 8) & 15) : 48 + ((r + 8) & 15); }
  const int addr = 4 * src;
  int v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = lane * 7 + i;
  __syncthreads();
  unsigned long long t0 = 0, t1 = 0;
  if (wave < nw) {
    t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 32; it++) {
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = __builtin_amdgcn_ds_bpermute(addr, v[i]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i];
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  unsigned long long* out; int* sink;
  hipMalloc(&out, 8 * 8 * 8); hipMalloc(&sink, 8 * 512 * 4);
  unsigned long long h[64];
  for (int nw = 8; nw >= 1; nw /= 8)
    for (int p = 0; p < 4; p++) {
      for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k, dim3(8), dim3(512), 0, 0, p, nw, out, sink);
      hipDeviceSynchronize();
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      unsigned long long mx = 0;
      for (int w = 0; w < nw; w++) mx = h[w] > mx ? h[w] : mx;
      printf("pattern %d, %d waves: 256 bpermutes per wave in %llu cycles (slowest wave) = %.2f cycles per wave-instruction per CU\n", p, nw, mx,
             (double)mx / (256.0 * nw));
    }
  return 0;
}
