// HBM bandwidth of the attention kernels' access pattern: the packed qkv / dqkv tensors are (lines * 256 rows) x 3072 bytes, a (line, head,
// q|k|v) tile is 256 rows x 256 bytes at a 3072-byte pitch.  Copies 1024 lines (805 MB) in four ways and prints TB/s (read + write).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_tilebw.hip -o tools/probe_tilebw.bin && tools/probe_tilebw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
// linear: consecutive 16-byte pieces
__global__ __launch_bounds__(256) void k_linear(const u4v* src, u4v* dst, long long n16, int mode) {
  u4v acc = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
    if (mode != 2) { u4v v = src[i]; if (mode == 0) dst[i] = v; else acc += v; }
    else dst[i] = (u4v){(unsigned)i, 1, 2, 3};
  }
  if (mode == 1 && acc[0] == 0x12345u) dst[0] = acc;
}
// tiles: one workgroup per (line, column block of `cb` bytes); a wave-instruction covers 1024 / cb ... rows: lanes (16 * cb / 256 ...)
// SEG = bytes per row covered by adjacent lanes (256: 16 lanes, 64: 4 lanes)
template <int SEG>
__global__ __launch_bounds__(256) void k_tile(const unsigned char* src, unsigned char* dst, int lines, int mode) {
  // tile = 256 rows x 256 B; tiles per line = 12; thread -> (row group, 16-byte piece)
  const long long tile = blockIdx.x;
  const long long line = tile / 12, cb = tile % 12;
  const unsigned char* s = src + line * 256 * 3072 + cb * 256;
  unsigned char* d = dst + line * 256 * 3072 + cb * 256;
  constexpr int LPR = SEG / 16;            // lanes per row segment
  constexpr int RPI = 256 / LPR;           // rows per 256-thread instruction
  const int r0 = threadIdx.x / LPR, c0 = (threadIdx.x % LPR) * 16;
  u4v acc = {0, 0, 0, 0};
  for (int cseg = 0; cseg < 256; cseg += SEG)
#pragma unroll 4
    for (int r = r0; r < 256; r += RPI) {
      const long long off = (long long)r * 3072 + cseg + c0;
      if (mode != 2) { u4v v = *(const u4v*)(s + off); if (mode == 0) *(u4v*)(d + off) = v; else acc += v; }
      else *(u4v*)(d + off) = (u4v){(unsigned)r, 1, 2, 3};
    }
  if (mode == 1 && acc[0] == 0x12345u) *(u4v*)d = acc;
}
int main() {
  const int lines = 1024;
  const size_t bytes = (size_t)lines * 256 * 3072;
  unsigned char *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* mn[3] = {"copy (read + write)", "read only", "write only"};
  for (int mode = 0; mode < 3; mode++) {
    for (int kind = 0; kind < 4; kind++) {
      float best = 1e9f;
      for (int it = 0; it < 6; it++) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_linear, dim3(4096), dim3(256), 0, 0, (const u4v*)a, (u4v*)b, (long long)(bytes / 16), mode);
        else if (kind == 1) hipLaunchKernelGGL(k_tile<256>, dim3(lines * 12), dim3(256), 0, 0, a, b, lines, mode);
        else if (kind == 2) hipLaunchKernelGGL(k_tile<64>, dim3(lines * 12), dim3(256), 0, 0, a, b, lines, mode);
        else hipLaunchKernelGGL(k_tile<128>, dim3(lines * 12), dim3(256), 0, 0, a, b, lines, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      const double moved = mode == 0 ? 2.0 * bytes : (double)bytes;
      printf("%-20s %-34s %8.1f us  %6.2f TB/s\n", mn[mode], kind == 0 ? "linear 16-byte pieces" : kind == 1 ? "tiles, 256-byte row segments" : kind == 2 ? "tiles, 64-byte row segments" : "tiles, 128-byte row segments",
             best * 1e3, moved / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
