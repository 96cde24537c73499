// Where does a burst of stores stall?  8 waves per CU issue 16 stores of 1 KiB each (fresh lines per repeat); s_memtime behind every
// store of the last repeat, relative to the first.  Patterns: rows per instruction x row pitch; storing waves 8 / 1.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_store3.hip -o tools/probe_store3.bin && tools/probe_store3.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int RPI>   // rows per instruction: 16 (64 B each), 8 (128 B), 1 (1 KiB contiguous)
__global__ __launch_bounds__(512) void k(unsigned char* C, long long pitch, int reps, int nw, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = 64 / RPI;
  const u4v v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  unsigned long long st[17];
  for (int r = 0; r < reps; r++) {
    unsigned char* tile = C + ((long long)blockIdx.x * reps + r) * 256 * pitch;
    // the wave's 16 instructions cover 16 * RPI rows x (16 * LPR) bytes; waves side by side in columns, then further down
    const int wcols = 512 / (16 * LPR) < 1 ? 1 : 512 / (16 * LPR);   // waves per row block
    unsigned char* base = tile + (long long)((wave / wcols) * 16 * RPI + lane / LPR) * pitch + (wave % wcols) * 16 * LPR + (lane % LPR) * 16;
    __syncthreads();
    if (wave < nw) {
      st[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        *(u4v*)(base + (long long)i * RPI * pitch) = v;
        st[i + 1] = __builtin_amdgcn_s_memtime();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (lane == 0)
    for (int i = 0; i < 17; i++) out[(blockIdx.x * 8 + wave) * 17 + i] = st[i] - st[0];
}
// the GEMM epilogue's ownership: wave = 128 rows x 128 B (whole lines), 16 rows x 64 B per instruction.  ORDER 0: the two halves of
// a line in consecutive instructions; 1: the eight first halves, then the eight second halves; 2: halves alternate but 2 row groups apart
template <int ORDER>
__global__ __launch_bounds__(512) void kg(unsigned char* C, long long pitch, int reps, int nw, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
  const u4v v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  unsigned long long st[17];
  for (int r = 0; r < reps; r++) {
    unsigned char* tile = C + ((long long)blockIdx.x * reps + r) * 256 * pitch;
    unsigned char* base = tile + (long long)((wave >> 2) * 128 + li) * pitch + (wave & 3) * 128 + lq * 16;
    __syncthreads();
    if (wave < nw) {
      st[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int rg, half;
        if (ORDER == 0) { rg = i >> 1; half = i & 1; }
        else if (ORDER == 1) { rg = i & 7; half = i >> 3; }
        else { rg = (i & 1) | ((i >> 2) << 1); half = (i >> 1) & 1; }   // rg0h0 rg1h0 rg0h1 rg1h1 rg2h0 ...
        *(u4v*)(base + (long long)rg * 16 * pitch + half * 64) = v;
        st[i + 1] = __builtin_amdgcn_s_memtime();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (lane == 0)
    for (int i = 0; i < 17; i++) out[(blockIdx.x * 8 + wave) * 17 + i] = st[i] - st[0];
}
// lane order of a conflict-free lane transpose: lanes 0-31 = (row L >> 1, piece L & 1), lanes 32-63 = (row (L - 32) >> 1, piece 2 + (L & 1)):
// adjacent PAIRS of lanes cover 32 contiguous bytes, the wave-instruction still 16 rows x 64 B.  MODE 1: quads (reference), 0: pairs
template <int MODE>
__global__ __launch_bounds__(512) void k2(unsigned char* C, long long pitch, int reps, int nw, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u4v v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  unsigned long long st[17];
  const int row = MODE ? (lane >> 2) : ((lane & 31) >> 1), piece = MODE ? (lane & 3) : (2 * (lane >> 5) + (lane & 1));
  for (int r = 0; r < reps; r++) {
    unsigned char* tile = C + ((long long)blockIdx.x * reps + r) * 256 * pitch;
    unsigned char* base = tile + (long long)((wave >> 2) * 128 + row) * pitch + (wave & 3) * 128 + piece * 16;
    __syncthreads();
    if (wave < nw) {
      st[0] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 16; i++) {
        *(u4v*)(base + (long long)(i >> 1) * 16 * pitch + (i & 1) * 64) = v;
        st[i + 1] = __builtin_amdgcn_s_memtime();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  if (lane == 0)
    for (int i = 0; i < 17; i++) out[(blockIdx.x * 8 + wave) * 17 + i] = st[i] - st[0];
}
int main() {
  const int G = 8, reps = 4;
  unsigned char* C; hipMalloc(&C, (size_t)G * reps * 4096 * 8192);
  unsigned long long* out; hipMalloc(&out, G * 8 * 17 * 8);
  std::vector<unsigned long long> h(G * 8 * 17);
#define RUN(RPI_, pitch_, nw_)                                                                              \
  do {                                                                                                      \
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k<RPI_>, dim3(G), dim3(512), 0, 0, C, (long long)(pitch_), reps, nw_, out); \
    hipDeviceSynchronize();                                                                                 \
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);                                          \
    printf("%2d rows / instruction, pitch %5d, %d storing waves: wave 0 store 16 issued at %5llu, last wave at %5llu; wave 0 steps:", RPI_, pitch_, nw_, h[16], h[(nw_ - 1) * 17 + 16]); \
    for (int i = 9; i <= 16; i++) printf(" %llu", h[i] - h[i - 1]);                                         \
    printf("\n");                                                                                           \
  } while (0)
  RUN(16, 4096, 8); RUN(16, 1024, 8); RUN(16, 8192, 8); RUN(16, 512, 8);
  RUN(8, 4096, 8); RUN(8, 1024, 8);
  RUN(1, 4096, 8); RUN(1, 1024, 8);
  RUN(16, 4096, 1); RUN(1, 4096, 1);
#define RUNG(ORD_, nw_)                                                                                     \
  do {                                                                                                      \
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(kg<ORD_>, dim3(G), dim3(512), 0, 0, C, 4096LL, reps, nw_, out); \
    hipDeviceSynchronize();                                                                                 \
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);                                          \
    printf("GEMM ownership, order %d, %d storing waves: wave 0 store 16 issued at %5llu, last wave at %5llu; wave 0 steps:", ORD_, nw_, h[16], h[(nw_ - 1) * 17 + 16]); \
    for (int i = 1; i <= 16; i++) printf(" %llu", h[i] - h[i - 1]);                                         \
    printf("\n");                                                                                           \
  } while (0)
  RUNG(0, 8); RUNG(1, 8); RUNG(2, 8); RUNG(0, 1); RUNG(1, 1);
#define RUN2(M_, nw_)                                                                                     \
  do {                                                                                                      \
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL(k2<M_>, dim3(G), dim3(512), 0, 0, C, 4096LL, reps, nw_, out); \
    hipDeviceSynchronize();                                                                                 \
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);                                          \
    printf("transposed layout, %s, %d storing waves: wave 0 store 16 issued at %5llu, last wave at %5llu; wave 0 steps:", M_ ? "quads (64 B)" : "pairs (32 B) per half", nw_, h[16], h[(nw_ - 1) * 17 + 16]); \
    for (int i = 1; i <= 16; i++) printf(" %llu", h[i] - h[i - 1]);                                         \
    printf("\n");                                                                                           \
  } while (0)
  RUN2(1, 8); RUN2(0, 8); RUN2(1, 1); RUN2(0, 1);
  return 0;
}
