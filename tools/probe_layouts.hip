// GPU probe: checks the gfx950 MFMA fragment maps, ds_read_b64_tr_b16 and global_load_lds
// semantics that the kernels in pero_pretraining_amd/csrc rely on. Exact integer data.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_layouts.hip -o gpurun_out/probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ inline __bf16 f2b(float f){ return (__bf16)f; }

// A [32][16], B [16][32] (B[k][n]), C [32][32]
__global__ void k_mfma32(const float* A, const float* B, float* C) {
  int l = threadIdx.x; bf8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = f2b(A[(l & 31) * 16 + 8 * (l >> 5) + j]); b[j] = f2b(B[(8 * (l >> 5) + j) * 32 + (l & 31)]); }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); C[row * 32 + (l & 31)] = c[r]; }
}
// A [16][32], B [32][16], C[16][16]
__global__ void k_mfma16(const float* A, const float* B, float* C) {
  int l = threadIdx.x; bf8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = f2b(A[(l & 15) * 32 + 8 * (l >> 4) + j]); b[j] = f2b(B[(8 * (l >> 4) + j) * 16 + (l & 15)]); }
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) { int row = (l >> 4) * 4 + r; C[row * 16 + (l & 15)] = c[r]; }
}
// f32: A[32][2], B[2][32]
__global__ void k_mfma32f(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)], b = B[(l >> 5) * 32 + (l & 31)];
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); C[row * 32 + (l & 31)] = c[r]; }
}
// f32: A[16][4], B[4][16]
__global__ void k_mfma16f(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) { int row = (l >> 4) * 4 + r; C[row * 16 + (l & 15)] = c[r]; }
}
// tr read: LDS tile [R=16][C=64] shorts, value = row*64+col. Lane l: group g=l>>4, i=l&15, q=i>>2, p=i&3.
// address = row (4*g + q) , cols 16*cb + 4p   (cb = column block param)
__global__ void k_tr(short* out, int rowstride) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 128];
  for (int i = threadIdx.x; i < 64 * 128; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  short* addr = lds + (4 * g + q) * rowstride + 4 * p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)addr);
  for (int e = 0; e < 4; e++) out[l * 4 + e] = v[e];
}
// global_load_lds 16B: 2 waves, each wave its own LDS base; source per lane permuted.
__global__ void k_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[2 * 64 * 4];
  int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 512; i += 128) lds[i] = -1;
  __syncthreads();
  int srclane = (l * 7) & 63;  // permuted source
  const int* g = src + w * 256 + srclane * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
      (__attribute__((address_space(3))) void*)(lds + w * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 128) out[i] = lds[i];
}
// accumulator (32x32 f32 result X = A1*B1, rows on regs) as B operand of the next mfma: Y = A2 * X.
// A2 is [32][32] (k index = row of X). k order inside step s: element j of lane half h is row 16s+8(j>>2)+4h+(j&3).
__global__ void k_acc_as_b(const float* A1, const float* B1, const float* A2, float* Y) {
  int l = threadIdx.x, h = l >> 5; bf8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = f2b(A1[(l & 31) * 16 + 8 * h + j]); b[j] = f2b(B1[(8 * h + j) * 32 + (l & 31)]); }
  f16v x = {0};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f16v y = {0};
  for (int s = 0; s < 2; s++) {
    bf8 xb, a2;
    for (int j = 0; j < 8; j++) {
      xb[j] = f2b(x[8 * s + j]);
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      a2[j] = f2b(A2[(l & 31) * 32 + k]);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  for (int r = 0; r < 16; r++) { int row = (r & 3) + 8 * (r >> 2) + 4 * h; Y[row * 32 + (l & 31)] = y[r]; }
}

static void matmul(const float* A, const float* B, float* C, int M, int N, int K) {
  for (int m = 0; m < M; m++) for (int n = 0; n < N; n++) { float s = 0; for (int k = 0; k < K; k++) s += A[m * K + k] * B[k * N + n]; C[m * N + n] = s; }
}
static int cmp(const float* a, const float* b, int n) { int bad = 0; for (int i = 0; i < n; i++) if (a[i] != b[i]) bad++; return bad; }
template <class F> static void run_mm(const char* name, int M, int N, int K, F launch) {
  std::vector<float> A(M * K), B(K * N), C(M * N), R(M * N);
  for (auto& v : A) v = (float)(rand() % 7 - 3);
  for (auto& v : B) v = (float)(rand() % 5 - 2);
  float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  launch(dA, dB, dC); CK(hipDeviceSynchronize());
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  matmul(A.data(), B.data(), R.data(), M, N, K);
  printf("%-28s mismatches=%d / %d  %s\n", name, cmp(C.data(), R.data(), M * N), M * N, cmp(C.data(), R.data(), M * N) ? "FAIL" : "PASS");
  hipFree(dA); hipFree(dB); hipFree(dC);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d lds/block=%zu clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.clockRate);
  run_mm("mfma_f32_32x32x16_bf16", 32, 32, 16, [](float* a, float* b, float* c) { k_mfma32<<<1, 64>>>(a, b, c); });
  run_mm("mfma_f32_16x16x32_bf16", 16, 16, 32, [](float* a, float* b, float* c) { k_mfma16<<<1, 64>>>(a, b, c); });
  run_mm("mfma_f32_32x32x2f32", 32, 32, 2, [](float* a, float* b, float* c) { k_mfma32f<<<1, 64>>>(a, b, c); });
  run_mm("mfma_f32_16x16x4f32", 16, 16, 4, [](float* a, float* b, float* c) { k_mfma16f<<<1, 64>>>(a, b, c); });
  {  // acc as B
    std::vector<float> A1(32 * 16), B1(16 * 32), A2(32 * 32), X(32 * 32), Y(32 * 32), R(32 * 32);
    for (auto& v : A1) v = (float)(rand() % 3 - 1); for (auto& v : B1) v = (float)(rand() % 3 - 1); for (auto& v : A2) v = (float)(rand() % 5 - 2);
    matmul(A1.data(), B1.data(), X.data(), 32, 32, 16); matmul(A2.data(), X.data(), R.data(), 32, 32, 32);
    float *d1, *d2, *d3, *d4; CK(hipMalloc(&d1, 2048)); CK(hipMalloc(&d2, 2048)); CK(hipMalloc(&d3, 4096)); CK(hipMalloc(&d4, 4096));
    CK(hipMemcpy(d1, A1.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, B1.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(d3, A2.data(), 4096, hipMemcpyHostToDevice));
    k_acc_as_b<<<1, 64>>>(d1, d2, d3, d4); CK(hipDeviceSynchronize());
    CK(hipMemcpy(Y.data(), d4, 4096, hipMemcpyDeviceToHost));
    printf("%-28s mismatches=%d / 1024  %s\n", "acc-as-B (Y=A2*X)", cmp(Y.data(), R.data(), 1024), cmp(Y.data(), R.data(), 1024) ? "FAIL" : "PASS");
  }
  for (int rs : {64, 128}) {  // tr read
    short* d; CK(hipMalloc(&d, 512)); k_tr<<<1, 64>>>(d, rs); CK(hipDeviceSynchronize());
    short h[256]; CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
    // hypothesis: lane (g,i) element e = tile[row 4g+e][col i]
    int bad = 0; for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) { int exp = (4 * (l >> 4) + e) * rs + (l & 15); if (h[l * 4 + e] != exp) bad++; }
    printf("ds_read_tr16_b64 rowstride=%d: hypothesis mismatches=%d %s\n", rs, bad, bad ? "FAIL" : "PASS");
    if (bad) for (int l = 0; l < 64; l++) printf("  lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, h[l*4]/rs, h[l*4]%rs, h[l*4+1]/rs, h[l*4+1]%rs, h[l*4+2]/rs, h[l*4+2]%rs, h[l*4+3]/rs, h[l*4+3]%rs);
    hipFree(d);
  }
  {  // glds
    int hs[512], ho[512]; for (int i = 0; i < 512; i++) hs[i] = i;
    int *ds, *dout; CK(hipMalloc(&ds, 2048)); CK(hipMalloc(&dout, 2048)); CK(hipMemcpy(ds, hs, 2048, hipMemcpyHostToDevice));
    k_glds<<<1, 128>>>(ds, dout); CK(hipDeviceSynchronize()); CK(hipMemcpy(ho, dout, 2048, hipMemcpyDeviceToHost));
    int bad = 0; for (int w = 0; w < 2; w++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) { int exp = w * 256 + ((l * 7) & 63) * 4 + e; if (ho[w * 256 + l * 4 + e] != exp) bad++; }
    printf("global_load_lds b128 (dest=base+lane*16, src per lane): mismatches=%d %s\n", bad, bad ? "FAIL" : "PASS");
    if (bad) { for (int i = 0; i < 32; i++) printf("%d ", ho[i]); printf("\n"); }
  }
  // quick HBM copy bandwidth
  {
    size_t n = 1ull << 30; char *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n)); CK(hipMemset(a, 1, n));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    CK(hipMemcpy(b, a, n, hipMemcpyDeviceToDevice)); hipEventRecord(e0);
    for (int i = 0; i < 10; i++) CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0));
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpy D2D 1GiB: %.1f GB/s (read+write)\n", 2.0 * n * 10 / ms / 1e6);
  }
  return 0;
}
