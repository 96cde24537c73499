// EXPERIMENT: 256 x 256 x 64 bf16 tile GEMM with FOUR waves (2 x 2, 128 x 128 outputs each = 256 accumulator registers per
// lane, one wave per SIMD), v_mfma_f32_32x32x16_bf16.  Half the LDS fragment bytes per MFMA of the 64 x 64 wave tile (8 KiB
// per 16 MFMAs of 32 cycles) and a quarter of the waves at every barrier.  K-contiguous operands only (NT products).
#include "gemm_common.hpp"

#define X_BM 256
#define X_BN 256
#define X_BK 64
#define X_ABYTES (256 * 64 * 2)      // 32 KiB operand tile
#define X_BUFBYTES (2 * X_ABYTES)    // 64 KiB per stage
#define X_EPI_PITCH 1040             // f32 staging pitch (256 * 4 + 16)
#define X_LDS_BYTES (2 * X_BUFBYTES) // 128 KiB (>= 64-row f32 staging of 66560 B)

// K-contiguous image [256 rows][64 k] = 128-byte rows; 16-byte chunk index XORed with ((row >> 1) & 7): a 32x32x16 operand
// fragment is read by 32 lanes = 32 consecutive rows with ONE chunk index; the ds_read_b128 lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}) then hold 8 even and 8 odd rows whose (row >> 1) & 7 are all different.
__device__ __forceinline__ int xsw(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ bf8v xfrag(const unsigned char* base, int row0, int ks16, int lane) {
  const int r = row0 + (lane & 31);
  const int chunk = ks16 * 2 + (lane >> 5);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ xsw(r)) << 4));
}
// 256-row operand tile x 64 k = 32 pieces of 1 KiB (8 rows x 128 B): eight LDS-DMA instructions per wave
__device__ __forceinline__ void xstage_glds(const bf16raw* X, long long ld, long long tile0, long long k0, unsigned char* lds_base, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int piece = i * 4 + wave;
    const int row = piece * 8 + (lane >> 3), chunk = (lane & 7) ^ xsw(row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (tile0 + row) * ld + k0 + chunk * 8),
                                     (__attribute__((address_space(3))) void*)(lds_base + piece * 1024), 16, 0, 0);
  }
}

template <bool OUTF32>
__global__ __launch_bounds__(256, 1) void gemm_bf16_x256(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (int)(p.N / X_BN);
  const int nt = (int)(p.M / X_BM) * ntn;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const long long tm0 = (long long)(id / ntn) * X_BM, tn0 = (long long)(id % ntn) * X_BN;
  const bf16raw* A = (const bf16raw*)p.A;
  const bf16raw* B = (const bf16raw*)p.B;
  const int nk = (int)(p.K / X_BK);

  f16v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f16v){0};

  // k16 step: 16 MFMAs (row-tile i x column-tile j); the fragments of the NEXT k16 step are read while it runs
  auto kstep = [&](const unsigned char* nsa, const unsigned char* nsb, int nks, bf8v (&fa)[4], bf8v (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);  // swapped: a lane owns 4 consecutive n
        if (i == 3) {
          fb[j] = xfrag(nsb, wn * 128 + j * 32, nks, lane);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      fa[i] = xfrag(nsa, wm * 128 + i * 32, nks, lane);
      if (i < 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  bf8v fa[4], fb[4];
  xstage_glds(A, p.lda, tm0, 0, smem, tid);
  xstage_glds(B, p.ldb, tn0, 0, smem + X_ABYTES, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if (nk > 1) {
    xstage_glds(A, p.lda, tm0, X_BK, smem + X_BUFBYTES, tid);
    xstage_glds(B, p.ldb, tn0, X_BK, smem + X_BUFBYTES + X_ABYTES, tid);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    fa[i] = xfrag(smem, wm * 128 + i * 32, 0, lane);
    fb[i] = xfrag(smem + X_ABYTES, wn * 128 + i * 32, 0, lane);
  }
  for (int t = 0; t < nk; t++) {
    unsigned char* s0 = smem + (t & 1) * X_BUFBYTES;         // stage t
    unsigned char* s1 = smem + ((t + 1) & 1) * X_BUFBYTES;   // stage t + 1
    kstep(s0, s0 + X_ABYTES, 1, fa, fb);
    kstep(s0, s0 + X_ABYTES, 2, fa, fb);
    kstep(s0, s0 + X_ABYTES, 3, fa, fb);                      // the last reads of stage t
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own DMA of stage t + 1
    lds_barrier();                                            // everyone's; every wave has finished reading stage t
    if (t + 2 < nk) {
      xstage_glds(A, p.lda, tm0, (long long)(t + 2) * X_BK, s0, tid);
      xstage_glds(B, p.ldb, tn0, (long long)(t + 2) * X_BK, s0 + X_ABYTES, tid);
    }
    kstep(s1, s1 + X_ABYTES, 0, fa, fb);                      // (t, 3) multiplies, (t + 1, 0) is read (unused after the last stage)
  }

  // ---- epilogue: four 64-row chunks through LDS -> whole 512-byte row segments (16-byte lanes)
  const int c8 = (tid & 31) * 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; e++) bias[e] = p.bias ? p.bias[tn0 + c8 + e] : 0.f;
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    lds_barrier();
    if (wm == (qq >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 2; ii++) {
        const int i = (qq & 1) * 2 + ii;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            const f4v v = {acc[i][j][4 * g4 + 0], acc[i][j][4 * g4 + 1], acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]};
            *(f4v*)(smem + (ii * 32 + (lane & 31)) * X_EPI_PITCH + (wn * 128 + j * 32 + 8 * g4 + 4 * (lane >> 5)) * 4) = v;
          }
      }
    }
    lds_barrier();
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
      const int row = (tid >> 5) + 8 * rr;
      const f4v v0 = *(const f4v*)(smem + row * X_EPI_PITCH + c8 * 4);
      const f4v v1 = *(const f4v*)(smem + row * X_EPI_PITCH + c8 * 4 + 16);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = v[e] * p.alpha + bias[e];
      const long long grow = tm0 + qq * 64 + row;
      if (p.resid) {
        const uint4 rr4 = *(const uint4*)((const bf16raw*)p.resid + grow * p.ldr + tn0 + c8);
        const unsigned w[4] = {rr4.x, rr4.y, rr4.z, rr4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
      if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.gate) {
        const uint4 gg = *(const uint4*)((const bf16raw*)p.gate + grow * p.ldg + tn0 + c8);
        const unsigned w[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(w[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      if (OUTF32) {
        float* C = (float*)p.C + grow * p.ldc + tn0 + c8;
        *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
        *(f4v*)(C + 4) = (f4v){v[4], v[5], v[6], v[7]};
      } else {
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        *(uint4*)((bf16raw*)p.C + grow * p.ldc + tn0 + c8) = o;
      }
    }
  }
}

// NT products with a stored output only (no batch, no split-K, no atomics / accumulate)
bool pero_launch_gemm_x256(const GemmP& p, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (ta || tb || batch != 1 || k_split > 1 || p.M % X_BM || p.N % X_BN || p.K % X_BK || (p.flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM)))
    return false;
  dim3 grid((unsigned)((p.M / X_BM) * (p.N / X_BN))), block(256);
  static bool attr[2] = {false, false};
  if (out_f32) {
    if (!attr[1]) { hipFuncSetAttribute((const void*)gemm_bf16_x256<true>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES); attr[1] = true; }
    hipLaunchKernelGGL((gemm_bf16_x256<true>), grid, block, X_LDS_BYTES, st, p);
  } else {
    if (!attr[0]) { hipFuncSetAttribute((const void*)gemm_bf16_x256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS_BYTES); attr[0] = true; }
    hipLaunchKernelGGL((gemm_bf16_x256<false>), grid, block, X_LDS_BYTES, st, p);
  }
  return true;
}
