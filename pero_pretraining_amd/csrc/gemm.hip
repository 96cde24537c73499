// GEMM kernels of libpero_hip (gfx950).
//
//   C[b] = alpha * op(A[b]) * op(B[b])^T (+bias) (+residual) (relu) (* gate>0)
//
// Two kernels:
//  * gemm_bf16_t128: the hot kernel.  bf16 operands, f32 accumulation on
//    v_mfma_f32_16x16x32_bf16, 128x128x64 tiles, 4 waves (2x2, 64x64 each), double-buffered LDS filled by
//    LDS-DMA (global_load_lds_dwordx4: tile t+1 is in flight while tile t is multiplied), XOR-swizzled
//    LDS images (conflict-free ds_read_b128 for K-contiguous operands, conflict-free
//    ds_read_b64_tr_b16 for operands stored K-major), XCD-aware tile order, optional split-K with f32
//    atomics (weight gradients: reduction over all tokens), f32 epilogue staged through LDS so that
//    global stores are whole 256-byte row segments.
//  * gemm_generic: any shape / stride / dtype, exact f32 arithmetic on v_mfma_f32_32x32x2_f32 (a k-ordered
//    fmaf chain).  Parity mode (PERO_F32), ragged shapes and small problems.
#include "common.hpp"

struct GemmP {
  const void* A; const void* B; void* C;
  const float* bias; const void* resid; const void* gate;
  long long M, N, K, lda, ldb, ldc, ldr, ldg;
  long long sAo, sAi, sBo, sBi, sCo, sCi;
  int binner; float alpha; int flags; long long kchunk;
};

// ------------------------------------------------------------------------------------------------
// fast bf16 kernel
// ------------------------------------------------------------------------------------------------
#define T_BM 128
#define T_BN 128
#define T_BK 64
#define T_OPBYTES (128 * 64 * 2)          // one operand tile: 16 KiB
#define T_BUFBYTES (2 * T_OPBYTES)        // A + B
#define T_EPI_PITCH 528                   // f32 epilogue row pitch in bytes (128*4 + 16)
#define T_LDS_BYTES (128 * T_EPI_PITCH)   // 67584 >= 2 * T_BUFBYTES (65536)

__device__ __forceinline__ int fk(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// K-contiguous image [128 rows][64 k] (128-byte rows), 16-byte chunk index XORed with (row & 7)
__device__ __forceinline__ bf8v frag_rowmajor(const unsigned char* base, int row, int ks, int lane) {
  const int r = row + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ (r & 7)) << 4));
}
// K-major image [64 k-rows][128 cols] (256-byte rows), 32-byte blocks XORed with fk(krow);
// two transposed 8-byte reads give the 8 consecutive k of one column.
__device__ __forceinline__ bf8v frag_kmajor(const unsigned char* base, int col, int ks, int lane) {
  const int i = lane & 15;
  const int krow = ks * 32 + 8 * (lane >> 4) + (i >> 2);
  const unsigned char* a = base + krow * 256 + ((((col >> 4) ^ fk(krow))) << 5) + 8 * (i & 3);
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a + 4 * 256));
  typedef short s8v __attribute__((ext_vector_type(8)));
  s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8v, v);
}

// Global -> LDS staging by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write traffic (register
// staging of a 128x128x64 step costs ~415 LDS cycles of ds_write_b128 against 512 MFMA cycles).  One wave
// instruction writes 1 KiB linearly (wave base + lane * 16), so the XOR swizzles of the two LDS images are
// applied to the per-lane SOURCE address instead (guide rule 21): piece (i, wave) covers image bytes
// [i*4096 + wave*1024, +1024) = rows 32i + 8*wave .. +7 of a K-contiguous image, or k-rows 16i + 4*wave .. +3
// of a K-major image, exactly the rows thread `tid` addresses below.
template <bool TR>
__device__ __forceinline__ void stage_glds(const bf16raw* X, long long ld, long long tile0, long long k0,
                                           unsigned char* lds_base, int tid) {
  const bf16raw* p;
  long long step;
  if (!TR) {  // stored [rows][K]: lane's LDS slot (tid & 7) of row (tid >> 3) holds logical chunk slot ^ (row & 7)
    const int row = tid >> 3, chunk = (tid & 7) ^ (row & 7);
    p = X + (tile0 + row) * ld + k0 + chunk * 8;
    step = 32 * ld;
  } else {    // stored [K][rows]: 32-byte blocks of k-row XORed with fk(krow)
    const int krow = tid >> 4, slot = tid & 15;
    const int chunk = ((((slot >> 1) ^ fk(krow))) << 1) | (slot & 1);
    p = X + (k0 + krow) * ld + tile0 + chunk * 8;
    step = 16 * ld;
  }
  unsigned char* dst = lds_base + (tid >> 6) * 1024;
#define GLDS16(src_, dst_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_), \
                                                            (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0)
  GLDS16(p, dst);
  GLDS16(p + step, dst + 4096);
  GLDS16(p + 2 * step, dst + 8192);
  GLDS16(p + 3 * step, dst + 12288);
#undef GLDS16
}

template <bool TA, bool TB, bool OUTF32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_t128(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
  // run of tiles (N fastest) so that neighbours share the A row panel.  Bijective for any tile count.
  const int ntn = (int)(p.N / T_BN);
  const int nt = (int)(p.M / T_BM) * ntn;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  const long long tm0 = (long long)(id / ntn) * T_BM, tn0 = (long long)(id % ntn) * T_BN;

  const int b = blockIdx.y;
  const long long bo = b / p.binner, bi = b % p.binner;
  const bf16raw* A = (const bf16raw*)p.A + bo * p.sAo + bi * p.sAi;
  const bf16raw* B = (const bf16raw*)p.B + bo * p.sBo + bi * p.sBi;
  const long long coff = bo * p.sCo + bi * p.sCi;

  const long long kbeg = (long long)blockIdx.z * p.kchunk;
  long long kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const int nk = (int)((kend - kbeg) / T_BK);

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    stage_glds<TA>(A, p.lda, tm0, kbeg, smem, tid);
    stage_glds<TB>(B, p.ldb, tn0, kbeg, smem + T_OPBYTES, tid);
  }
  for (int t = 0; t < nk; t++) {
    // tile t has landed (own DMA drained, then barrier: everyone's); all waves are past their reads of the
    // other buffer (tile t-1), so it can be refilled while tile t is multiplied
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* sa = smem + (t & 1) * T_BUFBYTES;
    const unsigned char* sb = sa + T_OPBYTES;
    if (t + 1 < nk) {
      unsigned char* da = smem + ((t + 1) & 1) * T_BUFBYTES;
      stage_glds<TA>(A, p.lda, tm0, kbeg + (long long)(t + 1) * T_BK, da, tid);
      stage_glds<TB>(B, p.ldb, tn0, kbeg + (long long)(t + 1) * T_BK, da + T_OPBYTES, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      bf8v fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        fa[i] = TA ? frag_kmajor(sa, wm * 64 + i * 16, ks, lane) : frag_rowmajor(sa, wm * 64 + i * 16, ks, lane);
        fb[i] = TB ? frag_kmajor(sb, wn * 64 + i * 16, ks, lane) : frag_rowmajor(sb, wn * 64 + i * 16, ks, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)  // swapped operands: D[n][m], so a lane holds 4 consecutive n of one m
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();  // all fragment reads done before the epilogue reuses the LDS

  // ---- epilogue: accumulators -> LDS (f32) -> coalesced global rows ---------------------------------
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = wm * 64 + i * 16 + (lane & 15);
      const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
      *(f4v*)(smem + m * T_EPI_PITCH + n * 4) = acc[i][j];
    }
  __syncthreads();

  if (OUTF32 && (p.flags & PERO_GEMM_ATOMIC)) {
    float* C = (float*)p.C + coff;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
      const int row = wave + 4 * i;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int col = lane + 64 * j;
        const float v = *(const float*)(smem + row * T_EPI_PITCH + col * 4) * p.alpha;
        atomicAdd(C + (tm0 + row) * p.ldc + tn0 + col, v);
      }
    }
    return;
  }

  const int c8 = (tid & 15) * 8;
  float bias[8];
#pragma unroll
  for (int e = 0; e < 8; e++) bias[e] = p.bias ? p.bias[tn0 + c8 + e] : 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = (tid >> 4) + 16 * i;
    const f4v v0 = *(const f4v*)(smem + row * T_EPI_PITCH + c8 * 4);
    const f4v v1 = *(const f4v*)(smem + row * T_EPI_PITCH + c8 * 4 + 16);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = v[e] * p.alpha + bias[e];
    const long long grow = tm0 + row;
    if (p.resid) {
      const uint4 rr = *(const uint4*)((const bf16raw*)p.resid + coff + grow * p.ldr + tn0 + c8);
      const unsigned w[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        v[2 * e] += __uint_as_float(w[e] << 16);
        v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
      }
    }
    if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.gate) {
      const uint4 gg = *(const uint4*)((const bf16raw*)p.gate + coff + grow * p.ldg + tn0 + c8);
      const unsigned w[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
        if (!(__uint_as_float(w[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
      }
    }
    if (OUTF32) {
      float* C = (float*)p.C + coff + grow * p.ldc + tn0 + c8;
      if (p.flags & PERO_GEMM_ACCUM) {
        const f4v o0 = *(const f4v*)C, o1 = *(const f4v*)(C + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] += o0[e]; v[4 + e] += o1[e]; }
      }
      *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
      *(f4v*)(C + 4) = (f4v){v[4], v[5], v[6], v[7]};
    } else {
      bf16raw* C = (bf16raw*)p.C + coff + grow * p.ldc + tn0 + c8;
      uint4 o;
      o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
      *(uint4*)C = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// generic exact-f32 kernel: 64x64x16 tiles, 4 waves (2x2), each a 32x32 v_mfma_f32_32x32x2_f32 chain
// ------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void gemm_generic(GemmP p) {
  __shared__ float As[64][17];
  __shared__ float Bs[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long tm0 = (long long)blockIdx.x * 64, tn0 = (long long)blockIdx.y * 64;
  const int b = blockIdx.z;
  const long long bo = b / p.binner, bi = b % p.binner;
  const TI* A = (const TI*)p.A + bo * p.sAo + bi * p.sAi;
  const TI* B = (const TI*)p.B + bo * p.sBo + bi * p.sBi;
  const long long coff = bo * p.sCo + bi * p.sCi;
  const bool ta = p.flags & PERO_GEMM_TRANS_A, tb = p.flags & PERO_GEMM_TRANS_B;
  const long long sam = ta ? 1 : p.lda, sak = ta ? p.lda : 1;
  const long long sbn = tb ? 1 : p.ldb, sbk = tb ? p.ldb : 1;

  f16v acc = {0};
  for (long long k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = tid + 256 * i;
      int m, k;
      if (ta) { m = idx & 63; k = idx >> 6; } else { m = idx >> 4; k = idx & 15; }
      float v = 0.f;
      if (tm0 + m < p.M && k0 + k < p.K) v = Elem<TI>::ld(A + (tm0 + m) * sam + (k0 + k) * sak);
      As[m][k] = v;
      int n, k2;
      if (tb) { n = idx & 63; k2 = idx >> 6; } else { n = idx >> 4; k2 = idx & 15; }
      float w = 0.f;
      if (tn0 + n < p.N && k0 + k2 < p.K) w = Elem<TI>::ld(B + (tn0 + n) * sbn + (k0 + k2) * sbk);
      Bs[n][k2] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      const float a = As[wm * 32 + (lane & 31)][k + (lane >> 5)];
      const float bb = Bs[wn * 32 + (lane & 31)][k + (lane >> 5)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const long long n = tn0 + wn * 32 + (lane & 31);
  if (n >= p.N) return;
  const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const long long m = tm0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m >= p.M) continue;
    float v = acc[r] * p.alpha + bias;
    if (p.resid) v += Elem<TI>::ld((const TI*)p.resid + coff + m * p.ldr + n);
    if (p.flags & PERO_GEMM_RELU) v = fmaxf(v, 0.f);
    if (p.gate && !(Elem<TI>::ld((const TI*)p.gate + coff + m * p.ldg + n) > 0.f)) v = 0.f;
    TO* c = (TO*)p.C + coff + m * p.ldc + n;
    if (sizeof(TO) == 4 && (p.flags & PERO_GEMM_ATOMIC)) atomicAdd((float*)c, v);
    else if (sizeof(TO) == 4 && (p.flags & PERO_GEMM_ACCUM)) *(float*)c += v;
    else Elem<TO>::st(c, v);
  }
}

// ------------------------------------------------------------------------------------------------
extern "C" int pero_gemm(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* gate,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldg,
                         int64_t batch, int64_t batch_inner,
                         int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi,
                         float alpha, int flags, int k_split, int in_dtype, int out_dtype, void* stream) {
  PERO_REQUIRE(A && B && C, "pero_gemm: null operand");
  PERO_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch_inner > 0, "pero_gemm: bad sizes M=%lld N=%lld K=%lld batch=%lld",
               (long long)M, (long long)N, (long long)K, (long long)batch);
  PERO_REQUIRE((in_dtype == PERO_F32 || in_dtype == PERO_BF16) && (out_dtype == PERO_F32 || out_dtype == PERO_BF16), "pero_gemm: bad dtype");
  PERO_REQUIRE(!((flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM)) && out_dtype != PERO_F32), "pero_gemm: ATOMIC/ACCUM need f32 C");
  PERO_REQUIRE(k_split >= 1 && (k_split == 1 || (flags & PERO_GEMM_ATOMIC)), "pero_gemm: k_split > 1 needs PERO_GEMM_ATOMIC");
  PERO_REQUIRE(batch < 65536, "pero_gemm: batch too large");
  const bool ta = flags & PERO_GEMM_TRANS_A, tb = flags & PERO_GEMM_TRANS_B;
  GemmP p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.resid = residual; p.gate = gate;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldg = ldg;
  p.sAo = sAo; p.sAi = sAi; p.sBo = sBo; p.sBi = sBi; p.sCo = sCo; p.sCi = sCi;
  p.binner = (int)batch_inner; p.alpha = alpha; p.flags = flags; p.kchunk = K;
  hipStream_t st = (hipStream_t)stream;

  const int esz_o = out_dtype == PERO_F32 ? 4 : 2;
  bool fast = in_dtype == PERO_BF16 && !(flags & 32) && M % T_BM == 0 && N % T_BN == 0 && K % T_BK == 0 &&
              lda % 8 == 0 && ldb % 8 == 0 && (ldc * esz_o) % 16 == 0 && aligned16(A) && aligned16(B) && aligned16(C) &&
              (sAo % 8 == 0) && (sAi % 8 == 0) && (sBo % 8 == 0) && (sBi % 8 == 0) && ((sCo * esz_o) % 16 == 0) &&
              ((sCi * esz_o) % 16 == 0) && (!residual || (ldr % 8 == 0 && aligned16(residual))) &&
              (!gate || (ldg % 8 == 0 && aligned16(gate))) && (!(residual || gate) || out_dtype == PERO_BF16 || true);
  if (fast) {
    if (k_split > 1) {
      long long steps = K / T_BK;
      long long per = (steps + k_split - 1) / k_split;
      p.kchunk = per * T_BK;
      k_split = (int)((steps + per - 1) / per);
    }
    dim3 grid((unsigned)((M / T_BM) * (N / T_BN)), (unsigned)batch, (unsigned)k_split), block(256);
#define LAUNCH_FAST(TA_, TB_, OF_)                                                                                        \
  do {                                                                                                                    \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      hipFuncSetAttribute((const void*)gemm_bf16_t128<TA_, TB_, OF_>, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS_BYTES); \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL((gemm_bf16_t128<TA_, TB_, OF_>), grid, block, T_LDS_BYTES, st, p);                                 \
  } while (0)
    const bool of = out_dtype == PERO_F32;
    if (!ta && !tb) { if (of) LAUNCH_FAST(false, false, true); else LAUNCH_FAST(false, false, false); }
    else if (!ta && tb) { if (of) LAUNCH_FAST(false, true, true); else LAUNCH_FAST(false, true, false); }
    else if (ta && tb) { if (of) LAUNCH_FAST(true, true, true); else LAUNCH_FAST(true, true, false); }
    else { if (of) LAUNCH_FAST(true, false, true); else LAUNCH_FAST(true, false, false); }
    PERO_CHECK_LAUNCH("pero_gemm(bf16 fast)");
    return PERO_OK;
  }
  PERO_REQUIRE((M + 63) / 64 < 2147483647LL && (N + 63) / 64 < 65536, "pero_gemm: N too large for the generic kernel");
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)batch), block(256);
  if (in_dtype == PERO_F32 && out_dtype == PERO_F32) hipLaunchKernelGGL((gemm_generic<float, float>), grid, block, 0, st, p);
  else if (in_dtype == PERO_BF16 && out_dtype == PERO_BF16) hipLaunchKernelGGL((gemm_generic<bf16raw, bf16raw>), grid, block, 0, st, p);
  else if (in_dtype == PERO_BF16 && out_dtype == PERO_F32) hipLaunchKernelGGL((gemm_generic<bf16raw, float>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((gemm_generic<float, bf16raw>), grid, block, 0, st, p);
  PERO_CHECK_LAUNCH("pero_gemm(generic)");
  return PERO_OK;
}
