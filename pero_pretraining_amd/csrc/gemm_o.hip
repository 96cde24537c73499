// 128 x 128 x 64 bf16 tile GEMM, simple form: one tile per workgroup, 2 workgroups per CU, LDS-DMA double buffer,
// f32 epilogue staged through LDS (coalesced 16-byte stores).  Kept as the A/B baseline of the tile-kernel policy
// (tools/step_ab.py) and used for split-K atomic products when selected.
#include "gemm_common.hpp"

#define O_BM 128
#define O_BN 128
#define O_BK 64
#define O_OPBYTES (128 * 64 * 2)          // one operand tile: 16 KiB
#define O_BUFBYTES (2 * O_OPBYTES)        // A + B
#define O_EPI_PITCH 528                   // f32 epilogue row pitch in bytes (128*4 + 16)
#define O_LDS_BYTES (128 * O_EPI_PITCH)   // 67584 >= 2 * O_BUFBYTES (65536)


// K-contiguous image [128 rows][64 k] (128-byte rows), 16-byte chunk index XORed with (row & 7)
__device__ __forceinline__ bf8v ofrag_rowmajor(const unsigned char* base, int row, int ks, int lane) {
  const int r = row + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ (r & 7)) << 4));
}
// K-major image [64 k-rows][128 cols] (256-byte rows), 32-byte blocks XORed with fk(krow);
// two transposed 8-byte reads give the 8 consecutive k of one column.
__device__ __forceinline__ bf8v ofrag_kmajor(const unsigned char* base, int col, int ks, int lane) {
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
__device__ __forceinline__ void ostage_glds(const bf16raw* X, long long ld, long long tile0, long long k0,
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
__global__ __launch_bounds__(256, 2) void gemm_bf16_o128(GemmP p, int ks_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b, b+8, ... share an XCD (and its L2); give each XCD a contiguous
  // run of tiles (N fastest) so that neighbours share the A row panel.  Bijective for any tile count.
  const int ntn = (int)(p.N / O_BN);
  const int nt = (int)(p.M / O_BM) * ntn;
  const int bid = blockIdx.x;
  int id, zslice;
  if (ks_xcd > 0) {
    // split-K products (weight gradients: the reduction runs over all tokens): put one k-slice per XCD, all of its
    // output tiles on that XCD.  Every dY / X panel of the slice is then fetched from HBM once and re-read from that
    // XCD's L2 by the other tiles (measured before: 408 MB fetched per launch for ~130 MB of unique operand bytes,
    // because the tiles of one slice were spread over all 8 L2s).  Placement is a speed choice only.
    const int xcd = bid & 7, r = bid >> 3;
    if (ks_xcd >= 8) { const int per = ks_xcd >> 3; zslice = xcd * per + (r % per); id = r / per; }
    else { zslice = xcd % ks_xcd; id = r * (8 / ks_xcd) + xcd / ks_xcd; }
  } else {
    const int q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
    id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    zslice = blockIdx.z;
  }
  const long long tm0 = (long long)(id / ntn) * O_BM, tn0 = (long long)(id % ntn) * O_BN;

  const int b = blockIdx.y;
  const long long bo = b / p.binner, bi = b % p.binner;
  const bf16raw* A = (const bf16raw*)p.A + bo * p.sAo + bi * p.sAi;
  const bf16raw* B = (const bf16raw*)p.B + bo * p.sBo + bi * p.sBi;
  const long long coff = bo * p.sCo + bi * p.sCi;

  const long long kbeg = (long long)zslice * p.kchunk;
  long long kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const int nk = (int)((kend - kbeg) / O_BK);

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

  // Main loop rotated by half a stage (see gemm_v.hip): the fragments of the next half-step are read from LDS while the
  // current one multiplies, the stage barrier sits between the two half-steps of a stage, so every wave leaves the barrier
  // into MFMAs.  With two workgroups of four waves per CU (two waves per SIMD) the plain form serialised the LDS-read and
  // the MFMA phases almost completely.
  auto half_step = [&](const unsigned char* nsa, const unsigned char* nsb, int nks, bf8v (&fa)[4], bf8v (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {  // swapped operands: D[n][m], so a lane holds 4 consecutive n of one m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (i == 3) {
          fb[j] = TB ? ofrag_kmajor(nsb, wn * 64 + j * 16, nks, lane) : ofrag_rowmajor(nsb, wn * 64 + j * 16, nks, lane);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, TB ? 2 : 1, 0);
        }
      }
      fa[i] = TA ? ofrag_kmajor(nsa, wm * 64 + i * 16, nks, lane) : ofrag_rowmajor(nsa, wm * 64 + i * 16, nks, lane);
      if (i < 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TA ? 2 : 1, 0);
    }
  };
  bf8v fa[4], fb[4];
  if (nk > 0) {
    ostage_glds<TA>(A, p.lda, tm0, kbeg, smem, tid);
    ostage_glds<TB>(B, p.ldb, tn0, kbeg, smem + O_OPBYTES, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (nk > 1) {
      ostage_glds<TA>(A, p.lda, tm0, kbeg + O_BK, smem + O_BUFBYTES, tid);
      ostage_glds<TB>(B, p.ldb, tn0, kbeg + O_BK, smem + O_BUFBYTES + O_OPBYTES, tid);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa[i] = TA ? ofrag_kmajor(smem, wm * 64 + i * 16, 0, lane) : ofrag_rowmajor(smem, wm * 64 + i * 16, 0, lane);
      fb[i] = TB ? ofrag_kmajor(smem + O_OPBYTES, wn * 64 + i * 16, 0, lane) : ofrag_rowmajor(smem + O_OPBYTES, wn * 64 + i * 16, 0, lane);
    }
  }
  for (int t = 0; t < nk; t++) {
    unsigned char* s0 = smem + (t & 1) * O_BUFBYTES;         // stage t
    unsigned char* s1 = smem + ((t + 1) & 1) * O_BUFBYTES;   // stage t + 1
    half_step(s0, s0 + O_OPBYTES, 1, fa, fb);                 // (t, 0) multiplies, (t, 1) is read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own DMA of stage t + 1
    lds_barrier();                                            // everyone's; every wave has finished reading stage t
    if (t + 2 < nk) {
      ostage_glds<TA>(A, p.lda, tm0, kbeg + (long long)(t + 2) * O_BK, s0, tid);
      ostage_glds<TB>(B, p.ldb, tn0, kbeg + (long long)(t + 2) * O_BK, s0 + O_OPBYTES, tid);
    }
    half_step(s1, s1 + O_OPBYTES, 0, fa, fb);                 // (t, 1) multiplies, (t + 1, 0) is read (unused after the last stage)
  }
  __syncthreads();  // all fragment reads done before the epilogue reuses the LDS

  // ---- epilogue: accumulators -> LDS (f32) -> coalesced global rows ---------------------------------
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = wm * 64 + i * 16 + (lane & 15);
      const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
      *(f4v*)(smem + m * O_EPI_PITCH + n * 4) = acc[i][j];
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
        const float v = *(const float*)(smem + row * O_EPI_PITCH + col * 4) * p.alpha;
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
    const f4v v0 = *(const f4v*)(smem + row * O_EPI_PITCH + c8 * 4);
    const f4v v1 = *(const f4v*)(smem + row * O_EPI_PITCH + c8 * 4 + 16);
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


bool pero_launch_gemm_o128(const GemmP& p0, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (p0.M % O_BM || p0.N % O_BN || p0.K % O_BK) return false;
  GemmP p = p0;
  if (k_split > 1) {
    const long long steps = p.K / O_BK;
    const long long per = (steps + k_split - 1) / k_split;
    p.kchunk = per * O_BK;
    k_split = (int)((steps + per - 1) / per);
  } else {
    p.kchunk = p.K;
    k_split = 1;
  }
  dim3 grid((unsigned)((p.M / O_BM) * (p.N / O_BN)), (unsigned)batch, (unsigned)k_split), block(256);
  int ks_xcd = 0;
  extern int g_pero_splitk_xcd;
  const long long tiles = (p.M / O_BM) * (p.N / O_BN);
  if (g_pero_splitk_xcd && batch == 1 && k_split > 1 && (k_split == 2 || k_split == 4 || k_split % 8 == 0) && (tiles * k_split) % 8 == 0 &&
      (k_split >= 8 || tiles % (8 / k_split) == 0)) {
    ks_xcd = k_split;
    grid = dim3((unsigned)(tiles * k_split), 1, 1);
  }
#define LAUNCH_O(TA_, TB_, OF_)                                                                                           \
  do {                                                                                                                    \
    PERO_LDS_ATTR((gemm_bf16_o128<TA_, TB_, OF_>), O_LDS_BYTES);                                                          \
    hipLaunchKernelGGL((gemm_bf16_o128<TA_, TB_, OF_>), grid, block, O_LDS_BYTES, st, p, ks_xcd);                                 \
  } while (0)
  if (!ta && !tb) { if (out_f32) LAUNCH_O(false, false, true); else LAUNCH_O(false, false, false); }
  else if (!ta && tb) { if (out_f32) LAUNCH_O(false, true, true); else LAUNCH_O(false, true, false); }
  else if (ta && tb) { if (out_f32) LAUNCH_O(true, true, true); else LAUNCH_O(true, true, false); }
  else { if (out_f32) LAUNCH_O(true, false, true); else LAUNCH_O(true, false, false); }
#undef LAUNCH_O
  return true;
}
