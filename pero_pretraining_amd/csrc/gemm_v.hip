// 256 x 256 x 64 bf16 tile GEMM, sixteen waves (4 x 4, 64 x 64 outputs each, <= 128 VGPRs), one workgroup per CU,
// LDS-DMA double buffer of 64 KiB stages.  Same wave code as gemm_bf16_q256 with BK = 64: a K-contiguous operand is then
// fetched in whole 128-byte row segments.  tools/probe_dma.hip: the pure LDS-DMA stream of a 256-row panel shared by two
// workgroups of an XCD runs at 10.5 TB/s with 64-byte segments (BK = 32) and 11.5 - 14.5 TB/s with 128-byte segments;
// the q256 ablation showed that stream, not the MFMA pipe, sets the k-step time of the large-K products.
#include "gemm_common.hpp"

#define V_BM 256
#define V_BN 256
#define V_BK 64
#define V_ABYTES (256 * 64 * 2)      // 32 KiB operand tile
#define V_BUFBYTES (2 * V_ABYTES)    // 64 KiB per stage
#define V_EPI_PITCH 1040             // f32 staging pitch (256 * 4 + 16)
#define V_LDS_BYTES (5 * V_ABYTES)   // 160 KiB: three A slots + two B slots (>= 64-row f32 staging of 66560 B)

// K-contiguous image [256 rows][64 k] = 128-byte rows, 16-byte chunk index XORed with (row & 7): conflict-free for the
// real ds_read_b128 lane groups ({0-3,12-15,20-27}, ...: rows {0-3,12-15} with chunk c and rows {4-11} with chunk c^1).
__device__ __forceinline__ bf8v vfrag_rowmajor(const unsigned char* base, int row, int ks, int lane) {
  const int r = row + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ (r & 7)) << 4));
}
// K-major image [64 k-rows][256 cols] (512-byte rows), 32-byte blocks XORed with fk(krow)
__device__ __forceinline__ bf8v vfrag_kmajor(const unsigned char* base, int col, int ks, int lane) {
  const int i = lane & 15;
  const int krow = ks * 32 + 8 * (lane >> 4) + (i >> 2);
  const unsigned char* a = base + krow * 512 + ((((col >> 4) ^ fk(krow))) << 5) + 8 * (i & 3);
  return lds_tr16_pair(a, a + 4 * 512);
}
// 256-row operand tile x 64 k = 32 pieces of 1 KiB: two LDS-DMA instructions per wave
template <bool TR>
__device__ __forceinline__ void vstage_glds(const bf16raw* X, long long ld, long long tile0, long long k0,
                                            unsigned char* lds_base, int tid) {
  unsigned char* dst = lds_base + (tid >> 6) * 1024;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const bf16raw* p;
    if (!TR) {  // piece = 8 rows x 128 B: thread -> row (tid >> 3) + 128 i, LDS slot tid & 7
      const int row = (tid >> 3) + 128 * i, chunk = (tid & 7) ^ (row & 7);
      p = X + (tile0 + row) * ld + k0 + chunk * 8;
    } else {    // piece = 2 k-rows x 512 B: k-row (tid >> 5) + 32 i, slot tid & 31
      const int krow = (tid >> 5) + 32 * i, slot = tid & 31;
      const int chunk = ((((slot >> 1) ^ fk(krow))) << 1) | (slot & 1);
      p = X + (k0 + krow) * ld + tile0 + chunk * 8;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p),
                                     (__attribute__((address_space(3))) void*)(dst + i * 16384), 16, 0, 0);
  }
}

template <bool TA, bool TB, bool OUTF32>
__global__ __launch_bounds__(1024, 4) void gemm_bf16_v256(GemmP p, int ks_xcd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = (int)(p.N / V_BN);
  const int nt = (int)(p.M / V_BM) * ntn;
  const int bid = blockIdx.x;
  int id, zslice;
  if (ks_xcd > 0) {  // split-K: one k-slice per XCD, all of its tiles on that XCD's L2 (as gemm_bf16_o128)
    const int xcd = bid & 7, r = bid >> 3;
    if (ks_xcd >= 8) { const int per = ks_xcd >> 3; zslice = xcd * per + (r % per); id = r / per; }
    else { zslice = xcd % ks_xcd; id = r * (8 / ks_xcd) + xcd / ks_xcd; }
  } else {
    const int q = nt >> 3, r8 = nt & 7, xcd = bid & 7, loc = bid >> 3;
    id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    zslice = blockIdx.z;
  }
  const long long tm0 = (long long)(id / ntn) * V_BM, tn0 = (long long)(id % ntn) * V_BN;
  const int b = blockIdx.y;
  const long long bo = b / p.binner, bi = b % p.binner;
  const bf16raw* A = (const bf16raw*)p.A + bo * p.sAo + bi * p.sAi;
  const bf16raw* B = (const bf16raw*)p.B + bo * p.sBo + bi * p.sBi;
  const long long coff = bo * p.sCo + bi * p.sCi;
  const long long kbeg = (long long)zslice * p.kchunk;
  long long kend = kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  const int nk = (int)((kend - kbeg) / V_BK);

  f4v acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

  // Main loop, rotated by half a stage.  A stage (BK = 64) is two half-steps of 16 MFMAs per wave; the fragments of the
  // NEXT half-step are read from LDS while the current one multiplies (each register set is refilled as soon as its last
  // MFMA has issued), and the stage barrier sits between the two half-steps.  So the code after the barrier starts with
  // MFMAs on fragments already in registers: in the plain form (barrier, 8 fragment reads, 16 MFMAs) all sixteen waves
  // leave the barrier together, read together and multiply together - the "LDS reads only" ablation cost 64 of 145 us.
  auto half_step = [&](const unsigned char* nsa, const unsigned char* nsb, int nks, bf8v (&fa)[4], bf8v (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (i == 3) {
          fb[j] = TB ? vfrag_kmajor(nsb, wn * 64 + j * 16, nks, lane) : vfrag_rowmajor(nsb, wn * 64 + j * 16, nks, lane);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, TB ? 2 : 1, 0);    // then the LDS read(s) of the register set it freed
        }
      }
      fa[i] = TA ? vfrag_kmajor(nsa, wm * 64 + i * 16, nks, lane) : vfrag_rowmajor(nsa, wm * 64 + i * 16, nks, lane);
      if (i < 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);       // 4 MFMAs of row i
      __builtin_amdgcn_sched_group_barrier(0x100, TA ? 2 : 1, 0);        // then the read(s) that refill fa[i]
    }
  };
  // LDS: THREE slots for the A operand (the streamed one: activations / gradients from HBM) and two for B (weights: L2
  // hits) = 160 KiB.  At the mid-stage barrier of stage t the wave issues B(t+2) and then A(t+3), and waits with vmcnt(2):
  // the two A instructions issued one barrier earlier may still be in flight - A gets a window of TWO stages to land, B
  // of one.  (With two slots each, every DMA had exactly one stage: the waves spent ~30 % of their cycles in that wait.)
  unsigned char* const abase = smem;
  unsigned char* const bbase = smem + 3 * V_ABYTES;
  bf8v fa[4], fb[4];
  if (nk > 0) {
    vstage_glds<TA>(A, p.lda, tm0, kbeg, abase, tid);
    vstage_glds<TB>(B, p.ldb, tn0, kbeg, bbase, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (nk > 1) {
      vstage_glds<TB>(B, p.ldb, tn0, kbeg + V_BK, bbase + V_ABYTES, tid);
      vstage_glds<TA>(A, p.lda, tm0, kbeg + V_BK, abase + V_ABYTES, tid);
    }
    if (nk > 2) vstage_glds<TA>(A, p.lda, tm0, kbeg + 2 * V_BK, abase + 2 * V_ABYTES, tid);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa[i] = TA ? vfrag_kmajor(abase, wm * 64 + i * 16, 0, lane) : vfrag_rowmajor(abase, wm * 64 + i * 16, 0, lane);
      fb[i] = TB ? vfrag_kmajor(bbase, wn * 64 + i * 16, 0, lane) : vfrag_rowmajor(bbase, wn * 64 + i * 16, 0, lane);
    }
  }
  int a0 = 0;  // A slot of stage t (t % 3)
  for (int t = 0; t < nk; t++) {
    const int a1 = a0 == 2 ? 0 : a0 + 1;
    unsigned char* sa0 = abase + a0 * V_ABYTES;
    unsigned char* sb0 = bbase + (t & 1) * V_ABYTES;
    half_step(sa0, sb0, 1, fa, fb);                           // (t, 0) multiplies, (t, 1) is read
    if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // own A(t+1), B(t+1) landed; A(t+2) (newest) may fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                            // everyone's; every wave has finished reading stage t
    if (t + 2 < nk) vstage_glds<TB>(B, p.ldb, tn0, kbeg + (long long)(t + 2) * V_BK, sb0, tid);
    if (t + 3 < nk) vstage_glds<TA>(A, p.lda, tm0, kbeg + (long long)(t + 3) * V_BK, sa0, tid);
    half_step(abase + a1 * V_ABYTES, bbase + ((t + 1) & 1) * V_ABYTES, 0, fa, fb);  // (t, 1) multiplies, (t + 1, 0) is read
    a0 = a1;
  }

  // ---- epilogue: four 64-row chunks through LDS -> whole 512-byte row segments (16-byte lanes)
  const int c8 = (tid & 31) * 8;
  float bias[8];
  const bool colsum = p.flags & PERO_GEMM_COLSUM;  // p.bias is then an OUTPUT (column sums of the stored result)
#pragma unroll
  for (int e = 0; e < 8; e++) bias[e] = (p.bias && !colsum) ? p.bias[tn0 + c8 + e] : 0.f;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // bit-mask gate: the tile's eight mask bytes of this thread are fetched HERE, ahead of the staging barriers - loaded where
  // they are used (like the bf16 gate rows) each of the eight dependent loads per tile exposed its full latency
  const bool bits_in = p.gate && (p.flags & PERO_GEMM_RELU_BITS) && !(p.flags & PERO_GEMM_RELU);
  unsigned gbits[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  if (bits_in) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      gbits[i] = *((const unsigned char*)p.gate + (tm0 + (i >> 1) * 64 + (tid >> 5) + 32 * (i & 1)) * p.ldg + ((tn0 + c8) >> 3));
  }
#pragma unroll
  for (int qq = 0; qq < 4; qq++) {
    lds_barrier();
    if (wm == qq) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          *(f4v*)(smem + (i * 16 + (lane & 15)) * V_EPI_PITCH + (wn * 64 + j * 16 + (lane >> 4) * 4) * 4) = acc[i][j];
    }
    lds_barrier();
    if (OUTF32 && (p.flags & PERO_GEMM_ATOMIC)) {
      // split-K partial sums: one wave instruction adds 64 CONSECUTIVE floats of a row (256 contiguous bytes); the
      // 8-floats-per-lane form of the stored path would spread an instruction's adds over 2 KiB at a 32-byte stride
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int row = wave * 4 + rr;
        float* C = (float*)p.C + coff + (tm0 + qq * 64 + row) * p.ldc + tn0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int col = e * 64 + lane;
          atomicAdd(C + col, *(const float*)(smem + row * V_EPI_PITCH + col * 4) * p.alpha);
        }
      }
      continue;
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const int row = (tid >> 5) + 32 * rr;
      const f4v v0 = *(const f4v*)(smem + row * V_EPI_PITCH + c8 * 4);
      const f4v v1 = *(const f4v*)(smem + row * V_EPI_PITCH + c8 * 4 + 16);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = v[e] * p.alpha + bias[e];
      const long long grow = tm0 + qq * 64 + row;
      if (p.resid) {
        const uint4 rr4 = *(const uint4*)((const bf16raw*)p.resid + coff + grow * p.ldr + tn0 + c8);
        const unsigned w[4] = {rr4.x, rr4.y, rr4.z, rr4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
      if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.gate && (p.flags & PERO_GEMM_RELU_BITS)) {
        // the ReLU gate as bits: one byte per thread (its 8 columns) and row
        unsigned char* gb = (unsigned char*)p.gate + grow * p.ldg + ((tn0 + c8) >> 3);
        if (p.flags & PERO_GEMM_RELU) {
          unsigned m = 0;
#pragma unroll
          for (int e = 0; e < 8; e++) m |= (bf2f(f2bf(v[e])) > 0.f ? 1u : 0u) << e;
          *gb = (unsigned char)m;
        } else {
          const unsigned m = gbits[qq * 2 + rr];
#pragma unroll
          for (int e = 0; e < 8; e++)
            if (!((m >> e) & 1)) v[e] = 0.f;
        }
      } else if (p.gate) {
        const uint4 gg = *(const uint4*)((const bf16raw*)p.gate + coff + grow * p.ldg + tn0 + c8);
        const unsigned w[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(w[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      if (colsum) {
#pragma unroll
        for (int e = 0; e < 8; e++) cs[e] += v[e];
      }
      if (OUTF32) {
        float* C = (float*)p.C + coff + grow * p.ldc + tn0 + c8;
        if (p.flags & PERO_GEMM_ATOMIC) {
#pragma unroll
          for (int e = 0; e < 8; e++) atomicAdd(C + e, v[e]);
        } else {
          if (p.flags & PERO_GEMM_ACCUM) {
            const f4v o0 = *(const f4v*)C, o1 = *(const f4v*)(C + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += o0[e]; v[4 + e] += o1[e]; }
          }
          *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
          *(f4v*)(C + 4) = (f4v){v[4], v[5], v[6], v[7]};
        }
      } else {
        uint4 o;
        o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
        *(uint4*)((bf16raw*)p.C + coff + grow * p.ldc + tn0 + c8) = o;
      }
    }
  }
  if (colsum) {
    // tile column sums: registers (8 rows per thread) -> lane ^ 32 (the other row of the wave) -> LDS over the 16 waves
#pragma unroll
    for (int e = 0; e < 8; e++) cs[e] += __shfl_xor(cs[e], 32, 64);
    lds_barrier();
    float* red = (float*)smem;
    if (lane < 32) {
#pragma unroll
      for (int e = 0; e < 8; e++) red[wave * 256 + lane * 8 + e] = cs[e];
    }
    lds_barrier();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 16; w++) t += red[w * 256 + tid];
      atomicAdd((float*)p.bias + tn0 + tid, t);
    }
  }
}

bool pero_launch_gemm_v256(const GemmP& p0, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (p0.M % V_BM || p0.N % V_BN || p0.K % V_BK) return false;
  GemmP p = p0;
  if (k_split > 1) {
    const long long steps = p.K / V_BK;
    const long long per = (steps + k_split - 1) / k_split;
    p.kchunk = per * V_BK;
    k_split = (int)((steps + per - 1) / per);
  } else {
    p.kchunk = p.K;
    k_split = 1;
  }
  dim3 grid((unsigned)((p.M / V_BM) * (p.N / V_BN)), (unsigned)batch, (unsigned)k_split), block(1024);
  int ks_xcd = 0;
  const long long tiles = (p.M / V_BM) * (p.N / V_BN);
  if (batch == 1 && k_split > 1 && (k_split == 2 || k_split == 4 || k_split % 8 == 0) && (tiles * k_split) % 8 == 0 &&
      (k_split >= 8 || tiles % (8 / k_split) == 0)) {
    ks_xcd = k_split;
    grid = dim3((unsigned)(tiles * k_split), 1, 1);
  }
#define LAUNCH_V(TA_, TB_, OF_)                                                                                            \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      hipFuncSetAttribute((const void*)gemm_bf16_v256<TA_, TB_, OF_>, hipFuncAttributeMaxDynamicSharedMemorySize, V_LDS_BYTES); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_v256<TA_, TB_, OF_>), grid, block, V_LDS_BYTES, st, p, ks_xcd);                                  \
  } while (0)
  if (!ta && !tb) { if (out_f32) LAUNCH_V(false, false, true); else LAUNCH_V(false, false, false); }
  else if (!ta && tb) { if (out_f32) LAUNCH_V(false, true, true); else LAUNCH_V(false, true, false); }
  else if (ta && tb) { if (out_f32) LAUNCH_V(true, true, true); else LAUNCH_V(true, true, false); }
  else { if (out_f32) LAUNCH_V(true, false, true); else LAUNCH_V(true, false, false); }
#undef LAUNCH_V
  return true;
}
