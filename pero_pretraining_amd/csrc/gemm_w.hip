// PERSISTENT 256 x 256 x 64 bf16 tile GEMM: one workgroup per CU (sixteen waves, 64 x 64 outputs each) walks tiles
// bid, bid + G, ...; the (tile, k-step) pairs form ONE LDS-DMA stream, so the first stage of the next tile is already in
// flight while the current tile's epilogue runs, and the epilogue's global stores drain under the next tile's main loop.
// Why: gemm_bf16_v256 (one tile per workgroup, same inner loop) spends 67 of 175 us of the 65536 x 2048 x 512 product in
// launch + pipeline fill + epilogue ("nothing" ablation): all 256 workgroups reach their epilogue together, the 32 MB of
// a round are written at ~4 TB/s while the MFMA pipe idles, and HBM idles during the main loops.
// vmcnt is ONE in-order counter for loads, LDS-DMA and stores: the wait for the next tile's first stage counts the
// epilogue's stores issued after it (W_EPI_STORES) instead of draining them.
#include "gemm_common.hpp"

#define W_BM 256
#define W_BN 256
#define W_BK 64
#define W_ABYTES (256 * 64 * 2)      // 32 KiB operand tile
#define W_BUFBYTES (2 * W_ABYTES)    // 64 KiB per stage
#define W_LDS_BYTES (5 * W_ABYTES)   // 160 KiB: three A slots + two B slots; the f32 epilogue staging (64 rows x 1 KiB, XOR-swizzled) takes one of each

// K-contiguous image [256 rows][64 k] = 128-byte rows, 16-byte chunk index XORed with (row & 7): conflict-free for the
// real ds_read_b128 lane groups ({0-3,12-15,20-27}, ...: rows {0-3,12-15} with chunk c and rows {4-11} with chunk c^1).
__device__ __forceinline__ bf8v wfrag_rowmajor(const unsigned char* base, int row, int ks, int lane) {
  const int r = row + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ (r & 7)) << 4));
}
// K-major image [64 k-rows][256 cols] (512-byte rows), 32-byte blocks XORed with fk(krow)
__device__ __forceinline__ bf8v wfrag_kmajor(const unsigned char* base, int col, int ks, int lane) {
  const int i = lane & 15;
  const int krow = ks * 32 + 8 * (lane >> 4) + (i >> 2);
  const unsigned char* a = base + krow * 512 + ((((col >> 4) ^ fk(krow))) << 5) + 8 * (i & 3);
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a + 4 * 512));
  typedef short s8v __attribute__((ext_vector_type(8)));
  s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8v, v);
}
// 256-row operand tile x 64 k = 32 pieces of 1 KiB: two LDS-DMA instructions per wave.  Addresses are a UNIFORM tile base
// (scalar registers) + one 32-bit per-lane byte offset that never changes (saddr form of global_load_lds): the persistent
// kernel has no VGPRs to spare for 64-bit per-lane pointers of the current and the next tile (the first version spilled
// 21-43 registers, with scratch reloads inside the k-loop).
template <bool TR>
__device__ __forceinline__ unsigned wlane_off(long long ld, int tid) {
  if (!TR) {  // piece = 8 rows x 128 B: thread -> row (tid >> 3) (+ 128 for the second piece), LDS slot tid & 7
    const int row = tid >> 3, chunk = (tid & 7) ^ (row & 7);
    return (unsigned)((row * ld + chunk * 8) * 2);
  } else {    // piece = 2 k-rows x 512 B: k-row (tid >> 5) (+ 32), slot tid & 31
    const int krow = tid >> 5, slot = tid & 31;
    const int chunk = ((((slot >> 1) ^ fk(krow))) << 1) | (slot & 1);
    return (unsigned)((krow * ld + chunk * 8) * 2);
  }
}
template <bool TR>
__device__ __forceinline__ void wstage_glds(const bf16raw* X, long long ld, long long tile0, long long k0, unsigned off,
                                            unsigned char* lds_base, int tid) {
  const unsigned char* base = (const unsigned char*)(TR ? X + k0 * ld + tile0 : X + tile0 * ld + k0);  // uniform
  const long long step = (TR ? 32 : 128) * ld * 2;                                                     // uniform
  unsigned char* dst = lds_base + (tid >> 6) * 1024;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + step + off),
                                   (__attribute__((address_space(3))) void*)(dst + 16384), 16, 0, 0);
}

// EPI: 0 = bias / residual / ReLU / ReLU-gate epilogue; 1 = + column sums of the stored result accumulated into p.bias (an
// OUTPUT then: PERO_GEMM_COLSUM); 2 = per-128-column row dots of the stored result with p.gate written to p.bias
// ([M][N / 128] f32: PERO_GEMM_ROWDOT); 3 / 4 = the ReLU gate as a bit mask (PERO_GEMM_RELU_BITS), written (with
// PERO_GEMM_RELU) / applied - two modes so that the forward one does not carry the backward one's registers.  Modes 1 - 4
// exist for the K-contiguous bf16 products only (the input-gradient
// products on transposed weight copies).
template <bool TA, bool TB, bool OUTF32, int EPI>
__global__ __launch_bounds__(1024, 4) void gemm_bf16_w256(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = (int)(p.N / W_BN);
  const int nt = (int)(p.M / W_BM) * ntn;
  const int G = gridDim.x;  // multiple of 8: T & 7 == blockIdx.x & 7 == the XCD of this workgroup
  const int q8 = nt >> 3, r8 = nt & 7;
  const bf16raw* A = (const bf16raw*)p.A;
  const bf16raw* B = (const bf16raw*)p.B;
  const int nk = (int)(p.K / W_BK);  // >= 3 (launcher)
  const int c8 = (tid & 31) * 8;
  const unsigned offA = wlane_off<TA>(p.lda, tid), offB = wlane_off<TB>(p.ldb, tid);

  auto tile_of = [&](int T, long long& tm0, long long& tn0) {
    const int xcd = T & 7, loc = T >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    tm0 = (long long)(id / ntn) * W_BM;
    tn0 = (long long)(id % ntn) * W_BN;
  };

  int T = blockIdx.x;
  if (T >= nt) return;
  long long tm0, tn0;
  tile_of(T, tm0, tn0);

  f4v acc[4][4];
  // half step (see gemm_v.hip): 16 MFMAs on the fragments in registers; every register set is re-read from (nsa, nsb, nks)
  // right after its last MFMA
  auto half_step = [&](const unsigned char* nsa, const unsigned char* nsb, int nks, bf8v (&fa)[4], bf8v (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (i == 3) {
          fb[j] = TB ? wfrag_kmajor(nsb, wn * 64 + j * 16, nks, lane) : wfrag_rowmajor(nsb, wn * 64 + j * 16, nks, lane);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, TB ? 2 : 1, 0);
        }
      }
      fa[i] = TA ? wfrag_kmajor(nsa, wm * 64 + i * 16, nks, lane) : wfrag_rowmajor(nsa, wm * 64 + i * 16, nks, lane);
      if (i < 3) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, TA ? 2 : 1, 0);
    }
  };

  // LDS ring: THREE 32 KiB slots for the A operand (streamed from HBM) + TWO for B (weights: L2 hits) = 160 KiB.  Stage g of
  // the (tile, k-step) stream lives in A slot g % 3 and B slot g & 1.  At the mid-stage barrier of stage g the workgroup
  // issues B(g + 2) and then A(g + 3), and waits with vmcnt(2): A(g + 2) - the newest two DMA instructions - may still be
  // in flight, so A has a window of TWO stages to land, B of one.  At a tile's last stage the two freed slots take the
  // epilogue staging instead (rows 0-31 in the A slot, 32-63 in the B slot) and the issue is made up after the epilogue.
  unsigned char* const abase = smem;
  unsigned char* const bbase = smem + 3 * W_ABYTES;
  int a0 = 0, b0 = 0;  // slots of the current stage
  bf8v fa[4], fb[4];
  wstage_glds<TA>(A, p.lda, tm0, 0, offA, abase, tid);
  wstage_glds<TB>(B, p.ldb, tn0, 0, offB, bbase, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  wstage_glds<TB>(B, p.ldb, tn0, W_BK, offB, bbase + W_ABYTES, tid);
  wstage_glds<TA>(A, p.lda, tm0, W_BK, offA, abase + W_ABYTES, tid);
  wstage_glds<TA>(A, p.lda, tm0, 2 * W_BK, offA, abase + 2 * W_ABYTES, tid);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    fa[i] = TA ? wfrag_kmajor(abase, wm * 64 + i * 16, 0, lane) : wfrag_rowmajor(abase, wm * 64 + i * 16, 0, lane);
    fb[i] = TB ? wfrag_kmajor(bbase, wn * 64 + i * 16, 0, lane) : wfrag_rowmajor(bbase, wn * 64 + i * 16, 0, lane);
  }

  for (;;) {
    long long nm0 = 0, nn0 = 0;
    const bool has_next = T + G < nt;
    if (has_next) tile_of(T + G, nm0, nn0);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; t++) {
      const int a1 = a0 == 2 ? 0 : a0 + 1;
      unsigned char* sa0 = abase + a0 * W_ABYTES;
      unsigned char* sb0 = bbase + b0 * W_ABYTES;
      half_step(sa0, sb0, 1, fa, fb);                                 // (t, 0) multiplies, (t, 1) is read
      if (t + 2 < nk || has_next) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // own A, B of the next stage (and older
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stores) are done; A two stages ahead may fly
      lds_barrier();                                                  // everyone's; every wave has finished reading stage t
      if (t + 1 < nk) {                                               // (last stage: the freed slots stage the epilogue)
        if (t + 2 < nk) wstage_glds<TB>(B, p.ldb, tn0, (long long)(t + 2) * W_BK, offB, sb0, tid);
        else if (has_next) wstage_glds<TB>(B, p.ldb, nn0, 0, offB, sb0, tid);
        if (t + 3 < nk) wstage_glds<TA>(A, p.lda, tm0, (long long)(t + 3) * W_BK, offA, sa0, tid);
        else if (has_next) wstage_glds<TA>(A, p.lda, nm0, (long long)(t + 3 - nk) * W_BK, offA, sa0, tid);
      }
      half_step(abase + a1 * W_ABYTES, bbase + (b0 ^ 1) * W_ABYTES, 0, fa, fb);  // (t, 1) multiplies; (t + 1, 0) / (next tile, 0, 0) is read
      a0 = a1;
      b0 ^= 1;
    }
    // (a0, b0) now name the next tile's stage 0; the slots of the stage just finished are free for the epilogue staging
    unsigned char* const stgA = abase + (a0 == 0 ? 2 : a0 - 1) * W_ABYTES;
    unsigned char* const stgB = bbase + (b0 ^ 1) * W_ABYTES;

    // ---- epilogue: four 64-row f32 chunks through the free stage -> whole 512-byte row segments in 16-byte lanes.
    // Image: [64 rows][64 chunks of 16 B], chunk index XORed with (row & 15): conflict-free writes (8-lane groups = 8 rows of
    // one chunk) and reads (32 lanes = the 64 chunks of one row) without padding: exactly one 64 KiB stage.
    int lane_e = lane, tid_e = tid;  // opaque copies: keeps the ~30 loop-invariant staging / output addresses out of the tile
    asm volatile("" : "+v"(lane_e), "+v"(tid_e));  // loop's live set (hoisted, they spilled 20-26 registers in the k-loop)
    const int c8e = (tid_e & 31) * 8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bias[e] = ((EPI == 0 || EPI == 3 || EPI == 4) && p.bias) ? p.bias[tn0 + c8e + e] : 0.f;  // (modes 1, 2: p.bias is an output)
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // EPI 3, mask as input: this thread's eight mask bytes of the tile, fetched ahead of the staging barriers (gemm_v.hip)
    unsigned gbits[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (EPI == 4) {
#pragma unroll
      for (int i = 0; i < 8; i++)
        gbits[i] = *((const unsigned char*)p.gate + (tm0 + (i >> 1) * 64 + (tid_e >> 5) + 32 * (i & 1)) * p.ldg + ((tn0 + c8e) >> 3));
    }
#pragma unroll
    for (int qq = 0; qq < 4; qq++) {
      if (qq > 0) lds_barrier();  // previous chunk's staging reads are done (chunk 0: the stage was released by the last mid-barrier)
      if (wm == qq) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int row = (i & 1) * 16 + (lane_e & 15), chunk = wn * 16 + j * 4 + (lane_e >> 4);
            *(f4v*)((i < 2 ? stgA : stgB) + row * 1024 + ((chunk ^ (row & 15)) << 4)) = acc[i][j];
          }
      }
      lds_barrier();
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
        const int lrow = tid_e >> 5, row = lrow + 32 * rr;
        const int ch = (tid_e & 31) * 2;
        const unsigned char* stg = rr ? stgB : stgA;
        const f4v v0 = *(const f4v*)(stg + lrow * 1024 + ((ch ^ (lrow & 15)) << 4));
        const f4v v1 = *(const f4v*)(stg + lrow * 1024 + (((ch + 1) ^ (lrow & 15)) << 4));
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = v[e] * p.alpha + bias[e];
        const long long grow = tm0 + qq * 64 + row;
        if (p.resid) {
          const uint4 rr4 = *(const uint4*)((const bf16raw*)p.resid + grow * p.ldr + tn0 + c8e);
          const unsigned w[4] = {rr4.x, rr4.y, rr4.z, rr4.w};
#pragma unroll
          for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
        }
        if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
        }
        if (EPI == 2) {
          // this thread's 8 columns of row `grow` (rounded to bf16 as they are stored) times the same columns of p.gate; the
          // 16 lanes of a 128-column block share the row and meet by shuffles
          const uint4 gg = *(const uint4*)((const bf16raw*)p.gate + grow * p.ldg + tn0 + c8e);
          const unsigned w[4] = {gg.x, gg.y, gg.z, gg.w};
          float dot = 0.f;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            dot += bf2f(f2bf(v[2 * e])) * __uint_as_float(w[e] << 16);
            dot += bf2f(f2bf(v[2 * e + 1])) * __uint_as_float(w[e] & 0xffff0000u);
          }
          dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64); dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);
          if ((tid_e & 15) == 0) ((float*)p.bias)[grow * (p.N >> 7) + (tn0 >> 7) + ((tid_e >> 4) & 1)] = dot;
        } else if (EPI == 3) {
          // the ReLU gate as bits, written: one byte per thread (its 8 columns) and row
          unsigned m = 0;
#pragma unroll
          for (int e = 0; e < 8; e++) m |= (bf2f(f2bf(v[e])) > 0.f ? 1u : 0u) << e;
          *((unsigned char*)p.gate + grow * p.ldg + ((tn0 + c8e) >> 3)) = (unsigned char)m;
        } else if (EPI == 4) {
          const unsigned m = gbits[qq * 2 + rr];  // ... applied: fetched ahead of the staging barriers
#pragma unroll
          for (int e = 0; e < 8; e++)
            if (!((m >> e) & 1)) v[e] = 0.f;
        } else if (p.gate) {
          const uint4 gg = *(const uint4*)((const bf16raw*)p.gate + grow * p.ldg + tn0 + c8e);
          const unsigned w[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
            if (!(__uint_as_float(w[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
          }
        }
        if (EPI == 1) {
#pragma unroll
          for (int e = 0; e < 8; e++) cs[e] += v[e];
        }
        if (OUTF32) {
          float* C = (float*)p.C + grow * p.ldc + tn0 + c8e;
          if (p.flags & PERO_GEMM_ACCUM) {
            const f4v o0 = *(const f4v*)C, o1 = *(const f4v*)(C + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += o0[e]; v[4 + e] += o1[e]; }
          }
          *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
          *(f4v*)(C + 4) = (f4v){v[4], v[5], v[6], v[7]};
        } else {
          uint4 o;
          o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
          *(uint4*)((bf16raw*)p.C + grow * p.ldc + tn0 + c8e) = o;
        }
      }
    }
    if (EPI == 1) {
      // tile column sums: registers (8 rows per thread) -> lane ^ 32 (the wave's other row) -> LDS over the 16 waves -> one
      // atomic per column and tile.  The scratch is the A staging slot (free: the chunk loop's reads are behind the barrier).
#pragma unroll
      for (int e = 0; e < 8; e++) cs[e] += __shfl_xor(cs[e], 32, 64);
      lds_barrier();
      float* red = (float*)stgA;
      if ((tid_e & 63) < 32) {
#pragma unroll
        for (int e = 0; e < 8; e++) red[(tid_e >> 6) * 256 + (tid_e & 31) * 8 + e] = cs[e];
      }
      lds_barrier();
      if (tid_e < 256) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) t += red[w * 256 + tid_e];
        atomicAdd((float*)p.bias + tn0 + tid_e, t);
      }
    }
    if (!has_next) break;
    T += G;
    tm0 = nm0;
    tn0 = nn0;
    lds_barrier();            // every wave is done with the staging slots: they take the new tile's B(1) and A(2)
    wstage_glds<TB>(B, p.ldb, tn0, W_BK, offB, stgB, tid);
    wstage_glds<TA>(A, p.lda, tm0, 2 * W_BK, offA, stgA, tid);
    // the new tile's first fragments (its stage 0 landed during the last stage).  They are re-read here rather than kept
    // from the last half step: 32 live fragment registers across the epilogue spilled 40 VGPRs.
    const unsigned char* f0 = abase + a0 * W_ABYTES;
    const unsigned char* f1 = bbase + b0 * W_ABYTES;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa[i] = TA ? wfrag_kmajor(f0, wm * 64 + i * 16, 0, lane) : wfrag_rowmajor(f0, wm * 64 + i * 16, 0, lane);
      fb[i] = TB ? wfrag_kmajor(f1, wn * 64 + i * 16, 0, lane) : wfrag_rowmajor(f1, wn * 64 + i * 16, 0, lane);
    }
  }
}

// Qualifies: one problem (batch 1), no split-K, no atomics, M % 256 == N % 256 == K % 64 == 0.
bool pero_launch_gemm_w256(const GemmP& p0, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (p0.M % W_BM || p0.N % W_BN || p0.K % W_BK || p0.K < 3 * W_BK || batch != 1 || k_split > 1 || (p0.flags & PERO_GEMM_ATOMIC)) return false;
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22)) return false;  // 32-bit per-lane byte offsets inside a tile
  static int num_cus = 0;
  if (!num_cus) {
    hipDeviceProp_t prop; int dev = 0;
    hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
    num_cus = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;
    if (num_cus < 8) num_cus = 8;
  }
  GemmP p = p0;
  p.kchunk = p.K;
  const long long nt = (p.M / W_BM) * (p.N / W_BN);
  const unsigned G = (unsigned)(nt < num_cus ? ((nt + 7) / 8) * 8 : num_cus);
  dim3 grid(G), block(1024);
#define LAUNCH_W(TA_, TB_, OF_, EP_)                                                                                            \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      hipFuncSetAttribute((const void*)gemm_bf16_w256<TA_, TB_, OF_, EP_>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_w256<TA_, TB_, OF_, EP_>), grid, block, W_LDS_BYTES, st, p);                                  \
  } while (0)
  const int epi = (p.flags & PERO_GEMM_ROWDOT) ? 2 : (p.flags & PERO_GEMM_COLSUM) ? 1 : (p.flags & PERO_GEMM_RELU_BITS) ? ((p.flags & PERO_GEMM_RELU) ? 3 : 4) : 0;
  if (epi) {
    // the fused row dots exist for the K-contiguous bf16 products only; the column-sum mode (EPI 1) is not instantiated:
    // its per-tile atomics stall the persistent loop (see gemm.hip) - gemm_bf16_v256 takes those products
    if (ta || tb || out_f32 || epi == 1) return false;
    if (epi == 2) LAUNCH_W(false, false, false, 2); else if (epi == 3) LAUNCH_W(false, false, false, 3); else LAUNCH_W(false, false, false, 4);
    return true;
  }
  if (!ta && !tb) { if (out_f32) LAUNCH_W(false, false, true, 0); else LAUNCH_W(false, false, false, 0); }
  else if (!ta && tb) { if (out_f32) LAUNCH_W(false, true, true, 0); else LAUNCH_W(false, true, false, 0); }
  else if (ta && tb) { if (out_f32) LAUNCH_W(true, true, true, 0); else LAUNCH_W(true, true, false, 0); }
  else { if (out_f32) LAUNCH_W(true, false, true, 0); else LAUNCH_W(true, false, false, 0); }
#undef LAUNCH_W
  return true;
}
