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

#include "gemm_common.hpp"
#include <string.h>

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

// Persistent tile loop: 2 workgroups per CU walk the work items (tile, batch, k-slice); the LDS-DMA of the NEXT
// item's first k-tile is issued before the current item's epilogue, and the epilogue stores straight from the
// accumulators (a lane owns 4 consecutive output columns), so its HBM writes drain while the next main loop runs.
// Measured before this change (M=32768, N=2048, K=512): loads 35 us + MFMA 42 us + epilogue 35 us ran almost
// serially (100 us); K=512 GEMMs with bf16 output sit at the ridge of the HBM-write and MFMA rooflines, so the
// three phases must overlap.  vmcnt counts stores too: the wait for the prefetched tile is a COUNTED vmcnt that
// leaves the epilogue's stores in flight, and barriers are raw s_barrier (a __syncthreads would drain vmcnt(0)).
struct WorkItem { long long tm0, tn0, kbeg; int nk; const bf16raw* A; const bf16raw* B; long long coff; };

__device__ __forceinline__ WorkItem work_item(const GemmP& p, long long w, int ntn, int nt, int nbatch) {
  WorkItem it;
  const int lin = (int)(w % nt);
  const long long rest = w / nt;
  const int b = (int)(rest % nbatch), z = (int)(rest / nbatch);
  // XCD-aware tile order (blocks b, b+8, ... share an XCD): each XCD gets a contiguous run of tiles, N fastest
  const int q = nt >> 3, r8 = nt & 7, xcd = lin & 7, loc = lin >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  it.tm0 = (long long)(id / ntn) * T_BM;
  it.tn0 = (long long)(id % ntn) * T_BN;
  const long long bo = b / p.binner, bi = b % p.binner;
  it.A = (const bf16raw*)p.A + bo * p.sAo + bi * p.sAi;
  it.B = (const bf16raw*)p.B + bo * p.sBo + bi * p.sBi;
  it.coff = bo * p.sCo + bi * p.sCi;
  it.kbeg = (long long)z * p.kchunk;
  long long kend = it.kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  it.nk = (int)((kend - it.kbeg) / T_BK);
  return it;
}

// Direct epilogue of one 16x16 accumulator block: lane -> row m, 4 consecutive columns n..n+3.
template <bool OUTF32>
__device__ __forceinline__ void store_block(const GemmP& p, const f4v& a, long long coff, long long m, long long n) {
  float v[4];
  f4v bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) bias = *(const f4v*)(p.bias + n);
#pragma unroll
  for (int e = 0; e < 4; e++) v[e] = a[e] * p.alpha + bias[e];
  if (p.resid) {
    const uint2 rr = *(const uint2*)((const bf16raw*)p.resid + coff + m * p.ldr + n);
    v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
    v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
  }
  if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
    for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.gate) {
    const uint2 gg = *(const uint2*)((const bf16raw*)p.gate + coff + m * p.ldg + n);
    if (!(__uint_as_float(gg.x << 16) > 0.f)) v[0] = 0.f;
    if (!(__uint_as_float(gg.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
    if (!(__uint_as_float(gg.y << 16) > 0.f)) v[2] = 0.f;
    if (!(__uint_as_float(gg.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
  }
  if (OUTF32) {
    float* C = (float*)p.C + coff + m * p.ldc + n;
    if (p.flags & PERO_GEMM_ACCUM) {
      const f4v o = *(const f4v*)C;
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] += o[e];
    }
    *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
  } else {
    uint2 o;
    o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
    *(uint2*)((bf16raw*)p.C + coff + m * p.ldc + n) = o;
  }
}

// Persistent tile loop with a DEFERRED epilogue.  vmcnt counts loads, LDS-DMA and stores in issue order, so a
// wave that has just issued a tile's 16 stores cannot wait for its next LDS-DMA without also waiting for those
// stores to reach HBM (measured: the 134 MB output of a 32768 x 2048 x 512 GEMM cost 30 us on top of the 73 us main
// loop, fully exposed, for 128- and 256-wide tiles alike).  Instead the finished accumulators are kept in registers
// and their stores are issued two blocks per k-step INSIDE the next work item's main loop, right after that
// k-step's wait: by the next wait (one k-step of MFMAs later) they have drained.  The next item's first k-tile is
// prefetched by LDS-DMA during the last k-step of the current one.  2 workgroups per CU walk the items.
template <bool TA, bool TB, bool OUTF32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_t128(GemmP p, int nbatch, long long nwork) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (int)(p.N / T_BN);
  const int nt = (int)(p.M / T_BM) * ntn;
  const bool atomic = OUTF32 && (p.flags & PERO_GEMM_ATOMIC);
  const int lm = wm * 64 + (lane & 15), ln = wn * 64 + (lane >> 4) * 4;  // lane's row / first column inside a tile

  long long w = blockIdx.x;
  if (w >= nwork) return;
  WorkItem it = work_item(p, w, ntn, nt, nbatch);
  int buf = 0;  // LDS buffer holding the current k-tile
  stage_glds<TA>(it.A, p.lda, it.tm0, it.kbeg, smem, tid);
  stage_glds<TB>(it.B, p.ldb, it.tn0, it.kbeg, smem + T_OPBYTES, tid);

  f4v prev[4][4];  // finished tile awaiting its stores
  bool have_prev = false;
  long long pm = 0, pn = 0, pcoff = 0;
#define STORE_PREV(I_, J_) store_block<OUTF32>(p, prev[I_][J_], pcoff, pm + (I_) * 16, pn + (J_) * 16)
#define STORE_PREV_BATCH(B_)                                                        \
  do { STORE_PREV((2 * (B_)) & 3, (2 * (B_)) >> 2); STORE_PREV((2 * (B_) + 1) & 3, (2 * (B_) + 1) >> 2); } while (0)

  while (true) {
    const long long wnext = w + gridDim.x;
    const bool have_next = wnext < nwork;
    WorkItem nx;
    if (have_next) nx = work_item(p, wnext, ntn, nt, nbatch);

    f4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < it.nk; t++) {
      // k-tile t has landed (own DMA drained; the <= 2 store blocks issued one k-step ago drained with it), the
      // barrier publishes everyone's pieces and retires all reads of the other buffer
      wait_vmcnt<0>();
      lds_barrier();
      if (have_prev) {
        switch (t) {
          case 0: STORE_PREV_BATCH(0); break;
          case 1: STORE_PREV_BATCH(1); break;
          case 2: STORE_PREV_BATCH(2); break;
          case 3: STORE_PREV_BATCH(3); break;
          case 4: STORE_PREV_BATCH(4); break;
          case 5: STORE_PREV_BATCH(5); break;
          case 6: STORE_PREV_BATCH(6); break;
          case 7: STORE_PREV_BATCH(7); break;
          default: break;
        }
      }
      const unsigned char* sa = smem + buf * T_BUFBYTES;
      const unsigned char* sb = sa + T_OPBYTES;
      unsigned char* da = smem + (buf ^ 1) * T_BUFBYTES;
      if (t + 1 < it.nk) {
        stage_glds<TA>(it.A, p.lda, it.tm0, it.kbeg + (long long)(t + 1) * T_BK, da, tid);
        stage_glds<TB>(it.B, p.ldb, it.tn0, it.kbeg + (long long)(t + 1) * T_BK, da + T_OPBYTES, tid);
      } else if (have_next && !atomic) {  // cross-item prefetch (the atomic epilogue needs the LDS itself)
        stage_glds<TA>(nx.A, p.lda, nx.tm0, nx.kbeg, da, tid);
        stage_glds<TB>(nx.B, p.ldb, nx.tn0, nx.kbeg, da + T_OPBYTES, tid);
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
      buf ^= 1;
    }
    if (have_prev && it.nk < 8) {  // short main loop: flush the store blocks that found no k-step
      if (it.nk <= 0) STORE_PREV_BATCH(0);
      if (it.nk <= 1) STORE_PREV_BATCH(1);
      if (it.nk <= 2) STORE_PREV_BATCH(2);
      if (it.nk <= 3) STORE_PREV_BATCH(3);
      if (it.nk <= 4) STORE_PREV_BATCH(4);
      if (it.nk <= 5) STORE_PREV_BATCH(5);
      if (it.nk <= 6) STORE_PREV_BATCH(6);
      if (it.nk <= 7) STORE_PREV_BATCH(7);
    }
    have_prev = false;

    if (atomic) {
      // split-K partial tile: stage through LDS so that every atomic wave-instruction adds 256 contiguous bytes
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          *(f4v*)(smem + (lm + i * 16) * T_EPI_PITCH + (ln + j * 16) * 4) = acc[i][j];
      lds_barrier();
      float* C = (float*)p.C + it.coff;
#pragma unroll 4
      for (int i = 0; i < 32; i++) {
        const int row = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int col = lane + 64 * j;
          const float v = *(const float*)(smem + row * T_EPI_PITCH + col * 4) * p.alpha;
          atomicAdd(C + (it.tm0 + row) * p.ldc + it.tn0 + col, v);
        }
      }
      if (have_next) {
        lds_barrier();  // staging reads done before the next item's DMA overwrites the LDS
        stage_glds<TA>(nx.A, p.lda, nx.tm0, nx.kbeg, smem, tid);
        stage_glds<TB>(nx.B, p.ldb, nx.tn0, nx.kbeg, smem + T_OPBYTES, tid);
        buf = 0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) prev[i][j] = acc[i][j];
      pm = it.tm0 + lm; pn = it.tn0 + ln; pcoff = it.coff;
      have_prev = true;
    }
    if (!have_next) break;
    w = wnext;
    it = nx;
  }
  if (have_prev) {
#pragma unroll
    for (int b = 0; b < 8; b++) { STORE_PREV_BATCH(b); }
  }
#undef STORE_PREV
#undef STORE_PREV_BATCH
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
// Tile-kernel policy ("gemm_policy"; measurements: DESIGN.md section 8):
//   0 = auto:
//         * stored bf16 products with >= "gemm_e256_min" (192) tiles of 256x256 - and those flagged PERO_GEMM_TILE256 -> the
//           eight-phase persistent 256x256x64 kernel gemm_bf16_e256 (gemm_e.hip) with its fused epilogues;
//         * split-K atomic products (weight gradients) over >= 32768 reduction rows with >= "gemm_e_splitk_min" (4) output tiles
//           -> gemm_bf16_e256 in its split-K mode; shorter ones -> gemm_bf16_o128 (128x128x64, one tile per workgroup, one
//           k-slice per XCD), k-slices aimed at "splitk_items" (512) workgroups;
//         * other stored products -> gemm_bf16_r256 (256x128x32, two workgroups per CU: small batches) when M is a multiple
//           of 256, else the persistent 128x128x64 kernel of this file (also: batched products, PERO_GEMM_TILE128);
//   forcing one family for A/B runs and tests: 1 = 128x128x64 persistent, 4 = o128, 7 = r256, 20 = e256 (any tile count).
extern int g_gemm_e_var;
extern int g_attn_bwd_pair;           // attention.hip
extern int g_attn_pipe;
extern int g_attn_order;
extern int g_attn_lh;
extern int g_gemm_splitk_ws;          // gemm_e.hip
extern int g_gemm_splitk_table;
extern int g_gemm_nw;
extern int g_gemm_d128;
extern int g_gemm_e_walk;
static int g_gemm_policy = 0;         // 0 = auto, 1 = 128x128x64 persistent kernel (this file), 4 = gemm_bf16_o128, 7 = gemm_bf16_r256, 20 = gemm_bf16_e256
static int g_gemm_e256_min = 192;     // auto: stored products with at least this many 256x256 tiles take the eight-phase kernel (0 = never)
static int g_gemm_e_splitk_min = 4;   // ... and split-K products (reduction >= 32768 rows) with at least this many output tiles
static int g_splitk_items = 512;      // split-K of the 128x128 kernel aims at this many work items
static int g_splitk_nearest = 0;
int g_pero_splitk_xcd = 1;            // one k-slice per XCD where the slice count allows it (gemm_o.hip)
extern "C" int pero_set_option(const char* name, int value) {
  if (name && !strcmp(name, "gemm_policy")) { g_gemm_policy = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_e_var")) { g_gemm_e_var = value; return PERO_OK; }
  if (name && !strcmp(name, "attn_bwd_pair")) { g_attn_bwd_pair = value; return PERO_OK; }
  if (name && !strcmp(name, "attn_pipe")) { g_attn_pipe = value; return PERO_OK; }
  if (name && !strcmp(name, "attn_order")) { g_attn_order = value; return PERO_OK; }
  if (name && !strcmp(name, "attn_lh")) { g_attn_lh = value; return PERO_OK; }
  if (name && !strcmp(name, "splitk_workspace")) { g_gemm_splitk_ws = value; return PERO_OK; }
  if (name && !strcmp(name, "splitk_table")) { g_gemm_splitk_table = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_nw")) { g_gemm_nw = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_d128")) { g_gemm_d128 = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_e_walk")) { g_gemm_e_walk = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_e256_min")) { g_gemm_e256_min = value; return PERO_OK; }
  if (name && !strcmp(name, "gemm_e_splitk_min")) { g_gemm_e_splitk_min = value; return PERO_OK; }
  if (name && !strcmp(name, "splitk_xcd")) { g_pero_splitk_xcd = value; return PERO_OK; }
  if (name && !strcmp(name, "splitk_nearest")) { g_splitk_nearest = value; return PERO_OK; }
  if (name && !strcmp(name, "splitk_items")) { g_splitk_items = value > 0 ? value : 512; return PERO_OK; }
  pero_set_error("pero_set_option: unknown option %s", name ? name : "(null)");
  return PERO_E_INVALID;
}

static int gemm_dispatch(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* gate,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldg,
                         int64_t batch, int64_t batch_inner,
                         int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi,
                         float alpha, int flags, int k_split, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                         void* stream, bool* colsum_fused);

// does a split-K product of this shape take the eight-phase kernel (whose partial tiles can go to a caller-owned workspace)?
static bool splitk_takes_e256(int64_t M, int64_t N, int64_t K, int flags) {
  const bool can256 = M % 256 == 0 && N % 256 == 0 && !(flags & PERO_GEMM_TILE128) && g_gemm_policy != 1;
  const long long t256 = can256 ? (M / 256) * (N / 256) : 0;
  return can256 && K % 64 == 0 && K >= 128 &&
         (g_gemm_policy == 20 || (g_gemm_policy == 0 && ((flags & PERO_GEMM_TILE256) || (g_gemm_e256_min > 0 && K >= 32768 && t256 >= g_gemm_e_splitk_min))));
}
// Linear + residual + LayerNorm in one launch (the row-complete 128 x 512 tile of gemm_e.hip): Y = A W^T + bias + R (bf16, stored: the backward
// needs it), T = (Y - mean) * rstd * gamma + beta over the ROUNDED rows of Y (layernorm_fwd4_k's arithmetic), mean / rstd (f32 per row).
extern "C" int pero_gemm_resid_layernorm(const void* A, const void* W, const float* bias, const void* R, const float* gamma, const float* beta,
                                         void* Y, void* T, float* mean, float* rstd, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                         int64_t ldy, int64_t ldr, int64_t ldt, float eps, void* stream) {
  PERO_REQUIRE(A && W && R && gamma && beta && T && mean && rstd, "pero_gemm_resid_layernorm: null pointer");   // Y may be null: not stored
  PERO_REQUIRE(N == 512 && M > 0 && M % 128 == 0 && K % 64 == 0 && K >= 192 && lda % 8 == 0 && ldw % 8 == 0 && (!Y || ldy % 8 == 0) && ldr % 8 == 0 &&
               ldt % 8 == 0 && aligned16(A) && aligned16(W) && aligned16(R) && (!Y || aligned16(Y)) && aligned16(T) && aligned16(gamma) && aligned16(beta) &&
               (!bias || aligned16(bias)),
               "pero_gemm_resid_layernorm: bf16, N = 512, M %% 128 == 0, K %% 64 == 0, K >= 192, 16-byte aligned rows");
  GemmP p;
  p.A = A; p.B = W; p.C = Y; p.bias = bias; p.resid = R; p.gate = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = Y ? ldy : 512; p.ldr = ldr; p.ldg = 0;
  p.sAo = p.sAi = p.sBo = p.sBi = p.sCo = p.sCi = 0;
  p.binner = 1; p.alpha = 1.0f; p.flags = 0; p.kchunk = K;
  PERO_REQUIRE(pero_launch_gemm_n512_ln(p, T, ldt, mean, rstd, gamma, beta, eps, (hipStream_t)stream), "pero_gemm_resid_layernorm: shape not taken");
  PERO_CHECK_LAUNCH("pero_gemm_resid_layernorm");
  return PERO_OK;
}
extern "C" int pero_gemm_resid_layernorm_bwd(const void* A, const void* Wt, const void* R, const void* T, const float* rstd, const float* gamma,
                                             const float* beta, void* DX, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t M, int64_t N,
                                             int64_t K, int64_t lda, int64_t ldw, int64_t ldr, int64_t ldt, int64_t lddx, void* stream) {
  PERO_REQUIRE(A && Wt && R && T && rstd && gamma && beta && DX && dgamma && dbeta && work, "pero_gemm_resid_layernorm_bwd: null pointer");
  PERO_REQUIRE(N == 512 && M > 0 && M % 128 == 0 && K % 64 == 0 && K >= 192 && lda % 8 == 0 && ldw % 8 == 0 && ldr % 8 == 0 && ldt % 8 == 0 &&
               lddx % 8 == 0 && aligned16(A) && aligned16(Wt) && aligned16(R) && aligned16(T) && aligned16(DX) && aligned16(gamma) && aligned16(beta),
               "pero_gemm_resid_layernorm_bwd: bf16, N = 512, M %% 128 == 0, K %% 64 == 0, K >= 192, 16-byte aligned rows");
  GemmP p;
  p.A = A; p.B = Wt; p.C = DX; p.bias = nullptr; p.resid = R; p.gate = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldw; p.ldc = lddx; p.ldr = ldr; p.ldg = 0;
  p.sAo = p.sAi = p.sBo = p.sBi = p.sCo = p.sCi = 0;
  p.binner = 1; p.alpha = 1.0f; p.flags = 0; p.kchunk = K;
  int grid = 0;
  PERO_REQUIRE(pero_launch_gemm_n512_lnb(p, T, ldt, rstd, gamma, beta, work, &grid, (hipStream_t)stream), "pero_gemm_resid_layernorm_bwd: shape not taken");
  pero_ln_bwd_reduce_launch(work, dgamma, dbeta, dxsum, grid, 512, (hipStream_t)stream);
  PERO_CHECK_LAUNCH("pero_gemm_resid_layernorm_bwd");
  return PERO_OK;
}
extern "C" int64_t pero_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t batch, int flags, int k_split, int in_dtype,
                                             int out_dtype) {
  if (!(flags & PERO_GEMM_ATOMIC) || batch != 1 || in_dtype != PERO_BF16 || out_dtype != PERO_F32 || (flags & PERO_GEMM_FORCE_GENERIC) || M <= 0 ||
      N <= 0 || K <= 0 || !splitk_takes_e256(M, N, K, flags))
    return 0;
  return pero_gemm_e256_splitk_ws_bytes(M, N, K, k_split);
}

extern "C" int pero_gemm(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* gate,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldg,
                         int64_t batch, int64_t batch_inner,
                         int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi,
                         float alpha, int flags, int k_split, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                         void* stream) {
  bool fused = false;
  PERO_REQUIRE(workspace_bytes >= 0 && (workspace || workspace_bytes == 0), "pero_gemm: workspace_bytes without a workspace");
  if (flags & PERO_GEMM_COLSUM)
    PERO_REQUIRE(bias && batch == 1 && !(flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM | PERO_GEMM_ROWDOT)),
                 "pero_gemm: PERO_GEMM_COLSUM needs the output pointer in `bias`, one problem, a stored result");
  if (flags & PERO_GEMM_ROWDOT)
    PERO_REQUIRE(bias && gate && batch == 1 && out_dtype == PERO_BF16 && N % 128 == 0 && !(flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM)),
                 "pero_gemm: PERO_GEMM_ROWDOT needs the output pointer in `bias`, the second matrix in `gate`, bf16 C, N %% 128 == 0");
  const int rc = gemm_dispatch(A, B, C, bias, residual, gate, M, N, K, lda, ldb, ldc, ldr, ldg, batch, batch_inner, sAo, sAi, sBo, sBi,
                               sCo, sCi, alpha, flags, k_split, in_dtype, out_dtype, workspace, workspace_bytes, stream, &fused);
  if (rc != PERO_OK || !(flags & (PERO_GEMM_COLSUM | PERO_GEMM_ROWDOT)) || fused) return rc;
  // kernels without the fused epilogue: a pass over C
  if (flags & PERO_GEMM_ROWDOT) return pero_rowdot_blocks(C, gate, (float*)bias, M, N, ldc, ldg, stream);
  return pero_colsum(C, (float*)bias, M, N, ldc, out_dtype, stream);
}

static int gemm_dispatch(const void* A, const void* B, void* C, const float* bias, const void* residual, const void* gate,
                         int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldg,
                         int64_t batch, int64_t batch_inner,
                         int64_t sAo, int64_t sAi, int64_t sBo, int64_t sBi, int64_t sCo, int64_t sCi,
                         float alpha, int flags, int k_split, int in_dtype, int out_dtype, void* workspace, int64_t workspace_bytes,
                         void* stream, bool* colsum_fused) {
  PERO_REQUIRE(A && B && C, "pero_gemm: null operand");
  PERO_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && batch_inner > 0, "pero_gemm: bad sizes M=%lld N=%lld K=%lld batch=%lld",
               (long long)M, (long long)N, (long long)K, (long long)batch);
  PERO_REQUIRE((in_dtype == PERO_F32 || in_dtype == PERO_BF16) && (out_dtype == PERO_F32 || out_dtype == PERO_BF16), "pero_gemm: bad dtype");
  PERO_REQUIRE(!((flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM)) && out_dtype != PERO_F32), "pero_gemm: ATOMIC/ACCUM need f32 C");
  PERO_REQUIRE(k_split >= 0 && (k_split == 1 || (flags & PERO_GEMM_ATOMIC)), "pero_gemm: k_split != 1 needs PERO_GEMM_ATOMIC");
  PERO_REQUIRE(batch < 65536, "pero_gemm: batch too large");
  const bool ta = flags & PERO_GEMM_TRANS_A, tb = flags & PERO_GEMM_TRANS_B;
  GemmP p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.resid = residual; p.gate = gate;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldg = ldg;
  p.sAo = sAo; p.sAi = sAi; p.sBo = sBo; p.sBi = sBi; p.sCo = sCo; p.sCi = sCi;
  p.binner = (int)batch_inner; p.alpha = alpha; p.flags = flags; p.kchunk = K;
  hipStream_t st = (hipStream_t)stream;
  // PERO_GEMM_COLSUM: only the r256 / v256 epilogues accumulate the column sums (`pc`); every other kernel gets `p`
  // without the flag and without the (output) bias pointer, and pero_gemm runs pero_colsum over C afterwards.
  // PERO_GEMM_ROWDOT is handled the same way (r256 only): `pc` keeps flag, gate and output pointer, `p` loses all three.
  const bool want_cs = flags & (PERO_GEMM_COLSUM | PERO_GEMM_ROWDOT);
  const bool want_rd = flags & PERO_GEMM_ROWDOT;
  const int cs_bits = flags & (PERO_GEMM_COLSUM | PERO_GEMM_ROWDOT);
  GemmP pc = p;
  if (want_cs) { flags &= ~(PERO_GEMM_COLSUM | PERO_GEMM_ROWDOT); p.flags = flags; p.bias = nullptr; if (want_rd) p.gate = nullptr; }

  const int esz_o = out_dtype == PERO_F32 ? 4 : 2;
  bool fast = in_dtype == PERO_BF16 && !(flags & 32) && M % T_BM == 0 && N % T_BN == 0 && K % T_BK == 0 &&
              lda % 8 == 0 && ldb % 8 == 0 && (ldc * esz_o) % 16 == 0 && aligned16(A) && aligned16(B) && aligned16(C) &&
              (sAo % 8 == 0) && (sAi % 8 == 0) && (sBo % 8 == 0) && (sBi % 8 == 0) && ((sCo * esz_o) % 16 == 0) &&
              ((sCi * esz_o) % 16 == 0) && (!residual || (ldr % 8 == 0 && aligned16(residual))) &&
              (!gate || (flags & PERO_GEMM_RELU_BITS) || (ldg % 8 == 0 && aligned16(gate))) && (!(residual || gate) || out_dtype == PERO_BF16 || true);
  if (fast && g_gemm_policy == 1) { flags |= PERO_GEMM_TILE128; p.flags = flags; pc.flags = flags | cs_bits; }
  const bool force128 = flags & PERO_GEMM_TILE128, force256 = flags & PERO_GEMM_TILE256;
  PERO_REQUIRE(!(flags & PERO_GEMM_MASK_TILED) || ((flags & PERO_GEMM_RELU_BITS) && N % 256 == 0), "pero_gemm: PERO_GEMM_MASK_TILED needs PERO_GEMM_RELU_BITS and N %% 256 == 0");
  if (flags & PERO_GEMM_RELU_BITS) {
    // bit-mask ReLU gate: only the e256 / r256 epilogues read or write it
    PERO_REQUIRE(fast && gate && batch == 1 && out_dtype == PERO_BF16 && !ta && !(flags & PERO_GEMM_ATOMIC) && !force128 && M % 256 == 0 &&
                 N % 128 == 0 && K % 32 == 0 && !(flags & PERO_GEMM_ROWDOT) && (g_gemm_policy == 0 || g_gemm_policy == 7 || g_gemm_policy == 20),
                 "pero_gemm: PERO_GEMM_RELU_BITS needs a bf16 product for the 256-row tile kernels (M %% 256, N %% 128, K %% 32, batch 1)");
  }
  if (fast) {
    const bool atomic = flags & PERO_GEMM_ATOMIC;
    const bool can256 = M % 256 == 0 && N % 256 == 0 && !force128;
    const long long t256 = can256 ? (M / 256) * (N / 256) * batch : 0;
    const long long t128 = (M / T_BM) * (N / T_BN) * batch;
    const int k_split_req = k_split;
    const bool auto_or = g_gemm_policy == 0;
    // the eight-phase persistent 256x256x64 kernel (gemm_e.hip): split-K weight gradients on long reductions ...
    if (atomic && out_dtype == PERO_F32 && batch == 1 && splitk_takes_e256(M, N, K, flags) &&
        pero_launch_gemm_e256(p, batch, k_split_req, ta, tb, true, st, -1, workspace, workspace_bytes)) {
      PERO_CHECK_LAUNCH("pero_gemm(e256 split-K)");
      return PERO_OK;
    }
    // the row-complete 128 x 512 tile (opt-in): N = 512 stored products with the plain / residual epilogue
    if (!atomic && g_gemm_nw && !want_cs && out_dtype == PERO_BF16 && pero_launch_gemm_n512(p, batch, ta, tb, false, st)) {
      PERO_CHECK_LAUNCH("pero_gemm(n512)");
      return PERO_OK;
    }
    // two workgroups per CU on 256 x 128 tiles (opt-in): stored products with few K-tiles per tile, plain / ReLU / bit-mask epilogues
    if (!atomic && g_gemm_d128 && out_dtype == PERO_BF16 && !want_rd && K <= g_gemm_d128 * 64LL && (auto_or || g_gemm_policy == 20) &&
        pero_launch_gemm_d128(pc, batch, ta, tb, false, st)) {
      *colsum_fused = want_cs;
      PERO_CHECK_LAUNCH("pero_gemm(d128)");
      return PERO_OK;
    }
    // ... and stored bf16 products with every fused epilogue
    if (!atomic && can256 && (g_gemm_policy == 20 || (auto_or && (force256 || (g_gemm_e256_min > 0 && t256 >= g_gemm_e256_min)))) &&
        pero_launch_gemm_e256(pc, batch, k_split, ta, tb, out_dtype == PERO_F32, st, -1)) {
      *colsum_fused = want_cs;
      PERO_CHECK_LAUNCH("pero_gemm(e256)");
      return PERO_OK;
    }
    PERO_REQUIRE(!(g_gemm_policy == 20 && (g_gemm_e_var & (8 | 64 | 128))), "pero_gemm: the stamp build did not take this product");  // its `gate` is a debug buffer
    if (atomic && k_split == 0) {
      long long ks = (g_splitk_items + t128 - 1) / t128;
      if (ks > K / 512) ks = K / 512;
      if (ks < 1) ks = 1;
      // powers of two (<= 8) or multiples of 8: lets the kernel place one k-slice per XCD
      if (ks >= 8) ks = ((ks + (g_splitk_nearest ? 4 : 7)) / 8) * 8;
      else if (ks > 4) ks = 8;
      else if (ks == 3) ks = 4;
      if (ks > K / 64) ks = 1;
      k_split = (int)ks;
    } else if (k_split < 1) {
      k_split = 1;
    }
    // split-K atomics: one 128x128x64 tile per workgroup
    if (!force128 && atomic && g_gemm_policy != 7 && pero_launch_gemm_o128(p, batch, k_split, ta, tb, out_dtype == PERO_F32, st)) {
      PERO_CHECK_LAUNCH("pero_gemm(bf16 o128)");
      return PERO_OK;
    }
    if (!force128 && !atomic && g_gemm_policy == 4 && !want_cs && !(flags & PERO_GEMM_RELU_BITS) &&
        pero_launch_gemm_o128(p, batch, k_split, ta, tb, out_dtype == PERO_F32, st)) {
      PERO_CHECK_LAUNCH("pero_gemm(bf16 o128)");
      return PERO_OK;
    }
    // 256x128x32 tiles, two workgroups per CU: stored products of small batches; carries the column-sum / row-dot / bit-mask epilogues
    if (!force128 && !atomic && g_gemm_policy != 4 && pero_launch_gemm_r256(want_cs ? pc : p, batch, k_split, ta, tb, out_dtype == PERO_F32, st)) {
      *colsum_fused = want_cs;
      PERO_CHECK_LAUNCH("pero_gemm(bf16 r256)");
      return PERO_OK;
    }
    PERO_REQUIRE(!(flags & PERO_GEMM_RELU_BITS), "pero_gemm: no kernel took the PERO_GEMM_RELU_BITS product (internal)");
    if (k_split > 1) {

      long long steps = K / T_BK;
      long long per = (steps + k_split - 1) / k_split;
      p.kchunk = per * T_BK;
      k_split = (int)((steps + per - 1) / per);
    }
    const long long nwork = (M / T_BM) * (N / T_BN) * batch * k_split;
    const int num_cus = pero_num_cus();
    const long long resident = 2LL * num_cus;  // 2 workgroups per CU (LDS-limited)
    dim3 grid((unsigned)(nwork < resident ? nwork : resident)), block(256);
    const int nbatch = (int)batch;
#define LAUNCH_FAST(TA_, TB_, OF_)                                                                                        \
  do {                                                                                                                    \
    PERO_LDS_ATTR((gemm_bf16_t128<TA_, TB_, OF_>), T_LDS_BYTES);                                                          \
    hipLaunchKernelGGL((gemm_bf16_t128<TA_, TB_, OF_>), grid, block, T_LDS_BYTES, st, p, nbatch, nwork);                                 \
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
