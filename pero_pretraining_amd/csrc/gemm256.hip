// 256 x 256 x 64 bf16 tile GEMM (gfx950).  Same contract as gemm_bf16_t128 (gemm.hip); used when the problem
// has enough 256x256 tiles to fill the chip.
//
// Why a second tile size: a 128x128 tile needs (128+128)*64*2 B = 32 KiB of operands per 512 MFMA-cycles; with
// two workgroups per CU that is 64 B/clk/CU of L2->LDS traffic, the whole vector-memory pipe of a CU (measured:
// "loads only" of the 128-tile kernel ran at ~30 TB/s aggregate, the L2 peak, and took as long as its MFMAs).
// A 256x256 tile halves the bytes per flop, so the LDS-DMA stream needs ~50 % of that pipe and can hide behind
// the matrix work.
//
// Structure: 8 waves (2 along M x 4 along N, 128 x 64 outputs each, 128 accumulator VGPRs), one workgroup per CU,
// two 64-KiB LDS buffers filled by global_load_lds_dwordx4 one k-tile ahead, v_mfma_f32_16x16x32_bf16, operand
// swizzles / transposed reads as in gemm.hip, persistent walk over the work items with the next item's first
// k-tile prefetched under the epilogue, direct epilogue from the accumulators (a lane owns 4 consecutive columns).
#include "gemm_common.hpp"

#define U_BM 256
#define U_BN 256
#define U_BK 64
#define U_OPBYTES (256 * 64 * 2)       // 32 KiB per operand tile
#define U_BUFBYTES (2 * U_OPBYTES)     // 64 KiB per stage
#define U_LDS_BYTES (2 * U_BUFBYTES)   // 128 KiB
#define U_EPI_PITCH (256 * 4 + 16)     // f32 staging pitch of the atomic epilogue (64-row chunks)

__device__ __forceinline__ bf8v ufrag_rowmajor(const unsigned char* base, int row, int ks, int lane) {
  const int r = row + (lane & 15);
  const int chunk = ks * 4 + (lane >> 4);
  return *(const bf8v*)(base + r * 128 + ((chunk ^ (r & 7)) << 4));
}
// K-major image [64 k-rows][256 cols]: 512-byte rows, 32-byte block index ^ fk(krow) (low 3 bits)
__device__ __forceinline__ bf8v ufrag_kmajor(const unsigned char* base, int col, int ks, int lane) {
  const int i = lane & 15;
  const int krow = ks * 32 + 8 * (lane >> 4) + (i >> 2);
  const unsigned char* a = base + krow * 512 + ((((col >> 4) ^ fk(krow))) << 5) + 8 * (i & 3);
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a + 4 * 512));
  typedef short s8v __attribute__((ext_vector_type(8)));
  s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8v, v);
}

// 32 KiB operand tile = 32 pieces of 1 KiB; wave w issues pieces i*8 + w (i = 0..3)
template <bool TR>
__device__ __forceinline__ void ustage_glds(const bf16raw* X, long long ld, long long tile0, long long k0,
                                            unsigned char* lds_base, int tid) {
  const bf16raw* p;
  long long step;
  if (!TR) {  // [rows][K]: piece = 8 rows x 128 B; thread -> row (tid >> 3) + 64 i, LDS slot tid & 7
    const int row = tid >> 3, chunk = (tid & 7) ^ (row & 7);
    p = X + (tile0 + row) * ld + k0 + chunk * 8;
    step = 64 * ld;
  } else {    // [K][rows]: piece = 2 k-rows x 512 B; thread -> k-row (tid >> 5) + 16 i, LDS slot tid & 31
    const int krow = tid >> 5, slot = tid & 31;
    const int chunk = ((((slot >> 1) ^ fk(krow))) << 1) | (slot & 1);
    p = X + (k0 + krow) * ld + tile0 + chunk * 8;
    step = 16 * ld;
  }
  unsigned char* dst = lds_base + (tid >> 6) * 1024;
#define GLDS16(src_, dst_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_), \
                                                            (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0)
  GLDS16(p, dst);
  GLDS16(p + step, dst + 8192);
  GLDS16(p + 2 * step, dst + 16384);
  GLDS16(p + 3 * step, dst + 24576);
#undef GLDS16
}

struct UWork { long long tm0, tn0, kbeg; int nk; const bf16raw* A; const bf16raw* B; long long coff; };

__device__ __forceinline__ UWork uwork_item(const GemmP& p, long long w, int ntn, int nt, int nbatch) {
  UWork it;
  const int lin = (int)(w % nt);
  const long long rest = w / nt;
  const int b = (int)(rest % nbatch), z = (int)(rest / nbatch);
  const int q = nt >> 3, r8 = nt & 7, xcd = lin & 7, loc = lin >> 3;
  const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
  it.tm0 = (long long)(id / ntn) * U_BM;
  it.tn0 = (long long)(id % ntn) * U_BN;
  const long long bo = b / p.binner, bi = b % p.binner;
  it.A = (const bf16raw*)p.A + bo * p.sAo + bi * p.sAi;
  it.B = (const bf16raw*)p.B + bo * p.sBo + bi * p.sBi;
  it.coff = bo * p.sCo + bi * p.sCi;
  it.kbeg = (long long)z * p.kchunk;
  long long kend = it.kbeg + p.kchunk;
  if (kend > p.K) kend = p.K;
  it.nk = (int)((kend - it.kbeg) / U_BK);
  return it;
}

template <bool TA, bool TB, bool OUTF32>
__global__ __launch_bounds__(512, 2) void gemm_bf16_t256(GemmP p, int nbatch, long long nwork) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // wave tile: rows wm*128 .. +127, cols wn*64 .. +63
  const int ntn = (int)(p.N / U_BN);
  const int nt = (int)(p.M / U_BM) * ntn;
  const bool atomic = OUTF32 && (p.flags & PERO_GEMM_ATOMIC);

  long long w = blockIdx.x;
  if (w >= nwork) return;
  UWork it = uwork_item(p, w, ntn, nt, nbatch);
  int buf = 0;
  ustage_glds<TA>(it.A, p.lda, it.tm0, it.kbeg, smem, tid);
  ustage_glds<TB>(it.B, p.ldb, it.tn0, it.kbeg, smem + U_OPBYTES, tid);
  bool first = true;

  while (true) {
    const long long wnext = w + gridDim.x;
    const bool have_next = wnext < nwork;
    UWork nx;
    if (have_next) nx = uwork_item(p, wnext, ntn, nt, nbatch);

    f4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = (f4v){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < it.nk; t++) {
      if (t == 0 && !first && !atomic) wait_vmcnt<8>();  // prefetched tile landed; >= 8 epilogue stores stay in flight
      else wait_vmcnt<0>();
      lds_barrier();
      const unsigned char* sa = smem + buf * U_BUFBYTES;
      const unsigned char* sb = sa + U_OPBYTES;
      unsigned char* da = smem + (buf ^ 1) * U_BUFBYTES;
      if (t + 1 < it.nk) {
        ustage_glds<TA>(it.A, p.lda, it.tm0, it.kbeg + (long long)(t + 1) * U_BK, da, tid);
        ustage_glds<TB>(it.B, p.ldb, it.tn0, it.kbeg + (long long)(t + 1) * U_BK, da + U_OPBYTES, tid);
      } else if (have_next && !atomic) {
        ustage_glds<TA>(nx.A, p.lda, nx.tm0, nx.kbeg, da, tid);
        ustage_glds<TB>(nx.B, p.ldb, nx.tn0, nx.kbeg, da + U_OPBYTES, tid);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf8v fb[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          fb[j] = TB ? ufrag_kmajor(sb, wn * 64 + j * 16, ks, lane) : ufrag_rowmajor(sb, wn * 64 + j * 16, ks, lane);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const bf8v fa = TA ? ufrag_kmajor(sa, wm * 128 + i * 16, ks, lane) : ufrag_rowmajor(sa, wm * 128 + i * 16, ks, lane);
#pragma unroll
          for (int j = 0; j < 4; j++)  // swapped operands: D[n][m]
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa, acc[i][j], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
    first = false;

    if (atomic) {
      // split-K partial tile -> f32 atomics, staged through LDS in 64-row chunks so that every atomic
      // wave-instruction adds 256 contiguous bytes
      float* C = (float*)p.C + it.coff;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        lds_barrier();
        if (wm == (c >> 1)) {
#pragma unroll
          for (int ii = 0; ii < 4; ii++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int i = 4 * (c & 1) + ii;
              const int m = ii * 16 + (lane & 15);           // row inside the 64-row chunk
              const int n = wn * 64 + j * 16 + (lane >> 4) * 4;
              *(f4v*)(smem + m * U_EPI_PITCH + n * 4) = acc[i][j];
            }
        }
        lds_barrier();
#pragma unroll 2
        for (int rr = 0; rr < 8; rr++) {
          const int row = wave + 8 * rr;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int col = lane + 64 * j;
            const float v = *(const float*)(smem + row * U_EPI_PITCH + col * 4) * p.alpha;
            atomicAdd(C + (it.tm0 + c * 64 + row) * p.ldc + it.tn0 + col, v);
          }
        }
      }
      if (have_next) {
        lds_barrier();
        ustage_glds<TA>(nx.A, p.lda, nx.tm0, nx.kbeg, smem, tid);
        ustage_glds<TB>(nx.B, p.ldb, nx.tn0, nx.kbeg, smem + U_OPBYTES, tid);
        buf = 0;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const long long n = it.tn0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        f4v bias = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bias = *(const f4v*)(p.bias + n);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const long long m = it.tm0 + wm * 128 + i * 16 + (lane & 15);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = acc[i][j][e] * p.alpha + bias[e];
          if (p.resid) {
            const uint2 rr = *(const uint2*)((const bf16raw*)p.resid + it.coff + m * p.ldr + n);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          }
          if (p.flags & PERO_GEMM_RELU) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
          }
          if (p.gate) {
            const uint2 gg = *(const uint2*)((const bf16raw*)p.gate + it.coff + m * p.ldg + n);
            if (!(__uint_as_float(gg.x << 16) > 0.f)) v[0] = 0.f;
            if (!(__uint_as_float(gg.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
            if (!(__uint_as_float(gg.y << 16) > 0.f)) v[2] = 0.f;
            if (!(__uint_as_float(gg.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
          }
          if (OUTF32) {
            float* C = (float*)p.C + it.coff + m * p.ldc + n;
            if (p.flags & PERO_GEMM_ACCUM) {
              const f4v o = *(const f4v*)C;
#pragma unroll
              for (int e = 0; e < 4; e++) v[e] += o[e];
            }
            *(f4v*)C = (f4v){v[0], v[1], v[2], v[3]};
          } else {
            uint2 o;
            o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *(uint2*)((bf16raw*)p.C + it.coff + m * p.ldc + n) = o;
          }
        }
      }
    }
    if (!have_next) break;
    w = wnext;
    it = nx;
  }
}

bool pero_launch_gemm_t256(const GemmP& p0, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (p0.M % U_BM || p0.N % U_BN || p0.K % U_BK) return false;
  GemmP p = p0;
  if (k_split > 1) {
    const long long steps = p.K / U_BK;
    const long long per = (steps + k_split - 1) / k_split;
    p.kchunk = per * U_BK;
    k_split = (int)((steps + per - 1) / per);
  } else {
    p.kchunk = p.K;
  }
  const long long nwork = (p.M / U_BM) * (p.N / U_BN) * batch * k_split;
  static int num_cus = 0;
  if (!num_cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    hipGetDevice(&dev);
    hipGetDeviceProperties(&prop, dev);
    num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  dim3 grid((unsigned)(nwork < num_cus ? nwork : num_cus)), block(512);
  const int nbatch = (int)batch;
#define LAUNCH_U(TA_, TB_, OF_)                                                                                            \
  do {                                                                                                                     \
    static bool attr_set = false;                                                                                          \
    if (!attr_set) {                                                                                                       \
      hipFuncSetAttribute((const void*)gemm_bf16_t256<TA_, TB_, OF_>, hipFuncAttributeMaxDynamicSharedMemorySize, U_LDS_BYTES); \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_t256<TA_, TB_, OF_>), grid, block, U_LDS_BYTES, st, p, nbatch, nwork);                   \
  } while (0)
  if (!ta && !tb) { if (out_f32) LAUNCH_U(false, false, true); else LAUNCH_U(false, false, false); }
  else if (!ta && tb) { if (out_f32) LAUNCH_U(false, true, true); else LAUNCH_U(false, true, false); }
  else if (ta && tb) { if (out_f32) LAUNCH_U(true, true, true); else LAUNCH_U(true, true, false); }
  else { if (out_f32) LAUNCH_U(true, false, true); else LAUNCH_U(true, false, false); }
#undef LAUNCH_U
  return true;
}
