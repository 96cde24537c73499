// Fused multi-head attention for the encoder layers (torch SDPA inside TransformerEncoderLayer._sa_block,
// reference models/transformers.py:36-43,86): softmax(q k^T / sqrt(hd)) v over all S keys of a line, no masks.
//
// bf16, head_dim 128, S a multiple of 128, operating directly on the packed qkv (N*S, 3d) tensor.
//
// Forward: one workgroup = 128 queries of one (line, head); 4 waves x 32 queries.  Keys are processed in
// tiles of 128 (online softmax across tiles).  Everything is computed TRANSPOSED so that a query lives on a
// LANE and keys / head-dim live on registers (guide section 3 "an accumulator tile as the next MFMA's operand"):
//   S^T tile (32 keys x 32 q)  = mfma_32x32x16(A = K rows from LDS, B = Q^T from registers)
//   row max / sum of a query    = in-lane reduction over its 64 score registers + ONE lane^32 exchange
//   P^T (bf16, packed in place) = the B operand of  O^T (32 d x 32 q) += mfma(A = V^T via ds_read_b64_tr_b16, B = P^T)
// so the probabilities never leave registers, the softmax statistics and the O rescale are lane-local, and the
// S x S score matrix is never written to memory (the unfused path moves 6*S^2 bytes per (line, head)).
// K and V tiles arrive by LDS-DMA; K image XOR-swizzled for conflict-free ds_read_b128 (256-byte rows), V image
// swizzled in 64-byte blocks for conflict-free transposed reads.  2 workgroups per CU (64 KiB LDS each).
#include "common.hpp"

#define AT_HALF_BYTES 16384   // 64 rows x 256 B
#ifndef AT_NO_PRIO
#define AT_PRIO(n_) __builtin_amdgcn_s_setprio(n_)
#else
#define AT_PRIO(n_)
#endif
#define AT_SUB_BYTES 8192     // 32 rows x 256 B
// Timing-only ablation builds of the backward kernels (tools/attn_ablate.sh; -DAT_ABL=mask, results wrong by design):
//   1 no LDS-DMA inside the loops (the prologue's tiles are reused), 2 no exponentials, 4 no gradient MFMAs (dQ / dK / dV products),
//   8 no output tiles (attn_store_tile), 16 no score MFMAs (S / dP), 32 output tiles staged but not stored, 64 no column sums (bias gradient)
#ifndef AT_COLSUM_MFMA
#define AT_COLSUM_MFMA 1   // column sums of the backward's output tiles by MFMA (0: round 3's vector-ALU sums + LDS reduction)
#endif
#ifndef AT_ABL
#define AT_ABL 0
#endif
#define AT_DKV2_LDS (4 * AT_SUB_BYTES + 128 * 128 * 2 + 512)  // Q / dO stages x 2, V tile, row statistics x 2
#define AT_TILE_BYTES (128 * 128 * 2)  // 32 KiB: 128 keys x 128 head-dim bf16

// piece p (0..31) of a 128x128 bf16 tile = rows 4p..4p+3 (1 KiB); lane -> (row, 16-byte slot)
template <bool VIMG>
__device__ __forceinline__ void attn_glds_tile(const bf16raw* g, long long ld, unsigned char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = wave + 4 * i;
    const int row = 4 * p + (lane >> 4), slot = lane & 15;
    // K image: 16-byte chunk index ^ (row & 15).  V image: 64-byte block index ^ (row & 3).
    const int chunk = VIMG ? ((((slot >> 2) ^ (row & 3)) << 2) | (slot & 3)) : (slot ^ (row & 15));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 8),
                                     (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf8v attn_k_frag(const unsigned char* kimg, int key, int ks, int h5) {
  const int chunk = 2 * ks + h5;
  return *(const bf8v*)(kimg + key * 256 + ((chunk ^ (key & 15)) << 4));
}
// A operand of O^T += V^T P^T for k-step (kb .. kb+15) and d-tile dt: element j <- V[kb + 8(j>>2) + 4h + (j&3)][dt*32 + (lane&31)]
__device__ __forceinline__ bf8v attn_vT_frag(const unsigned char* vimg, int kb, int dt, int lane) {
  const int i = lane & 15, g1 = (lane >> 4) & 1, h5 = lane >> 5;
  const int key = kb + 4 * h5 + (i >> 2);  // (key & 3) == (i >> 2) for both reads (kb, 4*h5, +8 are multiples of 4)
  const unsigned char* a = vimg + key * 256 + ((dt ^ (key & 3)) << 6) + g1 * 32 + (i & 3) * 8;
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a + 8 * 256));
  typedef short s8v __attribute__((ext_vector_type(8)));
  s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8v, v);
}
__device__ __forceinline__ bf8v pack8(const f16v& a, int s) {
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  u4v u = {pack2bf(a[8 * s + 0], a[8 * s + 1]), pack2bf(a[8 * s + 2], a[8 * s + 3]), pack2bf(a[8 * s + 4], a[8 * s + 5]),
           pack2bf(a[8 * s + 6], a[8 * s + 7])};
  return __builtin_bit_cast(bf8v, u);
}

// Workgroup -> ((line, head), block) so that the blocks of one (line, head) - which read the same K / V (or Q / dO) rows -
// run on ONE XCD, next to each other in dispatch order (hardware places workgroup b on XCD b & 7): the second reader then
// hits that XCD's L2 instead of fetching the rows again through the fabric.  Needs (lines x heads) % 8 == 0.
__device__ __forceinline__ void attn_block_map(int bid, int nblk, int nlh, int& lh, int& blk) {
  if ((nlh & 7) == 0) {
    const int xcd = bid & 7, u = bid >> 3;
    lh = (u / nblk) * 8 + xcd;
    blk = u % nblk;
  } else {
    lh = bid / nblk;
    blk = bid % nblk;
  }
}
// hpb = heads per workgroup: the (head, key tile) pairs of `hpb` heads of one line are walked as ONE stream, so the
// LDS-DMA of the next head's first K / V tile and the global loads of its Q rows run under the current head's last
// tile.  At S = 256 a (line, head, 128 queries) unit is only two key tiles: measured 439 TFLOP/s against 855 at S = 2048
// with identical inner loops - the difference was the per-workgroup prologue / epilogue.
__global__ __launch_bounds__(256, 2) void attn_fwd_k(const bf16raw* qkv, bf16raw* out, float* lse2, int S, int nh, int hpb, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + AT_TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nqb = S >> 7, ngrp = nh / hpb;
  int lg, qb;
  attn_block_map(blockIdx.x, nqb, gridDim.x / nqb, lg, qb);
  const int line = lg / ngrp, head0 = (lg % ngrp) * hpb;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* lbase = qkv + (long long)line * S * ld;  // + head * 128 : q ; + d : k ; + 2d : v
  const int q = qb * 128 + wave * 32 + r;  // this lane's query (both lane halves hold the same query)
  const int nkt = S >> 7, units = hpb * nkt;

  attn_glds_tile<false>(lbase + head0 * 128 + d, ld, kimg, wave, lane);
  attn_glds_tile<true>(lbase + head0 * 128 + 2 * d, ld, vimg, wave, lane);

  bf8v qf[8];
  {
    const bf16raw* qrow = lbase + head0 * 128 + (long long)q * ld + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) qf[ks] = *(const bf8v*)(qrow + 16 * ks);
  }
  f16v o[4];
#pragma unroll
  for (int t = 0; t < 4; t++) o[t] = (f16v){0};
  float m = -INFINITY, l = 0.f;

  for (int u = 0; u < units; u++) {
    const int head = head0 + u / nkt, kt = u % nkt;
    // K(u) (and the head's Q rows) must have landed; V(u) - the NEWEST eight DMA instructions, issued at the end of the
    // previous unit - may still fly: it is waited for after the scores, one phase later (in-order vmcnt: a full wait here
    // exposed V's whole latency at every unit)
    if (u > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // K(u) landed
    f16v s[4];
    AT_PRIO(1);   // this wave's MFMA cluster goes ahead of the other wave's softmax instructions on the same SIMD
#pragma unroll
    for (int t = 0; t < 4; t++) {
      s[t] = (f16v){0};
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_k_frag(kimg, t * 32 + r, ks, h5), qf[ks], s[t], 0, 0, 0);
    }
    AT_PRIO(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own part of V(u)
    __syncthreads();  // every wave is done with the K image; V(u) landed
    if (u + 1 < units) {
      const int nhd = head0 + (u + 1) / nkt, nkt_i = (u + 1) % nkt;
      attn_glds_tile<false>(lbase + nhd * 128 + d + (long long)nkt_i * 128 * ld, ld, kimg, wave, lane);
    }

    // ---- online softmax, all lane-local except one lane^32 exchange per reduction
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) mx = fmaxf(mx, s[t][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);  // exp2(-inf) = 0 on a head's first tile
    const float mc = mn * c;
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][e], c, -mc));
        s[t][e] = p;
        ps += p;
      }
    ps += __shfl_xor(ps, 32, 64);
    l = l * alpha + ps;
    m = mn;
    if (kt != 0) {  // (a head's first tile: O is still zero)
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[t][e] *= alpha;
    }

    // ---- O^T += V^T P^T
    AT_PRIO(1);
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        const bf8v pf = pack8(s[t], sub);
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(attn_vT_frag(vimg, t * 32 + sub * 16, dt, lane), pf, o[dt], 0, 0, 0);
      }
    }
    AT_PRIO(0);
    if (kt == nkt - 1) {
      // ---- head finished: O[q][d] = o[dt][reg] / l, staged through the (now free) V image so that HBM sees whole
      // 256-byte rows in 16-byte lanes: the direct form (16 scattered 8-byte stores per lane) cost 23 % of the kernel
      // at S = 256 - and vmcnt makes the next tile's DMA wait for them.  Image: 128 rows x 256 B, 8-byte granule index
      // XORed with (row & 31): conflict-free ds_write_b64 (lanes = rows) and ds_read_b128 (lanes = chunks of a row).
      __syncthreads();  // every wave is done with the V image
      const float inv = 1.0f / l;
      int qrow_l = wave * 32 + r, tid_l = tid;
      asm volatile("" : "+v"(qrow_l), "+v"(tid_l));  // keep the 24 staging addresses out of the unit loop's live set
#pragma unroll
      for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          uint2 w;
          w.x = pack2bf(o[dt][4 * g4 + 0] * inv, o[dt][4 * g4 + 1] * inv);
          w.y = pack2bf(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv);
          const int g = dt * 8 + 2 * g4 + h5;  // granule of d = dt*32 + 8*g4 + 4*h5
          *(uint2*)(vimg + qrow_l * 256 + ((g ^ (qrow_l & 31)) << 3)) = w;
        }
      if (h5 == 0) lse2[((long long)line * nh + head) * S + q] = m * c + __builtin_amdgcn_logf(l);  // base-2 LSE of c*scores
      __syncthreads();
      {
        const int ch = tid_l & 15;
#pragma unroll 2
        for (int i = 0; i < 8; i++) {
          const int row = (tid_l >> 4) + 16 * i;
          const int x = row & 31;
          uint4 v = *(const uint4*)(vimg + row * 256 + ((ch ^ (x >> 1)) << 4));
          if (x & 1) { const unsigned t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
          *(uint4*)(out + ((long long)line * S + qb * 128 + row) * d + head * 128 + ch * 8) = v;
        }
      }
      if (u + 1 < units) {  // next head: fresh statistics, its Q rows (the loads complete under the loop-top wait)
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = (f16v){0};
        m = -INFINITY;
        l = 0.f;
        const bf16raw* qrow = lbase + (head + 1) * 128 + (long long)q * ld + 8 * h5;
#pragma unroll
        for (int ks = 0; ks < 8; ks++) qf[ks] = *(const bf8v*)(qrow + 16 * ks);
      }
    }
    if (u + 1 < units) {
      __syncthreads();  // every wave is done with the V image (and with the O staging reads)
      const int nhd = head0 + (u + 1) / nkt, nkt_i = (u + 1) % nkt;
      attn_glds_tile<true>(lbase + nhd * 128 + 2 * d + (long long)nkt_i * 128 * ld, ld, vimg, wave, lane);
    }
  }
}


// heads per workgroup: as many as keep >= 2 workgroups per CU busy
static int attn_heads_per_block(long long N, long long S, long long nh) {
  const int num_cus = pero_num_cus();
  int hpb = 1;
  for (int cand = (int)nh; cand >= 1; cand--)
    if (nh % cand == 0 && N * (S / 128) * (nh / cand) >= 2LL * num_cus) { hpb = cand; break; }
  return hpb;
}

extern int g_attn_pipe;
__global__ void attn_fwd_p_k(const bf16raw* qkv, bf16raw* out, float* lse2, int S, int nh, int hpb, float c);   // (defined behind the asm helpers)
extern "C" int pero_attention_fwd(const void* qkv, void* out, float* lse, int64_t N, int64_t S, int64_t num_heads,
                                  int64_t head_dim, int dtype, void* stream) {
  PERO_REQUIRE(qkv && out && lse, "pero_attention_fwd: null pointer");
  PERO_REQUIRE(dtype == PERO_BF16 && head_dim == 128 && S % 128 == 0 && S > 0 && N > 0 && num_heads > 0,
               "pero_attention_fwd: fused kernel needs bf16, head_dim 128, S %% 128 == 0 (got hd=%lld S=%lld)", (long long)head_dim, (long long)S);
  PERO_REQUIRE(aligned16(qkv) && aligned16(out), "pero_attention_fwd: 16-byte alignment");
  PERO_LDS_ATTR(attn_fwd_k, 2 * AT_TILE_BYTES);
  PERO_LDS_ATTR(attn_fwd_p_k, 2 * AT_TILE_BYTES);
  const float c = (float)(1.4426950408889634 / sqrt((double)head_dim));
  const int hpb = attn_heads_per_block(N, S, num_heads);
  if (g_attn_pipe)
    hipLaunchKernelGGL(attn_fwd_p_k, dim3((unsigned)(N * (num_heads / hpb) * (S / 128))), dim3(256), 2 * AT_TILE_BYTES, (hipStream_t)stream,
                       (const bf16raw*)qkv, (bf16raw*)out, lse, (int)S, (int)num_heads, hpb, c);
  else
    hipLaunchKernelGGL(attn_fwd_k, dim3((unsigned)(N * (num_heads / hpb) * (S / 128))), dim3(256), 2 * AT_TILE_BYTES, (hipStream_t)stream,
                       (const bf16raw*)qkv, (bf16raw*)out, lse, (int)S, (int)num_heads, hpb, c);
  PERO_CHECK_LAUNCH("pero_attention_fwd");
  return PERO_OK;
}

// =================================================================================================
// Backward.  Two kernels, each recomputing S^T / P from q, k and the saved base-2 log-sum-exp (no S x S
// tensor is ever stored):
//   attn_bwd_dq_k : workgroup = 128 queries of a (line, head), query on the lane, sweeps the keys in 32-key
//                   sub-tiles: S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D) scale, dQ^T += K^T dS^T.
//                   Also writes D[q] = sum_d dO[q][d] O[q][d] for the second kernel.
//   attn_bwd_dkv_k: workgroup = 128 keys, key on the lane, sweeps the queries in 32-query sub-tiles:
//                   S = Q K^T, P, dP = dO V^T, dS; dV^T += dO^T P, dK^T += Q^T dS.
// No atomics, deterministic; costs 7 MFMA products instead of the minimal 5 (attention is ~8 % of the step's
// FLOPs).  All LDS tiles use ONE dual-use image (guide T10 image (b)): 256-byte rows, 16-byte chunk index XORed
// with f(row) = ((row&3)<<2)|((row>>2)&3): conflict-free both for ds_read_b128 row reads (32x32x16 A operand)
// and for ds_read_b64_tr_b16 transposed reads, so K (dq kernel) and Q, dO (dkv kernel) are staged once.
// =================================================================================================
__device__ __forceinline__ int img_f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ void attn_glds_img(const bf16raw* g, long long ld, unsigned char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = wave + 4 * i;
    const int row = 4 * p + (lane >> 4), slot = lane & 15;
    const int chunk = slot ^ img_f(row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 8),
                                     (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
  }
}
// 64-row half image (16 KiB): same layout, pieces 0..15
__device__ __forceinline__ void attn_glds_half(const bf16raw* g, long long ld, unsigned char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p = wave + 4 * i;
    const int row = 4 * p + (lane >> 4), slot = lane & 15;
    const int chunk = slot ^ img_f(row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 8),
                                     (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
  }
}
// 32-row stage (8 KiB): pieces 0..7
__device__ __forceinline__ void attn_glds_sub(const bf16raw* g, long long ld, unsigned char* lds, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = wave + 4 * i;
    const int row = 4 * p + (lane >> 4), slot = lane & 15;
    const int chunk = slot ^ img_f(row);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 8),
                                     (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
  }
}
// A operand, row-wise: lane holds M[row][16*ks + 8*h5 .. +8]
__device__ __forceinline__ bf8v img_row_frag(const unsigned char* img, int row, int ks, int h5) {
  return *(const bf8v*)(img + row * 256 + (((2 * ks + h5) ^ img_f(row)) << 4));
}
// A operand, transposed: element j <- M[rb + 8(j>>2) + 4h + (j&3)][dt*32 + (lane&31)]   (rb multiple of 16)
__device__ __forceinline__ bf8v img_tr_frag(const unsigned char* img, int rb, int dt, int lane) {
  const int i = lane & 15, g1 = (lane >> 4) & 1, h5 = lane >> 5;
  const int row = rb + 4 * h5 + (i >> 2);
  const int ch = 4 * dt + 2 * g1 + ((i & 3) >> 1);
  const unsigned char* a = img + row * 256 + ((ch ^ img_f(row)) << 4) + 8 * (i & 1);
  // second read: row + 8 -> (row&3) unchanged, (row>>2)&3 flips bit 1: f(row+8) = f(row) ^ 2
  const unsigned char* b = img + (row + 8) * 256 + ((ch ^ img_f(row + 8)) << 4) + 8 * (i & 1);
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, b));
  typedef short s8v __attribute__((ext_vector_type(8)));
  s8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8v, v);
}


// =================================================================================================
// Software-pipelined operand reads (round 3).  hipcc compiles the loops above to  read -> s_waitcnt lgkmcnt(0) -> MFMA  pairs - one LDS
// round trip exposed per one or two 32-cycle MFMAs - and, worse, puts an `s_waitcnt vmcnt(0)` in front of the first LDS read that
// follows an LDS-DMA (it cannot tell the DMA's destination from the tile being read), which makes the "prefetch" of the next tile a
// wait in the middle of the current one (ablation, tools/attn_ablate.py: the loop's DMA cost 24 % of the backward).  The `_p` bodies
// below issue every fragment read by inline asm the compiler neither waits for nor orders, SEVEN fragments ahead of the MFMA that
// consumes them, each MFMA tied to a counted `s_waitcnt lgkmcnt(n)` through its fragment register (LDS operations retire in order,
// so n = the LDS instructions issued after the fragment's own).  One sequence of reads runs through a whole LDS stage: the
// transposed fragments of the gradient products are in flight while the exponentials run, the next sub-tile's row fragments while
// the gradient products run.  Same MFMAs in the same order: results are bit-identical to the bodies above.
// =================================================================================================
#include <type_traits>
template <int I, int N, typename F>
__device__ __forceinline__ void at_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    at_static_for<I + 1, N>(f);
  }
}
typedef int at_i2v __attribute__((ext_vector_type(2)));
typedef unsigned at_u2v __attribute__((ext_vector_type(2)));
typedef unsigned at_u4v __attribute__((ext_vector_type(4)));
typedef int at_i4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned at_lds_addr(const void* p) { return (unsigned)(unsigned long long)LDS_PTR(const unsigned char, p); }
// address = per-lane offset (a loop-invariant VGPR) + uniform base (an SGPR: the LDS stage), added right in front of the read: the
// compiler otherwise hoists every (stage + offset) sum out of the loops into its own VGPR and spills (20-80 registers in these bodies)
template <int OFF, typename T>
__device__ __forceinline__ void at_rd128(T& d, unsigned lane_off, unsigned base) {
  static_assert(sizeof(T) == 16, "ds_read_b128");
  unsigned a;
  asm volatile("v_add_u32 %1, %3, %2\n\tds_read_b128 %0, %1 offset:%4" : "=v"(d), "=&v"(a) : "v"(lane_off), "s"(base), "i"(OFF) : "memory");
}
// one transposed fragment = two ds_read_b64_tr_b16 (rows x and x + 8 of the image), joined without register copies
template <int OFFA, int OFFB>
__device__ __forceinline__ void at_rdtr(bf8v& d, unsigned off_a, unsigned off_b, unsigned base) {
  at_i2v lo, hi;
  unsigned a, b;
  asm volatile("v_add_u32 %2, %6, %4\n\tv_add_u32 %3, %6, %5\n\tds_read_b64_tr_b16 %0, %2 offset:%7\n\tds_read_b64_tr_b16 %1, %3 offset:%8"
               : "=&v"(lo), "=&v"(hi), "=&v"(a), "=&v"(b) : "v"(off_a), "v"(off_b), "s"(base), "i"(OFFA), "i"(OFFB) : "memory");
  d = __builtin_bit_cast(bf8v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
// plain forms: the complete LDS address in a register
template <int OFF, typename T>
__device__ __forceinline__ void at_rd128a(T& d, unsigned addr) {
  static_assert(sizeof(T) == 16, "ds_read_b128");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFFA, int OFFB>
__device__ __forceinline__ void at_rdtra(bf8v& d, unsigned addr_a, unsigned addr_b) {
  at_i2v lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5"
               : "=&v"(lo), "=&v"(hi) : "v"(addr_a), "v"(addr_b), "i"(OFFA), "i"(OFFB) : "memory");
  d = __builtin_bit_cast(bf8v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
// the same with the per-lane offset XORed by a compile-time constant first: the swizzled fragment offsets of one lane differ from
// each other only by such a constant (k-step: 32 * ks, head-dim tile: 64 * dt), so ONE register per fragment kind serves all of them
template <int OFF, int KX, typename T>
__device__ __forceinline__ void at_rd128x(T& d, unsigned lane_off, unsigned base) {
  static_assert(sizeof(T) == 16, "ds_read_b128");
  unsigned a;
  asm volatile("v_xor_b32 %1, %4, %2\n\tv_add_u32 %1, %3, %1\n\tds_read_b128 %0, %1 offset:%5" : "=v"(d), "=&v"(a) : "v"(lane_off), "s"(base), "i"(KX), "i"(OFF) : "memory");
}
template <int OFFA, int OFFB, int KX>
__device__ __forceinline__ void at_rdtrx(bf8v& d, unsigned off_a, unsigned off_b, unsigned base) {
  at_i2v lo, hi;
  unsigned a, b;
  asm volatile("v_xor_b32 %2, %7, %4\n\tv_xor_b32 %3, %7, %5\n\tv_add_u32 %2, %6, %2\n\tv_add_u32 %3, %6, %3\n\tds_read_b64_tr_b16 %0, %2 offset:%8\n\tds_read_b64_tr_b16 %1, %3 offset:%9"
               : "=&v"(lo), "=&v"(hi), "=&v"(a), "=&v"(b) : "v"(off_a), "v"(off_b), "s"(base), "i"(KX), "i"(OFFA), "i"(OFFB) : "memory");
  d = __builtin_bit_cast(bf8v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
}
template <int N, typename T>
__device__ __forceinline__ void at_wait_lgkm(T& f) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(N) : "memory");
}
template <int N, typename T>
__device__ __forceinline__ void at_wait_lgkm2(T& f, T& g) {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f), "+v"(g) : "i"(N) : "memory");
}

// Epilogue of the backward kernels: a 128 x 128 gradient tile held as acc[dt][e] (row = this lane's query / key
// `wave * 32 + r`, columns d = dt*32 + 8*(e>>2) + 4*h5 + (e&3)) goes through LDS as bf16 rows and leaves in 16-byte row
// segments (the direct form was 16 scattered 8-byte stores per lane); the staged rows also give the tile's column sums -
// this (line, head) block's share of in_proj's bias gradient - for 128 (x2) atomics instead of a pass over dqkv.
// Image: 128 rows x 256 B, 8-byte granule index XORed with (row & 31): conflict-free ds_write_b64 and ds_read_b128.
__device__ __forceinline__ void attn_store_tile(const f16v (&acc)[4], unsigned char* stg, bf16raw* out_base, long long ld,
                                                float* colsum, int tid, int wave, int r, int h5) {
  if (AT_ABL & 8) {  // timing-only: keep the accumulators alive, write nothing
    float keep = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; dt++) keep += acc[dt][0] + acc[dt][15];
    if (keep == 1.2345e-33f) out_base[0] = 1;
    return;
  }
  __syncthreads();  // the staging region is free (every wave is past its last tile read)
  const int row_w = wave * 32 + r;
#pragma unroll
  for (int dt = 0; dt < 4; dt++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      uint2 w;
      w.x = pack2bf(acc[dt][4 * g4 + 0], acc[dt][4 * g4 + 1]);
      w.y = pack2bf(acc[dt][4 * g4 + 2], acc[dt][4 * g4 + 3]);
      const int g = dt * 8 + 2 * g4 + h5;
      *(uint2*)(stg + row_w * 256 + ((g ^ (row_w & 31)) << 3)) = w;
    }
  __syncthreads();
#if AT_COLSUM_MFMA
  const int ch = tid & 15;
#pragma unroll 2
  for (int i = 0; i < 8; i++) {
    const int row = (tid >> 4) + 16 * i;
    const int x = row & 31;
    uint4 v = *(const uint4*)(stg + row * 256 + ((ch ^ (x >> 1)) << 4));
    if (x & 1) { const unsigned t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
    if (!(AT_ABL & 32)) *(uint4*)(out_base + (long long)row * ld + ch * 8) = v;
    else if (v.x == 0x12345678u) out_base[0] = 1;   // timing-only build: the staged values stay live, nothing is stored
  }
  if (colsum && !(AT_ABL & 64)) {
    // Column sums of the staged tile (this (line, head) block's share of in_proj's bias gradient) on the MATRIX pipe, which idles through
    // the epilogue (round 4; lh_store_matrix's trick): wave w takes the 32 columns 32 w .., ones (32 x 16) times the 16 x 32 block of the
    // image read back TRANSPOSED (ds_read_b64_tr_b16: the row index becomes the MFMA's k - any order of the rows inside a k-step gives the
    // same sum), accumulated over the eight row blocks: every lane n then holds the sum of column 32 w + (n & 31), exact in f32, and writes
    // it - no cross-lane shuffles, no second pass through LDS, no barriers.  (As 128 vector adds + 16 shuffles per thread + an LDS
    // reduction over the four waves behind two barriers the sums were 7 % of the backward: tools/attn_ablate.py, mask 64.)
    const int lane = tid & 63;
    const int ti = lane & 15, tg = (lane >> 4) & 1;
    const int g = 8 * wave + 4 * tg + (ti & 3);                 // 8-byte granule (4 columns) this lane supplies
    const int q0 = 8 * h5 + (ti >> 2);                          // row inside a 16-row block; the second read takes row + 4
    const __bf16 one = (__bf16)1.0f;
    const bf8v ones = {one, one, one, one, one, one, one, one};
    f16v cs = {0};
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int ra = 16 * ks + q0, rb = ra + 4;
      const bf8v frag = lds_tr16_pair(stg + ra * 256 + ((g ^ (ra & 31)) << 3), stg + rb * 256 + ((g ^ (rb & 31)) << 3));
      cs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, frag, cs, 0, 0, 0);
    }
    if (lane < 32) colsum[32 * wave + lane] = cs[0];
  }
#else   // round 3: vector-ALU column sums (kept for the A/B: -DAT_COLSUM_MFMA=0)
  const int ch = tid & 15;
  float cs[8];  // column sums of this thread's 8 columns (chunk ch) over its 8 rows, taken from the values on their way out
#pragma unroll
  for (int e = 0; e < 8; e++) cs[e] = 0.f;
#pragma unroll 2
  for (int i = 0; i < 8; i++) {
    const int row = (tid >> 4) + 16 * i;
    const int x = row & 31;
    uint4 v = *(const uint4*)(stg + row * 256 + ((ch ^ (x >> 1)) << 4));
    if (x & 1) { const unsigned t0 = v.x, t1 = v.y; v.x = v.z; v.y = v.w; v.z = t0; v.w = t1; }
    if (!(AT_ABL & 32)) *(uint4*)(out_base + (long long)row * ld + ch * 8) = v;
    else if (v.x == 0x12345678u) out_base[0] = 1;   // timing-only build: the staged values stay live, nothing is stored
    if (colsum && !(AT_ABL & 64)) {
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) { cs[2 * k] += __uint_as_float(w[k] << 16); cs[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
    }
  }
  if (colsum && !(AT_ABL & 64)) {
    // the 16 threads with this chunk: lanes l, l ^ 16, l ^ 32, l ^ 48 of each wave (two exchanges per value), then the four waves
    // through LDS; one partial row per workgroup and gradient, summed by attn_bias_reduce_k (atomics into the 128 addresses of a
    // head from its 512 workgroups ran ~50 us longer per kernel than this)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      cs[e] += __shfl_xor(cs[e], 16, 64);
      cs[e] += __shfl_xor(cs[e], 32, 64);
    }
    __syncthreads();  // every thread is done reading the staged tile
    float* red = (float*)stg;
    if ((tid & 63) < 16) {
#pragma unroll
      for (int e = 0; e < 8; e++) red[wave * 128 + ch * 8 + e] = cs[e];
    }
    __syncthreads();
    if (tid < 128) colsum[tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
  }
#endif
}

// dbias[which * d + head * 128 + c] += sum over the workgroups (line, block) of partial[which][(lh, blk)][c], lh = line * nh + head.
// Grid (heads, 3, slices of the workgroup list): one atomic per address and slice.  64 slices at >= 4096 workgroups per head (16 slices
// were 192 blocks of two waves for 12.6 MB of partial rows: 37 us, latency-bound).
__global__ __launch_bounds__(128) void attn_bias_reduce_k(const float* partial, float* dbias, int nlines, int nh, int nblk) {
  const int c = threadIdx.x, head = blockIdx.x, which = blockIdx.y;
  const long long nwg = (long long)nlines * nh * nblk;
  const float* p = partial + (long long)which * nwg * 128;
  const int per_head = nlines * nblk;  // workgroups of this head
  const int chunk = (per_head + gridDim.z - 1) / gridDim.z;
  const int i0 = blockIdx.z * chunk, i1 = i0 + chunk < per_head ? i0 + chunk : per_head;
  // eight rows in flight per thread (two were 27 us for 25 MB of partial rows: one dependent load latency per pair)
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto at = [&](int i) -> float { return p[(((long long)(i / nblk) * nh + head) * nblk + i % nblk) * 128 + c]; };
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] += at(i + k);
  }
  for (; i < i1; i++) s[0] += at(i);
  if (i0 < i1) atomicAdd(dbias + (long long)which * nh * 128 + head * 128 + c, ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
}

__device__ __forceinline__ void attn_bwd_dq_body(unsigned char* smem, int lh, int qb, const bf16raw* qkv, const bf16raw* out,
                                                 const bf16raw* dout, const float* lse2, float* dvec, bf16raw* dqkv, float* dbias, int S,
                                                 int nh, float c, float scale) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nqb = S >> 7;
  const int line = lh / nh, head = lh % nh;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* base = qkv + (long long)line * S * ld + head * 128;
  const bf16raw* Kg = base + d;
  const bf16raw* Vg = base + 2 * d;
  const int q = qb * 128 + wave * 32 + r;

  // K / V are staged in 64-key HALF tiles, double-buffered (2 x (16 + 16) KiB = the LDS of one 128-key tile pair before):
  // the DMA of the next half runs under the current half's 48 MFMAs per wave.  With whole 128-key tiles and one buffer
  // the DMA was issued after the tile's last read and waited for at the top of the next one - its whole latency exposed
  // once per key tile, twice per workgroup at S = 256.
  attn_glds_half(Kg, ld, smem, wave, lane);
  attn_glds_half(Vg, ld, smem + AT_HALF_BYTES, wave, lane);

  // D[q] = sum_d dO[q][d] O[q][d] (row sums of the output gradient times the output), layout dvec[(line*S + q)*nh + head]:
  // either already there (out == nullptr: written by the epilogue of the product that produced dO, PERO_GEMM_ROWDOT) or
  // computed here from the O rows and stored for the dK / dV kernel.
  bf8v qf[8], gf[8];
  float dsum = 0.f;
  const long long dix = ((long long)line * S + q) * nh + head;
  {
    const bf16raw* qrow = base + (long long)q * ld + 8 * h5;
    const bf16raw* grow = dout + ((long long)line * S + q) * d + head * 128 + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      qf[ks] = *(const bf8v*)(qrow + 16 * ks);
      gf[ks] = *(const bf8v*)(grow + 16 * ks);
    }
    if (out) {
      const bf16raw* orow = out + ((long long)line * S + q) * d + head * 128 + 8 * h5;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const bf8v of = *(const bf8v*)(orow + 16 * ks);
#pragma unroll
        for (int e = 0; e < 8; e++) dsum += (float)gf[ks][e] * (float)of[e];
      }
      dsum += __shfl_xor(dsum, 32, 64);
      if (h5 == 0) dvec[dix] = dsum;
    } else {
      dsum = dvec[dix];
    }
  }
  const float lq = lse2[(long long)lh * S + q];

  f16v dq[4];
#pragma unroll
  for (int t = 0; t < 4; t++) dq[t] = (f16v){0};
  const int nhalf = S >> 6;
  for (int hk = 0; hk < nhalf; hk++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // half hk landed; every wave is done with the other buffer
    const unsigned char* kimg = smem + (hk & 1) * 2 * AT_HALF_BYTES;
    const unsigned char* vimg = kimg + AT_HALF_BYTES;
    if (hk + 1 < nhalf && !(AT_ABL & 1)) {
      unsigned char* nb = smem + ((hk + 1) & 1) * 2 * AT_HALF_BYTES;
      attn_glds_half(Kg + (long long)(hk + 1) * 64 * ld, ld, nb, wave, lane);
      attn_glds_half(Vg + (long long)(hk + 1) * 64 * ld, ld, nb + AT_HALF_BYTES, wave, lane);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {  // 32-key sub-tile
      f16v s = {0}, dp = {0};
      AT_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < ((AT_ABL & 16) ? 1 : 8); ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(kimg, t * 32 + r, ks, h5), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(vimg, t * 32 + r, ks, h5), gf[ks], dp, 0, 0, 0);
      }
      AT_PRIO(0);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float p = (AT_ABL & 2) ? fmaf(s[e], c, -lq) : __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq));
        s[e] = p * (dp[e] - dsum) * scale;  // dS^T
      }
      AT_PRIO(1);
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        const bf8v dsf = pack8(s, sub);
#pragma unroll
        for (int dt = 0; dt < ((AT_ABL & 4) ? 1 : 4); dt++)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag(kimg, t * 32 + sub * 16, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
      AT_PRIO(0);
    }
  }
  // dbias: partial-sum workspace [3][workgroups][128] (q, k, v); this kernel fills plane 0
  attn_store_tile(dq, smem, dqkv + ((long long)line * S + qb * 128) * ld + head * 128, ld,
                  dbias ? dbias + ((long long)lh * nqb + qb) * 128 : nullptr, tid, wave, r, h5);
}

// ---- dQ body, pipelined reads.  Per 64-key half (one LDS stage pair): 48 fragments = 2 sub-tiles x (16 row fragments K0 V0 K1 V1 ... for
// S^T / dP^T, then 8 transposed K fragments for dQ^T).  Fragment j is issued when fragment j - 7 has been consumed (pool of 8 registers
// sets), so at most 7 fragments (<= 14 LDS instructions) are in flight.
#define DQ_DEPTH 7
__device__ __forceinline__ constexpr int dq_ninstr(int j) { return (j % 24) < 16 ? 1 : 2; }
__device__ __forceinline__ constexpr int dq_after(int j) {   // LDS instructions issued after fragment j's when it is consumed
  int n = 0;
  for (int k = j + 1; k <= j + DQ_DEPTH && k < 48; k++) n += dq_ninstr(k);
  return n;
}
__device__ __forceinline__ void attn_bwd_dq_body_p(unsigned char* smem, int lh, int qb, const bf16raw* qkv, const bf16raw* out,
                                                   const bf16raw* dout, const float* lse2, float* dvec, bf16raw* dqkv, float* dbias, int S,
                                                   int nh, float c, float scale) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nqb = S >> 7;
  const int line = lh / nh, head = lh % nh;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* base = qkv + (long long)line * S * ld + head * 128;
  const bf16raw* Kg = base + d;
  const bf16raw* Vg = base + 2 * d;
  const int q = qb * 128 + wave * 32 + r;

  attn_glds_half(Kg, ld, smem, wave, lane);
  attn_glds_half(Vg, ld, smem + AT_HALF_BYTES, wave, lane);

  bf8v qf[8], gf[8];
  float dsum = 0.f;
  const long long dix = ((long long)line * S + q) * nh + head;
  {
    const bf16raw* qrow = base + (long long)q * ld + 8 * h5;
    const bf16raw* grow = dout + ((long long)line * S + q) * d + head * 128 + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      qf[ks] = *(const bf8v*)(qrow + 16 * ks);
      gf[ks] = *(const bf8v*)(grow + 16 * ks);
    }
    if (out) {
      const bf16raw* orow = out + ((long long)line * S + q) * d + head * 128 + 8 * h5;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const bf8v of = *(const bf8v*)(orow + 16 * ks);
#pragma unroll
        for (int e = 0; e < 8; e++) dsum += (float)gf[ks][e] * (float)of[e];
      }
      dsum += __shfl_xor(dsum, 32, 64);
      if (h5 == 0) dvec[dix] = dsum;
    } else {
      dsum = dvec[dix];
    }
  }
  const float lq = lse2[(long long)lh * S + q];

  // fragment addresses inside a stage (byte offsets from the stage's K image): row fragments per ks, transposed fragments per dt
  const unsigned s0 = at_lds_addr(smem);
  unsigned ra[8], ta[4], tb[4];
  {
    const int f = img_f(r);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) ra[ks] = (unsigned)(r * 256 + (((2 * ks + h5) ^ f) << 4));
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const int row = 4 * h5 + (i >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
      const int ch = 4 * dt + 2 * g1 + ((i & 3) >> 1);
      ta[dt] = (unsigned)(row * 256 + ((ch ^ img_f(row)) << 4) + 8 * (i & 1));
      tb[dt] = (unsigned)((row + 8) * 256 + ((ch ^ img_f(row + 8)) << 4) + 8 * (i & 1));
    }
  }

  f16v dq[4];
#pragma unroll
  for (int t = 0; t < 4; t++) dq[t] = (f16v){0};
  const int nhalf = S >> 6;
  for (int hk = 0; hk < nhalf; hk++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // half hk landed; every wave is done with the other buffer
    const unsigned stage = s0 + (hk & 1) * 2 * AT_HALF_BYTES;
    if (hk + 1 < nhalf && !(AT_ABL & 1)) {
      unsigned char* nb = smem + ((hk + 1) & 1) * 2 * AT_HALF_BYTES;
      attn_glds_half(Kg + (long long)(hk + 1) * 64 * ld, ld, nb, wave, lane);
      attn_glds_half(Vg + (long long)(hk + 1) * 64 * ld, ld, nb + AT_HALF_BYTES, wave, lane);
    }
    bf8v fr[8];
    f16v s, dp;
    bf8v dsf[2];
    auto issue = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int t = j / 24, qd = j % 24;
      if constexpr (qd < 16) {
        constexpr int ks = qd >> 1, isv = qd & 1;
        at_rd128<isv * AT_HALF_BYTES + t * 8192>(fr[j & 7], ra[ks], stage);
      } else {
        constexpr int sub = (qd - 16) >> 2, dt = (qd - 16) & 3;
        at_rdtr<t * 8192 + sub * 4096, t * 8192 + sub * 4096>(fr[j & 7], ta[dt], tb[dt], stage);
      }
    };
    at_static_for<0, DQ_DEPTH>(issue);
    AT_PRIO(1);
    at_static_for<0, 48>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int qd = j % 24;
      if constexpr (j + DQ_DEPTH < 48) issue(std::integral_constant<int, j + DQ_DEPTH>{});
      if constexpr (qd == 16) {
        // ---- dS^T = P^T (dP^T - D) scale for the sub-tile whose scores are complete; the transposed fragments are in flight
        AT_PRIO(0);
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq));
          s[e] = p * (dp[e] - dsum) * scale;
        }
        dsf[0] = pack8(s, 0);
        dsf[1] = pack8(s, 1);
        AT_PRIO(1);
      }
      at_wait_lgkm<dq_after(j)>(fr[j & 7]);
      if constexpr (qd < 16) {
        constexpr int ks = qd >> 1;
        if constexpr (qd == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 7], qf[0], (f16v){0}, 0, 0, 0);
        else if constexpr (qd == 1) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 7], gf[0], (f16v){0}, 0, 0, 0);
        else if constexpr ((qd & 1) == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 7], qf[ks], s, 0, 0, 0);
        else dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 7], gf[ks], dp, 0, 0, 0);
      } else {
        constexpr int sub = (qd - 16) >> 2, dt = (qd - 16) & 3;
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j & 7], dsf[sub], dq[dt], 0, 0, 0);
      }
    });
    AT_PRIO(0);
  }
  attn_store_tile(dq, smem, dqkv + ((long long)line * S + qb * 128) * ld + head * 128, ld,
                  dbias ? dbias + ((long long)lh * nqb + qb) * 128 : nullptr, tid, wave, r, h5);
}
template <bool PIPE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_k(const bf16raw* qkv, const bf16raw* out, const bf16raw* dout, const float* lse2,
                                                        float* dvec, bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nqb = S >> 7;
  int lh, qb;
  attn_block_map(blockIdx.x, nqb, gridDim.x / nqb, lh, qb);
  if (PIPE) attn_bwd_dq_body_p(smem, lh, qb, qkv, out, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
  else attn_bwd_dq_body(smem, lh, qb, qkv, out, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
}

// dK and dV in ONE pass (4 products: S, dP, dV^T += dO^T P, dK^T += Q^T dS; key on the lane).  The two-launch form read
// Q, dO and K twice and computed S twice (605 MB and 5 products per layer at S = 256, d = 512); both launches were HBM-bound
// to about half (row / tile reads of 256 KiB per workgroup for ~2.6 us of MFMA work).  What made the single pass spill before
// was register-resident V next to register-resident K and two accumulator sets; here the workgroup's 128 x 128 V tile
// lives in LDS (read as the B operand of dP) and Q / dO arrive in 32-query stages (8 + 8 KiB, double-buffered), so the
// footprint stays at 64.5 KiB = two workgroups per CU.
__device__ __forceinline__ void attn_bwd_dkv2_body(unsigned char* smem, int lh, int kb, long long nwg, const bf16raw* qkv, const bf16raw* dout,
                                                   const float* lse2, const float* dvec, bf16raw* dqkv, float* dbias, int S, int nh, float c,
                                                   float scale) {
  unsigned char* vimg = smem + 4 * AT_SUB_BYTES;
  float* lds_ld = (float*)(smem + 4 * AT_SUB_BYTES + AT_TILE_BYTES);  // [2 buffers][32 lse2 | 32 D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nkb = S >> 7;
  const int line = lh / nh, head = lh % nh;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* base = qkv + (long long)line * S * ld + head * 128;
  const bf16raw* Gg = dout + (long long)line * S * d + head * 128;
  const int key = kb * 128 + wave * 32 + r;
  // row statistics of a 32-query stage: threads 0-31 load lse2[lh][q], threads 32-63 load D[(line*S + q)*nh + head]
  const float* stat = tid < 32 ? lse2 + (long long)lh * S + tid : dvec + ((long long)line * S + (tid & 31)) * nh + head;
  const long long stat_step = tid < 32 ? 32 : 32LL * nh;

  if (tid < 64) lds_ld[tid] = stat[0];
  attn_glds_img(base + 2 * d + (long long)kb * 128 * ld, ld, vimg, wave, lane);  // this workgroup's V tile, resident
  attn_glds_sub(base, ld, smem, wave, lane);
  attn_glds_sub(Gg, d, smem + AT_SUB_BYTES, wave, lane);

  bf8v kf[8];
  {
    const bf16raw* krow = base + d + (long long)key * ld + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) kf[ks] = *(const bf8v*)(krow + 16 * ks);
  }
  f16v dv[4], dk[4];  // dV^T, dK^T: 32 d x 32 keys per tile, key on the lane
#pragma unroll
  for (int t = 0; t < 4; t++) { dv[t] = (f16v){0}; dk[t] = (f16v){0}; }
  const int nsub = S >> 5;
  for (int sq = 0; sq < nsub; sq++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // stage sq (Q / dO rows + statistics) landed; every wave is done with the other buffers
    const unsigned char* qimg = smem + (sq & 1) * 2 * AT_SUB_BYTES;
    const unsigned char* gimg = qimg + AT_SUB_BYTES;
    const float* lds_l = lds_ld + (sq & 1) * 64;
    const float* lds_d = lds_l + 32;
    float nstat = 0.f;
    if (sq + 1 < nsub && !(AT_ABL & 1)) {
      if (tid < 64) nstat = stat[(sq + 1) * stat_step];  // before the DMA: vmcnt is in-order
      unsigned char* nb = smem + ((sq + 1) & 1) * 2 * AT_SUB_BYTES;
      attn_glds_sub(base + (long long)(sq + 1) * 32 * ld, ld, nb, wave, lane);
      attn_glds_sub(Gg + (long long)(sq + 1) * 32 * d, d, nb + AT_SUB_BYTES, wave, lane);
    }
    // rows q = (e&3) + 8(e>>2) + 4*h5 of the 32-query stage on the registers, key on the lane
    f16v s = {0}, dp = {0};
    AT_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < ((AT_ABL & 16) ? 1 : 8); ks++) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(qimg, r, ks, h5), kf[ks], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(gimg, r, ks, h5), img_row_frag(vimg, wave * 32 + r, ks, h5), dp, 0, 0, 0);
    }
    AT_PRIO(0);
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const f4v l4 = *(const f4v*)(lds_l + 8 * g4 + 4 * h5);
      const f4v d4 = *(const f4v*)(lds_d + 8 * g4 + 4 * h5);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float p = (AT_ABL & 2) ? fmaf(s[4 * g4 + e], c, -l4[e]) : __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + e], c, -l4[e]));
        s[4 * g4 + e] = p;                                         // P
        dp[4 * g4 + e] = p * (dp[4 * g4 + e] - d4[e]) * scale;     // dS
      }
    }
    AT_PRIO(1);
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const bf8v pf = pack8(s, sub), dsf = pack8(dp, sub);
#pragma unroll
      for (int dt = 0; dt < ((AT_ABL & 4) ? 1 : 4); dt++) {
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag(gimg, sub * 16, dt, lane), pf, dv[dt], 0, 0, 0);   // dV^T += dO^T P
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag(qimg, sub * 16, dt, lane), dsf, dk[dt], 0, 0, 0);  // dK^T += Q^T dS
      }
    }
    AT_PRIO(0);
    if (sq + 1 < nsub && tid < 64) lds_ld[((sq + 1) & 1) * 64 + tid] = nstat;  // visible after the next barrier
  }
  bf16raw* tile_o = dqkv + ((long long)line * S + kb * 128) * ld + d + head * 128;  // dK tile; dV tile = + d columns
  // planes 1 (dK) and 2 (dV) of the partial-sum workspace, nwg = (line, head) x key blocks entries each
  attn_store_tile(dk, smem, tile_o, ld, dbias ? dbias + (nwg + (long long)lh * nkb + kb) * 128 : nullptr, tid, wave, r, h5);
  attn_store_tile(dv, smem, tile_o + d, ld, dbias ? dbias + (2 * nwg + (long long)lh * nkb + kb) * 128 : nullptr, tid, wave, r, h5);
}

// ---- dK / dV body, pipelined reads.  Per 32-query stage: 24 row fragments (per ks: Q, dO, V) for S / dP, then the stage's row statistics
// (compiler-visible LDS reads, issued while nothing else is in flight) and the exponentials, with the first transposed fragments of the
// gradient products issued between the four groups of the arithmetic, then 16 transposed fragments (per (sub, dt): dO^T, Q^T).
#ifndef DKV_POOL
#define DKV_POOL 6
#endif
#define DKV_TD (DKV_POOL - 1)        // transposed fragments in flight: 5 .. 7
#define DKV_RDEPTH (DKV_POOL - 2)   // row fragments in flight (a dP product holds two pool entries: dO and V); transposed fragments: 7
                                    // in flight (two LDS instructions each: 14 of the 15 the counter can hold)
__device__ __forceinline__ void attn_bwd_dkv2_body_p(unsigned char* smem, int lh, int kb, long long nwg, const bf16raw* qkv, const bf16raw* dout,
                                                     const float* lse2, const float* dvec, bf16raw* dqkv, float* dbias, int S, int nh, float c,
                                                     float scale) {
  unsigned char* vimg = smem + 4 * AT_SUB_BYTES;
  float* lds_ld = (float*)(smem + 4 * AT_SUB_BYTES + AT_TILE_BYTES);  // [2 buffers][32 lse2 | 32 D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nkb = S >> 7;
  const int line = lh / nh, head = lh % nh;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* base = qkv + (long long)line * S * ld + head * 128;
  const bf16raw* Gg = dout + (long long)line * S * d + head * 128;
  const int key = kb * 128 + wave * 32 + r;
  const float* stat = tid < 32 ? lse2 + (long long)lh * S + tid : dvec + ((long long)line * S + (tid & 31)) * nh + head;
  const long long stat_step = tid < 32 ? 32 : 32LL * nh;

  if (tid < 64) lds_ld[tid] = stat[0];
  attn_glds_img(base + 2 * d + (long long)kb * 128 * ld, ld, vimg, wave, lane);  // this workgroup's V tile, resident
  attn_glds_sub(base, ld, smem, wave, lane);
  attn_glds_sub(Gg, d, smem + AT_SUB_BYTES, wave, lane);

  bf8v kf[8];
  {
    const bf16raw* krow = base + d + (long long)key * ld + 8 * h5;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) kf[ks] = *(const bf8v*)(krow + 16 * ks);
  }
  // fragment addresses (byte offsets inside a stage's Q image; the dO image follows at + AT_SUB_BYTES)
  const unsigned s0 = at_lds_addr(smem);
  const unsigned vbase = s0 + 4 * AT_SUB_BYTES + __builtin_amdgcn_readfirstlane(wave) * 8192;    // this wave's 32 rows of the resident V image
  unsigned ra[8], ta[4], tb[4];
  {
    const int f = img_f(r);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) ra[ks] = (unsigned)(r * 256 + (((2 * ks + h5) ^ f) << 4));
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const int row = 4 * h5 + (i >> 2);
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
      const int ch = 4 * dt + 2 * g1 + ((i & 3) >> 1);
      ta[dt] = (unsigned)(row * 256 + ((ch ^ img_f(row)) << 4) + 8 * (i & 1));
      tb[dt] = (unsigned)((row + 8) * 256 + ((ch ^ img_f(row + 8)) << 4) + 8 * (i & 1));
    }
  }
  f16v dv[4], dk[4];
#pragma unroll
  for (int t = 0; t < 4; t++) { dv[t] = (f16v){0}; dk[t] = (f16v){0}; }
  const int nsub = S >> 5;
  for (int sq = 0; sq < nsub; sq++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // stage sq (Q / dO rows + statistics) landed; every wave is done with the other buffers
    const unsigned stage = s0 + (sq & 1) * 2 * AT_SUB_BYTES;
    const float* lds_l = lds_ld + (sq & 1) * 64;
    const float* lds_d = lds_l + 32;
    float nstat = 0.f;
    if (sq + 1 < nsub && !(AT_ABL & 1)) {
      if (tid < 64) nstat = stat[(sq + 1) * stat_step];  // before the DMA: vmcnt is in-order
      unsigned char* nb = smem + ((sq + 1) & 1) * 2 * AT_SUB_BYTES;
      attn_glds_sub(base + (long long)(sq + 1) * 32 * ld, ld, nb, wave, lane);
      attn_glds_sub(Gg + (long long)(sq + 1) * 32 * d, d, nb + AT_SUB_BYTES, wave, lane);
    }
    bf8v fr[DKV_POOL];
    f16v s, dp;
    // ---- S = Q K^T, dP = dO V^T: 24 row fragments
    auto issue_r = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int ks = j / 3, kind = j % 3;
      if constexpr (kind == 0) at_rd128<0>(fr[j % DKV_POOL], ra[ks], stage);
      else if constexpr (kind == 1) at_rd128<AT_SUB_BYTES>(fr[j % DKV_POOL], ra[ks], stage);
      else at_rd128<0>(fr[j % DKV_POOL], ra[ks], vbase);
    };
    at_static_for<0, DKV_RDEPTH>(issue_r);
    AT_PRIO(1);
    at_static_for<0, 24>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int ks = j / 3, kind = j % 3;
      if constexpr (j + DKV_RDEPTH < 24) issue_r(std::integral_constant<int, j + DKV_RDEPTH>{});
      constexpr int after = (24 - 1 - j) < DKV_RDEPTH ? (24 - 1 - j) : DKV_RDEPTH;
      if constexpr (kind == 0) {
        at_wait_lgkm<after>(fr[j % DKV_POOL]);
        if constexpr (ks == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % DKV_POOL], kf[0], (f16v){0}, 0, 0, 0);
        else s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % DKV_POOL], kf[ks], s, 0, 0, 0);
      } else if constexpr (kind == 2) {
        at_wait_lgkm<after>(fr[j % DKV_POOL]);   // LDS reads retire in order: the dO fragment (j - 1) has landed too
        asm volatile("" : "+v"(fr[(j - 1) % DKV_POOL]));
        if constexpr (ks == 0) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(j - 1) % DKV_POOL], fr[j % DKV_POOL], (f16v){0}, 0, 0, 0);
        else dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[(j - 1) % DKV_POOL], fr[j % DKV_POOL], dp, 0, 0, 0);
      }
    });
    AT_PRIO(0);
    // ---- P, dS (rows q = (e&3) + 8(e>>2) + 4*h5 of the stage on the registers); transposed fragments go out group by group
    auto issue_t = [&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int sub = m >> 3, dt = (m >> 1) & 3, kind = m & 1;   // kind 0: dO^T (-> dV), 1: Q^T (-> dK)
      constexpr int off = sub * 4096 + (kind == 0 ? AT_SUB_BYTES : 0);
      at_rdtr<off, off>(fr[m % DKV_POOL], ta[dt], tb[dt], stage);
    };
    // The stage's row statistics (lse2 and D of the 16 queries a lane holds: four groups of 4 + 4 floats) travel in the same counted stream
    // as the fragments - compiler-visible reads would be waited for with lgkmcnt(0), i.e. together with every transposed fragment
    // issued ahead of them.  Stream: S0 S1 | math 0 | S2 T0 T1 | math 1 | S3 T2 T3 | math 2 | T4 T5 | math 3 | T6   (Sg = 2 reads, Tm = 2)
    f4v l4[4], d4[4];
    const unsigned stb = s0 + 4 * AT_SUB_BYTES + AT_TILE_BYTES + (sq & 1) * 256, sto = 16 * h5;
    auto issue_s = [&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      at_rd128<32 * g>(l4[g], sto, stb);
      at_rd128<128 + 32 * g>(d4[g], sto, stb);
    };
    issue_s(std::integral_constant<int, 0>{});
    issue_s(std::integral_constant<int, 1>{});
    bf8v pf[2], dsf[2];
    at_static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g4 = decltype(gc)::value;
      // LDS instructions issued after this group's statistics: g0: S1 = 2; g1: S2 T0 T1 = 6; g2: T0 T1 S3 T2 T3 = 10; g3: T2 T3 T4 T5 = 8
      constexpr int after = g4 == 0 ? 2 : g4 == 1 ? 6 : g4 == 2 ? 10 : (DKV_TD >= 6 ? 8 : 4);
      at_wait_lgkm2<after>(l4[g4], d4[g4]);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + e], c, -l4[g4][e]));
        s[4 * g4 + e] = p;                                              // P
        dp[4 * g4 + e] = p * (dp[4 * g4 + e] - d4[g4][e]) * scale;     // dS
      }
      if constexpr (g4 == 1) { pf[0] = pack8(s, 0); dsf[0] = pack8(dp, 0); }
      if constexpr (g4 == 3) { pf[1] = pack8(s, 1); dsf[1] = pack8(dp, 1); }
      if constexpr (g4 + 2 < 4) issue_s(std::integral_constant<int, g4 + 2>{});
      if constexpr (2 * g4 < DKV_TD) issue_t(std::integral_constant<int, 2 * g4>{});
      if constexpr (2 * g4 + 1 < DKV_TD) issue_t(std::integral_constant<int, 2 * g4 + 1>{});
    });
    AT_PRIO(1);
    at_static_for<0, 16>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      constexpr int sub = m >> 3, dt = (m >> 1) & 3, kind = m & 1;
      if constexpr (m + DKV_TD < 16) issue_t(std::integral_constant<int, m + DKV_TD>{});
      constexpr int after = 2 * ((16 - 1 - m) < DKV_TD ? (16 - 1 - m) : DKV_TD);
      at_wait_lgkm<after>(fr[m % DKV_POOL]);
      if constexpr (kind == 0) dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % DKV_POOL], pf[sub], dv[dt], 0, 0, 0);    // dV^T += dO^T P
      else dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[m % DKV_POOL], dsf[sub], dk[dt], 0, 0, 0);                     // dK^T += Q^T dS
    });
    AT_PRIO(0);
    if (sq + 1 < nsub && tid < 64) lds_ld[((sq + 1) & 1) * 64 + tid] = nstat;  // visible after the next barrier
  }
  bf16raw* tile_o = dqkv + ((long long)line * S + kb * 128) * ld + d + head * 128;  // dK tile; dV tile = + d columns
  attn_store_tile(dk, smem, tile_o, ld, dbias ? dbias + (nwg + (long long)lh * nkb + kb) * 128 : nullptr, tid, wave, r, h5);
  attn_store_tile(dv, smem, tile_o + d, ld, dbias ? dbias + (2 * nwg + (long long)lh * nkb + kb) * 128 : nullptr, tid, wave, r, h5);
}
template <bool PIPE>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv2_k(const bf16raw* qkv, const bf16raw* dout, const float* lse2, const float* dvec,
                                                          bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkb = S >> 7;
  int lh, kb;
  attn_block_map(blockIdx.x, nkb, gridDim.x / nkb, lh, kb);
  if (PIPE) attn_bwd_dkv2_body_p(smem, lh, kb, gridDim.x, qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
  else attn_bwd_dkv2_body(smem, lh, kb, gridDim.x, qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
}
// Both backward kernels as ONE launch (D already computed: `out` is not read, so no workgroup depends on another): the 2 x (S / 128)
// workgroups of a (line, head) - its dQ blocks and its dK / dV blocks, which all read the same Q, K, V and dO rows - sit next to
// each other in one XCD's dispatch order, so the rows come from HBM once and the other readers find them in that XCD's L2
// (FETCH_SIZE of the backward at 256 lines: 534 MB as two launches, 308 MB paired, 267 MB = each row once; 789 -> 740 us at 1024 lines).
template <bool PIPE>
__global__ __launch_bounds__(256, 2) void attn_bwd_pair_k(const bf16raw* qkv, const bf16raw* dout, const float* lse2, float* dvec,
                                                          bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale, int order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = S >> 7;
  int lh, blk;
  attn_block_map(blockIdx.x, 2 * nb, gridDim.x / (2 * nb), lh, blk);
  {   // Dispatch order inside an XCD (pero_set_option("attn_order", n); default 32): chunks of 32 units whose 64 dK / dV blocks - one round of the XCD's 64 workgroup
      // places - go out ahead of their 64 dQ blocks, so that the CUs of an XCD run ONE kind of block at a time: medians of six launches (tools/attn_order_ab.py, 2048 lines)
      // 1 406 us side by side, 1 385 / 1 381 with chunks of 4 / 8, 1 414 with 16 (both kinds in one round again), 1 361 with 32, 1 388 with 64.  A chunk's rows are 8 MB:
      // the dQ blocks find half of them in the XCD's L2, the rest in the memory-side cache.
    const int nlh = gridDim.x / (2 * nb);
    if (order && (nlh & 7) == 0) {
      const int xcd = blockIdx.x & 7, u = blockIdx.x >> 3;
      if (order == 1) blk = (blk + nb) % (2 * nb);
      else {
        // order = 100 v + CH: chunks of CH units per XCD; v & 1: the dQ blocks of a chunk first (else the dK / dV blocks); v & 2: block-major inside a role (else unit-major)
        const int CH = order % 100, v = order / 100, per = CH * 2 * nb;
        if (CH > 0 && (nlh >> 3) % CH == 0) {
          const int cch = u / per, i = u % per, first = i / (CH * nb), j = i % (CH * nb);
          const int role_dkv = (v & 1) ? first : 1 - first;
          const int un = (v & 2) ? j % CH : j / nb, bl = (v & 2) ? j / CH : j % nb;
          lh = (cch * CH + un) * 8 + xcd;
          blk = (role_dkv ? nb : 0) + bl;
        }
      }
    }
  }
  if (blk < nb) {
    if (PIPE) attn_bwd_dq_body_p(smem, lh, blk, qkv, nullptr, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
    else attn_bwd_dq_body(smem, lh, blk, qkv, nullptr, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
  } else {
    if (PIPE) attn_bwd_dkv2_body_p(smem, lh, blk - nb, (long long)(gridDim.x >> 1), qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
    else attn_bwd_dkv2_body(smem, lh, blk - nb, (long long)(gridDim.x >> 1), qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
  }
}


// =================================================================================================
// Backward of ONE (line, head) per workgroup pass, S = 256: persistent, eight waves, one workgroup per CU (round 3).
//
// What the two-workgroups-per-CU kernels above are bound by is neither MFMA nor LDS nor HBM bandwidth: with every MFMA or every
// exponential compiled out they run at the same speed (tools/attn_ablate.py, realistic inputs: -1 %), while without the loop's LDS-DMA
// or without the output tiles they gain 11 % / 20 %.  They move 2.0 GB in ~740 us (2.7 TB/s) with at most one 32 KiB stage in flight per
// workgroup, and every workgroup pays its first loads and its last stores in full: memory-level parallelism is the bound.  Here a
// workgroup owns 128 KiB of LDS as a ring of four 32 KiB slots and streams a whole (line, head) through it with the loads THREE items
// ahead of the MFMAs, every vector-memory instruction issued by inline asm and waited for by a COUNTED vmcnt (the counter retires in
// order; the schedule below is static, so the counts are constants), and it loops over (line, head) units so that the next unit's
// first loads fly under the current unit's last stores:
//
//   phase Q (dQ^T of the 256 queries; wave w owns queries 32w..32w+31 on its lanes; Q / dO row fragments, lse and D in registers):
//       items H0..H3 = 64-key halves (K image 16 KiB + V image 16 KiB) in slots 0..3;   per half: the fragment sequence of attn_bwd_dq_body_p
//   phase K (dK^T / dV^T of the 256 keys; wave w owns keys 32w..32w+31; K row fragments in registers):
//       items V0, V1 = the V block (256 keys x 256 B) in slots 0, 1 (B operand of dP), items T0..T7 = 32-query stages (Q image 8 KiB +
//       dO image 8 KiB + 256 B of row statistics) in the four 16 KiB sub-slots of slots 2, 3;   per stage: the sequence of attn_bwd_dkv2_body_p
//
//   per-thread vector-memory program order of a unit in steady state ([n] = instructions; `newer` = issued after the awaited item):
//       (end of the previous unit)  H0 H1 H2 H3 [4 x 4]   dK rows [8]   F: Q / dO fragments, lse, D [18] (their registers are free once dK has left)   dV rows [8]
//       half 0: wait F - and with it the older H0..H3, which have had the whole epilogue to land - (newer 8) | barrier | previous unit's bias
//               partials [1] | MFMAs          halves 1..3: MFMAs (everything is resident: no barrier, no wave waits for another)
//       (the first unit of a workgroup issues F, then H0..H3, and waits for all of it)
//       barrier | issue V0 V1 [4 + 4], K fragments [8], T0 T1 T2 T3 [4 x 3] | epilogue Q: dQ rows [8]
//       stage 0: wait V0 V1, the K fragments and T0 (newer: T1 T2 T3 9 + 8 = 17) | barrier | MFMAs
//       stage j = 1..3: wait Tj (14) | barrier | issue T(j+3) [3] | MFMAs     stage 4: wait (6) | barrier | issue T7 | MFMAs     stages 5, 6, 7: wait (6, 3, 0)
//       barrier | issue the next unit's H0..H3 | epilogue K: dK rows, the next unit's F, dV rows
//   A slot is refilled only behind the barrier that follows its last reader; LDS reads are asm too (the compiler would put a vmcnt(0)
//   in front of any LDS read it can see behind an LDS-DMA).  The same MFMAs in the same order as the kernels above: dqkv is bit-identical.
// Output tiles leave through a wave-private 2 KiB staging block, 32 columns at a time (64-byte row segments on four adjacent lanes); the
// column sums of the stored values (in_proj's bias gradient) are reduced over the eight waves through LDS: one partial row per unit.
// =================================================================================================
#define LH_SLOT 32768
#define LH_RING (4 * LH_SLOT)
#define LH_STATS LH_RING                    // 4 sub-slots x 256 B: [32 lse2 | 32 D] of a stage
#define LH_STG (LH_RING + 1024)             // 8 waves x 2 KiB
#define LH_PART (LH_STG + 8 * 2048)         // 8 waves x 3 matrices x 128 floats
#define LH_LDS_BYTES (LH_PART + 8 * 1536)   // 160 768 B of the CU's 163 840

// LDS-DMA, 16 (4) bytes per lane: global address = sbase + voff, LDS address = dst + 16 (4) * lane
// Every asm vector-memory instruction below that takes an SGPR base opens with wait states: a v_readlane_b32 / v_readfirstlane_b32 that has
// just (re)written the SGPR - a restored spill - needs five of them before a vector-memory instruction reads it, and the compiler pads that
// hazard for its own instructions only (gemm_e.hip's E_BSTORE16 took stale row offsets that way in round 3; tools/check_async_loads.py
// checks the ISA of these kernels too: tests/test_cabi_and_host.py).  With the s_mov the LDS-DMA forms have s_nop 3 + 1.
__device__ __forceinline__ void lh_dma16(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(sbase), "v"(voff), "s"(dst) : "memory", "m0");
}
__device__ __forceinline__ void lh_dma4(const float* addr, unsigned dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(addr), "s"(dst) : "memory", "m0");
}
template <int IMM, typename T>
__device__ __forceinline__ void lh_gload16(T& d, const void* sbase, unsigned voff) {
  static_assert(sizeof(T) == 16 && IMM >= 0 && IMM < 4096, "global_load_dwordx4");
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "i"(IMM) : "memory");
}
__device__ __forceinline__ void lh_gload4(float& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// (the trailing wait states: the data registers are rewritten right behind the store - see E_BSTORE16 in gemm_e.hip)
template <int IMM, typename T>
__device__ __forceinline__ void lh_gstore16(const T& v, void* sbase, unsigned voff) {
  static_assert(sizeof(T) == 16 && IMM >= 0 && IMM < 4096, "global_store_dwordx4");
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 2" :: "v"(voff), "v"(v), "s"(sbase), "i"(IMM) : "memory");
}
__device__ __forceinline__ void lh_gstore4(float v, void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2\n\ts_nop 2" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void lh_wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
}
template <int N_STEADY, int N_FIRST>
__device__ __forceinline__ void lh_wait_vm2(bool first) {
  if (first) lh_wait_vm<N_FIRST>(); else lh_wait_vm<N_STEADY>();
}
__device__ __forceinline__ void lh_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <typename T>
__device__ __forceinline__ void lh_ds_write8(unsigned addr, const T& v) {
  static_assert(sizeof(T) == 8, "ds_write_b64");
  asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
template <typename T>
__device__ __forceinline__ void lh_ds_write16(unsigned addr, const T& v) {
  static_assert(sizeof(T) == 16, "ds_write_b128");
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
template <typename T>
__device__ __forceinline__ void lh_ds_read16(T& d, unsigned addr) {
  static_assert(sizeof(T) == 16, "ds_read_b128");
  asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lh_ds_write4(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "i"(OFF) : "memory");
}
__device__ __forceinline__ void lh_ds_read4(float& d, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void lh_wait_lgkm_plain() {
  asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory");
}

// One 128-column output matrix of a wave (acc[dt][e]: row = this lane's query / key r, columns dt*32 + 8*(e>>2) + 4*h5 + (e&3)) -> global
// rows (pitch `pitch_b` bytes from `obase`) through the wave's private 2 KiB staging block, and its column sums (of the stored bf16
// values) -> the wave's partial row in LDS.  8 stores per thread.
//   * a round = 16 rows x 64 columns, i.e. 128-BYTE row segments on eight adjacent lanes: 64-byte segments (32 rows x 32 columns per round)
//     write at 3.4 TB/s, 128- and 256-byte ones at 5.5 (tools/probe_tilebw.hip) - the first version of this kernel spent 40 % of a unit in
//     its epilogues.  Round order (columns 0-63: rows 0-15, rows 16-31; columns 64-127: ...); the 32 lanes that own a round's rows write.
//   * the rounds are software-pipelined: a wave's LDS operations execute in order, so round r + 1 is written into the SAME block right
//     behind round r's reads without waiting for their data (counted lgkmcnt: S0 S1 | wait S0 | finish 0 | S2 | wait S1 | ...);
//   * the column sums are MFMAs: ones (32 x 16) times the staged 16 x 32 tile read back TRANSPOSED (ds_read_b64_tr_b16: the row index
//     becomes the MFMA's k), accumulated over the two row halves - every lane n then holds the sum of column n, exact in f32.  As
//     cross-lane sums they cost 384 ds_bpermute per wave and unit through the CU's one LDS crossbar (10 us of a 52 us unit,
//     tools/attn_lh_stamps.py), as DPP / v_permlane*_swap arithmetic ~450 vector instructions per matrix (8 us).
__device__ __forceinline__ void lh_store_matrix(const f16v (&acc)[4], unsigned stg, unsigned part, void* obase, unsigned pitch_b, int lane) {
  // (every address below derives from this opaque copy of the lane index: otherwise the compiler computes them once in front of the unit
  //  loop, finds no registers for them and reloads them from scratch inside the epilogue - behind an `s_waitcnt vmcnt(0)` that also waits
  //  for every LDS-DMA in flight)
  asm volatile("" : "+v"(lane));
  const int r = lane & 31, h5 = lane >> 5, rl = r & 15;
  // staging image [16 rows][128 B], 16-byte chunk index XORed with (row & 7)
  const unsigned wa = stg + rl * 128 + 8 * h5;                        // + (((4 * (dt & 1) + g4) ^ (rl & 7)) << 4)
  const int rrow = lane >> 3, rc = lane & 7;                          // read-back: rows rrow, rrow + 8; chunk rc (8 lanes = 128 contiguous bytes)
  const unsigned ra0 = stg + rrow * 128 + ((rc ^ (rrow & 7)) << 4);
  const unsigned ra1 = ra0 + 8 * 128;                                 // (row + 8: same row & 7)
  const unsigned go = (unsigned)rrow * pitch_b + rc * 16;
  // transposed read (B operand, k = row): lane (i = lane & 15, g1, h5) supplies the 8 bytes of row 4 h5 + (i >> 2) (+ 8 for the second
  // read), columns 32 cb + 16 g1 + 4 (i & 3) .. + 3; the hardware hands lane i column 32 cb + 16 g1 + i
  const int ti = lane & 15, tg = (lane >> 4) & 1;
  const int trow = 4 * h5 + (ti >> 2), tch = 2 * tg + ((ti & 3) >> 1);
  const unsigned tq0 = stg + trow * 128 + ((tch ^ (trow & 7)) << 4) + 8 * (ti & 1);          // column block 0, rows trow / trow + 8
  const unsigned tq1 = stg + trow * 128 + (((4 + tch) ^ (trow & 7)) << 4) + 8 * (ti & 1);    // column block 1
  const __bf16 one = (__bf16)1.0f;
  const bf8v ones = {one, one, one, one, one, one, one, one};
  at_u4v v0[4], v1[4];
  bf8v t0[4], t1[4];
  float sums[4];                                                      // column sums of columns 32 b + (lane & 31), b = 0..3
  f16v cs0, cs1;
  auto stage_round = [&](auto rc_) __attribute__((always_inline)) {   // round 2 ch + rh: 8 writes (the 32 lanes of the rows) + 2 reads + 4 transposed reads
    constexpr int rd = decltype(rc_)::value;
    constexpr int ch = rd >> 1, rh = rd & 1;
    if ((r >> 4) == rh) {
#pragma unroll
      for (int dl = 0; dl < 2; dl++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          const at_u2v w = {pack2bf(acc[2 * ch + dl][4 * g4 + 0], acc[2 * ch + dl][4 * g4 + 1]),
                            pack2bf(acc[2 * ch + dl][4 * g4 + 2], acc[2 * ch + dl][4 * g4 + 3])};
          lh_ds_write8(wa + (((4 * dl + g4) ^ (rl & 7)) << 4), w);
        }
    }
    lh_ds_read16(v0[rd], ra0);
    lh_ds_read16(v1[rd], ra1);
    at_rdtr<0, 8 * 128>(t0[rd], tq0, tq0, 0u);
    at_rdtr<0, 8 * 128>(t1[rd], tq1, tq1, 0u);
  };
  auto finish_round = [&](auto rc_) __attribute__((always_inline)) {  // 2 global stores; the round's share of the column sums
    constexpr int rd = decltype(rc_)::value;
    constexpr int ch = rd >> 1, rh = rd & 1;
    void* ob = (unsigned char*)obase + (long long)(16 * rh) * pitch_b;
#ifndef LH_ABL_NOSTORE   // (timing-only ablation: tools/attn_lh_stamps.py)
    lh_gstore16<ch * 128>(v0[rd], ob, go);
    lh_gstore16<ch * 128>(v1[rd], (unsigned char*)ob + 8LL * pitch_b, go);
#else
    asm volatile("" :: "v"(v0[rd]), "v"(v1[rd]), "s"(ob), "v"(go));
#endif
    if constexpr (rh == 0) {
      cs0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, t0[rd], (f16v){0}, 0, 0, 0);
      cs1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, t1[rd], (f16v){0}, 0, 0, 0);
    } else {
      cs0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, t0[rd], cs0, 0, 0, 0);
      cs1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, t1[rd], cs1, 0, 0, 0);
      sums[2 * ch] = cs0[0] + 0.0f;        // (through the vector ALU: the compiler pads the MFMA -> read hazard for its own instructions)
      sums[2 * ch + 1] = cs1[0] + 0.0f;
    }
  };
  stage_round(std::integral_constant<int, 0>{});
  stage_round(std::integral_constant<int, 1>{});
  asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(v0[0]), "+v"(v1[0]), "+v"(t0[0]), "+v"(t1[0]) :: "memory");     // behind S0: S1 = 14
  finish_round(std::integral_constant<int, 0>{});
  stage_round(std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(v0[1]), "+v"(v1[1]), "+v"(t0[1]), "+v"(t1[1]) :: "memory");
  finish_round(std::integral_constant<int, 1>{});
  stage_round(std::integral_constant<int, 3>{});
  asm volatile("s_waitcnt lgkmcnt(14)" : "+v"(v0[2]), "+v"(v1[2]), "+v"(t0[2]), "+v"(t1[2]) :: "memory");
  finish_round(std::integral_constant<int, 2>{});
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0[3]), "+v"(v1[3]), "+v"(t0[3]), "+v"(t1[3]) :: "memory");
  finish_round(std::integral_constant<int, 3>{});
  const unsigned pa = part + r * 4;          // lanes n and n + 32 hold the same sums and write the same address
  lh_ds_write4<0>(pa, sums[0]);
  lh_ds_write4<128>(pa, sums[1]);
  lh_ds_write4<256>(pa, sums[2]);
  lh_ds_write4<384>(pa, sums[3]);
}

#ifndef LH_STAGGER
#define LH_STAGGER 0
#endif
#ifndef LH_STAGGER_K
#define LH_STAGGER_K 0
#endif
#ifndef LH_QPOOL
#define LH_QPOOL 4     // fragment register sets of phase Q (LH_QPOOL - 1 in flight): the kernel is bound by its memory pipeline, not by LDS latency
#endif
#ifndef LH_KPOOL
#define LH_KPOOL 4     // ... of phase K (row fragments LH_KPOOL - 2, transposed fragments LH_KPOOL - 1 in flight)
#endif
__device__ __forceinline__ constexpr int lh_q_after(int j) {   // LDS instructions issued after fragment j's when it is consumed (phase Q)
  int n = 0;
  for (int k = j + 1; k <= j + LH_QPOOL - 1 && k < 48; k++) n += dq_ninstr(k);
  return n;
}
// timing-only ablation builds of attn_bwd_lh_k (tools/attn_lh_stamps.py; results wrong by design): -DLH_ABL=mask: 1 no LDS-DMA, 2 no MFMAs,
// 4 no fragment loads from global memory; -DLH_ABL_NOSTORE: no output stores
#ifndef LH_ABL
#define LH_ABL 0
#endif
#if LH_ABL & 2
#define LH_MFMA(a_, b_, c_, x_, y_, z_) (c_)
#else
#define LH_MFMA(a_, b_, c_, x_, y_, z_) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, x_, y_, z_)
#endif
__global__ __launch_bounds__(512, 2) void attn_bwd_lh_k(const bf16raw* qkv, const bf16raw* dout, const float* lse2, const float* dvec, bf16raw* dqkv,
                                                        float* work, int nunits, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int S = 256;
  const int tid = threadIdx.x, lane = tid & 63, h5 = lane >> 5, r = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const unsigned pq = (unsigned)(ld * 2), pg = (unsigned)(d * 2);    // row pitches in bytes: qkv / dqkv rows, dO rows
  const unsigned s0 = at_lds_addr(smem);

  // ---- per-thread constants
  // LDS-DMA piece of a tile image: row 4 * wave + (lane >> 4) (+ 32 per further piece), 16-byte chunk (lane & 15) ^ img_f(row)
  const int drow = 4 * wave + (lane >> 4);
  const unsigned dq_off = (unsigned)drow * pq + (((lane & 15) ^ img_f(drow)) << 4);   // in q / k / v rows
  const unsigned dg_off = (unsigned)drow * pg + (((lane & 15) ^ img_f(drow)) << 4);   // in dO rows
  const unsigned ddst = wave * 1024;                                                   // piece `wave` of an image (+ 8192 per further piece)
  // row fragments from global memory: this lane's row (query / key 32 * wave + r), 16 bytes at 32 * ks + 16 * h5
  const unsigned fq_off = (unsigned)(32 * wave + r) * pq + 16 * h5;
  const unsigned fg_off = (unsigned)(32 * wave + r) * pg + 16 * h5;
  // fragment read offsets inside an LDS stage (as in the _p bodies)
  // (one register per kind: the offset of k-step ks is ra0 ^ (32 * ks), that of head-dim tile dt is ta0 ^ (64 * dt) - see at_rd128x)
  unsigned ra0, ta0, tb0;
  {
    ra0 = (unsigned)(r * 256 + ((h5 ^ img_f(r)) << 4));
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const int row = 4 * h5 + (i >> 2);
    const int ch = 2 * g1 + ((i & 3) >> 1);
    ta0 = (unsigned)(row * 256 + ((ch ^ img_f(row)) << 4) + 8 * (i & 1));
    tb0 = (unsigned)((row + 8) * 256 + ((ch ^ img_f(row + 8)) << 4) + 8 * (i & 1));
  }
  const unsigned stg = s0 + LH_STG + wave * 2048, part = s0 + LH_PART + wave * 1536;

  // ---- unit-dependent bases (uniform)
  const bf16raw *uq, *ug;          // Q rows of the unit's (line, head); dO rows
  const float *ul, *ud;            // lse2 row of the unit; D[(line * S + 0) * nh + head]
  auto set_unit = [&](int u) {
    const int line = u / nh, head = u % nh;
    uq = qkv + (long long)line * S * ld + head * 128;
    ug = dout + (long long)line * S * d + head * 128;
    ul = lse2 + (long long)u * S;
    ud = dvec + (long long)line * S * nh + head;
  };
  bf8v qf[8], gf[8], kf[8];
  float lq, dsum;
  // F: this lane's Q / dO row fragments, lse and D of unit u  [18]
  auto issue_F = [&](const bf16raw* q_, const bf16raw* g_, const float* l_, const float* d_) {
    if (LH_ABL & 4) { asm volatile("" : "=v"(qf[0]), "=v"(qf[1]), "=v"(qf[2]), "=v"(qf[3]), "=v"(qf[4]), "=v"(qf[5]), "=v"(qf[6]), "=v"(qf[7]));
                      asm volatile("" : "=v"(gf[0]), "=v"(gf[1]), "=v"(gf[2]), "=v"(gf[3]), "=v"(gf[4]), "=v"(gf[5]), "=v"(gf[6]), "=v"(gf[7]), "=v"(lq), "=v"(dsum)); return; }
    at_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { constexpr int ks = decltype(kc)::value; lh_gload16<32 * ks>(qf[ks], q_, fq_off); });
    at_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { constexpr int ks = decltype(kc)::value; lh_gload16<32 * ks>(gf[ks], g_, fg_off); });
    lh_gload4(lq, l_, (unsigned)(32 * wave + r) * 4);
    lh_gload4(dsum, d_, (unsigned)(32 * wave + r) * (unsigned)nh * 4);
  };
  // H_h: K rows 64h..64h+63 -> slot h + 0, V rows -> slot h + 16384  [4]
  auto issue_H = [&](const bf16raw* q_, int h) {
    if (LH_ABL & 1) return;
    const unsigned char* kb = (const unsigned char*)(q_ + d) + (long long)h * 64 * pq;
    const unsigned char* vb = (const unsigned char*)(q_ + 2 * d) + (long long)h * 64 * pq;
    const unsigned dst = s0 + h * LH_SLOT + ddst;
    lh_dma16(kb, dq_off, dst);
    lh_dma16(kb + 32LL * pq, dq_off, dst + 8192);
    lh_dma16(vb, dq_off, dst + 16384);
    lh_dma16(vb + 32LL * pq, dq_off, dst + 16384 + 8192);
  };
  // V_i: V rows 128i..128i+127 -> slot i  [4]
  auto issue_V = [&](int i) {
    if (LH_ABL & 1) return;
    const unsigned char* vb = (const unsigned char*)(uq + 2 * d) + (long long)i * 128 * pq;
    const unsigned dst = s0 + i * LH_SLOT + ddst;
#pragma unroll
    for (int k = 0; k < 4; k++) lh_dma16(vb + (long long)k * 32 * pq, dq_off, dst + k * 8192);
  };
  // T_j: Q rows 32j..32j+31 -> sub-slot j & 3, dO rows -> + 8192, statistics -> LH_STATS + 256 (j & 3)  [3]
  const float* stat_lane = nullptr;   // set per unit: lanes 0-31 -> lse2[q], lanes 32-63 -> D[q]
  auto issue_T = [&](int j) {
    if (LH_ABL & 1) return;
    const long long stat_step = lane < 32 ? 32 : 32LL * nh;
    const unsigned sub = s0 + 2 * LH_SLOT + (j & 3) * 16384;
    lh_dma16((const unsigned char*)uq + (long long)j * 32 * pq, dq_off, sub + ddst);
    lh_dma16((const unsigned char*)ug + (long long)j * 32 * pg, dg_off, sub + 8192 + ddst);
    lh_dma4(stat_lane + j * stat_step, s0 + LH_STATS + (j & 3) * 256);
  };
  auto issue_KF = [&]() {   // K row fragments of this lane's key  [8]
    if (LH_ABL & 4) { asm volatile("" : "=v"(kf[0]), "=v"(kf[1]), "=v"(kf[2]), "=v"(kf[3]), "=v"(kf[4]), "=v"(kf[5]), "=v"(kf[6]), "=v"(kf[7])); return; }
    at_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { constexpr int ks = decltype(kc)::value; lh_gload16<32 * ks>(kf[ks], uq + d, fq_off); });
  };

  // Diagnostic build (-DLH_STAMP, tools/attn_lh_stamps.py): s_memtime of wave 0 at up to 64 points of the workgroup's THIRD unit, through LDS
  // (a global store would enter the counted vmcnt stream) -> the upper half of `work` as u64 [workgroup][64] at the end of the kernel
#ifdef LH_STAMP
  int stamp_unit = 0;
#define LH_ST(n_)                                                                                                   \
  if (stamp_unit == 2 && tid == 0) {                                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                     \
    lh_ds_write8(s0 + LH_LDS_BYTES + 8 * (n_), __builtin_bit_cast(at_u2v, t_));                                     \
  }
#else
#define LH_ST(n_)
#endif
  // bias-gradient partials of a finished unit: the eight waves' rows summed in wave order -> work[which][unit][128]; called behind a barrier
  // that every wave passes after its epilogue K.  One store per thread (threads 384..511 repeat the first 128: one count for all waves).
  auto reduce_partials = [&](int un) {
    int tq = tid;
    asm volatile("" : "+v"(tq));                   // (opaque: see lh_store_matrix)
    const int e = tq < 384 ? tq : tq - 384;
    float v[8];
#pragma unroll
    for (int w = 0; w < 8; w++) lh_ds_read4(v[w], s0 + LH_PART + w * 1536 + e * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
    const float acc = ((((((v[0] + v[1]) + v[2]) + v[3]) + v[4]) + v[5]) + v[6]) + v[7];
    const int which = e >> 7, col = e & 127;
    lh_gstore4(acc, work + (long long)un * 128, ((unsigned)which * (unsigned)nunits * 128u + (unsigned)col) * 4u);   // (uniform base + per-thread offset)
  };
  int uprev = 0;
  int u = blockIdx.x;
  if (u >= nunits) return;
  set_unit(u);
  issue_F(uq, ug, ul, ud);
#pragma unroll
  for (int h = 0; h < 4; h++) issue_H(uq, h);
  bool first = true;

  for (;;) {
    const bool has_next = u + G < nunits;
    stat_lane = lane < 32 ? ul + lane : ud + (long long)(lane & 31) * nh;
    // ============================== phase Q ==============================
    f16v dq[4];
#pragma unroll
    for (int t = 0; t < 4; t++) dq[t] = (f16v){0};
    at_static_for<0, 4>([&](auto hc) __attribute__((always_inline)) {
      constexpr int h = decltype(hc)::value;
      LH_ST(3 * h);
      if constexpr (h == 0) {
        // F and - older - H0..H3 (steady state: newer are the 8 dV stores; a workgroup's first unit issued H0..H3 behind F: wait for all)
        lh_wait_vm2<8, 0>(first);
        // the fragments are registers the compiler tracks: tie them to the wait
        asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]) :: "memory");
        asm volatile("" : "+v"(gf[0]), "+v"(gf[1]), "+v"(gf[2]), "+v"(gf[3]), "+v"(gf[4]), "+v"(gf[5]), "+v"(gf[6]), "+v"(gf[7]), "+v"(lq), "+v"(dsum) :: "memory");
        lh_barrier();            // all four halves of every thread have landed: the halves below need no further barrier
        if (!first) reduce_partials(uprev);   // the previous unit's bias partials: every wave has left its epilogue (this barrier)  [1]
        // the two waves of a SIMD run the same instruction sequence from the same barrier: left alone they want the matrix pipe at the same
        // time and the vector ALU at the same time.  Waves 4-7 start one MFMA cluster late (LH_STAGGER x 64 cycles), so that one wave's
        // exponentials run under the other's MFMAs (s_setprio 1 inside the clusters keeps them apart)
        if (LH_STAGGER && wave >= 4) __builtin_amdgcn_s_sleep(LH_STAGGER);
      }
      LH_ST(3 * h + 1);
      unsigned stage = s0 + h * LH_SLOT;
      asm volatile("" : "+s"(stage));       // (opaque: the address registers below belong to this half only)
      // the fragment addresses of this half in registers: computing them in front of every read (xor + add per LDS instruction) made the
      // vector ALU as busy as the matrix pipe (~3 800 of a wave's instructions per unit against 448 MFMAs)
      unsigned ar[8], at4[4], bt4[4];
#pragma unroll
      for (int ks = 0; ks < 8; ks++) ar[ks] = (ra0 ^ (32u * ks)) + stage;
#pragma unroll
      for (int dt = 0; dt < 4; dt++) { at4[dt] = (ta0 ^ (64u * dt)) + stage; bt4[dt] = (tb0 ^ (64u * dt)) + stage; }
      bf8v fr[LH_QPOOL];
      f16v s, dp;
      bf8v dsf[2];
      auto issue = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int t = j / 24, qd = j % 24;
        if constexpr (qd < 16) {
          constexpr int ks = qd >> 1, isv = qd & 1;
          at_rd128a<isv * AT_HALF_BYTES + t * 8192>(fr[j % LH_QPOOL], ar[ks]);
        } else {
          constexpr int sub = (qd - 16) >> 2, dt = (qd - 16) & 3;
          at_rdtra<t * 8192 + sub * 4096, t * 8192 + sub * 4096>(fr[j % LH_QPOOL], at4[dt], bt4[dt]);
        }
      };
      at_static_for<0, LH_QPOOL - 1>(issue);
      AT_PRIO(1);
      at_static_for<0, 48>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int qd = j % 24;
        if constexpr (j + LH_QPOOL - 1 < 48) issue(std::integral_constant<int, j + LH_QPOOL - 1>{});
        if constexpr (qd == 16) {
          AT_PRIO(0);
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq));
            s[e] = p * (dp[e] - dsum) * scale;
          }
          dsf[0] = pack8(s, 0);
          dsf[1] = pack8(s, 1);
          AT_PRIO(1);
        }
        at_wait_lgkm<lh_q_after(j)>(fr[j % LH_QPOOL]);
        if constexpr (qd < 16) {
          constexpr int ks = qd >> 1;
          if constexpr (qd == 0) s = LH_MFMA(fr[j % LH_QPOOL], qf[0], (f16v){0}, 0, 0, 0);
          else if constexpr (qd == 1) dp = LH_MFMA(fr[j % LH_QPOOL], gf[0], (f16v){0}, 0, 0, 0);
          else if constexpr ((qd & 1) == 0) s = LH_MFMA(fr[j % LH_QPOOL], qf[ks], s, 0, 0, 0);
          else dp = LH_MFMA(fr[j % LH_QPOOL], gf[ks], dp, 0, 0, 0);
        } else {
          constexpr int sub = (qd - 16) >> 2, dt = (qd - 16) & 3;
          dq[dt] = LH_MFMA(fr[j % LH_QPOOL], dsf[sub], dq[dt], 0, 0, 0);
        }
      });
      AT_PRIO(0);
      LH_ST(3 * h + 2);
    });
    lh_barrier();                    // every wave is done with the four halves
    LH_ST(12);
    // phase K's first items go out here, in front of the dQ epilogue (refilling the slots half by half would need a barrier per half:
    // 0.9 us of wave skew each, tools/attn_lh_stamps.py)
    issue_V(0); issue_V(1);
    issue_KF();
    issue_T(0); issue_T(1); issue_T(2); issue_T(3);
    // dQ rows of this wave: dqkv[(line * S + 32 * wave + row)][head * 128 ..]
    bf16raw* uo = dqkv + (uq - qkv);
    lh_store_matrix(dq, stg, part, (unsigned char*)uo + (long long)(32 * wave) * pq, pq, lane);   // [8]
    LH_ST(13);

    // ============================== phase K ==============================
    f16v dv[4], dk[4];
#pragma unroll
    for (int t = 0; t < 4; t++) { dv[t] = (f16v){0}; dk[t] = (f16v){0}; }
    const unsigned vbase = s0 + wave * 8192;     // this wave's 32 rows of the V block (slots 0, 1)
    // (a run-time loop: one copy of the stage's code; the waits and the issues are uniform branches on j8)
    lh_wait_vm<17>();            // V0 V1, the K fragments, T0 (newer: T1 T2 T3 9 + dQ rows 8)
    asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]) :: "memory");
#pragma unroll 1
    for (int j8 = 0; j8 < 8; j8++) {
      LH_ST(14 + 3 * j8);
      if (j8 == 0) lh_wait_vm<17>();
      else if (j8 <= 3) lh_wait_vm<14>();
      else if (j8 <= 5) lh_wait_vm<6>();
      else if (j8 == 6) lh_wait_vm<3>();
      else lh_wait_vm<0>();
      lh_barrier();
      LH_ST(15 + 3 * j8);
      if (j8 >= 1 && j8 <= 4) issue_T(j8 + 3);
      if (LH_STAGGER_K && wave >= 4) __builtin_amdgcn_s_sleep(LH_STAGGER_K);
      const unsigned stage = s0 + 2 * LH_SLOT + (j8 & 3) * 16384;
      const unsigned stb = s0 + LH_STATS + (j8 & 3) * 256, sto = 16 * h5;
      bf8v fr[LH_KPOOL];
      f16v s, dp;
      auto issue_r = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int ks = j / 3, kind = j % 3;
        if constexpr (kind == 0) at_rd128x<0, 32 * ks>(fr[j % LH_KPOOL], ra0, stage);
        else if constexpr (kind == 1) at_rd128x<AT_SUB_BYTES, 32 * ks>(fr[j % LH_KPOOL], ra0, stage);
        else at_rd128x<0, 32 * ks>(fr[j % LH_KPOOL], ra0, vbase);
      };
      at_static_for<0, (LH_KPOOL - 2)>(issue_r);
      AT_PRIO(1);
      at_static_for<0, 24>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int ks = j / 3, kind = j % 3;
        if constexpr (j + (LH_KPOOL - 2) < 24) issue_r(std::integral_constant<int, j + (LH_KPOOL - 2)>{});
        constexpr int after = (24 - 1 - j) < (LH_KPOOL - 2) ? (24 - 1 - j) : (LH_KPOOL - 2);
        if constexpr (kind == 0) {
          at_wait_lgkm<after>(fr[j % LH_KPOOL]);
          if constexpr (ks == 0) s = LH_MFMA(fr[j % LH_KPOOL], kf[0], (f16v){0}, 0, 0, 0);
          else s = LH_MFMA(fr[j % LH_KPOOL], kf[ks], s, 0, 0, 0);
        } else if constexpr (kind == 2) {
          at_wait_lgkm<after>(fr[j % LH_KPOOL]);
          asm volatile("" : "+v"(fr[(j - 1) % LH_KPOOL]));
          if constexpr (ks == 0) dp = LH_MFMA(fr[(j - 1) % LH_KPOOL], fr[j % LH_KPOOL], (f16v){0}, 0, 0, 0);
          else dp = LH_MFMA(fr[(j - 1) % LH_KPOOL], fr[j % LH_KPOOL], dp, 0, 0, 0);
        }
      });
      AT_PRIO(0);
      auto issue_t = [&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int sub = m >> 3, dt = (m >> 1) & 3, kind = m & 1;
        constexpr int off = sub * 4096 + (kind == 0 ? AT_SUB_BYTES : 0);
        at_rdtrx<off, off, 64 * dt>(fr[m % LH_KPOOL], ta0, tb0, stage);
      };
      f4v l4[4], d4[4];
      auto issue_s = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        at_rd128<32 * g>(l4[g], sto, stb);
        at_rd128<128 + 32 * g>(d4[g], sto, stb);
      };
      issue_s(std::integral_constant<int, 0>{});
      issue_s(std::integral_constant<int, 1>{});
      bf8v pf[2], dsf[2];
      at_static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g4 = decltype(gc)::value;
        // LDS instructions issued after this group's statistics (S = 2 reads, T = 2 reads, T m issued only while m < LH_KPOOL - 1):
        //   g0: S1;  g1: S2 T0 T1;  g2: T0 T1 S3 T2 T3;  g3: T2 T3 T4 T5
        constexpr int TDK = LH_KPOOL - 1;
        constexpr int nt01 = 2 * ((0 < TDK) + (1 < TDK)), nt23 = 2 * ((2 < TDK) + (3 < TDK)), nt45 = 2 * ((4 < TDK) + (5 < TDK));
        constexpr int after = g4 == 0 ? 2 : g4 == 1 ? 2 + nt01 : g4 == 2 ? nt01 + 2 + nt23 : nt23 + nt45;
        at_wait_lgkm2<after>(l4[g4], d4[g4]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + e], c, -l4[g4][e]));
          s[4 * g4 + e] = p;
          dp[4 * g4 + e] = p * (dp[4 * g4 + e] - d4[g4][e]) * scale;
        }
        if constexpr (g4 == 1) { pf[0] = pack8(s, 0); dsf[0] = pack8(dp, 0); }
        if constexpr (g4 == 3) { pf[1] = pack8(s, 1); dsf[1] = pack8(dp, 1); }
        if constexpr (g4 + 2 < 4) issue_s(std::integral_constant<int, g4 + 2>{});
        if constexpr (2 * g4 < (LH_KPOOL - 1)) issue_t(std::integral_constant<int, 2 * g4>{});
        if constexpr (2 * g4 + 1 < (LH_KPOOL - 1)) issue_t(std::integral_constant<int, 2 * g4 + 1>{});
      });
      AT_PRIO(1);
      at_static_for<0, 16>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        constexpr int sub = m >> 3, dt = (m >> 1) & 3, kind = m & 1;
        if constexpr (m + (LH_KPOOL - 1) < 16) issue_t(std::integral_constant<int, m + (LH_KPOOL - 1)>{});
        constexpr int after = 2 * ((16 - 1 - m) < (LH_KPOOL - 1) ? (16 - 1 - m) : (LH_KPOOL - 1));
        at_wait_lgkm<after>(fr[m % LH_KPOOL]);
        if constexpr (kind == 0) dv[dt] = LH_MFMA(fr[m % LH_KPOOL], pf[sub], dv[dt], 0, 0, 0);
        else dk[dt] = LH_MFMA(fr[m % LH_KPOOL], dsf[sub], dk[dt], 0, 0, 0);
      });
      AT_PRIO(0);
      LH_ST(16 + 3 * j8);
    }
    lh_barrier();                    // every wave is done with the ring
    LH_ST(38);
    const int ucur = u;
    set_unit(has_next ? u + G : u);
    if (has_next) {
#pragma unroll
      for (int h = 0; h < 4; h++) issue_H(uq, h);
    }
    // dK / dV rows of this wave; the next unit's fragment loads go out between the two (their 64 registers become free with dK).  They
    // are issued unconditionally (a workgroup's last unit re-reads its own rows): a definition under `if (has_next)` keeps the old
    // fragments alive through phase K in the register allocator's eyes - 64 registers spilled and reloaded per unit
    lh_store_matrix(dk, stg, part + 512, (unsigned char*)(uo + d) + (long long)(32 * wave) * pq, pq, lane);       // [8]
    LH_ST(39);
    issue_F(uq, ug, ul, ud);
    lh_store_matrix(dv, stg, part + 1024, (unsigned char*)(uo + 2 * d) + (long long)(32 * wave) * pq, pq, lane);  // [8]
    LH_ST(40);
    lh_wait_lgkm_plain<0>();         // this wave's partial rows are in LDS: the next barrier publishes them (reduce_partials)
    uprev = ucur;
    LH_ST(41);
#ifdef LH_STAMP
    stamp_unit++;
#endif
    if (!has_next) break;
    u += G;
    first = false;
  }
  lh_barrier();
  reduce_partials(uprev);
  lh_wait_vm<0>();
#ifdef LH_STAMP
  if (tid < 64) {
    float lo, hi;
    lh_ds_read4(lo, s0 + LH_LDS_BYTES + 8 * tid);
    lh_ds_read4(hi, s0 + LH_LDS_BYTES + 8 * tid + 4);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) :: "memory");
    float* dbg = work + 3LL * nunits * 128 + ((long long)blockIdx.x * 64 + tid) * 2;
    dbg[0] = lo; dbg[1] = hi;
  }
#endif
}

// ---- forward, pipelined operand reads (round 3; see the note in front of the `_p` backward bodies).  hipcc's schedule of attn_fwd_k
// puts an `s_waitcnt vmcnt(0)` right behind the counted vmcnt(8) at the loop top and another one in front of the V reads - the
// K(u+1) tile issued after the scores is waited for before P V, the V(u) wait cannot be deferred - and serialises read -> wait -> MFMA.
// Here every LDS access of the loop (fragments, the O staging of a head's last tile) is inline asm: 32 K row fragments for the scores
// (six in flight), the first seven transposed V fragments in flight while the exponentials run, then rolling.  Same MFMAs in the
// same order: out and lse are bit-identical to attn_fwd_k.
#define FW_POOL 8
__global__ __launch_bounds__(256, 2) void attn_fwd_p_k(const bf16raw* qkv, bf16raw* out, float* lse2, int S, int nh, int hpb, float c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + AT_TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const int nqb = S >> 7, ngrp = nh / hpb;
  int lg, qb;
  attn_block_map(blockIdx.x, nqb, gridDim.x / nqb, lg, qb);
  const int line = lg / ngrp, head0 = (lg % ngrp) * hpb;
  const long long d = (long long)nh * 128, ld = 3 * d;
  const bf16raw* lbase = qkv + (long long)line * S * ld;  // + head * 128 : q ; + d : k ; + 2d : v
  const int q = qb * 128 + wave * 32 + r;  // this lane's query (both lane halves hold the same query)
  const int nkt = S >> 7, units = hpb * nkt;

  attn_glds_tile<false>(lbase + head0 * 128 + d, ld, kimg, wave, lane);
  attn_glds_tile<true>(lbase + head0 * 128 + 2 * d, ld, vimg, wave, lane);

  // this lane's Q row fragments by loads the compiler does not see (it waits vmcnt(0) for its own loads once LDS-DMA is in flight, which
  // would also wait for the V tile the loop top lets fly): uniform base + 32-bit lane offset
  bf8v qf[8];
  const unsigned qoff = (unsigned)((long long)q * ld * 2 + 16 * h5);
  auto load_q = [&](int head) {
    const bf16raw* qb_ = lbase + head * 128;
    at_static_for<0, 8>([&](auto kc) __attribute__((always_inline)) { constexpr int ks = decltype(kc)::value; lh_gload16<32 * ks>(qf[ks], qb_, qoff); });
  };
  load_q(head0);
  // fragment addresses: K row fragment (key 32 t + r, k-step ks) = ka0 ^ (32 ks) + 8192 t;  V^T fragment (keys 32 t + 16 sub + .., head-dim
  // tile dt) = va0 ^ (64 dt) + 8192 t + 4096 sub (second read + 2048)
  const unsigned kbase = at_lds_addr(kimg), vbase = at_lds_addr(vimg);
  const unsigned ka0 = (unsigned)(r * 256 + ((h5 ^ (r & 15)) << 4));
  unsigned va0;
  {
    const int i = lane & 15, g1 = (lane >> 4) & 1;
    const int key = 4 * h5 + (i >> 2);
    va0 = (unsigned)(key * 256 + ((key & 3) << 6) + g1 * 32 + (i & 3) * 8);
  }
  f16v o[4];
#pragma unroll
  for (int t = 0; t < 4; t++) o[t] = (f16v){0};
  float m = -INFINITY, l = 0.f;

  for (int u = 0; u < units; u++) {
    const int head = head0 + u / nkt, kt = u % nkt;
    if (u > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // K(u) (and the head's Q rows); V(u), the newest eight, may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]) :: "memory");   // (older than V(u): landed)
    lh_barrier();  // K(u) landed
    bf8v fr[FW_POOL];
    f16v s[4];
    // ---- S^T = K Q^T: 32 row fragments (t, ks), six in flight
    auto issue_k = [&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int t = j >> 3, ks = j & 7;
      at_rd128x<t * 8192, 32 * ks>(fr[j % FW_POOL], ka0, kbase);
    };
    at_static_for<0, 6>(issue_k);
    AT_PRIO(1);
    at_static_for<0, 32>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      constexpr int t = j >> 3, ks = j & 7;
      if constexpr (j + 6 < 32) issue_k(std::integral_constant<int, j + 6>{});
      constexpr int after = (31 - j) < 6 ? (31 - j) : 6;
      at_wait_lgkm<after>(fr[j % FW_POOL]);
      if constexpr (ks == 0) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % FW_POOL], qf[0], (f16v){0}, 0, 0, 0);
      else s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[j % FW_POOL], qf[ks], s[t], 0, 0, 0);
    });
    AT_PRIO(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own part of V(u)
    lh_barrier();  // every wave is done with the K image; V(u) landed
    if (u + 1 < units) {
      const int nhd = head0 + (u + 1) / nkt, nkt_i = (u + 1) % nkt;
      attn_glds_tile<false>(lbase + nhd * 128 + d + (long long)nkt_i * 128 * ld, ld, kimg, wave, lane);
    }
    // ---- transposed V fragments: m = (t, sub, dt), seven in flight; the first seven go out in front of the softmax arithmetic
    auto issue_v = [&](auto mc) __attribute__((always_inline)) {
      constexpr int mm = decltype(mc)::value;
      constexpr int t = mm >> 3, sub = (mm >> 2) & 1, dt = mm & 3;
      at_rdtrx<t * 8192 + sub * 4096, t * 8192 + sub * 4096 + 2048, 64 * dt>(fr[mm % FW_POOL], va0, va0, vbase);
    };
    at_static_for<0, 7>(issue_v);

    // ---- online softmax, all lane-local except one lane^32 exchange per reduction
    float mx = s[0][0];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) mx = fmaxf(mx, s[t][e]);
    {   // lanes l and l + 32 hold the two halves of a query's scores: v_permlane32_swap instead of a trip through the LDS crossbar
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);  // exp2(-inf) = 0 on a head's first tile
    const float mc = mn * c;
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[t][e], c, -mc));
        s[t][e] = p;
        ps += p;
      }
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
      ps = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    l = l * alpha + ps;
    m = mn;
    if (kt != 0) {  // (a head's first tile: O is still zero)
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) o[t][e] *= alpha;
    }

    // ---- O^T += V^T P^T
    AT_PRIO(1);
    at_static_for<0, 32>([&](auto mc_) __attribute__((always_inline)) {
      constexpr int mm = decltype(mc_)::value;
      constexpr int t = mm >> 3, sub = (mm >> 2) & 1, dt = mm & 3;
      if constexpr (mm + 7 < 32) issue_v(std::integral_constant<int, mm + 7>{});
      constexpr int after = 2 * ((31 - mm) < 7 ? (31 - mm) : 7);
      at_wait_lgkm<after>(fr[mm % FW_POOL]);
      const bf8v pf = pack8(s[t], sub);
      o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[mm % FW_POOL], pf, o[dt], 0, 0, 0);
    });
    AT_PRIO(0);
    if (kt == nkt - 1) {
      // ---- head finished: O[q][d] = o[dt][reg] / l, staged through the (now free) V image so that HBM sees whole 256-byte rows in
      // 16-byte lanes.  Image: 128 rows x 256 B, 8-byte granule index XORed with (row & 31).  (LDS accesses by asm: see the note above)
      lh_barrier();  // every wave is done with the V image
      const float inv = 1.0f / l;
      int qrow_l = wave * 32 + r, tid_l = tid;
      asm volatile("" : "+v"(qrow_l), "+v"(tid_l));  // keep the staging addresses out of the unit loop's live set
#pragma unroll
      for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          const at_u2v w = {pack2bf(o[dt][4 * g4 + 0] * inv, o[dt][4 * g4 + 1] * inv), pack2bf(o[dt][4 * g4 + 2] * inv, o[dt][4 * g4 + 3] * inv)};
          const int g = dt * 8 + 2 * g4 + h5;  // granule of d = dt*32 + 8*g4 + 4*h5
          lh_ds_write8(vbase + qrow_l * 256 + ((g ^ (qrow_l & 31)) << 3), w);
        }
      // base-2 LSE of c*scores (both lane halves hold it and store it: same value, same address)
      lh_gstore4(m * c + __builtin_amdgcn_logf(l), lse2 + ((long long)line * nh + head) * S, (unsigned)q * 4u);
      lh_wait_lgkm_plain<0>();
      lh_barrier();
      {
        const int ch = tid_l & 15;
        at_u4v v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int row = (tid_l >> 4) + 16 * i;
          lh_ds_read16(v[i], vbase + row * 256 + ((ch ^ ((row & 31) >> 1)) << 4));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int row = (tid_l >> 4) + 16 * i;
          at_u4v x = v[i];
          if (row & 1) x = (at_u4v){v[i][2], v[i][3], v[i][0], v[i][1]};
          lh_gstore16<0>(x, out + ((long long)line * S + qb * 128) * d + head * 128, (unsigned)(row * (int)d * 2 + ch * 16));
        }
      }
      if (u + 1 < units) {  // next head: fresh statistics, its Q rows (the loads complete under the loop-top wait)
#pragma unroll
        for (int t = 0; t < 4; t++) o[t] = (f16v){0};
        m = -INFINITY;
        l = 0.f;
        load_q(head + 1);
      }
    }
    if (u + 1 < units) {
      lh_barrier();  // every wave is done with the V image (and with the O staging reads)
      const int nhd = head0 + (u + 1) / nkt, nkt_i = (u + 1) % nkt;
      attn_glds_tile<true>(lbase + nhd * 128 + 2 * d + (long long)nkt_i * 128 * ld, ld, vimg, wave, lane);
    }
  }
}


int g_attn_bwd_pair = 1;  // pero_set_option("attn_bwd_pair", 0 / 1)
int g_attn_order = 32;    // pero_set_option("attn_order", n): dispatch order of the paired backward's blocks (see attn_bwd_pair_k; 0 = a unit's four blocks side by side)
int g_attn_lh = 0;        // pero_set_option("attn_lh", 0 / 1): S = 256 with D handed in and a bias gradient wanted -> the persistent (line, head) kernel
                          // attn_bwd_lh_k.  Same bits; measured 735-745 us against 725-735 us of the paired kernels at 1024 lines: off
int g_attn_pipe = 1;      // pero_set_option("attn_pipe", 0 / 1): the bodies with software-pipelined operand reads (default) / the compiler-scheduled ones
extern "C" int pero_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dvec, void* dqkv,
                                  float* dbias, float* work, int64_t N, int64_t S, int64_t num_heads, int64_t head_dim, int dtype,
                                  void* stream) {
  PERO_REQUIRE(qkv && dout && lse && dvec && dqkv, "pero_attention_bwd: null pointer");
  PERO_REQUIRE(dtype == PERO_BF16 && head_dim == 128 && S % 128 == 0 && S > 0 && N > 0 && num_heads > 0,
               "pero_attention_bwd: fused kernel needs bf16, head_dim 128, S %% 128 == 0");
  PERO_REQUIRE(aligned16(qkv) && (!out || aligned16(out)) && aligned16(dout) && aligned16(dqkv), "pero_attention_bwd: 16-byte alignment");
  PERO_REQUIRE(!dbias || work, "pero_attention_bwd: dbias needs the partial-sum workspace");
  PERO_LDS_ATTR(attn_bwd_dq_k<false>, 2 * AT_TILE_BYTES);
  PERO_LDS_ATTR(attn_bwd_dq_k<true>, 2 * AT_TILE_BYTES);
  PERO_LDS_ATTR(attn_bwd_dkv2_k<false>, AT_DKV2_LDS);
  PERO_LDS_ATTR(attn_bwd_dkv2_k<true>, AT_DKV2_LDS);
  PERO_LDS_ATTR(attn_bwd_pair_k<false>, AT_DKV2_LDS > 2 * AT_TILE_BYTES ? AT_DKV2_LDS : 2 * AT_TILE_BYTES);
  PERO_LDS_ATTR(attn_bwd_pair_k<true>, AT_DKV2_LDS > 2 * AT_TILE_BYTES ? AT_DKV2_LDS : 2 * AT_TILE_BYTES);
  const float scale = (float)(1.0 / sqrt((double)head_dim));
  const float c = (float)(1.4426950408889634 / sqrt((double)head_dim));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(N * num_heads * (S / 128))), block(256);
  if (!out && g_attn_bwd_pair && g_attn_lh && S == 256 && dbias && N * num_heads < (1LL << 20) && num_heads <= 1024) {   // (32-bit byte offsets inside a unit and inside the partial-sum workspace)
    // one persistent workgroup per CU, a (line, head) per pass (attn_bwd_lh_k); its bias partials: one row per unit
#ifdef LH_STAMP
#define LH_LAUNCH_LDS (LH_LDS_BYTES + 512)
#else
#define LH_LAUNCH_LDS LH_LDS_BYTES
#endif
    PERO_LDS_ATTR(attn_bwd_lh_k, LH_LAUNCH_LDS);
    const long long units = N * num_heads;
    const int cus = pero_num_cus();
    hipLaunchKernelGGL(attn_bwd_lh_k, dim3((unsigned)(units < cus ? units : cus)), dim3(512), LH_LAUNCH_LDS, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse,
                       (const float*)dvec, (bf16raw*)dqkv, work, (int)units, (int)num_heads, c, scale);
    hipLaunchKernelGGL(attn_bias_reduce_k, dim3((unsigned)num_heads, 3, N >= 4096 ? 128 : N >= 1024 ? 64 : 16), dim3(128), 0, st, work, dbias, (int)N, (int)num_heads, 1);
    PERO_CHECK_LAUNCH("pero_attention_bwd");
    return PERO_OK;
  }
  if (!out && g_attn_bwd_pair) {
    const size_t lds = AT_DKV2_LDS > 2 * AT_TILE_BYTES ? AT_DKV2_LDS : 2 * AT_TILE_BYTES;
    if (g_attn_pipe)
      hipLaunchKernelGGL(attn_bwd_pair_k<true>, dim3(2 * grid.x), block, lds, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec, (bf16raw*)dqkv,
                         dbias ? work : nullptr, (int)S, (int)num_heads, c, scale, g_attn_order);
    else
      hipLaunchKernelGGL(attn_bwd_pair_k<false>, dim3(2 * grid.x), block, lds, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec, (bf16raw*)dqkv,
                         dbias ? work : nullptr, (int)S, (int)num_heads, c, scale, g_attn_order);
  } else {
  if (g_attn_pipe)
    hipLaunchKernelGGL(attn_bwd_dq_k<true>, grid, block, 2 * AT_TILE_BYTES, st, (const bf16raw*)qkv, (const bf16raw*)out, (const bf16raw*)dout, lse,
                       dvec, (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  else
    hipLaunchKernelGGL(attn_bwd_dq_k<false>, grid, block, 2 * AT_TILE_BYTES, st, (const bf16raw*)qkv, (const bf16raw*)out, (const bf16raw*)dout, lse,
                       dvec, (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  if (g_attn_pipe)
    hipLaunchKernelGGL(attn_bwd_dkv2_k<true>, grid, block, AT_DKV2_LDS, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec,
                       (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  else
    hipLaunchKernelGGL(attn_bwd_dkv2_k<false>, grid, block, AT_DKV2_LDS, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec,
                       (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  }
  if (dbias)
    hipLaunchKernelGGL(attn_bias_reduce_k, dim3((unsigned)num_heads, 3, (N * (S / 128) >= 4096) ? 128 : (N * (S / 128) >= 1024) ? 64 : 16), dim3(128), 0, st, work, dbias, (int)N, (int)num_heads, (int)(S / 128));
  PERO_CHECK_LAUNCH("pero_attention_bwd");
  return PERO_OK;
}
