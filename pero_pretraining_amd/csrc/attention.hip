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

extern "C" int pero_attention_fwd(const void* qkv, void* out, float* lse, int64_t N, int64_t S, int64_t num_heads,
                                  int64_t head_dim, int dtype, void* stream) {
  PERO_REQUIRE(qkv && out && lse, "pero_attention_fwd: null pointer");
  PERO_REQUIRE(dtype == PERO_BF16 && head_dim == 128 && S % 128 == 0 && S > 0 && N > 0 && num_heads > 0,
               "pero_attention_fwd: fused kernel needs bf16, head_dim 128, S %% 128 == 0 (got hd=%lld S=%lld)", (long long)head_dim, (long long)S);
  PERO_REQUIRE(aligned16(qkv) && aligned16(out), "pero_attention_fwd: 16-byte alignment");
  PERO_LDS_ATTR(attn_fwd_k, 2 * AT_TILE_BYTES);
  const float c = (float)(1.4426950408889634 / sqrt((double)head_dim));
  const int hpb = attn_heads_per_block(N, S, num_heads);
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

// Epilogue of the backward kernels: a 128 x 128 gradient tile held as acc[dt][e] (row = this lane's query / key
// `wave * 32 + r`, columns d = dt*32 + 8*(e>>2) + 4*h5 + (e&3)) goes through LDS as bf16 rows and leaves in 16-byte row
// segments (the direct form was 16 scattered 8-byte stores per lane); the staged rows also give the tile's column sums -
// this (line, head) block's share of in_proj's bias gradient - for 128 (x2) atomics instead of a pass over dqkv.
// Image: 128 rows x 256 B, 8-byte granule index XORed with (row & 31): conflict-free ds_write_b64 and ds_read_b128.
__device__ __forceinline__ void attn_store_tile(const f16v (&acc)[4], unsigned char* stg, bf16raw* out_base, long long ld,
                                                float* colsum, int tid, int wave, int r, int h5) {
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
    *(uint4*)(out_base + (long long)row * ld + ch * 8) = v;
    if (colsum) {
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; k++) { cs[2 * k] += __uint_as_float(w[k] << 16); cs[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
    }
  }
  if (colsum) {
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
  float s0 = 0.f, s1 = 0.f;
  int i = i0;
  for (; i + 2 <= i1; i += 2) {
    s0 += p[(((long long)(i / nblk) * nh + head) * nblk + i % nblk) * 128 + c];
    s1 += p[(((long long)((i + 1) / nblk) * nh + head) * nblk + (i + 1) % nblk) * 128 + c];
  }
  if (i < i1) s0 += p[(((long long)(i / nblk) * nh + head) * nblk + i % nblk) * 128 + c];
  if (i0 < i1) atomicAdd(dbias + (long long)which * nh * 128 + head * 128 + c, s0 + s1);
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
    if (hk + 1 < nhalf) {
      unsigned char* nb = smem + ((hk + 1) & 1) * 2 * AT_HALF_BYTES;
      attn_glds_half(Kg + (long long)(hk + 1) * 64 * ld, ld, nb, wave, lane);
      attn_glds_half(Vg + (long long)(hk + 1) * 64 * ld, ld, nb + AT_HALF_BYTES, wave, lane);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {  // 32-key sub-tile
      f16v s = {0}, dp = {0};
      AT_PRIO(1);
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(kimg, t * 32 + r, ks, h5), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_row_frag(vimg, t * 32 + r, ks, h5), gf[ks], dp, 0, 0, 0);
      }
      AT_PRIO(0);
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[e], c, -lq));
        s[e] = p * (dp[e] - dsum) * scale;  // dS^T
      }
      AT_PRIO(1);
#pragma unroll
      for (int sub = 0; sub < 2; sub++) {
        const bf8v dsf = pack8(s, sub);
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(img_tr_frag(kimg, t * 32 + sub * 16, dt, lane), dsf, dq[dt], 0, 0, 0);
      }
      AT_PRIO(0);
    }
  }
  // dbias: partial-sum workspace [3][workgroups][128] (q, k, v); this kernel fills plane 0
  attn_store_tile(dq, smem, dqkv + ((long long)line * S + qb * 128) * ld + head * 128, ld,
                  dbias ? dbias + ((long long)lh * nqb + qb) * 128 : nullptr, tid, wave, r, h5);
}
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_k(const bf16raw* qkv, const bf16raw* out, const bf16raw* dout, const float* lse2,
                                                        float* dvec, bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nqb = S >> 7;
  int lh, qb;
  attn_block_map(blockIdx.x, nqb, gridDim.x / nqb, lh, qb);
  attn_bwd_dq_body(smem, lh, qb, qkv, out, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
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
    if (sq + 1 < nsub) {
      if (tid < 64) nstat = stat[(sq + 1) * stat_step];  // before the DMA: vmcnt is in-order
      unsigned char* nb = smem + ((sq + 1) & 1) * 2 * AT_SUB_BYTES;
      attn_glds_sub(base + (long long)(sq + 1) * 32 * ld, ld, nb, wave, lane);
      attn_glds_sub(Gg + (long long)(sq + 1) * 32 * d, d, nb + AT_SUB_BYTES, wave, lane);
    }
    // rows q = (e&3) + 8(e>>2) + 4*h5 of the 32-query stage on the registers, key on the lane
    f16v s = {0}, dp = {0};
    AT_PRIO(1);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
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
        const float p = __builtin_amdgcn_exp2f(fmaf(s[4 * g4 + e], c, -l4[e]));
        s[4 * g4 + e] = p;                                         // P
        dp[4 * g4 + e] = p * (dp[4 * g4 + e] - d4[e]) * scale;     // dS
      }
    }
    AT_PRIO(1);
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const bf8v pf = pack8(s, sub), dsf = pack8(dp, sub);
#pragma unroll
      for (int dt = 0; dt < 4; dt++) {
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
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv2_k(const bf16raw* qkv, const bf16raw* dout, const float* lse2, const float* dvec,
                                                          bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nkb = S >> 7;
  int lh, kb;
  attn_block_map(blockIdx.x, nkb, gridDim.x / nkb, lh, kb);
  attn_bwd_dkv2_body(smem, lh, kb, gridDim.x, qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
}
// Both backward kernels as ONE launch (D already computed: `out` is not read, so no workgroup depends on another): the 2 x (S / 128)
// workgroups of a (line, head) - its dQ blocks and its dK / dV blocks, which all read the same Q, K, V and dO rows - sit next to
// each other in one XCD's dispatch order, so the rows come from HBM once and the other readers find them in that XCD's L2
// (FETCH_SIZE of the backward at 256 lines: 534 MB as two launches, 308 MB paired, 267 MB = each row once; 789 -> 740 us at 1024 lines).
__global__ __launch_bounds__(256, 2) void attn_bwd_pair_k(const bf16raw* qkv, const bf16raw* dout, const float* lse2, float* dvec,
                                                          bf16raw* dqkv, float* dbias, int S, int nh, float c, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nb = S >> 7;
  int lh, blk;
  attn_block_map(blockIdx.x, 2 * nb, gridDim.x / (2 * nb), lh, blk);
  if (blk < nb)
    attn_bwd_dq_body(smem, lh, blk, qkv, nullptr, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
  else
    attn_bwd_dkv2_body(smem, lh, blk - nb, (long long)(gridDim.x >> 1), qkv, dout, lse2, dvec, dqkv, dbias, S, nh, c, scale);
}

int g_attn_bwd_pair = 1;  // pero_set_option("attn_bwd_pair", 0 / 1)
extern "C" int pero_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* dvec, void* dqkv,
                                  float* dbias, float* work, int64_t N, int64_t S, int64_t num_heads, int64_t head_dim, int dtype,
                                  void* stream) {
  PERO_REQUIRE(qkv && dout && lse && dvec && dqkv, "pero_attention_bwd: null pointer");
  PERO_REQUIRE(dtype == PERO_BF16 && head_dim == 128 && S % 128 == 0 && S > 0 && N > 0 && num_heads > 0,
               "pero_attention_bwd: fused kernel needs bf16, head_dim 128, S %% 128 == 0");
  PERO_REQUIRE(aligned16(qkv) && (!out || aligned16(out)) && aligned16(dout) && aligned16(dqkv), "pero_attention_bwd: 16-byte alignment");
  PERO_REQUIRE(!dbias || work, "pero_attention_bwd: dbias needs the partial-sum workspace");
  PERO_LDS_ATTR(attn_bwd_dq_k, 2 * AT_TILE_BYTES);
  PERO_LDS_ATTR(attn_bwd_dkv2_k, AT_DKV2_LDS);
  PERO_LDS_ATTR(attn_bwd_pair_k, AT_DKV2_LDS > 2 * AT_TILE_BYTES ? AT_DKV2_LDS : 2 * AT_TILE_BYTES);
  const float scale = (float)(1.0 / sqrt((double)head_dim));
  const float c = (float)(1.4426950408889634 / sqrt((double)head_dim));
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(N * num_heads * (S / 128))), block(256);
  if (!out && g_attn_bwd_pair) {
    const size_t lds = AT_DKV2_LDS > 2 * AT_TILE_BYTES ? AT_DKV2_LDS : 2 * AT_TILE_BYTES;
    hipLaunchKernelGGL(attn_bwd_pair_k, dim3(2 * grid.x), block, lds, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec, (bf16raw*)dqkv,
                       dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  } else {
  hipLaunchKernelGGL(attn_bwd_dq_k, grid, block, 2 * AT_TILE_BYTES, st, (const bf16raw*)qkv, (const bf16raw*)out, (const bf16raw*)dout, lse,
                     dvec, (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  hipLaunchKernelGGL(attn_bwd_dkv2_k, grid, block, AT_DKV2_LDS, st, (const bf16raw*)qkv, (const bf16raw*)dout, lse, dvec,
                     (bf16raw*)dqkv, dbias ? work : nullptr, (int)S, (int)num_heads, c, scale);
  }
  if (dbias)
    hipLaunchKernelGGL(attn_bias_reduce_k, dim3((unsigned)num_heads, 3, (N * (S / 128) >= 1024) ? 64 : 16), dim3(128), 0, st, work, dbias, (int)N, (int)num_heads, (int)(S / 128));
  PERO_CHECK_LAUNCH("pero_attention_bwd");
  return PERO_OK;
}
