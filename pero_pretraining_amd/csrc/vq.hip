// Codebook nearest-neighbour (VQ-VAE quantizer / Feature-Quantization labels) in exact f32.
//
//   indices[m] = argmin_k (|x_m|^2 + |e_k|^2) - 2 x_m . e_k        (first minimum wins)
//
// The (M, K) distance matrix and the one-hot matrix of the reference are never materialised: each block
// keeps 64 feature rows, streams the codebook through LDS in 128-code tiles, multiplies on
// v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain, so indices reproduce an f32 CPU evaluation)
// with the CODES on the MFMA row index and the FEATURE ROWS on the lane index, so that the running
// (min, argmin) of a feature row lives in one lane's registers and needs no cross-lane traffic until
// the very end.  Algorithmic bytes: codebook once (K*D*4) + features once (M*D*4) + 8 B/row out.
#include "common.hpp"

#define VQ_BX 64    // feature rows per block
#define VQ_BC 128   // codes per LDS tile
#define VQ_BK 16    // depth per LDS tile

__global__ __launch_bounds__(256) void sqnorm_k(const float* x, float* out, long long rows, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = threadIdx.x & 63; c < d; c += 64) { const float v = x[row * d + c]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void vq_argmin_k(const float* x, const float* e, const float* sx, const float* se,
                                                   int64_t* indices, float* best_out, long long M, int K, int D) {
  __shared__ float Es[VQ_BC][VQ_BK + 1];
  __shared__ float Xs[VQ_BX][VQ_BK + 1];
  __shared__ float cand_v[2][VQ_BX];
  __shared__ int cand_i[2][VQ_BX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wx = wave & 1;  // wave tile: codes [wc*64, +64) x rows [wx*32, +32)
  const long long m0 = (long long)blockIdx.x * VQ_BX;
  const long long mrow = m0 + wx * 32 + (lane & 31);
  const float sxm = mrow < M ? sx[mrow] : 0.f;
  float best = INFINITY;
  int besti = 0x7fffffff;

  for (int c0 = 0; c0 < K; c0 += VQ_BC) {
    f16v acc0 = {0}, acc1 = {0};
    for (int k0 = 0; k0 < D; k0 += VQ_BK) {
#pragma unroll
      for (int i = 0; i < 8; i++) {  // 128 x 16 code tile
        const int idx = tid + 256 * i, r = idx >> 4, k = idx & 15;
        Es[r][k] = (c0 + r < K && k0 + k < D) ? e[(long long)(c0 + r) * D + k0 + k] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {  // 64 x 16 feature tile
        const int idx = tid + 256 * i, r = idx >> 4, k = idx & 15;
        Xs[r][k] = (m0 + r < M && k0 + k < D) ? x[(m0 + r) * D + k0 + k] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < VQ_BK; k += 2) {
        const float xb = Xs[wx * 32 + (lane & 31)][k + (lane >> 5)];
        const float e0 = Es[wc * 64 + (lane & 31)][k + (lane >> 5)];
        const float e1 = Es[wc * 64 + 32 + (lane & 31)][k + (lane >> 5)];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, xb, acc0, 0, 0, 0);  // D[code][row]
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(e1, xb, acc1, 0, 0, 0);
      }
      __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int code = c0 + wc * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (code < K) {
          const float dot = t == 0 ? acc0[r] : acc1[r];
          const float dist = (sxm + se[code]) - 2.0f * dot;
          if (dist < best) { best = dist; besti = code; }  // codes visited in increasing order per lane
        }
      }
    }
  }
  // the two lane halves hold interleaved code rows of the same feature row
  {
    const float ov = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(besti, 32, 64);
    if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if (lane < 32) { cand_v[wc][wx * 32 + lane] = best; cand_i[wc][wx * 32 + lane] = besti; }
  __syncthreads();
  if (tid < VQ_BX && m0 + tid < M) {
    float b = cand_v[0][tid];
    int bi = cand_i[0][tid];
    const float ov = cand_v[1][tid];
    const int oi = cand_i[1][tid];
    if (ov < b || (ov == b && oi < bi)) { b = ov; bi = oi; }
    indices[m0 + tid] = bi;
    if (best_out) best_out[m0 + tid] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path (K % 128 == 0, D % 32 == 0, M % 64 == 0): 64 feature rows x 128 codes x 32 depth per stage, tiles
// staged by LDS-DMA into a double buffer (24 KiB per stage, three workgroups per CU), fragments read with ds_read_b128
// (4 consecutive depth values per lane: MFMA step e multiplies depth 8j+e on lanes 0-31 and 8j+4+e on lanes 32-63, so
// one 16-byte read feeds four v_mfma_f32_32x32x2_f32).  Wave w owns codes [32w, 32w+32) of the tile x all 64 rows.
// ---------------------------------------------------------------------------------------------------------------
#define VF_BX 64
#define VF_BC 128
#define VF_BK 32
#define VF_EBYTES (VF_BC * VF_BK * 4)   // 16 KiB
#define VF_XBYTES (VF_BX * VF_BK * 4)   // 8 KiB
#define VF_STAGE (VF_EBYTES + VF_XBYTES)

// rows x 32 floats tile (128-byte rows), 16-byte chunk index XORed with (row & 7); piece = 8 rows = 1 KiB
__device__ __forceinline__ void vf_glds(const float* g, long long ld, int rows, unsigned char* lds, int wave, int lane) {
  const int npieces = rows >> 3;
  for (int p = wave; p < npieces; p += 4) {
    const int row = 8 * p + (lane >> 3), slot = lane & 7;
    const int chunk = slot ^ (row & 7);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (long long)row * ld + chunk * 4),
                                     (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ f4v vf_frag(const unsigned char* img, int row, int j, int h5) {
  return *(const f4v*)(img + row * 128 + ((((2 * j + h5)) ^ (row & 7)) << 4));
}

__global__ __launch_bounds__(256, 3) void vq_argmin_fast_k(const float* x, const float* e, const float* sx, const float* se,
                                                           int64_t* indices, float* best_out, int K, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float cand_v[4][VF_BX];
  __shared__ int cand_i[4][VF_BX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h5 = lane >> 5, r = lane & 31;
  const long long m0 = (long long)blockIdx.x * VF_BX;
  const float sx0 = sx[m0 + r], sx1 = sx[m0 + 32 + r];
  float best0 = INFINITY, best1 = INFINITY;
  int bi0 = 0x7fffffff, bi1 = 0x7fffffff;
  const int nks = D / VF_BK, nct = K / VF_BC;
  const int nstage = nks * nct;

  // stage s = (code tile s / nks, depth block s % nks)
  vf_glds(e, D, VF_BC, smem, wave, lane);
  vf_glds(x + m0 * D, D, VF_BX, smem + VF_EBYTES, wave, lane);
  f16v acc0 = {0}, acc1 = {0};
  for (int s = 0; s < nstage; s++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* eimg = smem + (s & 1) * VF_STAGE;
    const unsigned char* ximg = eimg + VF_EBYTES;
    if (s + 1 < nstage) {
      const int ct = (s + 1) / nks, kb = (s + 1) % nks;
      unsigned char* d = smem + ((s + 1) & 1) * VF_STAGE;
      vf_glds(e + (long long)ct * VF_BC * D + kb * VF_BK, D, VF_BC, d, wave, lane);
      vf_glds(x + m0 * D + kb * VF_BK, D, VF_BX, d + VF_EBYTES, wave, lane);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const f4v ef = vf_frag(eimg, wave * 32 + r, j, h5);
      const f4v x0 = vf_frag(ximg, r, j, h5);
      const f4v x1 = vf_frag(ximg, 32 + r, j, h5);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[q], x0[q], acc0, 0, 0, 0);  // D[code][row]
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ef[q], x1[q], acc1, 0, 0, 0);
      }
    }
    if ((s % nks) == nks - 1) {  // a 128-code tile is complete: fold it into the running argmin, restart the sums
      const int c0 = (s / nks) * VF_BC + wave * 32;
#pragma unroll
      for (int rr = 0; rr < 16; rr++) {
        const int code = c0 + (rr & 3) + 8 * (rr >> 2) + 4 * h5;
        const float sec = se[code];
        const float d0 = (sx0 + sec) - 2.0f * acc0[rr];
        const float d1 = (sx1 + sec) - 2.0f * acc1[rr];
        if (d0 < best0) { best0 = d0; bi0 = code; }  // a lane visits its codes in increasing order
        if (d1 < best1) { best1 = d1; bi1 = code; }
      }
      acc0 = (f16v){0};
      acc1 = (f16v){0};
    }
  }
  // lane halves hold interleaved code rows of the same feature row
  {
    float ov = __shfl_xor(best0, 32, 64); int oi = __shfl_xor(bi0, 32, 64);
    if (ov < best0 || (ov == best0 && oi < bi0)) { best0 = ov; bi0 = oi; }
    ov = __shfl_xor(best1, 32, 64); oi = __shfl_xor(bi1, 32, 64);
    if (ov < best1 || (ov == best1 && oi < bi1)) { best1 = ov; bi1 = oi; }
  }
  if (lane < 32) { cand_v[wave][lane] = best0; cand_i[wave][lane] = bi0; cand_v[wave][32 + lane] = best1; cand_i[wave][32 + lane] = bi1; }
  __syncthreads();
  if (tid < VF_BX) {
    float b = cand_v[0][tid];
    int bi = cand_i[0][tid];
#pragma unroll
    for (int w = 1; w < 4; w++) {
      const float ov = cand_v[w][tid];
      const int oi = cand_i[w][tid];
      if (ov < b || (ov == b && oi < bi)) { b = ov; bi = oi; }
    }
    indices[m0 + tid] = bi;
    if (best_out) best_out[m0 + tid] = b;
  }
}

__global__ __launch_bounds__(256) void vq_gather_k(const float* x, const float* e, const int64_t* idx, float* q, long long M, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* er = e + idx[row] * D;
  for (int c = threadIdx.x & 63; c < D; c += 64) {
    const float xv = x[row * D + c];
    q[row * D + c] = xv + (er[c] - xv);
  }
}

extern "C" int pero_vq_argmin(const float* x, const float* codebook, int64_t* indices, float* best_dist, float* work,
                              int64_t M, int64_t K, int64_t D, void* stream) {
  PERO_REQUIRE(x && codebook && indices && work, "pero_vq_argmin: null pointer");
  PERO_REQUIRE(M > 0 && K > 0 && D > 0 && K < 2147483647LL && D < 2147483647LL, "pero_vq_argmin: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  float* sx = work;
  float* se = work + M;
  hipLaunchKernelGGL(sqnorm_k, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, sx, (long long)M, (int)D);
  hipLaunchKernelGGL(sqnorm_k, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, st, codebook, se, (long long)K, (int)D);
  if (M % VF_BX == 0 && K % VF_BC == 0 && D % VF_BK == 0 && aligned16(x) && aligned16(codebook)) {
    PERO_LDS_ATTR(vq_argmin_fast_k, 2 * VF_STAGE);
    hipLaunchKernelGGL(vq_argmin_fast_k, dim3((unsigned)(M / VF_BX)), dim3(256), 2 * VF_STAGE, st, x, codebook, sx, se, indices, best_dist,
                       (int)K, (int)D);
  } else {
    hipLaunchKernelGGL(vq_argmin_k, dim3((unsigned)((M + VQ_BX - 1) / VQ_BX)), dim3(256), 0, st, x, codebook, sx, se, indices, best_dist,
                       (long long)M, (int)K, (int)D);
  }
  PERO_CHECK_LAUNCH("pero_vq_argmin");
  return PERO_OK;
}
extern "C" int pero_vq_gather(const float* x, const float* codebook, const int64_t* indices, float* quantized, int64_t M, int64_t D,
                              void* stream) {
  PERO_REQUIRE(x && codebook && indices && quantized && M > 0 && D > 0, "pero_vq_gather: bad arguments");
  hipLaunchKernelGGL(vq_gather_k, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, codebook, indices, quantized, (long long)M, (int)D);
  PERO_CHECK_LAUNCH("pero_vq_gather");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// EMA codebook update of the tokenizer in training mode (models/autoencoders.py:225-237), without the (M, K) one-hot
// matrix and its two GEMMs:  counts[k] = #{m : idx[m] = k},  dw[k] = sum_{idx[m] = k} x[m]  (f32 atomics),
// cluster = cluster*decay + (1-decay)*counts;  n = sum(cluster);  cluster = (cluster + eps) / (n + K*eps) * n;
// ema_w = ema_w*decay + (1-decay)*dw;  weight = ema_w / cluster[k].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_ema_scatter_k(const float* x, const int64_t* idx, float* counts, float* dw, long long M, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const long long k = idx[row];
  if ((threadIdx.x & 63) == 0) atomicAdd(counts + k, 1.0f);
  for (int c = threadIdx.x & 63; c < D; c += 64) atomicAdd(dw + k * D + c, x[row * D + c]);
}
// one workgroup: the sum over K runs in a fixed order (deterministic given the counts)
__global__ __launch_bounds__(1024) void vq_ema_cluster_k(float* cluster, const float* counts, int K, float decay, float omd, float eps,
                                                         float keps) {
  __shared__ float red[16];
  __shared__ float total;
  float part = 0.f;
  for (int k = threadIdx.x; k < K; k += 1024) {
    const float c = cluster[k] * decay + omd * counts[k];
    cluster[k] = c;
    part += c;
  }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float n = 0.f;
    for (int i = 0; i < 16; i++) n += red[i];
    total = n;
  }
  __syncthreads();
  const float n = total;
  for (int k = threadIdx.x; k < K; k += 1024) cluster[k] = (cluster[k] + eps) / (n + keps) * n;
}
__global__ __launch_bounds__(256) void vq_ema_weights_k(float* ema_w, float* weight, const float* dw, const float* cluster, long long KD, int D,
                                                        float decay, float omd) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= KD) return;
  const float w = ema_w[i] * decay + omd * dw[i];
  ema_w[i] = w;
  weight[i] = w / cluster[i / D];
}

extern "C" int pero_vq_ema_update(const float* x, const int64_t* indices, float* ema_cluster_size, float* ema_w, float* codebook,
                                  float* work, int64_t M, int64_t K, int64_t D, double decay, double epsilon, void* stream) {
  PERO_REQUIRE(x && indices && ema_cluster_size && ema_w && codebook && work, "pero_vq_ema_update: null pointer");
  PERO_REQUIRE(M > 0 && K > 0 && D > 0 && K < 2147483647LL && D < 2147483647LL, "pero_vq_ema_update: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  float* counts = work;       // K
  float* dw = work + K;       // K * D
  if (hipMemsetAsync(work, 0, sizeof(float) * (size_t)(K + K * D), st) != hipSuccess) {
    pero_set_error("pero_vq_ema_update: memset failed");
    return PERO_E_LAUNCH;
  }
  hipLaunchKernelGGL(vq_ema_scatter_k, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, indices, counts, dw, (long long)M, (int)D);
  // python-scalar semantics of the reference: (1 - decay) and K * epsilon are formed in double, then rounded to f32
  const float fd = (float)decay, omd = (float)(1.0 - decay), feps = (float)epsilon, keps = (float)((double)K * epsilon);
  hipLaunchKernelGGL(vq_ema_cluster_k, dim3(1), dim3(1024), 0, st, ema_cluster_size, counts, (int)K, fd, omd, feps, keps);
  hipLaunchKernelGGL(vq_ema_weights_k, dim3((unsigned)((K * D + 255) / 256)), dim3(256), 0, st, ema_w, codebook, dw, ema_cluster_size,
                     (long long)(K * D), (int)D, fd, omd);
  PERO_CHECK_LAUNCH("pero_vq_ema_update");
  return PERO_OK;
}
