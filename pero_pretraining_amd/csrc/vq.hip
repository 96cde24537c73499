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
  hipLaunchKernelGGL(vq_argmin_k, dim3((unsigned)((M + VQ_BX - 1) / VQ_BX)), dim3(256), 0, st, x, codebook, sx, se, indices, best_dist,
                     (long long)M, (int)K, (int)D);
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
