// BatchNorm1d (+ ReLU) over the rows of an (M, d) matrix: the optional `use_bn=True` layers of the joint-embedding MLPHead
// (reference joint_embedding_pretraining/model.py:99-103: Linear -> torch.nn.BatchNorm1d(hidden) -> ReLU on the (N*S, hidden) rows).
// Off the benchmarked path (the reference default is use_bn=False): plain HBM-bound kernels, two-pass statistics, no atomics -
// one workgroup owns a strip of 64 columns and walks ALL rows (4 row lanes x 64 columns, 8 rows in flight per thread), so every
// column sum is added in one fixed order (deterministic).  Training statistics are per call (per data-parallel RANK: what
// torch.nn.BatchNorm1d does under DistributedDataParallel without SyncBatchNorm).
#include "common.hpp"

// mode 0: s0[c] = sum_r x[r][c]
// mode 1: s0[c] = sum_r (x[r][c] - m[c])^2
// mode 2: s0[c] = sum_r g[r][c], s1[c] = sum_r g[r][c] * (x[r][c] - m[c]);  g = dy, zeroed where the layer's output y is <= 0 (relu)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_colstats_k(const T* x, const T* dy, const T* y, const float* m, float* s0, float* s1, long long rows,
                                                     int d, bool relu) {
  __shared__ float red[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f;
  if (c < d) {
    const float mc = MODE ? m[c] : 0.f;
    for (long long r = rl; r < rows; r += 4) {
      const float xv = Elem<T>::ld(x + r * d + c);
      if (MODE == 0) a0 += xv;
      else if (MODE == 1) { const float t = xv - mc; a0 += t * t; }
      else {
        float g = Elem<T>::ld(dy + r * d + c);
        if (relu && !(Elem<T>::ld(y + r * d + c) > 0.f)) g = 0.f;
        a0 += g;
        a1 += g * (xv - mc);
      }
    }
  }
  red[0][rl][threadIdx.x & 63] = a0;
  red[1][rl][threadIdx.x & 63] = a1;
  __syncthreads();
  if (rl == 0 && c < d) {
    const int l = threadIdx.x;
    s0[c] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
    if (MODE == 2) s1[c] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
  }
}

// mean / rstd of the batch from the two column sums; running statistics as torch.nn.BatchNorm1d updates them (momentum m: running =
// (1 - m) running + m batch; the running variance takes the UNBIASED batch variance)
__global__ __launch_bounds__(256) void bn_finish_k(const float* sum, const float* sq, float* mean, float* rstd, float* rmean, float* rvar,
                                                   long long rows, int d, float eps, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  const float var = sq[c] / (float)rows;
  rstd[c] = 1.0f / sqrtf(var + eps);
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (rows > 1 ? sq[c] / (float)(rows - 1) : var);
}
__global__ __launch_bounds__(256) void bn_mean_k(const float* sum, float* mean, long long rows, int d) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) mean[c] = sum[c] / (float)rows;
}
__global__ __launch_bounds__(256) void bn_rstd_eval_k(const float* rvar, float* rstd, int d, float eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < d) rstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

// y = (x - mean) * rstd * w + b, optional ReLU
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_k(const T* x, const float* mean, const float* rstd, const float* w, const float* b, T* y,
                                                  long long n, int d, bool relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % d);
    float v = (Elem<T>::ld(x + i) - mean[c]) * rstd[c] * w[c] + b[c];
    if (relu) v = fmaxf(v, 0.f);
    Elem<T>::st(y + i, v);
  }
}
// dx = w rstd (g - mean_r(g) - xhat mean_r(g xhat)), xhat = (x - mean) rstd; sg = sum_r g, sgx = sum_r g (x - mean)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const T* x, const T* dy, const T* y, const float* mean, const float* rstd, const float* w,
                                                      const float* sg, const float* sgx, T* dx, long long n, long long rows, int d, bool relu) {
  const float inv = 1.0f / (float)rows;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % d);
    float g = Elem<T>::ld(dy + i);
    if (relu && !(Elem<T>::ld(y + i) > 0.f)) g = 0.f;
    const float xh = (Elem<T>::ld(x + i) - mean[c]) * rstd[c];
    const float m2 = sgx[c] * rstd[c] * inv;   // mean_r(g xhat)
    Elem<T>::st(dx + i, w[c] * rstd[c] * (g - sg[c] * inv - xh * m2));
  }
}
// dweight[c] += sum_r g xhat = sgx rstd;  dbias[c] += sum_r g
__global__ __launch_bounds__(256) void bn_bwd_params_k(const float* sg, const float* sgx, const float* rstd, float* dw, float* db, int d) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  if (dw) dw[c] += sgx[c] * rstd[c];
  if (db) db[c] += sg[c];
}

static inline unsigned bn_grid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

template <typename T>
static void bn_fwd_t(const void* x, const float* w, const float* b, float* rmean, float* rvar, void* y, float* mean, float* rstd, float* work,
                     long long rows, int d, float eps, float momentum, bool training, bool relu, hipStream_t st) {
  const dim3 cg((unsigned)((d + 63) / 64)), pg((unsigned)((d + 255) / 256)), blk(256);
  if (training) {
    hipLaunchKernelGGL((bn_colstats_k<T, 0>), cg, blk, 0, st, (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)nullptr, work,
                       (float*)nullptr, rows, d, false);
    hipLaunchKernelGGL(bn_mean_k, pg, blk, 0, st, (const float*)work, mean, rows, d);
    hipLaunchKernelGGL((bn_colstats_k<T, 1>), cg, blk, 0, st, (const T*)x, (const T*)nullptr, (const T*)nullptr, (const float*)mean, work + d,
                       (float*)nullptr, rows, d, false);
    hipLaunchKernelGGL(bn_finish_k, pg, blk, 0, st, (const float*)work, (const float*)(work + d), mean, rstd, rmean, rvar, rows, d, eps, momentum);
  } else {
    hipMemcpyAsync(mean, rmean, (size_t)d * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(bn_rstd_eval_k, pg, blk, 0, st, (const float*)rvar, rstd, d, eps);
  }
  hipLaunchKernelGGL((bn_apply_k<T>), dim3(bn_grid(rows * d)), blk, 0, st, (const T*)x, (const float*)mean, (const float*)rstd, w, b, (T*)y,
                     rows * d, d, relu);
}

extern "C" int pero_bn_fwd(const void* x, const float* weight, const float* bias, float* running_mean, float* running_var, void* y,
                           float* save_mean, float* save_rstd, float* work, int64_t rows, int64_t d, float eps, float momentum, int training,
                           int relu, int dtype, void* stream) {
  PERO_REQUIRE(x && weight && bias && y && save_mean && save_rstd && work, "pero_bn_fwd: null pointer");
  PERO_REQUIRE(rows > 0 && d > 0 && d < (1LL << 30), "pero_bn_fwd: bad shape");
  PERO_REQUIRE(training || (running_mean && running_var), "pero_bn_fwd: evaluation mode needs the running statistics");
  if (dtype == PERO_F32) bn_fwd_t<float>(x, weight, bias, running_mean, running_var, y, save_mean, save_rstd, work, rows, (int)d, eps, momentum, training != 0, relu != 0, (hipStream_t)stream);
  else if (dtype == PERO_BF16) bn_fwd_t<bf16raw>(x, weight, bias, running_mean, running_var, y, save_mean, save_rstd, work, rows, (int)d, eps, momentum, training != 0, relu != 0, (hipStream_t)stream);
  else PERO_REQUIRE(false, "pero_bn_fwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_bn_fwd");
  return PERO_OK;
}

template <typename T>
static void bn_bwd_t(const void* dy, const void* x, const void* y, const float* w, const float* mean, const float* rstd, void* dx, float* dw,
                     float* db, float* work, long long rows, int d, bool relu, hipStream_t st) {
  const dim3 cg((unsigned)((d + 63) / 64)), pg((unsigned)((d + 255) / 256)), blk(256);
  hipLaunchKernelGGL((bn_colstats_k<T, 2>), cg, blk, 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, work, work + d, rows, d, relu);
  hipLaunchKernelGGL((bn_bwd_apply_k<T>), dim3(bn_grid(rows * d)), blk, 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, w,
                     (const float*)work, (const float*)(work + d), (T*)dx, rows * d, rows, d, relu);
  hipLaunchKernelGGL(bn_bwd_params_k, pg, blk, 0, st, (const float*)work, (const float*)(work + d), rstd, dw, db, d);
}

extern "C" int pero_bn_bwd(const void* dy, const void* x, const void* y, const float* weight, const float* save_mean, const float* save_rstd,
                           void* dx, float* dweight, float* dbias, float* work, int64_t rows, int64_t d, int relu, int dtype, void* stream) {
  PERO_REQUIRE(dy && x && weight && save_mean && save_rstd && dx && work && (!relu || y), "pero_bn_bwd: null pointer");
  PERO_REQUIRE(rows > 0 && d > 0 && d < (1LL << 30), "pero_bn_bwd: bad shape");
  if (dtype == PERO_F32) bn_bwd_t<float>(dy, x, y, weight, save_mean, save_rstd, dx, dweight, dbias, work, rows, (int)d, relu != 0, (hipStream_t)stream);
  else if (dtype == PERO_BF16) bn_bwd_t<bf16raw>(dy, x, y, weight, save_mean, save_rstd, dx, dweight, dbias, work, rows, (int)d, relu != 0, (hipStream_t)stream);
  else PERO_REQUIRE(false, "pero_bn_bwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_bn_bwd");
  return PERO_OK;
}
