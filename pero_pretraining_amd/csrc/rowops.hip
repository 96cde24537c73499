// Row-wise HBM-bound kernels: LayerNorm (+positional encoding) fwd/bwd, softmax fwd/bwd, masked
// cross entropy, column sums.  One 64-lane wave per row (shuffle reductions, no LDS for the row
// statistics), 16-byte global accesses, f32 statistics whatever the storage dtype.
#include "gemm_common.hpp"

// ---------------------------------------------------------------------------------------------
// LayerNorm forward.  d % 8 == 0, d <= 512 * NCH.  4 waves per block, one row per wave.
// ---------------------------------------------------------------------------------------------
template <typename T, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_k(const T* x, const float* gamma, const float* beta, const float* pe,
                                                       const int64_t* offsets, T* y, float* mean, float* rstd,
                                                       long long rows, int d, long long S, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
      load8<T>(x + row * d + col, v[c]);
#pragma unroll
      for (int e = 0; e < 8; e++) s += v[c][e];
    }
  }
  const float mu = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
#pragma unroll
      for (int e = 0; e < 8; e++) { const float t = v[c][e] - mu; q += t * t; }
    }
  }
  const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  const float* perow = nullptr;
  if (pe) {
    const long long line = row / S, pos = row % S;
    perow = pe + ((offsets ? offsets[line] : 0) + pos) * d;
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int col = (c * 64 + lane) * 8;
    if (col < d) {
      float g[8], b[8], o[8];
      load8<float>(gamma + col, g);
      load8<float>(beta + col, b);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (v[c][e] - mu) * rs * g[e] + b[e];
      if (perow) {
        float pp[8];
        load8<float>(perow + col, pp);
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] += pp[e];
      }
      store8<T>(y + row * d + col, o);
    }
  }
}

// bf16, d <= 512 (one 8-column group per lane), no positional table: the arithmetic of layernorm_fwd_k<bf16raw, 1>, FOUR rows per wave
// with their loads issued together (the one-row-per-wave form is 65 536 workgroups of four short-lived waves at M = 262 144:
// 4.9 TB/s; four rows in flight per wave and a quarter of the waves: see DESIGN section 8.1).
__global__ __launch_bounds__(256) void layernorm_fwd4_k(const bf16raw* x, const float* gamma, const float* beta, bf16raw* y, float* mean,
                                                        float* rstd, long long rows, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (row0 >= rows) return;
  const int col = lane * 8;
  const bool act = col < d;
  uint4 raw[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    raw[u] = make_uint4(0, 0, 0, 0);
    if (act && row0 + u < rows) raw[u] = *(const uint4*)(x + (row0 + u) * d + col);
  }
  float g[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { g[e] = 0.f; b[e] = 0.f; }
  if (act) { load8<float>(gamma + col, g); load8<float>(beta + col, b); }
  float v[4][8], s[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const unsigned w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
    s[u] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      v[u][e] = (e & 1) ? __uint_as_float(w[e >> 1] & 0xffff0000u) : __uint_as_float(w[e >> 1] << 16);
      if (act) s[u] += v[u][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int u = 0; u < 4; u++) s[u] += __shfl_xor(s[u], o, 64);
  float mu[4], q[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    mu[u] = s[u] / (float)d;
    q[u] = 0.f;
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; e++) { const float t = v[u][e] - mu[u]; q[u] += t * t; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] += __shfl_xor(q[u], o, 64);
#pragma unroll
  for (int u = 0; u < 4; u++) {
    if (row0 + u >= rows) break;
    const float rs = 1.0f / sqrtf(q[u] / (float)d + eps);
    if (lane == 0) { mean[row0 + u] = mu[u]; rstd[row0 + u] = rs; }
    if (act) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (v[u][e] - mu[u]) * rs * g[e] + b[e];
      store8<bf16raw>(y + (row0 + u) * d + col, o);
    }
  }
}

// LayerNorm backward: dx per row; dgamma / dbeta / column sums of dx accumulated per lane over the rows
// a wave visits, combined across the block's 4 waves through LDS, then one f32 atomic per column.
// FROM_OUT (round 4, "memory-efficient" LayerNorm): `x` is the layer's OUTPUT t = xhat * gamma + beta and `mean` is the BETA vector (per
// column, not per row): xhat is recovered as (t - beta) / gamma, so the forward pass need not keep its input rows for the backward (in bf16
// mode the stored t carries the same 2^-9 relative rounding the stored input rows had; columns with gamma == 0 lose their gamma gradient).
template <typename T, int NCH, bool FROM_OUT>
__global__ __launch_bounds__(256) void layernorm_bwd_k(const T* dy, const T* x, const float* mean, const float* rstd,
                                                       const float* gamma, T* dx, float* work, const float* dxsum,
                                                       long long rows, int d) {
  __shared__ float red[4][NCH * 512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NCH][8], ab[NCH][8], ax[NCH][8], g[NCH][8], bt[NCH][8], ig[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const int col = (c * 64 + lane) * 8;
#pragma unroll
    for (int e = 0; e < 8; e++) { ag[c][e] = 0.f; ab[c][e] = 0.f; ax[c][e] = 0.f; g[c][e] = 0.f; bt[c][e] = 0.f; ig[c][e] = 0.f; }
    if (col < d) load8<float>(gamma + col, g[c]);
    if (FROM_OUT && col < d) {
      load8<float>(mean + col, bt[c]);
#pragma unroll
      for (int e = 0; e < 8; e++) ig[c][e] = g[c][e] != 0.f ? 1.0f / g[c][e] : 0.f;
    }
  }
  // two rows per iteration (independent loads and reduction chains in flight: the kernel is latency-, not bandwidth-bound
  // with one row per wave at a time)
  const long long stride = (long long)gridDim.x * 4;
  for (long long row0 = (long long)blockIdx.x * 4 + wave; row0 < rows; row0 += 2 * stride) {
    const long long rw[2] = {row0, row0 + stride};
    const bool has[2] = {true, row0 + stride < rows};
    float mu[2], rs[2], xh[2][NCH][8], gy[2][NCH][8], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!has[u]) continue;
      mu[u] = FROM_OUT ? 0.f : mean[rw[u]]; rs[u] = rstd[rw[u]];
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < d) {
          float xv[8], dv[8];
          load8<T>(x + rw[u] * d + col, xv);
          load8<T>(dy + rw[u] * d + col, dv);
#pragma unroll
          for (int e = 0; e < 8; e++) {
            xh[u][c][e] = FROM_OUT ? (xv[e] - bt[c][e]) * ig[c][e] : (xv[e] - mu[u]) * rs[u];
            gy[u][c][e] = dv[e] * g[c][e];
            s1[u] += gy[u][c][e];
            s2[u] += gy[u][c][e] * xh[u][c][e];
            ag[c][e] += dv[e] * xh[u][c][e];
            ab[c][e] += dv[e];
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {  // the four reductions of the two rows interleaved
      s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
      s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!has[u]) continue;
      const float c1 = s1[u] / (float)d, c2 = s2[u] / (float)d;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        const int col = (c * 64 + lane) * 8;
        if (col < d) {
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; e++) { o[e] = rs[u] * (gy[u][c][e] - c1 - xh[u][c][e] * c2); ax[c][e] += o[e]; }
          store8<T>(dx + rw[u] * d + col, o);
        }
      }
    }
  }
  // per-block partial sums -> workspace [3][gridDim.x][d]; a second kernel adds them to the gradients
  // (no contended atomics: 512 blocks adding into the same d addresses ran at ~1/14 of the atomic rate)
#pragma unroll
  for (int which = 0; which < 3; which++) {
    if (which == 2 && !dxsum) break;
#pragma unroll
    for (int c = 0; c < NCH; c++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int col = (c * 64 + lane) * 8 + e;
        red[wave][col] = which == 0 ? ag[c][e] : (which == 1 ? ab[c][e] : ax[c][e]);
      }
    __syncthreads();
    float* dst = work + ((long long)which * gridDim.x + blockIdx.x) * d;
    for (int col = threadIdx.x; col < d; col += 256)
      dst[col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    __syncthreads();
  }
}
// bf16, d <= 512 (one 8-column group per lane): the same arithmetic as layernorm_bwd_k<bf16raw, 1>, software-pipelined - the raw
// rows of the NEXT pair (x, dy: one 16-byte load each, mean, rstd) are requested before the current pair is reduced, so a
// wave always has a pair of rows in flight (the plain loop issued its loads, waited, reduced, stored: latency-bound at 8
// waves per CU).
template <bool FROM_OUT>
__global__ __launch_bounds__(256) void layernorm_bwd_pf_k(const bf16raw* dy, const bf16raw* x, const float* mean, const float* rstd,
                                                          const float* gamma, bf16raw* dx, float* work, const float* dxsum,
                                                          long long rows, int d) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane * 8;
  const bool act = col < d;
  float ag[8], ab[8], ax[8], g[8], bt[8], ig[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { ag[e] = 0.f; ab[e] = 0.f; ax[e] = 0.f; g[e] = 0.f; bt[e] = 0.f; ig[e] = 0.f; }
  if (act) load8<float>(gamma + col, g);
  if (FROM_OUT && act) {   // `mean` is the beta vector, `x` the LayerNorm output (see layernorm_bwd_k)
    load8<float>(mean + col, bt);
#pragma unroll
    for (int e = 0; e < 8; e++) ig[e] = g[e] != 0.f ? 1.0f / g[e] : 0.f;
  }
  const long long stride = (long long)gridDim.x * 4;
  long long row0 = (long long)blockIdx.x * 4 + wave;
  uint4 nx[2], nd[2];
  float nmu[2] = {0.f, 0.f}, nrs[2] = {0.f, 0.f};
  nx[0] = nx[1] = nd[0] = nd[1] = make_uint4(0, 0, 0, 0);
  auto fetch = [&](long long r0) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const long long r = r0 + u * stride;
      if (r < rows) {
        nmu[u] = FROM_OUT ? 0.f : mean[r]; nrs[u] = rstd[r];
        if (act) { nx[u] = *(const uint4*)(x + r * d + col); nd[u] = *(const uint4*)(dy + r * d + col); }
      }
    }
  };
  if (row0 < rows) fetch(row0);
  for (; row0 < rows; row0 += 2 * stride) {
    const bool has[2] = {true, row0 + stride < rows};
    uint4 cx[2] = {nx[0], nx[1]}, cd[2] = {nd[0], nd[1]};
    const float mu[2] = {nmu[0], nmu[1]}, rs[2] = {nrs[0], nrs[1]};
    if (row0 + 2 * stride < rows) fetch(row0 + 2 * stride);
    float xh[2][8], gy[2][8], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!has[u] || !act) continue;
      const unsigned wx[4] = {cx[u].x, cx[u].y, cx[u].z, cx[u].w}, wd[4] = {cd[u].x, cd[u].y, cd[u].z, cd[u].w};
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float xv = (e & 1) ? __uint_as_float(wx[e >> 1] & 0xffff0000u) : __uint_as_float(wx[e >> 1] << 16);
        const float dv = (e & 1) ? __uint_as_float(wd[e >> 1] & 0xffff0000u) : __uint_as_float(wd[e >> 1] << 16);
        xh[u][e] = FROM_OUT ? (xv - bt[e]) * ig[e] : (xv - mu[u]) * rs[u];
        gy[u][e] = dv * g[e];
        s1[u] += gy[u][e];
        s2[u] += gy[u][e] * xh[u][e];
        ag[e] += dv * xh[u][e];
        ab[e] += dv;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
      s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (!has[u] || !act) continue;
      const float c1 = s1[u] / (float)d, c2 = s2[u] / (float)d;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { o[e] = rs[u] * (gy[u][e] - c1 - xh[u][e] * c2); ax[e] += o[e]; }
      store8<bf16raw>(dx + (row0 + u * stride) * d + col, o);
    }
  }
#pragma unroll
  for (int which = 0; which < 3; which++) {
    if (which == 2 && !dxsum) break;
#pragma unroll
    for (int e = 0; e < 8; e++) red[wave][col + e] = which == 0 ? ag[e] : (which == 1 ? ab[e] : ax[e]);
    __syncthreads();
    float* dst = work + ((long long)which * gridDim.x + blockIdx.x) * d;
    for (int c = threadIdx.x; c < d; c += 256) dst[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_k(const float* work, float* dgamma, float* dbeta, float* dxsum,
                                                              int nblocks, int d) {
  // block = 64 columns x 4 slices of the partial-sum rows; 8 independent loads in flight per thread
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int which = blockIdx.y;
  float* dst = which == 0 ? dgamma : (which == 1 ? dbeta : dxsum);
  if (!dst) return;
  float s = 0.f;
  if (col < d) {
    const float* w = work + (long long)which * nblocks * d + col;
    const int per = (nblocks + 3) / 4, b0 = part * per, b1 = (b0 + per) < nblocks ? (b0 + per) : nblocks;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] += w[(long long)(b + u) * d];
    }
    for (; b < b1; b++) acc[0] += w[(long long)b * d];
    s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  red[part][c] = s;
  __syncthreads();
  if (part == 0 && col < d) dst[col] += (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

void pero_ln_bwd_reduce_launch(const float* work, float* dgamma, float* dbeta, float* dxsum, int nblocks, int d, hipStream_t st) {
  hipLaunchKernelGGL(layernorm_bwd_reduce_k, dim3((unsigned)((d + 63) / 64), 3), dim3(256), 0, st, work, dgamma, dbeta, dxsum, nblocks, d);
}

template <typename T>
static int ln_dispatch_fwd(const void* x, const float* gamma, const float* beta, const float* pe, const int64_t* offsets,
                           void* y, float* mean, float* rstd, int64_t rows, int64_t d, int64_t S, float eps, hipStream_t st) {
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const int nch = (int)((d + 511) / 512);
#define LN_F(N_) hipLaunchKernelGGL((layernorm_fwd_k<T, N_>), grid, block, 0, st, (const T*)x, gamma, beta, pe, offsets, (T*)y, mean, rstd, (long long)rows, (int)d, (long long)S, eps)
  if (nch == 1 && sizeof(T) == 2 && !pe && rows >= 4096)
    hipLaunchKernelGGL(layernorm_fwd4_k, dim3((unsigned)((rows + 15) / 16)), block, 0, st, (const bf16raw*)x, gamma, beta, (bf16raw*)y, mean, rstd,
                       (long long)rows, (int)d, eps);
  else if (nch == 1) LN_F(1); else if (nch == 2) LN_F(2); else if (nch <= 4) LN_F(4); else LN_F(8);
#undef LN_F
  return 0;
}
template <typename T, bool FROM_OUT>
static int ln_dispatch_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                           float* dgamma, float* dbeta, float* dxsum, float* work, int64_t rows, int64_t d, hipStream_t st) {
  long long blocks = (rows + 3) / 4;
  if (blocks > PERO_LN_BWD_BLOCKS) blocks = PERO_LN_BWD_BLOCKS;
  dim3 grid((unsigned)blocks), block(256);
  const int nch = (int)((d + 511) / 512);
#define LN_B(N_) hipLaunchKernelGGL((layernorm_bwd_k<T, N_, FROM_OUT>), grid, block, 0, st, (const T*)dy, (const T*)x, mean, rstd, gamma, (T*)dx, work, dxsum, (long long)rows, (int)d)
  if (nch == 1 && sizeof(T) == 2)
    hipLaunchKernelGGL(layernorm_bwd_pf_k<FROM_OUT>, grid, block, 0, st, (const bf16raw*)dy, (const bf16raw*)x, mean, rstd, gamma, (bf16raw*)dx, work, dxsum,
                       (long long)rows, (int)d);
  else if (nch == 1) LN_B(1); else if (nch == 2) LN_B(2); else LN_B(4);
#undef LN_B
  hipLaunchKernelGGL(layernorm_bwd_reduce_k, dim3((unsigned)((d + 63) / 64), 3), dim3(256), 0, st, work, dgamma, dbeta, dxsum, (int)blocks, (int)d);
  return 0;
}

extern "C" int pero_layernorm_fwd(const void* x, const float* gamma, const float* beta, const float* pe, const int64_t* offsets,
                                  void* y, float* mean, float* rstd, int64_t rows, int64_t d, int64_t S, float eps, int dtype,
                                  void* stream) {
  PERO_REQUIRE(x && gamma && beta && y && mean && rstd, "pero_layernorm_fwd: null pointer");
  PERO_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= 4096, "pero_layernorm_fwd: need d %% 8 == 0 and d <= 4096 (d=%lld)", (long long)d);
  PERO_REQUIRE(!pe || S > 0, "pero_layernorm_fwd: S must be > 0 with a positional table");
  PERO_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && (!pe || aligned16(pe)), "pero_layernorm_fwd: 16-byte alignment");
  if (dtype == PERO_F32) ln_dispatch_fwd<float>(x, gamma, beta, pe, offsets, y, mean, rstd, rows, d, S, eps, (hipStream_t)stream);
  else if (dtype == PERO_BF16) ln_dispatch_fwd<bf16raw>(x, gamma, beta, pe, offsets, y, mean, rstd, rows, d, S, eps, (hipStream_t)stream);
  else PERO_REQUIRE(false, "pero_layernorm_fwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_layernorm_fwd");
  return PERO_OK;
}

extern "C" int pero_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                                  void* dx, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t rows, int64_t d,
                                  int dtype, void* stream) {
  PERO_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && work, "pero_layernorm_bwd: null pointer");
  PERO_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= 2048, "pero_layernorm_bwd: need d %% 8 == 0 and d <= 2048 (d=%lld)", (long long)d);
  PERO_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(gamma), "pero_layernorm_bwd: 16-byte alignment");
  if (dtype == PERO_F32) ln_dispatch_bwd<float, false>(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dxsum, work, rows, d, (hipStream_t)stream);
  else if (dtype == PERO_BF16) ln_dispatch_bwd<bf16raw, false>(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, dxsum, work, rows, d, (hipStream_t)stream);
  else PERO_REQUIRE(false, "pero_layernorm_bwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_layernorm_bwd");
  return PERO_OK;
}

extern "C" int pero_layernorm_bwd_out(const void* dy, const void* t, const float* rstd, const float* gamma, const float* beta,
                                      void* dx, float* dgamma, float* dbeta, float* dxsum, float* work, int64_t rows, int64_t d,
                                      int dtype, void* stream) {
  PERO_REQUIRE(dy && t && rstd && gamma && beta && dx && dgamma && dbeta && work, "pero_layernorm_bwd_out: null pointer");
  PERO_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= 2048, "pero_layernorm_bwd_out: need d %% 8 == 0 and d <= 2048 (d=%lld)", (long long)d);
  PERO_REQUIRE(aligned16(dy) && aligned16(t) && aligned16(dx) && aligned16(gamma) && aligned16(beta), "pero_layernorm_bwd_out: 16-byte alignment");
  if (dtype == PERO_F32) ln_dispatch_bwd<float, true>(dy, t, beta, rstd, gamma, dx, dgamma, dbeta, dxsum, work, rows, d, (hipStream_t)stream);
  else if (dtype == PERO_BF16) ln_dispatch_bwd<bf16raw, true>(dy, t, beta, rstd, gamma, dx, dgamma, dbeta, dxsum, work, rows, d, (hipStream_t)stream);
  else PERO_REQUIRE(false, "pero_layernorm_bwd_out: bad dtype");
  PERO_CHECK_LAUNCH("pero_layernorm_bwd_out");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// softmax (attention probabilities): scores f32 -> p (T); one wave per row
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_k(const float* s, T* p, long long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* sr = s + row * cols;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, sr[c] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += expf(sr[c] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) Elem<T>::st(p + row * cols + c, expf(sr[c] * scale - mx) * inv);
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_k(const T* p, const float* dp, T* ds, long long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += Elem<T>::ld(p + row * cols + c) * dp[row * cols + c];
  dot = wave_sum(dot);
  for (int c = lane; c < cols; c += 64)
    Elem<T>::st(ds + row * cols + c, scale * Elem<T>::ld(p + row * cols + c) * (dp[row * cols + c] - dot));
}
extern "C" int pero_softmax_fwd(const float* s, void* p, int64_t rows, int64_t cols, float scale, int dtype, void* stream) {
  PERO_REQUIRE(s && p && rows > 0 && cols > 0, "pero_softmax_fwd: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == PERO_F32) hipLaunchKernelGGL((softmax_fwd_k<float>), grid, block, 0, (hipStream_t)stream, s, (float*)p, (long long)rows, (int)cols, scale);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((softmax_fwd_k<bf16raw>), grid, block, 0, (hipStream_t)stream, s, (bf16raw*)p, (long long)rows, (int)cols, scale);
  else PERO_REQUIRE(false, "pero_softmax_fwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_softmax_fwd");
  return PERO_OK;
}
extern "C" int pero_softmax_bwd(const void* p, const float* dp, void* ds, int64_t rows, int64_t cols, float scale, int dtype, void* stream) {
  PERO_REQUIRE(p && dp && ds && rows > 0 && cols > 0, "pero_softmax_bwd: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (dtype == PERO_F32) hipLaunchKernelGGL((softmax_bwd_k<float>), grid, block, 0, (hipStream_t)stream, (const float*)p, dp, (float*)ds, (long long)rows, (int)cols, scale);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((softmax_bwd_k<bf16raw>), grid, block, 0, (hipStream_t)stream, (const bf16raw*)p, dp, (bf16raw*)ds, (long long)rows, (int)cols, scale);
  else PERO_REQUIRE(false, "pero_softmax_bwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_softmax_bwd");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// masked cross entropy.  work[0..rows) = per-row loss, work[rows+0] = n_masked, work[rows+1] = n_unmasked,
// work[rows+8 .. 2*rows+8) = per-row logsumexp (kept for the backward kernel)
// ---------------------------------------------------------------------------------------------
// forward: per-row loss -> work[row], per-row logsumexp -> work[rows + 8 + row]
template <typename T>
__global__ __launch_bounds__(256) void ce_rows_k(const T* logits, const int64_t* labels, const int64_t* mask, float uw,
                                                 float* work, long long rows, int V, const int64_t* index) {
  __shared__ float red[4];
  const long long row = index ? index[blockIdx.x] : blockIdx.x;   // (compact form: the caller lists the rows that take part)
  const int tid = threadIdx.x;
  const long long lab = labels[row];
  const long long mk = mask[row];
  const bool active = (mk == 1) || (mk == 0 && lab >= 0 && uw >= 0.f);
  if (!active) {
    if (tid == 0) { work[row] = 0.f; work[rows + 8 + row] = 0.f; }
    return;
  }
  if (lab < 0 || lab >= V) {
    // a participating row whose label is outside [0, V) (masked_pretraining/model.py:82: F.cross_entropy raises there): the
    // loss becomes NaN instead of reading logits out of bounds - loud, no device fault
    if (tid == 0) { work[row] = __builtin_nanf(""); work[rows + 8 + row] = 0.f; }
    return;
  }
  const T* lr = logits + row * V;
  if (V == 4096 && (((uintptr_t)lr) & 15) == 0) {
    // the whole row in registers (2 x 8 values per thread): one pass over memory, one barrier pair
    float v[2][8];
    load8<T>(lr + tid * 8, v[0]);
    load8<T>(lr + 2048 + tid * 8, v[1]);
    float mx = v[0][0];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) mx = fmaxf(mx, v[i][e]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 8; e++) sum += expf(v[i][e] - mx);
    sum = wave_sum(sum);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float lse = logf(sum) + mx;
    if (tid == 0) { work[row] = lse - Elem<T>::ld(lr + lab); work[rows + 8 + row] = lse; }
    return;
  }
  float mx = -INFINITY;
  for (int c = tid; c < V; c += 256) mx = fmaxf(mx, Elem<T>::ld(lr + c));
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < V; c += 256) sum += expf(Elem<T>::ld(lr + c) - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float lse = logf(sum) + mx;
  if (tid == 0) { work[row] = lse - Elem<T>::ld(lr + lab); work[rows + 8 + row] = lse; }
}
// backward: dlogits[row] = (softmax(logits[row]) - onehot(label)) * weight(row) * dloss
// With `index` (the compact form): block i writes row i of a (gridDim.x, V) matrix - the gradient row of logits row index[i] for
// i < n_idx, a zero row behind (the padding up to whole GEMM tiles).  The head's two backward products then run over the selected
// rows only: every other row of the dense dlogits is an exact zero.
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_k(const T* logits, const int64_t* labels, const int64_t* mask, float uw,
                                                const float* dloss, const float* work, T* dlogits, long long rows, int V,
                                                const int64_t* index, long long n_idx) {
  const int tid = threadIdx.x;
  long long row = blockIdx.x;
  if (index) {
    const bool pad = (long long)blockIdx.x >= n_idx;
    const long long src = pad ? 0 : index[blockIdx.x];
    const long long lab0 = pad ? 0 : labels[src], mk0 = pad ? -1 : mask[src];
    if (pad || !((mk0 == 1) || (mk0 == 0 && lab0 >= 0 && uw >= 0.f))) {   // padding row / a listed row that takes no part: zeros
      T* o = dlogits + (long long)blockIdx.x * V;
      for (int c = tid; c < V; c += 256) Elem<T>::st(o + c, 0.f);
      return;
    }
    dlogits += ((long long)blockIdx.x - src) * V;   // row `src` of the arithmetic below lands in output row blockIdx.x
    row = src;
  }
  {
  const long long lab = labels[row];
  const long long mk = mask[row];
  float w = 0.f;
  if (mk == 1) w = 1.0f / work[rows];
  else if (mk == 0 && lab >= 0 && uw >= 0.f) w = uw / work[rows + 1];
  const bool active = (mk == 1) || (mk == 0 && lab >= 0 && uw >= 0.f);
  const bool vec = (V & 7) == 0 && ((((uintptr_t)(dlogits + row * V)) | ((uintptr_t)(logits + row * V))) & 15) == 0;
  if (!active) return;  // the buffer was zero-filled linearly before this launch: per-row zero stores from 65536 small
                        // blocks (or a grid-stride loop) reached only 2.7 TB/s

  if (dloss) w *= dloss[0];
  if (lab < 0 || lab >= V) w = __builtin_nanf("");  // out-of-range label on a participating row: NaN gradient row (see ce_rows_k)
  const float lse = work[rows + 8 + row];
  const T* lr = logits + row * V;
  if (vec) {
    for (int c = tid * 8; c < V; c += 2048) {
      float v[8];
      load8<T>(lr + c, v);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float g = expf(v[e] - lse);
        if (c + e == lab) g -= 1.0f;
        v[e] = g * w;
      }
      store8<T>(dlogits + row * V + c, v);
    }
    return;
  }
  for (int c = tid; c < V; c += 256) {
    float g = expf(Elem<T>::ld(lr + c) - lse);
    if (c == lab) g -= 1.0f;
    Elem<T>::st(dlogits + row * V + c, g * w);
  }
  }
}
// counts + deterministic loss sums: work[rows] = n_masked, work[rows+1] = n_unmasked (read by the backward kernel),
// loss = sum_m/n_m (+ uw * sum_u/n_u).  Two levels, both in a fixed order (deterministic): CE_PARTS blocks sum a contiguous slice of the rows each into
// work[rows + 8 + rows + 4 * part ..], the last block to finish (ticket in work[rows + 2]) adds the partials in index order.
#define CE_PARTS 64
__global__ __launch_bounds__(256) void ce_final_k(const int64_t* labels, const int64_t* mask, float uw, float* work,
                                                  float* loss, long long rows) {
  __shared__ float sm[4][4];
  __shared__ int last;
  float* partial = work + 2 * rows + 8;            // CE_PARTS x 4 floats (the launcher's workspace has room: see header)
  unsigned* ticket = (unsigned*)(work + rows + 2);
  const long long per = (rows + CE_PARTS - 1) / CE_PARTS;
  const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float a = 0.f, b = 0.f, na = 0.f, nb = 0.f;
  for (long long r = r0 + threadIdx.x; r < r1; r += 256) {
    const long long mk = mask[r], lb = labels[r];
    const float w = work[r];
    if (mk == 1) { a += w; na += 1.f; }
    else if (mk == 0 && lb >= 0) { b += w; nb += 1.f; }
  }
  a = wave_sum(a); b = wave_sum(b); na = wave_sum(na); nb = wave_sum(nb);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[0][wv] = a; sm[1][wv] = b; sm[2][wv] = na; sm[3][wv] = nb; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; q++) partial[4 * blockIdx.x + q] = (sm[q][0] + sm[q][1]) + (sm[q][2] + sm[q][3]);
    __threadfence();
    last = atomicAdd(ticket, 1u) == CE_PARTS - 1;
  }
  __syncthreads();
  __shared__ float pl[4 * CE_PARTS];
  if (last) {
    __threadfence();
    pl[threadIdx.x] = ((volatile float*)partial)[threadIdx.x];  // 256 = 4 * CE_PARTS values, one per thread
  }
  __syncthreads();
  if (last && threadIdx.x == 0) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < CE_PARTS; i++)
      for (int q = 0; q < 4; q++) t[q] += pl[4 * i + q];
    work[rows] = t[2];
    work[rows + 1] = t[3];
    float l = t[0] / t[2];
    if (uw >= 0.f) l += uw * (t[1] / t[3]);
    loss[0] = l;
    *ticket = 0u;  // ready for the next launch on this workspace
  }
}
extern "C" int pero_masked_ce_fwd(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                                  float* loss_out, float* work, int64_t rows, int64_t V, int dtype, void* stream) {
  PERO_REQUIRE(logits && labels && mask && loss_out && work, "pero_masked_ce_fwd: null pointer");
  PERO_REQUIRE(rows > 0 && V > 0 && rows < 16777216, "pero_masked_ce_fwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PERO_F32) hipLaunchKernelGGL((ce_rows_k<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, labels, mask, unmasked_weight, work, (long long)rows, (int)V, (const int64_t*)nullptr);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((ce_rows_k<bf16raw>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16raw*)logits, labels, mask, unmasked_weight, work, (long long)rows, (int)V, (const int64_t*)nullptr);
  else PERO_REQUIRE(false, "pero_masked_ce_fwd: bad dtype");
  hipMemsetAsync(work + rows + 2, 0, sizeof(float), st);  // the reduction's ticket
  hipLaunchKernelGGL(ce_final_k, dim3(CE_PARTS), dim3(256), 0, st, labels, mask, unmasked_weight, work, loss_out, (long long)rows);
  PERO_CHECK_LAUNCH("pero_masked_ce_fwd");
  return PERO_OK;
}
extern "C" int pero_masked_ce_fwd_rows(const void* logits, const int64_t* labels, const int64_t* mask, const int64_t* index, int64_t n_idx,
                                       float* loss_out, float* work, int64_t rows, int64_t V, int dtype, void* stream) {
  PERO_REQUIRE(logits && labels && mask && index && loss_out && work, "pero_masked_ce_fwd_rows: null pointer");
  PERO_REQUIRE(rows > 0 && V > 0 && rows < 16777216 && n_idx > 0 && n_idx <= rows, "pero_masked_ce_fwd_rows: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  // ce_final_k reads work[r] of every row with mask == 1 (all listed) and, into a sum this mode never uses, of the rows with mask == 0
  if (dtype == PERO_F32) hipLaunchKernelGGL((ce_rows_k<float>), dim3((unsigned)n_idx), dim3(256), 0, st, (const float*)logits, labels, mask, -1.0f, work, (long long)rows, (int)V, index);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((ce_rows_k<bf16raw>), dim3((unsigned)n_idx), dim3(256), 0, st, (const bf16raw*)logits, labels, mask, -1.0f, work, (long long)rows, (int)V, index);
  else PERO_REQUIRE(false, "pero_masked_ce_fwd_rows: bad dtype");
  hipMemsetAsync(work + rows + 2, 0, sizeof(float), st);  // the reduction's ticket
  hipLaunchKernelGGL(ce_final_k, dim3(CE_PARTS), dim3(256), 0, st, labels, mask, -1.0f, work, loss_out, (long long)rows);
  PERO_CHECK_LAUNCH("pero_masked_ce_fwd_rows");
  return PERO_OK;
}
extern "C" int pero_masked_ce_bwd(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                                  const float* dloss, const float* work, void* dlogits, int64_t rows, int64_t V, int dtype,
                                  void* stream) {
  PERO_REQUIRE(logits && labels && mask && work && dlogits, "pero_masked_ce_bwd: null pointer");
  PERO_REQUIRE(rows > 0 && V > 0, "pero_masked_ce_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dlogits, 0, (size_t)rows * (size_t)V * (dtype == PERO_F32 ? 4 : 2), st) != hipSuccess) {
    pero_set_error("pero_masked_ce_bwd: zero fill failed");
    return PERO_E_LAUNCH;
  }
  if (dtype == PERO_F32) hipLaunchKernelGGL((ce_bwd_k<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, labels, mask, unmasked_weight, dloss, work, (float*)dlogits, (long long)rows, (int)V, (const int64_t*)nullptr, 0LL);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((ce_bwd_k<bf16raw>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16raw*)logits, labels, mask, unmasked_weight, dloss, work, (bf16raw*)dlogits, (long long)rows, (int)V, (const int64_t*)nullptr, 0LL);
  else PERO_REQUIRE(false, "pero_masked_ce_bwd: bad dtype");
  PERO_CHECK_LAUNCH("pero_masked_ce_bwd");
  return PERO_OK;
}
extern "C" int pero_masked_ce_bwd_rows(const void* logits, const int64_t* labels, const int64_t* mask, float unmasked_weight,
                                       const float* dloss, const float* work, const int64_t* index, int64_t n_idx, int64_t n_rows_out,
                                       void* dlogits_rows, int64_t rows, int64_t V, int dtype, void* stream) {
  PERO_REQUIRE(logits && labels && mask && work && dlogits_rows && (index || n_idx == 0), "pero_masked_ce_bwd_rows: null pointer");
  PERO_REQUIRE(rows > 0 && V > 0 && n_idx >= 0 && n_idx <= n_rows_out && n_rows_out > 0 && n_rows_out < 2147483647LL, "pero_masked_ce_bwd_rows: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  // (index == null with n_idx == 0: every output row is padding; the kernel needs a non-null pointer to take the compact form)
  const int64_t* ix = index ? index : labels;
  if (dtype == PERO_F32) hipLaunchKernelGGL((ce_bwd_k<float>), dim3((unsigned)n_rows_out), dim3(256), 0, st, (const float*)logits, labels, mask, unmasked_weight, dloss, work, (float*)dlogits_rows, (long long)rows, (int)V, ix, (long long)n_idx);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((ce_bwd_k<bf16raw>), dim3((unsigned)n_rows_out), dim3(256), 0, st, (const bf16raw*)logits, labels, mask, unmasked_weight, dloss, work, (bf16raw*)dlogits_rows, (long long)rows, (int)V, ix, (long long)n_idx);
  else PERO_REQUIRE(false, "pero_masked_ce_bwd_rows: bad dtype");
  PERO_CHECK_LAUNCH("pero_masked_ce_bwd_rows");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums: out[n] += sum_m x[m][n]
// ---------------------------------------------------------------------------------------------
// fast path: cols % 8 == 0, 16-byte aligned rows.  Block = 32 column groups (8 cols each) x 8 row lanes over a
// 128-row slab; LDS combine of the 8 row lanes; one atomic per column per slab.
template <typename T>
__global__ __launch_bounds__(256) void colsum8_k(const T* x, float* out, long long rows, long long cols, long long ld) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long long col = ((long long)blockIdx.x * 32 + cg) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < cols) {
    const long long r0 = (long long)blockIdx.y * 128;
    const long long r1 = r0 + 128 < rows ? r0 + 128 : rows;
    for (long long r = r0 + rl; r < r1; r += 8) {
      float v[8];
      load8<T>(x + r * ld + col, v);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  const long long gc = (long long)blockIdx.x * 256 + c;
  if (gc < cols) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += red[k][c];
    atomicAdd(out + gc, s);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* x, float* out, long long rows, long long cols, long long ld) {
  const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  float s = 0.f;
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) s += Elem<T>::ld(x + r * ld + col);
  atomicAdd(out + col, s);
}
extern "C" int pero_colsum(const void* x, float* out, int64_t rows, int64_t cols, int64_t ld, int dtype, void* stream) {
  PERO_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "pero_colsum: bad arguments");
  const int esz = dtype == PERO_F32 ? 4 : 2;
  if (cols % 8 == 0 && (ld * esz) % 16 == 0 && aligned16(x) && (rows + 127) / 128 < 65536) {
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)((rows + 127) / 128)), block(256);
    if (dtype == PERO_F32) hipLaunchKernelGGL((colsum8_k<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, out, (long long)rows, (long long)cols, (long long)ld);
    else if (dtype == PERO_BF16) hipLaunchKernelGGL((colsum8_k<bf16raw>), grid, block, 0, (hipStream_t)stream, (const bf16raw*)x, out, (long long)rows, (long long)cols, (long long)ld);
    else PERO_REQUIRE(false, "pero_colsum: bad dtype");
    PERO_CHECK_LAUNCH("pero_colsum");
    return PERO_OK;
  }
  long long gy = rows < 128 ? rows : 128;
  dim3 grid((unsigned)((cols + 255) / 256), (unsigned)gy), block(256);
  if (dtype == PERO_F32) hipLaunchKernelGGL((colsum_k<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, out, (long long)rows, (long long)cols, (long long)ld);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((colsum_k<bf16raw>), grid, block, 0, (hipStream_t)stream, (const bf16raw*)x, out, (long long)rows, (long long)cols, (long long)ld);
  else PERO_REQUIRE(false, "pero_colsum: bad dtype");
  PERO_CHECK_LAUNCH("pero_colsum");
  return PERO_OK;
}
