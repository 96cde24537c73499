// Reduction kernels of the joint-embedding losses (joint_embedding_pretraining/losses.py of the reference).
// The dense contractions (VICReg covariance SYRK, its backward, the per-line NT-Xent similarity matrices and
// their backward products) run on pero_gemm; everything here is HBM-bound row/column work in f32 statistics.
#include "common.hpp"
#include <initializer_list>

// --------------------------------------------------------------------------------------------
// squared-difference sum of gathered rows (VICReg invariance, losses.py:14-16):
//   partial[row] = sum_c (x[ix[row]][c] - y[iy[row]][c])^2      (one wave per row, deterministic)
// --------------------------------------------------------------------------------------------
template <typename T, bool V8>
__global__ __launch_bounds__(256) void sqdiff_rows_k(const T* x, const int64_t* ix, const T* y, const int64_t* iy, float* partial,
                                                     long long n, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const T* a = x + ix[row] * d;
  const T* b = y + iy[row] * d;
  float s = 0.f;
  if (V8) {  // 16-byte accesses: d % 8 == 0, 16-byte aligned rows (the scalar loop moved 2 bytes per lane and instruction)
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float va[8], vb[8];
      load8<T>(a + c, va); load8<T>(b + c, vb);
#pragma unroll
      for (int e = 0; e < 8; e++) { const float t = va[e] - vb[e]; s += t * t; }
    }
  } else {
    for (int c = threadIdx.x & 63; c < d; c += 64) { const float t = Elem<T>::ld(a + c) - Elem<T>::ld(b + c); s += t * t; }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) partial[row] = s;
}
// out[0] = scale * sum(partial[0..n))  (single block, fixed order)
__global__ __launch_bounds__(256) void sum_scale_k(const float* partial, float* out, long long n, float scale) {
  __shared__ float sm[4];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) * scale;
}
// dx[ix[row]] += g*coef*(x-y), dy[iy[row]] -= g*coef*(x-y)   (rows of ix / iy are unique)
template <typename T, bool V8>
__global__ __launch_bounds__(256) void sqdiff_rows_bwd_k(const T* x, const int64_t* ix, const T* y, const int64_t* iy, T* dx, T* dy,
                                                         const float* g, float coef, long long n, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float c0 = coef * (g ? g[0] : 1.f);
  const long long ra = ix[row], rb = iy[row];
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float vx[8], vy[8], gx[8], gy[8];
      load8<T>(x + ra * d + c, vx); load8<T>(y + rb * d + c, vy); load8<T>(dx + ra * d + c, gx); load8<T>(dy + rb * d + c, gy);
#pragma unroll
      for (int e = 0; e < 8; e++) { const float t = c0 * (vx[e] - vy[e]); gx[e] += t; gy[e] -= t; }
      store8<T>(dx + ra * d + c, gx); store8<T>(dy + rb * d + c, gy);
    }
    return;
  }
  for (int c = threadIdx.x & 63; c < d; c += 64) {
    const float t = c0 * (Elem<T>::ld(x + ra * d + c) - Elem<T>::ld(y + rb * d + c));
    Elem<T>::st(dx + ra * d + c, Elem<T>::ld(dx + ra * d + c) + t);
    Elem<T>::st(dy + rb * d + c, Elem<T>::ld(dy + rb * d + c) - t);
  }
}

// --------------------------------------------------------------------------------------------
// VICReg statistics (losses.py:37-47).  z: (m_pad, d) gathered rows, rows >= m are zero padding.
// center: zc = z - mean (padding rows stay 0), sumsq[c] += sum_rows zc^2.   colsum (= m * mean) is an input.
// --------------------------------------------------------------------------------------------
template <typename T, bool V8>
__global__ __launch_bounds__(256) void center_cols_k(const T* z, const float* colsum, T* zc, float* sumsq, long long m, long long m_pad, int d) {
  __shared__ float red[8][256 + 8];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const long long col = ((long long)blockIdx.x * 32 + cg) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < d) {
    float mu[8];
#pragma unroll
    for (int e = 0; e < 8; e++) mu[e] = colsum[col + e] / (float)m;
    const long long r0 = (long long)blockIdx.y * 128;
    const long long r1 = r0 + 128 < m_pad ? r0 + 128 : m_pad;
    for (long long r = r0 + rl; r < r1; r += 8) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (r < m) {
        if (V8) load8<T>(z + r * d + col, v);
        else {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = Elem<T>::ld(z + r * d + col + e);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] -= mu[e];
      }
      if (V8) store8<T>(zc + r * d + col, v);
      else {
#pragma unroll
        for (int e = 0; e < 8; e++) Elem<T>::st(zc + r * d + col + e, v[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += v[e] * v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[rl][cg * 8 + e] = acc[e];
  __syncthreads();
  const int c = threadIdx.x;
  const long long gc = (long long)blockIdx.x * 256 + c;
  if (gc < d) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) s += red[k][c];
    atomicAdd(sumsq + gc, s);
  }
}
// variance hinge (losses.py:37-38) + the per-column coefficient of its gradient:
//   std_j = sqrt(sumsq_j/(m-1) + eps); loss_var = mean_j relu(thr - std_j);  cvar_j = std_j < thr ? -1/(d*std_j*(m-1)) : 0
__global__ __launch_bounds__(256) void vicreg_var_k(const float* sumsq, float* cvar, float* loss_var, long long m, int d, float thr, float eps) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int j = threadIdx.x; j < d; j += 256) {
    const float sd = sqrtf(sumsq[j] / (float)(m - 1) + eps);
    const float h = thr - sd;
    s += h > 0.f ? h : 0.f;
    cvar[j] = h > 0.f ? -1.0f / ((float)d * sd * (float)(m - 1)) : 0.f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss_var[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)d;
}
// One pass over cov (d x d, f32): row partial sums of the squared off-diagonal entries, and the backward
// operand G (compute dtype):  G_ij = wc * 4 * cov_ij / (d * (m-1))  (i != j),  G_jj = wv * cvar_j,
// so that d loss / d zc = zc @ G  (cov symmetric; centring backward vanishes because columns of zc sum to 0).
template <typename T>
__global__ __launch_bounds__(256) void vicreg_cov_k(const float* cov, const float* cvar, T* G, float* rowpart, int d, long long m,
                                                    float wv, float wc) {
  __shared__ float sm[4];
  const int i = blockIdx.x;
  const float a = wc * 4.0f / ((float)d * (float)(m - 1));
  float s = 0.f;
  for (int j = threadIdx.x; j < d; j += 256) {
    const float c = cov[(long long)i * d + j];
    if (j != i) { s += c * c; Elem<T>::st(G + (long long)i * d + j, a * c); }
    else Elem<T>::st(G + (long long)i * d + j, wv * cvar[i]);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rowpart[i] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// dst[index[i]] += g * src[i]
template <typename T, bool V8>
__global__ __launch_bounds__(256) void scatter_add_scaled_k(const T* src, const int64_t* index, T* dst, const float* g, long long n, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float gs = g ? g[0] : 1.f;
  T* o = dst + index[row] * d;
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float a[8], b[8];
      load8<T>(o + c, a); load8<T>(src + row * d + c, b);
#pragma unroll
      for (int e = 0; e < 8; e++) a[e] += gs * b[e];
      store8<T>(o + c, a);
    }
    return;
  }
  for (int c = threadIdx.x & 63; c < d; c += 64) Elem<T>::st(o + c, Elem<T>::ld(o + c) + gs * Elem<T>::ld(src + row * d + c));
}

// --------------------------------------------------------------------------------------------
// NT-Xent (losses.py:56-83): L2 row normalisation and the column-normalised softmax-CE of per-line S x S
// similarity matrices.
// --------------------------------------------------------------------------------------------
template <typename T, bool V8>
__global__ __launch_bounds__(256) void rownorm_fwd_k(const T* x, T* xn, float* inv, long long rows, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float v[8];
      load8<T>(x + row * d + c, v);
#pragma unroll
      for (int e = 0; e < 8; e++) s += v[e] * v[e];
    }
  } else {
    for (int c = threadIdx.x & 63; c < d; c += 64) { const float v = Elem<T>::ld(x + row * d + c); s += v * v; }
  }
  s = wave_sum(s);
  const float r = 1.0f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize eps
  if ((threadIdx.x & 63) == 0) inv[row] = r;
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float v[8];
      load8<T>(x + row * d + c, v);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] *= r;
      store8<T>(xn + row * d + c, v);
    }
    return;
  }
  for (int c = threadIdx.x & 63; c < d; c += 64) Elem<T>::st(xn + row * d + c, Elem<T>::ld(x + row * d + c) * r);
}
// dx = (dxn - xn * <xn, dxn>) * inv
template <typename T, bool V8>
__global__ __launch_bounds__(256) void rownorm_bwd_k(const T* xn, const T* dxn, const float* inv, const float* g, T* dx, long long rows, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float a[8], b[8];
      load8<T>(xn + row * d + c, a); load8<T>(dxn + row * d + c, b);
#pragma unroll
      for (int e = 0; e < 8; e++) s += a[e] * b[e];
    }
  } else {
    for (int c = threadIdx.x & 63; c < d; c += 64) s += Elem<T>::ld(xn + row * d + c) * Elem<T>::ld(dxn + row * d + c);
  }
  s = wave_sum(s);
  const float r = inv[row] * (g ? g[0] : 1.f);
  if (V8) {
    for (int c = (threadIdx.x & 63) * 8; c < d; c += 512) {
      float a[8], b[8];
      load8<T>(xn + row * d + c, a); load8<T>(dxn + row * d + c, b);
#pragma unroll
      for (int e = 0; e < 8; e++) b[e] = (b[e] - a[e] * s) * r;
      store8<T>(dx + row * d + c, b);
    }
    return;
  }
  for (int c = threadIdx.x & 63; c < d; c += 64)
    Elem<T>::st(dx + row * d + c, (Elem<T>::ld(dxn + row * d + c) - Elem<T>::ld(xn + row * d + c) * s) * r);
}
// sim: (lines, S, S) f32.  One block per line; thread = column j: lse_j = log sum_r exp(sim[r][j]);
// line_loss = mean_j (lse_j - sim[j][j]);   dsim[r][j] = (exp(sim[r][j] - lse_j) - [r==j]) / (S * lines)
template <typename T>
__global__ __launch_bounds__(256) void ntxent_cols_k(const float* sim, float* line_loss, T* dsim, int S, int lines) {
  __shared__ float sm[4];
  const float* s = sim + (long long)blockIdx.x * S * S;
  float acc = 0.f;
  for (int j = threadIdx.x; j < S; j += 256) {
    float mx = -INFINITY;
    for (int r = 0; r < S; r++) mx = fmaxf(mx, s[(long long)r * S + j]);
    float sum = 0.f;
    for (int r = 0; r < S; r++) sum += expf(s[(long long)r * S + j] - mx);
    const float lse = logf(sum) + mx;
    acc += lse - s[(long long)j * S + j];
    if (dsim) {
      T* o = dsim + (long long)blockIdx.x * S * S;
      const float w = 1.0f / ((float)S * (float)lines);
      for (int r = 0; r < S; r++) Elem<T>::st(o + (long long)r * S + j, (expf(s[(long long)r * S + j] - lse) - (r == j ? 1.f : 0.f)) * w);
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) line_loss[blockIdx.x] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)S;
}


// ---- cross-rank NT-Xent (NTXentLoss(cross_rank_negatives=True); no reference counterpart) -------------------------------------------
// pooled[l][c] = mean over the S rows of line l of x[(l*S + s)][c]   (f32 out; x in T)
template <typename T>
__global__ __launch_bounds__(256) void line_mean_k(const T* x, float* out, int S, int d) {
  const long long l = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= d) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const T* p = x + l * S * (long long)d + c;
  for (int s = 0; s < S; s++) {
    float v[8];
    load8<T>(p + (long long)s * d, v);
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] += v[e];
  }
  const float w = 1.0f / (float)S;
#pragma unroll
  for (int e = 0; e < 8; e++) acc[e] *= w;
  store8<float>(out + l * d + c, acc);
}
// dst[(l*S + s)][c] += scale * src[l][c]   (the backward of the mean: every row of a line receives the line's gradient / S)
template <typename T>
__global__ __launch_bounds__(256) void add_line_rows_k(T* dst, const float* src, int S, int d, float scale) {
  const long long l = blockIdx.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= d) return;
  float g[8];
  load8<float>(src + l * d + c, g);
  T* p = dst + l * S * (long long)d + c;
  for (int s = 0; s < S; s++) {
    float v[8];
    load8<T>(p + (long long)s * d, v);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] += scale * g[e];
    store8<T>(p + (long long)s * d, v);
  }
}
// One workgroup per line.  Column j of the line's similarity block (sim[l][i][j], i = 0..S-1) and row l*S + j of `cross` (the
// similarities of y_j with the L pooled embeddings; the line's own, index own0 + l, is left out) share ONE log-sum-exp:
//   lse_j = log( sum_i exp(sim[i][j]) + sum_{l' != own} exp(cross[j][l']) ),   line_loss = mean_j (lse_j - sim[j][j]),
//   dsim[i][j] = (exp(sim[i][j] - lse_j) - [i == j]) w,   dcross[j][l'] = exp(cross[j][l'] - lse_j) w (0 for the own line),  w = 1 / (S lines).
// Pass 1: thread j walks column j of sim (coalesced over the threads); pass 2: a wave per row of cross, lanes over l' (coalesced);
// the statistics meet in LDS.  Nothing is concatenated, masked or copied (torch did: cat, one_hot, masked_fill, logsumexp).
template <typename T>
__global__ __launch_bounds__(256) void ntxent_cols_cross_k(const float* sim, const float* cross, float* line_loss, T* dsim, T* dcross, int S, int L,
                                                           int lines, int own0) {
  extern __shared__ float sh[];          // [S] max, [S] sum, [S] lse
  float* cmx = sh; float* csum = sh + S; float* clse = sh + 2 * S;
  __shared__ float red[4];
  const long long l = blockIdx.x;
  const float* s = sim + l * S * (long long)S;
  const float* cr = cross + l * S * (long long)L;
  const int own = own0 + (int)l;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < S; j += 256) {
    float mx = -INFINITY;
    for (int r = 0; r < S; r++) mx = fmaxf(mx, s[(long long)r * S + j]);
    float sum = 0.f;
    for (int r = 0; r < S; r++) sum += expf(s[(long long)r * S + j] - mx);
    cmx[j] = mx; csum[j] = sum;
  }
  __syncthreads();
  for (int j = wave; j < S; j += 4) {
    const float* row = cr + (long long)j * L;
    float mx = -INFINITY;
    for (int c = lane; c < L; c += 64) if (c != own) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    const float m1 = cmx[j], m = fmaxf(m1, mx);
    float sum = 0.f;
    for (int c = lane; c < L; c += 64) if (c != own) sum += expf(row[c] - m);
    sum = wave_sum(sum);
    if (lane == 0) clse[j] = logf(csum[j] * expf(m1 - m) + sum) + m;
  }
  __syncthreads();
  const float w = 1.0f / ((float)S * (float)lines);
  float acc = 0.f;
  for (int j = tid; j < S; j += 256) {
    const float lse = clse[j];
    acc += lse - s[(long long)j * S + j];
    if (dsim) {
      T* o = dsim + l * S * (long long)S;
      for (int r = 0; r < S; r++) Elem<T>::st(o + (long long)r * S + j, (expf(s[(long long)r * S + j] - lse) - (r == j ? 1.f : 0.f)) * w);
    }
  }
  if (dcross) {
    for (int j = wave; j < S; j += 4) {
      const float* row = cr + (long long)j * L;
      T* o = dcross + (l * S + j) * (long long)L;
      const float lse = clse[j];
      for (int c = lane; c < L; c += 64) Elem<T>::st(o + c, c == own ? 0.f : expf(row[c] - lse) * w);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (tid == 0) line_loss[l] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)S;
}

#define DISPATCH_T(dtype, NAME, ...)                                                              \
  do {                                                                                            \
    if (dtype == PERO_F32) { NAME(float, __VA_ARGS__); }                                          \
    else if (dtype == PERO_BF16) { NAME(bf16raw, __VA_ARGS__); }                                  \
    else PERO_REQUIRE(false, "bad dtype");                                                        \
  } while (0)

// 16-byte row accesses are possible when every row starts 16-byte aligned
static inline bool v8_ok(int64_t d, int dtype, std::initializer_list<const void*> ptrs) {
  const int esz = dtype == PERO_F32 ? 4 : 2;
  if (d % 8 || (d * esz) % 16) return false;
  for (const void* q : ptrs) if (!aligned16(q)) return false;
  return true;
}

extern "C" int pero_sqdiff_rows(const void* x, const int64_t* ix, const void* y, const int64_t* iy, float* partial, float* out,
                                int64_t n, int64_t d, float scale, int dtype, void* stream) {
  PERO_REQUIRE(x && y && ix && iy && partial && out && n > 0 && d > 0, "pero_sqdiff_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((n + 3) / 4)), block(256);
  const bool v8 = v8_ok(d, dtype, {x, y});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((sqdiff_rows_k<T, true>), grid, block, 0, st, (const T*)x, ix, (const T*)y, iy, partial, (long long)n, (int)d); \
                        else hipLaunchKernelGGL((sqdiff_rows_k<T, false>), grid, block, 0, st, (const T*)x, ix, (const T*)y, iy, partial, (long long)n, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  hipLaunchKernelGGL(sum_scale_k, dim3(1), dim3(256), 0, st, partial, out, (long long)n, scale);
  PERO_CHECK_LAUNCH("pero_sqdiff_rows");
  return PERO_OK;
}
extern "C" int pero_sqdiff_rows_bwd(const void* x, const int64_t* ix, const void* y, const int64_t* iy, void* dx, void* dy,
                                    const float* g, float coef, int64_t n, int64_t d, int dtype, void* stream) {
  PERO_REQUIRE(x && y && ix && iy && dx && dy && n > 0 && d > 0, "pero_sqdiff_rows_bwd: bad arguments");
  dim3 grid((unsigned)((n + 3) / 4)), block(256);
  const bool v8 = v8_ok(d, dtype, {x, y, dx, dy});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((sqdiff_rows_bwd_k<T, true>), grid, block, 0, (hipStream_t)stream, (const T*)x, ix, (const T*)y, iy, (T*)dx, (T*)dy, g, coef, (long long)n, (int)d); \
                        else hipLaunchKernelGGL((sqdiff_rows_bwd_k<T, false>), grid, block, 0, (hipStream_t)stream, (const T*)x, ix, (const T*)y, iy, (T*)dx, (T*)dy, g, coef, (long long)n, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_sqdiff_rows_bwd");
  return PERO_OK;
}
extern "C" int pero_sum_scale(const float* partial, float* out, int64_t n, float scale, void* stream) {
  PERO_REQUIRE(partial && out && n > 0, "pero_sum_scale: bad arguments");
  hipLaunchKernelGGL(sum_scale_k, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, out, (long long)n, scale);
  PERO_CHECK_LAUNCH("pero_sum_scale");
  return PERO_OK;
}
extern "C" int pero_center_cols(const void* z, const float* colsum, void* zc, float* sumsq, int64_t m, int64_t m_pad, int64_t d,
                                int dtype, void* stream) {
  PERO_REQUIRE(z && colsum && zc && sumsq && m > 1 && m_pad >= m && d > 0 && d % 8 == 0, "pero_center_cols: bad arguments (d %% 8 == 0)");
  dim3 grid((unsigned)((d + 255) / 256), (unsigned)((m_pad + 127) / 128)), block(256);
  const bool v8 = v8_ok(d, dtype, {z, zc});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((center_cols_k<T, true>), grid, block, 0, (hipStream_t)stream, (const T*)z, colsum, (T*)zc, sumsq, (long long)m, (long long)m_pad, (int)d); \
                        else hipLaunchKernelGGL((center_cols_k<T, false>), grid, block, 0, (hipStream_t)stream, (const T*)z, colsum, (T*)zc, sumsq, (long long)m, (long long)m_pad, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_center_cols");
  return PERO_OK;
}
extern "C" int pero_vicreg_var(const float* sumsq, float* cvar, float* loss_var, int64_t m, int64_t d, float threshold, float eps,
                               void* stream) {
  PERO_REQUIRE(sumsq && cvar && loss_var && m > 1 && d > 0, "pero_vicreg_var: bad arguments");
  hipLaunchKernelGGL(vicreg_var_k, dim3(1), dim3(256), 0, (hipStream_t)stream, sumsq, cvar, loss_var, (long long)m, (int)d, threshold, eps);
  PERO_CHECK_LAUNCH("pero_vicreg_var");
  return PERO_OK;
}
extern "C" int pero_vicreg_cov(const float* cov, const float* cvar, void* G, float* rowpart, float* loss_cov, int64_t d, int64_t m,
                               float wv, float wc, int dtype, void* stream) {
  PERO_REQUIRE(cov && cvar && G && rowpart && loss_cov && d > 0 && m > 1, "pero_vicreg_cov: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define L_(T, ...) hipLaunchKernelGGL((vicreg_cov_k<T>), dim3((unsigned)d), dim3(256), 0, st, cov, cvar, (T*)G, rowpart, (int)d, (long long)m, wv, wc)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  hipLaunchKernelGGL(sum_scale_k, dim3(1), dim3(256), 0, st, rowpart, loss_cov, (long long)d, 1.0f / (float)d);
  PERO_CHECK_LAUNCH("pero_vicreg_cov");
  return PERO_OK;
}
extern "C" int pero_scatter_add_rows_scaled(const void* src, const int64_t* index, void* dst, const float* g, int64_t n, int64_t d,
                                            int dtype, void* stream) {
  PERO_REQUIRE(src && index && dst && n > 0 && d > 0, "pero_scatter_add_rows_scaled: bad arguments");
  dim3 grid((unsigned)((n + 3) / 4)), block(256);
  const bool v8 = v8_ok(d, dtype, {src, dst});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((scatter_add_scaled_k<T, true>), grid, block, 0, (hipStream_t)stream, (const T*)src, index, (T*)dst, g, (long long)n, (int)d); \
                        else hipLaunchKernelGGL((scatter_add_scaled_k<T, false>), grid, block, 0, (hipStream_t)stream, (const T*)src, index, (T*)dst, g, (long long)n, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_scatter_add_rows_scaled");
  return PERO_OK;
}
extern "C" int pero_rownorm_fwd(const void* x, void* xn, float* inv, int64_t rows, int64_t d, int dtype, void* stream) {
  PERO_REQUIRE(x && xn && inv && rows > 0 && d > 0, "pero_rownorm_fwd: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const bool v8 = v8_ok(d, dtype, {x, xn});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((rownorm_fwd_k<T, true>), grid, block, 0, (hipStream_t)stream, (const T*)x, (T*)xn, inv, (long long)rows, (int)d); \
                        else hipLaunchKernelGGL((rownorm_fwd_k<T, false>), grid, block, 0, (hipStream_t)stream, (const T*)x, (T*)xn, inv, (long long)rows, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_rownorm_fwd");
  return PERO_OK;
}
extern "C" int pero_rownorm_bwd(const void* xn, const void* dxn, const float* inv, const float* g, void* dx, int64_t rows, int64_t d,
                                int dtype, void* stream) {
  PERO_REQUIRE(xn && dxn && inv && dx && rows > 0 && d > 0, "pero_rownorm_bwd: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  const bool v8 = v8_ok(d, dtype, {xn, dxn, dx});
#define L_(T, ...) do { if (v8) hipLaunchKernelGGL((rownorm_bwd_k<T, true>), grid, block, 0, (hipStream_t)stream, (const T*)xn, (const T*)dxn, inv, g, (T*)dx, (long long)rows, (int)d); \
                        else hipLaunchKernelGGL((rownorm_bwd_k<T, false>), grid, block, 0, (hipStream_t)stream, (const T*)xn, (const T*)dxn, inv, g, (T*)dx, (long long)rows, (int)d); } while (0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_rownorm_bwd");
  return PERO_OK;
}
extern "C" int pero_ntxent_cols(const float* sim, float* line_loss, float* loss_out, void* dsim, int64_t lines, int64_t S, int dtype,
                                void* stream) {
  PERO_REQUIRE(sim && line_loss && loss_out && lines > 0 && S > 0, "pero_ntxent_cols: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define L_(T, ...) hipLaunchKernelGGL((ntxent_cols_k<T>), dim3((unsigned)lines), dim3(256), 0, st, sim, line_loss, (T*)dsim, (int)S, (int)lines)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  hipLaunchKernelGGL(sum_scale_k, dim3(1), dim3(256), 0, st, line_loss, loss_out, (long long)lines, 1.0f / (float)lines);
  PERO_CHECK_LAUNCH("pero_ntxent_cols");
  return PERO_OK;
}

extern "C" int pero_line_mean(const void* x, float* out, int64_t lines, int64_t S, int64_t d, int dtype, void* stream) {
  PERO_REQUIRE(x && out && lines > 0 && S > 0 && d > 0 && d % 8 == 0 && lines < 65536, "pero_line_mean: bad arguments (d %% 8 == 0)");
  PERO_REQUIRE(v8_ok(d, dtype, {x}) && aligned16(out), "pero_line_mean: 16-byte aligned rows");
  dim3 grid((unsigned)((d / 8 + 255) / 256), (unsigned)lines), block(256);
#define L_(T, ...) hipLaunchKernelGGL((line_mean_k<T>), grid, block, 0, (hipStream_t)stream, (const T*)x, out, (int)S, (int)d)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_line_mean");
  return PERO_OK;
}
extern "C" int pero_add_line_rows(void* dst, const float* src, int64_t lines, int64_t S, int64_t d, float scale, int dtype, void* stream) {
  PERO_REQUIRE(dst && src && lines > 0 && S > 0 && d > 0 && d % 8 == 0 && lines < 65536, "pero_add_line_rows: bad arguments (d %% 8 == 0)");
  PERO_REQUIRE(v8_ok(d, dtype, {dst}) && aligned16(src), "pero_add_line_rows: 16-byte aligned rows");
  dim3 grid((unsigned)((d / 8 + 255) / 256), (unsigned)lines), block(256);
#define L_(T, ...) hipLaunchKernelGGL((add_line_rows_k<T>), grid, block, 0, (hipStream_t)stream, (T*)dst, src, (int)S, (int)d, scale)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  PERO_CHECK_LAUNCH("pero_add_line_rows");
  return PERO_OK;
}
extern "C" int pero_ntxent_cols_cross(const float* sim, const float* cross, float* line_loss, float* loss_out, void* dsim, void* dcross,
                                      int64_t lines, int64_t S, int64_t L, int64_t own0, int dtype, void* stream) {
  PERO_REQUIRE(sim && cross && line_loss && loss_out && lines > 0 && S > 0 && L > 0 && own0 >= 0 && own0 + lines <= L && S <= 4096,
               "pero_ntxent_cols_cross: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define L_(T, ...) hipLaunchKernelGGL((ntxent_cols_cross_k<T>), dim3((unsigned)lines), dim3(256), (size_t)(3 * S * sizeof(float)), st, sim, cross, line_loss, \
                                      (T*)dsim, (T*)dcross, (int)S, (int)L, (int)lines, (int)own0)
  DISPATCH_T(dtype, L_, 0);
#undef L_
  hipLaunchKernelGGL(sum_scale_k, dim3(1), dim3(256), 0, st, line_loss, loss_out, (long long)lines, 1.0f / (float)lines);
  PERO_CHECK_LAUNCH("pero_ntxent_cols_cross");
  return PERO_OK;
}
