// Front end (u8 line images -> patch rows), casts, Adam, row gather / scatter.  All HBM-bound.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// patch rows from u8 NHWC line images.  One block = one line x 16 consecutive patches: the
// 40 x (16*P*C)-byte sub-image is read once with coalesced 4-byte loads into LDS, then written out as
// whole (c,h,p)-ordered patch rows (contiguous C*H*P elements per patch).
// ---------------------------------------------------------------------------------------------
#define PT_TOK 16
// x / 255.0f exactly as the reference's IEEE division (batch_operator.py:18), for byte values: reciprocal multiply plus one
// fma correction step (q = x*r; q += fma(-q, 255, x) * r) - equal to the correctly rounded quotient for all 256 inputs
// (tests/test_gpu_ops.py checks every value); a per-pixel division is a ~10-instruction sequence, a table in LDS conflicts.
__device__ __forceinline__ float div255(unsigned char x) {
  const float xf = (float)x, r = 1.0f / 255.0f;
  const float q = xf * r;
  return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, xf), r, q);
}
template <typename T>
__global__ __launch_bounds__(256) void patches_u8_k(const uint8_t* img, const int64_t* mask, const float* tile, T* out,
                                                    int H, int W, int C, int P, int S, int ldo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int n = blockIdx.y, s0 = blockIdx.x * PT_TOK;
  const int ntok = (S - s0) < PT_TOK ? (S - s0) : PT_TOK;
  const int rowbytes = ntok * P * C;
  const long long rowpitch = (long long)W * C;
  const uint8_t* src = img + ((long long)n * H) * rowpitch + (long long)s0 * P * C;
  // LDS row pitch: an ODD number of dwords - lanes walk the image rows (h) at a fixed column, and the unpadded 384-byte pitch
  // (96 dwords) put all 40 rows on two banks (32-way conflict on every pixel read)
  const int ldsp = (((PT_TOK * P * C + 3) >> 2) | 1) << 2;
  // the block's mask words next to the staged pixels (read per token below: from LDS, not one dependent global load per token)
  int* const lmask = (int*)(lds + H * ldsp);
  if (threadIdx.x < PT_TOK) lmask[threadIdx.x] = (mask && (int)threadIdx.x < ntok) ? (mask[(long long)n * S + s0 + threadIdx.x] == 1) : 0;
  if ((rowbytes & 15) == 0 && (rowpitch & 15) == 0 && ((s0 * P * C) & 15) == 0 && ((uintptr_t)img & 15) == 0) {
    // 16-byte pieces, four per thread in flight (a thread's loads are issued together: the loop below them waited for each one)
    const int rw = rowbytes >> 4, total = H * rw;
    for (int base = threadIdx.x; base < total; base += 256 * 4) {
      uint4 v[4];
      int off[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int i = base + 256 * k;
        off[k] = -1;
        if (i < total) {
          const int h = i / rw, b = i - h * rw;
          v[k] = *(const uint4*)(src + h * rowpitch + 16 * b);
          off[k] = h * ldsp + 16 * b;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (off[k] >= 0) {
          unsigned* d = (unsigned*)(lds + off[k]);
          d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
        }
    }
  } else if ((rowbytes & 3) == 0 && (rowpitch & 3) == 0 && ((s0 * P * C) & 3) == 0 && ((uintptr_t)img & 3) == 0) {
    const int rw = rowbytes >> 2;
    int h = threadIdx.x / rw, b = threadIdx.x - h * rw;  // element (h, b), advanced by 256 per iteration without divisions
    const int dh = 256 / rw, db = 256 - dh * rw;
    while (h < H) {
      *(unsigned*)(lds + h * ldsp + 4 * b) = *(const unsigned*)(src + h * rowpitch + 4 * b);
      h += dh; b += db;
      if (b >= rw) { b -= rw; h++; }
    }
  } else {
    for (int i = threadIdx.x; i < H * rowbytes; i += 256) {
      const int h = i / rowbytes, b = i - h * rowbytes;
      lds[h * ldsp + b] = src[h * rowpitch + b];
    }
  }
  __syncthreads();
  const int CH = C * H;  // one group = the P pixels of one (token, c, h); thread -> ch = tid % chp, tokens tid / chp, + step, ...
  const int pd = CH * P;
  int chp = 1;
  while (chp < CH) chp <<= 1;               // 128 for C*H = 120
  const int tstep = chp <= 256 ? 256 / chp : 1;
  if (chp <= 256) {
    const int ch = threadIdx.x & (chp - 1);
    if (ch < CH) {
      const int c = ch / H, h = ch - c * H;
      const unsigned char* lrow = lds + h * ldsp + c;
      for (int tok = threadIdx.x / chp; tok < ntok; tok += tstep) {
        const bool masked = lmask[tok] != 0;
        T* o = out + ((long long)n * S + s0 + tok) * ldo + (long long)ch * P;
        if (sizeof(T) == 2 && P == 8 && (ldo & 7) == 0 && (((uintptr_t)out) & 15) == 0) {
          float v[8];  // the 8 pixels of one (token, c, h) are 16 contiguous output bytes: one 16-byte store
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = masked ? tile[ch * 8 + e] : div255(lrow[(tok * 8 + e) * C]);
          uint4 w;
          w.x = pack2bf(v[0], v[1]); w.y = pack2bf(v[2], v[3]); w.z = pack2bf(v[4], v[5]); w.w = pack2bf(v[6], v[7]);
          *(uint4*)o = w;
        } else {
          for (int e = 0; e < P; e++) Elem<T>::st(o + e, masked ? tile[ch * P + e] : div255(lrow[(tok * P + e) * C]));
        }
      }
    }
  } else {  // very tall patches: generic index arithmetic
    for (int gidx = threadIdx.x; gidx < ntok * CH; gidx += 256) {
      const int tok = gidx / CH, ch = gidx - tok * CH;
      const int c = ch / H, h = ch - c * H;
      const bool masked = lmask[tok] != 0;
      T* o = out + ((long long)n * S + s0 + tok) * ldo + (long long)ch * P;
      for (int e = 0; e < P; e++) Elem<T>::st(o + e, masked ? tile[ch * P + e] : div255(lds[h * ldsp + (tok * P + e) * C + c]));
    }
  }
  const int padn = ldo - pd;  // zero the row padding (GEMM-friendly pitch)
  if (sizeof(T) == 2 && (padn & 7) == 0 && (pd & 7) == 0 && (ldo & 7) == 0 && (((uintptr_t)out) & 15) == 0) {
    const int pc = padn >> 3;
    for (int i = threadIdx.x; i < ntok * pc; i += 256) {
      const int tok = i / pc, e = i - tok * pc;
      *(uint4*)(out + ((long long)n * S + s0 + tok) * ldo + pd + 8 * e) = make_uint4(0, 0, 0, 0);
    }
  } else {
    for (int i = threadIdx.x; i < ntok * padn; i += 256) {
      const int tok = i / padn, e = i - tok * padn;
      Elem<T>::st(out + ((long long)n * S + s0 + tok) * ldo + pd + e, 0.f);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void patches_f32_k(const float* img, const int64_t* mask, const float* tile, T* out,
                                                     long long total, int H, int W, int C, int P, int S, int ldo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index over (token row, padded column)
  if (i >= total) return;
  const int colp = (int)(i % ldo);
  const long long tokrow = i / ldo;
  if (colp >= C * H * P) { Elem<T>::st(out + i, 0.f); return; }
  const int p = colp % P;
  long long r = tokrow * (C * H) + colp / P;
  const int h = (int)(r % H); r /= H;
  const int c = (int)(r % C); r /= C;
  const int s = (int)(r % S);
  const long long n = r / S;
  const bool masked = mask && mask[n * S + s] == 1;
  const float v = masked ? tile[(c * H + h) * P + p] : img[((n * C + c) * H + h) * W + (long long)s * P + p];
  Elem<T>::st(out + i, v);
}
__global__ __launch_bounds__(256) void add_rows2d_k(float* dst, const float* src, long long rows, long long cols, long long ldd, long long lds) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long long r = i / cols, c = i - r * cols;
  dst[r * ldd + c] += src[r * lds + c];
}
__global__ __launch_bounds__(256) void cast_pad_k(const float* src, bf16raw* dst, long long rows, long long cols, long long ldd) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * ldd) return;
  const long long r = i / ldd, c = i - r * ldd;
  dst[i] = c < cols ? f2bf(src[r * cols + c]) : (bf16raw)0;
}
__global__ __launch_bounds__(256) void apply_mask_k(float* img, const int64_t* mask, const float* tile, long long total,
                                                    int H, int W, int C, int P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // index over (n, c, h, w)
  if (i >= total) return;
  const int w = (int)(i % W);
  long long r = i / W;
  const int h = (int)(r % H); r /= H;
  const int c = (int)(r % C);
  const long long n = r / C;
  if (mask[n * (W / P) + w / P] == 1) img[i] = tile[(c * H + h) * P + (w % P)];
}

extern "C" int pero_patches_from_u8(const uint8_t* images, const int64_t* mask, const float* tile, void* patches,
                                    int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t ld_out, int dtype, void* stream) {
  PERO_REQUIRE(images && patches && (tile || !mask), "pero_patches_from_u8: null pointer");
  PERO_REQUIRE(ld_out >= C * H * P, "pero_patches_from_u8: ld_out < C*H*P");
  PERO_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && P > 0 && W % P == 0 && N < 65536, "pero_patches_from_u8: bad sizes (W %% P must be 0)");
  const int S = (int)(W / P);
  const size_t lds = (size_t)H * ((((PT_TOK * P * C + 3) >> 2) | 1) << 2) + PT_TOK * sizeof(int);
  PERO_REQUIRE(lds <= 65536, "pero_patches_from_u8: H*16*P*C = %zu bytes exceeds the LDS staging budget", lds);
  dim3 grid((unsigned)((S + PT_TOK - 1) / PT_TOK), (unsigned)N), block(256);
  if (dtype == PERO_F32) hipLaunchKernelGGL((patches_u8_k<float>), grid, block, lds, (hipStream_t)stream, images, mask, tile, (float*)patches, (int)H, (int)W, (int)C, (int)P, S, (int)ld_out);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((patches_u8_k<bf16raw>), grid, block, lds, (hipStream_t)stream, images, mask, tile, (bf16raw*)patches, (int)H, (int)W, (int)C, (int)P, S, (int)ld_out);
  else PERO_REQUIRE(false, "pero_patches_from_u8: bad dtype");
  PERO_CHECK_LAUNCH("pero_patches_from_u8");
  return PERO_OK;
}
extern "C" int pero_patches_from_f32(const float* images, const int64_t* mask, const float* tile, void* patches,
                                     int64_t N, int64_t H, int64_t W, int64_t C, int64_t P, int64_t ld_out, int dtype, void* stream) {
  PERO_REQUIRE(images && patches && (tile || !mask), "pero_patches_from_f32: null pointer");
  PERO_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && P > 0 && W % P == 0 && ld_out >= C * H * P, "pero_patches_from_f32: bad sizes");
  const long long total = (long long)N * (W / P) * ld_out;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == PERO_F32) hipLaunchKernelGGL((patches_f32_k<float>), grid, block, 0, (hipStream_t)stream, images, mask, tile, (float*)patches, total, (int)H, (int)W, (int)C, (int)P, (int)(W / P), (int)ld_out);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((patches_f32_k<bf16raw>), grid, block, 0, (hipStream_t)stream, images, mask, tile, (bf16raw*)patches, total, (int)H, (int)W, (int)C, (int)P, (int)(W / P), (int)ld_out);
  else PERO_REQUIRE(false, "pero_patches_from_f32: bad dtype");
  PERO_CHECK_LAUNCH("pero_patches_from_f32");
  return PERO_OK;
}
extern "C" int pero_apply_mask_f32(float* images, const int64_t* mask, const float* tile, int64_t N, int64_t H, int64_t W,
                                   int64_t C, int64_t P, void* stream) {
  PERO_REQUIRE(images && mask && tile, "pero_apply_mask_f32: null pointer");
  PERO_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && P > 0 && W % P == 0, "pero_apply_mask_f32: bad sizes");
  const long long total = (long long)N * C * H * W;
  hipLaunchKernelGGL(apply_mask_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, images, mask, tile, total, (int)H, (int)W, (int)C, (int)P);
  PERO_CHECK_LAUNCH("pero_apply_mask_f32");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// casts / scale
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_k(const float* src, bf16raw* dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * 8;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const f4v a = *(const f4v*)(src + i), b = *(const f4v*)(src + i + 4);
      uint4 o;
      o.x = pack2bf(a[0], a[1]); o.y = pack2bf(a[2], a[3]); o.z = pack2bf(b[0], b[1]); o.w = pack2bf(b[2], b[3]);
      *(uint4*)(dst + i) = o;
    } else {
      for (long long j = i; j < n; j++) dst[j] = f2bf(src[j]);
    }
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_k(const bf16raw* src, float* dst, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = bf2f(src[i]);
}
template <typename T>
__global__ __launch_bounds__(256) void scale_k(T* x, long long n, float s) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) Elem<T>::st(x + i, Elem<T>::ld(x + i) * s);
}
static unsigned grid_for(long long n, int per) {
  long long b = (n + (long long)256 * per - 1) / ((long long)256 * per);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}
extern "C" int pero_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  PERO_REQUIRE(src && dst && n > 0 && aligned16(src) && aligned16(dst), "pero_cast_f32_bf16: bad arguments");
  hipLaunchKernelGGL(cast_f32_bf16_k, dim3(grid_for(n, 8)), dim3(256), 0, (hipStream_t)stream, src, (bf16raw*)dst, (long long)n);
  PERO_CHECK_LAUNCH("pero_cast_f32_bf16");
  return PERO_OK;
}
extern "C" int pero_cast_pad_f32_bf16(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_dst, void* stream) {
  PERO_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_dst >= cols, "pero_cast_pad_f32_bf16: bad arguments");
  hipLaunchKernelGGL(cast_pad_k, dim3((unsigned)((rows * ld_dst + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16raw*)dst, (long long)rows, (long long)cols, (long long)ld_dst);
  PERO_CHECK_LAUNCH("pero_cast_pad_f32_bf16");
  return PERO_OK;
}
extern "C" int pero_add_rows2d(float* dst, const float* src, int64_t rows, int64_t cols, int64_t ld_dst, int64_t ld_src, void* stream) {
  PERO_REQUIRE(dst && src && rows > 0 && cols > 0 && ld_dst >= cols && ld_src >= cols, "pero_add_rows2d: bad arguments");
  hipLaunchKernelGGL(add_rows2d_k, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, src, (long long)rows, (long long)cols, (long long)ld_dst, (long long)ld_src);
  PERO_CHECK_LAUNCH("pero_add_rows2d");
  return PERO_OK;
}
extern "C" int pero_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
  PERO_REQUIRE(src && dst && n > 0, "pero_cast_bf16_f32: bad arguments");
  hipLaunchKernelGGL(cast_bf16_f32_k, dim3(grid_for(n, 1)), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)src, dst, (long long)n);
  PERO_CHECK_LAUNCH("pero_cast_bf16_f32");
  return PERO_OK;
}
// linear zero fill, 16 bytes per lane (torch's fill kernel wrote the step's 268 MB token-gradient matrix at 1.7 TB/s)
__global__ __launch_bounds__(256) void zero_fill_k(f4v* p, long long n16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = (f4v){0.f, 0.f, 0.f, 0.f};
}
extern "C" int pero_zero_fill(void* p, int64_t nbytes, void* stream) {
  PERO_REQUIRE(p && nbytes > 0, "pero_zero_fill: bad arguments");
  if (aligned16(p) && nbytes % 16 == 0) {
    const long long n16 = nbytes / 16;
    const long long blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(zero_fill_k, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, (f4v*)p, n16);
    PERO_CHECK_LAUNCH("pero_zero_fill");
    return PERO_OK;
  }
  if (hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream) != hipSuccess) { pero_set_error("pero_zero_fill: memset failed"); return PERO_E_LAUNCH; }
  return PERO_OK;
}
extern "C" int pero_scale(void* x, int64_t n, float scale, int dtype, void* stream) {
  PERO_REQUIRE(x && n > 0, "pero_scale: bad arguments");
  if (dtype == PERO_F32) hipLaunchKernelGGL((scale_k<float>), dim3(grid_for(n, 1)), dim3(256), 0, (hipStream_t)stream, (float*)x, (long long)n, scale);
  else if (dtype == PERO_BF16) hipLaunchKernelGGL((scale_k<bf16raw>), dim3(grid_for(n, 1)), dim3(256), 0, (hipStream_t)stream, (bf16raw*)x, (long long)n, scale);
  else PERO_REQUIRE(false, "pero_scale: bad dtype");
  PERO_CHECK_LAUNCH("pero_scale");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam over one flat buffer.  Algorithmic traffic: 16 B read + 12 B written per parameter (+2 B bf16 copy).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_k(float* p, const float* g, float* m, float* v, bf16raw* pb, long long n, float lr_bc1,
                                              float b1, float b2, float omb1, float omb2, float eps, float inv_sqrt_bc2, float gscale) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f4v pp = *(f4v*)(p + i), gg = *(const f4v*)(g + i), mm = *(f4v*)(m + i), vv = *(f4v*)(v + i);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float ge = gg[e] * gscale;
        mm[e] = mm[e] * b1 + omb1 * ge;
        vv[e] = vv[e] * b2 + omb2 * ge * ge;
        const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
        pp[e] = pp[e] - lr_bc1 * (mm[e] / denom);
      }
      *(f4v*)(p + i) = pp; *(f4v*)(m + i) = mm; *(f4v*)(v + i) = vv;
      if (pb) { uint2 o; o.x = pack2bf(pp[0], pp[1]); o.y = pack2bf(pp[2], pp[3]); *(uint2*)(pb + i) = o; }
    } else {
      for (long long j = i; j < n; j++) {
        const float ge = g[j] * gscale;
        const float mj = m[j] * b1 + omb1 * ge;
        const float vj = v[j] * b2 + omb2 * ge * ge;
        m[j] = mj; v[j] = vj;
        p[j] = p[j] - lr_bc1 * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
        if (pb) pb[j] = f2bf(p[j]);
      }
    }
  }
}
extern "C" int pero_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, double lr, double beta1,
                              double beta2, double eps, int64_t step, double grad_scale, void* stream) {
  PERO_REQUIRE(p && g && m && v && n > 0 && step >= 1, "pero_adam_step: bad arguments");
  PERO_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (!p_bf16 || (((uintptr_t)p_bf16) & 7) == 0), "pero_adam_step: alignment");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_k, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16raw*)p_bf16, (long long)n,
                     (float)(lr / bc1), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(1.0 / sqrt(bc2)), (float)grad_scale);
  PERO_CHECK_LAUNCH("pero_adam_step");
  return PERO_OK;
}

// ---------------------------------------------------------------------------------------------
// row gather / scatter-add (boolean-mask row selections of the losses; d % 8 == 0 fast path not needed:
// one wave per row, scalar elements, rows are >= 96 bytes)
// ---------------------------------------------------------------------------------------------
template <typename T, bool V16>
__global__ __launch_bounds__(256) void gather_rows_k(const T* src, const int64_t* index, T* dst, long long n_idx, long long n_out, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_out) return;
  const int lane = threadIdx.x & 63;
  if (V16) {  // rows of whole 16-byte pieces (d * sizeof(T) % 16 == 0, aligned bases): one uint4 per lane and step
    const int n16 = d * (int)sizeof(T) / 16;
    uint4* o = (uint4*)(dst + row * d);
    if (row < n_idx) {
      const uint4* s = (const uint4*)(src + index[row] * d);
      for (int c = lane; c < n16; c += 64) o[c] = s[c];
    } else {
      for (int c = lane; c < n16; c += 64) o[c] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  if (row < n_idx) {
    const T* s = src + index[row] * d;
    for (int c = lane; c < d; c += 64) dst[row * d + c] = s[c];
  } else {
    for (int c = lane; c < d; c += 64) Elem<T>::st(dst + row * d + c, 0.f);
  }
}
template <typename T, bool V16>
__global__ __launch_bounds__(256) void scatter_add_rows_k(const T* src, const int64_t* index, T* dst, long long n_idx, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_idx) return;
  const int lane = threadIdx.x & 63;
  T* o = dst + index[row] * d;
  if (V16) {
    for (int c = lane * 8; c < d; c += 512) {
      float a[8], b[8];
      load8<T>(o + c, a); load8<T>(src + row * d + c, b);
#pragma unroll
      for (int e = 0; e < 8; e++) a[e] += b[e];
      store8<T>(o + c, a);
    }
    return;
  }
  for (int c = lane; c < d; c += 64) Elem<T>::st(o + c, Elem<T>::ld(o + c) + Elem<T>::ld(src + row * d + c));
}
extern "C" int pero_gather_rows(const void* src, const int64_t* index, void* dst, int64_t n_idx, int64_t n_rows_out, int64_t d,
                                int dtype, void* stream) {
  PERO_REQUIRE(src && dst && (index || n_idx == 0) && n_rows_out >= n_idx && n_rows_out > 0 && d > 0, "pero_gather_rows: bad arguments");
  dim3 grid((unsigned)((n_rows_out + 3) / 4)), block(256);
  const bool v16 = d % 8 == 0 && aligned16(src) && aligned16(dst);
#define G_(T, V) hipLaunchKernelGGL((gather_rows_k<T, V>), grid, block, 0, (hipStream_t)stream, (const T*)src, index, (T*)dst, (long long)n_idx, (long long)n_rows_out, (int)d)
  if (dtype == PERO_F32) { if (v16) G_(float, true); else G_(float, false); }
  else if (dtype == PERO_BF16) { if (v16) G_(bf16raw, true); else G_(bf16raw, false); }
  else PERO_REQUIRE(false, "pero_gather_rows: bad dtype");
#undef G_
  PERO_CHECK_LAUNCH("pero_gather_rows");
  return PERO_OK;
}
extern "C" int pero_scatter_add_rows(const void* src, const int64_t* index, void* dst, int64_t n_idx, int64_t d, int dtype, void* stream) {
  PERO_REQUIRE(src && dst && index && n_idx > 0 && d > 0, "pero_scatter_add_rows: bad arguments");
  dim3 grid((unsigned)((n_idx + 3) / 4)), block(256);
  const bool v16 = d % 8 == 0 && aligned16(src) && aligned16(dst);
#define S_(T, V) hipLaunchKernelGGL((scatter_add_rows_k<T, V>), grid, block, 0, (hipStream_t)stream, (const T*)src, index, (T*)dst, (long long)n_idx, (int)d)
  if (dtype == PERO_F32) { if (v16) S_(float, true); else S_(float, false); }
  else if (dtype == PERO_BF16) { if (v16) S_(bf16raw, true); else S_(bf16raw, false); }
  else PERO_REQUIRE(false, "pero_scatter_add_rows: bad dtype");
#undef S_
  PERO_CHECK_LAUNCH("pero_scatter_add_rows");
  return PERO_OK;
}

// out[m][b] = sum over columns 128b .. 128b+127 of x[m][c] * y[m][c]  (bf16): the attention backward's D per head when the
// product that writes dO ran on a kernel without the PERO_GEMM_ROWDOT epilogue.  One wave per (row, block): 2 columns per lane.
__global__ __launch_bounds__(256) void rowdot_blocks_k(const bf16raw* x, const bf16raw* y, float* out, long long rows, int nblk,
                                                      long long ldx, long long ldy) {
  const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rows * nblk) return;
  const long long row = item / nblk;
  const int b = (int)(item - row * nblk), lane = threadIdx.x & 63;
  const unsigned xv = *(const unsigned*)(x + row * ldx + 128 * b + 2 * lane);
  const unsigned yv = *(const unsigned*)(y + row * ldy + 128 * b + 2 * lane);
  float s = __uint_as_float(xv << 16) * __uint_as_float(yv << 16) + __uint_as_float(xv & 0xffff0000u) * __uint_as_float(yv & 0xffff0000u);
  s = wave_sum(s);
  if (lane == 0) out[item] = s;
}
extern "C" int pero_rowdot_blocks(const void* x, const void* y, float* out, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy,
                                  void* stream) {
  PERO_REQUIRE(x && y && out && rows > 0 && cols > 0 && cols % 128 == 0 && ldx % 2 == 0 && ldy % 2 == 0, "pero_rowdot_blocks: bad arguments");
  const long long items = rows * (cols / 128);
  hipLaunchKernelGGL(rowdot_blocks_k, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)x,
                     (const bf16raw*)y, out, (long long)rows, (int)(cols / 128), (long long)ldx, (long long)ldy);
  PERO_CHECK_LAUNCH("pero_rowdot_blocks");
  return PERO_OK;
}

// --------------------------------------------------------------------------------------------
// Transposed bf16 weight copies, every matrix of a flat buffer in ONE launch.  The input gradient dX = dY W reads W
// K-contiguous from such a copy ([in][out]) instead of k-major from W itself - the 256x256x64 tile kernels run 10-20 %
// faster on K-contiguous operands.  table[t] = {src offset, dst offset, rows, cols, first tile}
// (elements / tiles of 64 x 64); a workgroup finds its matrix by scanning the <= a-few-dozen entries.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_multi_k(const bf16raw* src, bf16raw* dst, const long long* table, int n) {
  __shared__ bf16raw tile[64][66];  // pitch 33 dwords: a column walk touches 32 different banks
  const long long bid = blockIdx.x;
  int t = 0;
  while (t + 1 < n && table[(t + 1) * 5 + 4] <= bid) t++;
  const long long so = table[t * 5], dof = table[t * 5 + 1], rows = table[t * 5 + 2], cols = table[t * 5 + 3];
  const long long local = bid - table[t * 5 + 4];
  const long long tiles_c = (cols + 63) / 64;
  const long long r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
  const bf16raw* s = src + so;
  bf16raw* d = dst + dof;
  const int tid = threadIdx.x, c8 = (tid & 7) * 8;
  const bool vec = (rows % 8 == 0) && (cols % 8 == 0) && (so % 8 == 0) && (dof % 8 == 0);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int r = (tid >> 3) + 32 * h;
    if (r0 + r < rows) {
      if (vec && c0 + c8 + 8 <= cols) {
        const uint4 v = *(const uint4*)(s + (r0 + r) * cols + c0 + c8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) { tile[r][c8 + 2 * e] = (bf16raw)(w[e] & 0xffffu); tile[r][c8 + 2 * e + 1] = (bf16raw)(w[e] >> 16); }
      } else {
        for (int e = 0; e < 8; e++)
          if (c0 + c8 + e < cols) tile[r][c8 + e] = s[(r0 + r) * cols + c0 + c8 + e];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int c = (tid >> 3) + 32 * h;  // output row = source column
    if (c0 + c < cols) {
      if (vec && r0 + c8 + 8 <= rows) {
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; e++) w[e] = (unsigned)tile[c8 + 2 * e][c] | ((unsigned)tile[c8 + 2 * e + 1][c] << 16);
        *(uint4*)(d + (c0 + c) * rows + r0 + c8) = make_uint4(w[0], w[1], w[2], w[3]);
      } else {
        for (int e = 0; e < 8; e++)
          if (r0 + c8 + e < rows) d[(c0 + c) * rows + r0 + c8 + e] = tile[c8 + e][c];
      }
    }
  }
}
extern "C" int pero_transpose_multi(const void* src, void* dst, const int64_t* table, int64_t n_matrices, int64_t total_tiles, void* stream) {
  PERO_REQUIRE(src && dst && table && n_matrices > 0 && total_tiles > 0 && total_tiles < (1LL << 31), "pero_transpose_multi: bad arguments");
  PERO_REQUIRE(aligned16(src) && aligned16(dst), "pero_transpose_multi: 16-byte alignment");
  hipLaunchKernelGGL(transpose_multi_k, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16raw*)src, (bf16raw*)dst,
                     (const long long*)table, (int)n_matrices);
  PERO_CHECK_LAUNCH("pero_transpose_multi");
  return PERO_OK;
}
