// Shared device/host helpers for libpero_hip (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/pero_hip.h"

typedef unsigned short bf16raw;                                    // bf16 bit pattern in memory
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));           // MFMA bf16 operand fragment
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// Two transposed 8-byte LDS reads (ds_read_b64_tr_b16) -> one 8 x bf16 MFMA operand.  The halves are joined by a vector
// CONCATENATION of two 64-bit values: assembling the operand element by element from the 4 x i16 results made the
// compiler copy every dword (4 v_mov_b32 per fragment: 64 per k-step in the K-major GEMM loops).
__device__ __forceinline__ bf8v lds_tr16_pair(const unsigned char* a, const unsigned char* b) {
  typedef int i2v __attribute__((ext_vector_type(2)));
  typedef int i4v __attribute__((ext_vector_type(4)));
  const i2v lo = __builtin_bit_cast(i2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, a)));
  const i2v hi = __builtin_bit_cast(i2v, __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s4v, b)));
  const i4v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
  return __builtin_bit_cast(bf8v, v);
}

__device__ __forceinline__ float bf2f(bf16raw b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ bf16raw f2bf(float f) {  // round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16raw, h);
}
// two floats -> one dword of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  (Two scalar casts joined by shift / or compile to
// two conversions + a shift + an SDWA or: 4 instructions per dword in every epilogue.)
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  typedef float pk_f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pk_b2 __attribute__((ext_vector_type(2)));
  const pk_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_b2));
}

// eight consecutive elements as one (bf16) or two (f32) 16-byte accesses, f32 values in registers
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const f4v a = *(const f4v*)p, b = *(const f4v*)(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16raw>(const bf16raw* p, float (&v)[8]) {
  const uint4 r = *(const uint4*)p;
  const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int e = 0; e < 4; e++) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *(f4v*)p = (f4v){v[0], v[1], v[2], v[3]};
  *(f4v*)(p + 4) = (f4v){v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<bf16raw>(bf16raw* p, const float (&v)[8]) {
  uint4 o;
  o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
  *(uint4*)p = o;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16raw> {
  static __device__ __forceinline__ float ld(const bf16raw* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16raw* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// host side ------------------------------------------------------------------------------------
void pero_set_error(const char* fmt, ...);
#define PERO_REQUIRE(cond, ...)                   \
  do {                                            \
    if (!(cond)) {                                \
      pero_set_error(__VA_ARGS__);                \
      return PERO_E_INVALID;                      \
    }                                             \
  } while (0)
#define PERO_CHECK_LAUNCH(name)                                                     \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      pero_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));         \
      return PERO_E_LAUNCH;                                                         \
    }                                                                               \
  } while (0)
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// Host-side state of the library is limited to these two immutable, once-initialised values (C++11 static initialisation is
// thread-safe: the entry points are called from autograd worker threads) and the pero_set_option knobs.
//  * a kernel's dynamic-LDS limit above 64 KiB, set once per process and kernel
#define PERO_LDS_ATTR(kernel_, bytes_)                                                                                              \
  do {                                                                                                                              \
    static const hipError_t attr_once_ = hipFuncSetAttribute((const void*)(kernel_), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes_)); \
    (void)attr_once_;                                                                                                               \
  } while (0)
//  * the CU count of the process's device (one process per GPU)
static inline int pero_num_cus() {
  static const int n = [] {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}
