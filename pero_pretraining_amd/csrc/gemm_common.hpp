// Shared by the bf16 tile GEMM kernels (gemm.hip, gemm_o.hip, gemm_r.hip, gemm_e.hip).
#pragma once
#include "common.hpp"

struct GemmP {
  const void* A; const void* B; void* C;
  const float* bias; const void* resid; const void* gate;
  long long M, N, K, lda, ldb, ldc, ldr, ldg;
  long long sAo, sAi, sBo, sBi, sCo, sCi;
  int binner; float alpha; int flags; long long kchunk;
};


// K-major LDS images: 32-byte blocks of a k-row are XORed with fk(krow) so that the two transposed 8-byte reads of
// a fragment (k-rows 8g+q and 8g+q+4 of the 4 lane groups) hit 32 distinct 8-byte slots of the 256-byte bank row.
__device__ __forceinline__ int fk(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  if (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");  // keep later LDS accesses below the barrier
}

// host entry of the one-tile-per-workgroup 128x128x64 kernel (gemm_o.hip): split-K weight gradients of short reductions
bool pero_launch_gemm_o128(const GemmP& p, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st);
// host entry of the 256x128x32 eight-wave kernel (gemm_r.hip): stored products of small batches
bool pero_launch_gemm_r256(const GemmP& p, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st);
// host entry of the eight-phase ping-pong persistent 256x256x64 kernel (gemm_e.hip); var < 0: the "gemm_e_var" option
// ws / ws_bytes: the caller's workspace for split-K partial tiles (pero_gemm's `workspace`); null: f32 atomics
bool pero_launch_gemm_e256(const GemmP& p, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st, int var = 0,
                           void* ws = nullptr, long long ws_bytes = 0);
long long pero_gemm_e256_splitk_ws_bytes(long long M, long long N, long long K, int k_split);
// host entry of the row-complete 128 x 512 tile (gemm_e.hip, opt-in "gemm_nw"): N = 512 stored products with the plain / residual epilogue
bool pero_launch_gemm_n512(const GemmP& p, long long batch, bool ta, bool tb, bool out_f32, hipStream_t st);
// host entry of the 256 x 128 tile with two workgroups per CU (gemm_e.hip, opt-in "gemm_d128" = the largest K / 64 that takes it)
bool pero_launch_gemm_d128(const GemmP& p, long long batch, bool ta, bool tb, bool out_f32, hipStream_t st);
bool pero_launch_gemm_n512_ln(const GemmP& p, void* t, long long ldt, float* mean, float* rstd, const float* gamma, const float* beta, float eps,
                              hipStream_t st);
bool pero_launch_gemm_n512_lnb(const GemmP& p, const void* t, long long ldt, const float* rstd, const float* gamma, const float* beta, float* work,
                               int* grid_out, hipStream_t st);
// rowops.hip: dgamma / dbeta / dxsum (each may be null) += the sum of the `nblocks` partial rows of work [3][nblocks][d]
void pero_ln_bwd_reduce_launch(const float* work, float* dgamma, float* dbeta, float* dxsum, int nblocks, int d, hipStream_t st);
