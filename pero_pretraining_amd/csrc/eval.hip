// Evaluation-side kernels (masked_pretraining/tester.py).
//
// The reference's Tester copies the whole (B, S, V) logit tensor to the host and runs numpy argmax / argsort per row
// (tester.py:72-91) - 1 GiB per batch at B = 256, V = 4096.  Here the rank of the label inside its row is computed
// where the logits live, and the top-k error counters are accumulated on the device with integer atomics
// (deterministic), so a whole test() loop needs one 8-byte-per-counter read-back at the end.
#include "common.hpp"

// counters[0] += number of rows with mask == 1; counters[1 + i] += rows whose label is NOT among the top ks[i] logits.
// Tie rules (value comparisons in f32; a bf16 -> f32 widening is exact):
//   k == 1 : numpy argmax = FIRST maximum  -> error iff some j has logit > x, or logit == x with j < label
//   k  > 1 : numpy argsort(...)[:, -k:] (tester.py:94-96); among equal values a stable ascending sort puts higher
//            indices last, i.e. they rank higher: error iff #{logit > x} + #{logit == x, j > label} >= k
//            (numpy's default introsort leaves the order of equal elements unspecified; exact ties do not occur
//            with real-valued logits and are pinned here by this rule).
// ranks (optional, int32 [rows][3]): gt, eq_lo, eq_hi per row with mask == 1, -1 otherwise.
template <typename T>
__global__ __launch_bounds__(256) void label_rank_k(const T* logits, long long ld, const int64_t* labels, const int64_t* mask,
                                                   long long rows, int V, const int* ks, int nk, unsigned long long* counters,
                                                   int* ranks) {
  __shared__ int red[4][3];
  const long long row = blockIdx.x;
  const int tid = threadIdx.x;
  if (mask[row] != 1) {
    if (ranks && tid < 3) ranks[row * 3 + tid] = -1;
    return;
  }
  const long long lab = labels[row];
  const T* lr = logits + row * ld;
  int gt = 0, lo = 0, hi = 0;
  if (lab >= 0 && lab < V) {
    const float x = Elem<T>::ld(lr + lab);
    for (int c = tid; c < V; c += 256) {
      const float v = Elem<T>::ld(lr + c);
      gt += v > x;
      lo += (v == x) & (c < lab);
      hi += (v == x) & (c > lab);
    }
  } else {
    gt = tid == 0 ? V : 0;  // label outside the row (padding label under a set mask bit): never predicted
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gt += __shfl_xor(gt, o, 64);
    lo += __shfl_xor(lo, o, 64);
    hi += __shfl_xor(hi, o, 64);
  }
  if ((tid & 63) == 0) { red[tid >> 6][0] = gt; red[tid >> 6][1] = lo; red[tid >> 6][2] = hi; }
  __syncthreads();
  if (tid == 0) {
    gt = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    lo = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    hi = red[0][2] + red[1][2] + red[2][2] + red[3][2];
    if (ranks) { ranks[row * 3 + 0] = gt; ranks[row * 3 + 1] = lo; ranks[row * 3 + 2] = hi; }
    if (counters) {
      atomicAdd(counters, 1ULL);
      for (int i = 0; i < nk; i++) {
        const int k = ks[i];
        const bool err = k == 1 ? (gt + lo) >= 1 : (gt + hi) >= k;
        if (err) atomicAdd(counters + 1 + i, 1ULL);
      }
    }
  }
}

extern "C" int pero_label_rank(const void* logits, int64_t ld, const int64_t* labels, const int64_t* mask, int64_t rows,
                               int64_t V, const int32_t* ks, int32_t nk, uint64_t* counters, int32_t* ranks, int dtype,
                               void* stream) {
  PERO_REQUIRE(logits && labels && mask, "pero_label_rank: null pointer");
  PERO_REQUIRE(counters || ranks, "pero_label_rank: nothing to produce (counters and ranks are both null)");
  PERO_REQUIRE(rows > 0 && V > 0 && V < (1LL << 31) && ld >= V && rows < (1LL << 31), "pero_label_rank: bad sizes");
  PERO_REQUIRE(nk >= 0 && nk <= PERO_MAX_TOPK && (nk == 0 || ks), "pero_label_rank: 0..%d measured errors", PERO_MAX_TOPK);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PERO_F32)
    hipLaunchKernelGGL((label_rank_k<float>), dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, (long long)ld, labels,
                       mask, (long long)rows, (int)V, (const int*)ks, (int)nk, (unsigned long long*)counters, (int*)ranks);
  else if (dtype == PERO_BF16)
    hipLaunchKernelGGL((label_rank_k<bf16raw>), dim3((unsigned)rows), dim3(256), 0, st, (const bf16raw*)logits, (long long)ld,
                       labels, mask, (long long)rows, (int)V, (const int*)ks, (int)nk, (unsigned long long*)counters, (int*)ranks);
  else
    PERO_REQUIRE(false, "pero_label_rank: bad dtype");
  PERO_CHECK_LAUNCH("pero_label_rank");
  return PERO_OK;
}
