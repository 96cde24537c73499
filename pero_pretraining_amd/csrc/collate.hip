// Batch collation on the device (reference: common/dataloader.py:68-155
// BatchCreator.stack_images - host numpy: zero-filled (B, H, Wt, C) arrays, per-line slice copies, per-line mask loops).
//
// The host keeps only the decisions that consume its RNG stream (the left paddings, drawn exactly like the reference so
// that a seeded run collates identically) and uploads the RAGGED lines back to back: sum(w_i) * H * C bytes instead of
// B * Wt * H * C.  One kernel writes the padded uint8 batch (each output byte exactly once: line pixels or zero), one
// tiny kernel derives the four position masks.  Pure byte / integer work, HBM-bound, bit-exact.
#include "common.hpp"

// out (B, H, Wt, C) u8.  Row (b, h) = Wt*C bytes: zeros, except [lp*C, (lp + w)*C) <- packed[off[b] + h*w*C ...].
// A thread produces 16 output bytes (one uint4 store; rows are 16-byte multiples because Wt % 32 == 0, and the batch
// base is 16-byte aligned).  Source bytes are unaligned (line widths are arbitrary): two aligned dword loads + one
// v_alignbyte per output dword.  The packed buffer must be readable 8 bytes past its end (the wrapper allocates that).
__global__ __launch_bounds__(256) void stack_lines_k(const unsigned char* packed, const int64_t* off, const int* widths,
                                                     const int* left_px, unsigned char* out, int H, long long row_bytes, int C,
                                                     long long chunks_per_row) {
  const long long chunk = (long long)blockIdx.x * 256 + threadIdx.x;
  if (chunk >= chunks_per_row) return;
  const int h = blockIdx.y, b = blockIdx.z;
  const long long w_bytes = (long long)widths[b] * C;
  const long long lo = (long long)left_px[b] * C, hi = lo + w_bytes;  // line span inside the row, in bytes
  const unsigned char* src = packed + off[b] + (long long)h * w_bytes - lo;  // src[j] is the line byte at row byte j
  const long long j0 = chunk * 16;
  unsigned v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const long long j = j0 + 4 * q;
    unsigned word = 0;
    if (j >= lo && j + 4 <= hi) {
      const unsigned long long a = (unsigned long long)(src + j);
      const unsigned* ap = (const unsigned*)(a & ~3ULL);
      word = __builtin_amdgcn_alignbyte(ap[1], ap[0], (unsigned)(a & 3));
    } else if (j + 4 > lo && j < hi) {  // straddles an end of the line
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (j + e >= lo && j + e < hi) word |= (unsigned)src[j + e] << (8 * e);
    }
    v[q] = word;
  }
  *(uint4*)(out + ((long long)b * H + h) * row_bytes + j0) = make_uint4(v[0], v[1], v[2], v[3]);
}

// image_masks (B, S) u8 = 1 on the label positions that contain the line (dataloader.py:92-96);
// shift_masks (dataloader.py:124-138): shift = crop_shift + lp1 - lp2; sm1 = 1 on [shift, S) (shift >= 0) or on
// [0, S + shift) (shift < 0); sm2 = reverse(sm1); positions that are 1 but outside the view's own image mask become 2.
__global__ __launch_bounds__(256) void line_masks_k(const int* widths1, const int* widths2, const int* lp1, const int* lp2,
                                                    const int* crop_shifts, unsigned char* im1, unsigned char* im2,
                                                    unsigned char* sm1, unsigned char* sm2, int* shifts, int B, int S, int sub) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * S) return;
  const int b = idx / S, s = idx % S;
  const int a1 = lp1[b], e1 = a1 + (widths1[b] + sub - 1) / sub;
  const unsigned char m1 = (s >= a1 && s < e1) ? 1 : 0;
  im1[idx] = m1;
  if (!im2) return;
  const int a2 = lp2[b], e2 = a2 + (widths2[b] + sub - 1) / sub;
  const unsigned char m2 = (s >= a2 && s < e2) ? 1 : 0;
  im2[idx] = m2;
  const int shift = (crop_shifts ? crop_shifts[b] : 0) + a1 - a2;
  if (s == 0) shifts[b] = shift;
  // python slices clamp: [:shift] with shift < -S is empty, [shift:] with shift > S is empty
  const bool on1 = shift < 0 ? (s < S + shift) : (s >= shift);
  const int sr = S - 1 - s;  // sm2[s] = sm1[S - 1 - s]
  const bool on2 = shift < 0 ? (sr < S + shift) : (sr >= shift);
  sm1[idx] = on1 ? (m1 ? 1 : 2) : 0;
  sm2[idx] = on2 ? (m2 ? 1 : 2) : 0;
}

extern "C" int pero_stack_lines(const void* packed, const int64_t* offsets, const int32_t* widths, const int32_t* left_px,
                                void* out, int64_t B, int64_t H, int64_t Wt, int64_t C, void* stream) {
  PERO_REQUIRE(packed && offsets && widths && left_px && out, "pero_stack_lines: null pointer");
  PERO_REQUIRE(B > 0 && B < 65536 && H > 0 && H < 65536 && C > 0 && Wt > 0, "pero_stack_lines: bad sizes");
  PERO_REQUIRE((Wt * C) % 16 == 0 && aligned16(out), "pero_stack_lines: padded rows must be 16-byte multiples (Wt*C = %lld)",
               (long long)(Wt * C));
  const long long row_bytes = Wt * C, chunks = row_bytes / 16;
  hipLaunchKernelGGL(stack_lines_k, dim3((unsigned)((chunks + 255) / 256), (unsigned)H, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, (const unsigned char*)packed, offsets, widths, left_px, (unsigned char*)out, (int)H,
                     row_bytes, (int)C, chunks);
  PERO_CHECK_LAUNCH("pero_stack_lines");
  return PERO_OK;
}

extern "C" int pero_line_masks(const int32_t* widths1, const int32_t* widths2, const int32_t* left1, const int32_t* left2,
                               const int32_t* crop_shifts, uint8_t* image_masks1, uint8_t* image_masks2, uint8_t* shift_masks1,
                               uint8_t* shift_masks2, int32_t* shifts, int64_t B, int64_t S, int64_t subsampling, void* stream) {
  PERO_REQUIRE(widths1 && left1 && image_masks1, "pero_line_masks: null pointer");
  PERO_REQUIRE(!image_masks2 || (widths2 && left2 && shift_masks1 && shift_masks2 && shifts),
               "pero_line_masks: the second view needs widths2, left2, both shift masks and shifts");
  PERO_REQUIRE(B > 0 && S > 0 && subsampling > 0 && B * S < (1LL << 31), "pero_line_masks: bad sizes");
  hipLaunchKernelGGL(line_masks_k, dim3((unsigned)((B * S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, widths1, widths2,
                     left1, left2, crop_shifts, image_masks1, image_masks2, shift_masks1, shift_masks2, shifts, (int)B, (int)S,
                     (int)subsampling);
  PERO_CHECK_LAUNCH("pero_line_masks");
  return PERO_OK;
}
