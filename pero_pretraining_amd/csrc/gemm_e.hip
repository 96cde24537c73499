// PERSISTENT 256 x 256 x 64 bf16 tile GEMM on the EIGHT-PHASE PING-PONG schedule (cdna_hip_programming.md section 5, "The 256^2
// 8-phase template"): eight waves as 2 (M) x 4 (N), 128 x 64 outputs per wave, one workgroup per CU.
//
//  * A K-tile (256 x 64 of A, 256 x 64 of B) lives in LDS as FOUR half tiles of 16 KiB (A0, A1, B0, B1 = 128 rows x 64 k);
//    a wave owns 64 rows of each A half and 32 columns of each B half, i.e. its 128 x 64 outputs are 2 x 2 blocks of
//    64 x 32 = (A half) x (B half).  One PHASE = one block x K = 64 = 16 MFMAs; four phases per K-tile:
//        P1: read B0 (4 ds_read_b128), A0 (8)  -> A0 x B0        P3: read A1 (8) -> A1 x B1
//        P2: read B1 (4)                       -> A0 x B1        P4: (nothing)   -> A1 x B0
//    so a half tile is read exactly once per wave and at most 64 fragment registers are live beside the 128 accumulators.
//  * Every phase is  { fragment reads ; LDS-DMA of ONE half tile (2 x global_load_lds_dwordx4) ; s_barrier ; lgkmcnt(0) ;
//    s_setprio 1 ; 16 MFMA ; s_setprio 0 ; s_barrier }.  Waves 4-7 take one extra barrier before the loop: the two waves of
//    a SIMD run half a phase apart, one in its MFMA cluster while the other reads / stages (the ping-pong).
//  * The LDS-DMA stream is ONE sequence over (tile, K-tile) pairs, three half tiles ahead of the reads, never drained:
//    a counted s_waitcnt vmcnt once per K-tile (phase 4), raw s_barrier, lgkmcnt only.  Issue order B0(t+2) [P2], A0(t+2)
//    [P3], B1(t+2) [P4], A1(t+2) [P1 of t+1]: a slot is restaged two phases after its last read (B0: one phase, its reads are
//    retired by the lgkmcnt(8) in front of P1's first barrier); the wait in P4 of K-tile t retires everything up to
//    A1(t+1), read from P1 of t+1 on (one phase after the wait: staggered waves need the extra barrier).
//  * Persistent: one workgroup per CU walks tiles bid, bid + G, ...; the next tile's first K-tiles are part of the same
//    stream, so the pipeline never refills.  vmcnt is ONE in-order counter for LDS-DMA, loads and stores: the last
//    half tile the next tile needs before its second K-tile (A1 of K-tile 1) is issued AHEAD of the epilogue's stores, and
//    the first K-tile's wait counts them out - the stores then have two K-tiles of main loop to drain.
//  * Epilogue straight from the accumulators (no LDS memory): operands are swapped in the MFMA so that a lane owns 4 consecutive
//    output columns of one row; v_permlane16_swap between the two 16 x 16 tiles of a 32-column block gives it 8 consecutive
//    columns (16 bytes), and ONE lane transpose (4 x ds_bpermute_b32) moves the four 16-byte pieces of a row from lanes 16 apart
//    onto four ADJACENT lanes before the store: the memory pipeline merges adjacent lanes only, and a 1 KiB store whose adjacent
//    lanes sit on different rows is 64 separate requests (64 instead of 16 cycles of the CU's store path; tools/probe_store3.hip).
//    Side inputs (residual rows, the second matrix of the row dots, the ReLU bit mask) are fetched in that transposed layout by
//    inline-asm buffer loads the compiler does not count: half of them in phase 4 of the tile's last K-tile, half at the start
//    of the epilogue, each waited for by a counted vmcnt.  The bias comes from a per-wave 1 KiB LDS copy of the tile's bias row,
//    fetched by one more LDS-DMA of the same stream at the tile's start (an ordinary load would have to wait for every LDS-DMA
//    issued before it) and added in f32 before the rounding, exactly as the other tile kernels do (acc * alpha + bias): results
//    are bit-identical across kernels, i.e. across batch sizes.
//  * Both wave groups run their epilogues side by side: waves 0-3 take one extra barrier at the epilogue's start (waves 4-7 finish
//    their last MFMA cluster meanwhile), waves 4-7 one in front of the next tile, which staggers the groups again.  Staggered
//    through the epilogue, each group sat at a barrier through the other's epilogue (tools/gemm_e_ktiles.py).
#include "gemm_common.hpp"
#include <type_traits>
#include <vector>

#define E_BM 256
#define E_BN 256
#define E_BK 64
#define E_HALF 16384               // half tile: 128 rows x 64 k (bf16)
#define E_KTILE (4 * E_HALF)       // A0 A1 B0 B1
#define E_RING (2 * E_KTILE)       // 128 KiB: two K-tiles
#define E_BIAS E_RING              // 8 x 1 KiB: each wave's copy of the tile's 256 bias floats
#define E_LUT (E_RING + 8192 + 512)           // EP_GATE_BITS: 256 x 16 B, mask byte -> the four AND masks of its 8 bf16 columns
#define E_XSTG (E_RING + 8192 + 512 + 4096)    // 8 x 2 KiB: each wave's staging image of the epilogue's lane transpose
#define E_LDS_BYTES (E_XSTG + 8 * 2048)        // + 512 B of phase stamps (diagnostic build VAR 64) + the LUT + the staging
#ifndef E_XCHG_LDS
#define E_XCHG_LDS 1                           // 0: the lane transpose as 4 x ds_bpermute_b32 per unit (round 2)
#endif

// epilogue modes
#define EP_PLAIN 0       // bias
#define EP_RELU 1        // bias, ReLU
#define EP_RESID 2       // bias, + residual (bf16 rows of C's shape)
#define EP_RELU_BITS 3   // bias, ReLU, `gate` receives the bit mask (stored C > 0)
#define EP_GATE_BITS 4   // `gate` bit mask applied; with PERO_GEMM_COLSUM the column sums of the stored result are added to `bias`
#define EP_ROWDOT 6      // `bias`[m][n / 128] += row dots of the stored result with `gate` (bf16 rows of C's shape)
#define EP_SPLITK 7      // f32 C += partial product of ONE k-slice (atomics): one work item (tile, k-slice) per workgroup, not persistent
#define EP_RESID_LN 8    // (gemm_bf16_n512 only) bias, + residual -> y stored; LayerNorm of the stored rows -> t, mean, rstd (LnP)
#define EP_RESID_LN_T 9  // the same without the stores of y (the backward reads t: pero_layernorm_bwd_out): the epilogue's counted waits differ
#define EP_RESID_LNB 10  // (gemm_bf16_n512 only) input gradient of a Linear + residual gradient = dt, rows complete -> LayerNorm BACKWARD of the upstream norm
                         // in the epilogue (from its output t and rstd, pero_layernorm_bwd_out's arithmetic): dx stored, dt never; column sums -> LnP.work
// second argument of gemm_bf16_n512: what the LayerNorm epilogue writes and reads beside GemmP
struct LnP { void* t; long long ldt; float* mean; float* rstd; const float* gamma; const float* beta; float eps; float* work; };
typedef float ef2v __attribute__((ext_vector_type(2)));
#define E_BLOAD4(dst_, voff_, rs_, soff_, imm_) \
  asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(dst_) : "v"(voff_), "s"(rs_), "s"(soff_), "i"(imm_) : "memory")

typedef int ei4v __attribute__((ext_vector_type(4)));
typedef short es2v __attribute__((ext_vector_type(2)));
typedef unsigned short eus2v __attribute__((ext_vector_type(2)));
// two bf16 in a dword: ReLU as a signed 16-bit maximum with zero; (half != 0) per half for halves that are +0 or positive
__device__ __forceinline__ unsigned epk_relu(unsigned v) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(es2v, v), (es2v){0, 0}));
}
// acc + a.lo * b.lo + a.hi * b.hi of two bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float edot2(unsigned a, unsigned b, float acc) {
  typedef __bf16 eb2v __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(eb2v, a), __builtin_bit_cast(eb2v, b), acc, false);
}
__device__ __forceinline__ unsigned epk_nonzero(unsigned v) {   // (hipcc turns the vector minimum into compares and selects: asm)
  unsigned r;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(v), "s"(0x00010001u));
  return r;
}
typedef unsigned eu4v __attribute__((ext_vector_type(4)));
typedef unsigned eu2v __attribute__((ext_vector_type(2)));

// per-thread byte offset of its two LDS-DMA pieces inside a half tile's source (constant over the whole kernel).
// Rows of the half tiles are INTERLEAVED so that a wave's outputs are contiguous in memory: row r of A half h is row
// (r >> 6) * 128 + 64 h + (r & 63) of the tile, row r of B half h is column (r >> 5) * 64 + 32 h + (r & 31).
template <bool TR, bool ISA>
__device__ __forceinline__ unsigned elane_off(long long ld, int tid) {
  if (!TR) {  // K-contiguous operand: piece = 8 rows x 128 B; LDS slot (tid & 7) of row r holds chunk slot ^ (r & 7)
    const int r = tid >> 3, chunk = (tid & 7) ^ (r & 7);
    const int g = ISA ? r : ((r >> 5) * 64 + (r & 31));
    return (unsigned)((g * ld + chunk * 8) * 2);
  } else {    // K-major operand: piece = 4 k-rows x 256 B; 32-byte blocks of a k-row XORed with fk(krow)
    const int krow = tid >> 4, slot = tid & 15;
    const int c = ((((slot >> 1) ^ fk(krow)) << 1) | (slot & 1)) * 8;
    const int g = ISA ? ((c >> 6) * 128 + (c & 63)) : ((c >> 5) * 64 + (c & 31));
    return (unsigned)((krow * ld + g) * 2);
  }
}
// byte steps of an operand's half-tile stream (uniform): tile origin t0, K-tile u, half h, second piece
template <bool TR, int HS>
struct EStep {
  long long tile, ktile, half, piece;
  __device__ __forceinline__ EStep(long long ld) {
    if (!TR) { tile = ld * 2; ktile = E_BK * 2; half = HS * ld * 2; piece = 128 * ld * 2; }
    else { tile = 2; ktile = E_BK * ld * 2; half = HS * 2; piece = 32 * ld * 2; }
  }
};
__device__ __forceinline__ void eglds2(const unsigned char* base, long long piece, unsigned off, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + piece + off),
                                   (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
}
// buffer descriptor from uniform values (raw buffer, 32-bit offsets, no bounds beyond `bytes`)
__device__ __forceinline__ ei4v ersrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  ei4v r;
  r[0] = (int)(unsigned)b; r[1] = (int)(unsigned)((b >> 32) & 0xffffu); r[2] = (int)bytes; r[3] = 0x00020000;
  return r;
}
// loads the compiler does not see (no wait of its own, not in its vmcnt bookkeeping): waited for by E_WAIT* below
#define E_BLOAD16(dst_, voff_, rs_, soff_, imm_) \
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(dst_) : "v"(voff_), "s"(rs_), "s"(soff_), "i"(imm_) : "memory")
#define E_BLOAD8(dst_, voff_, rs_, soff_, imm_) \
  asm volatile("s_nop 4\n\tbuffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(dst_) : "v"(voff_), "s"(rs_), "s"(soff_), "i"(imm_) : "memory")
// 16-byte store: its data registers are rewritten by the next unit right behind it.  hipcc (ROCm 7.2) pads that hazard only for
// a constant soffset; with the row offset in an SGPR the store sent stale dwords for some lanes (measured: lanes 12-15 of the
// second data dword) - the wait states are in the string
// ... and IN FRONT of it: the compiler does not look into the string, so nothing keeps a v_readlane_b32 that restores a spilled SGPR (the row
// offset, the descriptor) apart from the store that reads it - a vector-ALU write of an SGPR needs five wait states before a vector-memory
// instruction uses it, and a store issued too early takes the SGPR's OLD value: rows of the LayerNorm epilogue (87 spilled SGPRs) landed in
// other row groups, run-to-run different (E_BLOAD16 has had its s_nop 4 for the same reason)
#define E_BSTORE16(src_, voff_, rs_, soff_, imm_) \
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4\n\ts_nop 2" :: "v"(src_), "v"(voff_), "s"(rs_), "s"(soff_), "i"(imm_) : "memory")
#define E_BSTORE16_NT(src_, voff_, rs_, soff_, imm_) \
  asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:%4 nt\n\ts_nop 2" :: "v"(src_), "v"(voff_), "s"(rs_), "s"(soff_), "i"(imm_) : "memory")
#define E_WAIT8(n_, r_) \
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r_[0]), "+v"(r_[1]), "+v"(r_[2]), "+v"(r_[3]), "+v"(r_[4]), "+v"(r_[5]), "+v"(r_[6]), "+v"(r_[7]) : "i"(n_) : "memory")
#define E_WAIT4(n_, r_) \
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r_[0]), "+v"(r_[1]), "+v"(r_[2]), "+v"(r_[3]) : "i"(n_) : "memory")

template <int EPI> struct ECnt {
  // vector-memory operations of the epilogue, in issue order: side loads of rows 0-63 (phase 4 of the last K-tile), [A1 of the
  // next tile's K-tile 1], side loads of rows 64-127, then the stores / atomics of the two halves
  static constexpr int L0 = (EPI == EP_RESID || EPI == EP_ROWDOT) ? 8 : 0;  // (EP_GATE_BITS: its 8 mask loads go out in phase 1 of the last K-tile, ahead of A1(t+1): every later wait retires them)
  static constexpr int L1 = L0;
#ifndef E_DUP
#define E_DUP 0    // timing probe (-DE_DUP=1): every B0 half tile is requested twice - ten LDS-DMA instructions per wave and K-tile instead of eight, the
#endif             // load of the row-complete 128 x 512 tile of DESIGN.md "what comes next"; every counted wait leaves two more operations in flight
#define E_DX (2 * E_DUP)
#ifndef E_ABL
#define E_ABL 0    // timing-only ablation builds (results wrong by design): 1 no bit-mask stores (ReLU + bits), 2 no bit-mask loads (gate)
#endif
  static constexpr int S_HALF = 8 + ((EPI == EP_RELU_BITS && !(E_ABL & 1)) ? 4 : 0) + (EPI == EP_ROWDOT ? 4 : 0);
};

// Work-item order for split-K slice counts that are no multiple of 8 (e.g. 12 tiles x 21 slices).  Workgroup T runs on XCD T & 7;
// slice z belongs to XCD z % 8, and an XCD takes the (tile, slice) items of its own slices first, slice by slice, so that the tiles
// which stream the same operand panels meet in one L2; what an XCD has too many of (an XCD with three slices has 36 items for 31-32
// workgroups) goes to the XCDs with room.  In plain order (item = T) every XCD fetched every panel: K-tiles of 4.2 k cycles against
// 3.6 k with aligned slices.  Computed per workgroup from (tiles, slices) alone - a few scalar loops - so no table, no allocation
// and no host copy stand behind the product (round 2 kept a device table per shape in a process-wide cache).
__device__ __forceinline__ void esplitk_xcd_item(int T, int tiles, int nsl, int& id, int& z) {
  const int n = tiles * nsl, x = T & 7, k = T >> 3;
  auto cnt = [&](int xx) { return tiles * ((nsl - xx + 7) / 8); };   // items of the slices z = xx, xx + 8, ... < nsl
  auto cap = [&](int xx) { return (n - xx + 7) / 8; };               // workgroups T = xx, xx + 8, ... < n
  int kk = k, xo = x;
  if (k >= cnt(x)) {
    // one of this XCD's free places: the q-th of all free places (in XCD order) takes the q-th surplus item (in XCD order)
    int q = k - cnt(x);
    for (int xx = 0; xx < x; xx++) { const int dfc = cap(xx) - cnt(xx); q += dfc > 0 ? dfc : 0; }
    for (int xx = 0; xx < 8; xx++) {
      const int sp = cnt(xx) - cap(xx);
      if (sp <= 0) continue;
      if (q < sp) { xo = xx; kk = cap(xx) + q; break; }
      q -= sp;
    }
  }
  z = xo + 8 * (kk / tiles);
  id = kk % tiles;
}

template <bool TA, bool TB, int EPI, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_bf16_e256(GemmP p, int ks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef ECnt<EPI> CN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int ntn = (int)(p.N / E_BN);
  const int nt = (int)(p.M / E_BM) * ntn;
  const int G = gridDim.x;  // multiple of 8
  const int q8 = nt >> 3, r8 = nt & 7;
  const bf16raw* A = (const bf16raw*)p.A;
  const bf16raw* B = (const bf16raw*)p.B;
  int nk = (int)(p.K / E_BK);  // >= 2 (launcher); split-K: the work item's K-tiles (set below)
  const unsigned offA = elane_off<TA, true>(p.lda, tid), offB = elane_off<TB, false>(p.ldb, tid);
  const bool colsum = EPI == EP_GATE_BITS && (p.flags & PERO_GEMM_COLSUM);

  // stored products: `ks` = N-tiles of one row panel a workgroup walks ONE AFTER THE OTHER (0 / 1: none - the ntn workgroups of an XCD that share a row panel
  // run its ntn N-tiles side by side and all wait for the SAME bytes from HBM; `seq` > 1 leaves ntn / seq sharers per panel and seq x as many panels in flight)
  const int ksq = ks & 15;
  const bool inter = ks & 16;   // the side-by-side workgroups take ADJACENT N-tiles (n = r * sharers + j instead of j * seq + r)
  const int seq = (EPI != EP_SPLITK && ksq > 1 && r8 == 0 && ntn % ksq == 0 && (G >> 3) % (ntn / ksq) == 0 && q8 % (((G >> 3) / (ntn / ksq)) * ntn) == 0) ? ksq : 1;
  auto tile_of = [&](int T, long long& tm0, long long& tn0) {
    const int xcd = T & 7;
    int loc = T >> 3;
    if (seq > 1) {
      const int per = G >> 3, sh = ntn / seq, l = loc % per, k = loc / per;
      loc = ((k / seq) * (per / sh) + l / sh) * ntn + (inter ? (k % seq) * sh + l % sh : (l % sh) * seq + k % seq);
    }
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    tm0 = (long long)(id / ntn) * E_BM;
    tn0 = (long long)(id % ntn) * E_BN;
  };
  int T = blockIdx.x;
  long long tm0, tn0, nm0, nn0, kbeg = 0;
  bool has_next = false;
  long long ws_slot = 0;   // split-K with a workspace: this work item's place among the partial tiles, (tile, slice) -> tile * slices + slice
  if (EPI == EP_SPLITK) {
    // work item = (tile, k-slice); the slices of one XCD's workgroups are the same few (operand panels fetched once per L2)
    const int xcd = T & 7, r = T >> 3;
    int id, z;
    if (ks < 0 && p.kchunk) esplitk_xcd_item(T, ntn * (int)(p.M / E_BM), -ks, id, z);  // any slice count, XCD by XCD
    else if (ks < 0) { id = T / (-ks); z = T % (-ks); }  // any slice count: plain order
    else if (ks >= 8) { const int per = ks >> 3; z = xcd * per + (r % per); id = r / per; }
    else { z = xcd % ks; id = r * (8 / ks) + xcd / ks; }
    tm0 = (long long)(id / ntn) * E_BM; tn0 = (long long)(id % ntn) * E_BN;
    nm0 = tm0; nn0 = tn0;
    // K-tiles are dealt evenly: the first (steps % slices) slices take one more
    const int steps = (int)(p.K / E_BK), nsl = ks < 0 ? -ks : ks, base = steps / nsl, rem = steps % nsl;
    nk = base + (z < rem ? 1 : 0);
    kbeg = (long long)(z * base + (z < rem ? z : rem)) * E_BK;
    ws_slot = (long long)id * nsl + z;
  } else {
    if (T >= nt) return;
    tile_of(T, tm0, tn0);
    has_next = T + G < nt;
    tile_of(has_next ? T + G : T, nm0, nn0);
  }

  // fragment read addresses (per lane, relative to a half tile's base)
  //  K-contiguous image [128 rows][128 B]: row = row0 + (lane & 15), 16-byte chunk (4 s + (lane >> 4)) ^ (row & 7)
  const unsigned rc0 = (unsigned)((lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4));  // s = 0; s = 1 is rc0 ^ 64
  //  K-major image [64 k-rows][256 B]: see frag_kmajor (gemm.hip); 32-byte block (col >> 4) ^ fk(krow)
  const int li = lane & 15, lq = lane >> 4;
  const int kf = (li >> 2) | ((lq & 1) << 2);
  const unsigned rk0 = (unsigned)((8 * lq + (li >> 2)) * 256 + 8 * (li & 3));
  // Transposed fragments are read by inline asm: behind an LDS-DMA hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every
  // ds_read_b64_tr_b16 it can see (three per K-tile in the K-major kernels: each phase's reads then waited for the half tile
  // issued a quarter of a K-tile earlier, i.e. for the whole memory latency, and the counted vmcnt(6) of phase 4 never mattered).
  // The ds_read_b128 of the K-contiguous images do not get that wait.  Every read here is retired by the phase's own
  // `s_waitcnt lgkmcnt(0)` in front of its MFMAs (E_LGKM0_F ties the fragment registers to that wait).
  auto rd_tr = [&](const unsigned char* a) -> bf8v {
    typedef int i2v_ __attribute__((ext_vector_type(2)));
    i2v_ lo, hi;
    const unsigned addr = (unsigned)(unsigned long long)LDS_PTR(const unsigned char, a);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr) : "memory");
    return __builtin_bit_cast(bf8v, __builtin_shufflevector(lo, hi, 0, 1, 2, 3));
  };
  auto rdA = [&](const unsigned char* base, int i, int s) -> bf8v {  // rows 64 wr + 16 i of an A half
    if (!TA) return *(const bf8v*)(base + (64 * wr + 16 * i) * 128 + (rc0 ^ (s << 6)));
    return rd_tr(base + rk0 + s * 32 * 256 + (((4 * wr + i) ^ kf) << 5));
  };
  auto rdB = [&](const unsigned char* base, int j, int s) -> bf8v {  // rows 32 wc + 16 j of a B half
    if (!TB) return *(const bf8v*)(base + (32 * wc + 16 * j) * 128 + (rc0 ^ (s << 6)));
    return rd_tr(base + rk0 + s * 32 * 256 + (((2 * wc + j) ^ kf) << 5));
  };

  f4v acc[2][2][4][2];  // [A half][B half][i][j]
  bf8v fa[4][2], fb0[2][2], fb1[2][2];

#define E_RD_A(H_)                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                          \
    fa[i_][0] = rdA(kt + (H_) * E_HALF, i_, 0);                               \
    fa[i_][1] = rdA(kt + (H_) * E_HALF, i_, 1);                               \
  }
#define E_RD_B(F_, H_)                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++) {                          \
    F_[j_][0] = rdB(kt + (2 + (H_)) * E_HALF, j_, 0);                         \
    F_[j_][1] = rdB(kt + (2 + (H_)) * E_HALF, j_, 1);                         \
  }
#define E_MFMA(HA_, HB_, F_)                                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                                          \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++)                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++)                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                                        \
    acc[HA_][HB_][i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F_[j_][s_], fa[i_][s_], acc[HA_][HB_][i_][j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define E_BAR()                                  \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);
#define E_LGKM0()                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);
// the same, with the fragments read by asm in this phase tied to the wait (an MFMA cannot move in front of it)
#define E_LGKM0_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), \
               "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]) :: "memory");                                              \
  __builtin_amdgcn_sched_barrier(0);
#define E_LGKM0_B(F_)                                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F_[0][0]), "+v"(F_[0][1]), "+v"(F_[1][0]), "+v"(F_[1][1]) :: "memory");     \
  __builtin_amdgcn_sched_barrier(0);

  // the LDS-DMA stream: half tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of K-tile u of the current tile (u >= nk: of the next
  // one; a workgroup's last tile "prefetches" its own first K-tiles again: the stream keeps its shape, no branches)
  const EStep<TA, 64> sa(p.lda);
  const EStep<TB, 32> sb(p.ldb);
  // (ks & 32, timing probe for stored products: every tile reads its A rows from the first 4096 rows - always in L2 -, results wrong by design)
  const long long amask = (EPI != EP_SPLITK && (ks & 32)) ? 4095 : -1LL;
  const unsigned char* cA = (const unsigned char*)A + (tm0 & amask) * sa.tile + (kbeg / E_BK) * sa.ktile;
  const unsigned char* cB = (const unsigned char*)B + tn0 * sb.tile + (kbeg / E_BK) * sb.ktile;
  const unsigned char* nA = (const unsigned char*)A + (nm0 & amask) * sa.tile;
  const unsigned char* nB = (const unsigned char*)B + nn0 * sb.tile;
  auto issue = [&](int u, int which, unsigned char* ktbase) {
    if (EPI == EP_SPLITK && u >= nk) return;  // one work item: nothing follows (the ring is the epilogue's staging area)
    const bool nx = u >= nk;
    const long long uu = nx ? u - nk : u;
    unsigned char* dst = ktbase + which * E_HALF + wave * 1024;
    if (which < 2) eglds2((nx ? nA : cA) + uu * sa.ktile + which * sa.half, sa.piece, offA, dst);
    else eglds2((nx ? nB : cB) + uu * sb.ktile + (which - 2) * sb.half, sb.piece, offB, dst);
    if (E_DUP && which == 2) eglds2((nx ? nB : cB) + uu * sb.ktile + (which - 2) * sb.half, sb.piece, offB, dst);   // timing probe: the same half tile again
  };
  const bool use_bias = (EPI <= EP_RELU_BITS) && p.bias;
  unsigned char* const biasl = smem + E_BIAS + wave * 1024;
  auto issue_bias = [&](long long bn) {  // 256 floats = 64 lanes x 16 B, this wave's private copy
    const float* src = use_bias ? p.bias + bn : (const float*)p.B;  // (no bias: any readable address; the copy is not used)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const unsigned char*)src + lane * 16),
                                     (__attribute__((address_space(3))) void*)(biasl), 16, 0, 0);
  };

  if (VAR & 2) {
    // de-phase the workgroups: all tiles take the same time, so without this every CU reaches its epilogue (128 KiB of
    // stores) at the same moment.  Workgroups that stream the same A panel (ntn consecutive ones of an XCD) stay together.
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, per = (G >> 3);
    const int ngr = (per + ntn - 1) / ntn, gr = loc / ntn;
    const int period = nk * 2900 + 6000;                       // cycles per tile, roughly
    const int delay = (int)((long long)period * (gr * 8 + xcd) / (ngr * 8));
    for (int i = 0; i < delay; i += 1024) __builtin_amdgcn_s_sleep(16);
  }
  if (EPI == EP_GATE_BITS) {
    // bit e of a mask byte keeps column e of the lane's 8: dword k holds columns 2k (low half) and 2k + 1
    if (tid < 256) {
      eu4v m;
#pragma unroll
      for (int k = 0; k < 4; k++) m[k] = (((tid >> (2 * k)) & 1) ? 0x0000ffffu : 0u) | (((tid >> (2 * k + 1)) & 1) ? 0xffff0000u : 0u);
      *(eu4v*)(smem + E_LUT + tid * 16) = m;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to every wave behind the prologue's barrier
  }
  // ---- prologue: K-tile 0 and all of K-tile 1 (a tile's A1(1) is always issued ahead of its first K-tile)
  issue(0, 2, smem); issue(0, 0, smem); issue(0, 3, smem); issue(0, 1, smem);
  issue(1, 2, smem + E_KTILE); issue(1, 0, smem + E_KTILE); issue(1, 3, smem + E_KTILE); issue(1, 1, smem + E_KTILE);
  if (E_DUP) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  E_BAR();
  if (wr == 1) { E_BAR(); }  // the stagger: waves 4-7 run one barrier behind

  int d = 0;  // ring slot of the current K-tile
  bool first = true;
  // VAR 8 (diagnostic build): s_memtime stamps of waves 0 and 4 per tile -> p.gate as u64 [G][2][64 tiles][4]
  unsigned long long* stamp = nullptr;
  int tix = 0;
  if (VAR & 8) stamp = (unsigned long long*)p.gate + ((size_t)blockIdx.x * 2 + wr) * 64 * 4;
#define E_STAMP(k_) if ((VAR & 8) && wc == 0 && lane == 0 && tix < 64) stamp[tix * 4 + (k_)] = (k_) == 3 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();

  // VAR 64 (diagnostic build): s_memtime at 17 points of ONE K-tile (tile PT_TILE, K-tile PT_K) of waves 0 and 4, through LDS
  // (a global store would enter the counted vmcnt stream) -> p.gate as u64 [G][2][32] at the end of the kernel
  bool pst_on = false;
#define E_PST(n_) if ((VAR & 64) && pst_on && wc == 0 && lane == 0) ((volatile unsigned long long*)(smem + E_RING + 8192))[wr * 32 + (n_)] = __builtin_amdgcn_s_memtime();
  // VAR 128 (diagnostic build): s_memtime at the start of every K-tile of the workgroup's THIRD tile, at its epilogue's start and end
  // and at the next tile's first two K-tiles (waves 0 and 4) -> p.gate as u64 [G][2][32]
#define E_KST(n_) if ((VAR & 128) && wc == 0 && lane == 0 && (n_) < 32) ((volatile unsigned long long*)(smem + E_RING + 8192))[wr * 32 + (n_)] = __builtin_amdgcn_s_memtime();

  // Epilogue addressing.  After the column swap lane (li, lq) holds the 8 columns 8 cq .. 8 cq + 7 of a 32-column block of row li:
  // the four 16-byte pieces of a row sit in lanes 16 apart and ADJACENT lanes belong to different rows.  The memory pipeline merges
  // the addresses of adjacent lanes only: stored (or loaded) like that, a 1 KiB wave-instruction is 64 separate 16-byte requests and
  // takes 64 cycles of the CU's store path instead of 16 (tools/probe_store3.hip: 128 KiB tile 7.5 k cycles against 2.1 k, at any
  // row pitch, with nothing else running).  So the packed values go through ONE lane transpose first (4 x ds_bpermute_b32 per unit):
  // lane L then holds piece L & 3 of row L >> 2 and four adjacent lanes cover 64 contiguous bytes; side inputs and the bit mask are
  // addressed in that layout as well.
  const int cq = ((lq & 1) << 1) | (lq >> 1);
  const int er = lane >> 2, ep = lane & 3;                                          // row and 16-byte piece of the transposed layout
  const int tsrc = 4 * (er + 16 * (((ep & 1) << 1) | (ep >> 1)));                  // ds_bpermute address: the lane that holds (row er, piece ep)
  const unsigned cvo = (unsigned)(((128 * wr + er) * p.ldc + 64 * wc + 8 * ep) * 2);   // C
  const unsigned gvo = (unsigned)(((128 * wr + er) * (EPI == EP_RESID ? p.ldr : p.ldg) + 64 * wc + 8 * ep) * 2);  // residual / row-dot matrix
  // bit mask (EP_RELU_BITS out, EP_GATE_BITS in): 8 bytes per row and wave; row pitch ldg, or - PERO_GEMM_MASK_TILED - 32 bytes inside the N-tile's own M x 32 plane
  const bool mtiled = (EPI == EP_RELU_BITS || EPI == EP_GATE_BITS) && (p.flags & PERO_GEMM_MASK_TILED);
  const long long mld = mtiled ? 32 : p.ldg;
  auto mask_base = [&](long long tm, long long tn) -> const unsigned char* {
    return (const unsigned char*)p.gate + (mtiled ? (tn >> 8) * p.M * 32 + tm * 32 : tm * p.ldg + (tn >> 3));
  };
  const unsigned mvo = (unsigned)((128 * wr + er) * mld + 8 * wc);
  auto lane_t = [&](const unsigned x) -> unsigned { return (unsigned)__builtin_amdgcn_ds_bpermute(tsrc, (int)x); };
  (void)cq; (void)tsrc;
  // The same transpose through LDS memory (E_XCHG_LDS): a unit is written as the lanes hold it (one ds_write_b128: row li, piece cq)
  // and read back in the transposed layout (one ds_read_b128: row er, piece ep) - 13 + 4 cycles of the LDS pipeline per unit against
  // 4 x 6 for the bpermutes (tools/probe_bperm.hip); the f32 sums of EP_RESID 2 x (13 + 4) against 8 x 6, and without their lane
  // swaps.  A wave's LDS instructions execute in order, so the read needs no wait of its own and a slot is reused without one.
  // Pieces are XOR-swizzled so that both directions are conflict-free (bf16: [16 rows][64 B], 2 slots; f32: [16][128 B]).
  unsigned char* const xstg = smem + E_XSTG + wave * 2048;
  const unsigned xw16 = (unsigned)(li * 64 + ((cq ^ ((li >> 1) & 3)) << 4));
  const unsigned xr16 = (unsigned)(er * 64 + ((ep ^ ((er >> 1) & 3)) << 4));
  const int xsw = (li ^ ((li >> 1) & 1)) & 7, xsr = (er ^ ((er >> 1) & 1)) & 7;
  const unsigned xw32 = (unsigned)(li * 128), xr32 = (unsigned)(er * 128);
  (void)xw16; (void)xr16; (void)xsw; (void)xsr; (void)xw32; (void)xr32;
  const unsigned char* const lutp = smem + E_LUT;
  // bf16 pairs (1, 0) and (0, 1) in registers the compiler cannot fold: as a constant operand hipcc (ROCm 7.2) prints the pair (1, 0) as the
  // inline constant `1.0`, which v_dot2c_f32_bf16 reads as the f32 pattern 0x3f800000 = the pair (0, 1) (measured: both sums took the odd column)
  unsigned sel_lo = 0x00003f80u, sel_hi = 0x3f800000u;
  asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
  float cs_run = 0.f;        // EP_GATE_BITS + column sums: the wave's running sums of its current N-tile (see the epilogue)
  long long cs_tn0 = -1;
  eu4v side0[8], side1[8];   // side inputs of rows 0-63 / 64-127 (EP_RESID, EP_ROWDOT: 16 B per unit)
  eu2v sm0[4], sm1[4];       // EP_GATE_BITS: the 8 mask bytes of a row (this wave's 64 columns), per row group

  for (;;) {
    E_STAMP(0);
    E_STAMP(3);
#pragma unroll
    for (int ha = 0; ha < 2; ha++)
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[ha][hb][i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    issue_bias(tn0);  // this tile's bias row for its epilogue: part of the stream (older than everything a later wait counts)

    // side-input descriptor of this tile
    const int spitch = (int)(EPI == EP_RESID ? p.ldr * 2 : EPI == EP_ROWDOT ? p.ldg * 2 : mld);  // bytes per row
    const ei4v srs = ersrc(EPI == EP_RESID ? (const void*)((const bf16raw*)p.resid + tm0 * p.ldr + tn0)
                           : EPI == EP_ROWDOT ? (const void*)((const bf16raw*)p.gate + tm0 * p.ldg + tn0)
                           : (const void*)mask_base(tm0, tn0),
                           (unsigned)(256 * spitch));

    // one K-tile; LAST = the tile's last one (peeled: the side loads of the epilogue start in its phase 4 and their registers
    // are live from there only)
    auto ktile = [&](auto last_c, const int t) __attribute__((always_inline)) {
      constexpr bool last = decltype(last_c)::value;
      unsigned char* const kt = smem + d * E_KTILE;         // K-tile t
      unsigned char* const kn = smem + (d ^ 1) * E_KTILE;   // K-tiles t + 1 (being completed) and, slot by slot, t + 2
      if (VAR & 64) pst_on = (tix == ((VAR & 8) ? 2 : 0)) && t == (nk > 4 ? 4 : 1);
      if ((VAR & 128) && (tix == 2 || EPI == EP_SPLITK)) { E_KST(t); }
      if ((VAR & 128) && tix == 3 && t < 3) { E_KST(nk + 2 + t); }
      // P1
      E_PST(0);
      E_RD_B(fb0, 0);
      __builtin_amdgcn_sched_barrier(0);
      E_RD_A(0);
      if (EPI == EP_GATE_BITS && last) {  // the tile's mask bytes (8 per row and wave): 8 small loads, four phases ahead of the epilogue
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int so0 = 16 * i * spitch, so1 = (64 + 16 * i) * spitch;
          if (E_ABL & 2) { sm0[i] = (eu2v){0xffffffffu, 0xffffffffu}; sm1[i] = sm0[i]; asm volatile("" : "+v"(sm0[i]), "+v"(sm1[i])); }
          else {
          E_BLOAD8(sm0[i], mvo, srs, so0, 0);
          E_BLOAD8(sm1[i], mvo, srs, so1, 0);
          }
        }
      }
      if (last || t > 0) issue(t + 1, 1, kn);                       // A1(t+1)  (a tile's A1(1) went out ahead of the previous epilogue)
      if (TA) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // the B0 reads (issued first) are done: B0 may be restaged in P2
      else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      E_PST(1);
      E_BAR();
      E_LGKM0_B(fb0);
      E_LGKM0_A();
      E_PST(2);
      E_MFMA(0, 0, fb0);
      E_PST(3);
      E_BAR();
      E_PST(4);
      // P2
      E_RD_B(fb1, 1);
      issue(t + 2, 2, kt);                                  // B0(t+2)
      E_PST(5);
      E_BAR();
      E_LGKM0_B(fb1);
      E_PST(6);
      E_MFMA(0, 1, fb1);
      E_PST(7);
      E_BAR();
      E_PST(8);
      // P3
      E_RD_A(1);
      issue(t + 2, 0, kt);                                  // A0(t+2)
      E_PST(9);
      E_BAR();
      E_LGKM0_A();
      E_PST(10);
      E_MFMA(1, 1, fb1);
      E_PST(11);
      E_BAR();
      E_PST(12);
      // P4
      if (CN::L0 && last) {  // side inputs of the wave's rows 0-63
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int so = 16 * i * spitch;
          E_BLOAD16(side0[2 * i], gvo, srs, so, 0);
          E_BLOAD16(side0[2 * i + 1], gvo, srs, so, 64);
        }
      }
      issue(t + 2, 3, kt);                                  // B1(t+2)
      // K-tile t+1 has landed (this wave's pieces).  What was issued after its last half tile A1(t+1) stays in flight:
      // normally the three half tiles of t+2; in a tile's first K-tile also the previous epilogue (side loads of rows 64-127,
      // stores) and the bias row; in its last K-tile the side loads issued just above.
      if (!last && t == 0 && first) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(7 + E_DX) : "memory");  // + the bias row
      // (the column-sum atomic of EP_GATE_BITS leaves only when the workgroup's N-tile changes: on that one tile the count below asks for
      //  one operation more than needed to have retired - never for one less)
      else if (!last && t == 0 && !first) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 + E_DX + 1 + CN::L1 + 2 * CN::S_HALF) : "memory");
      else if (CN::L0 && last) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 + E_DX + CN::L0) : "memory");
      else if (EPI == EP_SPLITK && t + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stream has ended
      else asm volatile("s_waitcnt vmcnt(%0)" :: "i"(6 + E_DX) : "memory");
      E_PST(13);
      if ((VAR & 128) && last && tix == 2) { E_KST(nk + 5); }
      E_BAR();
      if ((VAR & 128) && last && tix == 2) { E_KST(nk + 6); }
      E_PST(14);
      E_MFMA(1, 0, fb0);
      E_PST(15);
      if ((VAR & 128) && last && tix == 2) { E_KST(nk + 7); }
      E_BAR();
      E_PST(16);
      d ^= 1;
    };
    // (the first K-tile is its own copy of the code: its MFMAs take the zero accumulators as an inline constant, so the 128 registers
    //  are never cleared by moves, and its t == 0 cases fold)
    ktile(std::false_type{}, 0);
    for (int t = 1; t < nk - 1; t++) ktile(std::false_type{}, t);
    ktile(std::true_type{}, nk - 1);

    // ---- epilogue, straight from the accumulators
    E_STAMP(1);
    if ((VAR & 128) && tix == 2) { E_KST(nk); }
    first = false;
    issue(nk + 1, 1, smem + (d ^ 1) * E_KTILE);  // A1 of the next tile's K-tile 1: ahead of the stores in the in-order counter
    // Undo the stagger for the epilogue (VAR 16: keep it).  Staggered, waves 4-7 sit at the barrier behind their last MFMA
    // cluster until waves 0-3 reach the next tile's first barrier, i.e. through the whole epilogue of waves 0-3, and waves 0-3 then
    // sit through the epilogue of waves 4-7: the two epilogues ran one after the other (stamps at K = 512: 4.5 k + 5.3 k cycles of a
    // 40 k-cycle tile).  With one extra barrier here waves 0-3 wait the 256 cycles of that last cluster and both epilogues run
    // side by side; waves 4-7 take the extra barrier in front of the next tile, which staggers the groups again.
    if (EPI != EP_SPLITK && !(VAR & 16) && wr == 0) { E_BAR(); }
    if ((VAR & 128) && tix == 2) { E_KST(nk + 8); }
    if (EPI == EP_SPLITK) {
      // f32 tile added into C with atomics whose wave-instructions cover 256 contiguous bytes (full atomic rate): two rounds
      // through the (now free) 128 KiB ring, [128 rows][256 f32], 16-byte chunk index XORed with (row & 15)
      if (wr == 0) { E_BAR(); }  // undo the stagger: every wave has finished its last reads and MFMAs
      if ((VAR & 64) && wc == 0 && lane < 17)
        ((unsigned long long*)p.gate)[((size_t)blockIdx.x * 2 + wr) * 32 + lane] = ((unsigned long long*)(smem + E_RING + 8192))[wr * 32 + lane];
      if ((VAR & 128) && wc == 0 && lane < 32)
        ((unsigned long long*)p.gate)[((size_t)blockIdx.x * 2 + wr) * 32 + lane] = ((unsigned long long*)(smem + E_RING + 8192))[wr * 32 + lane];
      if (p.resid) {
        // partial tile -> workspace with plain 16-byte stores, accumulator by accumulator: every wave-instruction writes 1 KiB of
        // contiguous bytes ([accumulator][wave][lane] float4; pero_splitk_reduce_k knows the layout).  The slices of a tile are summed
        // by that kernel in slice order: deterministic, and 6 TB/s of stores + one pass over the partials instead of f32 atomics, which
        // the chip executes at 1.3 TB/s of added bytes (MI355X_MICROARCH.md): 64 MB per launch were ~49 us of every weight gradient.
        f4v* const W = (f4v*)p.resid + ws_slot * (E_BM * E_BN / 4) + wave * 64 + lane;
#pragma unroll
        for (int ha = 0; ha < 2; ha++)
#pragma unroll
          for (int hb = 0; hb < 2; hb++)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
              for (int j = 0; j < 2; j++) W[(((ha * 2 + hb) * 4 + i) * 2 + j) * 512] = acc[ha][hb][i][j];
        return;
      }
      float* const C = (float*)p.C;
#pragma unroll
      for (int ha = 0; ha < 2; ha++) {
        if (ha) { E_LGKM0(); E_BAR(); }
#pragma unroll
        for (int hb = 0; hb < 2; hb++)
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
              const int row = 64 * wr + 16 * i + li, chunk = 16 * wc + 8 * hb + 4 * j + lq;
              *(f4v*)(smem + row * 1024 + ((chunk ^ (row & 15)) << 4)) = acc[ha][hb][i][j];
            }
        E_LGKM0();
        E_BAR();
        // staged row r = 64 * wr' + rr  <->  tile row 128 * wr' + 64 * ha + rr
#pragma unroll 4
        for (int it = 0; it < 64; it++) {
          const int item = it * 8 + wave, row = item >> 2, seg = item & 3;
          const int col = 64 * seg + lane;
          const float v = *(const float*)(smem + row * 1024 + ((((col >> 2)) ^ (row & 15)) << 4) + (col & 3) * 4);
          const long long grow = tm0 + 128 * (row >> 6) + 64 * ha + (row & 63);
          atomicAdd(C + grow * p.ldc + tn0 + col, v * p.alpha);
        }
      }
      return;
    }
    {
      // (VAR 1, timing probe: every tile of a workgroup lands on the workgroup's FIRST tile - the stores are issued and acknowledged, the lines stay in L2)
      const ei4v crs = (VAR & 1) ? ersrc((bf16raw*)p.C + (long long)((blockIdx.x >> 3) * 256) * p.ldc + (blockIdx.x & 7) * 256, (unsigned)(256 * p.ldc * 2))
                                 : ersrc((bf16raw*)p.C + tm0 * p.ldc + tn0, (unsigned)(256 * p.ldc * 2));
      const int cpitch = (int)(p.ldc * 2);
      // bias of the lane's columns as the accumulators hold them: 64 wc + 32 hb + 16 j + 4 (lane >> 4) .. + 3
      f4v bx[2][2];
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          bx[hb][j] = (f4v){0.f, 0.f, 0.f, 0.f};
          if (use_bias) bx[hb][j] = *(const f4v*)(biasl + (64 * wc + 32 * hb + 16 * j + 4 * lq) * 4);
        }
      if ((VAR & 128) && tix == 2) { E_KST(nk + 9); }
      float cs[2][8];  // EP_GATE_BITS + column sums
      float rd[4];     // EP_ROWDOT: row dots of one row group
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int e = 0; e < 8; e++) cs[hb][e] = 0.f;
#pragma unroll
      for (int ha = 0; ha < 2; ha++) {
        if (CN::L1 && ha == 0) {  // side inputs of rows 64-127, then wait for those of rows 0-63
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int so = (64 + 16 * i) * spitch;
            E_BLOAD16(side1[2 * i], gvo, srs, so, 0);
            E_BLOAD16(side1[2 * i + 1], gvo, srs, so, 64);
          }
          E_WAIT8(2 + 2 + CN::L1, side0);
        }
        if ((VAR & 128) && tix == 2 && ha == 1) { E_KST(nk + 10); }
        if (CN::L1 && ha == 1) E_WAIT8(CN::S_HALF, side1);
        if (EPI == EP_GATE_BITS && ha == 0) {
          // retired by phase 4's wait of the last K-tile long ago (the statements tie the registers to a wait)
          E_WAIT4(63, sm0);
          E_WAIT4(63, sm1);
        }
        // Units (row group i, B half hb) are computed NI row groups at a time, instruction kind by instruction kind: one unit's chain
        // (add -> convert -> lane swap -> store, each waiting for the one before, and the next unit reusing the same registers)
        // took ~190 cycles of a wave that had the SIMD to itself (stamps); independent chains side by side fill those gaps.
        constexpr int NI = (EPI == EP_RESID || EPI == EP_ROWDOT) ? 1 : 2;
#pragma unroll
        for (int ib = 0; ib < 4; ib += NI) {
          eu4v o[NI][2];
          eu4v kp[NI][2];   // EP_GATE_BITS: the AND masks of the unit's 8 columns, requested with the staging reads (not in front of their use)
          (void)kp;
#pragma unroll
          for (int ii = 0; ii < NI; ii++)
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              const int i = ib + ii;
              if (EPI == EP_GATE_BITS) {
                const eu2v mm = ha ? sm1[i] : sm0[i];
                const unsigned byte = __builtin_amdgcn_ubfe(hb ? mm[1] : mm[0], 8u * (unsigned)ep, 8u);
                kp[ii][hb] = *(const eu4v*)(lutp + (byte << 4));
              }
              // (alpha = 1: the other kernels' acc * alpha + bias, bit for bit; the modes that never take a bias skip the add of zero)
              const f4v x = EPI <= EP_RELU_BITS ? acc[ha][hb][i][0] + bx[hb][0] : acc[ha][hb][i][0];
              const f4v y = EPI <= EP_RELU_BITS ? acc[ha][hb][i][1] + bx[hb][1] : acc[ha][hb][i][1];
              if (EPI == EP_RESID) {
                // f32 columns first (one rounding): even lane groups keep x and take the odd neighbour's x, odd ones y
                float v[8];
                if (E_XCHG_LDS) {
                  // x = columns 4 lq .. + 3 (piece lq), y = columns 16 + 4 lq .. (piece 4 + lq) of the row's 32 f32; read: columns 8 ep .. + 7
                  *(f4v*)(xstg + xw32 + ((lq ^ xsw) << 4)) = x;
                  *(f4v*)(xstg + xw32 + (((4 + lq) ^ xsw) << 4)) = y;
                  const f4v r0 = *(const f4v*)(xstg + xr32 + (((2 * ep) ^ xsr) << 4));
                  const f4v r1 = *(const f4v*)(xstg + xr32 + (((2 * ep + 1) ^ xsr) << 4));
#pragma unroll
                  for (int e = 0; e < 4; e++) { v[e] = r0[e]; v[4 + e] = r1[e]; }
                } else {
#pragma unroll
                  for (int e = 0; e < 4; e++) {
                    auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[e]), __float_as_uint(y[e]), false, false);
                    v[e] = __uint_as_float(sw[0]); v[4 + e] = __uint_as_float(sw[1]);
                  }
#pragma unroll
                  for (int e = 0; e < 8; e++) v[e] = __uint_as_float(lane_t(__float_as_uint(v[e])));   // the f32 sums move, rounded once below
                }
                const eu4v r4 = ha ? side1[2 * i + hb] : side0[2 * i + hb];
#pragma unroll
                for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(r4[e] << 16); v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u); }
                o[ii][hb][0] = pack2bf(v[0], v[1]); o[ii][hb][1] = pack2bf(v[2], v[3]); o[ii][hb][2] = pack2bf(v[4], v[5]); o[ii][hb][3] = pack2bf(v[6], v[7]);
              } else {
                unsigned px0 = pack2bf(x[0], x[1]), px1 = pack2bf(x[2], x[3]);
                unsigned py0 = pack2bf(y[0], y[1]), py1 = pack2bf(y[2], y[3]);
                if (EPI == EP_RELU || EPI == EP_RELU_BITS) {
                  // ReLU on the ROUNDED pairs: one v_pk_max_i16 per dword instead of two v_max_f32 (rounding keeps the sign, so
                  // max(round(x), 0) == round(max(x, 0)); a negative value or -0 is a negative int16 and becomes +0)
                  px0 = epk_relu(px0); px1 = epk_relu(px1); py0 = epk_relu(py0); py1 = epk_relu(py1);
                }
                auto s0 = __builtin_amdgcn_permlane16_swap(px0, py0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(px1, py1, false, false);
                if (E_XCHG_LDS) {
                  *(eu4v*)(xstg + hb * 1024 + xw16) = (eu4v){s0[0], s1[0], s0[1], s1[1]};
                  o[ii][hb] = *(const eu4v*)(xstg + hb * 1024 + xr16);
                } else {
                  o[ii][hb][0] = lane_t(s0[0]); o[ii][hb][1] = lane_t(s1[0]); o[ii][hb][2] = lane_t(s0[1]); o[ii][hb][3] = lane_t(s1[1]);
                }
              }
            }
#pragma unroll
          for (int ii = 0; ii < NI; ii++) {
            const int i = ib + ii;
            const int so = (64 * ha + 16 * i) * cpitch;
            unsigned mL = 0, mH = 0;
            if (EPI == EP_ROWDOT) rd[i] = 0.f;
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              eu4v& ou = o[ii][hb];
              if (EPI == EP_RELU_BITS) {
                // bit e = (stored column e > 0); after the ReLU every half word is +0, -0 or positive
                unsigned z = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  const unsigned m = epk_nonzero(ou[k]);  // bit 0: low half > 0, bit 16: high half (one v_pk_min_u16: the halves are >= +0)
                  z |= m << (2 * k);
                }
                const unsigned byte = (z & 0x55u) | ((z >> 15) & 0xaau);
                if (hb == 0) mL = byte << (8 * ep); else mH = byte << (8 * ep);
              }
              if (EPI == EP_GATE_BITS) {
                // column sums always (no branch per dword on the runtime flag; only the final atomic is conditional): one
                // v_dot2c_f32_bf16 per column pair and half, (x, y) . (1, 0) and (x, y) . (0, 1) - exact products, exact sums with zero
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  ou[k] &= kp[ii][hb][k];
                  cs[hb][2 * k] = edot2(ou[k], sel_lo, cs[hb][2 * k]);
                  cs[hb][2 * k + 1] = edot2(ou[k], sel_hi, cs[hb][2 * k + 1]);
                }
              }
              if (EPI == EP_ROWDOT) {
                const eu4v g4 = ha ? side1[2 * i + hb] : side0[2 * i + hb];
#pragma unroll
                for (int k = 0; k < 4; k++) rd[i] = edot2(ou[k], g4[k], rd[i]);   // one v_dot2c_f32_bf16 instead of 4 unpacks, 2 multiplies, 2 adds
              }
              if (VAR & 4) {  // ablation: no stores (the values stay live)
                asm volatile("" :: "v"(ou[0]), "v"(ou[1]), "v"(ou[2]), "v"(ou[3]));
                if (i + hb == 0) { if (hb) E_BSTORE16(ou, cvo, crs, so, 64); else E_BSTORE16(ou, cvo, crs, so, 0); }
              } else if (VAR & 32) {
                if (hb) E_BSTORE16_NT(ou, cvo, crs, so, 64); else E_BSTORE16_NT(ou, cvo, crs, so, 0);
              } else {
                if (hb) E_BSTORE16(ou, cvo, crs, so, 64); else E_BSTORE16(ou, cvo, crs, so, 0);
              }
            }
            if (EPI == EP_RELU_BITS) {
              // OR over the four lanes of a row (one quad): every one of them then holds the row's 8 bytes
              mL |= (unsigned)__builtin_amdgcn_mov_dpp((int)mL, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
              mH |= (unsigned)__builtin_amdgcn_mov_dpp((int)mH, 0xB1, 0xf, 0xf, false);
              mL |= (unsigned)__builtin_amdgcn_mov_dpp((int)mL, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
              mH |= (unsigned)__builtin_amdgcn_mov_dpp((int)mH, 0x4E, 0xf, 0xf, false);
              const eu2v mo = {mL, mH};
              const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)mask_base(tm0, tn0), 0, (int)(256 * mld), 0x00020000);
              // all four store them (same address, same data): one instruction, no branch (storing from the quad's first lane only
              // measured no faster)
              if (E_ABL & 1) asm volatile("" :: "v"(mo[0]), "v"(mo[1]));
              else __builtin_amdgcn_raw_buffer_store_b64(mo, mrs, mvo, (64 * ha + 16 * i) * (int)mld, 0);
            }
            if (EPI == EP_ROWDOT) {
              // sum over the four lanes of a row (one quad), then its first lane adds into [m][n / 128] (two waves per 128-column block)
              float s = rd[i];
              s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xf, 0xf, false));
              s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xf, 0xf, false));
              float* dst = (float*)p.bias + (tm0 + 128 * wr + 64 * ha + 16 * i + er) * (p.N >> 7) + ((tn0 + 64 * wc) >> 7);
              if (ep == 0) atomicAdd(dst, s);
            }
          }
        }
      }
      if (colsum) {
        // column sums of the tile's 128 rows of this wave.  A lane holds 16 partial sums (its 8 columns x 2 B halves) of its
        // rows; a halving butterfly over the 16 row indices of the wave (lane bits 2-5: xor 32, 16, 8, 4; each step a lane keeps half of
        // its values and adds the partner's copy of them - 15 exchanges instead of 64) leaves the total of value `er` on the lane,
        // and ONE atomic instruction with all 64 lanes adds the wave's 64 column sums
        // The exchanges stay in the vector ALU (round 3; as 15 ds_bpermute_b32 the four dependent LDS round trips at the very end of the
        // epilogue were ~1 us per tile = 6 % of the gated product): lanes 32 / 16 apart trade by v_permlane32_swap / v_permlane16_swap - the
        // swap of (a, b) followed by a + b IS "keep one, add the partner's copy of it" for both partners at once - lanes 8 / 4 apart by DPP
        // row rotations / shifts; a + partner(a) is the same on both partners, each then keeps the sum that is its to keep.
        float w8[8], w4[4], w2[2];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs[0][j]), __float_as_uint(cs[1][j]), false, false);
          w8[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // lanes 0-31 (er & 8 == 0): cs[0][j] of both; lanes 32-63: cs[1][j] of both
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(w8[j]), __float_as_uint(w8[j + 4]), false, false);
          w4[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // er & 4 == 0: w8[j] of both; else w8[j + 4] of both
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const float s0 = w4[j] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w4[j]), 0x128, 0xf, 0xf, false));          // row_ror:8 = lane ^ 8
          const float s1 = w4[j + 2] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w4[j + 2]), 0x128, 0xf, 0xf, false));
          w2[j] = (er & 2) ? s1 : s0;
        }
        // lane ^ 4 inside a row of 16: lanes with bit 2 clear take lane + 4 (row_shl:4 into banks 0, 2), the others lane - 4 (row_shr:4 into banks 1, 3)
        auto x4 = [&](float v) -> float {
          int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
          r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x114, 0xf, 0xa, false);
          return __int_as_float(r);
        };
        const float t0 = w2[0] + x4(w2[0]), t1 = w2[1] + x4(w2[1]);
        const float tot = (er & 1) ? t1 : t0;     // value index er = 8 * hb + e, of the lane's piece ep
        // A persistent workgroup keeps ONE N-tile while ntn divides its stride (ntn = 8 at N = 2048: always): the wave's 64 column sums run
        // on in ONE register over its tiles and leave by one atomic when the N-tile changes or the kernel ends (per tile: 131 072
        // wave-atomics per 2048-line launch onto 2048 addresses, each of them in the in-order vmcnt queue for ~3 k cycles)
        if (tn0 != cs_tn0) {
          if (cs_tn0 >= 0) atomicAdd((float*)p.bias + cs_tn0 + 64 * wc + 32 * (er >> 3) + 8 * ep + (er & 7), cs_run);
          cs_run = 0.f; cs_tn0 = tn0;
        }
        cs_run += tot;
      }
    }
    E_STAMP(2);
    if ((VAR & 128) && tix == 2) { E_KST(nk + 1); }
    tix++;
    if (!has_next) break;
    T += G;
    tm0 = nm0; tn0 = nn0; cA = nA; cB = nB;
    has_next = T + G < nt;
    tile_of(has_next ? T + G : T, nm0, nn0);
    nA = (const unsigned char*)A + (nm0 & amask) * sa.tile;
    nB = (const unsigned char*)B + nn0 * sb.tile;
    if (!(VAR & 16) && wr == 1) { E_BAR(); }  // the stagger again
  }
  if ((VAR & 64) && wc == 0 && lane < 17)
    ((unsigned long long*)p.gate)[(VAR & 8 ? (size_t)gridDim.x * 2 * 64 * 4 : 0) + ((size_t)blockIdx.x * 2 + wr) * 32 + lane] = ((unsigned long long*)(smem + E_RING + 8192))[wr * 32 + lane];
  if ((VAR & 128) && wc == 0 && lane < 32)
    ((unsigned long long*)p.gate)[((size_t)blockIdx.x * 2 + wr) * 32 + lane] = ((unsigned long long*)(smem + E_RING + 8192))[wr * 32 + lane];
  if (EPI == EP_GATE_BITS && colsum && cs_tn0 >= 0) atomicAdd((float*)p.bias + cs_tn0 + 64 * wc + 32 * (lane >> 5) + 8 * (lane & 3) + ((lane >> 2) & 7), cs_run);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last tile's surplus prefetches land before the LDS is released
  if ((VAR & 16) && wr == 0) { E_BAR(); }  // balance the stagger barrier (otherwise the last epilogue's extra barrier has done it)
#undef E_RD_A
#undef E_RD_B
#undef E_MFMA
#undef E_BAR
#undef E_LGKM0
#undef E_STAMP
}

// =====================================================================================================================================
// TWO INDEPENDENT WORKGROUPS PER CU (round 4, opt-in: pero_set_option("gemm_d128", 1)) for the stored K-contiguous products with FEW K-tiles per
// output tile (K = 512: eight).  In gemm_bf16_e256 the eight waves of the one workgroup per CU share one barrier domain: all of them are in
// the tile's epilogue together and the matrix pipe idles for a quarter of the tile (5-6 k of 35 k cycles, plus the main
// loop slowed by the whole chip storing at once).  Here a workgroup is FOUR waves (2 (M) x 2 (N), one per SIMD) on a 256 x 128 tile - the
// same 128 x 64 outputs, fragment reads, MFMAs and epilogue per wave as in the 256 x 256 tile - with half the LDS (80 KiB), so that two
// workgroups share a CU: two waves per SIMD as before, but from different barrier domains, and one workgroup's epilogue runs under the
// other's main loop.  Per K-tile A0 A1 (128 rows x 64 k, 16 KiB) and B0 B1 (64 columns x 64 k, 8 KiB); the A halves roll through a ring of
// THREE 16 KiB slots, the B halves through three 8 KiB slots (half a = 2 t + h of the stream sits in slot a % 3):
//     P1: wait A0(t) B0(t) | barrier | read B0 A0 -> A0 x B0                   P3: barrier | read A1 ; issue B0(t+2) -> B1(t)'s slot -> A1 x B1
//     P2: wait A1(t) B1(t) | barrier | read B1 ; issue A1(t+1) B1(t+1) -> the   P4: barrier | issue A0(t+2) -> A1(t)'s slot        -> A1 x B0
//         slots of A0(t) B0(t)                                    -> A0 x B1
// One barrier per phase: it stands behind every wave's lgkmcnt(0) of the phase before (the slot restaged behind it has been read by all)
// and behind every wave's counted vmcnt (the half tiles read behind it have landed for all).  The LDS-DMA stream is one sequence over
// (tile, K-tile) pairs as in gemm_bf16_e256, 12 instructions per wave and K-tile, never drained; the epilogue's stores (and the next
// tile's bias row) enter the same in-order counter and the first three waits of a tile count them out.
#define D_BM 256
#define D_BN 128
#define D_AH 16384                     // A half tile
#define D_BH 8192                      // B half tile
#define D_BRING (3 * D_AH)
#define D_AUX (D_BRING + 3 * D_BH)     // 4 KiB: two copies (tile parity) of the tile's 128 bias floats (1 KiB each: lanes 32-63 repeat lanes 0-31) | EP_GATE_BITS: the mask LUT
#define D_XSTG (D_AUX + 4096)          // 4 x 1 KiB: each wave's staging image of the epilogue's lane transpose (the two units of a row group one after the other)
#define D_LDS_BYTES (D_XSTG + 4096)    // 81 920 = half of the CU's LDS

template <int EPI> struct DCnt {
  static constexpr int ML = EPI == EP_GATE_BITS ? 8 : 0;                                   // mask loads (phase 1 of the last K-tile)
  static constexpr int L0 = (EPI == EP_RESID || EPI == EP_ROWDOT) ? 8 : 0;                 // side loads of rows 0-63 (phase 4 of the last K-tile, ahead of A0(t+2)) ...
  static constexpr int L1 = L0;                                                            // ... and of rows 64-127 (start of the epilogue)
  static constexpr int ST = 16 + (EPI == EP_RELU_BITS ? 8 : 0) + (EPI == EP_ROWDOT ? 8 : 0);   // stores / atomics of a tile
};

template <int EPI, int VAR>
__global__ __launch_bounds__(256, 2) void gemm_bf16_d128(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef DCnt<EPI> CN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = (int)(p.N / D_BN);
  const int nt = (int)(p.M / D_BM) * ntn;
  const int G = gridDim.x;  // multiple of 8
  const int q8 = nt >> 3, r8 = nt & 7;
  const int nk = (int)(p.K / E_BK);  // >= 3 (launcher)
  const long long lda2 = p.lda * 2, ldb2 = p.ldb * 2;
  // LDS-DMA: a wave-instruction fills 8 rows x 128 B (lane-linear; the chunk swizzle is on the SOURCE address); the 256 threads cover 32 rows,
  // an A half is four such pieces (rows 0, 32 | 128, 160 of the tile + 64 h), a B half two (columns 0 | 64 of the tile + 32 h)
  const unsigned offA = (unsigned)(((tid >> 3) * p.lda + (((tid & 7) ^ ((tid >> 3) & 7)) << 3)) * 2);
  const unsigned offB = (unsigned)(((tid >> 3) * p.ldb + (((tid & 7) ^ ((tid >> 3) & 7)) << 3)) * 2);
  const bool colsum = EPI == EP_GATE_BITS && (p.flags & PERO_GEMM_COLSUM);

  auto tile_of = [&](int T, long long& tm0, long long& tn0) {
    const int xcd = T & 7, loc = T >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    tm0 = (long long)(id / ntn) * D_BM;
    tn0 = (long long)(id % ntn) * D_BN;
  };
  int T = blockIdx.x;
  if (T >= nt) return;
  long long tm0, tn0, nm0, nn0;
  tile_of(T, tm0, tn0);
  bool has_next = T + G < nt;
  tile_of(has_next ? T + G : T, nm0, nn0);

  const unsigned rc0 = (unsigned)((lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4));  // fragment read offset inside a half tile (see gemm_bf16_e256)
  const int li = lane & 15, lq = lane >> 4;
  f4v acc[2][2][4][2];  // [A half][B half][i][j]
  bf8v fa[4][2], fb0[2][2], fb1[2][2];
#define D_RD_A(BASE_)                                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                                             \
    fa[i_][0] = *(const bf8v*)((BASE_) + (64 * wr + 16 * i_) * 128 + rc0);                       \
    fa[i_][1] = *(const bf8v*)((BASE_) + (64 * wr + 16 * i_) * 128 + (rc0 ^ 64u));               \
  }
#define D_RD_B(F_, BASE_)                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++) {                                             \
    F_[j_][0] = *(const bf8v*)((BASE_) + (32 * wc + 16 * j_) * 128 + rc0);                       \
    F_[j_][1] = *(const bf8v*)((BASE_) + (32 * wc + 16 * j_) * 128 + (rc0 ^ 64u));               \
  }
#define D_MFMA(HA_, HB_, F_)                                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                                          \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++)                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++)                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                                        \
    acc[HA_][HB_][i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F_[j_][s_], fa[i_][s_], acc[HA_][HB_][i_][j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define D_BAR()                                  \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);
#define D_LGKM0()                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);
#define D_VM(n_) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(n_) : "memory")

  const unsigned char* cA = (const unsigned char*)p.A + tm0 * lda2;
  const unsigned char* cB = (const unsigned char*)p.B + tn0 * ldb2;
  const unsigned char* nA = (const unsigned char*)p.A + nm0 * lda2;
  const unsigned char* nB = (const unsigned char*)p.B + nn0 * ldb2;
  auto glds = [&](const unsigned char* src, unsigned off, unsigned char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
  };
  // half h of K-tile u of the current tile (u >= nk: of the next one; a workgroup's last tile "prefetches" its own first K-tiles again) -> ring slot
  auto issueA = [&](int u, int h, int slot) {
    const bool nx = u >= nk;
    const unsigned char* src = (nx ? nA : cA) + (long long)(nx ? u - nk : u) * (E_BK * 2) + h * (64 * lda2);
    unsigned char* dst = smem + slot * D_AH + wave * 1024;
    glds(src, offA, dst); glds(src + 32 * lda2, offA, dst + 4096); glds(src + 128 * lda2, offA, dst + 8192); glds(src + 160 * lda2, offA, dst + 12288);
  };
  auto issueB = [&](int u, int h, int slot) {
    const bool nx = u >= nk;
    const unsigned char* src = (nx ? nB : cB) + (long long)(nx ? u - nk : u) * (E_BK * 2) + h * (32 * ldb2);
    unsigned char* dst = smem + D_BRING + slot * D_BH + wave * 1024;
    glds(src, offB, dst); glds(src + 64 * ldb2, offB, dst + 4096);
  };
  const bool use_bias = (EPI <= EP_RELU_BITS) && p.bias;
  int tix = 0;
  auto issue_bias = [&](long long bn) {  // 128 floats = 32 lanes x 16 B (lanes 32-63 fetch them again), into this tile's parity copy
    const float* src = use_bias ? p.bias + bn : (const float*)p.B;  // (no bias: any readable address; the copy is not used)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const unsigned char*)src + (lane & 31) * 16),
                                     (__attribute__((address_space(3))) void*)(smem + D_AUX + (EPI == EP_GATE_BITS ? 0 : (tix & 1) * 1024)), 16, 0, 0);
  };
  if (EPI == EP_GATE_BITS) {   // bit e of a mask byte keeps column e of the lane's 8: dword k holds columns 2k (low half) and 2k + 1
    eu4v m;
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = (((tid >> (2 * k)) & 1) ? 0x0000ffffu : 0u) | (((tid >> (2 * k + 1)) & 1) ? 0xffff0000u : 0u);
    *(eu4v*)(smem + D_AUX + tid * 16) = m;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to every wave behind the first barrier
  }
  // ---- prologue, in the stream's steady-state order: B0(0) A0(0) | A1(0) B1(0) | B0(1) | A0(1)
  issueB(0, 0, 0); issueA(0, 0, 0); issueA(0, 1, 1); issueB(0, 1, 1); issueB(1, 0, 2); issueA(1, 0, 2);
  int s = 0;           // ring slot of the current K-tile's A0 / B0
  bool first = true;

  // epilogue addressing: see gemm_bf16_e256 (the wave's outputs are rows 128 wr .., columns 64 wc .. of the tile)
  const int cq = ((lq & 1) << 1) | (lq >> 1);
  const int er = lane >> 2, ep = lane & 3;
  const unsigned cvo = (unsigned)(((128 * wr + er) * p.ldc + 64 * wc + 8 * ep) * 2);
  const unsigned gvo = (unsigned)(((128 * wr + er) * (EPI == EP_RESID ? p.ldr : p.ldg) + 64 * wc + 8 * ep) * 2);
  const unsigned mvo = (unsigned)((128 * wr + er) * p.ldg + 8 * wc);
  unsigned char* const xstg = smem + D_XSTG + wave * 1024;
  const unsigned xw16 = (unsigned)(li * 64 + ((cq ^ ((li >> 1) & 3)) << 4));
  const unsigned xr16 = (unsigned)(er * 64 + ((ep ^ ((er >> 1) & 3)) << 4));
  const unsigned char* const lutp = smem + D_AUX;
  unsigned sel_lo = 0x00003f80u, sel_hi = 0x3f800000u;
  asm volatile("" : "+s"(sel_lo), "+s"(sel_hi));
  float cs_run = 0.f;
  long long cs_tn0 = -1;
  eu4v side0[8], side1[8];
  eu2v sm0[4], sm1[4];
  (void)gvo; (void)side0; (void)side1;

  for (;;) {
    constexpr int EB = EPI == EP_GATE_BITS ? 0 : 1;   // the tile's bias row is part of the stream (the gate takes none: its LUT lives there)
    if (EB) issue_bias(tn0);
    const int spitch = (int)(EPI == EP_RESID ? p.ldr * 2 : EPI == EP_ROWDOT ? p.ldg * 2 : p.ldg);  // bytes per row
    const ei4v srs = ersrc(EPI == EP_RESID ? (const void*)((const bf16raw*)p.resid + tm0 * p.ldr + tn0)
                           : EPI == EP_ROWDOT ? (const void*)((const bf16raw*)p.gate + tm0 * p.ldg + tn0)
                           : (const void*)((const unsigned char*)p.gate + tm0 * p.ldg + (tn0 >> 3)),
                           (unsigned)(256 * spitch));
    // one K-tile; KIND 0 / 1: the tile's first two (their waits count the previous epilogue out), 3: its last (peeled: mask / side loads), 2: the others
    auto ktile = [&](auto kind_c, const int t) __attribute__((always_inline)) {
      constexpr int kind = decltype(kind_c)::value;
      constexpr bool last = kind == 3;
      const int s1 = s == 2 ? 0 : s + 1, s2 = s == 0 ? 2 : s - 1;
      const unsigned char* const A0p = smem + s * D_AH;
      const unsigned char* const A1p = smem + s1 * D_AH;
      const unsigned char* const B0p = smem + D_BRING + s * D_BH;
      const unsigned char* const B1p = smem + D_BRING + s1 * D_BH;
      // P1: A0(t), B0(t) have landed - in flight behind them: 12 of the stream (+ the previous epilogue and this tile's bias row for K-tiles 0, 1)
      if (kind == 0) { if (first) D_VM(12 + EB); else D_VM(12 + CN::ML + CN::L0 + CN::L1 + CN::ST + EB); }
      else if (kind == 1) { if (first) D_VM(12 + EB); else D_VM(12 + CN::L1 + CN::ST + EB); }
      else D_VM(12);
      D_BAR();
      D_RD_B(fb0, B0p);
      __builtin_amdgcn_sched_barrier(0);
      D_RD_A(A0p);
      if (EPI == EP_GATE_BITS && last) {  // the tile's mask bytes (8 per row and wave)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int so0 = 16 * i * spitch, so1 = (64 + 16 * i) * spitch;
          E_BLOAD8(sm0[i], mvo, srs, so0, 0);
          E_BLOAD8(sm1[i], mvo, srs, so1, 0);
        }
      }
      D_LGKM0();
      D_MFMA(0, 0, fb0);
      // P2: A1(t), B1(t) have landed - behind them B0(t+1), A0(t+1) (+ the mask loads)
      if (kind == 0) { if (first) D_VM(6 + EB); else D_VM(6 + CN::L0 + CN::L1 + CN::ST + EB); }
      else D_VM(6 + (last ? CN::ML : 0));
      D_BAR();
      D_RD_B(fb1, B1p);
      issueA(t + 1, 1, s); issueB(t + 1, 1, s);
      D_LGKM0();
      D_MFMA(0, 1, fb1);
      // P3
      D_BAR();
      D_RD_A(A1p);
      issueB(t + 2, 0, s1);
      D_LGKM0();
      D_MFMA(1, 1, fb1);
      // P4
      D_BAR();
      if (CN::L0 && last) {  // side inputs of the wave's rows 0-63
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int so = 16 * i * spitch;
          E_BLOAD16(side0[2 * i], gvo, srs, so, 0);
          E_BLOAD16(side0[2 * i + 1], gvo, srs, so, 64);
        }
      }
      issueA(t + 2, 0, s1);
      D_MFMA(1, 0, fb0);
      s = s2;
    };
#pragma unroll
    for (int ha = 0; ha < 2; ha++)
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[ha][hb][i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    ktile(std::integral_constant<int, 0>{}, 0);
    ktile(std::integral_constant<int, 1>{}, 1);
    for (int t = 2; t < nk - 1; t++) ktile(std::integral_constant<int, 2>{}, t);
    ktile(std::integral_constant<int, 3>{}, nk - 1);
    first = false;

    // ---- epilogue, straight from the accumulators: gemm_bf16_e256's, unit by unit (no barrier inside: the other workgroup of the CU has the matrix pipe meanwhile)
    {
      static_assert(EPI == EP_PLAIN || EPI == EP_RELU || EPI == EP_RELU_BITS || EPI == EP_GATE_BITS, "epilogue mode");
      const ei4v crs = ersrc((bf16raw*)p.C + tm0 * p.ldc + tn0, (unsigned)(256 * p.ldc * 2));
      const int cpitch = (int)(p.ldc * 2);
      const unsigned char* const biasl = smem + D_AUX + (tix & 1) * 1024;
      f4v bx[2][2];
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          bx[hb][j] = (f4v){0.f, 0.f, 0.f, 0.f};
          if (use_bias) bx[hb][j] = *(const f4v*)(biasl + (64 * wc + 32 * hb + 16 * j + 4 * lq) * 4);
        }
      float cs[2][8];
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int e = 0; e < 8; e++) cs[hb][e] = 0.f;
#pragma unroll
      for (int ha = 0; ha < 2; ha++) {
        if (EPI == EP_GATE_BITS && ha == 0) {   // behind the mask loads: phases 2-4 of the last K-tile (6 + 2 + 4)
          E_WAIT4(12, sm0);
          E_WAIT4(12, sm1);
        }
#pragma unroll
        for (int ib = 0; ib < 4; ib += 2) {
          eu4v o[2][2];
          eu4v kp[2][2];
          (void)kp;
#pragma unroll
          for (int ii = 0; ii < 2; ii++)
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              const int i = ib + ii;
              if (EPI == EP_GATE_BITS) {
                const eu2v mm = ha ? sm1[i] : sm0[i];
                const unsigned byte = __builtin_amdgcn_ubfe(hb ? mm[1] : mm[0], 8u * (unsigned)ep, 8u);
                kp[ii][hb] = *(const eu4v*)(lutp + (byte << 4));
              }
              const f4v x = EPI <= EP_RELU_BITS ? acc[ha][hb][i][0] + bx[hb][0] : acc[ha][hb][i][0];
              const f4v y = EPI <= EP_RELU_BITS ? acc[ha][hb][i][1] + bx[hb][1] : acc[ha][hb][i][1];
              unsigned px0 = pack2bf(x[0], x[1]), px1 = pack2bf(x[2], x[3]);
              unsigned py0 = pack2bf(y[0], y[1]), py1 = pack2bf(y[2], y[3]);
              if (EPI == EP_RELU || EPI == EP_RELU_BITS) { px0 = epk_relu(px0); px1 = epk_relu(px1); py0 = epk_relu(py0); py1 = epk_relu(py1); }
              auto s0 = __builtin_amdgcn_permlane16_swap(px0, py0, false, false);
              auto s1 = __builtin_amdgcn_permlane16_swap(px1, py1, false, false);
              // lane transpose through the wave's staging image (one image for every unit: a wave's LDS instructions execute in order)
              *(eu4v*)(xstg + xw16) = (eu4v){s0[0], s1[0], s0[1], s1[1]};
              o[ii][hb] = *(const eu4v*)(xstg + xr16);
            }
#pragma unroll
          for (int ii = 0; ii < 2; ii++) {
            const int i = ib + ii;
            const int so = (64 * ha + 16 * i) * cpitch;
            unsigned mL = 0, mH = 0;
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              eu4v& ou = o[ii][hb];
              if (EPI == EP_RELU_BITS) {
                unsigned z = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  const unsigned m = epk_nonzero(ou[k]);
                  z |= m << (2 * k);
                }
                const unsigned byte = (z & 0x55u) | ((z >> 15) & 0xaau);
                if (hb == 0) mL = byte << (8 * ep); else mH = byte << (8 * ep);
              }
              if (EPI == EP_GATE_BITS) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  ou[k] &= kp[ii][hb][k];
                  cs[hb][2 * k] = edot2(ou[k], sel_lo, cs[hb][2 * k]);
                  cs[hb][2 * k + 1] = edot2(ou[k], sel_hi, cs[hb][2 * k + 1]);
                }
              }
              if (VAR & 4) {  // ablation: no stores (the values stay live)
                asm volatile("" :: "v"(ou[0]), "v"(ou[1]), "v"(ou[2]), "v"(ou[3]));
                if (i + hb == 0) { if (hb) E_BSTORE16(ou, cvo, crs, so, 64); else E_BSTORE16(ou, cvo, crs, so, 0); }
              } else {
                if (hb) E_BSTORE16(ou, cvo, crs, so, 64); else E_BSTORE16(ou, cvo, crs, so, 0);
              }
            }
            if (EPI == EP_RELU_BITS) {
              mL |= (unsigned)__builtin_amdgcn_mov_dpp((int)mL, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
              mH |= (unsigned)__builtin_amdgcn_mov_dpp((int)mH, 0xB1, 0xf, 0xf, false);
              mL |= (unsigned)__builtin_amdgcn_mov_dpp((int)mL, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
              mH |= (unsigned)__builtin_amdgcn_mov_dpp((int)mH, 0x4E, 0xf, 0xf, false);
              const eu2v mo = {mL, mH};
              const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((unsigned char*)p.gate + tm0 * p.ldg + (tn0 >> 3), 0, (int)(256 * p.ldg), 0x00020000);
              __builtin_amdgcn_raw_buffer_store_b64(mo, mrs, mvo, (64 * ha + 16 * i) * (int)p.ldg, 0);
            }
          }
        }
      }
      if (colsum) {   // column sums of the wave's 128 rows: see gemm_bf16_e256
        float w8[8], w4[4], w2[2];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs[0][j]), __float_as_uint(cs[1][j]), false, false);
          w8[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(w8[j]), __float_as_uint(w8[j + 4]), false, false);
          w4[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const float s0 = w4[j] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w4[j]), 0x128, 0xf, 0xf, false));
          const float s1 = w4[j + 2] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w4[j + 2]), 0x128, 0xf, 0xf, false));
          w2[j] = (er & 2) ? s1 : s0;
        }
        auto x4 = [&](float v) -> float {
          int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);
          r = __builtin_amdgcn_update_dpp(r, __float_as_int(v), 0x114, 0xf, 0xa, false);
          return __int_as_float(r);
        };
        const float t0 = w2[0] + x4(w2[0]), t1 = w2[1] + x4(w2[1]);
        const float tot = (er & 1) ? t1 : t0;
        if (tn0 != cs_tn0) {
          if (cs_tn0 >= 0) atomicAdd((float*)p.bias + cs_tn0 + 64 * wc + 32 * (er >> 3) + 8 * ep + (er & 7), cs_run);
          cs_run = 0.f; cs_tn0 = tn0;
        }
        cs_run += tot;
      }
    }
    tix++;
    if (!has_next) break;
    T += G;
    tm0 = nm0; tn0 = nn0; cA = nA; cB = nB;
    has_next = T + G < nt;
    tile_of(has_next ? T + G : T, nm0, nn0);
    nA = (const unsigned char*)p.A + nm0 * lda2;
    nB = (const unsigned char*)p.B + nn0 * ldb2;
  }
  if (EPI == EP_GATE_BITS && colsum && cs_tn0 >= 0) atomicAdd((float*)p.bias + cs_tn0 + 64 * wc + 32 * (lane >> 5) + 8 * (lane & 3) + ((lane >> 2) & 7), cs_run);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last tile's surplus prefetches land before the LDS is released
#undef D_RD_A
#undef D_RD_B
#undef D_MFMA
#undef D_BAR
#undef D_LGKM0
#undef D_VM
}

int g_gemm_d128 = 0;   // pero_set_option("gemm_d128", K / 64): stored products up to that many K-tiles take this kernel
bool pero_launch_gemm_d128(const GemmP& p0, long long batch, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (!g_gemm_d128 || batch != 1 || ta || tb || out_f32 || p0.M % D_BM || p0.N % D_BN || p0.K % E_BK || p0.K < 3 * E_BK) return false;
  if (p0.alpha != 1.0f || (p0.flags & (PERO_GEMM_ATOMIC | PERO_GEMM_ACCUM | PERO_GEMM_ROWDOT | PERO_GEMM_MASK_TILED)) || p0.resid) return false;
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22) || p0.ldc >= (1LL << 22)) return false;
  const bool relu = p0.flags & PERO_GEMM_RELU, bits = p0.flags & PERO_GEMM_RELU_BITS, cs = p0.flags & PERO_GEMM_COLSUM;
  int epi;
  if (bits) {
    if (!p0.gate || (relu && cs) || p0.ldg >= (1LL << 22)) return false;
    if (!relu && p0.bias && !cs) return false;   // (as gemm_bf16_e256: the gate epilogue has no input-bias path)
    epi = relu ? EP_RELU_BITS : EP_GATE_BITS;
  } else if (cs || p0.gate) return false;
  else epi = relu ? EP_RELU : EP_PLAIN;
  int num_cus = (pero_num_cus() / 8) * 8;
  if (num_cus < 8) num_cus = 8;
  GemmP p = p0;
  p.kchunk = p.K;
  const long long nt = (p.M / D_BM) * (p.N / D_BN);
  if (nt < 2 * num_cus) return false;   // fewer tiles than workgroup places: the other kernels
  const unsigned G = (unsigned)(2 * num_cus);
  dim3 grid(G), block(256);
#define LAUNCH_D(EP_)                                                                              \
  do {                                                                                             \
    PERO_LDS_ATTR((gemm_bf16_d128<EP_, 0>), D_LDS_BYTES);                                          \
    hipLaunchKernelGGL((gemm_bf16_d128<EP_, 0>), grid, block, D_LDS_BYTES, st, p);                 \
  } while (0)
  switch (epi) {
    case EP_RELU: LAUNCH_D(EP_RELU); break;
    case EP_RELU_BITS: LAUNCH_D(EP_RELU_BITS); break;
    case EP_GATE_BITS: LAUNCH_D(EP_GATE_BITS); break;
    default: LAUNCH_D(EP_PLAIN); break;
  }
#undef LAUNCH_D
  return true;
}

// =====================================================================================================================================
// ROW-COMPLETE tile for the N = 512 products (DESIGN.md "what comes next"; opt-in: pero_set_option("gemm_nw", 1)): one workgroup owns 128 rows x
// ALL 512 columns, so that an epilogue can see whole rows (the residual LayerNorm of the out-projection / linear2 products, the LayerNorm
// backward behind linear1's / in_proj's input gradients).  The eight-phase compute code of gemm_bf16_e256 with another operand assignment:
//  * the two wave GROUPS take the two N-tiles of the same 128 rows: wave (g, wc) = all 128 rows (the two 64-row parts of ONE A half tile, which
//    both groups read: part 0 in phase 1, part 1 in phase 3) x the 32 wc columns of each half of N-tile g.  The same 128 x 64 outputs, the same
//    fragment reads and MFMAs per wave and K-tile as in the 256 x 256 tile.
//  * a K-tile is FIVE half tiles (A, B00, B10, B01, B11; Bgh = half h of N-tile g): A - the HBM stream - has its own ring of three slots, the four
//    B half tiles - the 512-row weight matrix, which every workgroup streams from L2 - roll through six: nine slots = 144 KiB + 16 KiB of epilogue
//    staging = all of the CU's LDS; the bias comes by side loads.  A slot is restaged two phases after its last read (B00 / B10: one phase,
//    their reads are retired by the lgkmcnt(8) in front of phase 1's first barrier):
//        phase 1 of K-tile t: A(t+2) -> A(t-1)'s slot          phase 2: B01(t+1), B11(t+1) -> the slots of B00(t), B10(t)
//        phase 4:             B00(t+2), B10(t+2) -> the slots of B01(t), B11(t)
//    (slots: A(t) = t % 3; with gb = 4 t % 6: B00 gb, B10 gb + 1, B01 gb + 2, B11 gb + 3, all mod 6 - every index has period three K-tiles).
//  * counted waits, two per K-tile: W2 in phase 1 (B01 / B11 of THIS K-tile, read from phase 2 on: newer are B00 / B10 of t + 1 and A(t+2) = 6
//    instructions) and W1 in phase 4 (A, B00, B10 of t + 1: newer are A(t+2), B01 / B11(t+1), B00 / B10(t+2) = 10); in a tile's first K-tile the
//    previous epilogue's side loads and stores, in its last the side loads of its own epilogue are counted out (constants at the waits).
// Epilogue: gemm_bf16_e256's plain / residual epilogue with `128 wr` gone from the row offsets and `256 wr` added to the columns.
#ifndef N_DBG
#define N_DBG 0   // bring-up switches of the LayerNorm epilogue: 1 no gamma / beta loads (gamma = 1, beta = 0), 2 no mean / rstd stores, 4 pass 2 recomputes nothing (stores y again)
#endif
#ifndef LNB_ABL
#define LNB_ABL 0   // timing-only ablation builds of the LayerNorm-backward epilogue (results wrong by design; tools/lnb_ab.py): 1 no column-sum
#endif             // butterflies, 2 no pass 1b arithmetic, 4 no pass 2 arithmetic (the packed dt rows are stored), 8 pass 2 without its second load of the t rows
#define N_BM 128
#define N_ASLOTS 3
#define N_BSLOTS 6
#define N_RING ((N_ASLOTS + N_BSLOTS) * E_HALF)     // 147 456
#define N_XSTG N_RING                                // 8 x 2 KiB: each wave's staging image of the epilogue's lane transpose
#define N_LDS_BYTES (N_XSTG + 8 * 2048)              // 163 840 = the CU's 160 KiB

template <int EPI, bool BIAS>
__global__ __launch_bounds__(512, 2) void gemm_bf16_n512(GemmP p, LnP q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(EPI == EP_PLAIN || EPI == EP_RESID || EPI == EP_RESID_LN || EPI == EP_RESID_LN_T || EPI == EP_RESID_LNB,
                "the row-complete tile has the plain, the residual, the residual + LayerNorm and the residual + LayerNorm-backward epilogue");
  static_assert(EPI != EP_RESID_LNB || !BIAS, "an input-gradient product has no bias");
  constexpr bool LN = EPI == EP_RESID_LN || EPI == EP_RESID_LN_T;
  constexpr bool LNB = EPI == EP_RESID_LNB;
  constexpr bool STORE_Y = EPI != EP_RESID_LN_T;
  constexpr bool RES = EPI == EP_RESID || LN || LNB;
  constexpr int LB = BIAS ? 4 : 0;                        // bias side loads (16 B per lane each)
  constexpr int L0 = LB + (RES ? 8 : 0);                  // side loads issued in phase 4 of the last K-tile: bias + residual rows 0-63
  constexpr int L1 = RES ? 8 : 0;                         // residual rows 64-127, issued halfway through rows 0-63
  constexpr int SH = STORE_Y ? 8 : 0;                     // stores per half of the epilogue
  constexpr int LNX = LN ? 8 + 2 + 16 : 0;                // LayerNorm: gamma / beta loads, mean / rstd stores, the 16 stores of t
  // vector-memory operations of an epilogue behind its phase-4 side loads.  LayerNorm backward: residual rows 64-127 (8), t rows 0-63 (8), gamma, beta,
  // rstd, t rows 64-127 (4 + 4 + 8 + 8), the rows of t again for pass 2 (16), the 16 stores of dx
  constexpr int EPO = LNB ? (8 + 8 + 8 + 8 + 16 + 16) : (L1 + 2 * SH + LNX);
  constexpr int cap63 = 63;                               // s_waitcnt vmcnt takes six bits: a larger count only asks for more than needed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;                // wr: N-tile (columns 256 wr ...), wc: its 64-column strip
  const int nt = (int)(p.M / N_BM);
  const int G = gridDim.x;                                // multiple of 8
  const int q8 = nt >> 3, r8 = nt & 7;
  const int nk = (int)(p.K / E_BK);                       // >= 3 (launcher)
  const unsigned offA = elane_off<false, true>(p.lda, tid), offB = elane_off<false, false>(p.ldb, tid);
  auto tile_of = [&](int T) -> long long {
    const int xcd = T & 7, loc = T >> 3;
    const int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    return (long long)id * N_BM;
  };
  int T = blockIdx.x;
  if (T >= nt) return;
  long long tm0 = tile_of(T);
  bool has_next = T + G < nt;
  long long nm0 = tile_of(has_next ? T + G : T);

  const unsigned rc0 = (unsigned)((lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4));
  const int li = lane & 15, lq = lane >> 4;
  auto rdA = [&](const unsigned char* base, int ha, int i, int s) -> bf8v {   // rows 64 ha + 16 i of the A half tile
    return *(const bf8v*)(base + (64 * ha + 16 * i) * 128 + (rc0 ^ (s << 6)));
  };
  auto rdB = [&](const unsigned char* base, int j, int s) -> bf8v {           // rows 32 wc + 16 j of a B half tile
    return *(const bf8v*)(base + (32 * wc + 16 * j) * 128 + (rc0 ^ (s << 6)));
  };
  f4v acc[2][2][4][2];  // [A part][B half][i][j]
  bf8v fa[4][2], fb0[2][2], fb1[2][2];
#define N_RD_A(HA_)                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++) {                          \
    fa[i_][0] = rdA(kA, (HA_), i_, 0);                                        \
    fa[i_][1] = rdA(kA, (HA_), i_, 1);                                        \
  }
#define N_RD_B(F_, BASE_)                                                     \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++) {                          \
    F_[j_][0] = rdB(BASE_, j_, 0);                                            \
    F_[j_][1] = rdB(BASE_, j_, 1);                                            \
  }
#define N_MFMA(HA_, HB_, F_)                                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                                          \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; s_++)                                                                        \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; i_++)                                                                        \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; j_++)                                                                        \
    acc[HA_][HB_][i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F_[j_][s_], fa[i_][s_], acc[HA_][HB_][i_][j_], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define N_BAR()                                  \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);
#define N_LGKM0()                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);
#define N_VMCNT(n_) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(n_) : "memory")

  // the two LDS-DMA streams (u >= nk: the K-tile u - nk of the workgroup's next tile; its last tile prefetches its own again)
  const unsigned char* cA = (const unsigned char*)p.A + tm0 * p.lda * 2;
  const unsigned char* nA = (const unsigned char*)p.A + nm0 * p.lda * 2;
  const unsigned char* const Bp = (const unsigned char*)p.B;
  const long long pieceA = 64 * p.lda * 2, pieceB = 128 * p.ldb * 2;
  unsigned char* const aring = smem;
  unsigned char* const bring = smem + N_ASLOTS * E_HALF;
  // LDS-DMA with a UNIFORM 64-bit base and a 32-bit lane offset (the builtin form keeps a 64-bit address pair per stream in VGPRs and adds
  // into it with the vector ALU; the compiler does not count these either)
  auto dma2 = [&](const unsigned char* sbase, long long piece, unsigned voff, unsigned char* dst) {
    const unsigned d0 = (unsigned)(unsigned long long)LDS_PTR(unsigned char, dst);
    const unsigned char* s1 = sbase + piece;
    // (s_nop 3: with the s_mov five wait states between a v_readlane_b32 that restores a spilled base and the load that reads it - the
    //  compiler pads that hazard for its own instructions only, tools/check_async_loads.py finds the unpadded ones)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(sbase), "v"(voff), "s"(d0) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(s1), "v"(voff), "s"(d0 + 8192u) : "memory", "m0");
  };
  auto issueA = [&](int u, int slot) {
    const bool nx = u >= nk;
    const long long uu = nx ? u - nk : u;
    dma2((nx ? nA : cA) + uu * (E_BK * 2), pieceA, offA, aring + slot * E_HALF + wave * 1024);
  };
  auto issueB = [&](int u, int g, int h, int slot) {
    const long long uu = u >= nk ? u - nk : u;
    dma2(Bp + uu * (E_BK * 2) + (long long)(256 * g + 32 * h) * p.ldb * 2, pieceB, offB, bring + slot * E_HALF + wave * 1024);
  };
  auto mod6 = [](int x) -> int { return x >= 6 ? x - 6 : x; };
  auto mod3 = [](int x) -> int { return x >= 3 ? x - 3 : x; };

  // ---- prologue: K-tile 0 and A, B00, B10 of K-tile 1 (stream order as in the steady state)
  issueA(0, 0); issueB(0, 0, 0, 0); issueB(0, 1, 0, 1); issueB(0, 0, 1, 2); issueB(0, 1, 1, 3);
  issueA(1, 1); issueB(1, 0, 0, 4); issueB(1, 1, 0, 5);
  N_VMCNT(6);
  N_BAR();
  if (wr == 1) { N_BAR(); }  // the stagger: waves 4-7 run one barrier behind

  int ga = 0, gb = 0;        // ring positions of the current K-tile
  bool first = true;

  unsigned char* const xstg = smem + N_XSTG + wave * 2048;
  const ei4v brs = ersrc(BIAS ? (const void*)(p.bias + 256 * wr) : (const void*)p.B, 256 * 4);
  eu4v side0[8], side1[8];   // residual rows 0-63 / 64-127 (EP_RESID)
  eu4v tside0[8], tside1[8]; // EP_RESID_LNB: the rows of t (the LayerNorm's output), same layout
  eu4v biasr[4];             // the lane's 16 bias values as the accumulators hold them: [hb * 2 + j]
  float run_g = 0.f, run_b = 0.f, run_x = 0.f;   // EP_RESID_LNB: the wave's column sums (dgamma, dbeta, sum of dx), one column per lane, over the workgroup's tiles
  (void)tside0; (void)tside1; (void)run_g; (void)run_b; (void)run_x;

  for (;;) {
#pragma unroll
    for (int ha = 0; ha < 2; ha++)
#pragma unroll
      for (int hb = 0; hb < 2; hb++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[ha][hb][i][j] = (f4v){0.f, 0.f, 0.f, 0.f};
    const int spitch = (int)(p.ldr * 2);
    const ei4v srs = ersrc(RES ? (const void*)((const bf16raw*)p.resid + tm0 * p.ldr) : (const void*)p.B, (unsigned)(128 * (RES ? spitch : 2)));
    const int tpitch_l = (int)(q.ldt * 2);
    const ei4v trs_l = ersrc(LNB ? (const void*)((const bf16raw*)q.t + tm0 * q.ldt) : (const void*)p.B, (unsigned)(128 * (LNB ? tpitch_l : 2)));
    (void)tpitch_l;

    auto ktile = [&](auto last_c, auto t0_c, const int t) __attribute__((always_inline)) {
      constexpr bool last = decltype(last_c)::value, t0 = decltype(t0_c)::value;
      unsigned char* const kA = aring + ga * E_HALF;
      unsigned char* const kB0 = bring + (gb + wr) * E_HALF;             // B(wr, 0): gb + wr <= 5
      unsigned char* const kB1 = bring + mod6(gb + 2 + wr) * E_HALF;     // B(wr, 1)
      // P1: A part 0 x B half 0
      N_RD_B(fb0, kB0);
      __builtin_amdgcn_sched_barrier(0);
      N_RD_A(0);
      issueA(t + 2, mod3(ga + 2));
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");   // the B half 0 reads (issued first) are done: their slots are restaged in P2
      // W2: B01 / B11 of this K-tile have landed (this wave's pieces).  Newer: B00 / B10 of t + 1, A(t+2); in a tile's first K-tile also the
      // previous tile's side loads and stores
      if (t0 && !first) N_VMCNT((6 + L0 + EPO) < cap63 ? (6 + L0 + EPO) : cap63);
      else N_VMCNT(6);
      N_BAR();
      N_LGKM0();
      N_MFMA(0, 0, fb0);
      N_BAR();
      // P2: A part 0 x B half 1
      N_RD_B(fb1, kB1);
      issueB(t + 1, 0, 1, gb);
      issueB(t + 1, 1, 1, gb + 1);
      N_BAR();
      N_LGKM0();
      N_MFMA(0, 1, fb1);
      N_BAR();
      // P3: A part 1 x B half 1
      N_RD_A(1);
      N_BAR();
      N_LGKM0();
      N_MFMA(1, 1, fb1);
      N_BAR();
      // P4: A part 1 x B half 0
      int lane_p = lane;   // (opaque copy: the side-load offsets are computed here, not carried through the main loop)
      if (last) asm volatile("" : "+v"(lane_p));
      const unsigned bvo = (unsigned)((64 * wc + 4 * (lane_p >> 4)) * 4);        // bias of the lane's accumulator columns: + (32 hb + 16 j) * 4
      const unsigned gvo = (unsigned)(((lane_p >> 2) * p.ldr + 256 * wr + 64 * wc + 8 * (lane_p & 3)) * 2);
      (void)bvo; (void)gvo;
      if (last && BIAS) {   // the epilogue's bias values and residual rows 0-63
        E_BLOAD16(biasr[0], bvo, brs, 0, 0);      // [hb * 2 + j]: columns + 32 hb + 16 j
        E_BLOAD16(biasr[1], bvo, brs, 0, 64);
        E_BLOAD16(biasr[2], bvo, brs, 0, 128);
        E_BLOAD16(biasr[3], bvo, brs, 0, 192);
      }
      if (last && RES) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int so = 16 * i * spitch;
          E_BLOAD16(side0[2 * i], gvo, srs, so, 0);
          E_BLOAD16(side0[2 * i + 1], gvo, srs, so, 64);
        }
      }
      issueB(t + 2, 0, 0, mod6(gb + 2));
      issueB(t + 2, 1, 0, mod6(gb + 3));
      // W1: A, B00, B10 of K-tile t + 1 have landed.  Newer: A(t+2), B01 / B11(t+1), B00 / B10(t+2) = 10; in a tile's first K-tile also the
      // previous epilogue's second-half side loads and its stores, in its last the side loads just issued
      if (t0 && !first) N_VMCNT((10 + EPO) < cap63 ? (10 + EPO) : cap63);
      else if (last) N_VMCNT(10 + L0);
      else N_VMCNT(10);
      N_BAR();
      N_MFMA(1, 0, fb0);
      N_BAR();
      ga = mod3(ga + 1);
      gb = mod6(gb + 4);
    };
    ktile(std::false_type{}, std::true_type{}, 0);
    for (int t = 1; t < nk - 1; t++) ktile(std::false_type{}, std::false_type{}, t);
    ktile(std::true_type{}, std::false_type{}, nk - 1);

    // ---- epilogue, straight from the accumulators
    first = false;
    if (wr == 0) { N_BAR(); }   // undo the stagger: both groups run their epilogues side by side
    if constexpr (LNB) {
      // ---- Linear input gradient + residual gradient = dt (rows complete in this workgroup) -> LayerNorm backward of the upstream norm, from its
      // output t and rstd (layernorm_bwd_pf_k<true>'s arithmetic on the ROUNDED dt, as the unfused pair has it): xhat = (t - beta) / gamma,
      // g = dt gamma, c1 = mean(g), c2 = mean(g xhat), dx = rstd (g - c1 - xhat c2); dgamma += dt xhat, dbeta += dt, dxsum += dx per column.
      //   pass 1a  accumulators + residual rows -> dt through the lane transpose, rounded and PACKED (the accumulators die here)
      //   pass 1b  the two row sums of the lane's 16 columns, all eight rows;  partials of the eight waves -> the A slot this tile's last K-tile has
      //            left free (the next tile's third K-tile refills it), ONE barrier
      //   pass 2   four chunks (32-column block hb, row half ha) of four rows: per row the totals from the exchange area, dx stored; per block the
      //            column sums, reduced over the wave's 16 row indices at once.
      // Vector-memory order (per lane): [phase 4 of the last K-tile: residual rows 0-63 x8] | t rows 0-63 x8 | residual rows 64-127 x8 (halfway through pass 1a of rows
      // 0-63) | gamma x4, beta x4 (in the last quarter of pass 1a) | t rows 64-127 x8 | rstd of chunks 0, 1, 2 (4 each) | 4 stores | rstd of chunk 3 | 4 + 4 + 4 stores.
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int li_e = lane_e & 15, lq_e = lane_e >> 4;
      const int er = lane_e >> 2, ep = lane_e & 3;
      const unsigned cvo = (unsigned)((er * p.ldc + 256 * wr + 64 * wc + 8 * ep) * 2);
      const unsigned gvo = (unsigned)((er * p.ldr + 256 * wr + 64 * wc + 8 * ep) * 2);
      const unsigned tvo = (unsigned)((er * q.ldt + 256 * wr + 64 * wc + 8 * ep) * 2);
      const int xsw = (li_e ^ ((li_e >> 1) & 1)) & 7, xsr = (er ^ ((er >> 1) & 1)) & 7;
      const unsigned xw32 = (unsigned)(li_e * 128), xr32 = (unsigned)(er * 128);
      const ei4v crs = ersrc((bf16raw*)p.C + tm0 * p.ldc, (unsigned)(128 * p.ldc * 2));
      const int cpitch = (int)(p.ldc * 2);
      float* const X = (float*)(aring + mod3(ga + 2) * E_HALF);   // exchange area: [8 waves][2 sums][128 rows] floats, 260 per wave (4 of padding: the 32 addresses of a half wave's ds_read_b32 fall on 32 banks)
      auto quad = [&](float v) -> float {
        v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, false));
        v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, false));
        return v;
      };
      auto lo2 = [](unsigned w) -> ef2v { return (ef2v){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; };
      eu4v dtp[2][4][2];
      // gamma / beta of the lane's 16 columns ([2 hb + half]), rstd of its eight rows
      const ei4v grs = ersrc(q.gamma + 256 * wr + 64 * wc, 64 * 4), ers = ersrc(q.beta + 256 * wr + 64 * wc, 64 * 4);
      const ei4v rrs = ersrc(q.rstd + tm0, 128 * 4);
      const unsigned gvoff = (unsigned)(8 * ep * 4), rvo = (unsigned)(er * 4);
      eu4v cg[4], cb[4];
      auto pass1a = [&](auto ha_c) __attribute__((always_inline)) {
        constexpr int ha = decltype(ha_c)::value;
        auto& side1_ = side1;   // (hipcc does not capture a variable that a generic lambda names only as an asm operand)
        auto& cg_ = cg; auto& cb_ = cb;
        const unsigned gvo_ = gvo, gvoff_ = gvoff;
        const ei4v srs_ = srs, grs_ = grs, ers_ = ers;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (ha == 0 && i == 2) {
            // residual rows 64-127 go out HALFWAY through rows 0-63: half of that half's accumulators and residual registers are free by now
            // (requested at the epilogue's start they sat beside all 128 accumulators: 16 spilled registers)
#pragma unroll
            for (int i2 = 0; i2 < 4; i2++) {
              const int so = (64 + 16 * i2) * spitch;
              E_BLOAD16(side1_[2 * i2], gvo_, srs_, so, 0);
              E_BLOAD16(side1_[2 * i2 + 1], gvo_, srs_, so, 64);
            }
          }
          if (ha == 1 && i == 3) {
            // the constants go out when three quarters of the accumulators are packed (40 registers; pass 1b needs them first thing)
            E_BLOAD16(cg_[0], gvoff_, grs_, 0, 0); E_BLOAD16(cg_[1], gvoff_, grs_, 0, 16); E_BLOAD16(cg_[2], gvoff_, grs_, 0, 128); E_BLOAD16(cg_[3], gvoff_, grs_, 0, 144);
            E_BLOAD16(cb_[0], gvoff_, ers_, 0, 0); E_BLOAD16(cb_[1], gvoff_, ers_, 0, 16); E_BLOAD16(cb_[2], gvoff_, ers_, 0, 128); E_BLOAD16(cb_[3], gvoff_, ers_, 0, 144);
          }
#pragma unroll
          for (int hb = 0; hb < 2; hb++) {
            const f4v x = acc[ha][hb][i][0], y = acc[ha][hb][i][1];
            *(f4v*)(xstg + xw32 + ((lq_e ^ xsw) << 4)) = x;
            *(f4v*)(xstg + xw32 + (((4 + lq_e) ^ xsw) << 4)) = y;
            const f4v r0 = *(const f4v*)(xstg + xr32 + (((2 * ep) ^ xsr) << 4));
            const f4v r1 = *(const f4v*)(xstg + xr32 + (((2 * ep + 1) ^ xsr) << 4));
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = r0[e]; v[4 + e] = r1[e]; }
            const eu4v r4 = ha ? side1[2 * i + hb] : side0[2 * i + hb];
#pragma unroll
            for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(r4[e] << 16); v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u); }
            dtp[ha][i][hb] = (eu4v){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
            // (the packed rows are made opaque HERE: left alone the compiler sinks the adds and conversions down to the rows' first use in pass 1b and
            //  parks the f32 rows of all eight units - 64 registers - in scratch meanwhile)
            asm volatile("" : "+v"(dtp[ha][i][hb]));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // t rows 0-63 first (they are needed after BOTH halves of pass 1a: issued behind rows 0-63 they had one half to arrive and the wave waited
      // for HBM; with every arithmetic instruction of this epilogue compiled out it still cost 128 us per launch over the plain residual
      // epilogue - tools/lnb_ab.py, LNB_ABL = 7: its loads' latency, not its 1 900 vector-ALU instructions, is what the epilogue costs)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int so = 16 * i * tpitch_l;
        E_BLOAD16(tside0[2 * i], tvo, trs_l, so, 0);
        E_BLOAD16(tside0[2 * i + 1], tvo, trs_l, so, 64);
      }
      E_WAIT8(12, side0);   // residual rows 0-63 (phase 4 of the last K-tile): newer are B00 / B10 (4) and the 8 loads above
      pass1a(std::integral_constant<int, 0>{});
      E_WAIT8(0, side1);    // residual rows 64-127 (issued halfway through pass 1a of rows 0-63): nothing newer
      pass1a(std::integral_constant<int, 1>{});
#pragma unroll
      for (int i = 0; i < 4; i++) {   // t rows 64-127 (beside rows 64-127 of the accumulators they spill 13 registers: they arrive under pass 1b of rows 0-63)
        const int so = (64 + 16 * i) * tpitch_l;
        E_BLOAD16(tside1[2 * i], tvo, trs_l, so, 0);
        E_BLOAD16(tside1[2 * i + 1], tvo, trs_l, so, 64);
      }
      // t rows 0-63 and the constants (issued in pass 1a's last quarter): newer are the 8 loads just issued
      E_WAIT8(8, tside0);
      E_WAIT4(8, cg);
      E_WAIT4(8, cb);
      // pair e of block hb <-> columns 32 hb + 8 ep + 2 e, + 1
      auto Gp = [&](int hb, int e) -> ef2v { return (ef2v){__uint_as_float(cg[2 * hb + (e >> 1)][2 * (e & 1)]), __uint_as_float(cg[2 * hb + (e >> 1)][2 * (e & 1) + 1])}; };
      auto Bp = [&](int hb, int e) -> ef2v { return (ef2v){__uint_as_float(cb[2 * hb + (e >> 1)][2 * (e & 1)]), __uint_as_float(cb[2 * hb + (e >> 1)][2 * (e & 1) + 1])}; };
      // ---- pass 1b: s1 = sum dt gamma, s2 = sum dt (t - beta) (= dt gamma xhat) over the lane's 16 columns of each of its eight rows
#pragma unroll
      for (int ha = 0; ha < 2; ha++) {
        if (ha == 1) E_WAIT8(0, tside1);   // (nothing newer; the older LDS-DMA of the next tile's first K-tiles has had the whole epilogue so far)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ef2v a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
          for (int hb = 0; hb < 2; hb++) {
            const eu4v tw = ha ? tside1[2 * i + hb] : tside0[2 * i + hb];
#pragma unroll
            for (int e = 0; e < 4; e++) {
              if (LNB_ABL & 2) { if (e == 0) { a1[0] += __uint_as_float(dtp[ha][i][hb][0]); a2[0] += __uint_as_float(tw[0]); } continue; }
              const ef2v d = lo2(dtp[ha][i][hb][e]);
              const ef2v u = lo2(tw[e]) - Bp(hb, e);
              a1 = __builtin_elementwise_fma(d, Gp(hb, e), a1);
              a2 = __builtin_elementwise_fma(d, u, a2);
            }
          }
          const float q1 = quad(a1[0] + a1[1]), q2 = quad(a2[0] + a2[1]);
          if (ep == 0) {
            X[wave * 260 + 64 * ha + 16 * i + er] = q1;
            X[wave * 260 + 128 + 64 * ha + 16 * i + er] = q2;
          }
        }
      }
      // ---- pass 2.  The rows of t again, chunk by chunk (c = 2 hb + ha: rows 64 ha + 16 i + er, the 16-byte piece of block hb), one chunk ahead
      unsigned rsc[3][4];   // rstd of a chunk's four rows, loaded two chunks ahead (held from pass 1a on they were 8 registers too many)
      auto tload = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value, ha = c & 1;
        auto& rsc_ = rsc;   // (hipcc does not capture a variable that a generic lambda names only as an asm operand)
        const unsigned rvo_ = rvo;
        const ei4v rrs_ = rrs;
        E_BLOAD4(rsc_[c % 3][0], rvo_, rrs_, 256 * ha, 0); E_BLOAD4(rsc_[c % 3][1], rvo_, rrs_, 256 * ha, 64);
        E_BLOAD4(rsc_[c % 3][2], rvo_, rrs_, 256 * ha, 128); E_BLOAD4(rsc_[c % 3][3], rvo_, rrs_, 256 * ha, 192);
      };
      tload(std::integral_constant<int, 0>{});
      tload(std::integral_constant<int, 1>{});
      tload(std::integral_constant<int, 2>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      N_BAR();
      auto colred8 = [&](const ef2v (&A)[4]) -> float {
        // 8 values per lane (columns 8 ep + j) summed over the wave's 16 row indices (lane bits 2-5): a halving butterfly over lanes 32, 16 and 8
        // apart (each step a lane keeps half of its values and adds the partner's copy of them), then lane ^ 4; lane (er, ep) ends with the
        // total of column 8 ep + (er >> 1)
        float w4[4], w2[2];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(A[j >> 1][j & 1]), __float_as_uint(A[2 + (j >> 1)][j & 1]), false, false);
          w4[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // er & 8 == 0: values j of both; else values j + 4 of both
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(w4[j]), __float_as_uint(w4[j + 2]), false, false);
          w2[j] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);      // er & 4 == 0: w4[j] of both; else w4[j + 2] of both
        }
        const float s0 = w2[0] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w2[0]), 0x128, 0xf, 0xf, false));   // row_ror:8 = lane ^ 8
        const float s1 = w2[1] + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w2[1]), 0x128, 0xf, 0xf, false));
        const float w1 = (er & 2) ? s1 : s0;
        int r = __builtin_amdgcn_update_dpp(0, __float_as_int(w1), 0x104, 0xf, 0x5, false);    // lane ^ 4 (see gemm_bf16_e256's column sums)
        r = __builtin_amdgcn_update_dpp(r, __float_as_int(w1), 0x114, 0xf, 0xa, false);
        return w1 + __int_as_float(r);
      };
      float tot_g[2], tot_b[2], tot_x[2];
      ef2v G[4], Bt[4], IG[4], AG[4], AB[4], AX[4];
      auto chunk = [&](auto c_c) __attribute__((always_inline)) {
        constexpr int c = decltype(c_c)::value, hb = c >> 1, ha = c & 1;
        const unsigned cvo_ = cvo;
        const ei4v crs_ = crs;
        if constexpr (ha == 0) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            G[e] = Gp(hb, e); Bt[e] = Bp(hb, e);
            asm volatile("" : "+v"(G[e]));   // (opaque here: the compiler otherwise takes the reciprocals of BOTH blocks in front of pass 1b and spills them)
            // (v_rcp_f32, one ulp: the correctly rounded quotient is a dozen instructions per column and tile)
            IG[e][0] = G[e][0] != 0.f ? __builtin_amdgcn_rcpf(G[e][0]) : 0.f;
            IG[e][1] = G[e][1] != 0.f ? __builtin_amdgcn_rcpf(G[e][1]) : 0.f;
            AG[e] = (ef2v){0.f, 0.f}; AB[e] = (ef2v){0.f, 0.f}; AX[e] = (ef2v){0.f, 0.f};
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int r = 64 * ha + 16 * i + er;
          const float c1 = quad(X[(2 * ep) * 260 + r] + X[(2 * ep + 1) * 260 + r]) * (1.0f / 512.f);
          const float c2 = quad(X[(2 * ep) * 260 + 128 + r] + X[(2 * ep + 1) * 260 + 128 + r]) * (1.0f / 512.f);
          const float rs = __uint_as_float(rsc[c % 3][i]);
          const ef2v c1v = {-c1, -c1}, c2v = {-c2, -c2}, rsv = {rs, rs};
          const int so = (64 * ha + 16 * i) * cpitch;
          // The rows of t stay in their registers from pass 1b (a second load of them was a second trip to HBM - a round of tiles sweeps the
          // XCD's L2 - and the launch runs at the memory system's rate: it was what the epilogue cost).  Both packed rows are made opaque again:
          // otherwise the unpacked f32 pairs of pass 1b are KEPT for this pass - in scratch - instead of two shifts per pair here
          if (ha) asm volatile("" : "+v"(tside1[2 * i + hb])); else asm volatile("" : "+v"(tside0[2 * i + hb]));
          const eu4v tw = ha ? tside1[2 * i + hb] : tside0[2 * i + hb];
          asm volatile("" : "+v"(dtp[ha][i][hb]));
          eu4v od;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (LNB_ABL & 4) { od[e] = dtp[ha][i][hb][e] ^ tw[e]; AX[e][0] += rs + c1 + c2; continue; }
            const ef2v d = lo2(dtp[ha][i][hb][e]);
            const ef2v xh = (lo2(tw[e]) - Bt[e]) * IG[e];
            ef2v o = __builtin_elementwise_fma(d, G[e], c1v);
            o = __builtin_elementwise_fma(xh, c2v, o) * rsv;
            AX[e] += o;
            AG[e] = __builtin_elementwise_fma(d, xh, AG[e]);
            AB[e] += d;
            od[e] = pack2bf(o[0], o[1]);
          }
          if (hb) E_BSTORE16(od, cvo_, crs_, so, 64); else E_BSTORE16(od, cvo_, crs_, so, 0);
        }
        if constexpr (ha == 1) {
          if (LNB_ABL & 1) { tot_g[hb] = AG[0][0] + AG[3][1]; tot_b[hb] = AB[0][0] + AB[3][1]; tot_x[hb] = AX[0][0] + AX[3][1]; }
          else {
          tot_g[hb] = colred8(AG);
          tot_b[hb] = colred8(AB);
          tot_x[hb] = colred8(AX);
          }
        }
      };
      E_WAIT4(8, rsc[0]);                                  // chunk 0's rstd: newer are chunks 1 and 2 (4 + 4)
      chunk(std::integral_constant<int, 0>{});
      tload(std::integral_constant<int, 3>{});             // (chunk 0's registers)
      E_WAIT4(12, rsc[1]);                                 // chunk 1: newer are chunk 2 (4), chunk 0's stores (4), chunk 3 (4)
      chunk(std::integral_constant<int, 1>{});
      E_WAIT4(12, rsc[2]);                                 // chunk 2: newer are chunk 0's stores, chunk 3, chunk 1's stores
      chunk(std::integral_constant<int, 2>{});
      E_WAIT4(8, rsc[0]);                                  // chunk 3: newer are chunk 1's and chunk 2's stores
      chunk(std::integral_constant<int, 3>{});
      // every wave has read the exchange area before any wave's next tile refills the slot (LDS-DMA in phase 1 of its first K-tile)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      N_BAR();
      // lane (er, ep) keeps column 32 (er & 1) + 8 ep + (er >> 1) of the wave's 64: three registers run on over the workgroup's tiles
      run_g += (er & 1) ? tot_g[1] : tot_g[0];
      run_b += (er & 1) ? tot_b[1] : tot_b[0];
      run_x += (er & 1) ? tot_x[1] : tot_x[0];
    } else
    {
      // epilogue addressing (see gemm_bf16_e256): after the column swap lane (li, lq) holds 8 columns of a 32-column block of row li; the values
      // go through one lane transpose in LDS so that four adjacent lanes store 64 contiguous bytes of a row.  Derived HERE from an opaque copy of
      // the lane index: as loop invariants the compiler parked two dozen of them in scratch across the main loop and reloaded them in the
      // epilogue - every reload a vector-memory operation in the counted queue with a vmcnt(0) behind it
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int li_e = lane_e & 15, lq_e = lane_e >> 4;
      const int cq = ((lq_e & 1) << 1) | (lq_e >> 1);
      const int er = lane_e >> 2, ep = lane_e & 3;
      const unsigned cvo = (unsigned)((er * p.ldc + 256 * wr + 64 * wc + 8 * ep) * 2);
      const unsigned gvo = (unsigned)((er * p.ldr + 256 * wr + 64 * wc + 8 * ep) * 2);
      const unsigned xw16 = (unsigned)(li_e * 64 + ((cq ^ ((li_e >> 1) & 3)) << 4));
      const unsigned xr16 = (unsigned)(er * 64 + ((ep ^ ((er >> 1) & 3)) << 4));
      const int xsw = (li_e ^ ((li_e >> 1) & 1)) & 7, xsr = (er ^ ((er >> 1) & 1)) & 7;
      const unsigned xw32 = (unsigned)(li_e * 128), xr32 = (unsigned)(er * 128);
      (void)xw16; (void)xr16; (void)xsw; (void)xsr; (void)xw32; (void)xr32; (void)gvo; (void)cq;
      const ei4v crs = ersrc((bf16raw*)p.C + tm0 * p.ldc, (unsigned)(128 * p.ldc * 2));
      const int cpitch = (int)(p.ldc * 2);
      unsigned ones2 = 0x3f803f80u;   // the bf16 pair (1, 1), in a register the compiler cannot fold into an inline constant (see the gate epilogue of gemm_bf16_e256)
      asm volatile("" : "+s"(ones2));
      (void)ones2;
      eu4v yk[2][4][2];   // LN: the packed rows of y, kept for the statistics and the normalisation (the accumulators die as they are consumed)
      float rsum[2][4];   // LN: this lane's share of the row sums (rows 64 ha + 16 i + er, its 16 columns)
      eu4v gb[8];         // LN: gamma / beta of the lane's 16 columns: [hb][half] then + 4
      (void)yk; (void)rsum; (void)gb;
#pragma unroll
      for (int ha = 0; ha < 2; ha++) {
        if (ha == 0) {
          // bias and rows 0-63: issued in phase 4 of the last K-tile; newer: B00 / B10 (4 instructions)
          if (BIAS) E_WAIT4(4, biasr);
          if (L1) E_WAIT8(4, side0);
        }
        if (L1 && ha == 1) E_WAIT8(SH / 2, side1);   // newer: the four stores of rows 32-63
        f4v bx[2][2];
#pragma unroll
        for (int hb = 0; hb < 2; hb++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const eu4v b4 = biasr[hb * 2 + j];
            bx[hb][j] = BIAS ? (f4v){__uint_as_float(b4[0]), __uint_as_float(b4[1]), __uint_as_float(b4[2]), __uint_as_float(b4[3])} : (f4v){0.f, 0.f, 0.f, 0.f};
          }
        constexpr int NI = RES ? 1 : 2;
#pragma unroll
        for (int ib = 0; ib < 4; ib += NI) {
          if (L1 && ha == 0 && ib == 2) {
            // rows 64-127 of the residual go out HALFWAY through rows 0-63: by then half of that half's accumulators and side registers are free
            // (requested at the epilogue's start they sat beside everything else: 30-40 spilled registers and a vmcnt(0) at every reload)
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const int so = (64 + 16 * i) * spitch;
              E_BLOAD16(side1[2 * i], gvo, srs, so, 0);
              E_BLOAD16(side1[2 * i + 1], gvo, srs, so, 64);
            }
          }
          if (LN && !(N_DBG & 1) && ha == 1 && ib == 2) {   // (three quarters of the accumulators and side registers are free by now; only the four stores of rows 96-127 follow)
            // gamma / beta of the lane's columns 256 wr + 64 wc + 32 hb + 8 ep .. + 7 (two 16-byte halves each): needed in pass 2
            const ei4v grs = ersrc(q.gamma + 256 * wr + 64 * wc, 64 * 4), ers = ersrc(q.beta + 256 * wr + 64 * wc, 64 * 4);
            const unsigned gvoff = (unsigned)(8 * ep * 4);
            E_BLOAD16(gb[0], gvoff, grs, 0, 0); E_BLOAD16(gb[1], gvoff, grs, 0, 16); E_BLOAD16(gb[2], gvoff, grs, 0, 128); E_BLOAD16(gb[3], gvoff, grs, 0, 144);
            E_BLOAD16(gb[4], gvoff, ers, 0, 0); E_BLOAD16(gb[5], gvoff, ers, 0, 16); E_BLOAD16(gb[6], gvoff, ers, 0, 128); E_BLOAD16(gb[7], gvoff, ers, 0, 144);
          }
          eu4v o[NI][2];
#pragma unroll
          for (int ii = 0; ii < NI; ii++)
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              const int i = ib + ii;
              const f4v x = acc[ha][hb][i][0] + bx[hb][0];
              const f4v y = acc[ha][hb][i][1] + bx[hb][1];
              if (RES) {
                float v[8];
                *(f4v*)(xstg + xw32 + ((lq_e ^ xsw) << 4)) = x;
                *(f4v*)(xstg + xw32 + (((4 + lq_e) ^ xsw) << 4)) = y;
                const f4v r0 = *(const f4v*)(xstg + xr32 + (((2 * ep) ^ xsr) << 4));
                const f4v r1 = *(const f4v*)(xstg + xr32 + (((2 * ep + 1) ^ xsr) << 4));
#pragma unroll
                for (int e = 0; e < 4; e++) { v[e] = r0[e]; v[4 + e] = r1[e]; }
                const eu4v r4 = ha ? side1[2 * i + hb] : side0[2 * i + hb];
#pragma unroll
                for (int e = 0; e < 4; e++) { v[2 * e] += __uint_as_float(r4[e] << 16); v[2 * e + 1] += __uint_as_float(r4[e] & 0xffff0000u); }
                o[ii][hb][0] = pack2bf(v[0], v[1]); o[ii][hb][1] = pack2bf(v[2], v[3]); o[ii][hb][2] = pack2bf(v[4], v[5]); o[ii][hb][3] = pack2bf(v[6], v[7]);
                if (LN) {
                  yk[ha][i][hb] = o[ii][hb];
                  float sacc = hb ? rsum[ha][i] : 0.f;   // the LayerNorm kernel sums the ROUNDED values (layernorm_fwd4_k); (x, y) . (1, 1): one instruction per pair
#pragma unroll
                  for (int e = 0; e < 4; e++) sacc = edot2(o[ii][hb][e], ones2, sacc);
                  rsum[ha][i] = sacc;
                }
              } else {
                const unsigned px0 = pack2bf(x[0], x[1]), px1 = pack2bf(x[2], x[3]);
                const unsigned py0 = pack2bf(y[0], y[1]), py1 = pack2bf(y[2], y[3]);
                auto s0 = __builtin_amdgcn_permlane16_swap(px0, py0, false, false);
                auto s1 = __builtin_amdgcn_permlane16_swap(px1, py1, false, false);
                *(eu4v*)(xstg + hb * 1024 + xw16) = (eu4v){s0[0], s1[0], s0[1], s1[1]};
                o[ii][hb] = *(const eu4v*)(xstg + hb * 1024 + xr16);
              }
            }
#pragma unroll
          for (int ii = 0; ii < NI; ii++) {
            const int i = ib + ii;
            const int so = (64 * ha + 16 * i) * cpitch;
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              eu4v& ou = o[ii][hb];
              if (!STORE_Y) continue;
              if (hb) E_BSTORE16(ou, cvo, crs, so, 64); else E_BSTORE16(ou, cvo, crs, so, 0);
            }
          }
        }
      }
      if (LN) {
        // ---- the LayerNorm of the stored rows (layernorm_fwd4_k's arithmetic: mean, then the centred squares, both over the rounded values).
        // A row's 512 columns are spread over the eight waves: every wave leaves its 128 row partials in its own staging block (idle now),
        // one workgroup barrier, and lane (er, ep) adds the partials of waves 2 ep, 2 ep + 1 for its eight rows; a quad sum gives the total.
        float* const mine = (float*)xstg;                                    // [2][128] floats of this wave
        const float* const all = (const float*)(smem + N_XSTG);              // wave w: + 512 w floats
        auto quad = [&](float v) -> float {
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, false));
          v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, false));
          return v;
        };
        auto exchange = [&](float (&part)[2][4], int which) {   // in: this lane's shares; out: the totals of its eight rows
#pragma unroll
          for (int ha = 0; ha < 2; ha++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
              const float w = quad(part[ha][i]);
              if (ep == 0) mine[which * 128 + 64 * ha + 16 * i + er + 4 * wave] = w;   // (+ 4 floats per wave: see the read below)
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          N_BAR();
#pragma unroll
          for (int ha = 0; ha < 2; ha++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
              // wave w's partials sit 4 w floats into its 2 KiB block: the four waves 2 ep a half wave reads from then fall on four different
              // groups of 8 banks (at the bare 512-float pitch all of them hit the same 8: 4-way conflicts, 6.3 M conflict cycles per launch)
              const int r = which * 128 + 64 * ha + 16 * i + er;
              part[ha][i] = quad(all[(2 * ep) * 512 + r + 8 * ep] + all[(2 * ep + 1) * 512 + r + 8 * ep + 4]);
            }
        };
        if (!(N_DBG & 8)) exchange(rsum, 0);
        float mu[2][4], qs[2][4];
        float xc[2][4][2][8];   // the centred values: unpacked and centred ONCE, used by the variance and by the normalisation (the accumulators are dead: 128 registers)
#pragma unroll
        for (int ha = 0; ha < 2; ha++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            mu[ha][i] = rsum[ha][i] / 512.f;
            float a = 0.f;
#pragma unroll
            for (int hb = 0; hb < 2; hb++)
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float t0 = __uint_as_float(yk[ha][i][hb][e] << 16) - mu[ha][i], t1 = __uint_as_float(yk[ha][i][hb][e] & 0xffff0000u) - mu[ha][i];
                xc[ha][i][hb][2 * e] = t0; xc[ha][i][hb][2 * e + 1] = t1;
                a += t0 * t0; a += t1 * t1;
              }
            qs[ha][i] = a;
          }
        if (!(N_DBG & 8)) exchange(qs, 1);
        // mean / rstd of the tile's rows: wave w writes the 16 rows of its (ha, i) = (w >> 2, w & 3)
        if (!(N_DBG & 1)) E_WAIT8(SH / 2, gb);   // gamma / beta: newer are the four stores of rows 96-127
        const ei4v trs = ersrc((bf16raw*)q.t + tm0 * q.ldt, (unsigned)(128 * q.ldt * 2));
        const int tpitch = (int)(q.ldt * 2);
        const unsigned tvo = (unsigned)((er * q.ldt + 256 * wr + 64 * wc + 8 * ep) * 2);
#pragma unroll
        for (int ha = 0; ha < 2; ha++)
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float rs = 1.0f / sqrtf(qs[ha][i] / 512.f + q.eps);
            if (!(N_DBG & 2) && wave == ha * 4 + i && ep == 0) {
              q.mean[tm0 + 64 * ha + 16 * i + er] = mu[ha][i];
              q.rstd[tm0 + 64 * ha + 16 * i + er] = rs;
            }
#pragma unroll
            for (int hb = 0; hb < 2; hb++) {
              eu4v ot;
#pragma unroll
              for (int e = 0; e < 4; e++) {
                const float g0 = (N_DBG & 1) ? 1.f : __uint_as_float(gb[2 * hb + (e >> 1)][2 * (e & 1)]), g1 = (N_DBG & 1) ? 1.f : __uint_as_float(gb[2 * hb + (e >> 1)][2 * (e & 1) + 1]);
                const float b0 = (N_DBG & 1) ? 0.f : __uint_as_float(gb[4 + 2 * hb + (e >> 1)][2 * (e & 1)]), b1 = (N_DBG & 1) ? 0.f : __uint_as_float(gb[4 + 2 * hb + (e >> 1)][2 * (e & 1) + 1]);
                ot[e] = (N_DBG & 4) ? yk[ha][i][hb][e] : pack2bf(xc[ha][i][hb][2 * e] * rs * g0 + b0, xc[ha][i][hb][2 * e + 1] * rs * g1 + b1);
              }
              const int so = (64 * ha + 16 * i) * tpitch;
              if (hb) E_BSTORE16(ot, tvo, trs, so, 64); else E_BSTORE16(ot, tvo, trs, so, 0);
            }
          }
      }
    }
    if (!has_next) break;
    T += G;
    tm0 = nm0; cA = nA;
    has_next = T + G < nt;
    nm0 = tile_of(has_next ? T + G : T);
    nA = (const unsigned char*)p.A + nm0 * p.lda * 2;
    if (wr == 1) { N_BAR(); }  // the stagger again
  }
  if constexpr (LNB) {
    // this workgroup's partial column sums -> work[which][workgroup][512] (plain stores; layernorm_bwd_reduce_k adds the workgroups' rows)
    const int er = lane >> 2, ep = lane & 3;
    const int col = 256 * wr + 64 * wc + 32 * (er & 1) + 8 * ep + (er >> 1);
    float* w = q.work + (size_t)blockIdx.x * 512 + col;
    w[0] = run_g;
    w[(size_t)gridDim.x * 512] = run_b;
    w[(size_t)2 * gridDim.x * 512] = run_x;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last tile's surplus prefetches land before the LDS is released
#undef N_RD_A
#undef N_RD_B
#undef N_MFMA
#undef N_BAR
#undef N_LGKM0
#undef N_VMCNT
}
int g_gemm_nw = 0;   // pero_set_option("gemm_nw", 1): N = 512 stored products with the plain / residual epilogue on the row-complete tile
bool pero_launch_gemm_n512(const GemmP& p0, long long batch, bool ta, bool tb, bool out_f32, hipStream_t st) {
  if (!g_gemm_nw || batch != 1 || ta || tb || out_f32 || p0.N != 512 || p0.M % N_BM || p0.K % E_BK || p0.K < 3 * E_BK) return false;
  if (p0.alpha != 1.0f || p0.gate || (p0.flags & ~(PERO_GEMM_TILE256))) return false;   // bias and residual only
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22) || p0.ldc >= (1LL << 22) || (p0.resid && p0.ldr >= (1LL << 22))) return false;
  int num_cus = (pero_num_cus() / 8) * 8;
  if (num_cus < 8) num_cus = 8;
  const long long nt = p0.M / N_BM;
  const unsigned G = (unsigned)(nt < num_cus ? ((nt + 7) / 8) * 8 : num_cus);
  GemmP p = p0;
  const LnP q = {nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr};
#define LAUNCH_N(EP_, BI_)                                                                          \
  do {                                                                                              \
    PERO_LDS_ATTR((gemm_bf16_n512<EP_, BI_>), N_LDS_BYTES);                                          \
    hipLaunchKernelGGL((gemm_bf16_n512<EP_, BI_>), dim3(G), dim3(512), N_LDS_BYTES, st, p, q);       \
  } while (0)
  if (p0.resid) { if (p0.bias) LAUNCH_N(EP_RESID, true); else LAUNCH_N(EP_RESID, false); }
  else { if (p0.bias) LAUNCH_N(EP_PLAIN, true); else LAUNCH_N(EP_PLAIN, false); }
  return true;
}
// y = A W^T + bias + resid (bf16, stored), t = LayerNorm(y) * gamma + beta (bf16), mean / rstd of every row: one launch on the row-complete
// tile.  false: the shape does not take it (N must be 512).
bool pero_launch_gemm_n512_ln(const GemmP& p0, void* t, long long ldt, float* mean, float* rstd, const float* gamma, const float* beta, float eps,
                              hipStream_t st) {
  if (p0.N != 512 || p0.M % N_BM || p0.K % E_BK || p0.K < 3 * E_BK || !p0.resid || !t || !mean || !rstd || !gamma || !beta) return false;
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22) || (p0.C && p0.ldc >= (1LL << 22)) || p0.ldr >= (1LL << 22) || ldt >= (1LL << 22)) return false;
  int num_cus = (pero_num_cus() / 8) * 8;
  if (num_cus < 8) num_cus = 8;
  const long long nt = p0.M / N_BM;
  const unsigned G = (unsigned)(nt < num_cus ? ((nt + 7) / 8) * 8 : num_cus);
  GemmP p = p0;
  const LnP q = {t, ldt, mean, rstd, gamma, beta, eps, nullptr};
  if (!p0.C) { if (p0.bias) LAUNCH_N(EP_RESID_LN_T, true); else LAUNCH_N(EP_RESID_LN_T, false); }   // y not stored
  else if (p0.bias) LAUNCH_N(EP_RESID_LN, true); else LAUNCH_N(EP_RESID_LN, false);
#undef LAUNCH_N
  return true;
}

// dx = LayerNorm backward (from the norm's output t and rstd) of dt = A W^T + R, column sums -> work [3][grid][512]; *grid_out = the rows of work that
// layernorm_bwd_reduce_k has to add.  false: the shape does not take it.
bool pero_launch_gemm_n512_lnb(const GemmP& p0, const void* t, long long ldt, const float* rstd, const float* gamma, const float* beta, float* work,
                               int* grid_out, hipStream_t st) {
  if (p0.N != 512 || p0.M % N_BM || p0.K % E_BK || p0.K < 3 * E_BK || !p0.resid || p0.bias || !p0.C || !t || !rstd || !gamma || !beta || !work) return false;
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22) || p0.ldc >= (1LL << 22) || p0.ldr >= (1LL << 22) || ldt >= (1LL << 22)) return false;
  int num_cus = (pero_num_cus() / 8) * 8;
  if (num_cus < 8) num_cus = 8;
  const long long nt = p0.M / N_BM;
  const unsigned G = (unsigned)(nt < num_cus ? ((nt + 7) / 8) * 8 : num_cus);
  // (workgroups beyond the tile count return at once: their rows of `work` are cleared first - the reduce kernel adds all G rows)
  if (nt < (long long)G) hipMemsetAsync(work, 0, (size_t)3 * G * 512 * sizeof(float), st);
  GemmP p = p0;
  const LnP q = {const_cast<void*>(t), ldt, nullptr, const_cast<float*>(rstd), gamma, beta, 0.f, work};
  PERO_LDS_ATTR((gemm_bf16_n512<EP_RESID_LNB, false>), N_LDS_BYTES);
  hipLaunchKernelGGL((gemm_bf16_n512<EP_RESID_LNB, false>), dim3(G), dim3(512), N_LDS_BYTES, st, p, q);
  *grid_out = (int)G;
  return true;
}

// C[tile] += alpha * sum over the slices (in slice order) of the partial tiles the split-K work items left in the workspace.
// One thread per float4 position of a tile: consecutive threads read consecutive 16 bytes of every partial tile.
__global__ __launch_bounds__(256) void pero_splitk_reduce_k(const f4v* ws, float* C, long long ldc, int ntn, int nsl, float alpha, bool vec4) {
  const int tile = blockIdx.x >> 6;                                   // 64 blocks of 256 threads per 256 x 256 tile
  const int pos = ((blockIdx.x & 63) << 8) + threadIdx.x;             // (accumulator * 8 + wave) * 64 + lane
  const f4v* src = ws + (long long)tile * nsl * (E_BM * E_BN / 4) + pos;
  f4v sum = src[0];
  int z = 1;
  for (; z + 8 <= nsl; z += 8) {   // eight partial tiles requested together, added in slice order
    f4v v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = src[(long long)(z + k) * (E_BM * E_BN / 4)];
#pragma unroll
    for (int k = 0; k < 8; k++) sum += v[k];
  }
  for (; z < nsl; z++) sum += src[(long long)z * (E_BM * E_BN / 4)];
  const int lane = pos & 63, wave = (pos >> 6) & 7, idx = pos >> 9;
  const int ha = idx >> 4, hb = (idx >> 3) & 1, i = (idx >> 1) & 3, j = idx & 1;
  const long long row = (long long)(tile / ntn) * E_BM + 128 * (wave >> 2) + 64 * ha + 16 * i + (lane & 15);
  const long long col = (long long)(tile % ntn) * E_BN + 64 * (wave & 3) + 32 * hb + 16 * j + 4 * (lane >> 4);
  // plain read-modify-write, 16 bytes per thread when C allows it: C must have ONE writer at a time in this mode (pero_hip.h,
  // PERO_GEMM_ATOMIC with a workspace).  As four f32 atomics per thread the pass took 21 us instead of 14 per launch (50 launches a step).
  float* dst = C + row * ldc + col;
  if (vec4) {
    f4v* d4 = (f4v*)dst;
    *d4 = *d4 + sum * alpha;
  } else {
#pragma unroll
    for (int e = 0; e < 4; e++) dst[e] += sum[e] * alpha;
  }
}
__global__ __launch_bounds__(256) void pero_zero16_k(f4v* p, long long n16) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) p[i] = (f4v){0.f, 0.f, 0.f, 0.f};
}
int g_gemm_splitk_ws = 1;   // pero_set_option("splitk_workspace", 0): atomic epilogue
int g_gemm_splitk_table = 1;   // pero_set_option("splitk_table", 0): plain item order for unaligned slice counts

// Slice count of a split-K product on this kernel (k_split <= 0: the library chooses) and whether the slices can be aligned to the XCDs.
static int e256_splitk_slices(long long tiles, long long steps, int k_split, bool* xcd_ok) {
  int ks = k_split;
  if (ks <= 0) {  // the library chooses: one round of workgroups over the CUs
    ks = (int)(256 / tiles);
    if (ks >= 8) ks = (ks / 8) * 8; else if (ks >= 4) ks = 4; else if (ks >= 2) ks = 2; else ks = 1;
    // an XCD-aligned slice count (one slice set per XCD: operand panels fetched once per L2) when it fills >= 90 % of the CUs
    // that the plain count fills, else the plain count
    int kx = (int)(256 / tiles);
    kx = kx < 1 ? 1 : kx;
    if (ks * tiles * 10 < kx * tiles * 9) ks = kx;
  }
  while (ks > 1 && steps / ks < 2) ks--;
  if (xcd_ok) *xcd_ok = (ks == 1 || ks == 2 || ks == 4 || ks % 8 == 0) && (tiles * ks) % 8 == 0 && (ks >= 8 || tiles % (8 / ks) == 0);
  return ks;
}
// Bytes of caller-owned workspace with which a split-K product of this shape runs DETERMINISTICALLY on this kernel: every (tile, slice)
// work item leaves its f32 partial tile there by plain stores and pero_splitk_reduce_k adds the slices of a tile in slice order.  0: the
// shape does not take this kernel or has one slice.  Without (enough) workspace the product falls back to f32 atomics.
long long pero_gemm_e256_splitk_ws_bytes(long long M, long long N, long long K, int k_split) {
  if (M % E_BM || N % E_BN || K % E_BK || K < 2 * E_BK) return 0;
  const long long tiles = (M / E_BM) * (N / E_BN);
  const int ks = e256_splitk_slices(tiles, K / E_BK, k_split, nullptr);
  return ks > 1 ? tiles * ks * (long long)(E_BM * E_BN * sizeof(float)) : 0;
}

// Qualifies: one problem (batch 1), bf16 operands; stored bf16 output (alpha == 1) or a split-K f32 product; M % 256 == N % 256 == K % 64 == 0, K >= 128.
// ws / ws_bytes: the caller's workspace for the split-K partial tiles (pero_gemm's `workspace`); never allocated here.
int g_gemm_e_var = 0;
int g_gemm_e_walk = 1;   // pero_set_option("gemm_e_walk", 0): every stored product in the side-by-side order
bool pero_launch_gemm_e256(const GemmP& p0, long long batch, int k_split, bool ta, bool tb, bool out_f32, hipStream_t st, int var, void* ws,
                           long long ws_bytes) {
  if (p0.M % E_BM || p0.N % E_BN || p0.K % E_BK || p0.K < 2 * E_BK || batch != 1) return false;
  if (var < 0) var = g_gemm_e_var;
  const int wg_cap = ((var >> 8) & 0xff) * 8;  // diagnostic: at most this many workgroups (bits 8-15 of the variant, in units of 8)
  const int walk = (var >> 16) & 0x3f;         // stored products (bit 5: the A-rows-from-L2 timing probe): N-tiles of a row panel per workgroup, one after the other (bits 16-19; see the kernel's tile_of)
  var &= 0xff;
  int ks = 0;
  if (p0.flags & PERO_GEMM_ATOMIC) {
    // split-K: f32 C, plain product, equal slices of whole K-tiles, one slice set per XCD
    if (!out_f32 || p0.bias || p0.resid || (p0.gate && !(var & (64 | 128))) || (p0.flags & ~(PERO_GEMM_ATOMIC | PERO_GEMM_TRANS_A | PERO_GEMM_TRANS_B | PERO_GEMM_TILE_V | PERO_GEMM_TILE256))) return false;
    const long long tiles = (p0.M / E_BM) * (p0.N / E_BN), steps = p0.K / E_BK;
    bool xcd_ok = false;
    ks = e256_splitk_slices(tiles, steps, k_split, &xcd_ok);
    if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22)) return false;
    GemmP p = p0;
    p.kchunk = 0;
    dim3 grid((unsigned)(tiles * ks)), block(512);
    const int nsl = ks;
    const bool ws_ok = ws && (((size_t)ws) & 15) == 0 && ws_bytes >= tiles * ks * (long long)(E_BM * E_BN * sizeof(float));
    p.resid = (g_gemm_splitk_ws && ws_ok && !(var & (64 | 128)) && nsl > 1) ? ws : nullptr;
    auto reduce = [&]() {
      if (p.resid)
        hipLaunchKernelGGL(pero_splitk_reduce_k, dim3((unsigned)(tiles * 64)), dim3(256), 0, st, (const f4v*)p.resid, (float*)p.C, (long long)p.ldc,
                           (int)(p.N / E_BN), nsl, p.alpha, p.ldc % 4 == 0 && (((size_t)p.C) & 15) == 0);
    };
    if (!xcd_ok) {
      ks = -ks;
      p.kchunk = (g_gemm_splitk_table && !(var & (64 | 128))) ? 1 : 0;   // work items handed out XCD by XCD (esplitk_xcd_item)
    }
#define LAUNCH_ES(TA_, TB_)                                                                                                \
  do {                                                                                                                     \
    PERO_LDS_ATTR((gemm_bf16_e256<TA_, TB_, EP_SPLITK, 0>), E_LDS_BYTES);                                                  \
    hipLaunchKernelGGL((gemm_bf16_e256<TA_, TB_, EP_SPLITK, 0>), grid, block, E_LDS_BYTES, st, p, ks);                    \
  } while (0)
    if ((var & 128) && ta && tb) {
      PERO_LDS_ATTR((gemm_bf16_e256<true, true, EP_SPLITK, 128>), E_LDS_BYTES);
      hipLaunchKernelGGL((gemm_bf16_e256<true, true, EP_SPLITK, 128>), grid, block, E_LDS_BYTES, st, p, ks);
      return true;
    }
    if ((var & 64) && ta && tb) {
      PERO_LDS_ATTR((gemm_bf16_e256<true, true, EP_SPLITK, 64>), E_LDS_BYTES);
      hipLaunchKernelGGL((gemm_bf16_e256<true, true, EP_SPLITK, 64>), grid, block, E_LDS_BYTES, st, p, ks);
      return true;
    }
    if (!ta && !tb) LAUNCH_ES(false, false); else if (!ta && tb) LAUNCH_ES(false, true); else if (ta && tb) LAUNCH_ES(true, true); else LAUNCH_ES(true, false);
#undef LAUNCH_ES
    reduce();
    return true;
  }
  if (k_split > 1 || out_f32 || (p0.flags & PERO_GEMM_ACCUM)) return false;
  if (p0.alpha != 1.0f) return false;
  if (p0.lda >= (1LL << 22) || p0.ldb >= (1LL << 22) || p0.ldc >= (1LL << 22)) return false;  // 32-bit byte offsets inside a tile
  const bool relu = p0.flags & PERO_GEMM_RELU, bits = p0.flags & PERO_GEMM_RELU_BITS, rowdot = p0.flags & PERO_GEMM_ROWDOT,
             cs = p0.flags & PERO_GEMM_COLSUM;
  int epi;
  if ((var & 128) && p0.resid && !relu && !bits && !rowdot && !cs) epi = EP_RESID;   // stamp build of the residual epilogue
  else if (var & (8 | 128)) epi = EP_PLAIN;          // stamp builds: `gate` is the stamp buffer
  else if (rowdot) { if (relu || bits || cs || p0.resid || !p0.gate || !p0.bias) return false; epi = EP_ROWDOT; }
  else if (bits) {
    if (!p0.gate || p0.resid || (relu && cs)) return false;
    // the gate epilogue has no input-bias path (its `bias` is the column-sum OUTPUT under PERO_GEMM_COLSUM): a gated product WITH an
    // input bias goes to gemm_bf16_r256, which adds it - both kernels then give the same bits at every tile count
    if (!relu && p0.bias && !cs) return false;
    epi = relu ? EP_RELU_BITS : EP_GATE_BITS;
  }
  else if (cs || p0.gate) return false;              // column sums without the bit mask, bf16 gate rows: other kernels
  else if (p0.resid) { if (relu) return false; epi = EP_RESID; }
  else epi = relu ? EP_RELU : EP_PLAIN;
  if ((epi == EP_RESID && p0.ldr >= (1LL << 22)) || ((epi == EP_ROWDOT || bits) && p0.ldg >= (1LL << 22))) return false;
  if (epi != EP_PLAIN && (ta || tb)) return false;   // the fused epilogues exist for the K-contiguous products only
  int num_cus = (pero_num_cus() / 8) * 8;
  if (num_cus < 8) num_cus = 8;
  GemmP p = p0;
  p.kchunk = p.K;
  const long long nt = (p.M / E_BM) * (p.N / E_BN);
  unsigned G = (unsigned)(nt < num_cus ? ((nt + 7) / 8) * 8 : num_cus);
  if (wg_cap && (unsigned)wg_cap < G) G = (unsigned)wg_cap;
  dim3 grid(G), block(512);
  // Walk of the stored K <= 512 products (the kernel's tile_of): by default the ntn workgroups of an XCD that share a 256-row panel of A run its ntn N-tiles side
  // by side and wait for the same bytes from HBM together.  With each workgroup taking `seq` N-tiles of its panel one after the other, seq x as many panels are in
  // flight per XCD and the panel's later passes come from the caches: 524 288 x 2048 x 512 plain / ReLU 1 056 -> 1 010 us, bit-mask gate 1 086 -> 1 054, N = 1536
  // 784 -> 772, N = 4096 2 117 -> 1 926 (seq 4 / 3; tools/e256_walk2.py).  NOT for the epilogue that writes the ReLU bit mask in ROWS (its 32 bytes per row and tile are a
  // quarter of a line: written rounds apart they cost more than the walk gains, 1 095 -> 1 147; with PERO_GEMM_MASK_TILED a tile's mask is whole lines), not at K = 2048 (+- 1 %).  Same tiles, same bits.
  ks = walk;
  if (!(walk & 31) && g_gemm_e_walk && p.K <= 512 && (epi == EP_PLAIN || epi == EP_RELU || epi == EP_GATE_BITS || (epi == EP_RELU_BITS && (p.flags & PERO_GEMM_MASK_TILED))) && nt >= 2LL * G) {
    const long long ntn = p.N / E_BN;
    ks = (walk & 32) | ((ntn >= 8 && ntn % 4 == 0) ? 4 : (ntn >= 6 && ntn % 3 == 0) ? 3 : 0);
  }
  if (epi == EP_ROWDOT) {  // the two waves of a 128-column block add into it: cleared first (hipMemsetAsync's fill kernel took 27 us for these 4 MB)
    const long long n = p.M * (p.N >> 7);
    if (n % 4 == 0 && aligned16(p.bias))
      hipLaunchKernelGGL(pero_zero16_k, dim3((unsigned)((n / 4 + 255) / 256 < 2048 ? (n / 4 + 255) / 256 : 2048)), dim3(256), 0, st, (f4v*)p.bias, n / 4);
    else
      hipMemsetAsync((void*)p.bias, 0, (size_t)n * sizeof(float), st);
  }
#define LAUNCH_E(TA_, TB_, EP_, VAR_)                                                                                            \
  do {                                                                                                                     \
    PERO_LDS_ATTR((gemm_bf16_e256<TA_, TB_, EP_, VAR_>), E_LDS_BYTES);                                                     \
    hipLaunchKernelGGL((gemm_bf16_e256<TA_, TB_, EP_, VAR_>), grid, block, E_LDS_BYTES, st, p, ks);                                  \
  } while (0)
  if (!ta && !tb) {
    switch (epi) {
      case EP_RELU: LAUNCH_E(false, false, EP_RELU, 0); break;
      case EP_RESID: if (var == 128) LAUNCH_E(false, false, EP_RESID, 128); else if (var == 2) LAUNCH_E(false, false, EP_RESID, 2); else if (var == 16) LAUNCH_E(false, false, EP_RESID, 16); else LAUNCH_E(false, false, EP_RESID, 0); break;
      case EP_RELU_BITS: LAUNCH_E(false, false, EP_RELU_BITS, 0); break;
      case EP_GATE_BITS: LAUNCH_E(false, false, EP_GATE_BITS, 0); break;
      case EP_ROWDOT: LAUNCH_E(false, false, EP_ROWDOT, 0); break;
      default:
        switch (var) {
          case 1: LAUNCH_E(false, false, EP_PLAIN, 1); break;
          case 2: LAUNCH_E(false, false, EP_PLAIN, 2); break;
          case 4: LAUNCH_E(false, false, EP_PLAIN, 4); break;
          case 8: LAUNCH_E(false, false, EP_PLAIN, 8); break;
          case 10: LAUNCH_E(false, false, EP_PLAIN, 10); break;
          case 12: LAUNCH_E(false, false, EP_PLAIN, 12); break;
          case 16: LAUNCH_E(false, false, EP_PLAIN, 16); break;
          case 32: LAUNCH_E(false, false, EP_PLAIN, 32); break;
          case 72: LAUNCH_E(false, false, EP_PLAIN, 72); break;
          case 128: LAUNCH_E(false, false, EP_PLAIN, 128); break;
          case 132: LAUNCH_E(false, false, EP_PLAIN, 132); break;
          default: LAUNCH_E(false, false, EP_PLAIN, 0); break;
        }
    }
  }
  else if (!ta && tb) LAUNCH_E(false, true, EP_PLAIN, 0);
  else if (ta && tb) LAUNCH_E(true, true, EP_PLAIN, 0);
  else LAUNCH_E(true, false, EP_PLAIN, 0);
#undef LAUNCH_E
  return true;
}
