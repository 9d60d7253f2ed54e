// chromegcn_amd/csrc/cgcn_common.hpp -- helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chromegcn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define TILE_NODES 16   // nodes per workgroup tile in the gather kernels
#ifndef BWD_MAX_PARTIALS
#define BWD_MAX_PARTIALS 256
#endif

// ------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores ("split products", round 6; profiles/r06_bf16x6_probe.txt).
// On gfx950 v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (64 flop/clk/SIMD) and keeps the SIMD's vector ALUs busy;
// v_mfma_f32_16x16x32_bf16 is 16 times faster and leaves half of its issue slots to vector work.  A float has 24
// significant bits = three bf16 numbers of 8:  x = h + m + l EXACTLY (h = bf16(x), m = bf16(x - h), l = x - h - m, which has
// <= 8 significant bits left; both subtractions are exact), so
//     a b = ah bh + (ah bm + am bh) + (ah bl + am bm + al bh) + [am bl + al bm + al bl  <= 2^-23 |a b|, dropped]
// is six bf16 MFMAs whose partial products are exact and whose sums are kept in three fp32 accumulators by magnitude
// (big / mid / small, added once at the end).  Measured against float64 (K = 256, 524 288 dot products, in units of
// 2^-24 sum |a_k b_k|): fp32 MFMA chain rms 0.43 / worst 4.0, this form rms 0.15 / worst 1.6 -- the chain rounds 64
// times, this form 8 times per accumulator.  Same fp32 range (bf16 has the fp32 exponent): no scaling, Inf / NaN propagate.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// Levels by ROUND-TO-NEAREST (v_cvt_pk_bf16_f32), not by truncation: h = bf16(x), m = bf16(x - h), l = x - h - m (<= 8
// significant bits: exact) -- the residuals are signed and half as large, so the three dropped partial products are zero-mean.
// With truncated levels every residual has the sign of x and the dropped terms the sign of a b: a relative bias of ~5e-8 on every
// product, COHERENT over rows -- the gate-bias gradients (sums of ~10^5 cancelling per-row terms, tests/test_gpu_fullsize_oracle.py)
// amplified it to 1.2e-4 at chr1 size with hubs, above the chain's 3.5e-5.  Same instruction count (one convert per pair and
// level instead of two masks and a byte permute).
__device__ __forceinline__ uint32_t sp_pack(float x0, float x1) {   // (bf16 x1) << 16 | (bf16 x0), round to nearest even
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
}
__device__ __forceinline__ float sp_lo(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float sp_up(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
// two consecutive floats -> one bf16 pair of each level
__device__ __forceinline__ void sp_split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = sp_pack(x0, x1);
  const float r0 = x0 - sp_lo(h), r1 = x1 - sp_up(h);
  m = sp_pack(r0, r1);
  l = sp_pack(r0 - sp_lo(m), r1 - sp_up(m));
}
// one float -> its three levels (bf16 bit patterns)
__device__ __forceinline__ void sp_split1(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
  uint32_t h2, m2, l2;
  sp_split2(x, 0.f, h2, m2, l2);
  h = (uint16_t)h2;
  m = (uint16_t)m2;
  l = (uint16_t)l2;
}
// four consecutive floats -> 4 bf16 of each level (8 bytes per level)
__device__ __forceinline__ void sp_split4(const f32x4 x, u32x2& h, u32x2& m, u32x2& l) {
  uint32_t h0, m0, l0, h1, m1, l1;
  sp_split2(x[0], x[1], h0, m0, l0);
  sp_split2(x[2], x[3], h1, m1, l1);
  h = (u32x2){h0, h1};
  m = (u32x2){m0, m1};
  l = (u32x2){l0, l1};
}
// eight consecutive K values -> one MFMA operand of each level
__device__ __forceinline__ void sp_split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
  u32x2 h0, m0, l0, h1, m1, l1;
  sp_split4((f32x4){x[0], x[1], x[2], x[3]}, h0, m0, l0);
  sp_split4((f32x4){x[4], x[5], x[6], x[7]}, h1, m1, l1);
  h = __builtin_bit_cast(bf16x8, (u32x4){h0[0], h0[1], h1[0], h1[1]});
  m = __builtin_bit_cast(bf16x8, (u32x4){m0[0], m0[1], m1[0], m1[1]});
  l = __builtin_bit_cast(bf16x8, (u32x4){l0[0], l0[1], l1[0], l1[1]});
}
struct SpAcc {   // the three accumulators of one 16 x 16 output block
  f32x4 big, mid, small;
  __device__ __forceinline__ void zero() { big = mid = small = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // one K-step of 32: the six partial products, ordered so that no MFMA follows one into the same accumulator
  __device__ __forceinline__ void step(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm,
                                       const bf16x8 bl) {
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, small, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, mid, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, big, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, small, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, mid, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, small, 0, 0, 0);
  }
  __device__ __forceinline__ f32x4 sum() const { return big + (mid + small); }
};
struct SpAcc2 {   // two accumulators (the leading product | the five others): four registers fewer, the same error to 2^-32
  f32x4 big, rest;
  __device__ __forceinline__ void zero() { big = rest = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void step(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh, const bf16x8 bm,
                                       const bf16x8 bl) {
    rest = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, rest, 0, 0, 0);
    rest = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, rest, 0, 0, 0);
    rest = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, rest, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, big, 0, 0, 0);
    rest = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, rest, 0, 0, 0);
    rest = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, rest, 0, 0, 0);
  }
  __device__ __forceinline__ f32x4 sum() const { return big + rest; }
};

// LDS image of a bf16 LEVEL TILE (16 rows x 128 columns, 256-byte rows, 4 KB; one per level and operand): the 16-byte chunk c
// of row m sits at c ^ sp_sigma(m), sp_sigma(m) = (m & 3) << 2 | tau[m >> 2], tau = {2, 0, 1, 3}.  Conflict-free for 8-byte
// row stores, for the ds_read_b128 row reads of a 16x16x32 operand (sigma of rows 4..11 is closed under ^ 1: the
// instruction's lane groups {0-3, 12-15 | 20-27} ... meet 16 distinct chunks) and for the TRANSPOSED reads a product over the
// ROW index needs (ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 columns, each lane receives one column's 4 rows;
// the 4 rows of a block differ in sigma >> 2: 16 distinct chunks per 32-lane half).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sp_sigma(int m) { return ((m & 3) << 2) | ((0xD2 >> ((m >> 2) << 1)) & 3); }
__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {   // rows 8h..8h+3 | 8h+4..8h+7
  const s16x4 a = lds_tr16(p0), b = lds_tr16(p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  return __builtin_bit_cast(bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// v of the lane that DPP control CTRL pairs this lane with; 0 in the rows ROW_MASK leaves out
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes, the same value in every lane.  Must be called with all lanes active.  Six DPP adds (data
// parallel primitives: register-to-register lane permutes on the VALU, no LDS crossbar) and one v_readlane: pairs,
// quads, half rows and rows of 16 by quad_perm / row_half_mirror / row_mirror, then row_bcast15 / row_bcast31 carry
// the row totals up to lane 63.  (Six dependent ds_bpermute round trips -- the __shfl_xor butterfly -- were a
// 600-cycle latency chain in every row-wise pass.)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_get<0x140, 0xF>(v);   // row_mirror: every lane holds its row's total
  v += dpp_get<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the wave's total
  return rl_f(v, 63);
}

// Sum over each 32-lane half of the wave, the half's total in every lane of the half (all lanes active).
__device__ __forceinline__ float half_sum(float v, bool upper) {
  v += dpp_get<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_get<0x140, 0xF>(v);   // row_mirror: every lane holds its row's total
  v += dpp_get<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3: they hold the totals of lanes 0-31 / 32-63
  const float lo = rl_f(v, 31), hi = rl_f(v, 63);
  return upper ? hi : lo;
}
// ... of N independent values at once, step by step: a DPP instruction needs two wait states behind the instruction that
// wrote its source, and the other values' steps are exactly that (one value alone: an s_nop per step)
template <int N>
__device__ __forceinline__ void half_sum_n(float (&v)[N], bool upper) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0xB1, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x4E, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x141, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x140, 0xF>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x142, 0xA>(v[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float lo = rl_f(v[i], 31), hi = rl_f(v[i], 63);
    v[i] = upper ? hi : lo;
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Chan's exact merge of two (count, mean, sum of squared deviations) summaries: A <- A u B
__device__ __forceinline__ void chan_combine(float& nA, float& meanA, float& m2A, float nB, float meanB, float m2B) {
  const float nAB = nA + nB;
  if (nAB > 0.f) {
    const float delta = meanB - meanA;
    meanA += delta * (nB / nAB);
    m2A += m2B + delta * delta * (nA * nB / nAB);
    nA = nAB;
  }
}


// Counter-based dropout RNG: a mask bit is a pure function of (seed, step counter, stream id, element
// index), so the backward regenerates the forward's mask instead of storing it.  rng_state lives in
// device memory ([0] = seed, [1] = step counter) so that a captured HIP graph sees a fresh counter on
// every replay.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// tanh of the gated layer's pre-activation.  CGCN_FAST_TANH (A/B builds; profiles/r05_fast_tanh_experiment.txt, measured again
// behind the split products in profiles/r06_split_products_ab.txt): the same two branches as the library's tanhf -- a series
// below 0.25, 1 - 2 / (e^{2|x|} + 1) above -- on the hardware exp2 / rcp without the range reduction around them: 15 instead
// of 27 vector instructions, |error| <= 1e-7 (<= 6 ulp between 0.25 and 1, <= 1.2 ulp elsewhere; tanhf: <= 1.4 ulp).
#ifndef CGCN_FAST_TANH
#define CGCN_FAST_TANH 0
#endif
__device__ __forceinline__ float layer_tanh(float x) {
#if CGCN_FAST_TANH
  const float a = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(a * 2.885390081777927f);   // e^(2a)
  const float r = __builtin_amdgcn_rcpf(e + 1.f);
  float z = __builtin_fmaf(-2.f, r, 1.f);
  const float x2 = x * x;
  const float p = a * (1.f + x2 * (-0.33333333f + x2 * (0.13333333f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  z = a < 0.25f ? p : z;
  return copysignf(z, x);
#else
  return tanhf(x);
#endif
}

__device__ __forceinline__ uint32_t dropout_key(const unsigned long long* rng_state, uint32_t stream_id) {
  const unsigned long long seed = rng_state[0], ctr = rng_state[1];
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9u);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k + (uint32_t)ctr * 0x85EBCA6Bu);
  k = mix32(k ^ (uint32_t)(ctr >> 32) ^ (stream_id * 0xC2B2AE35u));
  return k;
}
__device__ __forceinline__ bool dropout_keep(uint32_t key, uint32_t elem, uint32_t thresh) {
  return mix32(elem * 0x9E3779B1u + key) >= thresh;
}
static inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

// ------------------------------------------------------------------------------------------
// Accumulate mode of the head's BatchNorm batch statistics (round 5, d = 128, split-size tables): instead of one
// (mean, M2) record per k_layer_dense workgroup that k_head_bn_finalize merges in a launch of its own -- a launch that
// costs 6-7 us of the epoch per chromosome although it computes for 3 (profiles/r05_tiny_kernels_bound.txt) -- every
// workgroup ADDS its sum x and sum x^2 per (strand, column) to STAT_ACC_SLOTS copies of the totals as 64-bit FIXED-POINT
// integer atomics: integer addition is associative, so the totals are the same bits whatever order the workgroups arrive
// in (float atomics would not be), and the head kernel derives mean / invstd from them in its prologue.
//   word ((slot S + s) D + c) 2 + q,  q = 0: sum relu(x), q = 1: sum relu(x)^2, value * 2^STAT_ACC_FBITS
//   word STAT_ACC_SLOTS S D 2: nonzero when a workgroup's partial did not fit: the statistics then read NaN (loudly wrong
//   instead of silently wrapped)
// One word per value, 32 fraction bits: resolution 1.2e-10 per add (<= 6e-8 on a total: 4e-6 of sum x^2 even for |x| ~ 1e-3
// at n = 15 000); a workgroup's partial must stay below 2^22 so that the total of <= 512 workgroups stays below 2^31, i.e.
// sum relu(x)^2 < 2.1e9 per column -- rms |x| < 265 at n = 30 000.  (Two words would lift the range but cost 3-6 us per
// launch instead of 1: tools/micro/atomic_tail.hip; the engine falls back to records when a chromosome's features are
// outside the range, finetune.GCNStage.add_chromosome.)
// Slots: 512 adders on one word serialise (+10 us per word, same file); 8 copies cost +1 us.
// ------------------------------------------------------------------------------------------
constexpr int STAT_ACC_SLOTS = 8;
constexpr int STAT_ACC_FBITS = 32;
// Backward block behind it (same buffer, zeroed by the same launch): the head's BatchNorm-backward column sums
// sum dy / sum dy xhat per (strand, column) as ONE word each into the same 8 copies, so that k_head_train_finish is not
// launched either, then four header words: overflow flag, the two binary points, the loss total, a ticket.
//   Their rounding error is coherent over the rows and amplified ~1 000x downstream (head_part_stride below): they need
//   ~1e-10 relative precision AND have no fixed magnitude, so the binary point comes from bounds every workgroup reproduces:
//   sum |dy| <= keep_scale max|W_out| (|d loss / d pred| <= 1 / (n C) per element) and |xhat| <= sqrt(n).
__host__ __device__ __forceinline__ constexpr size_t stat_acc_fwd_words(int S, int d) { return (size_t)STAT_ACC_SLOTS * S * d * 2 + 2; }
__host__ __device__ __forceinline__ constexpr size_t stat_acc_bwd_words(int S, int d) { return (size_t)STAT_ACC_SLOTS * S * d * 2 + 4; }
static inline size_t stat_acc_words(int S, int d) { return stat_acc_fwd_words(S, d) + stat_acc_bwd_words(S, d); }
enum { BACC_FLAG = 0, BACC_EXP = 1, BACC_LOSS = 2, BACC_SPARE = 3 };   // header words behind the backward sums
constexpr unsigned long long BACC_LOSS_BAD = 1ull << 63;   // ORed into the loss word: a share or a backward sum was out of range
constexpr int BACC_LOSS_FBITS = 16;   // the loss total: per-element BCE terms (>= 0, <= ~30 each, n C < 2^31 of them) above bit 12 of its word,
                                      // the arrival count below (<= 4 095 workgroups): share and ticket are ONE atomic
__device__ __forceinline__ unsigned long long* bacc_base(const unsigned long long* acc, int S, int D) {
  return (unsigned long long*)acc + stat_acc_fwd_words(S, D);
}
// fraction bits for a sum bounded by `bound` (> 0): the bound sits below 2^61
__device__ __forceinline__ int bacc_fbits(double bound) { return 60 - (ilogb(bound < 1e-300 ? 1e-300 : bound) + 1); }
__device__ __forceinline__ void bacc_add(unsigned long long* b, int S, int D, int slot, int s, int c, double sdy, double sdyx,
                                         int fa, int fb) {
  unsigned long long* w = b + ((size_t)(slot * S + s) * D + c) * 2;
  const double va = ldexp(sdy, fa), vb = ldexp(sdyx, fb);
  if (!(__builtin_fabs(va) < 1.15e18) || !(__builtin_fabs(vb) < 1.15e18)) {   // 2^60: outside the bounds (or NaN): loud
    atomicOr(b + (size_t)STAT_ACC_SLOTS * S * D * 2 + BACC_FLAG, 1ull);          // the row-local backward's prologue reads NaN
    atomicOr(b + (size_t)STAT_ACC_SLOTS * S * D * 2 + BACC_LOSS, BACC_LOSS_BAD); // ... and so does the loss (head_loss_ticket)
    return;
  }
  atomicAdd(&w[0], (unsigned long long)(long long)__builtin_rint(va));
  atomicAdd(&w[1], (unsigned long long)(long long)__builtin_rint(vb));
}
// bnc of (strand s, column c): mean dy, mean dy xhat (for d loss = 1), from the totals
__device__ __forceinline__ void bacc_get(const unsigned long long* __restrict__ b, int S, int D, int s, int c, int n, float& c0, float& c1) {
  long long ta = 0, tb = 0;
#pragma unroll
  for (int slot = 0; slot < STAT_ACC_SLOTS; ++slot) {
    const unsigned long long* w = b + ((size_t)(slot * S + s) * D + c) * 2;
    ta += (long long)w[0];
    tb += (long long)w[1];
  }
  const unsigned long long* h = b + (size_t)STAT_ACC_SLOTS * S * D * 2;
  const int fa = (int)(unsigned)(h[BACC_EXP] & 0xFFFFFFFFull) - 1024, fb = (int)(unsigned)(h[BACC_EXP] >> 32) - 1024;
  const double invn = 1.0 / (double)n;
  c0 = (float)(ldexp((double)ta, -fa) * invn);
  c1 = (float)(ldexp((double)tb, -fb) * invn);
  if (h[BACC_FLAG] != 0ull) c0 = c1 = __builtin_nanf("");
}
// lim: what one adder may contribute so that the total of ALL adders stays below 2^31 -- 2^22 for the <= 512 workgroups of the
// row-local kernels; the one-launch forward has a workgroup per 16 / S-node tile and passes 2^31 / its grid
__device__ __forceinline__ void stat_acc_add(unsigned long long* acc, int S, int D, int slot, int s, int c, double sum1, double sum2,
                                             double lim = 4194304.0) {
  const double sc = (double)(1ll << STAT_ACC_FBITS);
  unsigned long long* w = acc + ((size_t)(slot * S + s) * D + c) * 2;
  if (!(__builtin_fabs(sum1) < lim) || !(sum2 < lim)) {
    atomicOr(acc + (size_t)STAT_ACC_SLOTS * S * D * 2, 1ull);
    return;
  }
  atomicAdd(&w[0], (unsigned long long)(long long)__builtin_rint(sum1 * sc));
  atomicAdd(&w[1], (unsigned long long)(long long)__builtin_rint(sum2 * sc));
}
// mean, biased variance sum M2 = sum (x - mean)^2 of (strand s, column c) over n rows from the totals
__device__ __forceinline__ void stat_acc_get(const unsigned long long* __restrict__ acc, int S, int D, int s, int c, int n,
                                             double& mean, double& m2) {
  long long t1 = 0, t2 = 0;
#pragma unroll
  for (int slot = 0; slot < STAT_ACC_SLOTS; ++slot) {
    const unsigned long long* w = acc + ((size_t)(slot * S + s) * D + c) * 2;
    t1 += (long long)w[0];
    t2 += (long long)w[1];
  }
  const double isc = 1.0 / (double)(1ll << STAT_ACC_FBITS);
  const double s1 = (double)t1 * isc, s2 = (double)t2 * isc;
  mean = s1 / (double)n;
  m2 = s2 - mean * s1;
  if (m2 < 0.0) m2 = 0.0;
  if (acc[(size_t)STAT_ACC_SLOTS * S * D * 2] != 0ull) mean = m2 = __builtin_nan("");
}

#define HEAD_STREAM_ID 0x4845u  // dropout stream of the classifier head ("HE")

// Optional prologue of k_bwd_rowlocal for the LAST gated layer: instead of reading dL/dXn it is
// recomputed per row from the head's backward state (BatchNorm backward + ReLU + dropout mask), which
// removes one launch and one write+read of an [S,n,d] tensor.  dym == nullptr disables it.
struct HeadApply {
  const float* dym;      // [n][d]   d loss / d (mean over strands of the dropped BatchNorm output)
  const float* bnc;      // [S][2][d] per-strand mean(dy), mean(dy * xhat)
  const unsigned long long* bacc;   // accumulate mode (STAT_ACC_*): the buffer whose backward block replaces bnc; else nullptr
  const float* mean;     // [S][d]   batch mean of relu(Xn)
  const float* invstd;   // [S][d]
  const float* bn_w;     // [d]
  const unsigned long long* rng_state;
  float keep_scale;
  uint32_t thresh;
  int S;
  // deferred second stage of the head's dW_out / db_out sums (run by extra workgroups of k_bwd_rowlocal)
  const float* hf_part;  // [P][head_part_stride(CP, D)] partials of k_head_fused / k_head_bwd
  float* hf_dWout;       // [C][D]
  float* hf_dbout;       // [C]
  int hf_P, hf_C, hf_CP, hf_accumulate;
  const float* dloss;    // [1] upstream d loss, or nullptr when dym / the partials already include it
  float* hf_dbn_w;       // [D] when set, the extra workgroups also finish d(bn weight) / d(bn bias) (workspace of
  float* hf_dbn_b;       //     cgcn_head_train: nobody has summed the BatchNorm columns for the parameters yet)
};

// Partial record of one k_head_fused / k_head_bwd workgroup (floats):
//   [CP*D dW_out][CP db_out][4*D doubles: sum dy (strand 0, 1), sum dy*xhat (strand 0, 1) per column]
// The BatchNorm-backward column sums are kept in float64 from the per-row term to the final mean: bnc = (mean dy,
// mean dy*xhat) is a per-COLUMN constant that enters every row's dL/dXn, so its rounding error is coherent over the
// rows and is amplified by the cancellation of the gate-bias sum dcg = sum_i gamma_i (|sum| / sum|.| = 1.7e-3 at
// chr21 size): measured, fp32 sums gave dW2.bias 2.2e-4 off the float64 truth where every other stage of the layer
// backward is at 5e-7 (tools/bias_sum_probe.py, profiles/r03_bias_sum_probe.txt).
__host__ __device__ __forceinline__ constexpr int head_part_stride(int CP, int D) { return CP * D + CP + 8 * D; }
#define HEAD_STAT_COLS 16   // BatchNorm columns per workgroup of the statistics second stage

// One 64-element slab of the head backward's second stage: elements [CP*D dW_out][CP db_out].
// EXT: the staging buffer is the caller's LDS (`scratch`, NT floats) instead of a static array of this function -- a
// kernel whose own tiles already fill the LDS lends them to its trailing workgroups.
template <int NT, bool EXT = false>
__device__ __forceinline__ void head_finalize_slab(int slab, int P, int D, int C, int CP,
                                                   const float* __restrict__ part, float* __restrict__ dWout,
                                                   float* __restrict__ dbout, int accumulate,
                                                   const float* __restrict__ dloss, void* scratch = nullptr) {
  constexpr int NS = NT / 64;
  const float gl = dloss ? dloss[0] : 1.f;  // everything summed here is linear in the upstream d loss
  const int PS = head_part_stride(CP, D);
  const int total = CP * D + CP;
  float (*hred)[64];
  if constexpr (EXT) {
    hred = (float (*)[64])scratch;
  } else {
    __shared__ __attribute__((aligned(16))) float hred_own[NS][64];
    hred = hred_own;
  }
  // lane = (el4, sub): 16 lanes x float4 cover the slab's 64 elements, the 4 sub-groups of a wave and the NS
  // waves each take a contiguous range of the P partials -> one or two batches of independent 16-byte loads
  // per thread instead of a long chain of 4-byte ones (the kernel is latency-, not bandwidth-limited).
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el4 = lane & 15, sub = lane >> 4;
  const int e0 = slab * 64 + el4 * 4;  // region boundaries are multiples of 4: a float4 never straddles one
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (e0 < total) {
    const int per = (P + NS * 4 - 1) / (NS * 4);
    const int p0 = (wave * 4 + sub) * per, p1 = min(P, p0 + per);
    for (int p = p0; p < p1; p += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(part + (size_t)min(p + u, p1 - 1) * PS + e0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + u < p1) a += t[u];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = a[k];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    a[k] = v;
  }
  if (sub == 0) *(f32x4*)&hred[wave][el4 * 4] = a;
  __syncthreads();
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = slab * 64 + el;
  if (slice != 0 || e >= total) return;
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NS; ++w) s += hred[w][el];
  s *= gl;
  if (e < CP * D) {
    const int i = e / D;
    if (i < C) dWout[e] = accumulate ? dWout[e] + s : s;
  } else {
    const int j = e - CP * D;
    if (j < C) dbout[j] = accumulate ? dbout[j] + s : s;
  }
}

// Second stage of the BatchNorm-backward column sums, in float64: columns [blk * 16, blk * 16 + 16).  Thread =
// (column, slice of the partial list); slices are merged through LDS in a fixed order => deterministic.
// Writes d(bn bias) = sum dy, d(bn weight) = sum dy*xhat (both strands; scaled by the upstream d loss) and/or
// bnc = the per-strand means for d loss = 1.
template <int NT, bool EXT = false, int BATCH = 8>
__device__ __forceinline__ void head_stats_finalize(int blk, int P, int n, int S, int D, int CP,
                                                    const float* __restrict__ part, float* __restrict__ dbn_w,
                                                    float* __restrict__ dbn_b, float* __restrict__ bnc, int accumulate,
                                                    const float* __restrict__ dloss, void* scratch = nullptr) {
  constexpr int NSL = NT / HEAD_STAT_COLS;
  double (*sred)[NSL][HEAD_STAT_COLS + 1];   // EXT: 4 * NSL * 17 doubles of the caller's LDS (see head_finalize_slab)
  if constexpr (EXT) {
    sred = (double (*)[NSL][HEAD_STAT_COLS + 1])scratch;
  } else {
    __shared__ double sred_own[4][NSL][HEAD_STAT_COLS + 1];
    sred = sred_own;
  }
  const int cl = threadIdx.x % HEAD_STAT_COLS, slice = threadIdx.x / HEAD_STAT_COLS;
  const int c = blk * HEAD_STAT_COLS + cl;
  const int PS = head_part_stride(CP, D);
  const int per = (P + NSL - 1) / NSL;
  const int p0 = slice * per, p1 = min(P, p0 + per);
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  if (c < D) {
    // batches of 8 records: all 32 loads of a batch are issued before the first add (the kernel is a latency chain:
    // with two records per trip the 8 records of a 256-partial launch were four dependent round trips); same
    // summation order as a plain loop
    for (int p = p0; p < p1; p += BATCH) {   // (BATCH = 4 inside kernels that live on 64 registers: same summation order)
      double t[BATCH][4];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const double* st = (const double*)(part + (size_t)min(p + u, p1 - 1) * PS + CP * D + CP);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) t[u][qd] = st[qd * D + c];
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
        if (p + u < p1) {
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) a[qd] += t[u][qd];
        }
    }
  }
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) sred[qd][slice][cl] = a[qd];
  __syncthreads();
  if (slice != 0 || c >= D) return;
  double s[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    double t = 0.0;
#pragma unroll 8
    for (int o = 0; o < NSL; ++o) t += sred[qd][o][cl];
    s[qd] = t;
  }
  // s[0], s[1] = sum dy (strand 0, 1); s[2], s[3] = sum dy*xhat (strand 0, 1); strand 1 is zero when S == 1
  const double gl = dloss ? (double)dloss[0] : 1.0;
  const float db_ = (float)((s[0] + s[1]) * gl), dg_ = (float)((s[2] + s[3]) * gl);
  if (dbn_b) dbn_b[c] = accumulate ? dbn_b[c] + db_ : db_;
  if (dbn_w) dbn_w[c] = accumulate ? dbn_w[c] + dg_ : dg_;
  if (bnc) {
    const double invn = 1.0 / (double)n;
    for (int st = 0; st < S; ++st) {
      bnc[(st * 2 + 0) * D + c] = (float)(s[st] * invn);
      bnc[(st * 2 + 1) * D + c] = (float)(s[2 + st] * invn);
    }
  }
}

// Row accesses of EPL = D/64 consecutive floats per lane.  Through a plain float* the compiler may only assume 4-byte
// alignment and emits one global_load_dword per element (256 B per wave-instruction); every such pointer in this
// library is 16-byte aligned at a multiple of EPL floats, so say so: one 8- or 16-byte access per lane.
template <int N> struct RowVec;
template <> struct RowVec<2> { typedef f32x2 T; };
template <> struct RowVec<4> { typedef f32x4 T; };
template <int N>
__device__ __forceinline__ void ld_row(float (&dst)[N], const float* __restrict__ p) {
  const typename RowVec<N>::T v = *(const typename RowVec<N>::T*)p;
#pragma unroll
  for (int e = 0; e < N; ++e) dst[e] = v[e];
}
template <int N>
__device__ __forceinline__ void st_row(float* __restrict__ p, const float (&src)[N]) {
  typename RowVec<N>::T v;
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = src[e];
  *(typename RowVec<N>::T*)p = v;
}
template <int N>
__device__ __forceinline__ void zero_row(float (&dst)[N]) {
#pragma unroll
  for (int e = 0; e < N; ++e) dst[e] = 0.f;
}

static inline bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? CGCN_OK : CGCN_ERR_LAUNCH; }
