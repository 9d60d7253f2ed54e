// chromegcn_amd/csrc/cgcn_metrics.hip
//
// Multi-label ranking metrics on the device.  The reference computes them on
// the CPU with one scikit-learn call per label per metric after every split (runner.py:41,45,51 ->
// utils/evals.py:89-92 -> utils/metrics.py:148-183,238-253): AUROC, area under the precision-recall curve
// (trapezoid over sklearn's precision_recall_curve points), recall at the first point with FDR <= cutoff,
// and average precision (mAP).  Here, two paths that share everything but the sort:
//   * scores that are probabilities (non-negative: what the reference passes, finetune.py:52): 32-bit keys
//     [descending score image | target], label c's keys in the contiguous segment [c n, (c + 1) n), sorted by the segmented
//     radix sort written in this file (cgcn_multilabel_metrics_nonneg; see "Non-negative scores" below);
//   * any float scores: ONE device-wide radix sort (rocprim::radix_sort_keys, called directly) of 64-bit keys
//     (label | descending score | target) -- every label's list comes out contiguous and sorted and the whole chip works on
//     it (a rocprim segmented sort with one long segment per label used a fraction of the chip) -- cgcn_multilabel_metrics;
// then a chunked scan of the sorted lists (one wave per 4096-element chunk), treating tied scores as one curve point exactly
// like sklearn's distinct-threshold curves.  All curve arithmetic is fp64 and summed in a fixed order; counts are integers
// (ballot + popcount).
#include <cstring>  // rocprim 4.x headers use memset without including it
#include <rocprim/rocprim.hpp>

#include "cgcn_common.hpp"

// Sort key of one (window i, label c) pair: [label c][~ordered(score)][target bit].  ordered() is the usual
// order-preserving map of IEEE floats to unsigned; complemented so that an ascending sort lists each label's scores in
// descending order.  -0 is folded onto +0 (sklearn compares scores as numbers).
__device__ __forceinline__ unsigned long long metric_key(int c, float score, float target) {
  unsigned u = __float_as_uint(score == 0.f ? 0.f : score);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
  return ((unsigned long long)(unsigned)c << 33) | ((unsigned long long)(~u) << 1) | (target > 0.5f ? 1ull : 0ull);
}
__device__ __forceinline__ float metric_key_score(unsigned long long k) {
  unsigned u = ~(unsigned)(k >> 1);
  u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
  return __uint_as_float(u);
}

// [n,C] row-major -> keys[c * n + i], transposed through LDS so that both sides are coalesced
__global__ __launch_bounds__(256) void k_metrics_pack(long long n, int C, const float* __restrict__ probs,
                                                      const float* __restrict__ targets,
                                                      unsigned long long* __restrict__ keys) {
  __shared__ float tp[32][33];
  __shared__ float tt[32][33];
  const long long i0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long long i = i0 + r;
    const int c = c0 + tx;
    const bool ok = i < n && c < C;
    tp[r][tx] = ok ? probs[i * C + c] : 0.f;
    tt[r][tx] = ok ? targets[i * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long i = i0 + tx;
    if (c < C && i < n) keys[(long long)c * n + i] = metric_key(c, tp[tx][r], tt[tx][r]);
  }
}

// sorted 64-bit keys -> the score / target arrays the curve kernels read
__global__ __launch_bounds__(256) void k_metrics_unpack(long long items, const unsigned long long* __restrict__ keys,
                                                        float* __restrict__ score, unsigned char* __restrict__ val) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= items) return;
  const unsigned long long k = keys[i];
  score[i] = metric_key_score(k);
  val[i] = (unsigned char)(k & 1ull);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// Curve points are the ends of runs of equal scores in the descending-sorted list of one label:
// (tp_k, fp_k), k = 1..K, plus the origin.
//   AUROC   = sum (fp_k - fp_{k-1}) (tp_k + tp_{k-1}) / 2 / (P N)                       roc_auc_score
//   AUPR    = sum (r_k - r_{k-1}) (p_k + p_{k-1}) / 2, (r_0, p_0) = (0, 1)              auc(recall, precision)
//   AP      = sum (r_k - r_{k-1}) p_k                                                   average_precision_score
//   R@FDR   = r_k of the LAST point with 1 - p_k <= cutoff, 0 if none                   utils/metrics.py:148-166
// with p_k = tp_k / (tp_k + fp_k), r_k = tp_k / P (sklearn: r_k = 1 when P = 0).
// Each label's list is cut into chunks of METRIC_CHUNK elements, one wave per (label, chunk):
//   k_metrics_summary  per chunk: positives, and the position / local tp of its last run end
//   k_metrics_prefix   per label (serial over <= a few hundred chunk records): positives before each chunk and
//                      the curve point preceding it; total positives
//   k_metrics_chunks   per chunk: the trapezoid / step sums of its own curve points (fp64) + R@FDR candidate
//   k_metrics_final    per label: fixed-order sum over chunks  => deterministic
#ifndef METRIC_CHUNK
#define METRIC_CHUNK 4096
#endif

struct ChunkRec {      // written by k_metrics_summary, completed by k_metrics_prefix
  double pos;          // positives in the chunk
  double end_tp;       // positives in the chunk up to and including its last run end
  long long end_idx;   // global index + 1 of that run end, 0 if the chunk has none
  double carry_tp;     // positives before the chunk
  double prev_tp, prev_fp;  // curve point preceding the chunk's first run end
};
struct ChunkOut { double auc, aupr, ap, rfdr; int has_fdr; int pad; };

// Source of a label's sorted list.  K32 = false: float scores + one byte per target (the general path, after
// k_metrics_unpack); K32 = true: the 32-bit keys of the non-negative-score path, read as they were sorted
// ([31-bit descending score image][target]).
template <bool K32>
struct MetricSrc {
  const float* k;
  const unsigned char* v;
  const unsigned* q;
  __device__ __forceinline__ MetricSrc(const void* keys, const unsigned char* vals, long long off)
      : k(K32 ? nullptr : (const float*)keys + off), v(K32 ? nullptr : vals + off), q(K32 ? (const unsigned*)keys + off : nullptr) {}
  // One element per lane and step, as a raw word; a chunk's loop requests step s + 1 before it works on step s (a wave
  // that waited for its own 256 bytes in each of its 64 steps spent the launch waiting), and takes element i + 1 from the
  // neighbouring lane -- lane 63 from lane 0 of the next step.  K32: the word is the sorted key; else: score bits, and the
  // target byte rides in `t`.
  struct Word { unsigned w; unsigned t; };
  __device__ __forceinline__ Word load(long long i, long long n) const {
    Word r = {0u, 0u};
    if (i < n) {
      if (K32) r.w = q[i];
      else { r.w = __float_as_uint(k[i]); r.t = v[i]; }
    }
    return r;
  }
  // (is element i a positive, does a run of tied scores end at i) for i < i1 <= n; i + 1 == n ends the last run
  __device__ __forceinline__ void at(const Word& cur, const Word& nxt_step, long long i, long long n, bool ok, int lane,
                                     bool& pos, bool& end) const {
    unsigned nb = __shfl_down(cur.w, 1, WAVE);
    const unsigned first_next = __shfl(nxt_step.w, 0, WAVE);
    if (lane == WAVE - 1) nb = first_next;
    if (K32) {
      pos = ok && (cur.w & 1u);
      end = ok && ((i + 1 == n) || (cur.w >> 1) != (nb >> 1));
    } else {
      pos = ok && cur.t != 0;
      end = ok && ((i + 1 == n) || __uint_as_float(cur.w) != __uint_as_float(nb));
    }
  }
};
__device__ __forceinline__ double readlane_d(double v, int l) {   // l: wave-uniform
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// 1 / x for x >= 1 (counts): the hardware estimate + two Newton steps (relative error ~1e-16)
__device__ __forceinline__ double rcp_d(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ unsigned long long lanes_le(int lane) { return lane >= 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull); }
__device__ __forceinline__ unsigned long long lanes_lt(int lane) { return (1ull << lane) - 1ull; }

// The positives up to a lane are a popcount of a ballot (targets are bits): no scan, and every count is an exact integer.
// Loads run MDEPTH steps ahead of the arithmetic: with one 256-byte request in flight per wave the launch moved 1 TB/s.
constexpr int MDEPTH = 4;
static_assert((METRIC_CHUNK / WAVE) % MDEPTH == 0, "a chunk is a whole number of load groups");
template <bool K32>
__global__ __launch_bounds__(64) void k_metrics_summary(long long n, int nch, const void* __restrict__ keys,
                                                        const unsigned char* __restrict__ vals, ChunkRec* __restrict__ rec) {
  using Word = typename MetricSrc<K32>::Word;
  const int c = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x;
  const MetricSrc<K32> src(keys, vals, (long long)c * n);
  const long long i0 = (long long)ch * METRIC_CHUNK, i1 = min(n, i0 + METRIC_CHUNK);
  int carry = 0, end_tp = 0;
  long long end_idx = 0;
  Word cur[MDEPTH], nxt[MDEPTH];
#pragma unroll
  for (int j = 0; j < MDEPTH; ++j) cur[j] = src.load(i0 + j * WAVE + lane, n);
  for (long long g0 = i0; g0 < i1; g0 += MDEPTH * WAVE) {
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) nxt[j] = src.load(g0 + (MDEPTH + j) * WAVE + lane, n);   // (past the chunk's end too: the next chunk's first element closes this one's last run)
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) {
      const long long base = g0 + j * WAVE, i = base + lane;
      bool pos, end;
      src.at(cur[j], j + 1 < MDEPTH ? cur[(j + 1) % MDEPTH] : nxt[0], i, n, i < i1, lane, pos, end);
      const unsigned long long pb = __ballot(pos), eb = __ballot(end);
      if (eb) {
        const int hi = 63 - __builtin_clzll(eb);
        end_tp = carry + __builtin_popcountll(pb & lanes_le(hi));
        end_idx = base + hi + 1;
      }
      carry += __builtin_popcountll(pb);
    }
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) cur[j] = nxt[j];
  }
  if (lane == 0) {
    ChunkRec& r = rec[(size_t)c * nch + ch];
    r.pos = (double)carry;
    r.end_tp = (double)end_tp;
    r.end_idx = end_idx;
  }
}

// one wave per label, 64 chunk records per step (every quantity is an integer-valued double: the sums are exact in any order)
__global__ __launch_bounds__(64) void k_metrics_prefix(int nch, ChunkRec* __restrict__ rec, double* __restrict__ Ptot) {
  const int c = blockIdx.x, lane = threadIdx.x;
  ChunkRec* r = rec + (size_t)c * nch;
  double carry = 0.0, ptp = 0.0, pfp = 0.0;   // positives / last curve point before this group of chunks
  for (int k0 = 0; k0 < nch; k0 += WAVE) {
    const int k = k0 + lane;
    const bool ok = k < nch;
    const double pos = ok ? r[k].pos : 0.0;
    const double end_tp = ok ? r[k].end_tp : 0.0;
    const long long end_idx = ok ? r[k].end_idx : 0;
    double incl = pos;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const double o = __shfl_up(incl, off, WAVE);
      if (lane >= off) incl += o;
    }
    const double carry_k = carry + (incl - pos);
    const bool has = ok && end_idx > 0;
    const double my_tp = carry_k + end_tp, my_fp = (double)end_idx - my_tp;   // the curve point this chunk leaves behind
    const unsigned long long hb = __ballot(has), before = hb & lanes_lt(lane);
    const int srcl = before ? 63 - __builtin_clzll(before) : 0;
    const double s_tp = __shfl(my_tp, srcl, WAVE), s_fp = __shfl(my_fp, srcl, WAVE);
    if (ok) {
      r[k].carry_tp = carry_k;
      r[k].prev_tp = before ? s_tp : ptp;
      r[k].prev_fp = before ? s_fp : pfp;
    }
    const int last = hb ? 63 - __builtin_clzll(hb) : 0;
    const double l_tp = __shfl(my_tp, last, WAVE), l_fp = __shfl(my_fp, last, WAVE);
    if (hb) {
      ptp = l_tp;
      pfp = l_fp;
    }
    carry += __shfl(incl, WAVE - 1, WAVE);
  }
  if (lane == 0) Ptot[c] = carry;
}

template <bool K32>
__global__ __launch_bounds__(64) void k_metrics_chunks(long long n, int nch, const void* __restrict__ keys,
                                                       const unsigned char* __restrict__ vals,
                                                       const ChunkRec* __restrict__ rec, const double* __restrict__ Ptot,
                                                       double fdr_cutoff, ChunkOut* __restrict__ outp) {
  using Word = typename MetricSrc<K32>::Word;
  const int c = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x;
  const MetricSrc<K32> src(keys, vals, (long long)c * n);
  const long long i0 = (long long)ch * METRIC_CHUNK, i1 = min(n, i0 + METRIC_CHUNK);
  const ChunkRec r = rec[(size_t)c * nch + ch];
  const double P = Ptot[c];
  const double invP = P > 0.0 ? 1.0 / P : 0.0;
  int carry_tp = (int)r.carry_tp;   // exact: counts (n C < 2^31)
  // the curve point preceding this wave-step's first run end: counts, and its precision / recall (the origin: 1 / 0)
  double prev_tp = r.prev_tp, prev_fp = r.prev_fp;
  double prev_prec = (prev_tp + prev_fp) > 0.0 ? prev_tp * rcp_d(prev_tp + prev_fp) : 1.0;
  double prev_rec = (prev_tp + prev_fp) > 0.0 ? (P > 0.0 ? prev_tp * invP : 1.0) : 0.0;
  double s_auc = 0.0, s_aupr = 0.0, s_ap = 0.0, r_fdr = 0.0;
  int has_fdr = 0;
  Word cur[MDEPTH], nxt[MDEPTH];
#pragma unroll
  for (int j = 0; j < MDEPTH; ++j) cur[j] = src.load(i0 + j * WAVE + lane, n);
  for (long long g0 = i0; g0 < i1; g0 += MDEPTH * WAVE) {
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) nxt[j] = src.load(g0 + (MDEPTH + j) * WAVE + lane, n);
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) {
      const long long base = g0 + j * WAVE, i = base + lane;
      bool pos, end;
      src.at(cur[j], j + 1 < MDEPTH ? cur[(j + 1) % MDEPTH] : nxt[0], i, n, i < i1, lane, pos, end);
      const unsigned long long pb = __ballot(pos), eb = __ballot(end);
      const double tp = (double)(carry_tp + (int)__builtin_popcountll(pb & lanes_le(lane)));
      const double tot = (double)(int)(i + 1);   // tp + fp
      const double fp = tot - tp;
      // every lane's own point (used by the lanes whose previous run end it is; garbage where `end` is false)
      const double prec = tp * rcp_d(tot);      // (quotients by a reciprocal refined to full double precision: the IEEE
      const double rec_ = P > 0.0 ? tp * invP : 1.0;   //  divisions were most of this kernel's instructions)
      // the previous run end: inside this wave-step (lane srcl), else the carried point
      const unsigned long long before = eb & lanes_lt(lane);
      const int srcl = before ? 63 - __builtin_clzll(before) : 0;
      const double q_tp = __shfl(tp, srcl, WAVE), q_fp = __shfl(fp, srcl, WAVE);
      const double q_pr = __shfl(prec, srcl, WAVE), q_rc = __shfl(rec_, srcl, WAVE);
      const double ptp = before ? q_tp : prev_tp, pfp = before ? q_fp : prev_fp;
      const double pp = before ? q_pr : prev_prec, pr = before ? q_rc : prev_rec;
      if (end) {   // per-lane partial sums, folded across the wave ONCE after the loop (fixed order: deterministic)
        s_auc += (fp - pfp) * (tp + ptp) * 0.5;
        s_aupr += (rec_ - pr) * (prec + pp) * 0.5;
        s_ap += (rec_ - pr) * prec;
      }
      // recall at FDR <= cutoff: utils/metrics.py:153-154 compares 1 - tps / (tps + fps) (an IEEE quotient) with the cutoff,
      // and precision is EXACTLY 1/2 whenever tp = fp at a run end -- 7 * (1/14 refined) is not.  Where the fast quotient
      // lands within 1e-9 of the cutoff the predicate is taken from the true division (a handful of lanes per label).
      double fdr = 1.0 - prec;
      if (end && __builtin_fabs(fdr - fdr_cutoff) < 1e-9) fdr = 1.0 - tp / tot;
      const unsigned long long qb = __ballot(end && fdr <= fdr_cutoff);
      const int last = eb ? 63 - __builtin_clzll(eb) : 0;
      const int lq = qb ? 63 - __builtin_clzll(qb) : 0;
      // (wave-uniform source lanes: v_readlane, not a cross-lane permute)
      const double l_tp = readlane_d(tp, last), l_fp = readlane_d(fp, last);
      const double l_pr = readlane_d(prec, last), l_rc = readlane_d(rec_, last);
      const double f_rc = readlane_d(rec_, lq);
      if (qb) {
        r_fdr = f_rc;
        has_fdr = 1;
      }
      if (eb) {
        prev_tp = l_tp; prev_fp = l_fp; prev_prec = l_pr; prev_rec = l_rc;
      }
      carry_tp += (int)__builtin_popcountll(pb);
    }
#pragma unroll
    for (int j = 0; j < MDEPTH; ++j) cur[j] = nxt[j];
  }
  s_auc = wave_sum_d(s_auc);
  s_aupr = wave_sum_d(s_aupr);
  s_ap = wave_sum_d(s_ap);
  if (lane == 0) {
    ChunkOut& o = outp[(size_t)c * nch + ch];
    o.auc = s_auc; o.aupr = s_aupr; o.ap = s_ap; o.rfdr = r_fdr; o.has_fdr = has_fdr;
  }
}

__global__ __launch_bounds__(64) void k_metrics_final(long long n, int C, int nch, const ChunkOut* __restrict__ outp,
                                                      const double* __restrict__ Ptot, float* __restrict__ out) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const ChunkOut* o = outp + (size_t)c * nch;
  double s_auc = 0.0, s_aupr = 0.0, s_ap = 0.0, r_fdr = 0.0;
  int best = -1;
  for (int k = lane; k < nch; k += WAVE) {   // lane-strided partial sums, then one butterfly: a fixed order => deterministic
    s_auc += o[k].auc; s_aupr += o[k].aupr; s_ap += o[k].ap;
    if (o[k].has_fdr) { best = k; r_fdr = o[k].rfdr; }   // the deepest chunk with a qualifying point wins
  }
  s_auc = wave_sum_d(s_auc);
  s_aupr = wave_sum_d(s_aupr);
  s_ap = wave_sum_d(s_ap);
  int top = best;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, WAVE));
  const unsigned long long wb = __ballot(best == top && top >= 0);
  r_fdr = wb ? __shfl(r_fdr, __builtin_ctzll(wb), WAVE) : 0.0;
  if (lane != 0) return;
  const double P = Ptot[c], N = (double)n - P;
  const float nanv = __int_as_float(0x7fc00000);
  out[0 * C + c] = (P > 0.0 && N > 0.0) ? (float)(s_auc / (P * N)) : nanv;  // undefined with one class present
  out[1 * C + c] = n > 0 ? (float)s_aupr : nanv;
  out[2 * C + c] = n > 0 ? (float)r_fdr : nanv;
  out[3 * C + c] = n > 0 ? (float)s_ap : nanv;  // 0 when the label has no positive (as sklearn)
}

// ------------------------------------------------------------------------------------------
// Non-negative scores (probabilities: what the reference's compute_metrics is given, utils/evals.py:26): the sign bit of
// the score is known, so [31-bit descending image of the score][target] is a 32-bit key, the label needs no key bits (the
// pack kernel already writes label c's keys to the contiguous segment [c n, (c + 1) n)), and the sort is written here: a
// SEGMENTED least-significant-digit radix sort, 8-bit digits over key bits 1 .. 31 (4 passes; the target bit does not
// take part -- ties are one curve point whatever their order), every pass three launches over all labels at once:
//   k_rs_pass<false>  per 4096-key tile: digit counts (LDS atomics, a lane's runs of equal digits merged) -> hist[label][tile][256]
//   k_rs_scan         per label: exclusive prefix of a digit's counts over the tiles (in place) and over the digits
//   k_rs_pass<true>   per tile: stable local ranks, keys staged in LDS in digit order, runs written out coalesced
// 12 B per key and pass (two reads, one write) instead of 16 B x 2 for five device-wide passes over 64-bit keys, and no
// unpack pass: the curve kernels read the sorted keys.
// Local ranks: a wave takes 64 consecutive keys per round; the lanes that share a digit find each other with eight
// ballots (one per digit bit), rank = the wave's running count of the digit (LDS, in-order per wave: no barrier) + the
// number of peers in lower lanes.  Stable by construction (tile order = index order).
// ------------------------------------------------------------------------------------------
#ifndef RS_TILE_KEYS
#define RS_TILE_KEYS 4096   // keys per tile (256 threads x 16).  8192 (512 threads): the scatter's runs double (32 keys per digit and tile) and a
                            // rank-free first pass drops 60 -> 48 us, but the ranked passes lose occupancy (60 -> 67 us): no gain overall
#endif
constexpr int RS_TILE = RS_TILE_KEYS, RS_THREADS = RS_TILE / 16, RS_ROUNDS = RS_TILE / RS_THREADS, RS_WAVES = RS_THREADS / WAVE;

// keys[c * n + i] = [0x7FFFFFFF - bits(score)][target]; *bad |= 1 when a score is negative or NaN (the result is then
// unspecified: the caller falls back to the general path).  -0 is folded onto +0.
__global__ __launch_bounds__(256) void k_metrics_pack32(long long n, int C, const float* __restrict__ probs,
                                                        const float* __restrict__ targets, unsigned* __restrict__ keys,
                                                        int* __restrict__ bad) {
  __shared__ float tp[32][33];
  __shared__ float tt[32][33];
  const long long i0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long long i = i0 + r;
    const int c = c0 + tx;
    const bool ok = i < n && c < C;
    tp[r][tx] = ok ? probs[i * C + c] : 0.f;
    tt[r][tx] = ok ? targets[i * C + c] : 0.f;
  }
  __syncthreads();
  bool neg = false;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r;
    const long long i = i0 + tx;
    if (c < C && i < n) {
      const float sc = tp[tx][r];
      const unsigned u = __float_as_uint(sc == 0.f ? 0.f : sc);
      neg |= u > 0x7F800000u;   // sign bit set, or a NaN
      keys[(long long)c * n + i] = ((0x7FFFFFFFu - (u & 0x7FFFFFFFu)) << 1) | (tt[tx][r] > 0.5f ? 1u : 0u);
    }
  }
  if (__any(neg) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// The same keys from FLAT reads: a block takes R consecutive rows of probs / targets as one contiguous range (16 bytes per
// lane when the arrays are 16-byte aligned: VEC = 4), turns every element into its key on the spot, parks it in LDS at
// [label][row] and writes every label's R keys as one run.  The 32 x 32-tile transposition above reads 128-byte row
// segments 412 bytes apart (C = 103): 3.3 TB/s; this form is bound by the stream.
template <int VEC>
__global__ __launch_bounds__(256) void k_metrics_pack32_flat(long long n, int C, int R, int rshift, const float* __restrict__ probs,
                                                             const float* __restrict__ targets, unsigned* __restrict__ keys,
                                                             int* __restrict__ bad) {
  extern __shared__ unsigned tk[];   // [C][R + 1]
  const long long i0 = (long long)blockIdx.x * R;
  const int rows = (int)min((long long)R, n - i0);
  const int total = rows * C;
  const float* pp = probs + i0 * C;
  const float* tt = targets + i0 * C;
  bool neg = false;
  for (int e = (int)threadIdx.x * VEC; e < total; e += 256 * VEC) {
    float pv[VEC], tv[VEC];
    if (VEC == 4 && e + 4 <= total) {
      const float4 a = *(const float4*)(pp + e), b = *(const float4*)(tt + e);
      pv[0] = a.x; pv[1 % VEC] = a.y; pv[2 % VEC] = a.z; pv[3 % VEC] = a.w;
      tv[0] = b.x; tv[1 % VEC] = b.y; tv[2 % VEC] = b.z; tv[3 % VEC] = b.w;
    } else {
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        pv[u] = e + u < total ? pp[e + u] : 0.f;
        tv[u] = e + u < total ? tt[e + u] : 0.f;
      }
    }
    int r = e / C, c = e - r * C;
#pragma unroll
    for (int u = 0; u < VEC; ++u) {
      if (e + u < total) {
        const unsigned b = __float_as_uint(pv[u] == 0.f ? 0.f : pv[u]);
        neg |= b > 0x7F800000u;   // sign bit set, or a NaN
        tk[c * (R + 1) + r] = ((0x7FFFFFFFu - (b & 0x7FFFFFFFu)) << 1) | (tv[u] > 0.5f ? 1u : 0u);
      }
      if (++c == C) { c = 0; ++r; }
    }
  }
  if (__any(neg) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
  __syncthreads();
  for (int idx = threadIdx.x; idx < (C << rshift); idx += 256) {
    const int c = idx >> rshift, r = idx & (R - 1);
    if (r < rows) keys[(long long)c * n + i0 + r] = tk[c * (R + 1) + r];
  }
}

template <bool SCATTER>
__global__ __launch_bounds__(RS_THREADS) void k_rs_pass(long long n, int T, int shift, const unsigned* __restrict__ in,
                                                         unsigned* __restrict__ out, unsigned* __restrict__ hist,
                                                         const unsigned* __restrict__ base) {
  const int tile = blockIdx.x, seg = blockIdx.y;
  const unsigned* src = in + (long long)seg * n;
  const long long i0 = (long long)tile * RS_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ unsigned cnt[RS_WAVES][256];                 // a wave's running count of every digit
  __shared__ unsigned stage[SCATTER ? RS_TILE : 1];       // the tile's keys in digit order
  __shared__ unsigned dstart[SCATTER ? 256 : 1];          // first staged position of a digit
  __shared__ long long gdst[SCATTER ? 256 : 1];           // global position of staged position 0 of a digit's run
  __shared__ unsigned wsum[4];
  for (int k = threadIdx.x; k < RS_WAVES * 256; k += RS_THREADS) (&cnt[0][0])[k] = 0u;
  __syncthreads();
  unsigned key[RS_ROUNDS];
  unsigned short rank[RS_ROUNDS];
  const long long w0 = i0 + (long long)w * (RS_ROUNDS * WAVE);
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const long long i = w0 + r * WAVE + lane;
    key[r] = i < n ? src[i] : 0xFFFFFFFFu;
  }
  unsigned* mycnt = cnt[w];
  if (!SCATTER) {
    // counts only: a lane adds a RUN of equal digits among its own 16 keys with one LDS atomic (no return value, nothing
    // waits for it).  Mantissa digits (about uniform): a run is one key and the 64 lanes of an atomic rarely meet in a word;
    // the top digit (exponent bits: a handful of values hold every key): a lane's 16 keys are one or two runs, so the
    // same-address serialisation that 16 atomics per lane would pay is paid once.
    unsigned cd = (key[0] >> shift) & 255u;
    unsigned cc = w0 + lane < n ? 1u : 0u;
#pragma unroll
    for (int r = 1; r < RS_ROUNDS; ++r) {
      const unsigned d = (key[r] >> shift) & 255u;
      if (w0 + r * WAVE + lane < n) {
        if (d == cd) ++cc;
        else {
          if (cc) atomicAdd(&mycnt[cd], cc);
          cd = d;
          cc = 1u;
        }
      }
    }
    if (cc) atomicAdd(&mycnt[cd], cc);
  } else {
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const bool valid = w0 + r * WAVE + lane < n;
      const unsigned d = (key[r] >> shift) & 255u;
      // peers = the valid lanes whose digit equals mine: per digit bit one ballot, and m &= ~(ballot ^ -bit) as ONE
      // v_bitop3_b32 per mask half (truth table 0x90: a & ~(b ^ c))
      const unsigned long long vb = __ballot(valid);
      unsigned mlo = (unsigned)vb, mhi = (unsigned)(vb >> 32);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int sb = ((int)(d << (31 - b))) >> 31;   // -bit
        const unsigned long long bb = __ballot(sb != 0);
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (unsigned)bb, (unsigned)sb, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (unsigned)(bb >> 32), (unsigned)sb, 0x90);
      }
      const unsigned below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));   // peers in lower lanes
      const unsigned peers = (unsigned)__builtin_popcount(mlo) + (unsigned)__builtin_popcount(mhi);
      // the wave's running count of the digit: read by every peer, rewritten by the first one.  LDS operations of one
      // wave complete in issue order, so the next round's read sees this write; relaxed atomics keep the compiler from
      // holding the word in a register across rounds.
      const unsigned prior = __hip_atomic_load(&mycnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      rank[r] = (unsigned short)(prior + below);
      if (valid && below == 0) __hip_atomic_store(&mycnt[d], prior + peers, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  __syncthreads();
  const int d = threadIdx.x & 255;   // threads 0 .. 255 <-> the 256 digits (the other threads mirror them and do not store)
  const bool owner = threadIdx.x < 256;
  unsigned c[RS_WAVES], tot = 0;
#pragma unroll
  for (int ww = 0; ww < RS_WAVES; ++ww) {
    c[ww] = cnt[ww][d];
    tot += c[ww];
  }
  unsigned* h = hist + ((size_t)seg * T + tile) * 256;
  if (!SCATTER) {
    if (owner) h[d] = tot;
    return;
  }
  // exclusive scan of the tile's digit totals over the 256 digits (wave scan + the waves' sums)
  unsigned incl = tot;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, WAVE);
    if (lane >= off) incl += o;
  }
  if (lane == WAVE - 1 && owner) wsum[w] = incl;
  __syncthreads();
  unsigned before = 0;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) before += ww < (w & 3) ? wsum[ww] : 0u;
  const unsigned ds = before + incl - tot;
  if (owner) {
    dstart[d] = ds;
    gdst[d] = (long long)base[(size_t)seg * 256 + d] + (long long)h[d] - (long long)ds;
  }
  if (owner) {
    unsigned run = 0;
#pragma unroll
    for (int ww = 0; ww < RS_WAVES; ++ww) {   // a wave's keys of digit d follow those of the waves before it
      cnt[ww][d] = run;
      run += c[ww];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const bool valid = w0 + r * WAVE + lane < n;
    const unsigned dd = (key[r] >> shift) & 255u;
    if (valid) stage[dstart[dd] + cnt[w][dd] + rank[r]] = key[r];
  }
  __syncthreads();
  const int count = (int)min((long long)RS_TILE, n - i0);
  unsigned* dst = out + (long long)seg * n;
  for (int k = threadIdx.x; k < count; k += RS_THREADS) {
    const unsigned kk = stage[k];
    dst[gdst[(kk >> shift) & 255u] + k] = kk;
  }
}

// per label: hist[t][d] := number of keys of digit d in the tiles before t (in place); base[d] := keys of smaller digits
__global__ __launch_bounds__(256) void k_rs_scan(int T, unsigned* __restrict__ hist, unsigned* __restrict__ base) {
  const int seg = blockIdx.x, d = threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __shared__ unsigned wsum[4];
  unsigned* h = hist + (size_t)seg * T * 256;
  unsigned run = 0;
  int t = 0;
  for (; t + 8 <= T; t += 8) {   // eight independent loads in flight
    unsigned cc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cc[u] = h[(size_t)(t + u) * 256 + d];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h[(size_t)(t + u) * 256 + d] = run;
      run += cc[u];
    }
  }
  for (; t < T; ++t) {
    const unsigned cc = h[(size_t)t * 256 + d];
    h[(size_t)t * 256 + d] = run;
    run += cc;
  }
  unsigned incl = run;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const unsigned o = __shfl_up(incl, off, WAVE);
    if (lane >= off) incl += o;
  }
  if (lane == WAVE - 1) wsum[w] = incl;
  __syncthreads();
  unsigned before = 0;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) before += ww < w ? wsum[ww] : 0u;
  base[(size_t)seg * 256 + d] = before + incl - run;
}

static inline int rs_tiles(long long n) { long long k = (n + RS_TILE - 1) / RS_TILE; return k < 1 ? 1 : (int)k; }

static inline int metric_chunks(long long n) { long long k = (n + METRIC_CHUNK - 1) / METRIC_CHUNK; return k < 1 ? 1 : (int)k; }

static inline int label_bits(int C) {
  int b = 1;
  while ((1 << b) < C) ++b;
  return b;
}
static size_t sort_temp_bytes(long long n, int C) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                  (size_t)(n * C), 1u, (unsigned)(33 + label_bits(C)));
  return bytes;
}
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

size_t cgcn_metrics_workspace_bytes(long long n, int C) {
  if (n < 0 || C < 1 || (double)n * C >= 2147483647.0) return 0;
  const size_t items = (size_t)n * C;
  const size_t nrec = (size_t)C * metric_chunks(n);
  // two 64-bit key buffers (the first is reused for the unpacked score / target arrays after the sort)
  const size_t general = 2 * al(items * 8) + al(nrec * sizeof(ChunkRec)) + al(nrec * sizeof(ChunkOut)) + al((size_t)C * 8) +
                         al(sort_temp_bytes(n, C)) + 256;
  // cgcn_multilabel_metrics_nonneg: two 32-bit key buffers, its tile histograms and digit bases (more than the above only
  // on tiny inputs)
  const size_t nonneg = 2 * al(items * 4) + al((size_t)C * rs_tiles(n) * 256 * 4) + al((size_t)C * 256 * 4) +
                        al(nrec * sizeof(ChunkRec)) + al(nrec * sizeof(ChunkOut)) + al((size_t)C * 8) + 256;
  return general > nonneg ? general : nonneg;
}

int cgcn_multilabel_metrics_nonneg(cgcn_stream_t stream, long long n, int C, const float* probs, const float* targets,
                                   float fdr_cutoff, float* out, int32_t* bad, void* workspace, size_t workspace_bytes) {
  if (n < 0 || C < 1 || !out || !bad) return CGCN_ERR_BAD_ARG;
  if ((double)n * C >= 2147483647.0) return CGCN_ERR_UNSUPPORTED;
  if (n > 0 && (!probs || !targets || !workspace)) return CGCN_ERR_BAD_ARG;
  if (workspace_bytes < cgcn_metrics_workspace_bytes(n, C)) return CGCN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t items = (size_t)n * C;
  const int T = rs_tiles(n), nch = metric_chunks(n);
  char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  unsigned* k_a = (unsigned*)w; w += al(items * 4);
  unsigned* k_b = (unsigned*)w; w += al(items * 4);
  unsigned* hist = (unsigned*)w; w += al((size_t)C * T * 256 * 4);
  unsigned* base = (unsigned*)w; w += al((size_t)C * 256 * 4);
  ChunkRec* rec = (ChunkRec*)w; w += al((size_t)C * nch * sizeof(ChunkRec));
  ChunkOut* outp = (ChunkOut*)w; w += al((size_t)C * nch * sizeof(ChunkOut));
  double* Ptot = (double*)w; w += al((size_t)C * 8);
  if ((size_t)(w - (char*)workspace) > workspace_bytes) return CGCN_ERR_WORKSPACE;   // (cannot happen: the size above covers this layout)
  if (hipMemsetAsync(bad, 0, sizeof(int), st) != hipSuccess) return CGCN_ERR_LAUNCH;
  if (n > 0) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((C + 31) / 32));
    // rows per block of the flat pack: a power of two, [C][R + 1] keys within 48 KB of LDS
    int R = 128, rshift = 7;
    while (R > 4 && (size_t)C * (R + 1) * 4 > 48 * 1024) { R >>= 1; --rshift; }
    if ((size_t)C * (R + 1) * 4 <= 48 * 1024) {
      const unsigned blocks = (unsigned)((n + R - 1) / R);
      const size_t lds = (size_t)C * (R + 1) * 4;
      if ((((uintptr_t)probs | (uintptr_t)targets) & 15) == 0)
        hipLaunchKernelGGL(k_metrics_pack32_flat<4>, dim3(blocks), dim3(256), lds, st, n, C, R, rshift, probs, targets, k_a, bad);
      else
        hipLaunchKernelGGL(k_metrics_pack32_flat<1>, dim3(blocks), dim3(256), lds, st, n, C, R, rshift, probs, targets, k_a, bad);
    } else   // thousands of labels: the tile transposition
      hipLaunchKernelGGL(k_metrics_pack32, grid, dim3(256), 0, st, n, C, probs, targets, k_a, bad);
    unsigned* a = k_a;
    unsigned* b = k_b;
    for (int shift = 1; shift < 32; shift += 8) {   // key bits 1 .. 31: four passes, the sorted keys end up in k_a again
      hipLaunchKernelGGL(k_rs_pass<false>, dim3(T, C), dim3(RS_THREADS), 0, st, n, T, shift, (const unsigned*)a, b, hist, (const unsigned*)base);
      hipLaunchKernelGGL(k_rs_scan, dim3(C), dim3(256), 0, st, T, hist, base);
      hipLaunchKernelGGL(k_rs_pass<true>, dim3(T, C), dim3(RS_THREADS), 0, st, n, T, shift, (const unsigned*)a, b, hist, (const unsigned*)base);
      unsigned* tsw = a; a = b; b = tsw;
    }
  }
  hipLaunchKernelGGL(k_metrics_summary<true>, dim3(nch, C), dim3(64), 0, st, n, nch, (const void*)k_a, (const unsigned char*)nullptr, rec);
  hipLaunchKernelGGL(k_metrics_prefix, dim3(C), dim3(64), 0, st, nch, rec, Ptot);
  hipLaunchKernelGGL(k_metrics_chunks<true>, dim3(nch, C), dim3(64), 0, st, n, nch, (const void*)k_a, (const unsigned char*)nullptr, rec, Ptot, (double)fdr_cutoff, outp);
  hipLaunchKernelGGL(k_metrics_final, dim3(C), dim3(64), 0, st, n, C, nch, outp, Ptot, out);
  return launch_status();
}

int cgcn_multilabel_metrics(cgcn_stream_t stream, long long n, int C, const float* probs, const float* targets,
                            float fdr_cutoff, float* out, void* workspace, size_t workspace_bytes) {
  if (n < 0 || C < 1 || !out) return CGCN_ERR_BAD_ARG;
  if ((double)n * C >= 2147483647.0) return CGCN_ERR_UNSUPPORTED;
  if (n > 0 && (!probs || !targets || !workspace)) return CGCN_ERR_BAD_ARG;
  if (workspace_bytes < cgcn_metrics_workspace_bytes(n, C)) return CGCN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const size_t items = (size_t)n * C;
  char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  unsigned long long* k_in = (unsigned long long*)w; w += al(items * 8);
  unsigned long long* k_out = (unsigned long long*)w; w += al(items * 8);
  float* keys_out = (float*)k_in;                                  // 4 bytes per item of the 8 ...
  unsigned char* vals_out = (unsigned char*)k_in + ((items * 4 + 15) & ~(size_t)15);  // ... and 1 more: fits in the dead input buffer
  const int nch = metric_chunks(n);
  ChunkRec* rec = (ChunkRec*)w; w += al((size_t)C * nch * sizeof(ChunkRec));
  ChunkOut* outp = (ChunkOut*)w; w += al((size_t)C * nch * sizeof(ChunkOut));
  double* Ptot = (double*)w; w += al((size_t)C * 8);
  size_t temp = sort_temp_bytes(n, C);
  if (n > 0) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_metrics_pack, grid, dim3(256), 0, st, n, C, probs, targets, k_in);
    // bit 0 (the target) does not take part: ties in score are one curve point whatever their order
    if (rocprim::radix_sort_keys((void*)w, temp, (const unsigned long long*)k_in, k_out, items, 1u, (unsigned)(33 + label_bits(C)), st) != hipSuccess)
      return CGCN_ERR_LAUNCH;
    hipLaunchKernelGGL(k_metrics_unpack, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (long long)items, k_out, keys_out, vals_out);
  }
  hipLaunchKernelGGL(k_metrics_summary<false>, dim3(nch, C), dim3(64), 0, st, n, nch, (const void*)keys_out, (const unsigned char*)vals_out, rec);
  hipLaunchKernelGGL(k_metrics_prefix, dim3(C), dim3(64), 0, st, nch, rec, Ptot);
  hipLaunchKernelGGL(k_metrics_chunks<false>, dim3(nch, C), dim3(64), 0, st, n, nch, (const void*)keys_out, (const unsigned char*)vals_out, rec, Ptot, (double)fdr_cutoff, outp);
  hipLaunchKernelGGL(k_metrics_final, dim3(C), dim3(64), 0, st, n, C, nch, outp, Ptot, out);
  return launch_status();
}

}  // extern "C"
