// chromegcn_amd/csrc/cgcn_metrics.hip
//
// Multi-label ranking metrics on the device.  The reference computes them on
// the CPU with one scikit-learn call per label per metric after every split (runner.py:41,45,51 ->
// utils/evals.py:89-92 -> utils/metrics.py:148-183,238-253): AUROC, area under the precision-recall curve
// (trapezoid over sklearn's precision_recall_curve points), recall at the first point with FDR <= cutoff,
// and average precision (mAP).  Here: ONE device-wide radix sort (rocprim::radix_sort_keys, called directly) of 64-bit keys
// (label | descending score | target) -- every label's list comes out contiguous and sorted and the whole chip
// works on it (a segmented sort with one long segment per label used a fraction of the chip) -- and a chunked scan
// of the sorted lists (one wave per 4096-element chunk),
// treating tied scores as one curve point exactly like sklearn's distinct-threshold curves.  All curve
// arithmetic is fp64 and summed in a fixed order.
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

__device__ __forceinline__ double wave_incl_scan_d(double v, int lane) {
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const double o = __shfl_up(v, off, WAVE);
    if (lane >= off) v += o;
  }
  return v;
}

__global__ __launch_bounds__(64) void k_metrics_summary(long long n, int nch, const float* __restrict__ keys,
                                                        const unsigned char* __restrict__ vals, ChunkRec* __restrict__ rec) {
  const int c = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x;
  const float* k = keys + (long long)c * n;
  const unsigned char* v = vals + (long long)c * n;
  const long long i0 = (long long)ch * METRIC_CHUNK, i1 = min(n, i0 + METRIC_CHUNK);
  double carry = 0.0, end_tp = 0.0;
  long long end_idx = 0;
  for (long long base = i0; base < i1; base += WAVE) {
    const long long i = base + lane;
    const bool ok = i < i1;
    const float sc = ok ? k[i] : 0.f;
    const float nx = (i + 1 < n) ? k[i + 1] : 0.f;
    const double t = ok ? (double)v[i] : 0.0;
    const double tp = carry + wave_incl_scan_d(t, lane);
    const bool end = ok && ((i + 1 == n) || sc != nx);
    const unsigned long long bal = __ballot(end);
    if (bal) {
      const int hi = 63 - __builtin_clzll(bal);
      end_tp = __shfl(tp, hi, WAVE);
      end_idx = base + hi + 1;
    }
    carry = __shfl(tp, WAVE - 1, WAVE);
  }
  if (lane == 0) {
    ChunkRec& r = rec[(size_t)c * nch + ch];
    r.pos = carry;
    r.end_tp = end_tp;
    r.end_idx = end_idx;
  }
}

__global__ __launch_bounds__(64) void k_metrics_prefix(int nch, ChunkRec* __restrict__ rec, double* __restrict__ Ptot) {
  const int c = blockIdx.x;
  if (threadIdx.x != 0) return;
  ChunkRec* r = rec + (size_t)c * nch;
  double carry = 0.0, ptp = 0.0, pfp = 0.0;
  for (int k = 0; k < nch; ++k) {
    r[k].carry_tp = carry;
    r[k].prev_tp = ptp;
    r[k].prev_fp = pfp;
    if (r[k].end_idx > 0) {
      ptp = carry + r[k].end_tp;
      pfp = (double)r[k].end_idx - ptp;
    }
    carry += r[k].pos;
  }
  Ptot[c] = carry;
}

__global__ __launch_bounds__(64) void k_metrics_chunks(long long n, int nch, const float* __restrict__ keys,
                                                       const unsigned char* __restrict__ vals,
                                                       const ChunkRec* __restrict__ rec, const double* __restrict__ Ptot,
                                                       double fdr_cutoff, ChunkOut* __restrict__ outp) {
  const int c = blockIdx.y, ch = blockIdx.x, lane = threadIdx.x;
  const float* k = keys + (long long)c * n;
  const unsigned char* v = vals + (long long)c * n;
  const long long i0 = (long long)ch * METRIC_CHUNK, i1 = min(n, i0 + METRIC_CHUNK);
  const ChunkRec r = rec[(size_t)c * nch + ch];
  const double P = Ptot[c];
  double carry_tp = r.carry_tp;
  double prev_tp = r.prev_tp, prev_fp = r.prev_fp;
  double s_auc = 0.0, s_aupr = 0.0, s_ap = 0.0, r_fdr = 0.0;
  int has_fdr = 0;
  for (long long base = i0; base < i1; base += WAVE) {
    const long long i = base + lane;
    const bool ok = i < i1;
    const float sc = ok ? k[i] : 0.f;
    const float nx = (i + 1 < n) ? k[i + 1] : 0.f;
    const double t = ok ? (double)v[i] : 0.0;
    const double tp = carry_tp + wave_incl_scan_d(t, lane);
    const double fp = (double)(i + 1) - tp;
    const bool end = ok && ((i + 1 == n) || sc != nx);   // last element of a run of tied scores
    // lane of the previous run end inside this wave-chunk (-1: it is the carried point)
    int pe = end ? lane : -1;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const int o = __shfl_up(pe, off, WAVE);
      if (lane >= off) pe = max(pe, o);
    }
    int pprev = __shfl_up(pe, 1, WAVE);
    if (lane == 0) pprev = -1;
    const int src = max(pprev, 0);
    const double ptp_l = __shfl(tp, src, WAVE), pfp_l = __shfl(fp, src, WAVE);
    const double ptp = pprev < 0 ? prev_tp : ptp_l;
    const double pfp = pprev < 0 ? prev_fp : pfp_l;
    double d_auc = 0.0, d_aupr = 0.0, d_ap = 0.0;
    double prec = 1.0, rec_ = 0.0;
    if (end) {
      prec = tp / (tp + fp);
      rec_ = P > 0.0 ? tp / P : 1.0;
      const double pp = (ptp + pfp) > 0.0 ? ptp / (ptp + pfp) : 1.0;              // origin: precision 1
      const double pr = (ptp + pfp) > 0.0 ? (P > 0.0 ? ptp / P : 1.0) : 0.0;      // origin: recall 0
      d_auc = (fp - pfp) * (tp + ptp) * 0.5;
      d_aupr = (rec_ - pr) * (prec + pp) * 0.5;
      d_ap = (rec_ - pr) * prec;
    }
    s_auc += d_auc;    // per-lane partial sums, folded across the wave ONCE after the loop (fixed order: deterministic);
    s_aupr += d_aupr;  // three fp64 wave reductions per 64 elements were a third of this kernel's instructions
    s_ap += d_ap;
    const bool q = end && (1.0 - prec) <= fdr_cutoff;
    const unsigned long long bal = __ballot(q);
    if (bal) {
      const int hi = 63 - __builtin_clzll(bal);
      r_fdr = __shfl(rec_, hi, WAVE);
      has_fdr = 1;
    }
    const int last_end = __shfl(pe, WAVE - 1, WAVE);
    if (last_end >= 0) {
      prev_tp = __shfl(tp, last_end, WAVE);
      prev_fp = __shfl(fp, last_end, WAVE);
    }
    carry_tp = __shfl(tp, WAVE - 1, WAVE);
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
  const int c = blockIdx.x;
  if (threadIdx.x != 0) return;
  const ChunkOut* o = outp + (size_t)c * nch;
  double s_auc = 0.0, s_aupr = 0.0, s_ap = 0.0, r_fdr = 0.0;
  for (int k = 0; k < nch; ++k) {
    s_auc += o[k].auc; s_aupr += o[k].aupr; s_ap += o[k].ap;
    if (o[k].has_fdr) r_fdr = o[k].rfdr;   // the deepest chunk with a qualifying point wins
  }
  const double P = Ptot[c], N = (double)n - P;
  const float nanv = __int_as_float(0x7fc00000);
  out[0 * C + c] = (P > 0.0 && N > 0.0) ? (float)(s_auc / (P * N)) : nanv;  // undefined with one class present
  out[1 * C + c] = n > 0 ? (float)s_aupr : nanv;
  out[2 * C + c] = n > 0 ? (float)r_fdr : nanv;
  out[3 * C + c] = n > 0 ? (float)s_ap : nanv;  // 0 when the label has no positive (as sklearn)
}

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
  return 2 * al(items * 8) + al(nrec * sizeof(ChunkRec)) + al(nrec * sizeof(ChunkOut)) + al((size_t)C * 8) +
         al(sort_temp_bytes(n, C)) + 256;
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
  hipLaunchKernelGGL(k_metrics_summary, dim3(nch, C), dim3(64), 0, st, n, nch, keys_out, vals_out, rec);
  hipLaunchKernelGGL(k_metrics_prefix, dim3(C), dim3(64), 0, st, nch, rec, Ptot);
  hipLaunchKernelGGL(k_metrics_chunks, dim3(nch, C), dim3(64), 0, st, n, nch, keys_out, vals_out, rec, Ptot, (double)fdr_cutoff, outp);
  hipLaunchKernelGGL(k_metrics_final, dim3(C), dim3(64), 0, st, n, C, nch, outp, Ptot, out);
  return launch_status();
}

}  // extern "C"
