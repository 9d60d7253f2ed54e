// chromegcn_amd/csrc/cgcn_metrics.hip
//
// Multi-label ranking metrics on the device.  The reference computes them on
// the CPU with one scikit-learn call per label per metric after every split (runner.py:41,45,51 ->
// utils/evals.py:89-92 -> utils/metrics.py:148-183,238-253): AUROC, area under the precision-recall curve
// (trapezoid over sklearn's precision_recall_curve points), recall at the first point with FDR <= cutoff,
// and average precision (mAP).  Here: one segmented radix sort of all (score, label) pairs by descending
// score (rocPRIM through hipCUB) and one wave per label that walks the sorted list once, treating tied scores
// as one curve point exactly like sklearn's distinct-threshold curves.  All curve arithmetic is fp64.
#include <hipcub/hipcub.hpp>

#include "cgcn_common.hpp"

// [n,C] row-major -> per-label contiguous keys[c][i] (score) and vals[c][i] (1 = positive)
__global__ __launch_bounds__(256) void k_metrics_pack(long long n, int C, const float* __restrict__ probs,
                                                      const float* __restrict__ targets, float* __restrict__ keys,
                                                      unsigned char* __restrict__ vals) {
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
    if (c < C && i < n) {
      keys[(long long)c * n + i] = tp[tx][r];
      vals[(long long)c * n + i] = tt[tx][r] > 0.5f ? 1 : 0;
    }
  }
}

__global__ void k_metrics_offsets(long long n, int C, int* __restrict__ off) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c <= C) off[c] = (int)((long long)c * n);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// One wave per label over its descending-sorted (score, positive) list.
// Curve points are the ends of runs of equal scores: (tp_k, fp_k), k = 1..K, plus the origin.
//   AUROC   = sum (fp_k - fp_{k-1}) (tp_k + tp_{k-1}) / 2 / (P N)                       roc_auc_score
//   AUPR    = sum (r_k - r_{k-1}) (p_k + p_{k-1}) / 2, (r_0, p_0) = (0, 1)              auc(recall, precision)
//   AP      = sum (r_k - r_{k-1}) p_k                                                   average_precision_score
//   R@FDR   = r_k of the LAST point with 1 - p_k <= cutoff, 0 if none                   utils/metrics.py:148-166
// with p_k = tp_k / (tp_k + fp_k), r_k = tp_k / P (sklearn: r_k = 1 when P = 0).  out[m*C + c], m = 0..3.
__global__ __launch_bounds__(64) void k_metrics_scan(long long n, int C, const float* __restrict__ keys,
                                                     const unsigned char* __restrict__ vals, double fdr_cutoff,
                                                     float* __restrict__ out) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  const float* k = keys + (long long)c * n;
  const unsigned char* v = vals + (long long)c * n;
  // pass 0: number of positives
  double P = 0.0;
  for (long long i = lane; i < n; i += WAVE) P += v[i];
  P = wave_sum_d(P);
  const double N = (double)n - P;

  double carry_tp = 0.0;                       // positives before this chunk
  double prev_tp = 0.0, prev_fp = 0.0;         // previous curve point (origin at start)
  double s_auc = 0.0, s_aupr = 0.0, s_ap = 0.0, r_fdr = 0.0;
  for (long long base = 0; base < n; base += WAVE) {
    const long long i = base + lane;
    const bool ok = i < n;
    const float sc = ok ? k[i] : 0.f;
    const float nx = (i + 1 < n) ? k[i + 1] : 0.f;
    const double t = ok ? (double)v[i] : 0.0;
    // inclusive scan of positives over the wave
    double cum = t;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const double o = __shfl_up(cum, off, WAVE);
      if (lane >= off) cum += o;
    }
    const double tp = carry_tp + cum;
    const double fp = (double)(i + 1) - tp;
    const bool end = ok && ((i + 1 == n) || sc != nx);   // last element of a run of tied scores
    // index of the previous run end inside this chunk (-1: it is the carried point)
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
    double prec = 1.0, rec = 0.0;
    if (end) {
      prec = tp / (tp + fp);
      rec = P > 0.0 ? tp / P : 1.0;
      const double pp = (ptp + pfp) > 0.0 ? ptp / (ptp + pfp) : 1.0;     // origin: precision 1
      const double pr = (ptp + pfp) > 0.0 ? (P > 0.0 ? ptp / P : 1.0) : 0.0;  // origin: recall 0
      d_auc = (fp - pfp) * (tp + ptp) * 0.5;
      d_aupr = (rec - pr) * (prec + pp) * 0.5;
      d_ap = (rec - pr) * prec;
    }
    s_auc += wave_sum_d(d_auc);
    s_aupr += wave_sum_d(d_aupr);
    s_ap += wave_sum_d(d_ap);
    // last point (deepest in the list) with FDR <= cutoff: keep the one with the largest index
    const bool q = end && (1.0 - prec) <= fdr_cutoff;
    const unsigned long long bal = __ballot(q);
    if (bal) {
      const int hi = 63 - __builtin_clzll(bal);
      r_fdr = __shfl(rec, hi, WAVE);
    }
    // carry to the next chunk: totals and the last run end seen so far
    const int last_end = __shfl(pe, WAVE - 1, WAVE);
    if (last_end >= 0) {
      prev_tp = __shfl(tp, last_end, WAVE);
      prev_fp = __shfl(fp, last_end, WAVE);
    }
    carry_tp = __shfl(tp, WAVE - 1, WAVE);
  }
  if (lane == 0) {
    const float nanv = __int_as_float(0x7fc00000);
    out[0 * C + c] = (P > 0.0 && N > 0.0) ? (float)(s_auc / (P * N)) : nanv;  // undefined with one class present
    out[1 * C + c] = n > 0 ? (float)s_aupr : nanv;
    out[2 * C + c] = n > 0 ? (float)r_fdr : nanv;
    out[3 * C + c] = n > 0 ? (float)s_ap : nanv;  // 0 when the label has no positive (as sklearn)
  }
}

static size_t sort_temp_bytes(long long n, int C) {
  size_t bytes = 0;
  hipcub::DeviceSegmentedRadixSort::SortPairsDescending(nullptr, bytes, (const float*)nullptr, (float*)nullptr,
                                                        (const unsigned char*)nullptr, (unsigned char*)nullptr,
                                                        (int)(n * C), C, (const int*)nullptr, (const int*)nullptr);
  return bytes;
}
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" {

size_t cgcn_metrics_workspace_bytes(long long n, int C) {
  if (n < 0 || C < 1 || (double)n * C >= 2147483647.0) return 0;
  const size_t items = (size_t)n * C;
  return 2 * al(items * 4) + 2 * al(items) + al((size_t)(C + 1) * 4) + al(sort_temp_bytes(n, C)) + 256;
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
  float* keys_in = (float*)w; w += al(items * 4);
  float* keys_out = (float*)w; w += al(items * 4);
  unsigned char* vals_in = (unsigned char*)w; w += al(items);
  unsigned char* vals_out = (unsigned char*)w; w += al(items);
  int* off = (int*)w; w += al((size_t)(C + 1) * 4);
  size_t temp = sort_temp_bytes(n, C);
  if (n > 0) {
    dim3 grid((unsigned)((n + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(k_metrics_pack, grid, dim3(256), 0, st, n, C, probs, targets, keys_in, vals_in);
    hipLaunchKernelGGL(k_metrics_offsets, dim3((C + 256) / 256), dim3(256), 0, st, n, C, off);
    if (hipcub::DeviceSegmentedRadixSort::SortPairsDescending(w, temp, keys_in, keys_out, vals_in, vals_out, (int)items, C, off,
                                                              off + 1, 0, 32, st) != hipSuccess)
      return CGCN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(k_metrics_scan, dim3(C), dim3(64), 0, st, n, C, keys_out, vals_out, (double)fdr_cutoff, out);
  return launch_status();
}

}  // extern "C"
