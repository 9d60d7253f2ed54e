// chromegcn_amd/csrc/cgcn_graph.hip
//
// Device-side adjacency normaliser: the work the reference does on the CPU
// with SciPy for every chromosome of every epoch -- process_graph, utils/util_methods.py:146-180 -- as
// three small kernels that run once per chromosome:
//     count  : per-row nnz of A-hat                       (one thread per row, sorted merge)
//     scan   : exclusive prefix sum -> rowptr              (one workgroup)
//     fill   : columns (+ values for 'both'), 1/rowsum     (same merge as count)
//     symm   : is A-hat == A-hat^T ? (decides whether the backward can reuse the same CSR)
// Input: canonical CSR (sorted columns, no duplicates) of the raw Hi-C matrix, fp32 values or NULL = ones.
//
// A-hat per adj_type (utils/util_methods.py:148-174):
//   CGCN_ADJ_HIC      binarise(hic + I): entry kept iff hic_ij + [i==j] > 0, value 1      (:152-165)
//   CGCN_ADJ_CONSTANT band(+-7) + I                                                        (:148-150)
//   CGCN_ADJ_BOTH     hic + band(+-7) + I, NOT binarised; exact zeros dropped              (:168-171)
//   CGCN_ADJ_NONE     I                                                                    (:173-174)
// row_scale = 1 / rowsum(A-hat) computed in double and rounded to fp32 once, inf -> 0      (:99-106,:122)
#include "cgcn_common.hpp"

#define BAND_RADIUS 7  // utils/util_methods.py:147

// Enumerate the merged, column-sorted entries (c, v) of row i of  hic*[uses hic] + band*[uses band] + I.
template <class F>
__device__ __forceinline__ void merged_row(int i, int n, int adj_type, const int* __restrict__ rowptr,
                                           const int* __restrict__ col, const float* __restrict__ val, F emit) {
  const bool use_hic = adj_type == CGCN_ADJ_HIC || adj_type == CGCN_ADJ_BOTH;
  const bool use_band = adj_type == CGCN_ADJ_CONSTANT || adj_type == CGCN_ADJ_BOTH;
  int p = 0, p1 = 0;
  if (use_hic) { p = rowptr[i]; p1 = rowptr[i + 1]; }
  int b = use_band ? max(0, i - BAND_RADIUS) : i;
  const int b1 = use_band ? min(n - 1, i + BAND_RADIUS) : i;
  while (p < p1 || b <= b1) {
    const int ch = p < p1 ? col[p] : 0x7fffffff;
    const int cb = b <= b1 ? b : 0x7fffffff;
    const int c = min(ch, cb);
    double v = 0.0;
    if (ch == c) { v += val ? (double)val[p] : 1.0; ++p; }
    if (cb == c) { v += 1.0; ++b; }  // band entry (c != i) or the identity (c == i): both contribute 1
    emit(c, v);
  }
}

__global__ __launch_bounds__(256) void k_graph_count(int n, int adj_type, const int* __restrict__ rowptr,
                                                     const int* __restrict__ col, const float* __restrict__ val,
                                                     int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cnt = 0;
  const bool binarise = adj_type == CGCN_ADJ_HIC;
  merged_row(i, n, adj_type, rowptr, col, val, [&](int, double v) { cnt += binarise ? (v > 0.0) : (v != 0.0); });
  counts[i] = cnt;
}

// exclusive scan of counts[0..n) into out[0..n], out[n] = total.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void k_graph_scan(int n, const int* __restrict__ counts, int* __restrict__ out) {
  __shared__ long long part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int i0 = min(n, t * per), i1 = min(n, i0 + per);
  long long s = 0;
  for (int i = i0; i < i1; ++i) s += counts[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    long long v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  long long run = t ? part[t - 1] : 0;
  for (int i = i0; i < i1; ++i) {
    out[i] = (int)run;
    run += counts[i];
  }
  if (t == 1023) out[n] = (int)part[1023];
}

__global__ __launch_bounds__(256) void k_graph_fill(int n, int adj_type, const int* __restrict__ rowptr,
                                                    const int* __restrict__ col, const float* __restrict__ val,
                                                    const int* __restrict__ rowptr_out, int* __restrict__ col_out,
                                                    float* __restrict__ val_out, float* __restrict__ row_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int w = rowptr_out[i];
  double sum = 0.0;
  const bool binarise = adj_type == CGCN_ADJ_HIC;
  merged_row(i, n, adj_type, rowptr, col, val, [&](int c, double v) {
    if (binarise ? (v > 0.0) : (v != 0.0)) {
      col_out[w] = c;
      if (val_out) val_out[w] = binarise ? 1.f : (float)v;
      sum += binarise ? 1.0 : v;
      ++w;
    }
  });
  row_scale[i] = sum != 0.0 ? (float)(1.0 / sum) : 0.f;  // np.power(rowsum,-1), inf -> 0 (:101-103)
}

// flag[0] &= (every stored (i,j,v) has a stored (j,i,v)); columns sorted, val NULL = ones.
__global__ __launch_bounds__(256) void k_graph_symmetric(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const float* __restrict__ val, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool ok = true;
  for (int p = rowptr[i]; p < rowptr[i + 1] && ok; ++p) {
    const int j = col[p];
    int lo = rowptr[j], hi = rowptr[j + 1] - 1, found = -1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const int cm = col[mid];
      if (cm == i) { found = mid; break; }
      if (cm < i) lo = mid + 1; else hi = mid - 1;
    }
    if (found < 0 || (val && val[found] != val[p])) ok = false;
  }
  if (!ok) atomicAnd(flag, 0);
}

extern "C" {

int cgcn_graph_count(cgcn_stream_t stream, int n, int adj_type, const int32_t* rowptr_in, const int32_t* col_in,
                     const float* val_in, int32_t* row_counts, int32_t* rowptr_out) {
  if (n < 0 || adj_type < CGCN_ADJ_HIC || adj_type > CGCN_ADJ_NONE) return CGCN_ERR_BAD_ARG;
  if (!row_counts || !rowptr_out) return CGCN_ERR_BAD_ARG;
  const bool use_hic = adj_type == CGCN_ADJ_HIC || adj_type == CGCN_ADJ_BOTH;
  if (use_hic && n > 0 && !rowptr_in) return CGCN_ERR_BAD_ARG;  // col_in may be NULL for an empty matrix
  hipStream_t st = (hipStream_t)stream;
  if (n > 0) hipLaunchKernelGGL(k_graph_count, dim3((n + 255) / 256), dim3(256), 0, st, n, adj_type, rowptr_in, col_in, val_in, row_counts);
  hipLaunchKernelGGL(k_graph_scan, dim3(1), dim3(1024), 0, st, n, row_counts, rowptr_out);
  return launch_status();
}

int cgcn_graph_fill(cgcn_stream_t stream, int n, int adj_type, const int32_t* rowptr_in, const int32_t* col_in,
                    const float* val_in, const int32_t* rowptr_out, int32_t* col_out, float* val_out, float* row_scale,
                    int32_t* symmetric_flag) {
  if (n < 0 || adj_type < CGCN_ADJ_HIC || adj_type > CGCN_ADJ_NONE) return CGCN_ERR_BAD_ARG;
  if (n == 0) return CGCN_OK;
  if (!rowptr_out || !col_out || !row_scale) return CGCN_ERR_BAD_ARG;
  if (adj_type == CGCN_ADJ_BOTH && !val_out) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_graph_fill, dim3((n + 255) / 256), dim3(256), 0, st, n, adj_type, rowptr_in, col_in, val_in, rowptr_out,
                     col_out, val_out, row_scale);
  if (symmetric_flag)  // caller initialises it to 1
    hipLaunchKernelGGL(k_graph_symmetric, dim3((n + 255) / 256), dim3(256), 0, st, n, rowptr_out, col_out,
                       adj_type == CGCN_ADJ_BOTH ? val_out : nullptr, symmetric_flag);
  return launch_status();
}

}  // extern "C"
