// tools/micro/lds_slice_gather.hip -- experiment (not product): neighbour gather out of an LDS-RESIDENT COLUMN SLICE.
//
// The product's aggregations move nnz * S * d * 4 bytes of 128-byte lines from the XCD L2s into the vector L1s at
// 16-21 TB/s: that rate, not HBM, bounds 42 % of the genome epoch.  LDS delivers 128 B / clk / CU =
// 78 TB/s over the chip.  Here a workgroup owns C consecutive features of one strand for ALL nodes -- the slice
// [n][C] floats: n * 4C bytes, which fits the 160 KB LDS for C = 4 up to n = 9 k, C = 2 up to 19 k, C = 1 up to 38 k, i.e.
// every chromosome -- stages it once (contiguous: the table is laid out slice-major, [S * d / C][n + 1][C], row n = 0),
// and then walks the rows: lane = row, 64 rows of similar degree per block (rows degree-sorted), the block's column
// indices transposed and padded ([K / 4][64 lanes][4 x uint16], pad index = n), so a wave instruction reads 64
// different rows' next neighbour from LDS.  R workgroups share a slice (row blocks dealt round robin over R x 16 waves).
// Cost model: LDS traffic = nnz * S * d * 4 bytes whatever C is; index traffic = (S * d / C) * nnz * 2 bytes from L2;
// staging = R * table bytes.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/lds_slice_gather.hip -o build/lds_slice_gather
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int C> struct vec_of;
template <> struct vec_of<4> { typedef f32x4 t; };
template <> struct vec_of<2> { typedef f32x2 t; };
template <> struct vec_of<1> { typedef float t; };

// baseline of the same arithmetic out of L2: one wave per row, 1 KiB (both strands) per neighbour, row-major table
__global__ __launch_bounds__(512) void k_base(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ rs, const float* __restrict__ X, float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lane_off = ((unsigned)(lane >> 5) * (unsigned)n * 128u + (lane & 31) * 4u) * 4u;
  const char* Xb = (const char*)X;
  const int i = blockIdx.x * 8 + wave;
  if (i >= n) return;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int k0 = rowptr[i], k1 = rowptr[i + 1];
  for (int kb = k0; kb < k1; kb += 64) {
    const int cnt = min(64, k1 - kb);
    const int myc = lane < cnt ? col[kb + lane] : 0;
    for (int b = 0; b < cnt; b += 2) {
      f32x4 t[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)__builtin_amdgcn_readlane(myc, min(b + u, cnt - 1)) * 512u + lane_off);
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (b + u < cnt) acc += t[u];
    }
  }
  *(f32x4*)((char*)H + (size_t)i * 512u + lane_off) = acc * rs[i];
}

// Xs / Hs: [slices][np][C] floats, np = n + 1 rounded up so that a slice is a multiple of 16 bytes; row n of Xs = 0.
// idx: block b's indices at idx + blk_off[b] * 256 (uint16): [K_b / 4][64][4];  blk_k[b] = K_b (multiple of 8).
// rows: [nblk * 64] node of (block, lane), -1 = no row.
template <int C>
__global__ __launch_bounds__(1024) void k_lds_slice(int n, int np, int nblk, int R, const unsigned* __restrict__ blk_off,
                                                    const int* __restrict__ blk_k, const unsigned short* __restrict__ idx,
                                                    const int* __restrict__ rows, const float* __restrict__ rs,
                                                    const float* __restrict__ Xs, float* __restrict__ Hs) {
  typedef typename vec_of<C>::t V;
  extern __shared__ __attribute__((aligned(16))) float tab[];
  const int slices = gridDim.x / R;
  const int slice = blockIdx.x % slices, part = blockIdx.x / slices;
  {
    const f32x4* src = (const f32x4*)(Xs + (size_t)slice * np * C);
    f32x4* dst = (f32x4*)tab;
    const int cnt = np * C / 4;
    for (int i = threadIdx.x; i < cnt; i += 1024) dst[i] = src[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* out = Hs + (size_t)slice * np * C;
  for (int b = part * 16 + wave; b < nblk; b += R * 16) {
    const int K = __builtin_amdgcn_readfirstlane(blk_k[b]);
    const uint2* ip = (const uint2*)(idx + (size_t)blk_off[b] * 256) + lane;   // 4 indices per lane per step
    V acc0 = {}, acc1 = {};
    uint2 a = ip[0], c = ip[64];
    for (int k = 0; k < K; k += 8) {
      const uint2 a_n = ip[(k + 8 < K ? (k / 4 + 2) : 0) * 64], c_n = ip[(k + 8 < K ? (k / 4 + 3) : 0) * 64];
      const V v0 = *(const V*)(tab + (a.x & 0xffffu) * C), v1 = *(const V*)(tab + (a.x >> 16) * C);
      const V v2 = *(const V*)(tab + (a.y & 0xffffu) * C), v3 = *(const V*)(tab + (a.y >> 16) * C);
      const V v4 = *(const V*)(tab + (c.x & 0xffffu) * C), v5 = *(const V*)(tab + (c.x >> 16) * C);
      const V v6 = *(const V*)(tab + (c.y & 0xffffu) * C), v7 = *(const V*)(tab + (c.y >> 16) * C);
      acc0 += v0; acc1 += v1; acc0 += v2; acc1 += v3;
      acc0 += v4; acc1 += v5; acc0 += v6; acc1 += v7;
      a = a_n; c = c_n;
    }
    const int row = rows[b * 64 + lane];
    if (row >= 0) *(V*)(out + (size_t)row * C) = (acc0 + acc1) * rs[row];
  }
}

struct Graph { int n, nnz; std::vector<int> rowptr, col; };

// uniform-random undirected pairs + self loops (the 'hic' graph of process_graph: A + I), like chromegcn_amd/synth.py
static Graph make_graph(int n, int pairs, unsigned seed, bool hic_like) {
  std::mt19937_64 rng(seed);
  std::vector<std::pair<int, int>> e;
  e.reserve((size_t)pairs * 2 + n);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  for (int p = 0; p < pairs; ++p) {
    int i, j;
    if (hic_like) {
      const int dist = std::min(n - 1, std::max(1, (int)std::floor(std::exp(U(rng) * std::log((double)std::max(2, n - 1))))));
      i = (int)(U(rng) * (n - dist));
      j = i + dist;
    } else {
      i = (int)(U(rng) * n); j = (int)(U(rng) * n);
      if (i == j) continue;
    }
    e.emplace_back(i, j); e.emplace_back(j, i);
  }
  for (int i = 0; i < n; ++i) e.emplace_back(i, i);
  std::sort(e.begin(), e.end());
  e.erase(std::unique(e.begin(), e.end()), e.end());
  Graph g; g.n = n; g.nnz = (int)e.size();
  g.rowptr.assign(n + 1, 0); g.col.resize(e.size());
  for (size_t k = 0; k < e.size(); ++k) { g.rowptr[e[k].first + 1]++; g.col[k] = e[k].second; }
  for (int i = 0; i < n; ++i) g.rowptr[i + 1] += g.rowptr[i];
  return g;
}

template <typename L>
static float time_us(L&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / reps;
}

template <int C>
static void run_lds(const Graph& g, const std::vector<float>& X, const std::vector<float>& ref, const std::vector<float>& rs,
                    const float* d_rs, int reps, float base_us) {
  const int n = g.n, S = 2, D = 128, slices = S * D / C;
  int np = n + 1;
  while ((np * C) % 4) ++np;
  const size_t lds = (size_t)np * C * 4;
  if (lds > 160 * 1024) { printf("  C=%d: slice %zu KB does not fit the LDS\n", C, lds / 1024); return; }
  // rows by descending degree, blocks of 64, transposed padded indices
  std::vector<int> perm(n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return g.rowptr[a + 1] - g.rowptr[a] > g.rowptr[b + 1] - g.rowptr[b]; });
  const int nblk = (n + 63) / 64;
  std::vector<unsigned> blk_off(nblk + 1, 0);
  std::vector<int> blk_k(nblk), rows((size_t)nblk * 64, -1);
  std::vector<unsigned short> idx;
  size_t padded = 0;
  for (int b = 0; b < nblk; ++b) {
    int K = 0;
    for (int l = 0; l < 64 && b * 64 + l < n; ++l) { const int r = perm[b * 64 + l]; rows[(size_t)b * 64 + l] = r; K = std::max(K, g.rowptr[r + 1] - g.rowptr[r]); }
    K = (K + 7) / 8 * 8;
    blk_k[b] = K;
    blk_off[b + 1] = blk_off[b] + K / 4;
    const size_t base = idx.size();
    idx.resize(base + (size_t)K * 64, (unsigned short)n);
    for (int l = 0; l < 64; ++l) {
      const int r = rows[(size_t)b * 64 + l];
      if (r < 0) continue;
      for (int k = g.rowptr[r]; k < g.rowptr[r + 1]; ++k) { const int kk = k - g.rowptr[r]; idx[base + ((size_t)(kk / 4) * 64 + l) * 4 + kk % 4] = (unsigned short)g.col[k]; }
    }
    padded += (size_t)K * 64;
  }
  // slice-major table
  std::vector<float> Xs((size_t)slices * np * C, 0.f), out(Xs.size());
  for (int s = 0; s < S; ++s)
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < D; ++c) Xs[((size_t)(s * (D / C) + c / C) * np + i) * C + c % C] = X[((size_t)s * n + i) * D + c];
  unsigned* d_off; int *d_k, *d_rows; unsigned short* d_idx; float *d_Xs, *d_Hs;
  CK(hipMalloc(&d_off, blk_off.size() * 4)); CK(hipMalloc(&d_k, blk_k.size() * 4)); CK(hipMalloc(&d_rows, rows.size() * 4));
  CK(hipMalloc(&d_idx, idx.size() * 2 + 4096)); CK(hipMalloc(&d_Xs, Xs.size() * 4)); CK(hipMalloc(&d_Hs, Xs.size() * 4));
  CK(hipMemcpy(d_off, blk_off.data(), blk_off.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_k, blk_k.data(), blk_k.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_idx, idx.data(), idx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_Xs, Xs.data(), Xs.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_lds_slice<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  printf("  C=%d: %d slices of %zu KB, %d row blocks, index padding %.1f %%\n", C, slices, lds / 1024, nblk, 100.0 * (padded - g.nnz) / g.nnz);
  for (int R : {1, 2, 4, 8}) {
    if (slices * R > 2048 || (slices * R < 128)) continue;
    CK(hipMemset(d_Hs, 0, Xs.size() * 4));
    const float t = time_us([&] { hipLaunchKernelGGL(k_lds_slice<C>, dim3(slices * R), dim3(1024), lds, 0, n, np, nblk, R, d_off, d_k, d_idx, d_rows, d_rs, d_Xs, d_Hs); }, reps);
    CK(hipMemcpy(out.data(), d_Hs, Xs.size() * 4, hipMemcpyDeviceToHost));
    double m = 0;
    for (int s = 0; s < S; ++s)
      for (int i = 0; i < n; ++i)
        for (int c = 0; c < D; ++c)
          m = std::max(m, (double)std::fabs(out[((size_t)(s * (D / C) + c / C) * np + i) * C + c % C] - ref[((size_t)s * n + i) * D + c]));
    printf("    R=%d (%4d workgroups): %6.1f us = %5.1f TB/s of gathered bytes (L2 baseline %.1f us), max |diff| vs baseline %.2e\n",
           R, slices * R, t, (double)g.nnz * S * D * 4 / t * 1e-6, base_us, m);
  }
  CK(hipFree(d_off)); CK(hipFree(d_k)); CK(hipFree(d_rows)); CK(hipFree(d_idx)); CK(hipFree(d_Xs)); CK(hipFree(d_Hs));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  const int sizes[] = {5776, 9369, 16264, 29910};
  for (int hic = 0; hic < 2; ++hic)
    for (int n : sizes) {
      Graph g = make_graph(n, 250000, 1000 + n, hic != 0);
      std::vector<float> rs(n), X((size_t)2 * n * 128);
      for (int i = 0; i < n; ++i) rs[i] = 1.f / (float)(g.rowptr[i + 1] - g.rowptr[i]);
      std::mt19937 rng(7);
      std::uniform_real_distribution<float> U(-1.f, 1.f);
      for (auto& v : X) v = U(rng);
      int *d_rowptr, *d_col; float *d_rs, *d_X, *d_H;
      CK(hipMalloc(&d_rowptr, (n + 1) * 4)); CK(hipMalloc(&d_col, (size_t)g.nnz * 4)); CK(hipMalloc(&d_rs, n * 4));
      CK(hipMalloc(&d_X, X.size() * 4)); CK(hipMalloc(&d_H, X.size() * 4));
      CK(hipMemcpy(d_rowptr, g.rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_col, g.col.data(), (size_t)g.nnz * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_rs, rs.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
      std::vector<float> ref(X.size());
      const float b = time_us([&] { hipLaunchKernelGGL(k_base, dim3((n + 7) / 8), dim3(512), 0, 0, n, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
      CK(hipMemcpy(ref.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
      printf("%s n=%d nnz=%d (%.1f per row) table %.1f MB, gathered %.0f MB: row-major gather out of L2, one wave per row %6.1f us = %.1f TB/s\n",
             hic ? "hic-like" : "uniform", n, g.nnz, (double)g.nnz / n, X.size() * 4 / 1e6, (double)g.nnz * 1024 / 1e6, b, (double)g.nnz * 1024 / b * 1e-6);
      run_lds<4>(g, X, ref, rs, d_rs, reps, b);
      run_lds<2>(g, X, ref, rs, d_rs, reps, b);
      run_lds<1>(g, X, ref, rs, d_rs, reps, b);
      CK(hipFree(d_rowptr)); CK(hipFree(d_col)); CK(hipFree(d_rs)); CK(hipFree(d_X)); CK(hipFree(d_H));
    }
  return 0;
}
