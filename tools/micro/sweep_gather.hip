// tools/micro/sweep_gather.hip -- experiment (not product): does a COLUMN-SWEEP ordering of the neighbour gather cut the
// traffic beyond L2 when the feature table (S*n*d*4 bytes) is several times an XCD's 4 MiB L2?
//
//   baseline  one wave per row, grid = n/8 workgroups of 8 waves (the shape of gather_tile in cgcn_kernels.hip): rows
//             sweep their sorted column lists independently; with more workgroups than fit on the chip at once the
//             sweeps of different rounds are out of phase.
//   seq<Q>    workgroup owns 8Q nodes, wave w gathers rows w, w+8, ... one after the other (Q sweeps per wave).
//   sweep<Q>  same ownership, but the wave walks ONE merged list of its Q rows' edges sorted by column (row id packed
//             into the top bits): every wave of the chip moves through the column space once, roughly in step, so the
//             rows being fetched at any time form a narrow band that fits the L2s.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/sweep_gather.hip -o /tmp/sg && /tmp/sg
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// S = 2, D = 128: one neighbour = 1 KiB = one dwordx4 wave-load (lanes 0-31 strand 0, 32-63 strand 1)
template <int Q, int GU, bool MERGED>
__global__ __launch_bounds__(512) void k_gather(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const int* __restrict__ gptr, const int* __restrict__ packed,
                                                const float* __restrict__ rs, const float* __restrict__ X,
                                                float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node0 = blockIdx.x * 8 * Q;
  const unsigned lane_off = ((unsigned)(lane >> 5) * (unsigned)n * 128u + (lane & 31) * 4u) * 4u;
  const char* Xb = (const char*)X;
  f32x4 acc[Q];
#pragma unroll
  for (int r = 0; r < Q; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (MERGED) {
    const int gid = blockIdx.x * 8 + wave;
    const int k0 = gptr[gid], k1 = gptr[gid + 1];
    for (int kb = k0; kb < k1; kb += 64) {
      const int cnt = min(64, k1 - kb);
      const int myp = lane < cnt ? packed[kb + lane] : 0;
      for (int b = 0; b < cnt; b += GU) {
        f32x4 t[GU];
        int rid[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          const int p = rl_i(myp, min(b + u, cnt - 1));
          rid[u] = (b + u < cnt) ? (int)((unsigned)p >> 28) : -1;
          t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)(p & 0x0FFFFFFF) * 512u + lane_off);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
#pragma unroll
          for (int r = 0; r < Q; ++r)
            if (rid[u] == r) acc[r] += t[u];   // rid is wave-uniform (SGPR)
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < Q; ++r) {
      const int i = node0 + wave + 8 * r;
      if (i >= n) continue;
      const int k0 = rowptr[i], k1 = rowptr[i + 1];
      for (int kb = k0; kb < k1; kb += 64) {
        const int cnt = min(64, k1 - kb);
        const int myc = lane < cnt ? col[kb + lane] : 0;
        for (int b = 0; b < cnt; b += GU) {
          f32x4 t[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u)
            t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)rl_i(myc, min(b + u, cnt - 1)) * 512u + lane_off);
#pragma unroll
          for (int u = 0; u < GU; ++u)
            if (b + u < cnt) acc[r] += t[u];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    const int i = node0 + wave + 8 * r;
    if (i < n) *(f32x4*)((char*)H + (size_t)i * 512u + lane_off) = acc[r] * rs[i];
  }
}

struct Graph;
typedef Graph Graph_;

__device__ __forceinline__ int xcd_contiguous(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

//   window<Q,W> workgroup owns 8Q nodes (wave w: rows w, w+8, ...) and first stages the rows of nodes
//               [node0 - W, node0 + 8Q + W) in LDS (1 KiB per node, both strands); neighbours inside that window are
//               read from LDS (ds_read_b128, conflict-free), the others from L2 as before.  XCD-contiguous tile order.
// ILV: node-major feature layout [n][2][128] (one contiguous 1 KiB segment per neighbour) instead of the strand-major
// [2][n][128] (two 512-B segments n*512 B apart)
template <int GU, bool ILV>
__global__ __launch_bounds__(512) void k_layout(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const float* __restrict__ rs, const float* __restrict__ X,
                                                float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node0 = xcd_contiguous(blockIdx.x, gridDim.x) * 8;
  const unsigned lane_off = ILV ? (unsigned)lane * 16u : ((unsigned)(lane >> 5) * (unsigned)n * 128u + (lane & 31) * 4u) * 4u;
  const unsigned row_b = ILV ? 1024u : 512u;
  const char* Xb = (const char*)X;
  const int i = node0 + wave;
  if (i >= n) return;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int k0 = rowptr[i], k1 = rowptr[i + 1];
  for (int kb = k0; kb < k1; kb += 64) {
    const int cnt = min(64, k1 - kb);
    const int myc = lane < cnt ? col[kb + lane] : 0;
    for (int b = 0; b < cnt; b += GU) {
      f32x4 t[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)rl_i(myc, min(b + u, cnt - 1)) * row_b + lane_off);
#pragma unroll
      for (int u = 0; u < GU; ++u)
        if (b + u < cnt) acc += t[u];
    }
  }
  *(f32x4*)((char*)H + (size_t)i * row_b + lane_off) = acc * rs[i];
}

// Premise test for "the last-arriving wave runs the tile's dense phase alone": the bare gather followed by a simulated
// dense phase of `ticks` (100 MHz) --  MODE 1: all 8 waves of the workgroup sit through it (today's structure: barrier,
// MFMA, epilogue);  MODE 2: the waves count their arrival in LDS, seven exit at once (their wave slots can take new
// workgroups), the last one sits through 1.6 x ticks alone.
template <int GU, int MODE>
__global__ __launch_bounds__(512) void k_tail(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ rs, const float* __restrict__ X,
                                              float* __restrict__ H, int ticks) {
  __shared__ int arrived;
  __shared__ __attribute__((aligned(16))) float T[16 * 132];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (MODE == 2 && threadIdx.x == 0) arrived = 0;
  if (MODE == 2) __syncthreads();
  const int node0 = xcd_contiguous(blockIdx.x, gridDim.x) * 8;
  const unsigned lane_off = ((unsigned)(lane >> 5) * (unsigned)n * 128u + (lane & 31) * 4u) * 4u;
  const char* Xb = (const char*)X;
  const int i = node0 + wave;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    for (int kb = k0; kb < k1; kb += 64) {
      const int cnt = min(64, k1 - kb);
      const int myc = lane < cnt ? col[kb + lane] : 0;
      for (int b = 0; b < cnt; b += GU) {
        f32x4 t[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)rl_i(myc, min(b + u, cnt - 1)) * 512u + lane_off);
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (b + u < cnt) acc += t[u];
      }
    }
    *(f32x4*)((char*)H + (size_t)i * 512u + lane_off) = acc * rs[i];
  }
  *(f32x4*)&T[((lane >> 5) * 8 + wave) * 132 + (lane & 31) * 4] = acc;
  if (MODE == 0) return;
  unsigned long long t0;
  if (MODE == 1) {
    __syncthreads();
    t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(4);
  } else {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my tile row is in LDS before I count myself in
    int old = 0;
    if (lane == 0) old = atomicAdd(&arrived, 1);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != 7) return;                // seven waves leave; their slots are free for the next workgroup
    t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < (ticks * 8) / 5) __builtin_amdgcn_s_sleep(4);
  }
  if (T[lane] == 123456.f) H[0] = 0.f;   // keep T alive
}

template <int GU, int MODE>
static float run_tail(const Graph_& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H, int ticks, int reps);

template <int Q, int W, int GU, bool USE_LDS>
__global__ __launch_bounds__(512) void k_window(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const float* __restrict__ rs, const float* __restrict__ X,
                                                float* __restrict__ H) {
  constexpr int NR = 8 * Q + 2 * W;
  extern __shared__ __attribute__((aligned(16))) char win[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node0 = xcd_contiguous(blockIdx.x, gridDim.x) * 8 * Q;
  const unsigned lane_off = ((unsigned)(lane >> 5) * (unsigned)n * 128u + (lane & 31) * 4u) * 4u;
  const char* Xb = (const char*)X;
  const int lo = max(0, node0 - W), hi = min(n, node0 + 8 * Q + W);
  if (USE_LDS) {
    f32x4 t[(NR + 7) / 8];
#pragma unroll
    for (int k = 0; k < (NR + 7) / 8; ++k) {
      const int r = wave + 8 * k;
      t[k] = *(const f32x4*)(Xb + (size_t)(unsigned)min(lo + r, n - 1) * 512u + lane_off);
    }
#pragma unroll
    for (int k = 0; k < (NR + 7) / 8; ++k) {
      const int r = wave + 8 * k;
      if (r < NR) *(f32x4*)(win + r * 1024 + lane * 16) = t[k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < Q; ++r) {
    const int i = node0 + wave + 8 * r;
    if (i >= n) continue;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    for (int kb = k0; kb < k1; kb += 64) {
      const int cnt = min(64, k1 - kb);
      const int myc = lane < cnt ? col[kb + lane] : 0;
      for (int b = 0; b < cnt; b += GU) {
        f32x4 t[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
          const int c = rl_i(myc, min(b + u, cnt - 1));
          if (USE_LDS && c >= lo && c < hi) t[u] = *(const f32x4*)(win + (c - lo) * 1024 + lane * 16);
          else t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)c * 512u + lane_off);
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
          if (b + u < cnt) acc += t[u];
      }
    }
    *(f32x4*)((char*)H + (size_t)i * 512u + lane_off) = acc * rs[i];
  }
}

struct Graph {
  int n, nnz;
  std::vector<int> rowptr, col;
};

static Graph make_graph(int n, int pairs, unsigned seed, bool hic_like) {
  std::mt19937_64 rng(seed);
  std::vector<std::pair<int, int>> e;
  e.reserve(2 * (size_t)pairs + n);
  std::uniform_int_distribution<int> U(0, n - 1);
  std::uniform_real_distribution<double> R(0.0, 1.0);
  for (int p = 0; p < pairs; ++p) {
    int i, j;
    if (hic_like) {
      const double kmax = std::max(2, n - 1);
      int dist = (int)std::floor(std::exp(R(rng) * std::log(kmax)));
      dist = std::min(std::max(dist, 1), n - 1);
      i = (int)(R(rng) * (n - dist));
      j = i + dist;
    } else {
      i = U(rng);
      j = U(rng);
    }
    if (i == j) continue;
    e.push_back({i, j});
    e.push_back({j, i});
  }
  for (int i = 0; i < n; ++i) e.push_back({i, i});
  std::sort(e.begin(), e.end());
  e.erase(std::unique(e.begin(), e.end()), e.end());
  Graph g;
  g.n = n;
  g.nnz = (int)e.size();
  g.rowptr.assign(n + 1, 0);
  g.col.resize(e.size());
  for (size_t k = 0; k < e.size(); ++k) {
    g.rowptr[e[k].first + 1]++;
    g.col[k] = e[k].second;
  }
  for (int i = 0; i < n; ++i) g.rowptr[i + 1] += g.rowptr[i];
  return g;
}

// merged, column-sorted edge lists per (workgroup, wave) for tile height 8Q
static void make_sweep(const Graph& g, int Q, std::vector<int>& gptr, std::vector<int>& packed) {
  const int groups = ((g.n + 8 * Q - 1) / (8 * Q)) * 8;
  gptr.assign(groups + 1, 0);
  packed.clear();
  packed.reserve(g.nnz);
  std::vector<std::pair<int, int>> tmp;
  for (int gid = 0; gid < groups; ++gid) {
    const int wg = gid / 8, w = gid % 8;
    tmp.clear();
    for (int r = 0; r < Q; ++r) {
      const int i = wg * 8 * Q + w + 8 * r;
      if (i >= g.n) continue;
      for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k) tmp.push_back({g.col[k], r});
    }
    std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
    for (auto& t : tmp) packed.push_back(t.first | (t.second << 28));
    gptr[gid + 1] = (int)packed.size();
  }
}

template <int Q, int GU, bool MERGED>
static float run(const Graph& g, const int* d_rowptr, const int* d_col, const int* d_gptr, const int* d_packed,
                 const float* d_rs, const float* d_X, float* d_H, int reps) {
  const int grid = (g.n + 8 * Q - 1) / (8 * Q);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((k_gather<Q, GU, MERGED>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((k_gather<Q, GU, MERGED>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

template <int Q>
static void bench_q(const Graph& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H,
                    std::vector<float>& h_ref, int reps) {
  std::vector<int> gptr, packed;
  make_sweep(g, Q, gptr, packed);
  int *d_gptr, *d_packed;
  CK(hipMalloc(&d_gptr, gptr.size() * 4));
  CK(hipMalloc(&d_packed, packed.size() * 4));
  CK(hipMemcpy(d_gptr, gptr.data(), gptr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_packed, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
  const size_t elems = (size_t)2 * g.n * 128;
  std::vector<float> h(elems);
  auto check = [&](const char* what) {
    CK(hipMemcpy(h.data(), d_H, elems * 4, hipMemcpyDeviceToHost));
    if (h_ref.empty()) { h_ref = h; return; }
    double md = 0;
    for (size_t i = 0; i < elems; ++i) md = std::max(md, (double)std::fabs(h[i] - h_ref[i]));
    if (md != 0.0) printf("   !! %s Q=%d differs from baseline by %g\n", what, Q, md);
  };
  const float s2 = run<Q, 2, false>(g, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H, reps); check("seq");
  const float s3 = run<Q, 3, false>(g, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H, reps);
  const float m2 = run<Q, 2, true>(g, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H, reps); check("sweep");
  const float m3 = run<Q, 3, true>(g, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H, reps);
  const float m4 = run<Q, 4, true>(g, d_rowptr, d_col, d_gptr, d_packed, d_rs, d_X, d_H, reps);
  printf("  Q=%d (grid %5d): seq GU2 %6.1f GU3 %6.1f | sweep GU2 %6.1f GU3 %6.1f GU4 %6.1f us\n", Q, (g.n + 8 * Q - 1) / (8 * Q), s2, s3, m2, m3, m4);
  CK(hipFree(d_gptr));
  CK(hipFree(d_packed));
}


template <int Q, int W, int GU, bool USE_LDS>
static float run_window(const Graph& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H, int reps) {
  const int grid = (g.n + 8 * Q - 1) / (8 * Q);
  const size_t lds = USE_LDS ? (size_t)(8 * Q + 2 * W) * 1024 : 0;
  auto kern = k_window<Q, W, GU, USE_LDS>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H);
  CK(hipGetLastError());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

template <int GU, bool ILV>
static float run_layout(const Graph& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H, int reps) {
  const int grid = (g.n + 7) / 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_layout<GU, ILV>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_layout<GU, ILV>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

template <int GU, int MODE>
static float run_tail(const Graph& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H, int ticks, int reps) {
  const int grid = (g.n + 7) / 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_tail<GU, MODE>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H, ticks);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_tail<GU, MODE>), dim3(grid), dim3(512), 0, 0, g.n, d_rowptr, d_col, d_rs, d_X, d_H, ticks);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / reps;
}

static void bench_window(const Graph& g, const int* d_rowptr, const int* d_col, const float* d_rs, const float* d_X, float* d_H,
                         const std::vector<float>& h_ref, int reps) {
  const size_t elems = (size_t)2 * g.n * 128;
  std::vector<float> h(elems);
  auto check = [&](const char* what) {
    CK(hipMemcpy(h.data(), d_H, elems * 4, hipMemcpyDeviceToHost));
    double md = 0;
    for (size_t i = 0; i < elems; ++i) md = std::max(md, (double)std::fabs(h[i] - h_ref[i]));
    if (md != 0.0) printf("   !! %s differs from baseline by %g\n", what, md);
  };
  // in-window share of the gathers for the (Q, W) geometries below
  auto share = [&](int Q, int W) {
    long long in = 0;
    for (int i = 0; i < g.n; ++i) {
      const int node0 = (i / (8 * Q)) * 8 * Q, lo = std::max(0, node0 - W), hi = std::min(g.n, node0 + 8 * Q + W);
      for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k) in += (g.col[k] >= lo && g.col[k] < hi);
    }
    return (double)in / g.nnz;
  };
#define WV(Q_, W_, GU_) do { \
    const float a = run_window<Q_, W_, GU_, false>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps); \
    const float b = run_window<Q_, W_, GU_, true>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps); check("window"); \
    printf("  window Q=%d W=%3d (%3d KB LDS, %4.1f%% of gathers in window) GU%d: L2 only %6.1f | LDS window %6.1f us\n", Q_, W_, (8 * Q_ + 2 * W_), 100.0 * share(Q_, W_), GU_, a, b); } while (0)
  WV(1, 28, 2); WV(1, 28, 4);
  WV(2, 24, 2); WV(2, 24, 4);
  WV(4, 16, 2); WV(4, 16, 4);
  WV(2, 56, 4); WV(4, 48, 4); WV(4, 48, 8);
#undef WV
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  const int sizes[] = {5776, 7563, 9369, 12304, 16264, 20534, 29184};
  for (int hic = 0; hic < 2; ++hic)
    for (int n : sizes) {
      Graph g = make_graph(n, 250000, 1000 + n, hic != 0);
      std::vector<float> rs(n), X((size_t)2 * n * 128);
      for (int i = 0; i < n; ++i) rs[i] = 1.f / (float)(g.rowptr[i + 1] - g.rowptr[i]);
      std::mt19937 rng(7);
      std::uniform_real_distribution<float> U(-1.f, 1.f);
      for (auto& v : X) v = U(rng);
      int *d_rowptr, *d_col;
      float *d_rs, *d_X, *d_H;
      CK(hipMalloc(&d_rowptr, (n + 1) * 4));
      CK(hipMalloc(&d_col, (size_t)g.nnz * 4));
      CK(hipMalloc(&d_rs, n * 4));
      CK(hipMalloc(&d_X, X.size() * 4));
      CK(hipMalloc(&d_H, X.size() * 4));
      CK(hipMemcpy(d_rowptr, g.rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_col, g.col.data(), (size_t)g.nnz * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_rs, rs.data(), n * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
      printf("%s n=%d nnz=%d table=%.1f MB\n", hic ? "hic-like" : "uniform", n, g.nnz, X.size() * 4 / 1e6);
      std::vector<float> h_ref;
      if (getenv("SG_TAIL_ONLY")) {
        printf("  dense phase 5.5 us: gather only %6.1f | all 8 waves wait %6.1f | 7 exit, last waits 8.8 us %6.1f us\n",
               run_tail<2, 0>(g, d_rowptr, d_col, d_rs, d_X, d_H, 550, reps), run_tail<2, 1>(g, d_rowptr, d_col, d_rs, d_X, d_H, 550, reps),
               run_tail<2, 2>(g, d_rowptr, d_col, d_rs, d_X, d_H, 550, reps));
      } else if (getenv("SG_LAYOUT_ONLY")) {
        printf("  layout: strand-major GU2 %6.1f GU3 %6.1f | node-major (1 KiB contiguous) GU2 %6.1f GU3 %6.1f us\n",
               run_layout<2, false>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps), run_layout<3, false>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps),
               run_layout<2, true>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps), run_layout<3, true>(g, d_rowptr, d_col, d_rs, d_X, d_H, reps));
      } else {
        bench_q<1>(g, d_rowptr, d_col, d_rs, d_X, d_H, h_ref, reps);
        bench_q<2>(g, d_rowptr, d_col, d_rs, d_X, d_H, h_ref, reps);
        bench_q<3>(g, d_rowptr, d_col, d_rs, d_X, d_H, h_ref, reps);
        bench_q<4>(g, d_rowptr, d_col, d_rs, d_X, d_H, h_ref, reps);
        bench_window(g, d_rowptr, d_col, d_rs, d_X, d_H, h_ref, reps);
      }
      CK(hipFree(d_rowptr)); CK(hipFree(d_col)); CK(hipFree(d_rs)); CK(hipFree(d_X)); CK(hipFree(d_H));
    }
  return 0;
}
