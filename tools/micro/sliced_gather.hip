// tools/micro/sliced_gather.hip -- experiment (not product): FEATURE-SLICED neighbour gather.
//
// The feature table of one chromosome is S*n*d*4 = n KiB (6..30 MB): several times an XCD's 4 MiB L2, so the fused
// layer kernels' row gathers miss L2 about half the time and run at the fabric / Infinity-Cache rate (7.9 TB/s at
// n = 29 k, 20 TB/s at n = 5.8 k where the table nearly fits).  Here every XCD owns ONE 128-byte column slice of the
// table -- (strand s, features 32q..32q+31), 8 slices = 2 strands x 4 quarter rows -- and aggregates ALL rows for that
// slice: its working set is n * 128 B (0.7..3.7 MB), L2 resident after first touch.  Workgroup b works on slice b & 7
// (workgroups are dispatched round-robin over the XCDs), row tile b >> 3.
//
//   base      one wave per row, 1 KiB per neighbour (the shape of gather_tile in cgcn_kernels.hip)
//   rows8     one wave = 8 rows x 8 lanes; each 8-lane group walks its own row's neighbour list, 16 B per lane =
//             one 128-B line per neighbour; column indices loaded 8 at a time and broadcast in the group by ds_swizzle
//   rows8/ctl the same kernel with slice = b / tiles (every XCD sees every slice): separates the L2 effect from the
//             access-shape effect
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/sliced_gather.hip -o build/sliced_gather
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

template <int GU>
__global__ __launch_bounds__(512) void k_base(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ rs, const float* __restrict__ X,
                                              float* __restrict__ H) {
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

// broadcast lane (8*group + U) of every 8-lane group: ds_swizzle bit-mask mode, lane' = (lane & 0x18) | U inside each 32
template <int U>
__device__ __forceinline__ int group8_bcast(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x18 | (U << 5)); }

template <int U>
struct Unroll {
  template <typename F>
  static __device__ __forceinline__ void run(F&& f) {
    Unroll<U - 1>::run(f);
    f(std::integral_constant<int, U - 1>());
  }
};
template <>
struct Unroll<0> {
  template <typename F>
  static __device__ __forceinline__ void run(F&&) {}
};

// RPW = row sets per wave (each wave handles RPW x 8 rows, one set after the other); IT = column index type (int32, or
// uint16 when n < 65 536: VERDICT r2 #5a -- half the index bytes re-read per slice)
template <bool XCD, int RPW, typename IT = int>
__global__ __launch_bounds__(512) void k_rows8(int n, int tiles, const int* __restrict__ rowptr, const IT* __restrict__ col,
                                               const float* __restrict__ rs, const float* __restrict__ X,
                                               float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = XCD ? (blockIdx.x & 7) : (blockIdx.x / tiles);
  const int tile = XCD ? (blockIdx.x >> 3) : (blockIdx.x % tiles);
  const int g = lane >> 3, j = lane & 7;
  const size_t slice_off = ((size_t)(slice >> 2) * (size_t)n * 128u + (slice & 3) * 32u + j * 4u) * 4u;
  const char* Xb = (const char*)X + slice_off;
#pragma unroll 1
  for (int r = 0; r < RPW; ++r) {
    const int i = (tile * RPW + r) * 64 + wave * 8 + g;
    int k0 = 0, k1 = 0;
    if (i < n) { k0 = rowptr[i]; k1 = rowptr[i + 1]; }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; k += 8) {
      const int myc = (int)col[min(k + j, k1 - 1)];
      f32x4 t[8];
      Unroll<8>::run([&](auto U) {
        constexpr int u = decltype(U)::value;
        t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
      });
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + u < k1) acc += t[u];
    }
    if (i < n) *(f32x4*)((char*)H + slice_off + (size_t)i * 512u) = acc * rs[i];
  }
}

// v2: (a) all 8 index broadcasts of a chunk issued back to back into 8 registers (the product kernel's ISA reuses one
// register and waits lgkmcnt(0) after every ds_swizzle: 8 serial LDS round trips per chunk), (b) 32-bit offsets from the
// uniform table base (global_load ... saddr form: one VALU per address instead of two 64-bit ones)
template <typename IT>
__global__ __launch_bounds__(512) void k_rows8_v2(int n, int tiles, const int* __restrict__ rowptr, const IT* __restrict__ col,
                                                  const float* __restrict__ rs, const float* __restrict__ X,
                                                  float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = blockIdx.x & 7, tile = blockIdx.x >> 3;
  const int g = lane >> 3, j = lane & 7;
  const unsigned slice_off = ((unsigned)(slice >> 2) * (unsigned)n * 128u + (slice & 3) * 32u + j * 4u) * 4u;
  const char* Xb = (const char*)X;
  const int i = tile * 64 + wave * 8 + g;
  int k0 = 0, k1 = 0;
  if (i < n) { k0 = rowptr[i]; k1 = rowptr[i + 1]; }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; k += 8) {
    const int myc = (int)col[min(k + j, k1 - 1)];
    unsigned off[8];
    Unroll<8>::run([&](auto U) {
      constexpr int u = decltype(U)::value;
      off[u] = ((unsigned)group8_bcast<u>(myc) << 9) + slice_off;
    });
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(Xb + (size_t)off[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < k1) acc += t[u];
  }
  if (i < n) *(f32x4*)((char*)H + (size_t)slice_off + (size_t)i * 512u) = acc * rs[i];
}

// PF 1: the column indices of the NEXT chunk are loaded before the current chunk's row loads are issued (the serial
//       chain per row becomes index latency + chunks x line latency instead of chunks x (index + line) latency)
// PF 2: additionally the lines of the next chunk are issued before the current chunk is summed (two chunks of 8 lines in
//       flight per group)
template <int PF>
__global__ __launch_bounds__(512) void k_rows8_pf(int n, int tiles, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const float* __restrict__ rs, const float* __restrict__ X,
                                                  float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = blockIdx.x & 7;
  const int tile = blockIdx.x >> 3;
  const int g = lane >> 3, j = lane & 7;
  const size_t slice_off = ((size_t)(slice >> 2) * (size_t)n * 128u + (slice & 3) * 32u + j * 4u) * 4u;
  const char* Xb = (const char*)X + slice_off;
  const int i = tile * 64 + wave * 8 + g;
  int k0 = 0, k1 = 0;
  if (i < n) { k0 = rowptr[i]; k1 = rowptr[i + 1]; }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (k0 < k1) {
    int myc = col[min(k0 + j, k1 - 1)];
    if (PF == 1) {
      for (int k = k0; k < k1; k += 8) {
        const int nextc = col[min(k + 8 + j, k1 - 1)];
        f32x4 t[8];
        Unroll<8>::run([&](auto U) {
          constexpr int u = decltype(U)::value;
          t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
        });
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k + u < k1) acc += t[u];
        myc = nextc;
      }
    } else {
      f32x4 t[8];
      Unroll<8>::run([&](auto U) {
        constexpr int u = decltype(U)::value;
        t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
      });
      myc = col[min(k0 + 8 + j, k1 - 1)];
      for (int k = k0; k < k1; k += 8) {
        const int nextc = col[min(k + 16 + j, k1 - 1)];
        f32x4 tn[8];
        Unroll<8>::run([&](auto U) {
          constexpr int u = decltype(U)::value;
          tn[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
        });
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k + u < k1) acc += t[u];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = tn[u];
        myc = nextc;
      }
    }
  }
  if (i < n) *(f32x4*)((char*)H + slice_off + (size_t)i * 512u) = acc * rs[i];
}

// rows of a 64-row tile dealt to the waves in order of their degree (perm[tile*64 + k] = row id, sorted by degree inside
// the tile): the 8 rows a wave walks together then have similar lengths, so fewer of its steps run with groups masked off
__global__ __launch_bounds__(512) void k_rows8_sorted(int n, int tiles, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const int* __restrict__ perm, const float* __restrict__ rs,
                                                      const float* __restrict__ X, float* __restrict__ H) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = blockIdx.x & 7;
  const int tile = blockIdx.x >> 3;
  const int g = lane >> 3, j = lane & 7;
  const size_t slice_off = ((size_t)(slice >> 2) * (size_t)n * 128u + (slice & 3) * 32u + j * 4u) * 4u;
  const char* Xb = (const char*)X + slice_off;
  const int slot = tile * 64 + wave * 8 + g;
  const int i = slot < n ? perm[slot] : n;
  int k0 = 0, k1 = 0;
  if (i < n) { k0 = rowptr[i]; k1 = rowptr[i + 1]; }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; k += 8) {
    const int myc = col[min(k + j, k1 - 1)];
    f32x4 t[8];
    Unroll<8>::run([&](auto U) {
      constexpr int u = decltype(U)::value;
      t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
    });
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < k1) acc += t[u];
  }
  if (i < n) *(f32x4*)((char*)H + slice_off + (size_t)i * 512u) = acc * rs[i];
}

// RW rows per wave (8 / RW groups of 8 lanes share a row, 8 * 8 / RW neighbours per step, summed over the groups by a
// butterfly at the end): fewer rows per wave = less time lost to the longest row of the wave, at a few shuffles per row
template <int RW>
__global__ __launch_bounds__(512) void k_rowsN(int n, int tiles, const int* __restrict__ rowptr, const int* __restrict__ col,
                                               const float* __restrict__ rs, const float* __restrict__ X,
                                               float* __restrict__ H) {
  constexpr int GPR = 8 / RW;          // groups per row
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slice = blockIdx.x & 7;
  const int tile = blockIdx.x >> 3;    // 8 * RW rows per workgroup
  const int g = lane >> 3, j = lane & 7;
  const int rw = g / GPR, gr = g % GPR;
  const size_t slice_off = ((size_t)(slice >> 2) * (size_t)n * 128u + (slice & 3) * 32u + j * 4u) * 4u;
  const char* Xb = (const char*)X + slice_off;
  const int i = tile * (8 * RW) + wave * RW + rw;
  int k0 = 0, k1 = 0;
  if (i < n) { k0 = rowptr[i]; k1 = rowptr[i + 1]; }
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = k0 + gr * 8; k < k1; k += 8 * GPR) {
    const int myc = col[min(k + j, k1 - 1)];
    f32x4 t[8];
    Unroll<8>::run([&](auto U) {
      constexpr int u = decltype(U)::value;
      t[u] = *(const f32x4*)(Xb + (size_t)(unsigned)group8_bcast<u>(myc) * 512u);
    });
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < k1) acc += t[u];
  }
#pragma unroll
  for (int o = 8; o < 8 * GPR; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (i < n && gr == 0) *(f32x4*)((char*)H + slice_off + (size_t)i * 512u) = acc * rs[i];
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

template <typename L>
static float time_us(L&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms * 1e3f / reps;
}

static double max_err(const std::vector<float>& a, const std::vector<float>& b) {
  double m = 0;
  for (size_t i = 0; i < a.size(); ++i) m = std::max(m, (double)std::fabs(a[i] - b[i]));
  return m;
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
      printf("%s n=%d nnz=%d table=%.1f MB (slice %.2f MB)\n", hic ? "hic-like" : "uniform", n, g.nnz, X.size() * 4 / 1e6, n * 128 / 1e6);
      std::vector<float> ref(X.size()), out(X.size());
      const float b2 = time_us([&] { hipLaunchKernelGGL((k_base<2>), dim3((n + 7) / 8), dim3(512), 0, 0, n, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
      CK(hipMemcpy(ref.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
      const float b3 = time_us([&] { hipLaunchKernelGGL((k_base<3>), dim3((n + 7) / 8), dim3(512), 0, 0, n, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
      printf("  base (1 KiB rows) GU2 %6.1f GU3 %6.1f us\n", b2, b3);
      auto sliced = [&](auto XCD, auto RPW, const char* label) {
        constexpr bool xcd = decltype(XCD)::value;
        constexpr int rpw = decltype(RPW)::value;
        const int tiles = (n + 64 * rpw - 1) / (64 * rpw);
        CK(hipMemset(d_H, 0, X.size() * 4));
        const float t = time_us([&] { hipLaunchKernelGGL((k_rows8<xcd, rpw>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
        CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
        printf("  %-34s %6.1f us  (max |diff| vs base %.2e)\n", label, t, max_err(ref, out));
      };
      sliced(std::true_type(), std::integral_constant<int, 1>(), "rows8 slice=b&7 (XCD-owned) RPW1");
      {   // the same kernel on 16-bit column indices (every chromosome here has n < 65 536)
        std::vector<unsigned short> c16(g.col.begin(), g.col.end());
        unsigned short* d_c16;
        CK(hipMalloc(&d_c16, (size_t)g.nnz * 2));
        CK(hipMemcpy(d_c16, c16.data(), (size_t)g.nnz * 2, hipMemcpyHostToDevice));
        const int tiles = (n + 63) / 64;
        CK(hipMemset(d_H, 0, X.size() * 4));
        const float a = time_us([&] { hipLaunchKernelGGL((k_rows8<true, 1, int>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
        const float b = time_us([&] { hipLaunchKernelGGL((k_rows8<true, 1, unsigned short>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_c16, d_rs, d_X, d_H); }, reps);
        const float a2 = time_us([&] { hipLaunchKernelGGL((k_rows8<true, 1, int>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
        const float b2 = time_us([&] { hipLaunchKernelGGL((k_rows8<true, 1, unsigned short>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_c16, d_rs, d_X, d_H); }, reps);
        CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
        printf("  %-34s int32 %6.1f / %6.1f us   uint16 %6.1f / %6.1f us  (interleaved; max |diff| vs base %.2e)\n", "rows8 XCD-owned, index width", a, a2, b, b2, max_err(ref, out));
        CK(hipMemset(d_H, 0, X.size() * 4));
        const float c1 = time_us([&] { hipLaunchKernelGGL((k_rows8_v2<unsigned short>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_c16, d_rs, d_X, d_H); }, reps);
        const float b3 = time_us([&] { hipLaunchKernelGGL((k_rows8<true, 1, unsigned short>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_c16, d_rs, d_X, d_H); }, reps);
        const float c2 = time_us([&] { hipLaunchKernelGGL((k_rows8_v2<unsigned short>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_c16, d_rs, d_X, d_H); }, reps);
        CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
        printf("  %-34s uint16 %6.1f us   v2 %6.1f / %6.1f us  (max |diff| vs base %.2e)\n", "rows8 v2 (batched swizzles, saddr)", b3, c1, c2, max_err(ref, out));
        CK(hipFree(d_c16));
      }
                  sliced(std::false_type(), std::integral_constant<int, 1>(), "rows8 slice=b/tiles (control) RPW1");
      {
        const int tiles = (n + 63) / 64;
        CK(hipMemset(d_H, 0, X.size() * 4));
        const float t1 = time_us([&] { hipLaunchKernelGGL((k_rows8_pf<1>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
        CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
        printf("  %-34s %6.1f us  (max |diff| vs base %.2e)\n", "rows8 + next-chunk index prefetch", t1, max_err(ref, out));
        {
          std::vector<int> perm(n);
          for (int i = 0; i < n; ++i) perm[i] = i;
          for (int t0 = 0; t0 < n; t0 += 64) {
            const int t1 = std::min(n, t0 + 64);
            std::stable_sort(perm.begin() + t0, perm.begin() + t1, [&](int a, int b) { return g.rowptr[a + 1] - g.rowptr[a] > g.rowptr[b + 1] - g.rowptr[b]; });
          }
          int* d_perm;
          CK(hipMalloc(&d_perm, n * 4));
          CK(hipMemcpy(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice));
          CK(hipMemset(d_H, 0, X.size() * 4));
          const float ts = time_us([&] { hipLaunchKernelGGL(k_rows8_sorted, dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_perm, d_rs, d_X, d_H); }, reps);
          CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
          printf("  %-34s %6.1f us  (max |diff| vs base %.2e)\n", "rows8, rows degree-sorted in tile", ts, max_err(ref, out));
          CK(hipFree(d_perm));
        }
        {
          auto rowsN = [&](auto RW, const char* label) {
            constexpr int rw = decltype(RW)::value;
            const int tl = (n + 8 * rw - 1) / (8 * rw);
            CK(hipMemset(d_H, 0, X.size() * 4));
            const float t = time_us([&] { hipLaunchKernelGGL((k_rowsN<rw>), dim3(8 * tl), dim3(512), 0, 0, n, tl, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
            CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
            printf("  %-34s %6.1f us  (max |diff| vs base %.2e)\n", label, t, max_err(ref, out));
          };
          rowsN(std::integral_constant<int, 4>(), "4 rows per wave (2 groups per row)");
          rowsN(std::integral_constant<int, 2>(), "2 rows per wave (4 groups per row)");
          rowsN(std::integral_constant<int, 1>(), "1 row per wave (8 groups per row)");
        }
        CK(hipMemset(d_H, 0, X.size() * 4));
        const float t2 = time_us([&] { hipLaunchKernelGGL((k_rows8_pf<2>), dim3(8 * tiles), dim3(512), 0, 0, n, tiles, d_rowptr, d_col, d_rs, d_X, d_H); }, reps);
        CK(hipMemcpy(out.data(), d_H, X.size() * 4, hipMemcpyDeviceToHost));
        printf("  %-34s %6.1f us  (max |diff| vs base %.2e)\n", "rows8 + index and line prefetch", t2, max_err(ref, out));
      }
      CK(hipFree(d_rowptr)); CK(hipFree(d_col)); CK(hipFree(d_rs)); CK(hipFree(d_X)); CK(hipFree(d_H));
    }
  return 0;
}
