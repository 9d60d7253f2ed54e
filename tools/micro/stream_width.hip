// Microbenchmark (tuning tool): row-wise streaming of [M][128] fp32 arrays, 3 reads + 1 write, with 8 B/lane
// (one 512-B row per wave-instruction) versus 16 B/lane (two rows per wave-instruction) accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int RPW>
__global__ __launch_bounds__(512) void k_w2(int M, const float* a, const float* b, const float* c, float* o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int tile = blockIdx.x; tile * (8 * RPW) < M; tile += gridDim.x) {
    f32x2 va[RPW], vb[RPW], vc[RPW];
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int m = tile * 8 * RPW + wave + t * 8;
      const size_t off = (size_t)m * 128 + lane * 2;
      const bool ok = m < M;
      va[t] = ok ? *(const f32x2*)&a[off] : (f32x2){0, 0};
      vb[t] = ok ? *(const f32x2*)&b[off] : (f32x2){0, 0};
      vc[t] = ok ? *(const f32x2*)&c[off] : (f32x2){0, 0};
    }
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int m = tile * 8 * RPW + wave + t * 8;
      if (m < M) *(f32x2*)&o[(size_t)m * 128 + lane * 2] = va[t] * vb[t] + vc[t];
    }
  }
}
template <int RPW>
__global__ __launch_bounds__(512) void k_w4(int M, const float* a, const float* b, const float* c, float* o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int tile = blockIdx.x; tile * (8 * RPW) < M; tile += gridDim.x) {
    f32x4 va[RPW / 2], vb[RPW / 2], vc[RPW / 2];
#pragma unroll
    for (int t = 0; t < RPW / 2; ++t) {
      const int m = tile * 8 * RPW + (wave * 2 + (lane >> 5)) + t * 16;
      const size_t off = (size_t)m * 128 + (lane & 31) * 4;
      const bool ok = m < M;
      va[t] = ok ? *(const f32x4*)&a[off] : (f32x4){0, 0, 0, 0};
      vb[t] = ok ? *(const f32x4*)&b[off] : (f32x4){0, 0, 0, 0};
      vc[t] = ok ? *(const f32x4*)&c[off] : (f32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < RPW / 2; ++t) {
      const int m = tile * 8 * RPW + (wave * 2 + (lane >> 5)) + t * 16;
      if (m < M) *(f32x4*)&o[(size_t)m * 128 + (lane & 31) * 4] = va[t] * vb[t] + vc[t];
    }
  }
}
template <class F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 50 * 1000.f;
}
int main() {
  for (int M : {11552, 59820}) {
    float *a, *b, *c, *o;
    const size_t bytes = (size_t)M * 128 * 4;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
    for (int blocks : {256, 512, 1024}) {
      const int tiles = (M + 31) / 32;
      const int g = tiles < blocks ? tiles : blocks;
      float t2 = timeit([&] { hipLaunchKernelGGL((k_w2<4>), dim3(g), dim3(512), 0, 0, M, a, b, c, o); });
      float t4 = timeit([&] { hipLaunchKernelGGL((k_w4<4>), dim3(g), dim3(512), 0, 0, M, a, b, c, o); });
      printf("M=%d blocks=%d  8B/lane %.1f us (%.2f TB/s)   16B/lane %.1f us (%.2f TB/s)\n", M, g, t2, 4.0 * bytes / t2 / 1e6, t4,
             4.0 * bytes / t4 / 1e6);
    }
    hipFree(a); hipFree(b); hipFree(c); hipFree(o);
  }
  return 0;
}
