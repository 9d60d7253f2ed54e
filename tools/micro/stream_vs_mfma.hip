// tools/micro/stream_vs_mfma.hip -- experiment (not product): does fp32-MFMA work on a CU slow a memory stream issued by
// OTHER waves of the same CU (and vice versa)?  The row-local kernels stream ~5 TB/s without their products and their
// products run at full rate without the stream, but together the kernel takes about the SUM (profiles/
// r03_rowlocal_rs_experiment.txt).  One 16-wave workgroup per CU: waves 0-7 copy a large array (16 B per lane, several
// loads in flight), waves 8-15 run an LDS-fed v_mfma_f32_16x16x4_f32 loop (9 operand vectors per 32 MFMAs, the dW-product
// ratio).  Three launches: stream only, MFMA only, both; reports TB/s (read + write) and TF/s.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/stream_vs_mfma.hip -o /tmp/svm && /tmp/svm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool STREAM, bool MFMA, int VALU>
__global__ __launch_bounds__(1024) void k(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4, int mfma_iters,
                                          float* out) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  if (wave < 8) {
    if (!STREAM) return;
    // 512 stream threads per workgroup; 4 independent 16-byte loads in flight per lane
    const size_t stride = (size_t)gridDim.x * 512;
    size_t i = (size_t)blockIdx.x * 512 + tid;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
      // VALU: vector instructions of "row math" per 16 bytes streamed (4 independent chains of packed-free FMAs per vector)
#pragma unroll
      for (int v = 0; v < VALU / 4; ++v) {
        a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f;
      }
      dst[i] = a * 1.5f; dst[i + stride] = b * 1.5f; dst[i + 2 * stride] = c * 1.5f; dst[i + 3 * stride] = d * 1.5f;
    }
    for (; i < n4; i += stride) dst[i] = src[i] * 1.5f;
  } else {
    if (!MFMA) return;
    const int t = tid - 512;
    f32x4 acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a = *(const f32x4*)&lds[t * 4], bv[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) bv[b] = *(const f32x4*)&lds[((t + 64 * b) & 4095) * 4];
    for (int it = 0; it < mfma_iters; ++it) {
      const int o = (it & 3) * 4096;
      f32x4 an = *(const f32x4*)&lds[o + t * 4], bn[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) bn[b] = *(const f32x4*)&lds[o + ((t + 64 * (b + 1)) & 4095) * 4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[b][u], acc[b], 0, 0, 0);
      a = an;
#pragma unroll
      for (int b = 0; b < 8; ++b) bv[b] = bn[b];
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int b = 1; b < 8; ++b) s += acc[b];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = s[0];
  }
}

int main() {
  const size_t bytes = 160u << 20;   // 160 MB read + 160 MB written per launch (the row-local kernel at chr1 size: 153 MB)
  const size_t n4 = bytes / 16;
  f32x4 *src, *dst;
  float* out;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t lds_bytes = 16 * 1024 * 4;
  auto timeit = [&](auto launch) {
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 100.0;   // us per launch
  };
  auto sweep = [&](auto tag, const char* name) {
    constexpr int V = decltype(tag)::value;
    for (int iters : {45, 60}) {   // MFMA work per launch: iters x 32 MFMAs per wave, 8 waves per CU
      const double flop = (double)iters * 32 * 2048 * 8 * 256;
      const double t_s = timeit([&] { hipLaunchKernelGGL((k<true, false, V>), dim3(256), dim3(1024), lds_bytes, 0, src, dst, n4, iters, out); });
      const double t_m = timeit([&] { hipLaunchKernelGGL((k<false, true, V>), dim3(256), dim3(1024), lds_bytes, 0, src, dst, n4, iters, out); });
      const double t_b = timeit([&] { hipLaunchKernelGGL((k<true, true, V>), dim3(256), dim3(1024), lds_bytes, 0, src, dst, n4, iters, out); });
      printf("%-34s MFMA %.2f GFLOP: stream alone %6.1f us (%.2f TB/s) | MFMA alone %6.1f us (%.1f TF/s) | both %6.1f us (sum %.1f, max %.1f)\n",
             name, flop / 1e9, t_s, 2.0 * bytes / (t_s * 1e-6) / 1e12, t_m, flop / (t_m * 1e-6) / 1e12, t_b, t_s + t_m, t_s > t_m ? t_s : t_m);
    }
  };
  sweep(std::integral_constant<int, 0>{}, "stream waves: copy only");
  sweep(std::integral_constant<int, 16>{}, "+ 16 x 4 FMAs per 16 B (64 VALU)");
  sweep(std::integral_constant<int, 32>{}, "+ 32 x 4 FMAs per 16 B (128 VALU)");
  sweep(std::integral_constant<int, 64>{}, "+ 64 x 4 FMAs per 16 B (256 VALU)");
  return 0;
}
