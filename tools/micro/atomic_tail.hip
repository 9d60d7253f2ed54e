// tools/micro/atomic_tail.hip -- experiment (not product): what order-independent accumulation by integer atomics costs at
// the tail of a streaming kernel (candidate (d): BatchNorm column sums without a finalize launch).
// A persistent streaming kernel of k_layer_dense's shape (512 workgroups x 512 threads, ~62 MB of traffic) ends with every
// workgroup adding V values as W 64-bit integer atomics each into SLOTS copies of the totals; compared with the same kernel
// writing one 2 KB record per workgroup (what ships), in captured chains [kernel ; tiny consumer].
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_tail.hip -o build/atomic_tail && ./build/atomic_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: one record per workgroup (plain stores); 1: W atomics per value into totals[slot]
template <int MODE, int W>
__global__ __launch_bounds__(512) void k_stream(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n4, float* __restrict__ rec,
                                                unsigned long long* __restrict__ tot, int values, int slots) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t k = i; k < n4; k += stride) {
    const f32x4 v = in[k];
    acc += v;
    out[k] = v * 2.f;
  }
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if ((int)threadIdx.x < values) {
    if (MODE == 0) rec[(size_t)blockIdx.x * values + threadIdx.x] = s;
    else {
      const double d = (double)s * 1.0000001;
      const long long hi = (long long)__builtin_floor(d);
      const long long lo = (long long)((d - (double)hi) * 1099511627776.0);
      unsigned long long* t = tot + ((size_t)(blockIdx.x % slots) * values + threadIdx.x) * W;
      atomicAdd(&t[0], (unsigned long long)hi);
      if (W > 1) atomicAdd(&t[1], (unsigned long long)lo);
      if (W > 2) atomicAdd(&t[2], (unsigned long long)(lo >> 7));
    }
  }
}
__global__ void k_consume(const unsigned long long* tot, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)tot[i];
}

template <typename F>
static float chain_us(hipStream_t st, int reps, F body) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int r = 0; r < reps; ++r) body();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(a, st));
  for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / (10.f * reps);
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t n4 = (31u << 20) / 16;   // 31 MB in, 31 MB out
  f32x4 *in, *out; float *rec, *o2; unsigned long long* tot;
  CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&out, n4 * 16)); CK(hipMalloc(&rec, 512 * 512 * 4)); CK(hipMalloc(&o2, 1 << 20));
  CK(hipMalloc(&tot, 8 * 512 * 3 * 8)); CK(hipMemset(tot, 0, 8 * 512 * 3 * 8)); CK(hipMemset(in, 0, n4 * 16));
  const int G = 512;
  printf("streaming kernel (31 MB in + 31 MB out, %d x 512) followed by a tiny consumer, us per pair in a captured chain of 20:\n", G);
  float base = chain_us(st, 20, [&] { hipLaunchKernelGGL((k_stream<0, 1>), dim3(G), dim3(512), 0, st, in, out, n4, rec, tot, 0, 1); hipLaunchKernelGGL(k_consume, dim3(2), dim3(512), 0, st, tot, o2, 1024); });
  printf("  no statistics at all                                  %6.2f\n", base);
  float r = chain_us(st, 20, [&] { hipLaunchKernelGGL((k_stream<0, 1>), dim3(G), dim3(512), 0, st, in, out, n4, rec, tot, 512, 1); hipLaunchKernelGGL(k_consume, dim3(2), dim3(512), 0, st, tot, o2, 1024); });
  printf("  one 2 KB record per workgroup (512 values, plain)     %6.2f\n", r);
  for (int values : {256, 512})
    for (int slots : {1, 2, 8}) {
      float a1 = chain_us(st, 20, [&] { hipLaunchKernelGGL((k_stream<1, 1>), dim3(G), dim3(512), 0, st, in, out, n4, rec, tot, values, slots); hipLaunchKernelGGL(k_consume, dim3(2), dim3(512), 0, st, tot, o2, 1024); });
      float a2 = chain_us(st, 20, [&] { hipLaunchKernelGGL((k_stream<1, 2>), dim3(G), dim3(512), 0, st, in, out, n4, rec, tot, values, slots); hipLaunchKernelGGL(k_consume, dim3(2), dim3(512), 0, st, tot, o2, 1024); });
      float a3 = chain_us(st, 20, [&] { hipLaunchKernelGGL((k_stream<1, 3>), dim3(G), dim3(512), 0, st, in, out, n4, rec, tot, values, slots); hipLaunchKernelGGL(k_consume, dim3(2), dim3(512), 0, st, tot, o2, 1024); });
      printf("  %3d values per workgroup, %d slot(s): 1 word %6.2f   2 words %6.2f   3 words %6.2f\n", values, slots, a1, a2, a3);
    }
  return 0;
}
