// tools/micro/launch_floor.hip -- experiment (not product): what one dependent kernel launch costs inside a captured HIP
// graph on this runtime, by launch shape.  A chain of N empty kernels is captured on one stream and replayed; the time per
// node is the launch floor (dispatch + completion + the dependency to the next node).  The genome epoch has 176 dependent
// launches; rocprofv3 shows 4.6 us for k_head_bn_finalize / k_head_train_finish with their bodies removed
// (profiles/r04_launch_floor.txt).  Shapes: workgroups x threads, static LDS, with / without a touch of memory.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/launch_floor.hip -o build/launch_floor && ./build/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int LDS_FLOATS>
__global__ void k_empty(float* p, int touch) {
  __shared__ float s[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
  if (LDS_FLOATS > 0 && touch == 12345) s[threadIdx.x % (LDS_FLOATS > 0 ? LDS_FLOATS : 1)] = 1.f;   // keep the allocation
  if (touch == 1 && threadIdx.x == 0) p[blockIdx.x] = (float)blockIdx.x;                              // one 4-byte store per workgroup
  if (LDS_FLOATS > 0 && touch == 12345) p[0] = s[0];
}

// a kernel that leaves `bytes` dirty (plain stores), to see what the next boundary pays for them
__global__ void k_dirty(float* p, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4* q = (float4*)p;
  for (size_t k = i; k < n4; k += stride) q[k] = make_float4(1.f, 2.f, 3.f, 4.f);
}

__global__ void k_nt(float* p, size_t n4) {   // non-temporal stores
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4* q = (f32x4*)p;
  for (size_t k = i; k < n4; k += stride) __builtin_nontemporal_store((f32x4){1.f, 2.f, 3.f, 4.f}, &q[k]);
}

// write-through stores (sc1: the line leaves the XCD's L2 with the store instead of staying dirty in it) -- round 5
template <int MODE>   // 1: sc1, 2: sc0 sc1, 3: sc1 nt
__global__ void k_wt(float* p, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4* q = (f32x4*)p;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t k = i; k < n4; k += stride) {
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&q[k]), "v"(v) : "memory");
    else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&q[k]), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(&q[k]), "v"(v) : "memory");
  }
}

__global__ void k_read(const float* p, size_t n4, float* out) {   // workgroups >= 16 stream n4 float4 and keep a sum
  if (blockIdx.x < 16) return;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const size_t i = (size_t)(blockIdx.x - 16) * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)(gridDim.x - 16) * blockDim.x;
  const f32x4* q = (const f32x4*)p;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (size_t k = i; k < n4; k += stride) a += __builtin_nontemporal_load(&q[k]);
  if (a[0] + a[1] + a[2] + a[3] == 12345.678f) out[0] = 1.f;
}

__global__ void k_dirty2(float* p, size_t n4) {   // the same work as k_dirty under another name
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4* q = (float4*)p;
  for (size_t k = i; k < n4; k += stride) q[k] = make_float4(4.f, 3.f, 2.f, 1.f);
}

template <typename L>
static float graph_us_per_node(L&& enqueue, int nodes, int reps) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) enqueue(st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  CK(hipStreamDestroy(st));
  return ms * 1e3f / (reps * nodes);
}

int main() {
  float* buf;
  const size_t bytes = 64u << 20;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 0, bytes));
  const int nodes = 200, reps = 20;
  printf("chain of %d dependent launches in one captured graph, %d replays: us per launch\n", nodes, reps);
  struct Shape { int grid, block; };
  const Shape shapes[] = {{1, 64}, {1, 1024}, {16, 1024}, {9, 512}, {256, 64}, {256, 512}, {256, 1024}, {1024, 512}, {2048, 512}, {4096, 256}};
  for (const Shape& s : shapes) {
    const float a = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_empty<0>, dim3(s.grid), dim3(s.block), 0, st, buf, 0); }, nodes, reps);
    const float b = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_empty<0>, dim3(s.grid), dim3(s.block), 0, st, buf, 1); }, nodes, reps);
    const float c = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_empty<16000>, dim3(s.grid), dim3(s.block), 0, st, buf, 0); }, nodes, reps);
    printf("  %5d workgroups x %4d threads: empty %5.2f   one store per workgroup %5.2f   64 KB static LDS %5.2f\n", s.grid, s.block, a, b, c);
  }
  {   // does it matter that consecutive launches are DIFFERENT kernels?  (every boundary of the epoch is one)
    const float same = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0); }, nodes, reps);
    int flip = 0;
    const float alt = graph_us_per_node([&](hipStream_t st) {
      if (flip++ & 1) hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0);
      else hipLaunchKernelGGL(k_empty<4>, dim3(16), dim3(1024), 0, st, buf, 0);
    }, nodes, reps);
    int flip2 = 0;
    const float altw = graph_us_per_node([&](hipStream_t st) {
      if (flip2++ & 1) hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, (size_t)(1u << 20) / 16);
      else hipLaunchKernelGGL(k_dirty2, dim3(1024), dim3(256), 0, st, buf, (size_t)(1u << 20) / 16);
    }, nodes, reps);
    const float samew = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, (size_t)(1u << 20) / 16); }, nodes, reps);
    printf("  16 x 1024 empty: same kernel %5.2f, two kernels alternating %5.2f;  1 MB writers: same kernel %5.2f, two writers alternating %5.2f us per launch\n", same, alt, samew, altw);
  }
  // what a boundary pays for dirty bytes left by the predecessor: pairs (writer of B bytes, empty kernel), per pair minus the writer alone
  for (size_t mb : {1u, 4u, 16u, 32u, 64u}) {
    const size_t n4 = (mb << 20) / 16;
    const float w = graph_us_per_node([&](hipStream_t st) { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, n4); }, nodes, reps);
    const float p = graph_us_per_node([&](hipStream_t st) {
      hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, n4);
      hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0);
    }, nodes / 2, reps) * 2.f;
    printf("  writer of %2zu MB: %6.2f us per launch back to back; writer + empty 16 x 1024 kernel: %6.2f us per pair (the empty one adds %5.2f)\n", mb, w, p, p - w);
  }
  // round 5: the same pairs with WRITE-THROUGH stores in the writer (nothing stays dirty in the L2s at its end)
  for (size_t mb : {16u, 32u}) {
    const size_t n4 = (mb << 20) / 16;
    auto E = [&](hipStream_t st) { hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0); };
    auto pair = [&](auto W) {
      const float w = graph_us_per_node([&](hipStream_t st) { W(st); }, nodes, reps);
      const float p = graph_us_per_node([&](hipStream_t st) { W(st); E(st); }, nodes / 2, reps) * 2.f;
      return std::make_pair(w, p);
    };
    auto r0 = pair([&](hipStream_t st) { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, n4); });
    auto rn = pair([&](hipStream_t st) { hipLaunchKernelGGL(k_nt, dim3(1024), dim3(256), 0, st, buf, n4); });
    auto r1 = pair([&](hipStream_t st) { hipLaunchKernelGGL(k_wt<1>, dim3(1024), dim3(256), 0, st, buf, n4); });
    auto r2 = pair([&](hipStream_t st) { hipLaunchKernelGGL(k_wt<2>, dim3(1024), dim3(256), 0, st, buf, n4); });
    auto r3 = pair([&](hipStream_t st) { hipLaunchKernelGGL(k_wt<3>, dim3(1024), dim3(256), 0, st, buf, n4); });
    printf("  %2zu MB writer alone / writer + empty 16 x 1024 (us):  plain %5.2f / %5.2f   nt %5.2f / %5.2f   sc1 %5.2f / %5.2f   sc0 sc1 %5.2f / %5.2f   sc1 nt %5.2f / %5.2f\n",
           mb, r0.first, r0.second, rn.first, rn.second, r1.first, r1.second, r2.first, r2.second, r3.first, r3.second);
  }
  {   // does the ORDER matter?  W = 32 MB of plain stores (k_layer_dense-like), N = 16 MB of non-temporal stores (k_aggregate_sliced-
      // like), E = empty 16 x 1024 (k_head_bn_finalize-like).  Per triple: W E N (today's order) against W N E.
    const size_t w4 = (32u << 20) / 16, n4 = (16u << 20) / 16;
    float* buf2 = buf + (32u << 20) / 4;
    auto W = [&](hipStream_t st) { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, w4); };
    auto N = [&](hipStream_t st) { hipLaunchKernelGGL(k_nt, dim3(1024), dim3(256), 0, st, buf2, n4); };
    auto E = [&](hipStream_t st) { hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0); };
    const float wen = graph_us_per_node([&](hipStream_t st) { W(st); E(st); N(st); }, 60, reps) * 3.f;
    const float wne = graph_us_per_node([&](hipStream_t st) { W(st); N(st); E(st); }, 60, reps) * 3.f;
    const float wn = graph_us_per_node([&](hipStream_t st) { W(st); N(st); }, 60, reps) * 2.f;
    printf("  per triple: W E N %6.2f us, W N E %6.2f us (W N alone %6.2f)\n", wen, wne, wn);
  }
  {   // is work that rides in the tiny kernel's launch free while the predecessor drains?  R(B) = a kernel whose workgroups
      // stream B bytes of reads (a stand-in for a share of another chromosome's aggregation) + the 16 empty workgroups
    const size_t w4 = (32u << 20) / 16;
    float* src = buf + (32u << 20) / 4;   // the other half of the buffer
    auto W = [&](hipStream_t st) { hipLaunchKernelGGL(k_dirty, dim3(1024), dim3(256), 0, st, buf, w4); };
    for (size_t mb : {0u, 4u, 8u, 16u, 32u}) {
      const size_t n4 = (mb << 20) / 16;
      auto R = [&](hipStream_t st) { hipLaunchKernelGGL(k_read, dim3(16 + (mb ? 1024 : 0)), dim3(256), 0, st, src, n4, buf + (60u << 20) / 4); };
      const float wr = graph_us_per_node([&](hipStream_t st) { W(st); R(st); }, 100, reps) * 2.f;
      const float rr = graph_us_per_node([&](hipStream_t st) { R(st); }, 100, reps);
      printf("  W(32 MB) then [16 empty workgroups + reads of %2zu MB]: %6.2f us per pair; the read kernel alone, back to back: %6.2f us\n", mb, wr, rr);
    }
  }
  // eager (no graph) for comparison
  {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL(k_empty<0>, dim3(16), dim3(1024), 0, st, buf, 0);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  eager stream, 16 x 1024 empty: %5.2f us per launch\n", ms * 1e3f / 4000);
  }
  return 0;
}
