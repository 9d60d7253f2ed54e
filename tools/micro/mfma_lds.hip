// tools/micro/mfma_lds.hip -- experiment (not product): what fp32-MFMA rate does the chip HOLD when the operands are
// re-read from LDS, by instruction shape and by LDS bytes per MFMA?
//
// The row-local backward kernel's two products run at ~98 TF/s (profiles/r03_rowlocal_rs_experiment.txt: 3.9 GFLOP in
// 44 us, back-to-back v_mfma_f32_16x16x4_f32 with ds_read_b128 one step ahead) against 157 TF/s nominal.  Candidates:
//   v_mfma_f32_16x16x4_f32 : 2 048 flop, 32 cyc/SIMD; a wave that owns 16 rows x 128 columns reads 9 operand vectors per
//                            32 MFMAs (A once, B per column block)
//   v_mfma_f32_32x32x2_f32 : 4 096 flop, 64 cyc/SIMD; 32 rows x 64 columns: 3 vectors per 8 MFMAs (= 0.67 of the LDS bytes
//                            per flop); 32 rows x 128 columns (64 accumulator registers): 5 per 16 (0.56)
// Each lane reads its own 16 bytes (conflict-free by construction; the VALUES are irrelevant to the timing), so the
// kernel isolates instruction shape x LDS read volume x waves per SIMD.  Prints TF/s chip-wide for bursts of ~50 us
// (the length of the real kernels) and ~1 ms.
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// SHAPE 16: 8 accumulator blocks of 16x16 (32 regs); NREAD operand vectors from LDS per 32 MFMAs (0: registers only)
template <int NREAD>
__global__ __launch_bounds__(1024) void k16(int iters, float* out) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 a = *(const f32x4*)&lds[tid * 4], bv[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) bv[b] = *(const f32x4*)&lds[((tid + 64 * b) & 4095) * 4];
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 3) * 4096;
    f32x4 an = a, bn[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) bn[b] = bv[b];
    if (NREAD > 0) an = *(const f32x4*)&lds[o + tid * 4];
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b + 1 < NREAD) bn[b] = *(const f32x4*)&lds[o + ((tid + 64 * (b + 1)) & 4095) * 4];
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

// SHAPE 32: NB accumulator blocks of 32x32 (16 regs each); NB + 1 operand vectors from LDS per 4 * NB MFMAs (LDSFED)
template <int NB, bool LDSFED>
__global__ __launch_bounds__(1024) void k32(int iters, float* out) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 16 * 1024; i += blockDim.x) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
  f32x4 a = *(const f32x4*)&lds[tid * 4], bv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) bv[b] = *(const f32x4*)&lds[((tid + 64 * b) & 4095) * 4];
  for (int it = 0; it < iters; ++it) {
    const int o = (it & 3) * 4096;
    f32x4 an = a, bn[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) bn[b] = bv[b];
    if (LDSFED) {
      an = *(const f32x4*)&lds[o + tid * 4];
#pragma unroll
      for (int b = 0; b < NB; ++b) bn[b] = *(const f32x4*)&lds[o + ((tid + 64 * (b + 1)) & 4095) * 4];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[b][u], acc[b], 0, 0, 0);
    a = an;
#pragma unroll
    for (int b = 0; b < NB; ++b) bv[b] = bn[b];
  }
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[b][e];
  if (s == 123.456f) out[0] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_iter_per_wave, int waves_per_cu) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int target_us : {50, 1000}) {
    // calibrate iters for the target duration at ~100 TF/s
    const double flop_target = 100e12 * target_us * 1e-6;
    int iters = (int)(flop_target / (flop_per_iter_per_wave * waves_per_cu * 256));
    if (iters < 1) iters = 1;
    launch(iters);
    CK(hipDeviceSynchronize());
    const int reps = target_us == 50 ? 40 : 5;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch(iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("  %-44s %2d waves/CU  %7.1f us per launch  %6.1f TF/s\n", name, waves_per_cu, us,
           flop_per_iter_per_wave * waves_per_cu * 256 * iters / (us * 1e-6) / 1e12);
  }
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  const size_t lds_bytes = 16 * 1024 * 4;
  for (int wpc : {8, 16}) {
    const dim3 g(256), b(wpc * 64);
#define L16(NR) [&](int it) { hipLaunchKernelGGL((k16<NR>), g, b, lds_bytes, 0, it, out); }
#define L32(NB, F) [&](int it) { hipLaunchKernelGGL((k32<NB, F>), g, b, lds_bytes, 0, it, out); }
    run("16x16x4, operands in registers", L16(0), 32 * 2048.0, wpc);
    run("16x16x4, 9 LDS vectors / 32 MFMA (288 B/MFMA)", L16(9), 32 * 2048.0, wpc);
    run("16x16x4, 5 LDS vectors / 32 MFMA", L16(5), 32 * 2048.0, wpc);
    run("16x16x4, 2 LDS vectors / 32 MFMA", L16(2), 32 * 2048.0, wpc);
    run("32x32x2, operands in registers (2 blocks)", L32(2, false), 8 * 4096.0, wpc);
    run("32x32x2, 3 LDS vectors / 8 MFMA (2 blocks)", L32(2, true), 8 * 4096.0, wpc);
    if (wpc == 8) {
      run("32x32x2, operands in registers (4 blocks)", L32(4, false), 16 * 4096.0, wpc);
      run("32x32x2, 5 LDS vectors / 16 MFMA (4 blocks)", L32(4, true), 16 * 4096.0, wpc);
    }
  }
  return 0;
}
