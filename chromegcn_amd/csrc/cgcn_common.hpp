// chromegcn_amd/csrc/cgcn_common.hpp -- helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chromegcn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64
#define TILE_NODES 16   // nodes per workgroup tile in the gather kernels
#define BWD_TILE_ROWS 64  // rows per tile of k_bwd_rowlocal
#ifndef BWD_MAX_PARTIALS
#define BWD_MAX_PARTIALS 256
#endif

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

__device__ __forceinline__ int rl_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }


// Counter-based dropout RNG: a mask bit is a pure function of (seed, step counter, stream id, element
// index), so the backward regenerates the forward's mask instead of storing it.  rng_state lives in
// device memory ([0] = seed, [1] = step counter) so that a captured HIP graph sees a fresh counter on
// every replay.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t dropout_key(const unsigned long long* rng_state, uint32_t stream_id) {
  const unsigned long long seed = rng_state[0], ctr = rng_state[1];
  uint32_t k = mix32((uint32_t)seed ^ 0x9E3779B9u);
  k = mix32(k ^ (uint32_t)(seed >> 32));
  k = mix32(k + (uint32_t)ctr * 0x85EBCA6Bu);
  k = mix32(k ^ (uint32_t)(ctr >> 32) ^ (stream_id * 0xC2B2AE35u));
  return k;
}
__device__ __forceinline__ bool dropout_keep(uint32_t key, uint32_t elem, uint32_t thresh) {
  return mix32(elem * 0x9E3779B1u + key) >= thresh;
}
static inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

#define HEAD_STREAM_ID 0x4845u  // dropout stream of the classifier head ("HE")

// Optional prologue of k_bwd_rowlocal for the LAST gated layer: instead of reading dL/dXn it is
// recomputed per row from the head's backward state (BatchNorm backward + ReLU + dropout mask), which
// removes one launch and one write+read of an [S,n,d] tensor.  dym == nullptr disables it.
struct HeadApply {
  const float* dym;      // [n][d]   d loss / d (mean over strands of the dropped BatchNorm output)
  const float* bnc;      // [S][2][d] per-strand mean(dy), mean(dy * xhat)
  const float* mean;     // [S][d]   batch mean of relu(Xn)
  const float* invstd;   // [S][d]
  const float* bn_w;     // [d]
  const unsigned long long* rng_state;
  float keep_scale;
  uint32_t thresh;
  int S;
};

static inline bool misaligned16(const void* p) { return ((uintptr_t)p & 15u) != 0; }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? CGCN_OK : CGCN_ERR_LAUNCH; }
