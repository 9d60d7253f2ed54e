// bf16x6_probe (round 6): can the fp32 products of the row-local kernels run on the bf16 matrix cores at fp32 accuracy?
//   x = h + m + l EXACTLY (three bf16 numbers: the float's 24 significant bits cut 8 | 8 | 8 by truncation), and
//   a * b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)        (6 of the 9 partial products; the three dropped
//   ones are <= 2^-23 |a b| together) on v_mfma_f32_16x16x32_bf16 with fp32 accumulators.
// Part A: error of a 16 x 16 x K product against float64 for: the fp32 MFMA chain (what the library ships), bf16x6 into one
//         accumulator (small terms first), bf16x6 into three accumulators by magnitude, bf16x9, bf16x3.
// Part B: time of the product phase of a k_layer_dense-shaped loop (LDS tile -> MFMA -> vector filler, 16 waves per CU).
// Build: hipcc -O3 --offload-arch=gfx950 -o bf16x6_probe tools/micro/bf16x6_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__device__ __forceinline__ uint32_t pack_hi(float x0, float x1) {   // (bf16 trunc x1) << 16 | (bf16 trunc x0)
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float hi_part(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// eight floats -> the three bf16x8 levels
__device__ __forceinline__ void split8(const float* x, bf16x8& h, bf16x8& m, bf16x8& l) {
  u32x4 H, M, L;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a0 = x[2 * p], a1 = x[2 * p + 1];
    const float r0 = a0 - hi_part(a0), r1 = a1 - hi_part(a1);
    const float s0 = r0 - hi_part(r0), s1 = r1 - hi_part(r1);
    H[p] = pack_hi(a0, a1);
    M[p] = pack_hi(r0, r1);
    L[p] = pack_hi(s0, s1);
  }
  h = __builtin_bit_cast(bf16x8, H);
  m = __builtin_bit_cast(bf16x8, M);
  l = __builtin_bit_cast(bf16x8, L);
}

// ---------------------------------------------------------------- part A
// one wave per 16 x 16 output tile; A [tiles][16][K], B [tiles][K][16]; out [variant][tiles][16][16]
template <int K>
__global__ __launch_bounds__(64) void k_numerics(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out,
                                                  int tiles) {
  const int t = blockIdx.x, lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const float* a = A + (size_t)t * 16 * K;
  const float* b = B + (size_t)t * K * 16;
  const size_t vs = (size_t)tiles * 256;
  auto store = [&](int v, f32x4 acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) out[v * vs + (size_t)t * 256 + (4 * q + e) * 16 + r] = acc[e];
  };
  {   // 0: the fp32 chain
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * K + k + q], b[(k + q) * 16 + r], acc, 0, 0, 0);
    store(0, acc);
  }
  f32x4 one = {0.f, 0.f, 0.f, 0.f}, big = one, mid = one, small = one, nine = one, three = one, one_kfirst = one;
  for (int k = 0; k < K; k += 32) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = a[r * K + k + 8 * q + u];
      bv[u] = b[(k + 8 * q + u) * 16 + r];
    }
    bf16x8 ah, am, al, bh, bm, bl;
    split8(av, ah, am, al);
    split8(bv, bh, bm, bl);
    // 1: one accumulator, small terms first inside the K-step
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, one, 0, 0, 0);
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, one, 0, 0, 0);
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, one, 0, 0, 0);
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, one, 0, 0, 0);
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, one, 0, 0, 0);
    one = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, one, 0, 0, 0);
    // 2: three accumulators by magnitude
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, small, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, small, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, mid, 0, 0, 0);
    mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, mid, 0, 0, 0);
    big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, big, 0, 0, 0);
    // 3: all nine
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, nine, 0, 0, 0);
    nine = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, nine, 0, 0, 0);
    // 4: three
    three = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, three, 0, 0, 0);
    three = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, three, 0, 0, 0);
    three = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, three, 0, 0, 0);
  }
  store(1, one);
#pragma unroll
  for (int e = 0; e < 4; ++e) big[e] += mid[e] + small[e];
  store(2, big);
  store(3, nine);
  store(4, three);
  // 5: one accumulator, level by level over the whole K (all small terms, then all mid terms, then the big ones)
  for (int lvl = 0; lvl < 3; ++lvl)
    for (int k = 0; k < K; k += 32) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        av[u] = a[r * K + k + 8 * q + u];
        bv[u] = b[(k + 8 * q + u) * 16 + r];
      }
      bf16x8 ah, am, al, bh, bm, bl;
      split8(av, ah, am, al);
      split8(bv, bh, bm, bl);
      if (lvl == 0) {
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, one_kfirst, 0, 0, 0);
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, one_kfirst, 0, 0, 0);
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, one_kfirst, 0, 0, 0);
      } else if (lvl == 1) {
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, one_kfirst, 0, 0, 0);
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, one_kfirst, 0, 0, 0);
      } else {
        one_kfirst = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, one_kfirst, 0, 0, 0);
      }
    }
  store(5, one_kfirst);
}

// ---------------------------------------------------------------- part B
// A k_layer_dense-shaped loop: NW waves, every wave owns NCB blocks of 16 output columns with its B operands resident;
// per tile: wave w writes row(s) of the 16 x K tile into LDS (split into the three levels for MODE 1), barrier, products,
// NF dependent fma per lane as the stand-in for tanh + row pass, barrier.
template <int MODE, int K, int NW, int NCB>
__global__ __launch_bounds__(NW * 64) void k_loop(const float* __restrict__ W, float* __restrict__ out, int tiles, int nf) {
  constexpr int LD = K + 4;          // fp32 tile row pitch (floats)
  constexpr int LDB = K + 8;         // bf16 tile row pitch (elements): 16-byte aligned rows, conflict-light
  __shared__ __attribute__((aligned(16))) float T[16 * LD];
  __shared__ __attribute__((aligned(16))) uint16_t Tb[3][16 * LDB];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  float bw[MODE == 0 ? NCB * K / 4 : 1];
  bf16x8 bh[MODE == 1 ? NCB * K / 32 : 1], bm[MODE == 1 ? NCB * K / 32 : 1], bl[MODE == 1 ? NCB * K / 32 : 1];
  for (int cb = 0; cb < NCB; ++cb) {
    const int j = (wave * NCB + cb) * 16 + r;
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < K / 16; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) bw[cb * K / 4 + 4 * t + u] = W[(size_t)(16 * t + 4 * q + u) * (NW * NCB * 16) + j];
    } else {
#pragma unroll
      for (int s = 0; s < K / 32; ++s) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = W[(size_t)(32 * s + 8 * q + u) * (NW * NCB * 16) + j];
        split8(v, bh[cb * K / 32 + s], bm[cb * K / 32 + s], bl[cb * K / 32 + s]);
      }
    }
  }
  float carry = (float)lane * 1e-3f;
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  for (int tile = 0; tile < tiles; ++tile) {
    // this wave's share of the tile: rows wave, wave + NW, ... ; K floats per row = K / 64 per lane
    for (int row = wave; row < 16; row += NW) {
#pragma unroll
      for (int c = 0; c < K / 256 + (K % 256 ? 1 : 0); ++c) {
        const int k0 = c * 256 + lane * 4;
        if (k0 < K) {
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = carry * (float)(e + 1) + (float)(tile + row) * 0.37f;
          if (MODE == 0) {
            *(f32x4*)&T[row * LD + k0] = (f32x4){x[0], x[1], x[2], x[3]};
          } else {
            const float r0 = x[0] - hi_part(x[0]), r1 = x[1] - hi_part(x[1]), r2 = x[2] - hi_part(x[2]), r3 = x[3] - hi_part(x[3]);
            const float s0 = r0 - hi_part(r0), s1 = r1 - hi_part(r1), s2 = r2 - hi_part(r2), s3 = r3 - hi_part(r3);
            *(uint2*)&Tb[0][row * LDB + k0] = make_uint2(pack_hi(x[0], x[1]), pack_hi(x[2], x[3]));
            *(uint2*)&Tb[1][row * LDB + k0] = make_uint2(pack_hi(r0, r1), pack_hi(r2, r3));
            *(uint2*)&Tb[2][row * LDB + k0] = make_uint2(pack_hi(s0, s1), pack_hi(s2, s3));
          }
        }
      }
    }
    __syncthreads();
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
      const float* Ta = T + r * LD + 4 * q;
#pragma unroll
      for (int t = 0; t < K / 16; ++t) {
        const f32x4 a = *(const f32x4*)&Ta[16 * t];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bw[cb * K / 4 + 4 * t + u], acc[cb], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < K / 32; ++s) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)&Tb[0][r * LDB + 32 * s + 8 * q]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)&Tb[1][r * LDB + 32 * s + 8 * q]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)&Tb[2][r * LDB + 32 * s + 8 * q]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const int i = cb * K / 32 + s;
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[i], acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[i], acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[i], acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[i], acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[i], acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[i], acc[cb], 0, 0, 0);
        }
      }
    }
    // the stand-in for tanh + row pass: nf dependent fma per lane
    float f = acc[0][0];
    for (int i = 0; i < nf; ++i) f = fmaf(f, 0.999f, 1e-3f);
    carry = f * 1e-6f + 0.5f;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) total += acc[cb];
    __syncthreads();
  }
  out[(size_t)blockIdx.x * NW * 64 + threadIdx.x] = total[0] + total[1] + total[2] + total[3] + carry;
}

template <int MODE, int K, int NW, int NCB>
static void run_loop(const char* name, const float* W, float* out, int tiles) {
  for (int nf : {0, 256, 1024}) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_loop<MODE, K, NW, NCB>), dim3(256), dim3(NW * 64), 0, 0, W, out, tiles, nf);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_loop<MODE, K, NW, NCB>), dim3(256), dim3(NW * 64), 0, 0, W, out, tiles, nf);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us_tile = ms * 1e3 / 5 / tiles;
    const double flops = 2.0 * 16 * K * (NW * NCB * 16) * 256.0 * tiles;
    printf("%-34s nf %4d: %.3f us per tile per CU   (%.1f TF/s of useful fp32 product)\n", name, nf, us_tile, flops / (ms * 1e-3 / 5) * 1e-12);
  }
}

int main() {
  // ---------------- part A
  constexpr int K = 256, TILES = 2048;
  std::mt19937_64 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(-0.108f, 0.108f);   // xavier-uniform bound of a 256 x 256 matrix
  std::vector<float> hA((size_t)TILES * 16 * K), hB((size_t)TILES * K * 16);
  for (auto& v : hA) v = nd(rng);
  for (auto& v : hB) v = ud(rng);
  // a second family with a wide dynamic range inside a row (hub rows, large activations)
  for (size_t i = hA.size() / 2; i < hA.size(); ++i) hA[i] *= std::exp(4.f * nd(rng));
  float *dA, *dB, *dO;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dB, hB.size() * 4));
  CK(hipMalloc(&dO, (size_t)6 * TILES * 256 * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((k_numerics<K>), dim3(TILES), dim3(64), 0, 0, dA, dB, dO, TILES);
  CK(hipDeviceSynchronize());
  std::vector<float> hO((size_t)6 * TILES * 256);
  CK(hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost));
  const char* names[6] = {"fp32 MFMA chain (shipped)", "bf16x6, one accumulator", "bf16x6, three accumulators", "bf16x9", "bf16x3",
                          "bf16x6, one acc, levels over K"};
  for (int fam = 0; fam < 2; ++fam) {
    printf("family %d (%s), K = %d, %d tiles: error against float64, in units of 2^-24 * sum_k |a_k b_k|\n", fam,
           fam == 0 ? "N(0,1) x xavier" : "log-normal-scaled rows x xavier", K, TILES / 2);
    double worst[6] = {0}, sumsq[6] = {0};
    size_t cnt = 0, host_chain_equal = 0;
    for (int t = fam * TILES / 2; t < (fam + 1) * TILES / 2; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0, mag = 0;
          for (int k = 0; k < K; ++k) {
            const double p = (double)hA[((size_t)t * 16 + i) * K + k] * (double)hB[((size_t)t * K + k) * 16 + j];
            ref += p;
            mag += std::fabs(p);
          }
          const double unit = mag * std::ldexp(1.0, -24);
          for (int v = 0; v < 6; ++v) {
            const double e = std::fabs((double)hO[((size_t)v * TILES + t) * 256 + i * 16 + j] - ref) / unit;
            worst[v] = std::max(worst[v], e);
            sumsq[v] += e * e;
          }
          ++cnt;
          (void)host_chain_equal;
        }
    for (int v = 0; v < 6; ++v) printf("  %-32s worst %8.3f   rms %7.4f\n", names[v], worst[v], std::sqrt(sumsq[v] / cnt));
  }
  // ---------------- part B
  float *dW, *dOut;
  CK(hipMalloc(&dW, 256 * 512 * 4));
  CK(hipMalloc(&dOut, 256 * 1024 * 4));
  std::vector<float> hW(256 * 512);
  for (auto& v : hW) v = ud(rng);
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  const int tiles = 200;
  printf("product phase of a k_layer_dense-shaped loop, 256 workgroups, %d tiles each\n", tiles);
  run_loop<0, 128, 8, 1>("fp32   K128  8 waves x 16 col", dW, dOut, tiles);
  run_loop<1, 128, 8, 1>("bf16x6 K128  8 waves x 16 col", dW, dOut, tiles);
  run_loop<0, 256, 16, 1>("fp32   K256 16 waves x 16 col", dW, dOut, tiles);
  run_loop<1, 256, 16, 1>("bf16x6 K256 16 waves x 16 col", dW, dOut, tiles);
  run_loop<1, 256, 8, 2>("bf16x6 K256  8 waves x 32 col", dW, dOut, tiles);
  return 0;
}
