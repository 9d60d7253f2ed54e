// layer_tanh (cgcn_common.hpp) against the device library's tanhf (bit for bit) and against double-precision tanh (ulp),
// over every 2^-? step of the float line that matters: all floats with |x| in [2^-30, 128) in steps of 64 ulps, plus the
// neighbourhood of the path switch at 0.625.   hipcc -O3 --offload-arch=gfx950 -I include -I chromegcn_amd/csrc
// tools/micro/tanh_check.hip -o /tmp/tanh_check && /tmp/tanh_check
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "cgcn_common.hpp"
__global__ void k(const float* x, float* a, float* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = layer_tanh(x[i]); b[i] = tanhf(x[i]); }
}
int main() {
  std::vector<float> xs;
  for (uint32_t bits = 0x30800000u; bits < 0x43000000u; bits += 64) { float v; std::memcpy(&v, &bits, 4); xs.push_back(v); xs.push_back(-v); }
  for (uint32_t bits = 0x3f200000u - 4096; bits < 0x3f200000u + 4096; ++bits) { float v; std::memcpy(&v, &bits, 4); xs.push_back(v); xs.push_back(-v); }
  const float sp[] = {0.f, -0.f, 1e-40f, -1e-40f, 200.f, -200.f, INFINITY, -INFINITY, 88.7f, 44.4f, 44.36f, NAN};
  for (float v : sp) xs.push_back(v);
  const int n = (int)xs.size();
  float *dx, *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
  hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, n);
  std::vector<float> a(n), b(n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  long long diff = 0; double worst_ulp = 0, worst_abs = 0; float wx = 0;
  for (int i = 0; i < n; ++i) {
    if (std::memcmp(&a[i], &b[i], 4) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) { if (diff < 5) printf("differs: x=%a mine=%a tanhf=%a\n", xs[i], a[i], b[i]); ++diff; }
    if (std::isfinite(xs[i])) {
      const double t = std::tanh((double)xs[i]);
      const double ulp = std::ldexp(1.0, std::ilogb(std::fabs(t) > 1e-300 ? t : 1e-300) - 23);
      const double e = std::fabs((double)a[i] - t);
      if (e / ulp > worst_ulp) { worst_ulp = e / ulp; wx = xs[i]; }
      if (e > worst_abs) worst_abs = e;
    }
  }
  printf("n=%d  bit differences vs tanhf: %lld  worst error vs double tanh: %.2f ulp (at x=%g), %.3g absolute\n", n, diff, worst_ulp, wx, worst_abs);
  return diff ? 1 : 0;
}
