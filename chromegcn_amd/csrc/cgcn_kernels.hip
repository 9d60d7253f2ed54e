// chromegcn_amd/csrc/cgcn_kernels.hip
//
// gfx950 (MI355X, CDNA4) kernels for ChromeGCN's gated graph-convolution layer and the
// C ABI declared in include/chromegcn.h.  Written for wave64 / MFMA / 160 KiB LDS; there is
// no other target.
//
// Kernel inventory (DESIGN.md has the roofline for each):
//   k_spmm<S,D>             unfused aggregation  Y = diag(rs) Ahat X, whole-row gather  (tables that fit the L2s)
//   k_aggregate_sliced<S,D> the same aggregation, feature-sliced: one 128-byte column slice per XCD (large tables)
//   k_layer_fwd<S,D>        fused layer forward: gather -> LDS tile -> MFMA (x W) -> bias/tanh/gate/mix epilogue
//   k_layer_dense<S,D>      the row-local half of the layer forward on an H that is in memory (after k_aggregate_sliced)
//   k_bwd_rowlocal_ring     (d = 128) per-row gate/tanh derivative by a row team, H^T dU and dHs = diag(rs) dU W^T on MFMA by a
//                           matrix team, the teams meeting through a flag-synchronised ring of LDS slots (no per-tile barrier)
//   The dense fp32 products of the d = 128 kernels come in two forms (template parameter PROD; cgcn_common.hpp):
//   split products (default) -- six v_mfma_f32_*_bf16 partial products of an exact 3-way split of both fp32 operands, fp32
//   accumulators -- or the fp32 MFMA chain v_mfma_f32_16x16x4_f32 (rounds 1-5; CGCN_PRODUCTS=fp32); d = 256 runs the chain.
//   k_bwd_rowlocal256s      the same work at d = 256: column-slab workgroups, both products per 32-row tile
//   k_reduce_partials       deterministic second stage of the column / dW sums
//   k_bwd_sliced<S,D>       dX = mask ((1-g) dXn + Ahat^T dHs): feature-sliced gather + element-wise epilogue
//
// Reference semantics: models/SubLayers.py:42-52, models/ChromeModels.py:34-46 (forward);
// SURVEY.md Appendix A (backward).
#include <atomic>
#include <type_traits>
#include <cstdlib>

#include "cgcn_common.hpp"

// Geometry of one node's payload (S strands x D features) over a 64-lane wave of float4 loads.
//   PAY = S*D floats.  PAY >= 256: NV = PAY/256 float4 per lane per neighbour.
//   PAY == 128: the two half-waves take alternate neighbours (HALF) and are summed at the end.
template <int S, int D>
struct Geo {
  static constexpr int PAY = S * D;
  static constexpr bool HALF = (PAY == 128);
  static constexpr int NV = HALF ? 1 : PAY / 256;
  static_assert(PAY == 128 || PAY == 256 || PAY == 512, "unsupported payload");
  // strand / column of float4 slot v on this lane
  __device__ static __forceinline__ int strand(int v, int lane) { return HALF ? 0 : ((v * 64 + lane) * 4) / D; }
  __device__ static __forceinline__ int column(int v, int lane) { return HALF ? (lane & 31) * 4 : ((v * 64 + lane) * 4) % D; }
};

// Sum val[k] * X[s, col[k], :] over k in [k0, k1) for one output node; the whole wave cooperates.
// lane_off[v]: this lane's byte offset (strand * n_cols * D + column) * 4 into X.
// Column indices of up to 64 neighbours are read with one coalesced load and broadcast to the scalar
// unit with v_readlane.  Neighbour rows are fetched GU wave-loads at a time, all issued before the first
// add.  The code is branch-free inside a 64-neighbour chunk: a ragged tail re-reads the row's last valid
// neighbour (an L1 hit) and adds a selected zero, so it never degenerates into a serial
// load-wait-add chain.  GATHER_DB = 1 additionally double-buffers the batches.
// Batch depth (GU wave-loads in flight per wave) is a property of the payload and of where the rows come from,
// measured on MI355X (whole chr21-like step, ms): the sweet spot is about 2 KiB in flight per wave with as many
// resident waves as the registers then allow (66 instead of 86 VGPRs: three instead of two workgroups per CU, so
// 722 tiles run as ONE round):  S*D = 256 floats: GU 1 / 2 / 3 / 8 = .249 / .225 / .229 / .236;  S*D = 512
// (d = 256, L = 4): GU 1 / 2 / 3 / 8 = .887 / 1.04 / 1.08 / 1.13.  When the feature table is much larger than the
// L2s (chr1-like, 30 MB) the longer miss latency wants one more load in flight: GU 2 / 3 / 8 = .670 / .647 / .668.
// S*D = 128 (one strand, two neighbours per wave-load): layer forward 26.5 / 25.4 / 27.1 us at GU 3 / 4 / 6.
#ifndef GU_OVERRIDE
#define GU_OVERRIDE 0  // tuning: force one depth everywhere
#endif
// Z and H are written for the backward only.  Non-temporal stores for them (NT_SAVED=1) were measured neutral
// for the kernel (34.3 vs 34.5 us) and slightly negative for the whole step, so plain stores are the default.
#ifndef NT_SAVED
#define NT_SAVED 0
#endif
#if NT_SAVED
#define NT_STORE4(p, v) __builtin_nontemporal_store((v), (f32x4*)(p))
#define NT_STORE1(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NT_STORE4(p, v) (*(f32x4*)(p) = (v))
#define NT_STORE1(p, v) (*(p) = (v))
#endif
#ifndef CBW128
#define CBW128 1   // column blocks per wave at D = 128: 1 -> 8-wave workgroups, 2 -> 4-wave
#endif
#ifndef GATHER_DB
#define GATHER_DB 0
#endif
template <int S, int D, bool HAS_VAL, bool DEEP>
struct Gather {
  using G = Geo<S, D>;
  static constexpr int NV = G::NV;
  static constexpr int GU = GU_OVERRIDE ? GU_OVERRIDE : (G::HALF ? 4 : (NV == 2 ? 1 : (DEEP ? 3 : 2)));
  static constexpr int NPL = G::HALF ? 2 : 1;   // neighbours per wave-load
  static constexpr unsigned ROWB = D * 4;       // bytes per (strand,node) row

  // issue the GU wave-loads of batch b (neighbour slots b*GU .. b*GU+GU-1, clamped to the chunk)
  static __device__ __forceinline__ void issue(f32x4 (&t)[GU][NV], float (&w)[GU], int b, int cnt, int myc, float myv,
                                               const char* __restrict__ Xb, const unsigned (&lane_off)[NV], int lane) {
    const int last = cnt - 1;
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int slot = b * GU + u;
      if (!G::HALF) {
        const int idx = min(slot, last);
        const char* rowp = Xb + (size_t)(unsigned)rl_i(myc, idx) * ROWB;
#pragma unroll
        for (int v = 0; v < NV; ++v) t[u][v] = *(const f32x4*)(rowp + lane_off[v]);
        if (HAS_VAL) w[u] = rl_f(myv, idx);
      } else {
        const int sub = lane >> 5;
        const int i0 = min(2 * slot, last), i1 = min(2 * slot + 1, last);
        const unsigned c = sub ? (unsigned)rl_i(myc, i1) : (unsigned)rl_i(myc, i0);
        if (HAS_VAL) { const float wa = rl_f(myv, i0), wb = rl_f(myv, i1); w[u] = sub ? wb : wa; }
        t[u][0] = *(const f32x4*)(Xb + (size_t)c * ROWB + lane_off[0]);
      }
    }
  }
  static __device__ __forceinline__ void consume(const f32x4 (&t)[GU][NV], const float (&w)[GU], int b, int cnt,
                                                 f32x4 (&acc)[NV], int lane) {
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int slot = b * GU + u;
      const bool ok = G::HALF ? (2 * slot + (lane >> 5) < cnt) : (slot < cnt);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4 x = HAS_VAL ? w[u] * t[u][v] : t[u][v];
        acc[v] += ok ? x : zero;
      }
    }
  }
};

// accumulate neighbours k0..k1 of one row into acc (no initialisation, no half-wave fold)
template <int S, int D, bool HAS_VAL, bool DEEP>
__device__ __forceinline__ void gather_range(const int* __restrict__ col, const float* __restrict__ val,
                                             int k0, int k1, const char* __restrict__ Xb,
                                             const unsigned (&lane_off)[Geo<S, D>::NV], f32x4 (&acc)[Geo<S, D>::NV],
                                             int lane) {
  using GA = Gather<S, D, HAS_VAL, DEEP>;
  constexpr int NV = GA::NV;
  constexpr int GU = GA::GU;
  for (int kb = k0; kb < k1; kb += WAVE) {
    const int cnt = min(WAVE, k1 - kb);
    int myc = 0;
    float myv = 0.f;
    if (lane < cnt) {
      myc = col[kb + lane];
      if (HAS_VAL) myv = val[kb + lane];
    }
    const int nbat = (cnt + GU * GA::NPL - 1) / (GU * GA::NPL);
#if GATHER_DB
    f32x4 ta[GU][NV], tb[GU][NV];
    float wa[GU], wb[GU];
    GA::issue(ta, wa, 0, cnt, myc, myv, Xb, lane_off, lane);
    for (int b = 0; b < nbat; b += 2) {
      GA::issue(tb, wb, b + 1, cnt, myc, myv, Xb, lane_off, lane);   // past the end: clamped re-reads, adds zero
      GA::consume(ta, wa, b, cnt, acc, lane);
      GA::issue(ta, wa, b + 2, cnt, myc, myv, Xb, lane_off, lane);
      GA::consume(tb, wb, b + 1, cnt, acc, lane);
    }
#else
    for (int b = 0; b < nbat; ++b) {
      f32x4 t[GU][NV];
      float w[GU];
      GA::issue(t, w, b, cnt, myc, myv, Xb, lane_off, lane);
      GA::consume(t, w, b, cnt, acc, lane);
    }
#endif
  }
}

template <int S, int D>
__device__ __forceinline__ void gather_fold(int lane, f32x4 (&acc)[Geo<S, D>::NV]) {
  (void)lane;
  if (Geo<S, D>::HALF) {
    // fold the odd-neighbour half onto the even one; afterwards both halves hold the row sum
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] += __shfl_xor(acc[0][e], 32, WAVE);
  }
}

template <int S, int D, bool HAS_VAL, bool DEEP>
__device__ __forceinline__ void gather_node(const int* __restrict__ col, const float* __restrict__ val,
                                            int k0, int k1, const char* __restrict__ Xb,
                                            const unsigned (&lane_off)[Geo<S, D>::NV], f32x4 (&acc)[Geo<S, D>::NV],
                                            int lane) {
#pragma unroll
  for (int v = 0; v < Geo<S, D>::NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  gather_range<S, D, HAS_VAL, DEEP>(col, val, k0, k1, Xb, lane_off, acc, lane);
  gather_fold<S, D>(lane, acc);
}

// Phase 1 of the two gather kernels: aggregate the R nodes of a tile into the LDS tile T (and Hout).
// Ordinary rows: one wave per row.  Rows with more than LONG_ROW neighbours (Hi-C hubs) are split across all
// NW waves of the workgroup in 64-neighbour chunks and combined through LDS in wave order, so one hub row
// costs len/NW instead of len serial batches and the result stays bit-reproducible.
#ifndef LONG_ROW
#define LONG_ROW 512
#endif
template <int S, int D, bool HAS_VAL, bool DEEP, int R, int NW, int LD>
__device__ __forceinline__ void gather_tile(int n, int node0, const int* __restrict__ rowptr,
                                            const int* __restrict__ col, const float* __restrict__ val,
                                            const float* __restrict__ rs, const char* __restrict__ Xb,
                                            const unsigned (&lane_off)[Geo<S, D>::NV], float* __restrict__ T,
                                            float* __restrict__ Hout, float* __restrict__ scratch, int wave, int lane) {
  using G = Geo<S, D>;
  constexpr int NV = G::NV;
  constexpr int PAY = S * D;
  bool any_long = false;
  for (int rr = wave; rr < R; rr += NW) {
    const int i = node0 + rr;
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool is_long = false;
    if (i < n) {
      const int k0 = rowptr[i], k1 = rowptr[i + 1];
      is_long = (k1 - k0) > LONG_ROW;
      if (!is_long) {
        gather_range<S, D, HAS_VAL, DEEP>(col, val, k0, k1, Xb, lane_off, acc, lane);
        gather_fold<S, D>(lane, acc);
        const float sc = rs ? rs[i] : 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] *= sc;
      }
    }
    any_long |= is_long;
    if (!is_long && (!G::HALF || lane < 32)) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int s = G::strand(v, lane), c = G::column(v, lane);
        *(f32x4*)&T[(s * R + rr) * LD + c] = acc[v];
        if (Hout && i < n) NT_STORE4(&Hout[((size_t)s * n + i) * D + c], acc[v]);
      }
    }
  }
  if (!__syncthreads_or(any_long)) return;  // block-uniform: no hub row in this tile
  for (int rr = 0; rr < R; ++rr) {
    const int i = node0 + rr;
    if (i >= n) break;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    if (k1 - k0 <= LONG_ROW) continue;  // same decision in every wave
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kb = k0 + WAVE * wave; kb < k1; kb += WAVE * NW)
      gather_range<S, D, HAS_VAL, DEEP>(col, val, kb, min(kb + WAVE, k1), Xb, lane_off, acc, lane);
    gather_fold<S, D>(lane, acc);
    if (!G::HALF || lane < 32) {
#pragma unroll
      for (int v = 0; v < NV; ++v) *(f32x4*)&scratch[wave * PAY + G::strand(v, lane) * D + G::column(v, lane)] = acc[v];
    }
    __syncthreads();
    if (wave == rr % NW && (!G::HALF || lane < 32)) {
      const float sc = rs ? rs[i] : 1.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int s = G::strand(v, lane), c = G::column(v, lane);
        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < NW; ++w) t += *(const f32x4*)&scratch[w * PAY + s * D + c];
        t *= sc;
        *(f32x4*)&T[(s * R + rr) * LD + c] = t;
        if (Hout) NT_STORE4(&Hout[((size_t)s * n + i) * D + c], t);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// k_spmm: one wave per output node, grid-stride over nodes.
// ------------------------------------------------------------------------------------------
template <int S, int D, bool HAS_VAL, bool DEEP>
__global__ __launch_bounds__(256) void k_spmm(int n_rows, int n_cols, const int* __restrict__ rowptr,
                                              const int* __restrict__ col, const float* __restrict__ val,
                                              const float* __restrict__ rs, const float* __restrict__ X,
                                              float* __restrict__ Y) {
  using G = Geo<S, D>;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  unsigned lane_off[G::NV];
#pragma unroll
  for (int v = 0; v < G::NV; ++v)
    lane_off[v] = ((unsigned)G::strand(v, lane) * (unsigned)n_cols * D + G::column(v, lane)) * 4u;
  for (int i = wave; i < n_rows; i += nwaves) {
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    f32x4 acc[G::NV];
    gather_node<S, D, HAS_VAL, DEEP>(col, val, k0, k1, (const char*)X, lane_off, acc, lane);
    const float sc = rs ? rs[i] : 1.f;
    if (!G::HALF || lane < 32) {
#pragma unroll
      for (int v = 0; v < G::NV; ++v) {
        float* dst = Y + ((size_t)G::strand(v, lane) * n_rows + i) * D + G::column(v, lane);
        *(f32x4*)dst = acc[v] * sc;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_spmm_any: the same aggregation for ANY feature width d that is a multiple of 4 (d <= 4096) -- the
// GraphConvolution drop-in accepts arbitrary in/out widths (models/SubLayers.py:8-12), only the fused gated
// layer is specialised to d = 128 / 256.  One wave per (strand, node) row; lane l owns the float4 column
// groups l, l + 64, ...; column indices are broadcast with v_readlane, four neighbour rows are in flight per wave.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_spmm_any(int n_rows, int n_cols, int S, int d, const int* __restrict__ rowptr,
                                                  const int* __restrict__ col, const float* __restrict__ val,
                                                  const float* __restrict__ rs, const float* __restrict__ X,
                                                  float* __restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int d4 = d >> 2;
  for (int row = wave; row < S * n_rows; row += nwaves) {
    const int s = row / n_rows, i = row - s * n_rows;
    const float* Xs = X + (size_t)s * n_cols * d;
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    const float sc = rs ? rs[i] : 1.f;
    for (int c4b = 0; c4b < d4; c4b += WAVE) {   // every lane stays in the loop (the index broadcast needs all 64)
      const int c4 = c4b + lane;
      const bool on = c4 < d4;
      const int c4c = on ? c4 : 0;                // lanes past the row's end re-read column group 0 and drop it
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kb = k0; kb < k1; kb += WAVE) {
        const int cnt = min(WAVE, k1 - kb);
        const int myc = lane < cnt ? col[kb + lane] : 0;
        const float myv = (val && lane < cnt) ? val[kb + lane] : 1.f;
        for (int j = 0; j < cnt; j += 4) {
          f32x4 t[4];
          float w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = min(j + u, cnt - 1);
            const int cj = rl_i(myc, idx);
            w[u] = rl_f(myv, idx);
            t[u] = *(const f32x4*)&Xs[(size_t)cj * d + c4c * 4];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (j + u < cnt) acc += w[u] * t[u];
        }
      }
      if (on) *(f32x4*)&Y[((size_t)s * n_rows + i) * d + c4 * 4] = acc * sc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Shared pieces of the fused kernels (k_layer_fwd; round 1: also the fused backward gather).
//
// Tile: TN = 16*MB/S nodes x S strands = 16*MB MFMA rows, staged in LDS as T[row][D+4].
// MFMA: v_mfma_f32_16x16x4_f32.  A lane l: A[row l&15][k-slot l>>4]; B lane l: B[k-slot l>>4][col l&15];
// C: col l&15, row 4*(l>>4)+reg.  The K index is permuted so that operand reads are 16 bytes wide and the
// four k-slots of one step read 64 contiguous bytes of a row: step (t,u) of k-slot q (= l>>4) uses
// k = 16t + 4q + u, for A and B alike.
// Wave w owns output columns [32w, 32w+32) as two 16-wide blocks.
//   TRANS_W == false: B(k,j) = W[k][j]   (forward,  U = H W)
//   TRANS_W == true : B(k,j) = W[j][k]   (backward, dS W^T)
// For D = 128 a wave's whole B operand (2 x 32 floats per lane) is fetched into registers BEFORE the
// gather phase, so its L2 latency hides behind the gather; for D = 256 it is read in the K loop.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int xcd_contiguous(int b, int nblk) {
  // Workgroups are dealt round-robin over the 8 XCDs; give each XCD (each private L2) a contiguous
  // range of tiles so that near-diagonal Hi-C neighbourhoods stay L2-resident.  Bijective for any nblk.
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int D, int CBW, bool TRANS_W>
__device__ __forceinline__ void load_wfrag(const float* __restrict__ W, int wave, int lane, float (&bw)[CBW][D / 4]) {
  constexpr int KQ = D / 4;
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) {
    const int j = wave * (16 * CBW) + cb * 16 + r;
    if (TRANS_W) {
#pragma unroll
      for (int t = 0; t < KQ / 4; ++t) {
        const f32x4 v = *(const f32x4*)&W[(size_t)j * D + 16 * t + 4 * q];
#pragma unroll
        for (int u = 0; u < 4; ++u) bw[cb][4 * t + u] = v[u];
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) bw[cb][kk] = W[(size_t)(16 * (kk >> 2) + 4 * q + (kk & 3)) * D + j];
    }
  }
}

template <int MB, int D, int CBW, int LD, bool TRANS_W, bool PRELOADED>
__device__ __forceinline__ void tile_mfma(const float* __restrict__ T, const float* __restrict__ W,
                                          const float (&bw)[CBW][D / 4], int wave, int lane, f32x4 (&acc)[MB][CBW]) {
  constexpr int KQ = D / 4;
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) acc[mb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (PRELOADED) {
#pragma unroll
    for (int t = 0; t < KQ / 4; ++t) {
      f32x4 a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = *(const f32x4*)&T[(mb * 16 + r) * LD + 16 * t + 4 * q];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int cb = 0; cb < CBW; ++cb)
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], bw[cb][4 * t + u], acc[mb][cb], 0, 0, 0);
    }
  } else {
    const int j0 = wave * (16 * CBW) + r;
#pragma unroll 2
    for (int t = 0; t < KQ / 4; ++t) {
      f32x4 a[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[mb] = *(const f32x4*)&T[(mb * 16 + r) * LD + 16 * t + 4 * q];
      f32x4 b[CBW];
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) {
        if (TRANS_W) {
          b[cb] = *(const f32x4*)&W[(size_t)(j0 + 16 * cb) * D + 16 * t + 4 * q];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) b[cb][u] = W[(size_t)(16 * t + 4 * q + u) * D + j0 + 16 * cb];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int cb = 0; cb < CBW; ++cb)
            acc[mb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], b[cb][u], acc[mb][cb], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_layer_fwd: one workgroup = TN nodes x S strands.
//   phase 0  W fragments -> registers (D = 128)
//   phase 1  each wave gathers whole rows of H = diag(rs) Ahat X into the LDS tile (+ Hout)
//   phase 2  U = H W on the matrix cores; the residual rows of X are prefetched meanwhile
//   phase 3  Z = tanh(U + b) -> LDS -> row-wise gate (wave reduction), mix, optional dropout, stores
// ------------------------------------------------------------------------------------------
// 8 waves per SIMD (<= 64 VGPRs; the two-strand 16-row variants need 62-66 as it is): four workgroups per CU
// resident.  Measured with it: chr10-like layer forward 73.4 -> 68.9 us, chr1-like 92.5 -> 87.3 us, chr21-like
// unchanged.  Not for the single-strand geometry (deeper batches, more registers: 25.4 us free vs 29.2 us forced).
#ifndef DENSE_HALF_WAVE_ROWS
#define DENSE_HALF_WAVE_ROWS 1   // d = 128: the row-wise passes of k_layer_fwd / k_layer_dense take a row per half-wave, 16 B per lane
#endif
#ifndef FWD_HALF_WAVE_ROWS
#define FWD_HALF_WAVE_ROWS 1
#endif
#ifndef FWD_WAVES_PER_SIMD
#define FWD_WAVES_PER_SIMD(S_, D_) (((S_) * (D_) == 128) ? 1 : 8)
#endif
#ifndef FWD_OCC
#define FWD_OCC
#endif
#ifdef KT_TIMING  // tuning build only (tools/khead.py --stamps-rowlocal): phase timestamps of a few workgroups
__device__ unsigned long long kt_stamps[8 * 16];
#ifndef KT_STRIDE
#define KT_STRIDE 32
#endif
#define KT_STAMP(i)                                                                      \
  do {                                                                                   \
    __builtin_amdgcn_s_waitcnt(0);                                                       \
    if (threadIdx.x == 0 && (blockIdx.x % KT_STRIDE) == 0 && blockIdx.x < 8 * KT_STRIDE) kt_stamps[(blockIdx.x / KT_STRIDE) * 16 + (i)] = wall_clock64(); \
  } while (0)
#define KT_STAMP_T(i, tid)                                                               \
  do {                                                                                   \
    __builtin_amdgcn_s_waitcnt(0);                                                       \
    if (threadIdx.x == (tid) && (blockIdx.x % KT_STRIDE) == 0 && blockIdx.x < 8 * KT_STRIDE) kt_stamps[(blockIdx.x / KT_STRIDE) * 16 + (i)] = wall_clock64(); \
  } while (0)
extern "C" int cgcn_debug_kt_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kt_stamps), sizeof(unsigned long long) * 8 * 16) == hipSuccess ? 0 : -1;
}
// stamp WITHOUT a vector-memory wait (loads in flight across the stamp stay in flight), period KT_PERIOD
#ifndef KT_PERIOD
#define KT_PERIOD 2
#endif
#define KT_STAMP_NW(i, tid, cond)                                                        \
  do {                                                                                   \
    if ((cond) && threadIdx.x == (tid) && (blockIdx.x % KT_STRIDE) == 0 && blockIdx.x < 8 * KT_STRIDE) { \
      __builtin_amdgcn_sched_barrier(0);                                                 \
      kt_stamps[(blockIdx.x / KT_STRIDE) * 16 + (i)] = wall_clock64();                   \
      __builtin_amdgcn_sched_barrier(0);                                                 \
    }                                                                                    \
  } while (0)
#else
#define KT_STAMP(i)
#define KT_STAMP_T(i, tid)
#define KT_STAMP_NW(i, tid, cond)
#endif

template <int S, int D, int MB, bool HAS_VAL, bool DEEP, int PROD>
__global__ __launch_bounds__((D == 128 && CBW128 == 2) ? 256 : 512, PROD ? 4 : FWD_WAVES_PER_SIMD(S, D)) FWD_OCC void k_layer_fwd(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                     const float* __restrict__ val, const float* __restrict__ rs,
                                                     const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ wg,
                                                     const float* __restrict__ cg, float* __restrict__ Xn,
                                                     float* __restrict__ Zout, float* __restrict__ Hout,
                                                     float* __restrict__ gate, float keep_scale, uint32_t thresh,
                                                     const unsigned long long* __restrict__ rng_state,
                                                     uint32_t stream_id, float* __restrict__ colstats, int stat_acc,
                                                     unsigned long long* __restrict__ zero_words, int zero_count) {
  using G = Geo<S, D>;
  // (statistics totals of a LATER launch of the step -- cgcn_layer_fwd's colstats_rows = -2 -- zeroed here like k_aggregate_sliced does)
  if (zero_words && (int)blockIdx.x < min((int)gridDim.x, 8))
    for (int i = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x; i < zero_count; i += min((int)gridDim.x, 8) * (int)blockDim.x) zero_words[i] = 0ull;
  constexpr int ROWS = 16 * MB;      // MFMA rows in the tile
  constexpr int R = ROWS / S;        // nodes in the tile
  constexpr int CBW = (D == 128) ? CBW128 : 2;  // 16-wide output column blocks per wave
  constexpr int NW = D / (16 * CBW); // 8 waves per workgroup
  constexpr int LD = D + 4;          // LDS row stride (floats); keeps 16-byte alignment
  constexpr int EPL = D / 64;        // floats per lane in the row-wise epilogue
  constexpr int RPW = (ROWS + NW - 1) / NW;  // epilogue rows per wave
  constexpr bool PRE = (D == 128);
  __shared__ __attribute__((aligned(16))) float T[ROWS * LD];
  __shared__ __attribute__((aligned(16))) float LR[NW * S * D];  // hub-row partial sums (gather_tile)
  // PROD == 1 (split products, cgcn_common.hpp; d = 128): the gathered tile is re-staged as three bf16 levels (k_layer_dense's
  // Tb, the same image) and the product is formed exactly as k_layer_dense forms it -- the same bits on both routes.  The W operands are 48
  // registers instead of 32: 4 waves per SIMD (this kernel only serves tables below the split threshold).
  constexpr int LVT = ROWS * 256;   // bytes of one level tile (swizzled image of cgcn_common.hpp, sp_sigma)
  __shared__ __attribute__((aligned(16))) unsigned char Tb[PROD ? 3 * LVT : 16];
  static_assert(PROD == 0 || (D == 128 && MB == 1 && CBW == 1 && FWD_HALF_WAVE_ROWS), "split products: d = 128, the half-wave-row form");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node0 = xcd_contiguous(blockIdx.x, gridDim.x) * R;

  KT_STAMP(8);
  float bw[PROD ? 1 : CBW][PROD ? 1 : D / 4];  // W fragments: loaded after the gather (below)
  bf16x8 wh[PROD ? D / 32 : 1], wm[PROD ? D / 32 : 1], wl[PROD ? D / 32 : 1];

  unsigned lane_off[G::NV];
#pragma unroll
  for (int v = 0; v < G::NV; ++v) lane_off[v] = ((unsigned)G::strand(v, lane) * (unsigned)n * D + G::column(v, lane)) * 4u;

  KT_STAMP(9);
  // ---- phase 1
  gather_tile<S, D, HAS_VAL, DEEP, R, NW, LD>(n, node0, rowptr, col, val, rs, (const char*)X, lane_off, T, Hout, LR, wave, lane);
  // W fragments -> registers once the gather's loads are issued: in front of it they delay the first neighbour rows
  // (vector loads return in order) and hold 32 registers through the gather loop; here they land during the barrier
  // and the residual prefetch (measured: -0.3 ... -1.5 % per step, every workload)
  if constexpr (PROD != 0) {   // B operand of K-step s: W[32 s + 8 q + u][16 wave + r], u < 8, as three levels
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = W[(size_t)(32 * s + 8 * q + u) * D + wave * 16 + r];
      sp_split8(v, wh[s], wm[s], wl[s]);
    }
  } else {
    if (PRE) load_wfrag<D, CBW, false>(W, wave, lane, bw);
  }
  KT_STAMP(10);
  // prefetch the residual rows this wave will mix in phase 3 (latency hides behind the MFMA phase)
  // D = 128: the row-wise epilogue takes a row per HALF-wave, 16 bytes per lane (see k_layer_dense): the wave's two rows
  // of the 16-row tile are 2 wave + {0, 1}
  constexpr bool HW = (D == 128 && MB == 1 && NW == 8 && FWD_HALF_WAVE_ROWS);
  const bool upper = lane >= 32;
  const int l4 = (lane & 31) * 4;
  const int hm = 2 * wave + (upper ? 1 : 0);
  const int hs = hm / R, hr = hm % R;
  f32x4 xres4 = {0.f, 0.f, 0.f, 0.f};
  float xres[RPW][EPL];
  if (HW) {
    if (node0 + hr < n) xres4 = *(const f32x4*)&X[((size_t)hs * n + node0 + hr) * D + l4];
  } else {
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int m = wave + t * NW;
      const int i = node0 + (m % R);
      if (m < ROWS && i < n) ld_row<EPL>(xres[t], &X[((size_t)(m / R) * n + i) * D + lane * EPL]);
      else zero_row<EPL>(xres[t]);
    }
  }
  // the epilogue's small operands too (bias of this lane's columns, gate weights): their latency would otherwise sit
  // between the MFMA phase and the tanh
  float bjv[CBW], wgl[EPL];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) bjv[cb] = bias[wave * (16 * CBW) + cb * 16 + (lane & 15)];
  ld_row<EPL>(wgl, &wg[lane * EPL]);
  const float c0 = cg[0];
  const uint32_t key = thresh ? dropout_key(rng_state, stream_id) : 0u;
  __syncthreads();

  KT_STAMP(11);
  // ---- phase 2
  f32x4 acc[MB][CBW];
  if constexpr (PROD != 0) {
    {   // this half-wave's row of the gathered tile -> the three levels
      u32x2 h2, m2, l2;
      sp_split4(*(const f32x4*)&T[hm * LD + l4], h2, m2, l2);
      unsigned char* w = Tb + hm * 256 + ((((lane & 31) >> 1) ^ sp_sigma(hm)) << 4) + ((lane & 1) << 3);
      *(u32x2*)w = h2;
      *(u32x2*)(w + LVT) = m2;
      *(u32x2*)(w + 2 * LVT) = l2;
    }
    __syncthreads();   // (also: every wave is done reading T, which the tanh tile overwrites)
    const int r = lane & 15, q = lane >> 4;
    const unsigned char* __restrict__ Ta = Tb + r * 256 + ((q ^ (sp_sigma(r) & 3)) << 4);
    const int hi = sp_sigma(r) >> 2;
    SpAcc sa;
    sa.zero();
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      const int o = (s ^ hi) << 6;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + o));
      const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + LVT + o));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + 2 * LVT + o));
      sa.step(ah, am, al, wh[s], wm[s], wl[s]);
    }
    acc[0][0] = sa.sum();
  } else {
    tile_mfma<MB, D, CBW, LD, false, PRE>(T, W, bw, wave, lane, acc);
    __syncthreads();  // every wave is done reading T as the A operand
  }

  KT_STAMP(12);
  // ---- phase 3a: Z = tanh(U + b) back into the tile
  // (the gate weights of the half-wave form are requested here, behind the products: they would cost four registers
  // through the gather and the MFMA phase -- two spilled ones in the deep-batch instantiations at the 64-register cap)
  const f32x4 wgl4 = HW ? *(const f32x4*)&wg[l4] : (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < CBW; ++cb) {
      const int j = wave * (16 * CBW) + cb * 16 + r;
      const float bj = bjv[cb];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[(mb * 16 + q * 4 + e) * LD + j] = layer_tanh(acc[mb][cb][e] + bj);
    }
  }
  __syncthreads();

  KT_STAMP(13);
  // ---- phase 3b: row-wise gate + residual mix, coalesced stores
  if (HW) {
    const int i = node0 + hr;
    const f32x4 z4 = *(const f32x4*)&T[hm * LD + l4];
    float dot = z4[0] * wgl4[0] + z4[1] * wgl4[1] + z4[2] * wgl4[2] + z4[3] * wgl4[3];
    dot = half_sum(dot, upper);
    const float g = sigmoidf_(dot + c0);
    if (i < n) {
      const size_t g_off = ((size_t)hs * n + i) * D + l4;
      f32x4 xo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xo[e] = (1.f - g) * xres4[e] + g * z4[e];
        if (thresh) xo[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? xo[e] * keep_scale : 0.f;
      }
      if (colstats) *(f32x4*)&T[hm * LD + l4] = (f32x4){fmaxf(xo[0], 0.f), fmaxf(xo[1], 0.f), fmaxf(xo[2], 0.f), fmaxf(xo[3], 0.f)};  // own row, already consumed above
      *(f32x4*)&Xn[g_off] = xo;
      if (Zout) *(f32x4*)&Zout[g_off] = z4;
      if ((lane & 31) == 0) gate[(size_t)hs * n + i] = g;
    }
  } else
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int m = wave + t * NW;
    const int s = m / R, rr = m % R;
    const int i = node0 + rr;
    if (m >= ROWS || i >= n) continue;  // wave-uniform
    float z[EPL];
    float dot = 0.f;
    const size_t g_off = ((size_t)s * n + i) * D + lane * EPL;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      z[e] = T[m * LD + lane * EPL + e];
      dot += z[e] * wgl[e];
    }
    dot = wave_sum(dot);
    const float g = sigmoidf_(dot + c0);
    float xo[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      xo[e] = (1.f - g) * xres[t][e] + g * z[e];
      if (thresh) xo[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? xo[e] * keep_scale : 0.f;
      if (colstats) T[m * LD + lane * EPL + e] = fmaxf(xo[e], 0.f);  // own row, already consumed above
    }
    st_row<EPL>(&Xn[g_off], xo);
    if (Zout) st_row<EPL>(&Zout[g_off], z);
    if (lane == 0) gate[(size_t)s * n + i] = g;
  }
  // ---- optional: first stage of the classifier head's BatchNorm statistics (k_head_colstats' job) while the tile
  // is still on chip: per (strand, column) the exact two-pass (mean, M2) of relu(Xn) over this tile's nodes
  if (colstats && stat_acc) {
    // accumulate mode (cgcn_common.hpp, STAT_ACC_*; totals zeroed by an EARLIER launch of the step): this tile's sum x and
    // sum x^2 per (strand, column), the squares in double, straight into the slot of this workgroup's XCD
    __syncthreads();
    const int cnt = min(R, n - node0);
    for (int idx = threadIdx.x; idx < S * D; idx += blockDim.x) {
      const int s = idx / D, c = idx % D;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const double v = rr < cnt ? (double)T[(s * R + rr) * LD + c] : 0.0;
        s1 += v;
        s2 += v * v;
      }
      stat_acc_add((unsigned long long*)colstats, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), s, c, s1, s2,
                   2147483648.0 / (double)gridDim.x);   // (loud, not wrapped, whatever the number of tiles)
    }
  } else if (colstats) {
    __syncthreads();
    const int cnt = min(R, n - node0);
    const float inv = 1.f / (float)cnt;
    float* out = colstats + (size_t)(node0 / R) * S * D * 2;
    for (int idx = threadIdx.x; idx < S * D; idx += blockDim.x) {
      const int s = idx / D, c = idx % D;
      float v[R];
      float sum = 0.f;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        v[rr] = T[(s * R + rr) * LD + c];
        sum += rr < cnt ? v[rr] : 0.f;
      }
      const float mean = sum * inv;
      float m2 = 0.f;
#pragma unroll
      for (int rr = 0; rr < R; ++rr) m2 += rr < cnt ? (v[rr] - mean) * (v[rr] - mean) : 0.f;
      out[idx * 2] = mean;
      out[idx * 2 + 1] = m2;
    }
  }
  KT_STAMP(14);
}

// ------------------------------------------------------------------------------------------
// k_layer_dense: the row-local half of the layer forward for an aggregation H that is already in memory (the split
// path: k_aggregate_sliced wrote it; or the engine's cached A X of the first layer): U = H W + b on MFMA, tanh, gate,
// mix, optional dropout and BatchNorm column statistics -- phases 2 and 3 of k_layer_fwd.  Persistent: a workgroup
// walks node tiles (R nodes x S strands = 16 MB rows) with its W fragments resident in registers (D = 128), so W is
// read once per workgroup instead of once per tile, and several workgroups per CU overlap each other's load, MFMA
// and store phases.  Streams H and X in, Xn, Z and the gate out.
// ------------------------------------------------------------------------------------------
#ifndef DENSE_WAVES_PER_SIMD
#define DENSE_WAVES_PER_SIMD 4
#endif
template <int S, int D, int MB, int PROD>
__global__ __launch_bounds__(512, DENSE_WAVES_PER_SIMD) void k_layer_dense(int n, int ntiles, const float* __restrict__ Hin,
                                                    const float* __restrict__ X, const float* __restrict__ W,
                                                    const float* __restrict__ bias, const float* __restrict__ wg,
                                                    const float* __restrict__ cg, float* __restrict__ Xn,
                                                    float* __restrict__ Zout, float* __restrict__ gate,
                                                    float keep_scale, uint32_t thresh,
                                                    const unsigned long long* __restrict__ rng_state,
                                                    uint32_t stream_id, float* __restrict__ colstats, int stat_chunk,
                                                    int stat_acc, unsigned long long* __restrict__ zero_words, int zero_count) {
  if (zero_words && (int)blockIdx.x < min((int)gridDim.x, 8))   // (totals of a LATER launch of the step: colstats_rows = -2)
    for (int i = (int)blockIdx.x * 512 + (int)threadIdx.x; i < zero_count; i += min((int)gridDim.x, 8) * 512) zero_words[i] = 0ull;
  constexpr int ROWS = 16 * MB;      // MFMA rows in the tile
  constexpr int R = ROWS / S;        // nodes in the tile
  static_assert(D == 128, "d = 256 has its own kernel (k_layer_dense256)");
  constexpr int CBW = 1;             // 16-wide output column blocks per wave
  constexpr int NW = 8;
  constexpr int LD = D + 4;          // LDS row stride (floats); keeps 16-byte alignment
  constexpr int EPL = D / 64;        // floats per lane in the row-wise passes
  constexpr int RPW = ROWS / NW;     // rows per wave
  constexpr bool PRE = true;         // W fragments resident in registers
  static_assert(D / (16 * CBW) == NW && ROWS % NW == 0, "geometry");
  __shared__ __attribute__((aligned(16))) float T[ROWS * LD];
  // PROD == 1 (split products, cgcn_common.hpp): the A operand lives in its own tile as three bf16 levels (the swizzled level-tile
  // image sp_sigma: padded 272-byte rows cost 2 800 bank-conflict cycles per CU and launch on the ds_read_b128 operand reads,
  // whose lane groups are not the contiguous sixteen) -- so T only ever holds the tanh tile and two of the four barriers of a
  // tile go (nobody reads T as an operand, nobody rewrites it early)
  constexpr int LVT = ROWS * 256;
  __shared__ __attribute__((aligned(16))) unsigned char Tb[PROD ? 3 * LVT : 16];
  static_assert(PROD == 0 || (D == 128 && MB == 1 && DENSE_HALF_WAVE_ROWS), "split products: the half-wave-row form");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  KT_STAMP(8);
  float bw[PROD ? 1 : CBW][PROD ? 1 : D / 4];
  bf16x8 wh[PROD ? D / 32 : 1], wm[PROD ? D / 32 : 1], wl[PROD ? D / 32 : 1];
  if constexpr (PROD != 0) {   // B operand of K-step s: W[32 s + 8 q + u][16 wave + r], u < 8, as three levels
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = W[(size_t)(32 * s + 8 * q + u) * D + wave * 16 + r];
      sp_split8(v, wh[s], wm[s], wl[s]);
    }
  } else {
    if (PRE) load_wfrag<D, CBW, false>(W, wave, lane, bw);
  }
  float bjv[CBW], wgl[EPL];
#pragma unroll
  for (int cb = 0; cb < CBW; ++cb) bjv[cb] = bias[wave * (16 * CBW) + cb * 16 + (lane & 15)];
  ld_row<EPL>(wgl, &wg[lane * EPL]);
  const float c0 = cg[0];
  const uint32_t key = thresh ? dropout_key(rng_state, stream_id) : 0u;
  // D = 128 (round 4): the row-wise passes take a row per HALF-wave, 16 bytes per lane -- one wave instruction moves two
  // rows (k_bwd_rowlocal_ring's row team streams the same rows at 6.5 TB/s that way against 5.3 with 8 bytes per lane
  // and a row per wave): half the loads / stores / LDS accesses, one DPP reduction for two rows.  The wave's two rows of
  // a 16-row tile are 2 wave + {0, 1}.
  constexpr bool HW = (D == 128 && MB == 1 && DENSE_HALF_WAVE_ROWS);
  const bool upper = lane >= 32;
  const int l4 = (lane & 31) * 4;
  const int hm = 2 * wave + (upper ? 1 : 0);            // this half-wave's row of the tile
  const int hs = hm / R, hr = hm % R;                   // its strand and its node inside the tile
  const f32x4 wgl4 = HW ? *(const f32x4*)&wg[l4] : (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 hrow4 = {0.f, 0.f, 0.f, 0.f}, xres4 = hrow4, xnext4 = hrow4;
  auto load_rows4 = [&](f32x4& dst, const float* __restrict__ src, int tile) {
    const int i = tile * R + hr;
    dst = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n) dst = *(const f32x4*)&src[((size_t)hs * n + i) * D + l4];
  };

  // rows of the tile this wave streams in / finishes: m = wave + t * NW  ->  (strand m / R, node node0 + m % R).
  // Both input streams run one whole tile ahead: the loads of tile t+1 are issued at the top of tile t, so every
  // resident workgroup keeps a full tile of reads in flight through its MFMA, tanh and store phases.
  float hrow[RPW][EPL], xres[RPW][EPL], xnext[RPW][EPL];
  auto load_rows = [&](float (&dst)[RPW][EPL], const float* __restrict__ src, int tile) {
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int m = wave + t * NW;
      const int i = tile * R + (m % R);
      if (i < n) ld_row<EPL>(dst[t], &src[((size_t)(m / R) * n + i) * D + lane * EPL]);
      else zero_row<EPL>(dst[t]);
    }
  };
  // Tile walk.  Without column statistics: tiles b, b + G, b + 2G, ... (G = grid).  With them: the CONTIGUOUS tiles
  // [b * stat_chunk, (b + 1) * stat_chunk), whose statistics are merged on chip (Chan) into ONE record per workgroup --
  // a record then covers stat_chunk * R consecutive nodes, which is all the head's finalize kernel needs to know
  // (cgcn_layer_fwd_colstats_plan), and there are <= 512 of them instead of one per 16 / S nodes.
  const int tfirst = colstats ? (int)blockIdx.x * stat_chunk : (int)blockIdx.x;
  const int tstep = colstats ? 1 : (int)gridDim.x;
  const int tend = colstats ? min(ntiles, tfirst + stat_chunk) : ntiles;
  float st_cnt = 0.f, st_mean = 0.f, st_m2 = 0.f;   // running statistics of this thread's (strand, column)
  if (tfirst < tend) {
    if (HW) {
      load_rows4(hrow4, Hin, tfirst);
      load_rows4(xnext4, X, tfirst);
    } else {
      load_rows(hrow, Hin, tfirst);
      load_rows(xnext, X, tfirst);
    }
  }
  KT_STAMP(9);
  for (int tile = tfirst; tile < tend; tile += tstep) {
    const int node0 = tile * R;
    KT_STAMP(10);
    if (HW) {
      if (PROD) {
        u32x2 h2, m2, l2;
        sp_split4(hrow4, h2, m2, l2);
        unsigned char* w = Tb + hm * 256 + ((((lane & 31) >> 1) ^ sp_sigma(hm)) << 4) + ((lane & 1) << 3);
        *(u32x2*)w = h2;
        *(u32x2*)(w + LVT) = m2;
        *(u32x2*)(w + 2 * LVT) = l2;
      } else {
        *(f32x4*)&T[hm * LD + l4] = hrow4;
      }
      xres4 = xnext4;
      if (tile + tstep < tend) {
        load_rows4(hrow4, Hin, tile + tstep);
        load_rows4(xnext4, X, tile + tstep);
      }
    } else {
#pragma unroll
      for (int t = 0; t < RPW; ++t) {
        st_row<EPL>(&T[(wave + t * NW) * LD + lane * EPL], hrow[t]);
#pragma unroll
        for (int e = 0; e < EPL; ++e) xres[t][e] = xnext[t][e];
      }
      if (tile + tstep < tend) {
        load_rows(hrow, Hin, tile + tstep);
        load_rows(xnext, X, tile + tstep);
      }
    }
    __syncthreads();
    KT_STAMP(11);
    // ---- U = H W
    f32x4 acc[MB][CBW];
    if constexpr (PROD != 0) {
      const int r = lane & 15, q = lane >> 4;
      const unsigned char* __restrict__ Ta = Tb + r * 256 + ((q ^ (sp_sigma(r) & 3)) << 4);
      const int hi = sp_sigma(r) >> 2;
      SpAcc sa;
      sa.zero();
#pragma unroll
      for (int s = 0; s < D / 32; ++s) {
        const int o = (s ^ hi) << 6;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + o));
        const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + LVT + o));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ta + 2 * LVT + o));
        sa.step(ah, am, al, wh[s], wm[s], wl[s]);
      }
      acc[0][0] = sa.sum();
    } else {
      tile_mfma<MB, D, CBW, LD, false, PRE>(T, W, bw, wave, lane, acc);
      __syncthreads();  // every wave is done reading T as the A operand
    }
    KT_STAMP(12);
    // ---- Z = tanh(U + b) back into the tile
    {
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int cb = 0; cb < CBW; ++cb) {
        const int j = wave * (16 * CBW) + cb * 16 + r;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) T[(mb * 16 + q * 4 + e) * LD + j] = layer_tanh(acc[mb][cb][e] + bjv[cb]);
      }
    }
    __syncthreads();
    KT_STAMP(13);
    // ---- row-wise gate + residual mix, coalesced stores
    if (HW) {
      const int i = node0 + hr;
      const f32x4 z4 = *(const f32x4*)&T[hm * LD + l4];
      float dot = z4[0] * wgl4[0] + z4[1] * wgl4[1] + z4[2] * wgl4[2] + z4[3] * wgl4[3];
      dot = half_sum(dot, upper);
      const float g = sigmoidf_(dot + c0);
      if (i < n) {
        const size_t g_off = ((size_t)hs * n + i) * D + l4;
        f32x4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xo[e] = (1.f - g) * xres4[e] + g * z4[e];
          if (thresh) xo[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? xo[e] * keep_scale : 0.f;
        }
        if (colstats) *(f32x4*)&T[hm * LD + l4] = (f32x4){fmaxf(xo[0], 0.f), fmaxf(xo[1], 0.f), fmaxf(xo[2], 0.f), fmaxf(xo[3], 0.f)};  // own row, already consumed above
        *(f32x4*)&Xn[g_off] = xo;
        if (Zout) *(f32x4*)&Zout[g_off] = z4;
        if ((lane & 31) == 0) gate[(size_t)hs * n + i] = g;
      }
    } else
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int m = wave + t * NW;
      const int s = m / R, rr = m % R;
      const int i = node0 + rr;
      if (i >= n) continue;  // wave-uniform
      float z[EPL];
      float dot = 0.f;
      const size_t g_off = ((size_t)s * n + i) * D + lane * EPL;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        z[e] = T[m * LD + lane * EPL + e];
        dot += z[e] * wgl[e];
      }
      dot = wave_sum(dot);
      const float g = sigmoidf_(dot + c0);
      float xo[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        xo[e] = (1.f - g) * xres[t][e] + g * z[e];
        if (thresh) xo[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? xo[e] * keep_scale : 0.f;
        if (colstats) T[m * LD + lane * EPL + e] = fmaxf(xo[e], 0.f);  // own row, already consumed above
      }
      st_row<EPL>(&Xn[g_off], xo);
      if (Zout) st_row<EPL>(&Zout[g_off], z);
      if (lane == 0) gate[(size_t)s * n + i] = g;
    }
    KT_STAMP(14);
    // ---- optional: first stage of the classifier head's BatchNorm statistics while the tile is on chip: per
    // (strand, column) the exact two-pass (mean, M2) of relu(Xn) over this tile's nodes
    if (colstats) {
      __syncthreads();
      static_assert(MB * S * D <= 512, "one (strand, column) per thread");
      const int idx = threadIdx.x;
      if (idx < S * D) {
        const int s = idx / D, c = idx % D;
        const int cnt = min(R, n - node0);
        const float inv = 1.f / (float)cnt;
        float v[R];
        float sum = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          v[rr] = T[(s * R + rr) * LD + c];
          sum += rr < cnt ? v[rr] : 0.f;
        }
        const float mean = sum * inv;
        float m2 = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) m2 += rr < cnt ? (v[rr] - mean) * (v[rr] - mean) : 0.f;
        chan_combine(st_cnt, st_mean, st_m2, (float)cnt, mean, m2);
      }
    }
    if (!PROD) __syncthreads();  // T is rewritten by the next tile (PROD: only behind the next tile's first barrier)
  }
  if (colstats && threadIdx.x < S * D && tfirst < tend) {
    if (stat_acc) {   // accumulate mode (cgcn_common.hpp, STAT_ACC_*): this workgroup's sum x and sum x^2, formed in double
      const double nb = (double)st_cnt, mb = (double)st_mean;
      stat_acc_add((unsigned long long*)colstats, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), (int)threadIdx.x / D,
                   (int)threadIdx.x % D, nb * mb, (double)st_m2 + nb * mb * mb);
    } else {
      float* out = colstats + ((size_t)blockIdx.x * S * D + threadIdx.x) * 2;
      out[0] = st_mean;
      out[1] = st_m2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_layer_dense256 (round 6): k_layer_dense for D = 256 as ONE 16-wave workgroup per CU.
// W is 256 KB -- half of a CU's register file -- so it has to be spread over all 16 waves that fit a CU at 128 registers:
// wave w owns the 16 output columns [16 w, 16 w + 16) and keeps their 64 B-operand registers resident (k_layer_dense<S,256>
// held 128 per wave in 8 waves, spilled 20 registers and could not co-reside with a second workgroup).  The fragments arrive
// COALESCED: every wave reads whole 1 KiB rows of W (16 float4 per lane, all requested before the first wait) and the
// workgroup transposes them through LDS in four 64-row rounds -- the 8-wave kernel's prologue issued 128 four-byte loads
// per lane against a matrix that all 2 048 waves of the launch wanted at the same moment (12 us of a 33 us kernel,
// profiles/r05_d256_dense_experiment.txt; here 5 us).
// Per 16-row tile (R = 16 / S nodes x S strands): wave w puts row w of H into the LDS tile and holds row w of X for the
// residual mix (both one tile ahead), 64 fp32 MFMAs per wave (one 16 x 16 accumulator, K = 256 in the permuted order of
// tile_mfma), tanh(U + b) -> LDS, then wave w finishes row w: gate (DPP wave sum), mix, dropout, 16-byte stores.
// Column statistics (the last layer): one (strand, column) per thread over the tile's nodes, Chan-merged over the
// workgroup's contiguous tile chunk, emitted as one record or as fixed-point integer totals -- exactly k_layer_dense's.
// What bounds it (profiles/r06_dense256_experiment.txt): the ALUs.  A tile costs a SIMD 256 MFMAs x 32 cycles plus ~4 800
// cycles of vector work (tanh 27 instructions x 16 wave-elements, the row pass), and the two ADD UP: on this chip the fp32
// MFMA rate equals the fp32 vector rate (MI355X_MICROARCH.md) and a version of this kernel that issued every wave's tanh and
// row pass BETWEEN the MFMAs of the next tile (three-stage pipeline, one barrier per tile, straight-line block with buffer-
// descriptor stores and LDS-DMA loads) ran at the same 5.7-6.1 us per tile as this phase-by-phase one -- removed.
// ------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(1024) void k_layer_dense256(int n, int ntiles, const float* __restrict__ Hin,
                                                          const float* __restrict__ X, const float* __restrict__ W,
                                                          const float* __restrict__ bias, const float* __restrict__ wg,
                                                          const float* __restrict__ cg, float* __restrict__ Xn,
                                                          float* __restrict__ Zout, float* __restrict__ gate,
                                                          float keep_scale, uint32_t thresh,
                                                          const unsigned long long* __restrict__ rng_state,
                                                          uint32_t stream_id, float* __restrict__ colstats, int stat_chunk,
                                                          int stat_acc, unsigned long long* __restrict__ zero_words, int zero_count) {
  if (zero_words && (int)blockIdx.x < min((int)gridDim.x, 8))   // (totals of a LATER launch of the step: colstats_rows = -2)
    for (int i = (int)blockIdx.x * 1024 + (int)threadIdx.x; i < zero_count; i += min((int)gridDim.x, 8) * 1024) zero_words[i] = 0ull;
  constexpr int D = 256, ROWS = 16, R = ROWS / S, LD = D + 4, KCH = 64, TILE = ROWS * LD;
  // one W staging round (64 rows); afterwards the A tile T, the tanh tile Zt and the relu(Xn) tile St of the statistics
  __shared__ __attribute__((aligned(16))) float smem[KCH * LD];
  static_assert(KCH * LD >= 3 * TILE, "the tiles fit the staging buffer");
  float* const T = smem;
  float* const Zt = smem + TILE;
  float* const St = smem + 2 * TILE;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int j = wave * 16 + r;                // this lane's output column
  const int l4 = lane * 4;
  const int ms = wave / R, mr = wave % R;     // this wave's row of a tile: strand, node inside the tile
  // Tile walk as k_layer_dense: without column statistics tiles b, b + G, ...; with them a CONTIGUOUS chunk per workgroup
  const int tfirst = colstats ? (int)blockIdx.x * stat_chunk : (int)blockIdx.x;
  const int tstep = colstats ? 1 : (int)gridDim.x;
  const int tend = colstats ? min(ntiles, tfirst + stat_chunk) : ntiles;
  auto load_row = [&](f32x4& dst, const float* __restrict__ src, int tile) {
    const int i = tile * R + mr;
    dst = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n) dst = *(const f32x4*)&src[((size_t)ms * n + i) * D + l4];
  };
  KT_STAMP(8);
  // ---- W -> registers through LDS (round c: rows 64 c .. 64 c + 63)
  float bw[64];
  {
    f32x4 wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = *(const f32x4*)&W[(size_t)(wave + 16 * u) * D + l4];   // row wave + 16 u, whole
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c) __syncthreads();   // the round before has been read
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) *(f32x4*)&smem[(wave + 16 * uu) * LD + l4] = wv[4 * c + uu];
      __syncthreads();
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int u = 0; u < 4; ++u) bw[16 * c + 4 * tt + u] = smem[(16 * tt + 4 * q + u) * LD + j];   // k = 64 c + 16 tt + 4 q + u
    }
  }
  f32x4 hrow = {0.f, 0.f, 0.f, 0.f}, xnext = hrow, xres = hrow;
  if (tfirst < tend) {
    load_row(hrow, Hin, tfirst);
    load_row(xnext, X, tfirst);
  }
  const float bj = bias[j];
  const f32x4 wg4 = *(const f32x4*)&wg[l4];
  const float c0 = cg[0];
  const uint32_t key = thresh ? dropout_key(rng_state, stream_id) : 0u;
  float st_cnt = 0.f, st_mean = 0.f, st_m2 = 0.f;   // running statistics of this thread's (strand, column)
  __syncthreads();   // the last round has been read: the buffer becomes the tiles
  KT_STAMP(9);
  for (int tile = tfirst; tile < tend; tile += tstep) {
    const int node0 = tile * R;
    KT_STAMP(10);
    *(f32x4*)&T[wave * LD + l4] = hrow;
    xres = xnext;
    if (tile + tstep < tend) {   // both input streams one tile ahead
      load_row(hrow, Hin, tile + tstep);
      load_row(xnext, X, tile + tstep);
    }
    __syncthreads();
    KT_STAMP(11);
    // ---- U = H W (this wave's 16 columns)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
      const float* __restrict__ Ta = T + r * LD + 4 * q;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const f32x4 a = *(const f32x4*)&Ta[16 * t];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bw[4 * t + u], acc, 0, 0, 0);
      }
    }
    KT_STAMP(12);
    // ---- Z = tanh(U + b) -> the tanh tile (its last readers, the row pass of the tile before, are behind the barrier above)
#pragma unroll
    for (int e = 0; e < 4; ++e) Zt[(q * 4 + e) * LD + j] = layer_tanh(acc[e] + bj);
    __syncthreads();   // (also: every wave is done reading T)
    KT_STAMP(13);
    // ---- row `wave`: gate, residual mix, dropout, coalesced stores
    {
      const int i = node0 + mr;
      const f32x4 z4 = *(const f32x4*)&Zt[wave * LD + l4];
      float dot = z4[0] * wg4[0] + z4[1] * wg4[1] + z4[2] * wg4[2] + z4[3] * wg4[3];
      dot = wave_sum(dot);
      const float g = sigmoidf_(dot + c0);
      f32x4 xo = {0.f, 0.f, 0.f, 0.f};
      if (i < n) {
        const size_t g_off = ((size_t)ms * n + i) * D + l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xo[e] = (1.f - g) * xres[e] + g * z4[e];
          if (thresh) xo[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? xo[e] * keep_scale : 0.f;
        }
        *(f32x4*)&Xn[g_off] = xo;
        if (Zout) *(f32x4*)&Zout[g_off] = z4;
        if (lane == 0) gate[(size_t)ms * n + i] = g;
      }
      if (colstats) *(f32x4*)&St[wave * LD + l4] = (f32x4){fmaxf(xo[0], 0.f), fmaxf(xo[1], 0.f), fmaxf(xo[2], 0.f), fmaxf(xo[3], 0.f)};
    }
    KT_STAMP(14);
    // ---- optional: first stage of the classifier head's BatchNorm statistics while the tile is on chip: per (strand,
    // column) the exact two-pass (mean, M2) of relu(Xn) over this tile's nodes, Chan-merged into the chunk's
    if (colstats) {
      __syncthreads();
      const int idx = threadIdx.x;
      if (idx < S * D) {
        const int s = idx / D, c = idx % D;
        const int cnt = min(R, n - node0);
        const float inv = 1.f / (float)cnt;
        float v[R];
        float sum = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          v[rr] = St[(s * R + rr) * LD + c];
          sum += rr < cnt ? v[rr] : 0.f;
        }
        const float mean = sum * inv;
        float m2 = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) m2 += rr < cnt ? (v[rr] - mean) * (v[rr] - mean) : 0.f;
        chan_combine(st_cnt, st_mean, st_m2, (float)cnt, mean, m2);
      }
    }
  }
  if (colstats && threadIdx.x < S * D && tfirst < tend) {
    if (stat_acc) {   // accumulate mode (cgcn_common.hpp, STAT_ACC_*): this workgroup's sum x and sum x^2, formed in double
      const double nb = (double)st_cnt, mb = (double)st_mean;
      stat_acc_add((unsigned long long*)colstats, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), (int)threadIdx.x / D,
                   (int)threadIdx.x % D, nb * mb, (double)st_m2 + nb * mb * mb);
    } else {
      float* out = colstats + ((size_t)blockIdx.x * S * D + threadIdx.x) * 2;
      out[0] = st_mean;
      out[1] = st_m2;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The row-local launch of the layer backward: everything that is local to a (strand, node) row, plus the two dense
// products on MFMA:  dW = H^T dU  and  dHs = diag(row_scale) dU W^T  (dL/dH, pre-scaled: the operand of the gather
// over Ahat^T that follows in k_bwd_sliced).  Rows are the flattened [S*n] axis; every workgroup keeps its share of dW
// in accumulators and writes one partial:
//   partial layout per workgroup: [D*D dW][D db][D dwg][1 dcg][3 pad]
// d = 128: k_bwd_rowlocal_ring (below).  d = 256: k_bwd_rowlocal256s (after it).
// (Rounds 1-4 ran d = 256 as k_bwd_rowlocal256 -- 8 waves, the whole 256 x 256 dW in 128 accumulator registers per wave,
// one 263 KB record per workgroup -- plus k_dh_dense<256> for the dHs product: in the history at the round-4 tag.)
// ------------------------------------------------------------------------------------------
// k_bwd_rowlocal_ring (D = 128): the work of k_bwd_rowlocal<128, 32> as a PRODUCER / CONSUMER pair of wave teams that
// meet only through a ring of LDS slots with counted flags -- no workgroup barrier between the first tile and the last.
//   row team    (waves 0-7):  streams Z, X, H, dL/dXn (16 B per lane, one wave instruction = 2 rows), RING_PF slots of
//                             rows in flight in registers, all the row math (head prologue, gate / tanh derivatives,
//                             column sums), writes the H and dU operand tiles of a 16-row slot, then bumps FULL[slot];
//   matrix team (waves 8-15): waits for FULL[slot], both fp32 MFMA products from the slot -- dW += H^T dU (16 rows of dW
//                             per wave) and dHs = diag(row_scale) dU W^T (16 columns per wave, W^T fragments resident)
//                             -- interleaved MFMA by MFMA (the single dHs accumulation chain never waits for its own
//                             result), bumps FREE[slot] as soon as its last operand has left the LDS, stores dHs from
//                             the accumulators.
// Why (profiles/r03_stream_vs_mfma_microbench.txt, r03_rowlocal_rs_experiment.txt): a 5.5 TB/s stream and 120-130 TF/s
// of LDS-fed fp32 MFMA share a CU at max(stream, MFMA) when they run in waves that never wait for one another; the
// shipped kernel and three role-split versions of it with ONE workgroup barrier per tile all ran at stream + MFMA (47 %
// of the wave-cycles parked at s_barrier / s_waitcnt): behind a barrier every interval lasts as long as the slowest of
// 16 waves.  Here a wave only ever waits for the slot it needs: RING_SLOTS x 16 rows of slack between the teams.
//   flags: FULL[s] / FREE[s] are monotonic arrival counters in LDS (8 arrivals per use of the slot); an arrival is
//   s_waitcnt lgkmcnt(0) (the wave's own LDS writes / reads of the slot have completed) + one ds_add_u32 by lane 0; a
//   wait is a ds_read_b32 poll with s_sleep.  The LDS is one coherent memory for the workgroup: no fence, no cache.
//   slots: 16 rows x 128 floats, UNPADDED, 16-byte chunks XOR-swizzled by the row so that every access pattern is
//   bank-conflict free: H chunk c of row k sits at c ^ 4(k & 1) (column-wise ds_read_b32 A operands), dU chunk c of row
//   k at c ^ k (ds_read_b128 of 4 consecutive columns of row 4kk+q as B operands of four dW MFMAs -- the MFMA's column
//   index r stands for column 64h + 4r + u, the partial is stored accordingly -- and ds_read_b128 of 4 consecutive K
//   values of row r as A operands of four dHs MFMAs).
//   work split: workgroup b owns the contiguous slots [b NST / P, (b + 1) NST / P): every workgroup within one 16-row
//   slot of the mean (32-row tiles dealt round robin: 7.3 tiles per workgroup ran as 8 at chr1 size).
//   partial layout per workgroup: [D*D dW][D db][D dwg][1 dcg][3 pad]   (as k_bwd_rowlocal)
// Results equal k_bwd_rowlocal's up to the summation order inside a row and over the rows; bit-reproducible run to run
// (every sum has a fixed order: no atomics on data).
// ------------------------------------------------------------------------------------------
#ifndef RING_SLOTS
#define RING_SLOTS 8
#endif
#ifndef RING_ROW_WAVES
#define RING_ROW_WAVES 8 // 8: 16-wave workgroups, 2 row + 2 matrix waves per SIMD, <= 128 registers (needs RING_OPBUF 2, RING_PRIME 0);
#endif                   // 4: 12 waves, 1 row + 2 matrix per SIMD, <= 168.  Measured (profiles/r04_rowlocal_ring_experiment.txt): 8 wins
#define RING_THREADS ((RING_ROW_WAVES + 8) * 64)
#ifndef RING_PF
#define RING_PF 2        // slots of rows a row wave keeps in flight in registers (3: 4.65 vs 4.63 ms genome epoch)
#endif
#ifndef RING_PF_HEAD
#define RING_PF_HEAD 2   // ... in head mode (the head prologue needs ~25 more registers per row pair)
#endif
#ifndef RING_PRIO
#define RING_PRIO 1      // s_setprio of the matrix team (measured: 0 -> 1: 51.3 -> 45.6 us at n = 29 910)
#endif
#ifndef RING_OPBUF
#define RING_OPBUF 2     // matrix team: operand sets in flight (3: two steps ahead, 2: one step ahead, 8 registers fewer)
#endif
#ifndef RING_EARLY_FLAG
#define RING_EARLY_FLAG 1 // matrix team: read the next slot's FULL flag under the current slot's MFMAs; poll only if it was not up yet
#endif
#ifndef RING_PRIME
#define RING_PRIME 0     // matrix team: request the next slot's first operands under the current slot's last MFMAs (measured: no gain)
#endif
// Every switch below changes WHAT the kernel computes or how its teams meet (decomposition builds of tools/ring_decomp.sh:
// garbage results by design; the slowed teams of tests/test_gpu_ring_stress.py: same results, other timing).  None of
// them can be reached by a stray -D: they need -DCGCN_EXPERIMENT_BUILD, which chromegcn_amd/_build.py refuses for the
// in-tree library.
#if (defined(RING_NO_WAIT) || defined(RING_SKIP_MFMA) || defined(RING_SKIP_ROWTEAM) || defined(RING_SKIP_LOADS) || \
     defined(RING_TEST_SLOW_ROW) || defined(RING_TEST_SLOW_MATRIX) || defined(RL256_SKIP_MFMA) || defined(RL256_SKIP_LOADS) || \
     defined(BSX_NORIDERS) || defined(BSX_NODXN) || defined(BSX_NOSTORE)) && \
    !defined(CGCN_EXPERIMENT_BUILD)
#error "RING_NO_WAIT / RING_SKIP_* / RING_TEST_SLOW_* are experiment switches: build a variant with -DCGCN_EXPERIMENT_BUILD (tools/mkvariant.py), never the shipped library"
#endif
#if defined(RING_TEST_SLOW_ROW) || defined(RING_TEST_SLOW_MATRIX)
// test builds only: hold this wave back by a wave- and slot-dependent time (0 ... ~4 us), so that the waves of a team
// fall out of step with each other and one team runs the other one's flags dry (SLOW_ROW: the matrix team polls FULL
// on every slot; SLOW_MATRIX: the ring fills up and the row team polls FREE on every slot)
__device__ __forceinline__ void ring_test_delay(int wave, int it) {
  unsigned h = (unsigned)wave * 2654435761u ^ (unsigned)it * 40503u ^ (unsigned)blockIdx.x * 97u;
  h ^= h >> 7;
  for (unsigned i = 0; i < (h & 15u); ++i) __builtin_amdgcn_s_sleep(10);
}
#endif
__device__ __forceinline__ void ring_wait(const unsigned* flag, unsigned target) {
#ifdef RING_NO_WAIT   // experiment only (results are garbage): the two teams run free of each other -> their pure interference
  return;
#endif
  // Every spin is bounded by WALL time, not by a poll count (a debugger, thread trace or heavy instrumentation slows the
  // polls by orders of magnitude): the 100 MHz real-time counter is looked at once per 2^16 polls (a healthy launch never
  // gets that far: a poll is ~200 cycles, a launch ~10^5), and 10 s without progress can only mean a broken protocol --
  // trap (the launch fails with an error) instead of hanging the device.
  unsigned long long t0 = 0;
  for (unsigned spins = 1;; ++spins) {
    const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(v - target) >= 0) break;
    if ((spins & 0xFFFFu) == 0u) {
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (t0 == 0) t0 = now;
      else if (now - t0 > 1000000000ull) __builtin_trap();   // 10 s at 100 MHz
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");   // nothing of the slot is read or written ahead of the poll
}
__device__ __forceinline__ void ring_arrive(unsigned* flag, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS accesses of the slot have completed
  if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// HEAD: the last layer (dL/dXn recomputed from the head's backward state, HeadApply); DROP: ... with the head's dropout
// PROD == 1 (split products, cgcn_common.hpp): both products on the bf16 matrix cores.  The row team splits every H and
// dU row into its three bf16 levels on the way into the slot (the fp32 tiles are gone: a slot is 2 operands x 3 levels x
// 4 KB, RING_SLOTS_SP = 6 of them); the matrix team forms
//   dW  with v_mfma_f32_32x32x16_bf16 -- K = the slot's 16 rows; wave `own` owns H columns [32 (own >> 1), +32) x dU columns
//        [64 (own & 1), +64): two 32 x 32 accumulators; both operands are K-major reads of row-major tiles, i.e.
//        ds_read_b64_tr_b16 (the hardware transpose: a 16-lane group fetches 4 rows x 16 columns and every lane receives
//        one column's 4 rows); the six partial products go into ONE accumulator per block, small terms first --
//   dHs with v_mfma_f32_16x16x32_bf16 -- dU rows by ds_read_b128, the W^T levels resident (48 registers), two
//        accumulators (the leading product | the other five: SpAcc2).
// Level tiles in the swizzled image of cgcn_common.hpp (sp_sigma): every access pattern above is bank-conflict free.
// 768 matrix-core cycles per slot and wave instead of 2 048: the matrix team, which bounded the fp32 form (39 us alone
// against the row team's 18 at n = 29 910), no longer does.
#ifndef RING_SLOTS_SP
#define RING_SLOTS_SP 6
#endif
template <bool HEAD, bool DROP, int PROD>
__global__ __launch_bounds__(RING_THREADS) void k_bwd_rowlocal_ring(int M, int n, const float* __restrict__ dXn,
                                                            const float* __restrict__ Z, const float* __restrict__ X,
                                                            const float* __restrict__ gate, const float* __restrict__ dgate,
                                                            const float* __restrict__ H, const float* __restrict__ wg,
                                                            const float* __restrict__ rs, float* __restrict__ dHs,
                                                            float* __restrict__ part, HeadApply hp,
                                                            float* __restrict__ dxn_store, int row_blocks, int head_slabs,
                                                            const float* __restrict__ W) {
  constexpr int D = 128, SR = 16, NSL = PROD ? RING_SLOTS_SP : RING_SLOTS, PF = HEAD ? RING_PF_HEAD : RING_PF;
  constexpr int SLOT_F = PROD ? 3 * SR * D / 2 : SR * D;   // floats per slot and operand (PROD: three bf16 level tiles)
  constexpr int LVB = SR * D * 2;                          // bytes of one level tile
  constexpr int NRW = RING_ROW_WAVES;          // row-team waves (the matrix team always has 8: one per 16 rows of dW)
  constexpr int NT = RING_THREADS;
  constexpr int RT = SR / (2 * NRW);           // row PAIRS (one wave instruction = 2 rows) per row wave per slot
  static_assert(NRW * 2 * RT == SR && (NRW == 4 || NRW == 8), "row-team geometry");
  constexpr int PSTRIDE = D * D + 2 * D + 4;
  constexpr int RS = 2 * D + 4;
  __shared__ __attribute__((aligned(16))) float Hs[NSL][SLOT_F];
  if ((int)blockIdx.x >= row_blocks) {   // extra workgroups: the head's deferred second stage (see k_bwd_rowlocal); they
    const int extra = (int)blockIdx.x - row_blocks;      // stage through the (here unused) ring memory
    const int wslabs = (hp.hf_CP * D + hp.hf_CP) / 64;
    static_assert(sizeof(Hs) >= 4 * (NT / HEAD_STAT_COLS) * (HEAD_STAT_COLS + 1) * sizeof(double), "finalize staging fits the H ring");
    if (extra < wslabs)
      head_finalize_slab<NT, true>(extra, hp.hf_P, D, hp.hf_C, hp.hf_CP, hp.hf_part, hp.hf_dWout, hp.hf_dbout,
                                   hp.hf_accumulate, hp.dloss, &Hs[0][0]);
    else
      head_stats_finalize<NT, true>(extra - wslabs, hp.hf_P, n, hp.S, D, hp.hf_CP, hp.hf_part, hp.hf_dbn_w, hp.hf_dbn_b,
                                    nullptr, hp.hf_accumulate, hp.dloss, &Hs[0][0]);
    return;
  }
  __shared__ __attribute__((aligned(16))) float Us[NSL][SLOT_F];
  __shared__ __attribute__((aligned(16))) float Sc[NSL][SR];   // row_scale of the slot's rows (0 past the end)
  __shared__ __attribute__((aligned(16))) float Hc[HEAD ? 9 * D : 4];   // head mode: BatchNorm constants (row team)
  __shared__ __attribute__((aligned(16))) float red[NRW][RS];  // column sums of the row waves
  __shared__ unsigned flg[2 * NSL + 1];                        // FULL[0..NSL), FREE[NSL..2 NSL), DONE (row team's column sums)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NST = (M + SR - 1) / SR;
  // (the 64-bit divisions run on the vector unit: tell the compiler their results are uniform)
  const int s_begin = __builtin_amdgcn_readfirstlane((int)((long long)blockIdx.x * NST / row_blocks));
  const int ns = __builtin_amdgcn_readfirstlane((int)((long long)(blockIdx.x + 1) * NST / row_blocks)) - s_begin;   // slots of this workgroup
  float* P = part + (size_t)blockIdx.x * PSTRIDE;
  if (threadIdx.x < 2 * NSL + 1) flg[threadIdx.x] = 0u;
  unsigned* const FULL = flg;
  unsigned* const FREE = flg + NSL;
  unsigned* const DONE = flg + 2 * NSL;

  if (wave < NRW) {
    // =============================================================== row team
    const bool upper = lane >= 32;
    const int l = lane & 31;
    const int row0 = 2 * wave + (upper ? 1 : 0);   // this half-wave's rows inside a slot: row0 + 2 NRW t, t < RT
    const f32x4 wgl = *(const f32x4*)&wg[4 * l];
    f32x4 db_acc = {0.f, 0.f, 0.f, 0.f}, dwg_acc = {0.f, 0.f, 0.f, 0.f};
    float dcg_acc = 0.f;
    const uint32_t hkey = (HEAD && DROP) ? dropout_key(hp.rng_state, HEAD_STREAM_ID) : 0u;
    const float hgl = (HEAD && hp.dloss) ? hp.dloss[0] : 1.f;
    if (HEAD) {   // per-column BatchNorm constants of both strands: [s][invstd, mean, c0, c1][D], then bn weight
      for (int i = threadIdx.x; i < 2 * D; i += NRW * 64) {
        const int s = i / D, c = i % D, ss = s < hp.S ? s : 0;
        Hc[(s * 4 + 0) * D + c] = hp.invstd[ss * D + c];
        Hc[(s * 4 + 1) * D + c] = hp.mean[ss * D + c];
        if (hp.bacc) {   // accumulate mode: the column means from the integer totals the head kernel left (cgcn_common.hpp)
          float c0, c1;
          bacc_get(bacc_base(hp.bacc, hp.S, D), hp.S, D, ss, c, n, c0, c1);
          Hc[(s * 4 + 2) * D + c] = c0;
          Hc[(s * 4 + 3) * D + c] = c1;
        } else {
          Hc[(s * 4 + 2) * D + c] = hp.bnc[(ss * 2 + 0) * D + c];
          Hc[(s * 4 + 3) * D + c] = hp.bnc[(ss * 2 + 1) * D + c];
        }
        if (s == 0) Hc[8 * D + c] = hp.bn_w[c];
      }
    }
#ifndef RING_NT_LOADS
#define RING_NT_LOADS 0   // bit mask of row streams loaded non-temporally (every one is a last use): 1 Z, 2 X, 4 H, 8 dXn, 16 dym
#endif
#define RING_LD(bit, p) ((RING_NT_LOADS & (bit)) ? __builtin_nontemporal_load((const f32x4*)(p)) : *(const f32x4*)(p))
    struct Rows { f32x4 z[RT], x[RT], h[RT], g[RT]; float gt[RT], sc[RT]; };
    Rows R_[PF];
    // this lane's element of a slot's rows, as 32-bit offsets from UNIFORM per-slot bases (scalar base + vector offset
    // addressing: no 64-bit vector address arithmetic per load)
    const unsigned lane_el = (unsigned)(row0 * D + 4 * l);
    auto load_slot = [&](int it, Rows& w) {
      const int m0 = (s_begin + it) * SR;   // first row of the slot (uniform)
#ifdef RING_SKIP_LOADS
      if (false) {
#else
      if (m0 + SR <= M && (m0 >= n || m0 + SR <= n)) {
#endif
        // the whole slot lies inside the table and inside one strand (all but two slots of a launch): no predicates
        const int mi0 = m0 >= n ? m0 - n : m0;   // S <= 2
        const float* __restrict__ Zb = Z + (size_t)m0 * D;
        const float* __restrict__ Xb = X + (size_t)m0 * D;
        const float* __restrict__ Hb = H + (size_t)m0 * D;
        const float* __restrict__ Gb = HEAD ? hp.dym + (size_t)mi0 * D : dXn + (size_t)m0 * D;
        const float* __restrict__ gb = gate + m0;
        const float* __restrict__ rb = rs ? rs + mi0 : nullptr;
#pragma unroll
        for (int t = 0; t < RT; ++t) {   // BYTE offsets: base + zero-extended 32-bit vector offset is one addressing mode
          unsigned ob = 4u * lane_el + (unsigned)(2 * NRW * t * D * 4);
          unsigned rb4 = 4u * (unsigned)(row0 + 2 * NRW * t);
          asm volatile("" : "+v"(ob), "+v"(rb4));   // keep the zero-extension next to the loads (a hoisted 64-bit offset defeats the mode)
          w.z[t] = RING_LD(1, (const char*)Zb + ob);
          w.x[t] = RING_LD(2, (const char*)Xb + ob);
          w.h[t] = RING_LD(4, (const char*)Hb + ob);
          w.g[t] = RING_LD(HEAD ? 16 : 8, (const char*)Gb + ob);
          w.gt[t] = *(const float*)((const char*)gb + rb4);
          w.sc[t] = rb ? *(const float*)((const char*)rb + rb4) : 1.f;
        }
        return;
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int m = m0 + row0 + 2 * NRW * t;
#ifdef RING_SKIP_LOADS
        const bool ok = false;
#else
        const bool ok = m < M;
#endif
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        w.z[t] = w.x[t] = w.h[t] = w.g[t] = zero;
        w.gt[t] = 0.f;
        w.sc[t] = 0.f;
        if (ok) {
          const unsigned off = (unsigned)(m * D + 4 * l);
          const int mi = m >= n ? m - n : m;   // S <= 2
          w.z[t] = *(const f32x4*)&Z[off];
          w.x[t] = *(const f32x4*)&X[off];
          w.h[t] = *(const f32x4*)&H[off];
          w.g[t] = HEAD ? *(const f32x4*)&hp.dym[(unsigned)(mi * D + 4 * l)] : *(const f32x4*)&dXn[off];
          w.gt[t] = gate[m];
          w.sc[t] = rs ? rs[mi] : 1.f;
        }
      }
    };
    auto row_pass = [&](int it, Rows& w) {
      f32x4 du[RT], gupv[RT];
      float dgv[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int m = (s_begin + it) * SR + row0 + 2 * NRW * t;
        const unsigned off = (unsigned)(m * D + 4 * l);
        const float g = w.gt[t];
        f32x4 gup = w.g[t];
        if (HEAD) {   // dL/dXn of the last layer from the head's backward state (see HeadApply)
          const int s = m >= n ? 1 : 0;   // S <= 2
          const float invS = 1.f / (float)hp.S;
          const float* __restrict__ hc = Hc + s * 4 * D + 4 * l;
          const f32x4 is4 = *(const f32x4*)&hc[0], mu4 = *(const f32x4*)&hc[D], c04 = *(const f32x4*)&hc[2 * D], c14 = *(const f32x4*)&hc[3 * D];
          const f32x4 bw4 = *(const f32x4*)&Hc[8 * D + 4 * l];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xn = (1.f - g) * w.x[t][e] + g * w.z[t][e];
            float dy = gup[e] * invS * hgl;
            if (DROP) dy = dropout_keep(hkey, (uint32_t)(off + e), hp.thresh) ? dy * hp.keep_scale : 0.f;
            const float xh = (fmaxf(xn, 0.f) - mu4[e]) * is4[e];
            const float dr = bw4[e] * is4[e] * (dy - hgl * c04[e] - xh * (hgl * c14[e]));
            gup[e] = xn > 0.f ? dr : 0.f;
          }
          if (dxn_store && m < M) *(f32x4*)&dxn_store[off] = gup;   // k_bwd_sliced reads it back as dL/dXn for the (1-g) dXn term
        }
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) a += gup[e] * (w.z[t][e] - w.x[t][e]);
        dgv[t] = a;
        gupv[t] = gup;
      }
      half_sum_n<RT>(dgv, upper);   // the RT rows' reductions side by side
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int m = (s_begin + it) * SR + row0 + 2 * NRW * t;
        const float g = w.gt[t];
        float dg = dgv[t];
        if (dgate) dg += m < M ? dgate[m] : 0.f;   // an upstream gradient on the gate output itself: rare, read in place
        const float gamma = g * (1.f - g) * dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dz = g * gupv[t][e] + gamma * wgl[e];
          du[t][e] = dz * (1.f - w.z[t][e] * w.z[t][e]);
          db_acc[e] += du[t][e];
          dwg_acc[e] += gamma * w.z[t][e];
        }
        dcg_acc += gamma;
      }
      const int slot = it % NSL;
#ifdef RING_TEST_SLOW_ROW
      ring_test_delay(wave, it);
#endif
      if (it >= NSL) ring_wait(&FREE[slot], 8u * (unsigned)(it / NSL));   // the matrix team is done with the slot's last use
#ifndef RING_SKIP_MFMA
#pragma unroll
      for (int t = 0; t < RT; ++t) {   // swizzled chunk positions (see the header)
        const int rowi = row0 + 2 * NRW * t;
        if constexpr (PROD != 0) {
          const int wo = rowi * 256 + (((l >> 1) ^ sp_sigma(rowi)) << 4) + ((l & 1) << 3);   // bytes inside a level tile
          u32x2 h2, m2, l2;
          sp_split4(w.h[t], h2, m2, l2);
          unsigned char* hb = (unsigned char*)Hs[slot] + wo;
          *(u32x2*)hb = h2;
          *(u32x2*)(hb + LVB) = m2;
          *(u32x2*)(hb + 2 * LVB) = l2;
          sp_split4(du[t], h2, m2, l2);
          unsigned char* ub = (unsigned char*)Us[slot] + wo;
          *(u32x2*)ub = h2;
          *(u32x2*)(ub + LVB) = m2;
          *(u32x2*)(ub + 2 * LVB) = l2;
        } else {
          *(f32x4*)&Hs[slot][rowi * D + ((l ^ ((rowi & 1) << 2)) << 2)] = w.h[t];
          *(f32x4*)&Us[slot][rowi * D + ((l ^ rowi) << 2)] = du[t];
        }
        if (l == 0) Sc[slot][rowi] = w.sc[t];
      }
#endif
      ring_arrive(&FULL[slot], lane);
    };
#ifdef RING_SKIP_ROWTEAM
    __syncthreads();
    for (int it = 0; it < ns; ++it) {
      if (it >= NSL) ring_wait(&FREE[it % NSL], 8u * (unsigned)(it / NSL));
      ring_arrive(&FULL[it % NSL], lane);
    }
#else
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (j < ns) load_slot(j, R_[j]);
    __syncthreads();   // flags zeroed, head constants staged (matched by the matrix team): THE barrier of this kernel; the
                       // first slots' rows are already on their way
    for (int it0 = 0; it0 < ns; it0 += PF) {   // PF slots per trip: static register-set indexing
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int it = it0 + j;
        if (it < ns) {
          row_pass(it, R_[j]);
          if (it + PF < ns) load_slot(it + PF, R_[j]);
        }
      }
    }
#endif
    // column sums: the two halves of a wave hold the same columns; combine them, then the row waves through LDS in a fixed order
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      db_acc[e] += __shfl_xor(db_acc[e], 32);
      dwg_acc[e] += __shfl_xor(dwg_acc[e], 32);
    }
    dcg_acc += __shfl_xor(dcg_acc, 32);
    if (!upper) {
      *(f32x4*)&red[wave][4 * l] = db_acc;
      *(f32x4*)&red[wave][D + 4 * l] = dwg_acc;
      if (l == 0) red[wave][2 * D] = dcg_acc;
    }
    ring_arrive(DONE, lane);   // among the row waves only: the matrix team is still on its last slots
    ring_wait(DONE, (unsigned)NRW);
    for (int c = threadIdx.x; c < 2 * D + 1; c += NRW * 64) {
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < NRW; ++w) sacc += red[w][c];
      P[D * D + c] = sacc;
    }
  } else if constexpr (PROD != 0) {
    // =============================================================== matrix team, split products (see the header)
    const int own = wave - NRW;
    int lq = lane;
    asm volatile("" : "+v"(lq));
    const int r16 = lq & 15, q4 = lq >> 4;
    // W^T levels of this wave's 16 output columns of dHs: B operand of K-step s = W[16 own + r][32 s + 8 q .. + 7]
    bf16x8 th[4], tm[4], tl[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      float v[8];
      const f32x4 v0 = dHs ? *(const f32x4*)&W[(size_t)(16 * own + r16) * D + 32 * s4 + 8 * q4] : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 v1 = dHs ? *(const f32x4*)&W[(size_t)(16 * own + r16) * D + 32 * s4 + 8 * q4 + 4] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = v0[u];
        v[4 + u] = v1[u];
      }
      sp_split8(v, th[s4], tm[s4], tl[s4]);
    }
    f32x16 accW[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 16; ++e) accW[bb][e] = 0.f;
    __syncthreads();   // flags zeroed (and the row team's staging of the head constants)
    if (RING_PRIO) __builtin_amdgcn_s_setprio(RING_PRIO);
    // transposed reads (32x32x16 operands): lane = 32 h + 16 c16 + i; its 16-lane group fetches rows 8 h + 4 rd + (i >> 2),
    // the 8-byte piece (i & 3) of columns [16 c16, +16) of the wave's block, and receives column 16 c16 + i of those rows
    const int hh = lq >> 5, c16 = (lq >> 4) & 1, qp = (lq & 15) >> 2, pp = lq & 3;
    const int hcb = own >> 1, ucb = own & 1;
    int trH[2], trU[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int row = 8 * hh + 4 * rd + qp, sg = sp_sigma(row);
      trH[rd] = row * 256 + (((4 * hcb + 2 * c16 + (pp >> 1)) ^ sg) << 4) + ((pp & 1) << 3);
      trU[rd] = row * 256 + (((8 * ucb + 2 * c16 + (pp >> 1)) ^ sg) << 4) + ((pp & 1) << 3);   // block bb: ^ (bb << 6)
    }
    // row reads (16x16x32 A operand of dHs): row r16, chunk 4 s + q4
    const int ua0 = r16 * 256 + ((q4 ^ (sp_sigma(r16) & 3)) << 4), ua_hi = sp_sigma(r16) >> 2;
    const unsigned dh_lane = (unsigned)(4 * q4 * D + own * 16 + r16);   // this lane's element of a slot's dHs rows
    bool next_full = false;
    auto slot_loop = [&](auto DH_) {
      constexpr bool DH = decltype(DH_)::value;
      for (int it = 0; it < ns; ++it) {
        const int slot = it % NSL;
#ifdef RING_TEST_SLOW_MATRIX
        ring_test_delay(wave, it);
#endif
#ifndef RING_SKIP_MFMA
        const unsigned char* Hb = (const unsigned char*)Hs[slot];
        const unsigned char* Ub = (const unsigned char*)Us[slot];
        if (!(RING_EARLY_FLAG && next_full)) ring_wait(&FULL[slot], (unsigned)NRW * (unsigned)(it / NSL + 1));
        bf16x8 hf[3], uf[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) hf[v] = tr_pair(Hb + v * LVB + trH[0], Hb + v * LVB + trH[1]);
        SpAcc2 sa;
        sa.zero();
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
          for (int v = 0; v < 3; ++v) uf[v] = tr_pair(Ub + v * LVB + (trU[0] ^ (bb << 6)), Ub + v * LVB + (trU[1] ^ (bb << 6)));
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[2], uf[0], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[0], uf[2], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[1], uf[1], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[1], uf[0], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[0], uf[1], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[0], uf[0], accW[bb], 0, 0, 0);
          if (DH) {   // two K-steps of the dHs product beside each dW block
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int s4 = 2 * bb + s2;
              const int o = ua0 + ((s4 ^ ua_hi) << 6);
              const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ub + o));
              const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ub + LVB + o));
              const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ub + 2 * LVB + o));
              sa.step(ah, am, al, th[s4], tm[s4], tl[s4]);
            }
          }
        }
        f32x4 sc_cur = {0.f, 0.f, 0.f, 0.f};
        if (DH) sc_cur = *(const f32x4*)&Sc[slot][4 * q4];
        if (RING_EARLY_FLAG) {   // the next slot's flag, looked at once: usually up already -> no poll at the top of the next trip
          const unsigned nflag = __hip_atomic_load(&FULL[(it + 1) % NSL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          next_full = it + 1 < ns && (int)(__builtin_amdgcn_readfirstlane(nflag) - (unsigned)NRW * (unsigned)((it + 1) / NSL + 1)) >= 0;
          asm volatile("" ::: "memory");
        }
        ring_arrive(&FREE[slot], lane);   // every operand of the slot is in registers
        if (DH) {
          const f32x4 hacc = sa.sum();
          const int m0 = (s_begin + it) * SR + 4 * q4;
          const unsigned o = (unsigned)((s_begin + it) * SR * D) + dh_lane;
          if ((s_begin + it) * SR + SR <= M) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dHs[o + e * D] = hacc[e] * sc_cur[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m0 + e < M) dHs[o + e * D] = hacc[e] * sc_cur[e];
          }
        }
#else
        ring_wait(&FULL[slot], (unsigned)NRW * (unsigned)(it / NSL + 1));
        ring_arrive(&FREE[slot], lane);
#endif
      }
    };
    if (dHs) slot_loop(std::true_type{});
    else slot_loop(std::false_type{});
    // ---- this workgroup's dW partial: accW[bb][reg] = dW[32 hcb + (reg & 3) + 8 (reg >> 2) + 4 hh][64 ucb + 32 bb + (lane & 31)]
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        P[(32 * hcb + (e & 3) + 8 * (e >> 2) + 4 * hh) * D + 64 * ucb + 32 * bb + (lq & 31)] = accW[bb][e];
  } else {
    // =============================================================== matrix team
    const int own = wave - NRW;   // 16 rows of dW / 16 columns of dHs
    int lq = lane;
    asm volatile("" : "+v"(lq));
    const int r = lq & 15, q = lq >> 4;
    f32x4 accW[8], WT[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // W^T fragments of this wave's 16 output columns of dHs: B operand of k-step (t, u) = W[16 own + r][16 t + 4 q + u]
#pragma unroll
    for (int t = 0; t < 8; ++t) WT[t] = dHs ? *(const f32x4*)&W[(size_t)(16 * own + r) * D + 16 * t + 4 * q] : (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // flags zeroed (and the row team's staging of the head constants)
    if (RING_PRIO) __builtin_amdgcn_s_setprio(RING_PRIO);
    // operand offsets inside a slot (floats), see the header for the swizzles
    //   dW A: H[4 kk + q][16 own + r]            -> + kk * 4 D
    const int ha_off = q * D + (((4 * own + (r >> 2)) ^ ((q & 1) << 2)) << 2) + (r & 3);
    //   dW B: dU[4 kk + q][64 h + 4 r .. + 3]    -> chunk (16 h + r) ^ (4 kk + q) = 16 h + 4 ((r >> 2) ^ kk) + ((r & 3) ^ q)
    const int ub_lo = q * D + (((r & 3) ^ q) << 2);
    //   dHs A: dU[r][16 t + 4 q .. + 3]          -> chunk (4 t + q) ^ r = 4 (t ^ (r >> 2)) + (q ^ (r & 3))
    const int ua_lo = r * D + ((q ^ (r & 3)) << 2);
    const int rq = r >> 2;
    const unsigned dh_lane = (unsigned)(4 * q * D + own * 16 + r);   // this lane's element of a slot's dHs rows
    // Software pipeline across slots: under the last MFMA step of a slot the next slot's FULL flag is looked at (not
    // waited for) and, if it is up, its first operand set is requested, so a slot whose rows are ready starts without
    // an LDS round trip; otherwise the wave polls at the top of the next trip.
    constexpr int OB = RING_OPBUF;   // operand sets in flight: step s + OB - 1 is requested before the MFMAs of step s
    float av[4];
    f32x4 bv[OB], uv[OB];
    bool primed = false, next_full = false;
    // DH: with the dHs product (a gather over Ahat^T follows) or without (dHs == NULL: the layer's input is a leaf nobody
    // differentiates -- the engine's default for the first layer -- so the matrix team runs the dW product alone: half its MFMAs)
    auto slot_loop = [&](auto DH_) {
      constexpr bool DH = decltype(DH_)::value;
      auto rd = [&](const float* __restrict__ Hb, const float* __restrict__ Ub, int s) {
        const int kk = s >> 1, h = s & 1;
        if (h == 0) av[kk] = Hb[ha_off + kk * 4 * D];
        bv[s % OB] = *(const f32x4*)&Ub[ub_lo + kk * 4 * D + 64 * h + ((rq ^ kk) << 4)];
        if (DH) uv[s % OB] = *(const f32x4*)&Ub[ua_lo + (((s & 3) ^ rq) << 4) + (s & 4) * 16];
      };
      for (int it = 0; it < ns; ++it) {
        const int slot = it % NSL;
#ifdef RING_TEST_SLOW_MATRIX
        ring_test_delay(wave, it);
#endif
#ifndef RING_SKIP_MFMA
        const float* __restrict__ Hb = Hs[slot];
        const float* __restrict__ Ub = Us[slot];
        if (!primed) {
          if (!(RING_EARLY_FLAG && next_full)) ring_wait(&FULL[slot], (unsigned)NRW * (unsigned)(it / NSL + 1));
          rd(Hb, Ub, 0);
        }
        if (OB == 3) rd(Hb, Ub, 1);
        f32x4 sc_cur;
        primed = false;
        const int nslot = (it + 1) % NSL;
        unsigned nflag = 0u;
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
        // step s = (kk, h) = (s >> 1, s & 1) of the dW product and k-step block t = s of the dHs product; operands of step
        // s + OB - 1 are requested before the MFMAs of step s
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          if (s + OB - 1 < 8) rd(Hb, Ub, s + OB - 1);
          if (s == 5 && (RING_PRIME || RING_EARLY_FLAG)) nflag = __hip_atomic_load(&FULL[nslot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (DH && s == 7) sc_cur = *(const f32x4*)&Sc[slot][4 * q];   // (requested here, not at the top: four registers less through the loop)
          if (s == 7 && RING_EARLY_FLAG) {   // the next slot's flag was looked at two steps ago: usually up already -> no poll,
            next_full = it + 1 < ns &&       // no exposed LDS round trip at the top of the next trip
                        (int)(__builtin_amdgcn_readfirstlane(nflag) - (unsigned)NRW * (unsigned)((it + 1) / NSL + 1)) >= 0;
            asm volatile("" ::: "memory");
          }
          if (s == 7 && RING_PRIME && it + 1 < ns &&
              (int)(__builtin_amdgcn_readfirstlane(nflag) - (unsigned)NRW * (unsigned)((it + 1) / NSL + 1)) >= 0) {
            asm volatile("" ::: "memory");
            primed = true;
            rd(Hs[nslot], Us[nslot], 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          const int kk = s >> 1, h = s & 1;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            accW[h * 4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bv[s % OB][u], accW[h * 4 + u], 0, 0, 0);
            if (DH) {
              hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(uv[s % OB][u], WT[s][u], hacc, 0, 0, 0);
              if (u < 3) __builtin_amdgcn_sched_barrier(0);   // keep the alternation: two links of the dHs chain are 64 cycles apart
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        ring_arrive(&FREE[slot], lane);   // every operand of the slot is in registers (the last step's MFMAs covered the reads)
        if (DH) {   // 64-byte row segments straight from the accumulators (32-bit offsets from the uniform base)
          const int m0 = (s_begin + it) * SR + 4 * q;
          const unsigned o = (unsigned)((s_begin + it) * SR * D) + dh_lane;
          if ((s_begin + it) * SR + SR <= M) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dHs[o + e * D] = hacc[e] * sc_cur[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m0 + e < M) dHs[o + e * D] = hacc[e] * sc_cur[e];
          }
        }
#else
        ring_wait(&FULL[slot], (unsigned)NRW * (unsigned)(it / NSL + 1));
        ring_arrive(&FREE[slot], lane);
#endif
      }
    };
    if (dHs) slot_loop(std::true_type{});
    else slot_loop(std::false_type{});
    // ---- this workgroup's dW partial: accW[4 h + u][e] = dW[16 own + 4 q + e][64 h + 4 r + u]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *(f32x4*)&P[(own * 16 + q * 4 + e) * D + 64 * h + 4 * r] = (f32x4){accW[h * 4 + 0][e], accW[h * 4 + 1][e], accW[h * 4 + 2][e], accW[h * 4 + 3][e]};
  }
}

// ------------------------------------------------------------------------------------------
// k_bwd_rowlocal256s (d = 256, round 5): the row-local launch with BOTH dense products, like d = 128's ring kernel, and a
// quarter of the partial-record traffic.  What was wrong with k_bwd_rowlocal256 + k_dh_dense (profiles/r04_chr21_d256L4_*):
// a workgroup kept the whole 256 x 256 dW in 128 accumulator registers per wave, so (i) there was no room for the W^T
// fragments -- diag(1/deg) dU went to memory and came back through a second launch -- and (ii) every workgroup wrote a
// 263 KB partial record: 256 records = 67 MB out and 67 MB back in through the second stage, MORE than the row tensors
// themselves at chr21 size (47 MB), where a workgroup owns 45 rows.
// Here the OUTPUT COLUMNS are cut instead: workgroup (range, cs) owns column slab cs (64 of the 256 columns of dU / of the
// columns of dW and dHs) of a contiguous range of 32-row tiles.  It streams the range's FULL rows (the gate's row sum needs
// them; the four slab workgroups of a range sit on one XCD -- b, b + 8, b + 16, b + 24 -- so three of the four reads are L2
// hits), recomputes the row math (cheap VALU, 4x redundant), and per tile runs
//     dW[:, slab]  += H^T dU[:, slab]             matrix wave w: H columns [32 w, 32 w + 32) x 64  -> 32 accumulator registers
//     dHs[:, slab]  = diag(1/deg) dU W^T[:, slab]  matrix wave w: rows 16 (w >> 2).., columns 16 (w & 3)..; W rows resident: 64
// 64 + 64 MFMAs per wave and tile, interleaved one by one (the single dHs chain never waits for its own result).  The
// four slabs of a range write disjoint column slabs of ONE record: 64 records (16.8 MB) instead of 256 (67 MB), no dU round
// trip, no k_dh_dense launch.
// Two TEAMS, as in k_bwd_rowlocal_ring: a row team (waves 0-3: streams, row math, fills a 32-row slot of H and dU) and a
// matrix team (waves 4-11: both products from the slot), meeting only through counted FULL / FREE flags of two slots -- the
// barrier-per-tile form of this kernel ran row pass + products (52 us at chr21 size: 24 + 29), not their maximum
// (profiles/r05_d256_rowlocal_experiment.txt).  A row wave keeps 4 rows in registers, half a tile ahead of their use (a row's loads
// are issued the moment the registers of the row 4 before it are free).  Sums are in fixed order:
// bit-reproducible.
// ------------------------------------------------------------------------------------------
#ifndef RL256_ROW_WAVES
#define RL256_ROW_WAVES 8   // 8: 16 waves, 128 registers; 4: 12 waves, 168 (measured: the row team of 4 is latency-bound, 6.8 us per tile against 4.2 of MFMA)
#endif
#define RL256_THREADS ((RL256_ROW_WAVES + 8) * 64)
#ifndef RL256_PRIO
#define RL256_PRIO 1       // s_setprio of the matrix team
#endif
#ifndef RL256_ROW_PRIO
#define RL256_ROW_PRIO 0   // ... of the row team
#endif
template <int TR>
__global__ __launch_bounds__(RL256_THREADS) void k_bwd_rowlocal256s(int M, int n, const float* __restrict__ dXn,
                                                          const float* __restrict__ Z, const float* __restrict__ X,
                                                          const float* __restrict__ gate, const float* __restrict__ dgate,
                                                          const float* __restrict__ H, const float* __restrict__ wg,
                                                          const float* __restrict__ rs, float* __restrict__ dHs,
                                                          float* __restrict__ part, HeadApply hp,
                                                          float* __restrict__ dxn_store, int row_blocks, int head_slabs,
                                                          const float* __restrict__ W) {
  constexpr int D = 256, NRW = RL256_ROW_WAVES, NT = RL256_THREADS, SW = 64;   // SW: slab width (columns)
  static_assert(TR == 32, "one 16 x 16 dHs tile per matrix wave needs 32-row tiles");
  constexpr int LD = D + 16, EPL = D / 64, RPW = TR / NRW, PSTRIDE = D * D + 2 * D + 4, RS = 2 * D + 4;
  static_assert(EPL == 4, "one 16-byte chunk per lane");
  __shared__ __attribute__((aligned(16))) float Hs[2][TR * LD];
  if ((int)blockIdx.x >= row_blocks) {   // extra workgroups: the head's deferred second stage (independent work, fused
    const int extra = (int)blockIdx.x - row_blocks;      // "horizontally"); they stage through the (here unused) slot memory
    const int wslabs = (hp.hf_CP * D + hp.hf_CP) / 64;
    static_assert(sizeof(Hs) >= 4 * (NT / HEAD_STAT_COLS) * (HEAD_STAT_COLS + 1) * sizeof(double), "finalize staging fits the H slots");
    if (extra < wslabs)
      head_finalize_slab<NT, true>(extra, hp.hf_P, D, hp.hf_C, hp.hf_CP, hp.hf_part, hp.hf_dWout, hp.hf_dbout,
                                   hp.hf_accumulate, hp.dloss, &Hs[0][0]);
    else
      head_stats_finalize<NT, true>(extra - wslabs, hp.hf_P, n, hp.S, D, hp.hf_CP, hp.hf_part, hp.hf_dbn_w, hp.hf_dbn_b,
                                    nullptr, hp.hf_accumulate, hp.dloss, &Hs[0][0]);
    return;
  }
  __shared__ __attribute__((aligned(16))) float Us[2][TR * LD];
  __shared__ __attribute__((aligned(16))) float red[NRW][RS];   // column sums of the row waves
  __shared__ unsigned flg[5];                                    // FULL[2], FREE[2], DONE
  __shared__ float Hc[2 * 2 * D];                                // accumulate mode: bnc decoded from the integer totals, [S][2][D]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (hp.dym && hp.bacc) {   // (complete at THE barrier below; cgcn_common.hpp, bacc_get)
    for (int idx = threadIdx.x; idx < hp.S * D; idx += NT) {
      const int s = idx / D, c = idx % D;
      float c0, c1;
      bacc_get(bacc_base(hp.bacc, hp.S, D), hp.S, D, s, c, n, c0, c1);
      Hc[(s * 2 + 0) * D + c] = c0;
      Hc[(s * 2 + 1) * D + c] = c1;
    }
  }
  // workgroup -> (range, slab): the four slabs of a range are 8 workgroup ids apart (one XCD under round-robin dispatch)
  const int b = (int)blockIdx.x, NR = row_blocks >> 2;
  const int cs = (b >> 3) & 3, range = (b & 7) + 8 * (b >> 5);
  const int ntiles = (M + TR - 1) / TR;
  const int t_begin = __builtin_amdgcn_readfirstlane((int)((long long)range * ntiles / NR));
  const int nt = __builtin_amdgcn_readfirstlane((int)((long long)(range + 1) * ntiles / NR)) - t_begin;   // tiles of this workgroup
  float* P = part + (size_t)range * PSTRIDE;
  if (threadIdx.x < 5) flg[threadIdx.x] = 0u;
  unsigned* const FULL = flg;
  unsigned* const FREE = flg + 2;
  unsigned* const DONE = flg + 4;

  if (wave < NRW) {
    // =============================================================== row team
    float wgl[EPL], db_acc[EPL], dwg_acc[EPL];
    float dcg_acc = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      wgl[e] = wg[lane * EPL + e];
      db_acc[e] = 0.f;
      dwg_acc[e] = 0.f;
    }
    const uint32_t hkey = (hp.dym && hp.thresh) ? dropout_key(hp.rng_state, HEAD_STREAM_ID) : 0u;
    const float hgl = (hp.dym && hp.dloss) ? hp.dloss[0] : 1.f;
    constexpr int RIF = RPW < 4 ? RPW : 4;   // rows a wave holds in registers (in flight)
    float gup[RIF][EPL], z[RIF][EPL], x[RIF][EPL], h[RIF][EPL], gt[RIF], dgt[RIF];
    auto load_row = [&](int it, int tt) {   // row (wave + NRW tt) of the workgroup's it-th tile, into register set tt % RIF
      const int t = tt % RIF;
      const int m = (t_begin + it) * TR + wave + tt * NRW;
#ifdef RL256_SKIP_LOADS   // decomposition build (garbage results): no row streams
      const bool ok = false;
#else
      const bool ok = it < nt && m < M;
#endif
      const size_t off = (size_t)m * D + lane * EPL;
      if (ok) {
        ld_row<EPL>(z[t], &Z[off]);
        ld_row<EPL>(x[t], &X[off]);
        ld_row<EPL>(h[t], &H[off]);
        ld_row<EPL>(gup[t], hp.dym ? &hp.dym[(size_t)(m >= n ? m - n : m) * D + lane * EPL] : &dXn[off]);
      } else {
        zero_row<EPL>(z[t]);
        zero_row<EPL>(x[t]);
        zero_row<EPL>(h[t]);
        zero_row<EPL>(gup[t]);
      }
      gt[t] = ok ? gate[m] : 0.f;
      dgt[t] = (ok && dgate) ? dgate[m] : 0.f;
    };
#pragma unroll
    for (int t = 0; t < RIF; ++t) load_row(0, t);
    __syncthreads();   // flags zeroed: THE barrier of this kernel (matched by the matrix team)
    if (RL256_ROW_PRIO) __builtin_amdgcn_s_setprio(RL256_ROW_PRIO);
    KT_STAMP_NW(0, 0, true);
    for (int it = 0; it < nt; ++it) {
      const int slot = it & 1;
      if (it >= 2) ring_wait(&FREE[slot], 8u * (unsigned)(it >> 1));
      KT_STAMP_NW(1 + 2 * it, 0, it < 3);   // the matrix team is done with the slot's last use
      float* __restrict__ Ht = Hs[slot];
      float* __restrict__ Ut = Us[slot];
#pragma unroll
      for (int tt = 0; tt < RPW; ++tt) {
        const int t = tt % RIF;
        const int trow = wave + tt * NRW;
        const int m = (t_begin + it) * TR + trow;
        float du[EPL];
        if (m < M) {
          const size_t off = (size_t)m * D + lane * EPL;
          const float g = gt[t];
          if (hp.dym) {   // dL/dXn of the last layer from the head's backward state (see HeadApply)
            const int s = m >= n ? 1 : 0;   // S <= 2
            const float invS = 1.f / (float)hp.S;
            float is[EPL], mu[EPL], bw_[EPL], c0[EPL], c1[EPL];
            ld_row<EPL>(is, &hp.invstd[s * D + lane * EPL]);
            ld_row<EPL>(mu, &hp.mean[s * D + lane * EPL]);
            ld_row<EPL>(bw_, &hp.bn_w[lane * EPL]);
            if (hp.bacc) {
              ld_row<EPL>(c0, &Hc[(s * 2 + 0) * D + lane * EPL]);
              ld_row<EPL>(c1, &Hc[(s * 2 + 1) * D + lane * EPL]);
            } else {
              ld_row<EPL>(c0, &hp.bnc[(s * 2 + 0) * D + lane * EPL]);
              ld_row<EPL>(c1, &hp.bnc[(s * 2 + 1) * D + lane * EPL]);
            }
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              const float xn = (1.f - g) * x[t][e] + g * z[t][e];
              float dy = gup[t][e] * invS * hgl;
              if (hp.thresh) dy = dropout_keep(hkey, (uint32_t)(off + e), hp.thresh) ? dy * hp.keep_scale : 0.f;
              const float xh = (fmaxf(xn, 0.f) - mu[e]) * is[e];
              const float dr = bw_[e] * is[e] * (dy - hgl * c0[e] - xh * (hgl * c1[e]));
              gup[t][e] = xn > 0.f ? dr : 0.f;
            }
            if (dxn_store && cs == 0) st_row<EPL>(&dxn_store[off], gup[t]);  // one of the four slab workgroups stores the row
          }
          float dg = 0.f;
#pragma unroll
          for (int e = 0; e < EPL; ++e) dg += gup[t][e] * (z[t][e] - x[t][e]);
          dg = wave_sum(dg) + dgt[t];
          const float gamma = g * (1.f - g) * dg;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const float dz = g * gup[t][e] + gamma * wgl[e];
            du[e] = dz * (1.f - z[t][e] * z[t][e]);
            db_acc[e] += du[e];
            dwg_acc[e] += gamma * z[t][e];
          }
          dcg_acc += gamma;
        } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e) du[e] = 0.f;
        }
        *(f32x4*)&Ht[trow * LD + lane * EPL] = (f32x4){h[t][0], h[t][1], h[t][2], h[t][3]};
        *(f32x4*)&Ut[trow * LD + lane * EPL] = (f32x4){du[0], du[1], du[2], du[3]};
        // this row's registers are free: the row RIF rows ahead (of this tile or the next), half a tile ahead of its use
        if (tt + RIF < RPW) load_row(it, tt + RIF);
        else load_row(it + 1, tt + RIF - RPW);
      }
      ring_arrive(&FULL[slot], lane);
      KT_STAMP_NW(2 + 2 * it, 0, it < 3);
    }
    KT_STAMP_NW(7, 0, true);
    // column sums: every slab workgroup holds all 256 columns' sums (it computed the full rows); it writes its slab's
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      red[wave][lane * EPL + e] = db_acc[e];
      red[wave][D + lane * EPL + e] = dwg_acc[e];
    }
    if (lane == 0) red[wave][2 * D] = dcg_acc;
    ring_arrive(DONE, lane);   // among the row waves only: the matrix team is still on its last slots
    ring_wait(DONE, (unsigned)NRW);
    for (int c = threadIdx.x; c < 2 * D + 1; c += NRW * 64) {
      const int col = c < D ? c : (c < 2 * D ? c - D : -1);
      const bool mine = col >= 0 ? (col >= SW * cs && col < SW * cs + SW) : (cs == 0);
      if (!mine) continue;
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < NRW; ++w) sacc += red[w][c];
      P[D * D + c] = sacc;
    }
  } else {
    // =============================================================== matrix team
    const int own = wave - NRW;
    const int r = lane & 15, q = lane >> 4;
    const int rb = own >> 2, cb = own & 3;   // this wave's dHs tile: rows 16 rb.., slab columns 16 cb..
    f32x4 R[2][4];   // dW accumulators: H columns (2 own + ib) * 16.., slab columns jb * 16..
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) R[ib][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // W^T fragments of this wave's 16 output columns of dHs: B operand of k-step (t, u) = W[64 cs + 16 cb + r][16 t + 4 q + u]
    f32x4 WT[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
      WT[t] = dHs ? *(const f32x4*)&W[(size_t)(SW * cs + 16 * cb + r) * D + 16 * t + 4 * q] : (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // flags zeroed
    if (RL256_PRIO) __builtin_amdgcn_s_setprio(RL256_PRIO);
    // DH: with the dHs product (a gather over Ahat^T follows) or without (dHs == NULL: nobody differentiates the layer's
    // input): a compile-time switch, so that the interleaved MFMA stream is straight-line code
    auto tile_loop = [&](auto DH_) {
      constexpr bool DH = decltype(DH_)::value;
      for (int it = 0; it < nt; ++it) {
        const int slot = it & 1;
        const float* __restrict__ Ht = Hs[slot];
        const float* __restrict__ Ut = Us[slot];
        KT_STAMP_NW(8, NRW * 64, it == 0);
        ring_wait(&FULL[slot], (unsigned)NRW * (unsigned)((it >> 1) + 1));
        KT_STAMP_NW(9 + 2 * it, NRW * 64, it < 3);
        // the two products of the tile, one MFMA of each in turn: dW step kk (4 rows of K) holds 8 MFMAs, dHs k-step blocks
        // t = 2 kk, 2 kk + 1 hold 4 each
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#ifndef RL256_SKIP_MFMA   // (decomposition build, garbage results: no products)
#pragma unroll
        for (int kk = 0; kk < TR / 4; ++kk) {
          const int k = 4 * kk + q;
          float a[2], bq[4];
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) a[ib] = Ht[k * LD + (2 * own + ib) * 16 + r];
#pragma unroll
          for (int jb = 0; jb < 4; ++jb) bq[jb] = Ut[k * LD + SW * cs + jb * 16 + r];
          f32x4 ua[2];
          if (DH) {
            ua[0] = *(const f32x4*)&Ut[(16 * rb + r) * LD + 16 * (2 * kk) + 4 * q];
            ua[1] = *(const f32x4*)&Ut[(16 * rb + r) * LD + 16 * (2 * kk + 1) + 4 * q];
          }
#pragma unroll
          for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
              R[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ib], bq[jb], R[ib][jb], 0, 0, 0);
              if (DH) {
                const int i = jb * 2 + ib;   // 0..7 -> (t, u) = (2 kk + i / 4, i % 4)
                hacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[i >> 2][i & 3], WT[2 * kk + (i >> 2)][i & 3], hacc, 0, 0, 0);
              }
            }
        }
#endif
        ring_arrive(&FREE[slot], lane);   // every operand of the slot is in registers (s_waitcnt lgkmcnt(0) inside)
        KT_STAMP_NW(10 + 2 * it, NRW * 64, it < 3);
        if (DH) {   // rows 16 rb + 4 q + e, column 64 cs + 16 cb + r: 64-byte row segments straight from the accumulators
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = (t_begin + it) * TR + 16 * rb + 4 * q + e;
            if (m < M) dHs[(size_t)m * D + SW * cs + 16 * cb + r] = hacc[e] * (rs ? rs[m >= n ? m - n : m] : 1.f);
          }
        }
      }
    };
    if (dHs) tile_loop(std::true_type{});
    else tile_loop(std::false_type{});
    KT_STAMP_NW(15, NRW * 64, true);
    // ---- this workgroup's column slab of the range's partial record: [D*D dW][D db][D dwg][1 dcg][3 pad]
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) P[((2 * own + ib) * 16 + q * 4 + e) * D + SW * cs + jb * 16 + r] = R[ib][jb][e];
  }
}

// Fused optimizer step (cgcn_sgd_fuse): torch.optim.SGD on the flat arenas, element `idx`, given its final gradient g.
struct SgdFuse {
  float* param;          // nullptr = no fusion
  const float* grad;
  float* mom;
  int count;
  float lr, mu, wd, grad_scale;
  int nesterov;
  unsigned long long* rng_state;
};
__device__ __forceinline__ void sgd_apply(const SgdFuse& sg, int idx, float g) {
  const float pi = sg.param[idx];
  const float d = g * sg.grad_scale + sg.wd * pi;
  float upd = d;
  if (sg.mom) {
    const float b = sg.mu * sg.mom[idx] + d;
    sg.mom[idx] = b;
    upd = sg.nesterov ? d + sg.mu * b : b;
  }
  sg.param[idx] = pi - sg.lr * upd;
}

// One 64-element slab of the second-stage sum, computed by a workgroup of NT threads (NT/64 partial slices).
// Used by k_reduce_partials and, fused "horizontally", by the extra workgroups at the end of k_bwd_sliced's grid.
template <int NT>
__device__ __forceinline__ void reduce_slab(int slab, int P, int D, const float* __restrict__ part,
                                            float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dwg,
                                            float* __restrict__ dcg, int accumulate, const SgdFuse& sg) {
  constexpr int NS = NT / 64;
  const int PSTRIDE = D * D + 2 * D + 4;
  const int total = D * D + 2 * D + 1;
  __shared__ __attribute__((aligned(16))) float red[NS][64];
  // lane = (el4, sub): 16 lanes x float4 cover the slab, the 4 sub-groups of each of the NS waves take contiguous
  // ranges of the P partials (see head_finalize_slab): one batch of independent 16-byte loads per thread at P <= 256
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el4 = lane & 15, sub = lane >> 4;
  const int e0 = slab * 64 + el4 * 4;  // PSTRIDE is a multiple of 4 and the row is padded to it
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (e0 < total) {
    const int per = (P + NS * 4 - 1) / (NS * 4);
    const int p0 = (wave * 4 + sub) * per, p1 = min(P, p0 + per);
    for (int p = p0; p < p1; p += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(part + (size_t)min(p + u, p1 - 1) * PSTRIDE + e0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + u < p1) a += t[u];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = a[k];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    a[k] = v;
  }
  if (sub == 0) *(f32x4*)&red[wave][el4 * 4] = a;
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = slab * 64 + el;
  float s;
  __syncthreads();
  if (slice == 0 && e < total) {
    s = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += red[w][el];
    float* dst;
    if (e < D * D) dst = dW + e;
    else if (e < D * D + D) dst = db + (e - D * D);
    else if (e < D * D + 2 * D) dst = dwg + (e - D * D - D);
    else dst = dcg;
    *dst = accumulate ? (*dst + s) : s;
    if (sg.param) sgd_apply(sg, (int)(dst - sg.grad), s);   // this element's gradient is final: step it right here
  }
}

// The same second stage for FEW partial records (d = 256: 64 records of 263 KB): one workgroup finishes 256 elements instead
// of 64 -- a wave-row of float4 covers them, wave w sums records [w P / 8, (w + 1) P / 8) with all of its loads in flight, the
// eight waves' sums are added in wave order.  At d = 256 the 64-element form meant 1 033 rider workgroups of 16 KB each at the
// tail of k_bwd_sliced (round 6: 259 of 64 KB).  Deterministic: fixed order, no atomics.
#ifndef REDUCE_WIDE_MAX_P
#define REDUCE_WIDE_MAX_P 64   // (0: the 64-element form at every record count: A/B)
#endif
template <int NT>
__device__ __forceinline__ void reduce_slab_wide(int wg, int P, int D, const float* __restrict__ part,
                                                 float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dwg,
                                                 float* __restrict__ dcg, int accumulate, const SgdFuse& sg) {
  constexpr int NW = NT / 64;
  const int PSTRIDE = D * D + 2 * D + 4;
  const int total = D * D + 2 * D + 1;
  __shared__ __attribute__((aligned(16))) float redw[NW][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = wg * 256 + lane * 4;   // PSTRIDE is a multiple of 4 and the row is padded to it
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (e0 < PSTRIDE) {
    const int per = (P + NW - 1) / NW;
    const int p0 = wave * per, p1 = min(P, p0 + per);
    for (int p = p0; p < p1; p += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(part + (size_t)min(p + u, p1 - 1) * PSTRIDE + e0);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + u < p1) a += t[u];
    }
  }
  *(f32x4*)&redw[wave][lane * 4] = a;
  __syncthreads();
  const int e = wg * 256 + (int)threadIdx.x;
  if (threadIdx.x < 256 && e < total) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += redw[w][threadIdx.x];
    float* dst;
    if (e < D * D) dst = dW + e;
    else if (e < D * D + D) dst = db + (e - D * D);
    else if (e < D * D + 2 * D) dst = dwg + (e - D * D - D);
    else dst = dcg;
    *dst = accumulate ? (*dst + s) : s;
    if (sg.param) sgd_apply(sg, (int)(dst - sg.grad), s);   // this element's gradient is final: step it right here
  }
}

// Extra workgroups of a launch that carries the optimizer step (cgcn_sgd_fuse): every arena element whose gradient an
// EARLIER launch finished, i.e. all but this layer's own dW / db / dwg / dcg (the reduce slabs step those as they finish).
__device__ __forceinline__ void sgd_other_elements(const SgdFuse& sg, int block, int D, const float* dW, const float* db,
                                                   const float* dwg, const float* dcg) {
  const int i = block * (int)blockDim.x + (int)threadIdx.x;
  if (i == 0 && sg.rng_state) sg.rng_state[1] += 1ull;
  if (i < sg.count) {
    const float* gp = sg.grad + i;
    const bool mine = (gp >= dW && gp < dW + D * D) || (gp >= db && gp < db + D) || (gp >= dwg && gp < dwg + D) || gp == dcg;
    if (!mine) sgd_apply(sg, i, *gp);
  }
}

// Second stage: out[e] (+)= sum_p part[p][e], fixed order => deterministic.  (Stand-alone form: the first layer's
// backward when nobody wants d loss / d features, so there is no gather launch to ride in.)
__global__ __launch_bounds__(256) void k_reduce_partials(int P, int D, const float* __restrict__ part,
                                                         float* __restrict__ dW, float* __restrict__ db,
                                                         float* __restrict__ dwg, float* __restrict__ dcg,
                                                         int accumulate, SgdFuse sg, int reduce_slabs) {
  if ((int)blockIdx.x >= reduce_slabs) {
    sgd_other_elements(sg, (int)blockIdx.x - reduce_slabs, D, dW, db, dwg, dcg);
    return;
  }
  reduce_slab<256>(blockIdx.x, P, D, part, dW, db, dwg, dcg, accumulate, sg);
}

// ------------------------------------------------------------------------------------------
// Feature-sliced aggregation (the standalone gathers: k_aggregate_sliced, k_bwd_sliced).
//
// One node's payload is S*D*4 bytes (1 KiB at S = 2, D = 128), a chromosome's table 6..30 MB: several times an XCD's
// 4 MiB L2, so a gather of whole rows misses L2 about half the time and runs at the fabric / Infinity-Cache rate
// (measured: 7.9 TB/s at n = 29 k against 20 TB/s at n = 5.8 k, tools/micro/sliced_gather.hip).  Here the table is cut
// into S*D/32 column slices of one 128-byte line per node -- (strand s, features 32q .. 32q+31) -- and workgroup b
// aggregates slice b % NSL for 64 rows: workgroups are dealt round-robin over the 8 XCDs, so every XCD (every private
// L2) only ever touches its own slice(s), n * 128 B = 0.7..3.7 MB: L2 resident after the first touch.  The column
// indices are re-read once per slice (2 MB x 8), which is the price.
//   one wave = 8 rows x 8 lanes; a lane holds 16 B of its row's slice; an 8-lane group walks its own row's
//   neighbour list 8 at a time: one coalesced 32-byte load of column indices (+ values), broadcast inside the
//   group with ds_swizzle, 8 line loads in flight per group (64 per wave), summed in list order.
//   Waves that own a row longer than SLICED_HUB neighbours (Hi-C hubs) switch to a cooperative walk: the 8 groups
//   share each of the wave's rows (64 neighbours per step) and are summed in a fixed butterfly order, so one hub
//   costs len/8 instead of len serial steps and the result stays bit-reproducible.
// ------------------------------------------------------------------------------------------
#ifndef SLICED_HUB
#define SLICED_HUB 192
#endif
#ifndef SLICED_COST_MODEL
#define SLICED_COST_MODEL 1
#endif
#ifndef SLICED_COOP_OVERHEAD
#define SLICED_COOP_OVERHEAD 3
#endif
// lane (8*group + U) of every 8-lane group: ds_swizzle bit-mask mode, lane' = (lane & 0x18) | U inside each 32
template <int U>
__device__ __forceinline__ int group8_bcast(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x18 | (U << 5)); }

// index (and value) broadcasts of one chunk of 8 neighbours: all eight ds_swizzle issued back to back into eight
// registers (interleaved with the loads the compiler reused one register and waited lgkmcnt(0) after every swizzle:
// eight serial LDS round trips per chunk), 32-bit byte offsets from the uniform table base (global_load ... saddr form:
// one v_lshl_add_u32 per address instead of two 64-bit operations).  tools/micro/sliced_gather.hip "v2": -1 ... -2.5 %.
template <int U, bool HAS_VAL>
struct SlicedBcast {
  static __device__ __forceinline__ void run(unsigned (&off)[8], float (&w)[8], int myc, float myv, unsigned rowsh,
                                             unsigned lane_off) {
    SlicedBcast<U - 1, HAS_VAL>::run(off, w, myc, myv, rowsh, lane_off);
    off[U - 1] = ((unsigned)group8_bcast<U - 1>(myc) << rowsh) + lane_off;
    if (HAS_VAL) w[U - 1] = __int_as_float(group8_bcast<U - 1>(__float_as_int(myv)));
  }
};
template <bool HAS_VAL>
struct SlicedBcast<0, HAS_VAL> {
  static __device__ __forceinline__ void run(unsigned (&)[8], float (&)[8], int, float, unsigned, unsigned) {}
};

// neighbours k, k + step, ... of [k0, k1) in chunks of 8 (k0 already offset by the caller for the cooperative walk);
// base = the table (uniform), lane_off = byte offset of this lane's 16 bytes of row 0 of its slice, rowsh = log2(bytes per
// (strand, node) row).
// IT: column index type -- int32, or uint16 when the graph has at most 65 536 columns (every chromosome of the genome):
// the index list is re-read once per column slice (8 x 2 MB per launch at int32), and the 16-bit list halves that:
// 0 ... -7 % per launch (profiles/r03_u16_index_experiment.txt), sums bit-identical.
template <bool HAS_VAL, typename IT>
__device__ __forceinline__ f32x4 sliced_walk(const IT* __restrict__ col, const float* __restrict__ val, int k0, int k1,
                                             int step, const char* __restrict__ base, unsigned lane_off, unsigned rowsh, int j) {
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k = k0; k < k1; k += step) {
    const int kk = min(k + j, k1 - 1);   // ragged tail: re-read the last neighbour (an L1 hit), add a selected zero
    const int myc = (int)col[kk];
    const float myv = HAS_VAL ? val[kk] : 0.f;
    unsigned off[8];
    float w[8];
    SlicedBcast<8, HAS_VAL>::run(off, w, myc, myv, rowsh, lane_off);
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(base + (size_t)off[u]);
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 x = HAS_VAL ? w[u] * t[u] : t[u];
      acc += (k + u < k1) ? x : zero;
    }
  }
  return acc;
}

// sum over the neighbours of this lane's group's row (k0, k1: that row's range; equal inside a group)
template <bool HAS_VAL, typename IT>
__device__ __forceinline__ f32x4 sliced_row_sum(const IT* __restrict__ col, const float* __restrict__ val, int k0, int k1,
                                                const char* __restrict__ base, unsigned lane_off, unsigned rowsh, int lane) {
  const int g = lane >> 3, j = lane & 7;
#if SLICED_COST_MODEL
  // per-row walk: every group steps until the wave's longest row is done (8 neighbours per row and step); cooperative
  // walk: the rows one after the other, 64 neighbours per step, plus eight butterflies.  Take the cheaper one: a wave
  // whose rows are 5, 5, ..., 5, 200 neighbours long walks 11 steps together instead of 25 apart.
  int mx = k1 - k0, sm = (k1 - k0 + 63) >> 6;
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    mx = max(mx, __shfl_xor(mx, o, WAVE));
    sm += __shfl_xor(sm, o, WAVE);
  }
  if (sm + SLICED_COOP_OVERHEAD >= ((mx + 7) >> 3)) return sliced_walk<HAS_VAL, IT>(col, val, k0, k1, 8, base, lane_off, rowsh, j);
#else
  if (!__any(k1 - k0 > SLICED_HUB)) return sliced_walk<HAS_VAL, IT>(col, val, k0, k1, 8, base, lane_off, rowsh, j);
#endif
  f32x4 mine = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int rr = 0; rr < 8; ++rr) {   // wave-uniform: every group helps with row rr of the wave
    const int a0 = __shfl(k0, rr * 8, WAVE), a1 = __shfl(k1, rr * 8, WAVE);
    f32x4 acc = sliced_walk<HAS_VAL, IT>(col, val, a0 + g * 8, a1, 64, base, lane_off, rowsh, j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] += __shfl_xor(acc[e], 8, WAVE);
      acc[e] += __shfl_xor(acc[e], 16, WAVE);
      acc[e] += __shfl_xor(acc[e], 32, WAVE);
    }
    if (g == rr) mine = acc;
  }
  return mine;
}

// The 64 rows of a workgroup's tile, rows of more than SLICED_SUPER neighbours (top-K Hi-C hubs: thousands) included.
// One CU pulls at most 64 B/clk through its L1, and one wave walking a 10 000-neighbour row exposes an L2 round trip
// per 64 neighbours: such rows are walked by all 8 waves of the workgroup, wave w taking the w-th eighth of the list
// (64-aligned), combined through LDS in wave order (bit-reproducible).
// Every wave reads the tile's 65 row pointers itself (lane l: row 64*tile + l; one coalesced load of lines its
// neighbours read too) and ballots the super rows, so all 8 waves know the same mask without a barrier: tiles without a
// super row -- every tile of a regular graph -- pay nothing.
#ifndef SLICED_SUPER
#define SLICED_SUPER 768
#endif
struct SlicedTile {
  int t0, t1;                 // lane l: neighbour range of the tile's l-th row (empty past n)
  unsigned long long super;   // wave-uniform (and workgroup-uniform): rows of the tile with more than SLICED_SUPER neighbours
  int i;                      // this lane's own row: the (8*wave + lane/8)-th of the tile (n if past the end)
  int k0, k1;                 // its neighbour range
  int g0;                     // first row of the 64-row group the tile holds (a tile's rows are a permutation of one group)
};
// order (may be null): the rows in the order the tiles take them (cgcn_graph_aux::row_order): position p -> tile p / 64,
// wave (p % 64) / 8.  The engine sorts the rows of every 64-row tile by length, so that the 8 rows a wave walks side by
// side are about equally long (a wave steps until its longest row is done) while a tile keeps its rows.
__device__ __forceinline__ SlicedTile sliced_tile(const int* __restrict__ rowptr, const int* __restrict__ order, int n,
                                                  int tile, int wave, int lane) {
  SlicedTile t;
  const int r = tile * 64 + lane;
  int row = n;
  t.t0 = t.t1 = 0;
  if (r < n) {
    row = order ? order[r] : r;
    t.t0 = rowptr[row];
    t.t1 = rowptr[row + 1];
  }
  t.super = __ballot(t.t1 - t.t0 > SLICED_SUPER);
  t.g0 = __builtin_amdgcn_readfirstlane(row) & ~63;   // (position 64 tile always holds a row)
  const int mine = wave * 8 + (lane >> 3);
  t.i = __shfl(row, mine, WAVE);
  t.k0 = __shfl(t.t0, mine, WAVE);
  t.k1 = __shfl(t.t1, mine, WAVE);
  return t;
}
// the tile's sums when it holds super rows (t.super != 0: every wave of the workgroup takes this branch)
template <bool HAS_VAL, typename IT>
__device__ __forceinline__ f32x4 sliced_super_sum(const IT* __restrict__ col, const float* __restrict__ val, const SlicedTile& t,
                                                  const char* __restrict__ base, unsigned lane_off, unsigned rowsh, int lane,
                                                  int wave) {
  __shared__ float s_part[8][32];
  __shared__ float s_res[64][32];
  const int g = lane >> 3, j = lane & 7;
  // the super rows first, results parked in LDS (nothing but the row ranges stays live across these walks)
  unsigned long long m = t.super;
  while (m) {
    const int r = __builtin_ctzll(m);
    m &= m - 1;
    const int a0 = __builtin_amdgcn_readlane(t.t0, r), a1 = __builtin_amdgcn_readlane(t.t1, r);
    const int seg = (((a1 - a0 + 7) >> 3) + 63) & ~63;
    const int w0 = a0 + wave * seg, w1 = min(a1, w0 + seg);
    f32x4 acc = sliced_walk<HAS_VAL, IT>(col, val, w0 + g * 8, w1, 64, base, lane_off, rowsh, j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] += __shfl_xor(acc[e], 8, WAVE);
      acc[e] += __shfl_xor(acc[e], 16, WAVE);
      acc[e] += __shfl_xor(acc[e], 32, WAVE);
    }
    if (lane < 8) *(f32x4*)&s_part[wave][lane * 4] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
      float x = s_part[0][threadIdx.x];
#pragma unroll
      for (int ww = 1; ww < 8; ++ww) x += s_part[ww][threadIdx.x];
      s_res[r][threadIdx.x] = x;
    }
    __syncthreads();
  }
  // the ordinary rows (super rows walk an empty range here and pick their sum up from LDS)
  const bool super = (t.k1 - t.k0) > SLICED_SUPER;
  f32x4 mine = sliced_row_sum<HAS_VAL, IT>(col, val, t.k0, super ? t.k0 : t.k1, base, lane_off, rowsh, lane);
  if (super) mine = *(const f32x4*)&s_res[wave * 8 + g][j * 4];
  return mine;
}

// workgroup b -> (column slice, 64-row tile).  Up to 8 slices: slice = b mod NSL, so XCD x (workgroups b = x mod 8) owns
// slice x.  16 slices (S*D = 512 floats): two passes over the tiles, XCD x works on slice x in the first half of the
// grid and on slice x + 8 in the second, so that one slice (not two) is hot in its L2 at a time.
// Tuning: non-temporal accesses on the sliced kernels' once-touched streams, so that they compete less
// with the L2-resident slice of the gathered table.  Bit 0: H store of k_aggregate_sliced; bit 1: the (1-g) dXn operand
// of k_bwd_sliced; bit 2: its dX store.
#ifndef SLICED_NT
#define SLICED_NT 7   // measured (profiles/r04_sliced_nt_experiment.txt): genome epoch 4.67 -> 4.63 ms, traffic beyond L2 -6 %
#endif
__device__ __forceinline__ f32x4 ld_stream4(const float* p) {
  if (SLICED_NT & 2) return __builtin_nontemporal_load((const f32x4*)p);
  return *(const f32x4*)p;
}
template <int NSL>
__device__ __forceinline__ void sliced_block(int b, int tiles, int& slice, int& tile) {
  if (NSL <= 8) {
    slice = b % NSL;
    tile = b / NSL;
  } else {
    const int per = 8 * tiles, pass = b / per, rem = b - pass * per;
    slice = (rem & 7) + 8 * pass;
    tile = rem >> 3;
  }
}

constexpr int BAND_W = 7;   // the reference's constant_range (utils/util_methods.py:147)
// ------------------------------------------------------------------------------------------
// BP ("band plus", adj_type 'both': utils/util_methods.py:168-171 -- Hi-C + the +-7 band + I, values 1 or 2, not
// binarised).  The merged CSR walks 15 band entries per row on top of the Hi-C ones, every one a 128-byte line through the
// vL1D like any other neighbour, with explicit values (int32 indices, 6 waves per SIMD): 1.45x the line loads of 'hic',
// 1.5x its time.  But the band half of the sum needs no indices and no L1: with cgcn_graph_aux::bp_* the CSR handed to
// these kernels holds only the UNIT entries that are not the band's own (Hi-C entries outside the band, and the second
// unit of the value-2 entries inside it), and the band + I part is added from a (64 + 14)-row x 128-byte window of the
// table that the workgroup stages in LDS once -- 78 line loads per tile instead of 15 x 64 -- so 'both' costs what 'hic'
// costs plus the staging.  sum_j w_ij x_j = sum_{unit entries} x_j + sum_{|j - i| <= 7} x_j, exact in real arithmetic;
// fp32: the unit entries in list order from zero, then the window rows in ascending order from zero, then their sum.
// ------------------------------------------------------------------------------------------
constexpr int BANDPLUS_CHUNKS = (64 + 2 * BAND_W) * 8;   // [window row][16-byte chunk of the slice's 128-byte line]
// stage rows [g0 - BW, g0 + 64 + BW) of this workgroup's slice (lane_el_base: element offset of chunk 0 of row 0); zeros
// outside the strand.  512 threads; followed by a workgroup barrier in the caller.
__device__ __forceinline__ void bandplus_stage(f32x4* __restrict__ bt, const float* __restrict__ T, size_t slice_el, int n, int g0, int D) {
  constexpr int NCH = BANDPLUS_CHUNKS;
#pragma unroll
  for (int u = 0; u < (NCH + 511) / 512; ++u) {
    const int idx = (int)threadIdx.x + 512 * u;
    const int j = g0 - BAND_W + (idx >> 3);
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (idx < NCH && j >= 0 && j < n) v = *(const f32x4*)&T[slice_el + (size_t)j * D + (idx & 7) * 4];
    if (idx < NCH) bt[idx] = v;
  }
}
__device__ __forceinline__ f32x4 bandplus_sum(const f32x4* __restrict__ bt, int i, int g0, int lane) {
  const int li = i - g0, c = lane & 7;   // window rows i - BW .. i + BW sit at tile rows li .. li + 2 BW
  f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k <= 2 * BAND_W; ++k) a += bt[(li + k) * 8 + c];
  return a;
}

// H = diag(rs) Ahat X, [S, n, D] -> [S, n, D]  (grid: NSL * ceil(n / 64) workgroups of 512)
template <int S, int D, bool HAS_VAL, typename IT = int, bool BP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(HAS_VAL ? 6 : 8))) void k_aggregate_sliced(int n, const int* __restrict__ rowptr, const IT* __restrict__ col,
                                                          const float* __restrict__ val, const float* __restrict__ rs,
                                                          const float* __restrict__ X, float* __restrict__ H,
                                                          const int* __restrict__ order,
                                                          unsigned long long* __restrict__ zero_words, int zero_count) {
  static_assert(!(BP && HAS_VAL), "the band-plus CSR holds unit entries");
  constexpr int NSL = S * D / 32, QPR = D / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the accumulators of the statistics the row-local launch behind this one adds to (STAT_ACC_*): zeroed here, by the launch
  // that precedes it in the same call, so that no launch of its own is needed
  const int zblocks = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  if (zero_words && (int)blockIdx.x < zblocks)
    for (int i = (int)blockIdx.x * 512 + (int)threadIdx.x; i < zero_count; i += zblocks * 512) zero_words[i] = 0ull;
  int slice, tile;
  sliced_block<NSL>(blockIdx.x, (n + 63) / 64, slice, tile);
  const size_t slice_el = (size_t)(slice / QPR) * n * D + (slice % QPR) * 32;
  const size_t lane_el = slice_el + (lane & 7) * 4;
  const SlicedTile t = sliced_tile(rowptr, order, n, tile, wave, lane);
  __shared__ f32x4 bt[BP ? BANDPLUS_CHUNKS : 1];
  if (BP) {
    bandplus_stage(bt, X, slice_el, n, t.g0, D);
    __syncthreads();
  }
  const int i = t.i;
  const float sc = (i < n && rs) ? rs[i] : 1.f;
  const unsigned lane_off = (unsigned)(lane_el * 4), rowsh = D == 128 ? 9u : 10u;
  f32x4 acc = !t.super ? sliced_row_sum<HAS_VAL, IT>(col, val, t.k0, t.k1, (const char*)X, lane_off, rowsh, lane)
                       : sliced_super_sum<HAS_VAL, IT>(col, val, t, (const char*)X, lane_off, rowsh, lane, wave);
  if (i < n) {
    if (BP) acc += bandplus_sum(bt, i, t.g0, lane);
    if (SLICED_NT & 1) __builtin_nontemporal_store(acc * sc, (f32x4*)&H[lane_el + (size_t)i * D]);
    else *(f32x4*)&H[lane_el + (size_t)i * D] = acc * sc;
  }
}

// Horizontal fusion: the workgroups past the gather tiles of the layer backward's last launch (k_bwd_sliced, k_bwd_band)
// do the (independent) second-stage sum of the row-local kernel's partials, so that reduction costs no launch of its own
// and overlaps the gather's tail.  `extra` = index of the workgroup past the gather tiles; 512 threads.
// EXT: the head slabs stage through the caller's LDS (`scratch`: >= 4 * 32 * 17 doubles) instead of static arrays of their own.
template <int S, int D, bool EXT = false>
__device__ __forceinline__ void bwd_riders(int extra, int n, int P, const float* __restrict__ part, float* __restrict__ dW,
                                           float* __restrict__ db, float* __restrict__ dwg, float* __restrict__ dcg,
                                           int accumulate, const SgdFuse& sg, int reduce_slabs, const HeadApply& hp,
                                           int head_slabs, void* scratch = nullptr) {
  if (extra < reduce_slabs) {   // (the host counts wide slabs when P <= REDUCE_WIDE_MAX_P: layer_bwd_impl)
    if (P <= REDUCE_WIDE_MAX_P) reduce_slab_wide<512>(extra, P, D, part, dW, db, dwg, dcg, accumulate, sg);
    else reduce_slab<512>(extra, P, D, part, dW, db, dwg, dcg, accumulate, sg);
    return;
  }
  extra -= reduce_slabs;
  if (extra < head_slabs) {
    // the head's deferred second stage (dW_out / db_out slabs, BatchNorm column sums): the row-local launch of the last
    // layer used to carry these 266 workgroups, but every workgroup of that kernel needs the whole LDS of a CU, so they
    // ran as a second wave behind the 256 resident ones (+5 us); here they share CUs with the gather workgroups
    const int wslabs = (hp.hf_CP * D + hp.hf_CP) / 64;
    if (extra < wslabs)
      head_finalize_slab<512, EXT>(extra, hp.hf_P, D, hp.hf_C, hp.hf_CP, hp.hf_part, hp.hf_dWout, hp.hf_dbout, hp.hf_accumulate, hp.dloss, scratch);
    else
      head_stats_finalize<512, EXT, 4>(extra - wslabs, hp.hf_P, n, S, D, hp.hf_CP, hp.hf_part, hp.hf_dbn_w, hp.hf_dbn_b, nullptr,
                                       hp.hf_accumulate, hp.dloss, scratch);
    return;
  }
  // fused optimizer step (cgcn_sgd_fuse) for every arena element whose gradient an EARLIER launch finished: all but
  // this layer's own dW / db / dwg / dcg, which the slabs above step as they finish them
  sgd_other_elements(sg, extra - head_slabs, D, dW, db, dwg, dcg);
}

// ------------------------------------------------------------------------------------------
// k_bwd_sliced: dX = mask * ((1-g) dXn + Ahat^T dHs), dHs = diag(rs) dU W^T from k_bwd_rowlocal: the aggregation over
// the transposed adjacency with an element-wise epilogue, feature-sliced (above).
// mask: the dropout the PREVIOUS layer applied to this layer's input (stream_id of that layer).
// ------------------------------------------------------------------------------------------
template <int S, int D, bool HAS_VAL, typename IT = int, bool BP = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(HAS_VAL ? 6 : 8))) void k_bwd_sliced(int n, const int* __restrict__ rowptr, const IT* __restrict__ col,
                                                    const float* __restrict__ val, const float* __restrict__ dHs,
                                                    const float* dXn, const float* __restrict__ gate, float* dX,
                                                    float keep_scale, uint32_t thresh,
                                                    const unsigned long long* __restrict__ rng_state,
                                                    uint32_t stream_id, int gather_blocks, int P,
                                                    const float* __restrict__ part, float* __restrict__ dW,
                                                    float* __restrict__ db, float* __restrict__ dwg,
                                                    float* __restrict__ dcg, int accumulate, SgdFuse sg, int reduce_slabs,
                                                    const int* __restrict__ order, HeadApply hp, int head_slabs) {
  // BP: one LDS pool serves the band window of a gather workgroup and the staging of a rider workgroup (never both)
  constexpr int RIDER_STAGE = 4 * (512 / HEAD_STAT_COLS) * (HEAD_STAT_COLS + 1) * (int)sizeof(double) / 16;
  __shared__ f32x4 bt[BP ? (BANDPLUS_CHUNKS > RIDER_STAGE ? BANDPLUS_CHUNKS : RIDER_STAGE) : 1];
  if ((int)blockIdx.x >= gather_blocks) {
#ifndef BSX_NORIDERS   // (decomposition build, profiles/r06_bwd_sliced_gap.txt: the riders return at once -- gradients are garbage)
    bwd_riders<S, D, BP>((int)blockIdx.x - gather_blocks, n, P, part, dW, db, dwg, dcg, accumulate, sg, reduce_slabs, hp, head_slabs, bt);
#endif
    return;
  }
  static_assert(!(BP && HAS_VAL), "the band-plus CSR holds unit entries");
  constexpr int NSL = S * D / 32, QPR = D / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int slice, tile;
  sliced_block<NSL>(blockIdx.x, (n + 63) / 64, slice, tile);
  const int s = slice / QPR;
  const size_t slice_el = (size_t)s * n * D + (slice % QPR) * 32;
  const size_t lane_el = slice_el + (lane & 7) * 4;
  const SlicedTile t = sliced_tile(rowptr, order, n, tile, wave, lane);
  if (BP) {   // the band + I half of Ahat^T dHs (Ahat is symmetric) from a window of the table staged once per workgroup
    bandplus_stage(bt, dHs, slice_el, n, t.g0, D);
    __syncthreads();
  }
  const int i = t.i;
  const unsigned lane_off = (unsigned)(lane_el * 4), rowsh = D == 128 ? 9u : 10u;
  f32x4 res = (f32x4){0.f, 0.f, 0.f, 0.f}, acc;
  if (!t.super) {
    // the (1-g) dXn term first: its loads are in flight during the walk
#ifndef BSX_NODXN      // (decomposition build: no (1-g) dXn operand)
    if (i < n) res = ld_stream4(&dXn[lane_el + (size_t)i * D]) * (1.f - gate[(size_t)s * n + i]);
#endif
    acc = sliced_row_sum<HAS_VAL, IT>(col, val, t.k0, t.k1, (const char*)dHs, lane_off, rowsh, lane);
  } else {
    acc = sliced_super_sum<HAS_VAL, IT>(col, val, t, (const char*)dHs, lane_off, rowsh, lane, wave);
    if (i < n) res = ld_stream4(&dXn[lane_el + (size_t)i * D]) * (1.f - gate[(size_t)s * n + i]);
  }
  if (BP && i < n) acc += bandplus_sum(bt, i, t.g0, lane);
  if (i >= n) return;
  const size_t g_off = lane_el + (size_t)i * D;
  f32x4 o = res + acc;
  if (thresh) {
    const uint32_t key = dropout_key(rng_state, stream_id);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? o[e] * keep_scale : 0.f;
  }
#ifdef BSX_NOSTORE     // (decomposition build: the result leaves only if it is a number no input produces)
  if (o[0] != 1.2345e33f) return;
#endif
  if (SLICED_NT & 4) __builtin_nontemporal_store(o, (f32x4*)&dX[g_off]);
  else *(f32x4*)&dX[g_off] = o;
}

// ------------------------------------------------------------------------------------------
// Band graphs (adj_type 'constant': utils/util_methods.py:137-150 -- the +-7 diagonals plus I, row-normalised; the band
// half of 'both' too).  Row i of Ahat is the contiguous window [max(0, i - BW), min(n - 1, i + BW)] with unit values, so
// the aggregation is a (2 BW + 1)-row sliding-window sum: a STREAM, not a gather.  The CSR route walks the same rows as
// 15 index-driven line loads per output line (15x the table through the vL1D, index list on top: the random-gather
// rate); here a workgroup stages the R + 2 BW rows its R-row tile needs ONCE in LDS (zero rows past either end of the
// strand) and every output is 15 LDS reads: the table moves through the memory system once (+ 2 BW / R halo rows, served
// by the neighbouring tile's L2 -- tiles are dealt to the XCDs in contiguous ranges), which prices the launch at the
// algorithmic bytes of SURVEY 8(d): read X, write H.
//   tile: 512 threads; a thread owns one 16-byte column chunk (CH = D / 4 chunks per row) of RPT = R / (512 / CH)
//   consecutive rows; loads: all of a thread's tile rows in flight at once, ds_write_b128, ONE barrier, then RPT + 2 BW
//   ds_read_b128 (conflict-free: a 16-lane read group covers 256 contiguous bytes) and the window sums in ascending row
//   order starting from zero -- the CSR kernels' list order, so the band route and the CSR route give the same bits.
// k_band_aggregate: H = diag(rs) Ahat X.   k_bwd_band: dX = mask * ((1-g) dXn + Ahat^T dHs) (Ahat is symmetric; dHs comes
// pre-scaled), with k_bwd_sliced's riders (second-stage sums, the head's slabs, the fused SGD step) in trailing workgroups.
// ------------------------------------------------------------------------------------------
#ifndef BAND_R
#define BAND_R 32
#endif
template <int D, int R>
struct BandGeo {
  static constexpr int CH = D / 4, RL = 512 / CH, RPT = R / RL, ROWS = R + 2 * BAND_W, NL = (ROWS * CH + 511) / 512;
  static constexpr int TROWS = NL * RL;   // rows of the LDS tile: ROWS rounded up to whole staging passes (no predicated store)
  static_assert(RL * RPT == R && RPT >= 1, "tile height is a multiple of the row lanes");
};
// window sums of tile `tile` of one strand's table T [n][D]: acc[p] = sum_{j in window(r0 + rl RPT + p)} T[j][4 c .. 4 c + 3]
template <int D, int R>
__device__ __forceinline__ void band_tile_sums(const float* __restrict__ T, int n, int r0, float* __restrict__ tile,
                                               f32x4 (&acc)[BandGeo<D, R>::RPT]) {
  using G = BandGeo<D, R>;
  // branch-free staging: every load has a valid (clamped) address and is issued before the first wait; rows past either
  // end of the strand become zeros by a select
  f32x4 v[G::NL];
  const int c0 = (int)threadIdx.x % G::CH, q0 = (int)threadIdx.x / G::CH;
#pragma unroll
  for (int u = 0; u < G::NL; ++u) {
    const int j = r0 - BAND_W + q0 + (512 / G::CH) * u;
    v[u] = *(const f32x4*)&T[(size_t)min(max(j, 0), n - 1) * D + 4 * c0];
  }
#pragma unroll
  for (int u = 0; u < G::NL; ++u) {
    const int row = q0 + (512 / G::CH) * u;
    const int j = r0 - BAND_W + row;
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    *(f32x4*)&tile[row * D + 4 * c0] = (j >= 0 && j < n) ? v[u] : z;   // (rows >= ROWS of the padded tile: written, never read)
  }
  __syncthreads();
  const int c = (int)threadIdx.x % G::CH, rl = (int)threadIdx.x / G::CH;
  // one pass down the thread's RPT + 2 BW tile rows; row k belongs to the windows of outputs p with 0 <= k - p <= 2 BW, and
  // every output adds its rows in ascending order (few registers live: the kernel runs 8 waves per SIMD)
#pragma unroll
  for (int p = 0; p < G::RPT; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ col = &tile[rl * G::RPT * D + 4 * c];
  constexpr int NK = G::RPT + 2 * BAND_W, KB = 4;   // KB rows per batch of LDS reads (all in flight, then added)
#pragma unroll
  for (int k0 = 0; k0 < NK; k0 += KB) {
    f32x4 w[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u)
      if (k0 + u < NK) w[u] = *(const f32x4*)&col[(k0 + u) * D];
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int p = 0; p < G::RPT; ++p)
        if (k0 + u < NK && k0 + u - p >= 0 && k0 + u - p <= 2 * BAND_W) acc[p] += w[u];
    // the sums of this batch happen HERE (left alone the compiler sinks them into the callers' store branches and keeps
    // every row of the column live: 64+ registers, spills at 8 waves)
#pragma unroll
    for (int p = 0; p < G::RPT; ++p) asm volatile("" : "+v"(acc[p]));
  }
}

template <int S, int D, int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8))) void k_band_aggregate(int n, const float* __restrict__ rs, const float* __restrict__ X,
                                                        float* __restrict__ H, unsigned long long* __restrict__ zero_words,
                                                        int zero_count) {
  using G = BandGeo<D, R>;
  // (the statistics accumulators of the row-local launch behind this one: zeroed here, like k_aggregate_sliced does)
  const int zblocks = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  if (zero_words && (int)blockIdx.x < zblocks)
    for (int i = (int)blockIdx.x * 512 + (int)threadIdx.x; i < zero_count; i += zblocks * 512) zero_words[i] = 0ull;
  const int tiles = (n + R - 1) / R;
  const int b = xcd_contiguous((int)blockIdx.x, S * tiles);
  const int s = b / tiles, r0 = (b - s * tiles) * R;
  __shared__ __attribute__((aligned(16))) float tile[G::TROWS * D];
  f32x4 acc[G::RPT];
  band_tile_sums<D, R>(X + (size_t)s * n * D, n, r0, tile, acc);
  const int c = (int)threadIdx.x % G::CH, rl = (int)threadIdx.x / G::CH;
#pragma unroll
  for (int p = 0; p < G::RPT; ++p) {
    const int i = r0 + rl * G::RPT + p;
    if (i < n) {
      const f32x4 o = acc[p] * (rs ? rs[i] : 1.f);
      float* dst = &H[((size_t)s * n + i) * D + 4 * c];
      if (SLICED_NT & 1) __builtin_nontemporal_store(o, (f32x4*)dst);
      else *(f32x4*)dst = o;
    }
  }
}

template <int S, int D, int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8))) void k_bwd_band(int n, const float* __restrict__ dHs, const float* dXn,
                                                  const float* __restrict__ gate, float* dX, float keep_scale, uint32_t thresh,
                                                  const unsigned long long* __restrict__ rng_state, uint32_t stream_id,
                                                  int gather_blocks, int P, const float* __restrict__ part,
                                                  float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dwg,
                                                  float* __restrict__ dcg, int accumulate, SgdFuse sg, int reduce_slabs,
                                                  HeadApply hp, int head_slabs) {
  using G = BandGeo<D, R>;
  __shared__ __attribute__((aligned(16))) float tile[G::TROWS * D];
  static_assert(sizeof(tile) >= 4 * (512 / HEAD_STAT_COLS) * (HEAD_STAT_COLS + 1) * sizeof(double), "rider staging fits the tile");
  if ((int)blockIdx.x >= gather_blocks) {
    bwd_riders<S, D, true>((int)blockIdx.x - gather_blocks, n, P, part, dW, db, dwg, dcg, accumulate, sg, reduce_slabs, hp, head_slabs, tile);
    return;
  }
  const int tiles = (n + R - 1) / R;
  const int b = xcd_contiguous((int)blockIdx.x, gather_blocks);
  const int s = b / tiles, r0 = (b - s * tiles) * R;
  const int c = (int)threadIdx.x % G::CH, rl = (int)threadIdx.x / G::CH;
  // the (1-g) dXn term first: its loads are in flight during the tile's staging (dXn may be dX itself: every thread reads
  // exactly the elements it writes)
  f32x4 res[G::RPT];
#pragma unroll
  for (int p = 0; p < G::RPT; ++p) {
    const int i = r0 + rl * G::RPT + p;
    res[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n) res[p] = ld_stream4(&dXn[((size_t)s * n + i) * D + 4 * c]) * (1.f - gate[(size_t)s * n + i]);
  }
  f32x4 acc[G::RPT];
  band_tile_sums<D, R>(dHs + (size_t)s * n * D, n, r0, tile, acc);
  const uint32_t key = thresh ? dropout_key(rng_state, stream_id) : 0u;
#pragma unroll
  for (int p = 0; p < G::RPT; ++p) {
    const int i = r0 + rl * G::RPT + p;
    if (i >= n) continue;
    const size_t g_off = ((size_t)s * n + i) * D + 4 * c;
    f32x4 o = res[p] + acc[p];
    if (thresh) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = dropout_keep(key, (uint32_t)(g_off + e), thresh) ? o[e] * keep_scale : 0.f;
    }
    if (SLICED_NT & 4) __builtin_nontemporal_store(o, (f32x4*)&dX[g_off]);
    else *(f32x4*)&dX[g_off] = o;
  }
}

// ------------------------------------------------------------------------------------------
// k_sddmm: out[k] = sum_s < A[s,i,:], B[s,col[k],:] >  for every stored entry k of row i  (SURVEY.md row f4:
// the adjacency-saliency product of scripts/visualize.py:29-49 restricted to the sparsity pattern, instead
// of a dense n x n autograd gradient).  One wave per row; the row of A stays in registers, neighbour rows of
// B are fetched 8 at a time like the gather kernels.
// ------------------------------------------------------------------------------------------
template <int S, int D>
__global__ __launch_bounds__(256) void k_sddmm(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                               const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ out, int accumulate) {
  using G = Geo<S, D>;
  constexpr int NV = G::NV;
  constexpr unsigned ROWB = D * 4;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  unsigned lane_off[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) lane_off[v] = ((unsigned)G::strand(v, lane) * (unsigned)n * D + G::column(v, lane)) * 4u;
  for (int i = wave; i < n; i += nwaves) {
    f32x4 a[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      a[v] = *(const f32x4*)((const char*)A + (size_t)i * ROWB + lane_off[v]);
      if (G::HALF && lane >= 32) a[v] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the row is held once, by lanes 0-31
    }
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    for (int kb = k0; kb < k1; kb += WAVE) {
      const int cnt = min(WAVE, k1 - kb);
      const int myc = lane < cnt ? col[kb + lane] : 0;
      float mine = 0.f;  // lane u ends up with the dot product of neighbour kb + u
      for (int j = 0; j < cnt; j += 8) {
        f32x4 t[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const char* rowp = (const char*)B + (size_t)(unsigned)rl_i(myc, min(j + u, cnt - 1)) * ROWB;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const unsigned off = G::HALF ? (unsigned)((lane & 31) * 16) : lane_off[v];
            t[u][v] = *(const f32x4*)(rowp + off);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float d = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) d += a[v][e] * t[u][v][e];
          d = wave_sum(d);
          if (lane == j + u) mine = d;
        }
      }
      if (lane < cnt) out[kb + lane] = accumulate ? out[kb + lane] + mine : mine;   // (the saliency sums one product per layer)
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_saliency_rows: the row normalisation of the adjacency saliency (scripts/visualize.py:49-55) on the CSR pattern, in the
// reference's order of operations: v = |val * raw| per stored entry, divided by the row's sum (1 where the sum is 0), then by
// the row's maximum of those quotients (1 where it is 0).  Entries outside the pattern are zeros of the dense matrix the
// reference builds: they add nothing to a sum and never raise a maximum of non-negative numbers.  One wave per row, 64
// entries per step, two passes over the row (the second one hits the L1).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_saliency_rows(int n, const int* __restrict__ rowptr, const float* __restrict__ val,
                                                       const float* __restrict__ raw, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n; i += nwaves) {
    const int k0 = rowptr[i], k1 = rowptr[i + 1];
    float s = 0.f;
    for (int kb = k0; kb < k1; kb += WAVE) {
      const int k = kb + lane;
      s += k < k1 ? fabsf((val ? val[k] : 1.f) * raw[k]) : 0.f;
    }
    s = wave_sum(s);
    if (s == 0.f) s = 1.f;
    float m = 0.f;
    for (int kb = k0; kb < k1; kb += WAVE) {
      const int k = kb + lane;
      m = fmaxf(m, k < k1 ? fabsf((val ? val[k] : 1.f) * raw[k]) / s : 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if (m == 0.f) m = 1.f;
    for (int kb = k0; kb < k1; kb += WAVE) {
      const int k = kb + lane;
      if (k < k1) out[k] = fabsf((val ? val[k] : 1.f) * raw[k]) / s / m;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_sgd: torch.optim.SGD semantics on one flat buffer (utils/util_methods.py:14-19 builds
// SGD(lr, momentum=0.9, weight_decay=1e-6)):  d = g + wd p;  m = mu m + d;  p -= lr (nesterov ? d + mu m : m).
// A zero-initialised m reproduces torch's first step (m = d).  Also advances the dropout step counter:
// this is the last kernel of a train step.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sgd(int count, float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ m, float lr, float mu, float wd, int nesterov,
                                             float grad_scale, unsigned long long* __restrict__ rng_state) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && rng_state) rng_state[1] += 1ull;
  if (i >= count) return;
  const float pi = p[i];
  float d = g[i] * grad_scale + wd * pi;
  float upd = d;
  if (m) {
    const float b = mu * m[i] + d;
    m[i] = b;
    upd = nesterov ? d + mu * b : b;
  }
  p[i] = pi - lr * upd;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

static int check_shape(int n, int S, int d) {
  if (n < 0) return CGCN_ERR_BAD_ARG;
  if (!(S == 1 || S == 2) || !(d == 128 || d == 256)) return CGCN_ERR_UNSUPPORTED;
  if ((double)n * (double)S * (double)d * 4.0 >= 4294967296.0) return CGCN_ERR_UNSUPPORTED;  // 32-bit byte offsets
  return CGCN_OK;
}


#define DISPATCH_SDV(S_, d_, hasval_, CALL)                                      \
  do {                                                                           \
    if ((S_) == 1 && (d_) == 128) { if (hasval_) { CALL(1, 128, true); } else { CALL(1, 128, false); } } \
    else if ((S_) == 2 && (d_) == 128) { if (hasval_) { CALL(2, 128, true); } else { CALL(2, 128, false); } } \
    else if ((S_) == 1 && (d_) == 256) { if (hasval_) { CALL(1, 256, true); } else { CALL(1, 256, false); } } \
    else { if (hasval_) { CALL(2, 256, true); } else { CALL(2, 256, false); } }  \
  } while (0)

// Deeper gather batches when the gathered table is much larger than the L2s (see Gather::GU).
#ifndef DEEP_TABLE_BYTES
#define DEEP_TABLE_BYTES (12u << 20)
#endif
static inline bool pick_deep(int n, int S, int d) { return (double)n * S * d * 4.0 > (double)DEEP_TABLE_BYTES; }

// Tile height of the gather kernels: 16 MFMA rows (MB = 1: 8 nodes x 2 strands, or 16 nodes of one strand) at every
// size.  A 32-row variant (the MB template parameter) existed for large chromosomes; with the shallow gather batches
// it is slower even at 30 k nodes (layer forward 102.7 vs 92.5 us: more, smaller workgroups pack the CUs better),
// so it is no longer instantiated.
static inline int pick_mb(int, int) { return 1; }

// Events for fork/join edges between the main and the auxiliary stream.  A small pool reused round-robin;
// never destroyed (an event may still be referenced by a captured graph).  This is the only state the
// library keeps, and it holds no caller memory.
static hipEvent_t pooled_event() {
  // thread-safe: the pool is created once (C++11 static initialisation), the cursor is atomic.  Only reached when the
  // caller passes an auxiliary stream (opt-in, CHROMEGCN_AUX_STREAM); the default path creates no events at all.
  struct Pool {
    hipEvent_t ev[64];
    Pool() {
      for (auto& e : ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
    }
  };
  static Pool pool;
  static std::atomic<unsigned> next{0};
  return pool.ev[next.fetch_add(1u, std::memory_order_relaxed) & 63u];
}

static int dropout_args(float p, const unsigned long long* rng_state, float* keep_scale, uint32_t* thresh) {
  *keep_scale = 1.f;
  *thresh = 0u;
  if (p <= 0.f) return CGCN_OK;
  if (!rng_state || p >= 1.f) return CGCN_ERR_BAD_ARG;
  *keep_scale = 1.f / (1.f - p);
  *thresh = dropout_threshold(p);
  return CGCN_OK;
}

// H = diag(rs) Ahat X, feature-sliced: int32 column indices, or the 16-bit copy when given (implicit values only)
static void launch_aggregate_sliced(hipStream_t st, int gblocks, int n, int S, int d, const int32_t* rowptr, const int32_t* col,
                                    const uint16_t* col16, const float* val, const float* rs, const float* X, float* H,
                                    const int32_t* order, bool bp = false, unsigned long long* zero_words = nullptr,
                                    int zero_count = 0) {
#define SD4(CALLX)                         \
  do {                                     \
    if (S == 1 && d == 128) CALLX(1, 128); \
    else if (S == 2 && d == 128) CALLX(2, 128); \
    else if (S == 1 && d == 256) CALLX(1, 256); \
    else CALLX(2, 256);                    \
  } while (0)
  if (bp) {   // band-plus: (rowptr, col / col16) is the unit-entry CSR, the band + I half comes from the LDS window
#define CALLBP16(S_, D_) hipLaunchKernelGGL((k_aggregate_sliced<S_, D_, false, uint16_t, true>), dim3(gblocks), dim3(512), 0, st, n, rowptr, col16, nullptr, rs, X, H, order, zero_words, zero_count)
#define CALLBP32(S_, D_) hipLaunchKernelGGL((k_aggregate_sliced<S_, D_, false, int, true>), dim3(gblocks), dim3(512), 0, st, n, rowptr, col, nullptr, rs, X, H, order, zero_words, zero_count)
    if (col16) SD4(CALLBP16);
    else SD4(CALLBP32);
#undef CALLBP16
#undef CALLBP32
    return;
  }
  if (col16) {
#define CALL16(S_, D_) hipLaunchKernelGGL((k_aggregate_sliced<S_, D_, false, uint16_t>), dim3(gblocks), dim3(512), 0, st, n, rowptr, col16, val, rs, X, H, order, zero_words, zero_count)
    SD4(CALL16);
#undef CALL16
    return;
  }
#define CALL(S_, D_, V_) hipLaunchKernelGGL((k_aggregate_sliced<S_, D_, V_, int>), dim3(gblocks), dim3(512), 0, st, n, rowptr, col, val, rs, X, H, order, zero_words, zero_count)
  DISPATCH_SDV(S, d, val != nullptr, CALL);
#undef CALL
}
// the arrays the sliced kernels walk for this graph: the band-plus unit-entry CSR when the graph carries one
struct SlicedCsr {
  const int32_t* rowptr;
  const int32_t* col;
  const uint16_t* col16;
  const float* val;
  const int32_t* order;
  bool bp;
};
static inline SlicedCsr sliced_csr(const cgcn_graph_aux* aux, const int32_t* rowptr, const int32_t* col, const float* val, int n_cols);

extern "C" {

int cgcn_abi_version(void) { return CGCN_ABI_VERSION; }

const char* cgcn_strerror(int code) {
  switch (code) {
    case CGCN_OK: return "ok";
    case CGCN_ERR_BAD_ARG: return "bad argument (null pointer, negative size or misaligned buffer)";
    case CGCN_ERR_UNSUPPORTED: return "unsupported shape (need S in {1,2}; d in {128,256} for the fused kernels, a multiple of 4 <= 4096 for cgcn_spmm; S*n*d*4 < 4 GiB)";
    case CGCN_ERR_LAUNCH: return "HIP kernel launch failed";
    case CGCN_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown chromegcn error";
  }
}

// the split forward (see cgcn_layer_fwd): tables from this size on; workgroups and tile height of k_layer_dense.
// 6 MiB since the split products (rounds 2-6: 8): the fused kernel's W operands grew from 32 to 48 registers (6 instead of 8
// waves per SIMD under its gather) while the row-local kernel got 17 % faster, and inside a genome epoch -- inputs cold, not
// the L2-warm loop of a single-chromosome benchmark -- the two launches win from the smallest training chromosome (6.0 MiB)
// on: epoch 3.97 -> 3.90 ms; a 5.6 MiB table alone still prefers the fused kernel by 1 % (profiles/r06_split_products_ab.txt).
#ifndef FWD_SPLIT_TABLE_BYTES
#define FWD_SPLIT_TABLE_BYTES (6u << 20)
#endif
#ifndef DENSE_MAX_BLOCKS
#define DENSE_MAX_BLOCKS (256 * (DENSE_WAVES_PER_SIMD / 2))   // exactly the workgroups resident at once: one round
#endif
static inline int dense_max_blocks(int d) { return d == 256 ? 256 : DENSE_MAX_BLOCKS; }   // d = 256 (k_layer_dense256): one workgroup per CU
#ifndef DENSE_MB
#define DENSE_MB 1
#endif
static long long fwd_split_default() {   // tuning: CGCN_FWD_SPLIT_BYTES in the environment overrides the built-in threshold
  const char* e = getenv("CGCN_FWD_SPLIT_BYTES");
  return (e && *e) ? atoll(e) : (long long)FWD_SPLIT_TABLE_BYTES;
}
static std::atomic<long long> g_fwd_split_bytes{fwd_split_default()};
void cgcn_debug_set_fwd_split_bytes(long long bytes) { g_fwd_split_bytes.store(bytes < 0 ? fwd_split_default() : bytes); }
// How the dense products are formed (cgcn_common.hpp, "split products"): six bf16 MFMA partial products of an exact 3-way
// split (default) or the fp32 MFMA chain of rounds 1-5.  Both are fp32 arithmetic (the split form is the more accurate one);
// they differ in the last bits, so the choice is process-wide, made once from CGCN_PRODUCTS=fp32|split and movable only by
// the measurement hook below (bench.py's A/B, the tests that run both forms).
static int products_default() {
  const char* e = getenv("CGCN_PRODUCTS");
  return (e && (e[0] == 'f' || e[0] == '0')) ? CGCN_PRODUCTS_FP32_CHAIN : CGCN_PRODUCTS_SPLIT;
}
static std::atomic<int> g_products{products_default()};
void cgcn_debug_set_products(int mode) {
  g_products.store((mode == CGCN_PRODUCTS_FP32_CHAIN || mode == CGCN_PRODUCTS_SPLIT) ? mode : products_default());
}
int cgcn_debug_get_products(void) { return g_products.load(); }

// 16-bit column indices serve the feature-sliced kernels of implicit-value graphs (HAS_VAL = false) with <= 65 536 columns
static inline const uint16_t* use_col16(const cgcn_graph_aux* aux, const float* val, int n_cols) {
  return (aux && aux->col16 && !val && n_cols <= 65536) ? aux->col16 : nullptr;
}
// graphs with a row this long take the feature-sliced route at every table size (cgcn_graph_aux::max_row_len)
#ifndef FWD_HUB_ROW
#define FWD_HUB_ROW 2048
#endif
static inline bool hub_graph(const cgcn_graph_aux* aux) { return aux && aux->max_row_len > FWD_HUB_ROW; }
static inline const int32_t* row_order(const cgcn_graph_aux* aux) { return aux ? aux->row_order : nullptr; }
static inline SlicedCsr sliced_csr(const cgcn_graph_aux* aux, const int32_t* rowptr, const int32_t* col, const float* val, int n_cols) {
  if (aux && aux->bp_rowptr && aux->bp_col && val)
    return SlicedCsr{aux->bp_rowptr, aux->bp_col, n_cols <= 65536 ? aux->bp_col16 : nullptr, nullptr, aux->bp_row_order, true};
  return SlicedCsr{rowptr, col, use_col16(aux, val, n_cols), val, row_order(aux), false};
}
// band graphs (cgcn_graph_aux::band_halfwidth; implicit unit values): the sliding-window kernels instead of the CSR walk
static inline bool band_graph(const cgcn_graph_aux* aux, const float* val) { return aux && aux->band_halfwidth == BAND_W && !val; }
// 'both' graphs with a band-plus decomposition (cgcn_graph_aux::bp_*): the unit-entry CSR + the LDS window (see BP above)
static inline bool bandplus_graph(const cgcn_graph_aux* aux, const float* val) { return aux && aux->bp_rowptr && aux->bp_col && val; }
static void launch_band_aggregate(hipStream_t st, int n, int S, int d, const float* rs, const float* X, float* H,
                                  unsigned long long* zw = nullptr, int zc = 0) {
  const int blocks = S * ((n + BAND_R - 1) / BAND_R);
  if (S == 1 && d == 128) hipLaunchKernelGGL((k_band_aggregate<1, 128, BAND_R>), dim3(blocks), dim3(512), 0, st, n, rs, X, H, zw, zc);
  else if (S == 2 && d == 128) hipLaunchKernelGGL((k_band_aggregate<2, 128, BAND_R>), dim3(blocks), dim3(512), 0, st, n, rs, X, H, zw, zc);
  else if (S == 1 && d == 256) hipLaunchKernelGGL((k_band_aggregate<1, 256, BAND_R>), dim3(blocks), dim3(512), 0, st, n, rs, X, H, zw, zc);
  else hipLaunchKernelGGL((k_band_aggregate<2, 256, BAND_R>), dim3(blocks), dim3(512), 0, st, n, rs, X, H, zw, zc);
}

int cgcn_spmm(cgcn_stream_t stream, int n_rows, int n_cols, int S, int d, const int32_t* rowptr, const int32_t* col,
              const float* val, const float* row_scale, const float* X, float* Y, const cgcn_graph_aux* aux) {
  const int nmax = n_rows > n_cols ? n_rows : n_cols;
  if (nmax < 0) return CGCN_ERR_BAD_ARG;
  // the bare aggregation takes any width that is a multiple of 4 (k_spmm_any); S*D in {128, 256, 512} has tuned kernels
  if (!(S == 1 || S == 2) || d < 4 || (d & 3) || d > 4096) return CGCN_ERR_UNSUPPORTED;
  if ((double)nmax * (double)S * (double)d * 4.0 >= 4294967296.0) return CGCN_ERR_UNSUPPORTED;
  if (n_rows == 0) return CGCN_OK;
  if (!rowptr || !col || !X || !Y || X == Y) return CGCN_ERR_BAD_ARG;
  if (misaligned16(X) || misaligned16(Y)) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!(d == 128 || d == 256)) {
    const long long rows = (long long)S * n_rows;
    const int blocks = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipLaunchKernelGGL(k_spmm_any, dim3(blocks), dim3(256), 0, st, n_rows, n_cols, S, d, rowptr, col, val, row_scale, X, Y);
    return launch_status();
  }
  if (n_rows == n_cols && band_graph(aux, val)) {   // a band operator: the sliding-window stream (see k_band_aggregate)
    launch_band_aggregate(st, n_rows, S, d, row_scale, X, Y);
    return launch_status();
  }
  if (n_rows == n_cols && ((double)n_rows * S * d * 4.0 >= (double)g_fwd_split_bytes.load() || hub_graph(aux) || bandplus_graph(aux, val))) {
    // square operator on a table too large for the L2s (or a band-plus graph): the feature-sliced aggregation (see k_aggregate_sliced)
    const int gblocks = (S * d / 32) * ((n_rows + 63) / 64);
    const SlicedCsr c = sliced_csr(aux, rowptr, col, val, n_cols);
    launch_aggregate_sliced(st, gblocks, n_rows, S, d, c.rowptr, c.col, c.col16, c.val, row_scale, X, Y, c.order, c.bp);
    return launch_status();
  }
  const int blocks = (n_rows + 3) / 4 < 4096 ? (n_rows + 3) / 4 : 4096;
  const bool deep = false;  // measured: the bare SpMM never gains from the deeper batches (chr10-like 48.7 vs 55.8 us)
#define CALL(S_, D_, V_)                                                                                                  \
  do {                                                                                                                    \
    if (deep) hipLaunchKernelGGL((k_spmm<S_, D_, V_, true>), dim3(blocks), dim3(256), 0, st, n_rows, n_cols, rowptr, col, val, row_scale, X, Y); \
    else hipLaunchKernelGGL((k_spmm<S_, D_, V_, false>), dim3(blocks), dim3(256), 0, st, n_rows, n_cols, rowptr, col, val, row_scale, X, Y); \
  } while (0)
  DISPATCH_SDV(S, d, val != nullptr, CALL);
#undef CALL
  return launch_status();
}

// Nodes per column-statistics record of cgcn_layer_fwd(n, S, d).  The fused kernel emits one record per 16 / S-node
// tile.  Tables that take the split route (and the same tables when an H_in is streamed) go through k_layer_dense,
// which merges the tiles of one workgroup into one record: chunk tiles, so that at most DENSE_MAX_BLOCKS records exist.
static bool fwd_split_shape(int n, int S, int d) {
  const long long split_bytes = g_fwd_split_bytes.load();
  const double table = (double)n * S * d * 4.0;
  if (split_bytes == 0) return true;   // the debug hook's 0 forces the split at every shape
  // Payloads of 2 KiB per node (d = 256, both strands: 16 column slices, two per XCD) take the same threshold since round 6.
  // Rounds 3-5 kept the fused k_layer_fwd<2,256> at every size (tools/strand_split_probe.py, profiles/
  // r03_d256_strand_split_experiment.txt: the two launches won on uniform graphs above 7 000 nodes and lost on distance-decay
  // graphs, 207 vs 183 us at n = 29 910) because the row-local launch was k_layer_dense<2,256>: 33 us at n = 5 776, ~130 at
  // 29 910.  k_layer_dense256 takes 27 / 93 us: config 4's chr21-size step 0.742 -> 0.727 ms on uniform graphs, 0.692 -> 0.673
  // on distance-decay graphs (profiles/r06_dense256_experiment.txt).
  return table >= (double)split_bytes;
}
static int dense_stat_chunk(int n, int S, int d) {
  if (!fwd_split_shape(n, S, d)) return 1;
  const int tn = 16 * DENSE_MB / S;
  const int ntiles = (n + tn - 1) / tn;
  return (ntiles + dense_max_blocks(d) - 1) / dense_max_blocks(d);
}

// Column statistics: RECORDS (one (mean, M2) record per node tile -- per contiguous chunk of tiles on the two-launch route)
// or ACCUMULATE (cgcn_common.hpp, STAT_ACC_*: fixed-point integer totals; the two-launch route only, whose aggregation
// launch zeroes them).  The mode is the CALLER's, per call (ABI v24): the plan below says what to allocate for it,
// cgcn_layer_fwd is told what the plan said (colstats_rows) and never consults process state about it.
static int acc_chunk(int n, int S, int d) {   // contiguous tiles per k_layer_dense workgroup: one round of workgroups
  const int tn = 16 * DENSE_MB / S;
  const int ntiles = (n + tn - 1) / tn;
  return (ntiles + dense_max_blocks(d) - 1) / dense_max_blocks(d);
}

int cgcn_layer_fwd_colstats_plan(int n, int S, int d, int mode, int* rows_per_tile) {
  if (check_shape(n, S, d) != CGCN_OK || n == 0) return 0;
  if (mode != CGCN_COLSTATS_RECORDS && mode != CGCN_COLSTATS_ACCUMULATE) return 0;
  if (mode == CGCN_COLSTATS_ACCUMULATE && n >= 2) {   // the buffer holds the integer accumulators; rows_per_tile = -1 says so
    if (rows_per_tile) *rows_per_tile = -1;
    const size_t tile_bytes = (size_t)S * d * 2 * sizeof(float);
    return (int)((stat_acc_words(S, d) * 8 + tile_bytes - 1) / tile_bytes);
  }
  const int tn = (16 * pick_mb(n, S) / S) * dense_stat_chunk(n, S, d);
  if (rows_per_tile) *rows_per_tile = tn;
  return (n + tn - 1) / tn;
}

int cgcn_debug_layer_fwd_route(int n, int S, int d, const cgcn_graph_aux* aux, int colstats_rows) {
  const int rc = check_shape(n, S, d);
  if (rc) return rc;
  if (band_graph(aux, nullptr)) return 2;
  const bool need_dense = colstats_rows == CGCN_COLSTATS_ROWS_ACCUMULATE || colstats_rows > 16 * DENSE_MB / S;   // see cgcn_layer_fwd
  return (fwd_split_shape(n, S, d) || hub_graph(aux) || (aux && aux->bp_rowptr && aux->bp_col) || need_dense) ? 1 : 0;
}

int cgcn_layer_fwd(cgcn_stream_t stream, int n, int S, int d, const int32_t* rowptr, const int32_t* col, const float* val,
                   const float* row_scale, const float* X, const float* W, const float* b, const float* wg,
                   const float* cg, float* Xn, float* Z, float* H, float* gate, float dropout_p,
                   const unsigned long long* rng_state, unsigned int stream_id, const float* H_in, float* colstats,
                   int colstats_rows, const cgcn_graph_aux* aux) {
  int rc = check_shape(n, S, d);
  if (rc) return rc;
  if (n == 0) return CGCN_OK;
  if (!rowptr || !col || !X || !W || !b || !wg || !cg || !Xn || !gate || X == Xn) return CGCN_ERR_BAD_ARG;
  if (misaligned16(X) || misaligned16(Xn) || misaligned16(W) || (Z && misaligned16(Z)) || (H && misaligned16(H)))
    return CGCN_ERR_BAD_ARG;
  if (H_in && (misaligned16(H_in) || H_in == H)) return CGCN_ERR_BAD_ARG;
  float ks;
  uint32_t th;
  if ((rc = dropout_args(dropout_p, rng_state, &ks, &th))) return rc;
  hipStream_t st = (hipStream_t)stream;
  // Three routes.  H_in given: the row-local kernel alone.  Training (H is wanted anyway) on a table that does not
  // fit the L2s: feature-sliced aggregation into H, then the row-local kernel on it.  Otherwise the fused kernel.
  const bool band = band_graph(aux, val);
  // Column statistics as the caller planned them (cgcn_layer_fwd_colstats_plan): -1 = accumulate mode, else the nodes per
  // record.  One record per 16 / S-node tile is what the fused kernel writes; merged records (k_layer_dense's contiguous
  // tile chunks) and the integer totals come from the row-local kernel, i.e. need the two-launch route: H or H_in.
  //   -1 accumulate, THIS call zeroes the totals (its aggregation launch does: two-launch route);
  //   -2 no statistics from this call: its first launch zeroes the totals in `colstats` for a LATER call of the step;
  //   -3 accumulate into totals an earlier call zeroed (-2): any route, the fused kernel included.
  const int tn0 = 16 * DENSE_MB / S;
  const bool zero_only = colstats && colstats_rows == CGCN_COLSTATS_ROWS_ZERO_ONLY;
  const bool acc_pre = colstats && colstats_rows == CGCN_COLSTATS_ROWS_ACCUMULATE_ZEROED;
  const bool acc = colstats && (colstats_rows == CGCN_COLSTATS_ROWS_ACCUMULATE || acc_pre);
  int chunk = 1;
  if (colstats) {
    if (acc || zero_only) {
      if (n < 2 || ((uintptr_t)colstats & 7)) return CGCN_ERR_BAD_ARG;
      chunk = acc_chunk(n, S, d);
    } else {
      if (colstats_rows < tn0 || colstats_rows % tn0) return CGCN_ERR_BAD_ARG;
      chunk = colstats_rows / tn0;
    }
  }
  unsigned long long* zw = zero_only ? (unsigned long long*)colstats : nullptr;   // zeroed by this call's FIRST launch
  const int zc = zero_only ? (int)stat_acc_words(S, d) : 0;
  if (zero_only) colstats = nullptr;                                              // ... which produces no statistics
  const bool need_dense = colstats && ((acc && !acc_pre) || (!acc && chunk != 1));
  const bool split = !H_in && H && (fwd_split_shape(n, S, d) || hub_graph(aux) || band || bandplus_graph(aux, val) || need_dense);
  if (!split && !H_in) chunk = 1;   // (the fused kernel: one tile per workgroup; accumulate mode has no records to count)
  bool acc_zeroed = acc_pre;
  if (need_dense && !split && !H_in) return CGCN_ERR_BAD_ARG;
  if (split) {
    const int gblocks = (S * d / 32) * ((n + 63) / 64);
    // (the aggregation launch zeroes: this call's own totals, or a later call's)
    unsigned long long* zq = zw ? zw : ((acc && !acc_pre) ? (unsigned long long*)colstats : nullptr);
    const int zn = zq ? (int)stat_acc_words(S, d) : 0;
    if (band) {
      launch_band_aggregate(st, n, S, d, row_scale, X, H, zq, zn);
    } else {
      const SlicedCsr c = sliced_csr(aux, rowptr, col, val, n);
      launch_aggregate_sliced(st, gblocks, n, S, d, c.rowptr, c.col, c.col16, c.val, row_scale, X, H, c.order, c.bp, zq, zn);
    }
    acc_zeroed = acc;
    zw = nullptr;
    if ((rc = launch_status())) return rc;
    H_in = H;
  }
  if (H_in) {
    // (accumulate mode without a sliced aggregation launch in this call -- a caller's H_in, a band graph: a memset node)
    if (acc && !acc_zeroed && hipMemsetAsync(colstats, 0, stat_acc_words(S, d) * 8, st) != hipSuccess) return CGCN_ERR_LAUNCH;
    constexpr int MB = DENSE_MB;
    const int tn = 16 * MB / S;
    const int ntiles = (n + tn - 1) / tn;
    const int grid = colstats ? (ntiles + chunk - 1) / chunk : (ntiles < dense_max_blocks(d) ? ntiles : dense_max_blocks(d));
    if (d == 256) {   // one 16-wave workgroup per CU (k_layer_dense256)
      if (S == 1) hipLaunchKernelGGL((k_layer_dense256<1>), dim3(grid), dim3(1024), 0, st, n, ntiles, H_in, X, W, b, wg, cg, Xn, Z, gate,
                                     ks, th, rng_state, stream_id, colstats, chunk, acc ? 1 : 0, zw, zc);
      else hipLaunchKernelGGL((k_layer_dense256<2>), dim3(grid), dim3(1024), 0, st, n, ntiles, H_in, X, W, b, wg, cg, Xn, Z, gate,
                              ks, th, rng_state, stream_id, colstats, chunk, acc ? 1 : 0, zw, zc);
      return launch_status();
    }
    const bool sp = g_products.load() != CGCN_PRODUCTS_FP32_CHAIN;
#define CALL(S_, P_) \
    hipLaunchKernelGGL((k_layer_dense<S_, 128, MB, P_>), dim3(grid), dim3(512), 0, st, n, ntiles, H_in, X, W, b, wg, cg, Xn, Z, gate, \
                       ks, th, rng_state, stream_id, colstats, chunk, acc ? 1 : 0, zw, zc)
    if (S == 1) { if (sp) CALL(1, 1); else CALL(1, 0); }
    else { if (sp) CALL(2, 1); else CALL(2, 0); }
#undef CALL
    return launch_status();
  }
  const int mb = pick_mb(n, S);
  const int tn = 16 * mb / S;
  const int blocks = (n + tn - 1) / tn;
  const bool deep = pick_deep(n, S, d);
  const bool spf = g_products.load() != CGCN_PRODUCTS_FP32_CHAIN;   // (d = 128 only: d = 256 keeps the chain on every route)
#define FWD(S_, D_, MB_, V_, DP_, P_)                                                                                 \
  hipLaunchKernelGGL((k_layer_fwd<S_, D_, MB_, V_, DP_, (D_ == 128) ? P_ : 0>), dim3(blocks), blk, 0, st, n, rowptr, col, val, row_scale, \
                     X, W, b, wg, cg, Xn, Z, H, gate, ks, th, rng_state, stream_id, colstats, acc ? 1 : 0, zw, zc)
#define CALL(S_, D_, V_)                                                                                              \
  do {                                                                                                                \
    const dim3 blk((D_ == 128 && CBW128 == 2) ? 256 : 512);                                                           \
    if (deep) { if (spf) FWD(S_, D_, 1, V_, true, 1); else FWD(S_, D_, 1, V_, true, 0); }                             \
    else { if (spf) FWD(S_, D_, 1, V_, false, 1); else FWD(S_, D_, 1, V_, false, 0); }                                \
  } while (0)
  DISPATCH_SDV(S, d, val != nullptr, CALL);
#undef CALL
#undef FWD
  return launch_status();
}

#ifndef BWD256_MAX_RANGES
#define BWD256_MAX_RANGES 64
#endif
// Workgroups (= partial records) of the row-local launch: one persistent workgroup per CU at most (BWD_MAX_PARTIALS).
// d = 128: k_bwd_rowlocal_ring, 16-row slots dealt in contiguous, balanced ranges; d = 256: k_bwd_rowlocal256s, 32-row tiles,
// four column-slab workgroups per record.
static int bwd_partials(int n, int S, int d) {
  const int M = n * S;
  const int tr = d == 128 ? 16 : 32;
  const int ntiles = (M + tr - 1) / tr;
  if (d == 256) {   // k_bwd_rowlocal256s: 4 column-slab workgroups per record; ranges in multiples of 8 (XCD pairing), <= 64
    const int nr = ((ntiles < BWD256_MAX_RANGES ? ntiles : BWD256_MAX_RANGES) + 7) / 8 * 8;
    return nr < 8 ? 8 : nr;
  }
  const int P = ntiles < BWD_MAX_PARTIALS ? ntiles : BWD_MAX_PARTIALS;
  return P < 1 ? 1 : P;
}

int cgcn_debug_layer_bwd_route(int n, int S, int d) {
  const int rc = check_shape(n, S, d);
  if (rc) return rc;
  return d == 128 ? 2 : 0;
}

size_t cgcn_layer_bwd_workspace_bytes(int n, int S, int d) {
  if (check_shape(n, S, d) != CGCN_OK) return 0;
  // partial blocks of the row-local kernel
  return (size_t)bwd_partials(n, S, d) * ((size_t)d * d + 2 * d + 4) * sizeof(float);
}

// phases: bit 0 = the row-local launch (k_bwd_rowlocal_ring / k_bwd_rowlocal256s), bit 1 = the launch that follows it
// (k_bwd_sliced with the second-stage sums / the optimizer step in its trailing workgroups, or k_reduce_partials).
// cgcn_layer_bwd runs both; cgcn_debug_layer_bwd_phases lets a profiler time them one at a time.
static int layer_bwd_impl(cgcn_stream_t stream, int n, int S, int d, const int32_t* rowptr_t, const int32_t* col_t,
                          const float* val_t, const float* row_scale, const float* X, const float* Z, const float* H,
                          const float* gate, const float* W, const float* wg, const float* dXn, const float* dgate, float* dX,
                          float* dHs, float* dW, float* db, float* dwg, float* dcg, int accumulate, float in_dropout_p,
                          const unsigned long long* rng_state, unsigned int in_stream_id, const cgcn_head_grad* head,
                          void* workspace, size_t workspace_bytes, cgcn_stream_t aux_stream, const cgcn_sgd_fuse* sgd,
                          int phases, const cgcn_graph_aux* aux_t) {
  int rc = check_shape(n, S, d);
  if (rc) return rc;
  if (!rowptr_t || !col_t || !X || !Z || !H || !gate || !W || !wg || !dW || !db || !dwg || !dcg) return CGCN_ERR_BAD_ARG;
  if (dX && !dHs) return CGCN_ERR_BAD_ARG;  // the gather's operand; without dX it is optional (NULL: not computed)
  SgdFuse sg = {nullptr, nullptr, nullptr, 0, 0.f, 0.f, 0.f, 1.f, 0, nullptr};
  if (sgd) {
    // the launch that finishes this layer's sums carries the step: it must overwrite them and run on the main stream
    if (n == 0 || accumulate || aux_stream) return CGCN_ERR_BAD_ARG;
    if (!sgd->param || !sgd->grad || sgd->count <= 0 || sgd->count > 2147483647LL) return CGCN_ERR_BAD_ARG;
    if ((sgd->momentum != 0.f) != (sgd->momentum_buf != nullptr) || (sgd->nesterov && sgd->momentum == 0.f)) return CGCN_ERR_BAD_ARG;
    const float* g0 = sgd->grad;
    const float* g1 = sgd->grad + sgd->count;
    if (!(dW >= g0 && dW + (size_t)d * d <= g1 && db >= g0 && db + d <= g1 && dwg >= g0 && dwg + d <= g1 && dcg >= g0 && dcg < g1))
      return CGCN_ERR_BAD_ARG;  // the layer's gradient outputs must live inside the flat gradient arena
    sg = SgdFuse{sgd->param, sgd->grad, sgd->momentum_buf, (int)sgd->count, sgd->lr, sgd->momentum, sgd->weight_decay,
                 sgd->grad_scale, sgd->nesterov, sgd->rng_state};
  }
  if ((dXn == nullptr) == (head == nullptr)) return CGCN_ERR_BAD_ARG;  // exactly one source of dL/dXn
  // the launch that carries the optimizer step also advances the dropout step counter (rng_state[1]) in a trailing
  // workgroup, unordered against the gather workgroups of the same launch that would read it for the input-dropout mask
  if (sgd && dX && in_dropout_p > 0.f) return CGCN_ERR_BAD_ARG;
  if ((dX && dX == dXn) || (dHs && (misaligned16(dHs) || dHs == dX)) || (dX && misaligned16(dX)) || misaligned16(W)) return CGCN_ERR_BAD_ARG;
  if (misaligned16(X) || misaligned16(Z) || misaligned16(H) || (dXn && misaligned16(dXn))) return CGCN_ERR_BAD_ARG;  // vector row accesses
  HeadApply hp = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, 0u, S, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, nullptr};
  int head_slabs = 0;
  if (head) {
    if (!head->dym || !head->bnc || !head->save_mean || !head->save_invstd || !head->bn_w) return CGCN_ERR_BAD_ARG;
    if (!head->part || !head->dW_out || !head->db_out || head->C < 1 || head->C > 256) return CGCN_ERR_BAD_ARG;
    float hks;
    uint32_t hth;
    if ((rc = dropout_args(head->dropout_p, head->rng_state, &hks, &hth))) return rc;
    const int CP = head->C <= 128 ? 128 : 256;
    // accumulate mode (ABI v23; d = 256 since v24): the BatchNorm-backward column means are decoded from the integer totals
    // the head kernel left (k_bwd_rowlocal_ring / k_bwd_rowlocal256s prologue)
    if (head->stat_acc && ((uintptr_t)head->stat_acc & 7)) return CGCN_ERR_BAD_ARG;
    hp = HeadApply{head->dym, head->bnc, (const unsigned long long*)head->stat_acc, head->save_mean, head->save_invstd, head->bn_w, head->rng_state, hks, hth, S,
                   head->part, head->dW_out, head->db_out, head->n_partials, head->C, CP, head->accumulate, head->dloss,
                   head->dbn_w, head->dbn_b};
    if ((head->dbn_w == nullptr) != (head->dbn_b == nullptr)) return CGCN_ERR_BAD_ARG;
    // dW_out / db_out slabs, plus the BatchNorm-column slabs when their parameter gradients are still to be summed
    head_slabs = (CP * d + CP) / 64 + (head->dbn_w ? d / HEAD_STAT_COLS : 0);
  }
  // the head's second-stage workgroups ride in the gather launch when there is one on this stream and it does not carry
  // the optimizer step (whose workgroups would step the head's gradients while these finish them); else in the row-local one
#ifndef HEAD_SLABS_IN_GATHER
#define HEAD_SLABS_IN_GATHER 1   // 0: A/B, round 3's placement (profiles/r04_head_slabs_in_gather.txt)
#endif
  const bool head_in_gather = HEAD_SLABS_IN_GATHER && head_slabs > 0 && n > 0 && dX != nullptr && !sg.param && !(aux_stream && aux_stream != stream);
  const int head_slabs_rl = head_in_gather ? 0 : head_slabs, head_slabs_g = head_in_gather ? head_slabs : 0;
  if (!workspace || workspace_bytes < cgcn_layer_bwd_workspace_bytes(n, S, d)) return CGCN_ERR_WORKSPACE;
  if (misaligned16(workspace)) return CGCN_ERR_BAD_ARG;
  float ks;
  uint32_t th;
  if ((rc = dropout_args(in_dropout_p, rng_state, &ks, &th))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int P = bwd_partials(n, S, d);
  float* part = (float*)workspace;
  const int M = n * S;
  if (!(phases & 1)) {
    // profiling only: the partials / dHs / dL/dXn of an earlier full call are still in place
  } else if (d == 128) {
    const bool spb = g_products.load() != CGCN_PRODUCTS_FP32_CHAIN;
#define RING_(H_, D_, P_) hipLaunchKernelGGL((k_bwd_rowlocal_ring<H_, D_, P_>), dim3(P + head_slabs_rl), dim3(RING_THREADS), 0, st, M, n, dXn, Z, X, gate, dgate, H, wg, row_scale, dHs, part, hp, dX, P, head_slabs_rl, W)
#define RING(H_, D_) do { if (spb) RING_(H_, D_, 1); else RING_(H_, D_, 0); } while (0)
    if (!head) RING(false, false);
    else if (hp.thresh) RING(true, true);
    else RING(true, false);
#undef RING
#undef RING_
  } else
    hipLaunchKernelGGL((k_bwd_rowlocal256s<32>), dim3(4 * P + head_slabs_rl), dim3(RL256_THREADS), 0, st, M, n, dXn, Z, X, gate, dgate, H, wg, row_scale, dHs, part, hp, dX, 4 * P, head_slabs_rl, W);
  if ((rc = launch_status())) return rc;
  if (!(phases & 2)) return CGCN_OK;
  // The reduction of the per-tile partials and the gather kernel are independent: with an auxiliary stream
  // they run side by side (fork after k_bwd_rowlocal, join before returning; both edges are events, so the
  // fork/join is captured as graph dependencies under HIP-graph capture).
  const int total = d * d + 2 * d + 1;
  hipStream_t rs_stream = st;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  if (aux_stream && aux_stream != stream && n > 0) {
    ev_fork = pooled_event();
    ev_join = pooled_event();
    if (ev_fork && ev_join && hipEventRecord(ev_fork, st) == hipSuccess &&
        hipStreamWaitEvent((hipStream_t)aux_stream, ev_fork, 0) == hipSuccess)
      rs_stream = (hipStream_t)aux_stream;
  }
  const int slabs = (total + 63) / 64;
  // (riding in a gather launch, few partial records are finished 256 elements per workgroup: reduce_slab_wide)
  const int rslabs = P <= REDUCE_WIDE_MAX_P ? (total + 255) / 256 : slabs;
  // default: the sum rides at the end of the gather launch; no gather when the caller does not want dX
  const bool fuse_reduce = (rs_stream == st) && n > 0 && dX != nullptr;
  if (!fuse_reduce) {
    const int sgd_blocks_r = sg.param ? (sg.count + 255) / 256 : 0;   // no gather launch: the step rides here instead
    hipLaunchKernelGGL(k_reduce_partials, dim3(slabs + sgd_blocks_r), dim3(256), 0, rs_stream, P, d, part, dW, db, dwg, dcg,
                       accumulate, sg, slabs);
    if ((rc = launch_status())) return rc;
    if (rs_stream != st && hipEventRecord(ev_join, rs_stream) != hipSuccess) return CGCN_ERR_LAUNCH;
  }
  if (n == 0 || !dX) {
    if (rs_stream != st && hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return CGCN_ERR_LAUNCH;
    return CGCN_OK;  // parameter gradients only (the input is a leaf nobody differentiates)
  }
  const int blocks = (S * d / 32) * ((n + 63) / 64);   // slices x 64-row tiles (k_bwd_sliced)
  if (head) dXn = dX;  // k_bwd_rowlocal left dL/dXn there; each thread reads its elements before overwriting them
  const int sgd_blocks = sg.param ? (sg.count + 511) / 512 : 0;
  if (band_graph(aux_t, val_t)) {   // band operator (symmetric): the sliding-window stream with the same riders
    const int bblocks = S * ((n + BAND_R - 1) / BAND_R);
#define CALLB(S_, D_)                                                                                                 \
  hipLaunchKernelGGL((k_bwd_band<S_, D_, BAND_R>), dim3(bblocks + (fuse_reduce ? rslabs : 0) + head_slabs_g + sgd_blocks), dim3(512), 0, st, \
                     n, dHs, dXn, gate, dX, ks, th, rng_state, in_stream_id, bblocks, P, part, dW, db, dwg, dcg, accumulate, sg, \
                     fuse_reduce ? rslabs : 0, hp, head_slabs_g)
    if (S == 1 && d == 128) CALLB(1, 128);
    else if (S == 2 && d == 128) CALLB(2, 128);
    else if (S == 1 && d == 256) CALLB(1, 256);
    else CALLB(2, 256);
#undef CALLB
  } else if (bandplus_graph(aux_t, val_t)) {   // 'both': the unit-entry CSR + the band window from LDS (BP)
    const SlicedCsr c = sliced_csr(aux_t, rowptr_t, col_t, val_t, n);
#define CALLBP(S_, D_, IT_, COL_)                                                                                    \
  hipLaunchKernelGGL((k_bwd_sliced<S_, D_, false, IT_, true>), dim3(blocks + (fuse_reduce ? rslabs : 0) + head_slabs_g + sgd_blocks), dim3(512), 0, \
                     st, n, c.rowptr, COL_, nullptr, dHs, dXn, gate, dX, ks, th, rng_state, in_stream_id, blocks, P,  \
                     part, dW, db, dwg, dcg, accumulate, sg, fuse_reduce ? rslabs : 0, c.order, hp, head_slabs_g)
#define CALLBP16(S_, D_) CALLBP(S_, D_, uint16_t, c.col16)
#define CALLBP32(S_, D_) CALLBP(S_, D_, int, c.col)
    if (c.col16) { if (S == 1 && d == 128) CALLBP16(1, 128); else if (S == 2 && d == 128) CALLBP16(2, 128); else if (S == 1 && d == 256) CALLBP16(1, 256); else CALLBP16(2, 256); }
    else { if (S == 1 && d == 128) CALLBP32(1, 128); else if (S == 2 && d == 128) CALLBP32(2, 128); else if (S == 1 && d == 256) CALLBP32(1, 256); else CALLBP32(2, 256); }
#undef CALLBP16
#undef CALLBP32
#undef CALLBP
  } else if (const uint16_t* col16_t = use_col16(aux_t, val_t, n)) {
#define CALL16(S_, D_)                                                                                               \
  hipLaunchKernelGGL((k_bwd_sliced<S_, D_, false, uint16_t>), dim3(blocks + (fuse_reduce ? rslabs : 0) + head_slabs_g + sgd_blocks), dim3(512), 0, \
                     st, n, rowptr_t, col16_t, val_t, dHs, dXn, gate, dX, ks, th, rng_state, in_stream_id, blocks, P, \
                     part, dW, db, dwg, dcg, accumulate, sg, fuse_reduce ? rslabs : 0, row_order(aux_t), hp, head_slabs_g)
    if (S == 1 && d == 128) CALL16(1, 128);
    else if (S == 2 && d == 128) CALL16(2, 128);
    else if (S == 1 && d == 256) CALL16(1, 256);
    else CALL16(2, 256);
#undef CALL16
  } else {
#define CALL(S_, D_, V_)                                                                                             \
  hipLaunchKernelGGL((k_bwd_sliced<S_, D_, V_, int>), dim3(blocks + (fuse_reduce ? rslabs : 0) + head_slabs_g + sgd_blocks), dim3(512), 0, \
                     st, n, rowptr_t, col_t, val_t, dHs, dXn, gate, dX, ks, th, rng_state, in_stream_id, blocks, P,  \
                     part, dW, db, dwg, dcg, accumulate, sg, fuse_reduce ? rslabs : 0, row_order(aux_t), hp, head_slabs_g)
    DISPATCH_SDV(S, d, val_t != nullptr, CALL);
#undef CALL
  }
  if ((rc = launch_status())) return rc;
  if (rs_stream != st && hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return CGCN_ERR_LAUNCH;  // join
  return CGCN_OK;
}

int cgcn_layer_bwd(cgcn_stream_t stream, int n, int S, int d, const int32_t* rowptr_t, const int32_t* col_t,
                   const float* val_t, const float* row_scale, const float* X, const float* Z, const float* H,
                   const float* gate, const float* W, const float* wg, const float* dXn, const float* dgate, float* dX,
                   float* dHs, float* dW, float* db, float* dwg, float* dcg, int accumulate, float in_dropout_p,
                   const unsigned long long* rng_state, unsigned int in_stream_id, const cgcn_head_grad* head,
                   void* workspace, size_t workspace_bytes, cgcn_stream_t aux_stream, const cgcn_sgd_fuse* sgd,
                   const cgcn_graph_aux* aux_t) {
  return layer_bwd_impl(stream, n, S, d, rowptr_t, col_t, val_t, row_scale, X, Z, H, gate, W, wg, dXn, dgate, dX, dHs, dW, db,
                        dwg, dcg, accumulate, in_dropout_p, rng_state, in_stream_id, head, workspace, workspace_bytes,
                        aux_stream, sgd, 3, aux_t);
}

int cgcn_debug_layer_bwd_phases(cgcn_stream_t stream, int n, int S, int d, const int32_t* rowptr_t, const int32_t* col_t,
                                const float* val_t, const float* row_scale, const float* X, const float* Z, const float* H,
                                const float* gate, const float* W, const float* wg, const float* dXn, const float* dgate,
                                float* dX, float* dHs, float* dW, float* db, float* dwg, float* dcg, int accumulate,
                                float in_dropout_p, const unsigned long long* rng_state, unsigned int in_stream_id,
                                const cgcn_head_grad* head, void* workspace, size_t workspace_bytes, int phases,
                                const cgcn_graph_aux* aux_t) {
  if (phases < 1 || phases > 3) return CGCN_ERR_BAD_ARG;
  return layer_bwd_impl(stream, n, S, d, rowptr_t, col_t, val_t, row_scale, X, Z, H, gate, W, wg, dXn, dgate, dX, dHs, dW, db,
                        dwg, dcg, accumulate, in_dropout_p, rng_state, in_stream_id, head, workspace, workspace_bytes,
                        nullptr, nullptr, phases, aux_t);
}

int cgcn_sddmm(cgcn_stream_t stream, int n, int S, int d, const int32_t* rowptr, const int32_t* col, const float* A,
               const float* B, float* out, int accumulate) {
  int rc = check_shape(n, S, d);
  if (rc) return rc;
  if (n == 0) return CGCN_OK;
  if (!rowptr || !col || !A || !B || !out || misaligned16(A) || misaligned16(B)) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  if (S == 1 && d == 128) hipLaunchKernelGGL((k_sddmm<1, 128>), dim3(blocks), dim3(256), 0, st, n, rowptr, col, A, B, out, accumulate);
  else if (S == 2 && d == 128) hipLaunchKernelGGL((k_sddmm<2, 128>), dim3(blocks), dim3(256), 0, st, n, rowptr, col, A, B, out, accumulate);
  else if (S == 1 && d == 256) hipLaunchKernelGGL((k_sddmm<1, 256>), dim3(blocks), dim3(256), 0, st, n, rowptr, col, A, B, out, accumulate);
  else hipLaunchKernelGGL((k_sddmm<2, 256>), dim3(blocks), dim3(256), 0, st, n, rowptr, col, A, B, out, accumulate);
  return launch_status();
}

int cgcn_saliency_normalize(cgcn_stream_t stream, int n, const int32_t* rowptr, const float* val, const float* raw, float* out) {
  if (n < 0) return CGCN_ERR_BAD_ARG;
  if (n == 0) return CGCN_OK;
  if (!rowptr || !raw || !out) return CGCN_ERR_BAD_ARG;
  const int blocks = (n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096;
  hipLaunchKernelGGL(k_saliency_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, rowptr, val, raw, out);
  return launch_status();
}

int cgcn_sgd_step(cgcn_stream_t stream, long long count, float* param, const float* grad, float* momentum_buf, float lr,
                  float momentum, float weight_decay, int nesterov, float grad_scale, unsigned long long* rng_state) {
  if (count < 0 || count > 2147483647LL) return CGCN_ERR_UNSUPPORTED;
  if (count > 0 && (!param || !grad)) return CGCN_ERR_BAD_ARG;
  if (momentum != 0.f && !momentum_buf) return CGCN_ERR_BAD_ARG;
  if (nesterov && momentum == 0.f) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = count > 0 ? (int)((count + 255) / 256) : 1;
  hipLaunchKernelGGL(k_sgd, dim3(blocks), dim3(256), 0, st, (int)count, param, grad, momentum != 0.f ? momentum_buf : nullptr,
                     lr, momentum, weight_decay, nesterov, grad_scale, rng_state);
  return launch_status();
}

}  // extern "C"
