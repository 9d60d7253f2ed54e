// chromegcn_amd/csrc/cgcn_head.hip
//
// Fused classifier head of ChromeGCN for the GCN-stage train/eval step (gfx950 only):
//     relu -> BatchNorm1d over the node axis -> dropout -> Linear(d, C)      models/ChromeModels.py:48-51
//     pred = mean over strands of the logits                                 finetune.py:43
//     loss = binary_cross_entropy_with_logits(pred, target) (mean)           finetune.py:45
//     probs = sigmoid(pred)                                                  finetune.py:52
// and its backward.  Because Linear is affine, mean_s(Linear(y_s)) = Linear(mean_s y_s): the strands are
// averaged BEFORE the d x C contraction, which halves the MFMA work and the dW_out reduction.
//
// Kernels:
//   k_head_colstats<D>     per-strand column mean / M2 of relu(X) (Welford per thread, Chan combine)
//   k_head_bn_finalize     combine partials -> batch mean / invstd, running-stat update (strand 0 then 1,
//                          like the reference's two forward calls), num_batches_tracked += S
//   k_head_fwd<D>          16-node tile: ym = mean_s dropout(BN(relu(x_s))) -> LDS -> fp32 MFMA x W_out^T
//                          -> sigmoid / BCE / d(loss)/d(pred) epilogue
//   k_sum_scale            deterministic sum of the per-workgroup loss partials
//   k_head_bwd<D,CBMAX>    persistent, 32-node tiles: dym = dpred W_out (MFMA), dW_out += dpred^T ym (MFMA,
//                          accumulated in registers), column sums for db_out and the BatchNorm backward
//   k_head_bwd_finalize    deterministic second stage: dW_out, db_out, d(bn weight), d(bn bias), BN constants
//   k_head_bn_bwd_apply<D> dX = bn_w invstd (dy - mean(dy) - xhat mean(dy xhat)) [x > 0]
// Training path of the stage engine (cgcn_head_train; 3 launches, none in the backward):
//   k_head_bn_finalize     on the per-tile statistics the last cgcn_layer_fwd emitted (colstats)
//   k_head_fused<D,CBMAX>  k_head_fwd + the tile-local half of k_head_bwd in one pass: d loss / d pred lives only
//                          in LDS; leaves dym and the per-workgroup partial sums in the workspace
//   (d = 128: k_head_fused_rs, two wave teams one tile apart, or -- the default -- k_head_fused_sp, the same with its three
//   products as split products on the bf16 matrix cores: cgcn_common.hpp)
//   k_head_train_finish    loss sum + BatchNorm-backward column means; every parameter sum of the head is then
//                          finished by extra workgroups of the layer backward (head_finalize_slab, cgcn_common.hpp)
#include "cgcn_common.hpp"

#define HEAD_STAT_BLOCKS 128
#define HEAD_TILE 16
#define HEADB_TILE 32
#ifndef HEAD_MAX_PARTIALS
#define HEAD_MAX_PARTIALS 256  // one 32-node tile per workgroup up to 8k nodes: the kernel is a latency chain per tile
#endif

// ------------------------------------------------------------------------------------------
// BatchNorm statistics
// ------------------------------------------------------------------------------------------
// (chan_combine: cgcn_common.hpp)

// Per-block (mean, M2) of relu(X) per column and strand over a contiguous chunk of nodes.  Sums are taken
// relative to a pivot (the chunk's first row), so there is no division in the loop and no catastrophic
// cancellation; blocks are merged exactly with Chan's formula in k_head_bn_finalize.
template <int D>
__global__ __launch_bounds__(256) void k_head_colstats(int n, int S, int rows_per_blk, const float* __restrict__ X,
                                                       float* __restrict__ part) {
  constexpr int RL = 256 / D;  // row lanes per column
  __shared__ float sm[2][256];
  const int c = threadIdx.x % D, rl = threadIdx.x / D;
  const int r0 = blockIdx.x * rows_per_blk, r1 = min(n, r0 + rows_per_blk);
  for (int s = 0; s < S; ++s) {
    const float* Xs = X + (size_t)s * n * D + c;
    const float pivot = r0 < n ? fmaxf(Xs[(size_t)r0 * D], 0.f) : 0.f;
    float s1 = 0.f, s2 = 0.f;
    int i = r0 + rl;
    for (; i + 3 * RL < r1; i += 4 * RL) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = Xs[(size_t)(i + u * RL) * D];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dlt = fmaxf(v[u], 0.f) - pivot;
        s1 += dlt;
        s2 += dlt * dlt;
      }
    }
    for (; i < r1; i += RL) {
      const float dlt = fmaxf(Xs[(size_t)i * D], 0.f) - pivot;
      s1 += dlt;
      s2 += dlt * dlt;
    }
    if (RL > 1) {
      if (rl > 0) { sm[0][threadIdx.x] = s1; sm[1][threadIdx.x] = s2; }
      __syncthreads();
      if (rl == 0) {
        for (int o = 1; o < RL; ++o) { s1 += sm[0][o * D + c]; s2 += sm[1][o * D + c]; }
      }
      __syncthreads();
    }
    if (rl == 0) {
      const float k = (float)max(r1 - r0, 1);
      float* p = part + (((size_t)blockIdx.x * S + s) * D + c) * 2;
      p[0] = pivot + s1 / k;
      p[1] = fmaxf(s2 - s1 * s1 / k, 0.f);
    }
  }
}

// Second stage of the statistics.  One workgroup = COLS (strand, channel) columns x 1024/COLS slices of the partial
// list: every thread Chan-combines its few partials (one batch of independent loads), the slices are merged through
// LDS in two fixed-order steps, and one thread per channel finishes both strands (the running-stat update is
// sequential in the strand index).  The kernel is a pure latency chain, so it is laid out wide and shallow:
// COLS = 16 for the <= 128 partials of k_head_colstats, COLS = 4 for the per-tile partials of cgcn_layer_fwd.
// Chan's merge in float64 (the second stage below): the batch mean is a per-COLUMN constant of every row's BatchNorm
// output, so its rounding error is coherent over the rows -- the same amplification channel as the BatchNorm-backward
// column means (cgcn_common.hpp, head_part_stride): merged in fp32, dW2.bias stayed at 1.4e-4 of the float64 truth.
__device__ __forceinline__ void chan_combine_d(double& nA, double& meanA, double& m2A, double nB, double meanB, double m2B) {
  const double nAB = nA + nB;
  if (nAB > 0.0) {
    const double delta = meanB - meanA;
    meanA += delta * (nB / nAB);
    m2A += m2B + delta * delta * (nA * nB / nAB);
    nA = nAB;
  }
}

template <int COLS>
__global__ __launch_bounds__(1024) void k_head_bn_finalize(int n, int S, int D, int nblk, int rows_per_blk,
                                                           const float* __restrict__ part, float momentum, float eps,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           long long* __restrict__ nbt, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd) {
  // Two passes over the records, no division inside them: mean = sum_b n_b m_b / n, then
  // M2 = sum_b [M2_b + n_b (m_b - mean)^2] (the exact identity Chan's pairwise formula telescopes to), everything in
  // float64.  A thread keeps its first 16 records in registers between the passes (<= 1 024 records: every size the
  // engine produces) and re-reads the rest.  Slices of one column: 4 per wave (lanes cl + 16 j) -> two DPP-free
  // shuffles, then the 16 waves through LDS in wave order.  (Round 2's pairwise Chan merges cost ~32 dependent float64
  // divisions per thread: 9-11 us per launch, now ~6.)
  constexpr int NSL = 1024 / COLS;          // slices
  constexpr int SPW = 64 / COLS;            // slices inside one wave (lanes cl, cl + COLS, ...)
  constexpr int NWV = 16;                   // waves
  static_assert(COLS * SPW == 64 && NSL == SPW * NWV, "thread layout");
  constexpr int KEEP = 16;
  __shared__ double sm[2][NWV][COLS + 1];
  __shared__ double stat[2][COLS + 1];      // mean, M2 of this workgroup's columns
  const int cl = threadIdx.x % COLS, slice = threadIdx.x / COLS, wave = threadIdx.x >> 6;
  const int CPB = COLS / S;  // channels per workgroup (S is 1 or 2)
  const int s = cl / CPB, c = blockIdx.x * CPB + cl % CPB;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += S;
  // the running statistics the tail updates: fetched now, under the record loads (they were a second dependent round trip)
  float rm = 0.f, rv = 0.f;
  if (slice == 0 && s == 0 && c < D) { rm = run_mean[c]; rv = run_var[c]; }
  const int per = (nblk + NSL - 1) / NSL;
  const int b0 = slice * per, b1 = min(nblk, b0 + per);
  auto rows_of = [&](int b) { return (double)max(0, min(n, (b + 1) * rows_per_blk) - b * rows_per_blk); };
  auto wg_sum = [&](double v, int which) -> double {   // sum over the 64 slices of column cl; every thread gets it
#pragma unroll
    for (int o = COLS; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) < COLS) sm[which][wave][cl] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) t += sm[which][w][cl];
    return t;
  };
  float km[KEEP], k2[KEEP];
  double s1 = 0.0;
  if (c < D) {
#pragma unroll
    for (int u = 0; u < KEEP; ++u) {
      const int bb = min(b0 + u, max(b1 - 1, 0));
      const float* p = part + (((size_t)bb * S + s) * D + c) * 2;
      km[u] = p[0];
      k2[u] = p[1];
    }
#pragma unroll
    for (int u = 0; u < KEEP; ++u)
      if (b0 + u < b1) s1 += rows_of(b0 + u) * (double)km[u];
    for (int b = b0 + KEEP; b < b1; ++b) s1 += rows_of(b) * (double)part[(((size_t)b * S + s) * D + c) * 2];
  }
  const double mean = wg_sum(s1, 0) / (double)n;
  double q = 0.0;
  if (c < D) {
#pragma unroll
    for (int u = 0; u < KEEP; ++u)
      if (b0 + u < b1) {
        const double dm = (double)km[u] - mean;
        q += (double)k2[u] + rows_of(b0 + u) * dm * dm;
      }
    for (int b = b0 + KEEP; b < b1; ++b) {
      const float* p = part + (((size_t)b * S + s) * D + c) * 2;
      const double dm = (double)p[0] - mean;
      q += (double)p[1] + rows_of(b) * dm * dm;
    }
  }
  const double m2 = wg_sum(q, 1);
  if (slice == 0) {
    stat[0][cl] = mean;
    stat[1][cl] = m2;
    if (c < D) {
      save_mean[s * D + c] = (float)mean;
      save_invstd[s * D + c] = (float)(1.0 / sqrt(m2 / (double)n + (double)eps));
    }
  }
  __syncthreads();
  if (slice != 0 || s != 0 || c >= D) return;
  for (int st = 0; st < S; ++st) {
    // sequential update: the reference calls the model on the forward strand, then the reverse one
    rm = (1.f - momentum) * rm + momentum * (float)stat[0][st * CPB + cl];
    rv = (1.f - momentum) * rv + momentum * (float)(stat[1][st * CPB + cl] / (double)(n - 1));
  }
  run_mean[c] = rm;
  run_var[c] = rv;
}

static void launch_bn_finalize(hipStream_t st, int n, int S, int d, int nblk, int rpb, const float* part, float momentum,
                               float eps, float* run_mean, float* run_var, long long* nbt, float* save_mean,
                               float* save_invstd) {
#ifndef BNFIN_WIDE_FROM
#define BNFIN_WIDE_FROM 2048
#endif
  if (nblk > BNFIN_WIDE_FROM)  // measured at 722 partials: 16 columns x 64 slices (2 load batches) 9.6 us, 4 x 256 (1 batch) 11 us
    hipLaunchKernelGGL(k_head_bn_finalize<4>, dim3((S * d + 3) / 4), dim3(1024), 0, st, n, S, d, nblk, rpb, part, momentum, eps,
                       run_mean, run_var, nbt, save_mean, save_invstd);
  else
    hipLaunchKernelGGL(k_head_bn_finalize<16>, dim3((S * d + 15) / 16), dim3(1024), 0, st, n, S, d, nblk, rpb, part, momentum,
                       eps, run_mean, run_var, nbt, save_mean, save_invstd);
}

// ------------------------------------------------------------------------------------------
// ym row of one node: mean over strands of dropout(BN(relu(x_s))).  Lane owns EPL consecutive columns.
// ------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void head_row(int n, int S, int i, int lane, const float* __restrict__ X,
                                         const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                         const float* __restrict__ mean, const float* __restrict__ vr, int use_running,
                                         float eps, float keep_scale, uint32_t thresh, uint32_t key,
                                         float (&ym)[D / 64]) {
  constexpr int EPL = D / 64;
#pragma unroll
  for (int e = 0; e < EPL; ++e) ym[e] = 0.f;
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane * EPL + e;
      const uint32_t el = (uint32_t)(((size_t)s * n + i) * D + c);
      const float x = X[el];
      const float mu = use_running ? mean[c] : mean[s * D + c];
      const float is = use_running ? rsqrtf(vr[c] + eps) : vr[s * D + c];
      float y = (fmaxf(x, 0.f) - mu) * is * bn_w[c] + bn_b[c];
      if (thresh) y = dropout_keep(key, el, thresh) ? y * keep_scale : 0.f;
      ym[e] += y;
    }
  }
  const float invS = 1.f / (float)S;
#pragma unroll
  for (int e = 0; e < EPL; ++e) ym[e] *= invS;
}

// ------------------------------------------------------------------------------------------
// k_head_fwd
// MFMA K index is permuted so that both operands are 16-byte reads and the four k-slots of a step read
// 64 contiguous bytes: step (t,u) of k-slot q (= l>>4) uses k = 16t + 4q + u for A (LDS) and B (W_out row).
// ------------------------------------------------------------------------------------------
template <int D, int NCBW>
__global__ __launch_bounds__(512) void k_head_fwd(int n, int S, int C, const float* __restrict__ X,
                                                  const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                  const float* __restrict__ mean, const float* __restrict__ vr,
                                                  int use_running, float eps, const float* __restrict__ Wout,
                                                  const float* __restrict__ bout, const float* __restrict__ target,
                                                  float keep_scale, uint32_t thresh,
                                                  const unsigned long long* __restrict__ rng_state, float inv_count,
                                                  float* __restrict__ probs, float* __restrict__ dpred,
                                                  float* __restrict__ loss_part, float* __restrict__ logits) {
  // logits != NULL (cgcn_head_logits): pred is the output; target, probs, dpred and loss_part are then NULL.
  // 8 waves; wave w owns label blocks w, w+8 (NCBW of them; C <= 128*NCBW) and rows w, w+8 of the 16-node tile.
  constexpr int R = HEAD_TILE, LD = D + 4, EPL = D / 64, NW = 8, KQ = D / 4, RPW = R / NW;
  constexpr bool PRE = (D == 128);
  __shared__ __attribute__((aligned(16))) float Y[R * LD];
  __shared__ float lsum[NW];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int node0 = blockIdx.x * R;
  const int r = lane & 15, q = lane >> 4;
  const int CB = (C + 15) / 16;

  // B fragments of W_out for this wave's label blocks, fetched before anything else (D = 128)
  float bw[NCBW][PRE ? KQ : 1];
  if (PRE) {
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi) {
      const int j = (wave + NW * cbi) * 16 + r;
#pragma unroll
      for (int t = 0; t < KQ / 4; ++t) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (j < C) v = *(const f32x4*)&Wout[(size_t)j * D + 16 * t + 4 * q];
#pragma unroll
        for (int u = 0; u < 4; ++u) bw[cbi][4 * t + u] = v[u];
      }
    }
  }
  const uint32_t key = thresh ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;

  // ---- ym tile: all X loads of this wave's rows first, then the BatchNorm / dropout arithmetic
  float xv[RPW][2][EPL];
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int i = node0 + wave + t * NW;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (i < n && s < S) ld_row<EPL>(xv[t][s], &X[((size_t)s * n + i) * D + lane * EPL]);
      else zero_row<EPL>(xv[t][s]);
    }
  }
  float mu[2][EPL], is[2][EPL], gw[EPL], gb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane * EPL + e;
    gw[e] = bn_w[c];
    gb[e] = bn_b[c];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int ss = s < S ? s : 0;
      mu[s][e] = use_running ? mean[c] : mean[ss * D + c];
      is[s][e] = use_running ? rsqrtf(vr[c] + eps) : vr[ss * D + c];
    }
  }
  const float invS = 1.f / (float)S;
#pragma unroll
  for (int t = 0; t < RPW; ++t) {
    const int rr = wave + t * NW;
    const int i = node0 + rr;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      float ym = 0.f;
      if (i < n) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < S) {
            float y = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e] * gw[e] + gb[e];
            if (thresh) y = dropout_keep(key, (uint32_t)(((size_t)s * n + i) * D + lane * EPL + e), thresh) ? y * keep_scale : 0.f;
            ym += y;
          }
        }
      }
      Y[rr * LD + lane * EPL + e] = ym * invS;
    }
  }
  // targets / bias of this lane's output elements: issued now, consumed after the MFMA phase
  float tgv[NCBW][4], bjv[NCBW];
#pragma unroll
  for (int cbi = 0; cbi < NCBW; ++cbi) {
    const int j = (wave + NW * cbi) * 16 + r;
    bjv[cbi] = j < C ? bout[j] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = node0 + q * 4 + e;
      tgv[cbi][e] = (target && i < n && j < C) ? target[(size_t)i * C + j] : 0.f;
    }
  }
  __syncthreads();

  // ---- pred = ym W_out^T on the matrix cores (K permuted, see above)
  f32x4 acc[NCBW];
#pragma unroll
  for (int cbi = 0; cbi < NCBW; ++cbi) acc[cbi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KQ / 4; ++t) {
    const f32x4 a = *(const f32x4*)&Y[r * LD + 16 * t + 4 * q];
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi) {
      if (wave + NW * cbi < CB) {
        f32x4 b;
        if (PRE) {
#pragma unroll
          for (int u = 0; u < 4; ++u) b[u] = bw[cbi][4 * t + u];
        } else {
          const int j = (wave + NW * cbi) * 16 + r;
          b = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (j < C) b = *(const f32x4*)&Wout[(size_t)j * D + 16 * t + 4 * q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[cbi] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[cbi], 0, 0, 0);
      }
    }
  }
  // ---- sigmoid / BCE / d loss / d pred epilogue straight from the accumulators
  float lacc = 0.f;
#pragma unroll
  for (int cbi = 0; cbi < NCBW; ++cbi) {
    const int cb = wave + NW * cbi;
    const int j = cb * 16 + r;
    if (cb < CB && j < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = node0 + q * 4 + e;
        if (i < n) {
          const float pred = acc[cbi][e] + bjv[cbi];
          if (logits) {
            logits[(size_t)i * C + j] = pred;
            continue;
          }
          const float en = __expf(-fabsf(pred));            // exp(-|x|) in (0,1]
          const float inv = 1.f / (1.f + en);
          const float p = pred >= 0.f ? inv : en * inv;      // sigmoid(x), no overflow
          lacc += fmaxf(pred, 0.f) - pred * tgv[cbi][e] + __logf(1.f + en);
          probs[(size_t)i * C + j] = p;
          if (dpred) dpred[(size_t)i * C + j] = (p - tgv[cbi][e]) * inv_count;
        }
      }
    }
  }
  if (logits) return;
  lacc = wave_sum(lacc);
  if (lane == 0) lsum[wave] = lacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += lsum[w];
    loss_part[blockIdx.x] = t;
  }
}

// out[0] = scale * sum(part[0..m)), one workgroup, fixed-order tree => deterministic
__global__ __launch_bounds__(256) void k_sum_scale(int m, const float* __restrict__ part, float scale, float* __restrict__ out) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < m; i += 256) s += part[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0] * scale;
}

// Accumulate mode of the training head kernels (k_head_fused, k_head_fused_rs; cgcn_common.hpp, STAT_ACC_*; sa.acc != nullptr):
// the batch statistics come from the integer totals
// cgcn_layer_fwd left, every lane decodes the columns it needs in its prologue (`mean` / `invstd` are then not read), and the
// first workgroup of the first label pass does what k_head_bn_finalize did besides: save_mean / save_invstd for the backward,
// the running statistics (forward strand, then reverse: the reference calls the model once per strand) and the call count.
struct HeadStatAcc {
  const unsigned long long* acc;
  float eps, momentum;
  float* run_mean;
  float* run_var;
  long long* nbt;
  float* save_mean;
  float* save_invstd;
  float* loss_out;   // the caller's loss word: written by the last workgroup to arrive (k_head_train_finish is not launched)
};
// Accumulate mode: the workgroup decodes the totals ONCE, cooperatively -- thread t the (strand, column) pair t: 16 eight-byte
// loads, an int64 sum, a float64 division and 1 / sqrt -- into an LDS stash [mean 2 D][invstd 2 D][unbiased variance 2 D]; every thread then picks
// the few columns it needs (head_stat, after a workgroup barrier).  (Round 5 / the first round-6 version had every thread
// decode its own columns: 4 pairs per thread of the d = 128 kernel, 8 at d = 256 -- 13 us of a 52 us launch there,
// profiles/r06_stat_acc_experiment.txt.)
template <int D>
__device__ __forceinline__ void head_stat_stage(const HeadStatAcc& sa, int n, int S, float* __restrict__ stash, int nthreads) {
  for (int idx = threadIdx.x; idx < S * D; idx += nthreads) {
    double m, m2;
    stat_acc_get(sa.acc, S, D, idx / D, idx % D, n, m, m2);
    stash[idx] = (float)m;
    stash[2 * D + idx] = (float)(1.0 / sqrt(m2 / (double)n + (double)sa.eps));
    stash[4 * D + idx] = (float)(m2 / (double)(n - 1));   // the unbiased variance of the running statistics (head_stat_bookkeeping)
  }
}
template <int D>
__device__ __forceinline__ void head_stat(const HeadStatAcc& sa, const float* __restrict__ stash, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, int s, int c, float& mu, float& is) {
  if (sa.acc) {
    mu = stash[s * D + c];
    is = stash[2 * D + s * D + c];
  } else {
    mu = mean[s * D + c];
    is = invstd[s * D + c];
  }
}

// The bookkeeping k_head_bn_finalize did besides (one wave of the first workgroup of the first label pass; lane l: columns
// l EPL .. l EPL + EPL - 1, both strands): save_mean / save_invstd for the backward, the running statistics (forward strand,
// then reverse: the reference calls the model once per strand), the call count -- from the workgroup's stash (head_stat_stage),
// behind its barrier: decoding the totals a second time held the first workgroup back by 2-3 us, the launch's serial tail.
template <int D>
__device__ __forceinline__ void head_stat_bookkeeping(const HeadStatAcc& sa, const float* __restrict__ stash, int S, int lane) {
  constexpr int EPL = D / 64;
  if (lane == 0 && sa.nbt) sa.nbt[0] += S;
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
    const int c = lane * EPL + u;
    float rm = sa.run_mean[c], rv = sa.run_var[c];
    for (int st = 0; st < S; ++st) {
      const float m = stash[st * D + c];
      sa.save_mean[st * D + c] = m;
      sa.save_invstd[st * D + c] = stash[2 * D + st * D + c];
      rm = (1.f - sa.momentum) * rm + sa.momentum * m;
      rv = (1.f - sa.momentum) * rv + sa.momentum * stash[4 * D + st * D + c];
    }
    sa.run_mean[c] = rm;
    sa.run_var[c] = rv;
  }
}
// This workgroup's share of the loss joins the total as a fixed-point integer, and the SAME atomic draws its ticket (the share
// sits above bit 12, the arrival count below: BCE terms are >= 0, so nothing borrows): whoever finds gridDim.x - 1 earlier
// arrivals in the returned word holds the complete total -- every other share was added by an atomic that precedes this one at
// the memory side -- and writes the caller's loss.  One round trip at the workgroup's tail instead of three.
// "Something went wrong" travels in the SAME word, as bit 63 (BACC_LOSS_BAD; the sum stays far below it): a share that is
// NaN / absurd (statistics outside the fixed-point range upstream) ORs it in before its add, and so does bacc_add for a
// backward sum that does not fit (the thread that does is ordered before this workgroup's ticket by the workgroup barrier in
// between).  Operations on ONE word are totally ordered, so the last arriver sees every such OR -- no fence: an agent-scope
// fence writes the L2 back on this chip (private L2 per XCD) and cost 40 us per launch when every thread issued one
// (profiles/r06_stat_acc_experiment.txt); round 5 kept the flag in a word of its own, whose update could pass the ticket's.
// ONE thread of the workgroup calls this, behind a workgroup barrier.
__device__ __forceinline__ void head_loss_ticket(const HeadStatAcc& sa, int S, int D, float tl, float inv_count) {
  unsigned long long* h = bacc_base(sa.acc, S, D) + (size_t)STAT_ACC_SLOTS * S * D * 2;
  const bool bad = !(tl >= 0.f && tl < 1e9f);   // the total of <= 4 095 realistic shares stays far below 2^36 (x 2^16 x 2^12 = 2^64)
  if (bad) {
    atomicOr(&h[BACC_FLAG], 2ull);              // (for the consumers behind the kernel boundary)
    atomicOr(&h[BACC_LOSS], BACC_LOSS_BAD);
  }
  const unsigned long long mine = ((bad ? 0ull : (unsigned long long)__builtin_rint(ldexp((double)tl, BACC_LOSS_FBITS))) << 12) | 1ull;
  const unsigned long long old = atomicAdd(&h[BACC_LOSS], mine);
  if ((old & 0xFFFull) == (unsigned long long)gridDim.x - 1ull) {
    const unsigned long long tot = old + mine;
    sa.loss_out[0] = (tot & BACC_LOSS_BAD) ? __builtin_nanf("")
                                           : (float)(ldexp((double)((tot & ~BACC_LOSS_BAD) >> 12), -BACC_LOSS_FBITS) * (double)inv_count);
  }
}
// The binary points of the backward sums (cgcn_common.hpp): from max |W_out| over ALL labels (per-wave maxima in wmax[nw]).
__device__ __forceinline__ void head_bacc_points(const float* wmax, int nw, float keep_scale, int n, int& fa, int& fb) {
  float mw = 0.f;
  for (int w = 0; w < nw; ++w) mw = fmaxf(mw, wmax[w]);
  const double bound = fmax((double)keep_scale * (double)mw, 1e-30) * 1.001;
  fa = bacc_fbits(bound);
  fb = bacc_fbits(bound * sqrt((double)n) * 1.001);
}
__device__ __forceinline__ float head_wout_absmax(const float* __restrict__ Wout, int count4, int nthreads) {
  float mw = 0.f;
  for (int idx = threadIdx.x; idx < count4; idx += nthreads) {
    const f32x4 v = *(const f32x4*)&Wout[(size_t)idx * 4];
    mw = fmaxf(fmaxf(mw, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, WAVE));
  return mw;
}

// Cross-wave merge of the per-thread BatchNorm-backward column sums into the workgroup's partial record, in float64
// from here on (a thread's own sum covers 4 rows per tile -- a few dozen fp32 terms whose rounding is random across
// threads; it is the LARGE partial sums of the later stages whose rounding is coherent over a column, cgcn_common.hpp):
// two rounds (sum dy, then sum dy*xhat) through an LDS scratch of NW * 2 * D doubles, waves added in a fixed order.
// Accumulate mode (bacc != nullptr): the same totals also join the integer totals of the statistics buffer (bacc_add; binary
// points fa / fb), for the row-local backward's prologue.
template <int D, int NW, int SCRATCH_FLOATS>
__device__ __forceinline__ void head_stats_partial(float* __restrict__ scratch, double* __restrict__ out,
                                                   const float (&sdy)[2][D / 64], const float (&sdyx)[2][D / 64],
                                                   int wave, int lane, unsigned long long* bacc = nullptr, int S = 0,
                                                   int fa = 0, int fb = 0) {
  constexpr int EPL = D / 64, RS = 2 * D;
  static_assert(NW * RS * 2 <= SCRATCH_FLOATS, "reduction scratch must fit in the tile buffer");
  static_assert(RS == NW * 64 || RS == 2 * NW * 64 || RS * 2 == NW * 64, "a thread's columns are the same in both rounds");
  double* red = (double*)scratch;
  double keep[2] = {0.0, 0.0};   // round 0's total of this thread's first two (strand, column) pairs (RS <= 2 NW 64)
  for (int round = 0; round < 2; ++round) {
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < EPL; ++e) red[wave * RS + s * D + lane * EPL + e] = (double)(round ? sdyx[s][e] : sdy[s][e]);
    __syncthreads();
    int k = 0;
    for (int c = threadIdx.x; c < RS; c += NW * 64, ++k) {
      double t = 0.0;
      for (int w = 0; w < NW; ++w) t += red[w * RS + c];
      out[round * RS + c] = t;
      if (bacc) {
        if (round == 0) keep[k & 1] = t;
        else if (c / D < S) bacc_add(bacc, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), c / D, c % D, keep[k & 1], t, fa, fb);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_head_bwd: persistent over 32-node tiles.
//   partial layout per workgroup: head_part_stride (cgcn_common.hpp): [CPT*D dW_out][CPT db_out][4*D float64 column sums]
// Labels are walked in PASSES of at most 128 (one launch per pass, labels [c0, c0 + Cp)): the accumulators of a
// 256-label variant do not fit the register file (112 / 70 / 13 spilled registers in round 2's <*,16> instantiations).
// A pass adds its share of dym = dpred W_out to what the earlier passes left in memory; the LAST pass, which sees the
// complete dym, takes the BatchNorm-backward column sums.  CPT = padded label count of the partial layout (128 / 256).
// ------------------------------------------------------------------------------------------
template <int D, int CBMAX>
__global__ __launch_bounds__(512) void k_head_bwd(int n, int S, int C, const float* __restrict__ X,
                                                  const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  const float* __restrict__ Wout, const float* __restrict__ dpred,
                                                  const float* __restrict__ dloss, float keep_scale, uint32_t thresh,
                                                  const unsigned long long* __restrict__ rng_state,
                                                  float* __restrict__ dym, float* __restrict__ part,
                                                  int c0, int Cp, int CPT, int first, int last) {
  constexpr int TR = HEADB_TILE, NW = 8, EPL = D / 64, RPW = TR / NW;
  constexpr int CP = CBMAX * 16;
  constexpr int LDP = CP + ((CP & 16) ? 2 : 18);  // = 18 (mod 32): row reads and transposed reads both (nearly) conflict-free
  constexpr int LDY = D + 16;                      // = 16 (mod 32): conflict-free transposed reads
  constexpr int JBW = D / 128;                     // 16-wide column blocks of D owned by one wave
  const int PS = head_part_stride(CPT, D);
  constexpr int NLD = TR * CP / 512;               // dpred elements staged per thread per tile
  __shared__ __attribute__((aligned(16))) float Pt[TR * LDP];
  __shared__ __attribute__((aligned(16))) float Yt[TR * LDY];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const uint32_t key = thresh ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;
  const float gl = dloss ? dloss[0] : 1.f;
  const int CB = (Cp + 15) / 16;
  const float invS = 1.f / (float)S;

  f32x4 accW[CBMAX][JBW];
#pragma unroll
  for (int ib = 0; ib < CBMAX; ++ib)
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb) accW[ib][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbo = 0.f;                       // thread j < CP: column sum of dpred
  float sdy[2][EPL], sdyx[2][EPL];   // per thread: <= a few dozen rows in fp32; float64 from the cross-wave merge on
  float mu[2][EPL], is[2][EPL], gw[EPL], gb[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane * EPL + e;
    gw[e] = bn_w[c];
    gb[e] = bn_b[c];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sdy[s][e] = sdyx[s][e] = 0.f;
      mu[s][e] = mean[(s < S ? s : 0) * D + c];
      is[s][e] = invstd[(s < S ? s : 0) * D + c];
    }
  }

  // B operand of dym = dpred W_out: B[k = label][col] -- fetched once per workgroup (D = 128; at D = 256
  // the register budget goes to the accumulators and it is read in the K loop instead)
  constexpr bool PREB = (D == 128);
  float bwo[JBW][PREB ? CBMAX * 4 : 1];
  if (PREB) {
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
      for (int kk = 0; kk < CBMAX * 4; ++kk) {
        const int k = 4 * kk + q;
        bwo[jb][kk] = (k < Cp) ? Wout[(size_t)(c0 + k) * D + (wave * JBW + jb) * 16 + r] : 0.f;
      }
  }

  const int ntiles = (n + TR - 1) / TR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int node0 = tile * TR;
    // ---- issue every global load of the tile first: dpred elements and this wave's X rows
    float pv[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + u * 512;
      const int row = idx / CP, j = idx % CP;
      const int i = node0 + row;
      pv[u] = (i < n && j < Cp) ? dpred[(size_t)i * C + c0 + j] : 0.f;
    }
    float xv[RPW][2][EPL];
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int i = node0 + wave + t * NW;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (i < n && s < S) ld_row<EPL>(xv[t][s], &X[((size_t)s * n + i) * D + lane * EPL]);
        else zero_row<EPL>(xv[t][s]);
      }
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int idx = threadIdx.x + u * 512;
      Pt[(idx / CP) * LDP + (idx % CP)] = pv[u] * gl;
    }
    // ym rows (mean over strands of the dropped BatchNorm output), recomputed
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int rr = wave + t * NW;
      const int i = node0 + rr;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float ym = 0.f;
        if (i < n) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s < S) {
              float y = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e] * gw[e] + gb[e];
              if (thresh) y = dropout_keep(key, (uint32_t)(((size_t)s * n + i) * D + lane * EPL + e), thresh) ? y * keep_scale : 0.f;
              ym += y;
            }
          }
        }
        Yt[rr * LDY + lane * EPL + e] = ym * invS;
      }
    }
    __syncthreads();
#ifndef HB_SKIP_DBO
    if (threadIdx.x < CP) {
      float sacc = 0.f;
#pragma unroll 8
      for (int row = 0; row < TR; ++row) sacc += Pt[row * LDP + threadIdx.x];
      dbo += sacc;
    }
#endif
#ifndef HB_SKIP_DW
    // ---- dW_out += Pt^T Yt   (K = TR rows)
#pragma unroll 2
    for (int kk = 0; kk < TR / 4; ++kk) {
      const int k = 4 * kk + q;
      float b[JBW];
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) b[jb] = Yt[k * LDY + (wave * JBW + jb) * 16 + r];
#pragma unroll
      for (int ib = 0; ib < CBMAX; ++ib) {
        if (ib < CB) {
          const float a = Pt[k * LDP + ib * 16 + r];
#pragma unroll
          for (int jb = 0; jb < JBW; ++jb) accW[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[jb], accW[ib][jb], 0, 0, 0);
        }
      }
    }
#endif
    // ---- dym tile = Pt W_out   (M = TR rows, K = labels, N = D)
    f32x4 accY[2][JBW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) accY[mb][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef HB_SKIP_DYM
    if (PREB) {
#pragma unroll
      for (int kk = 0; kk < CBMAX * 4; ++kk) {
        if (kk < CB * 4) {  // wave-uniform
          const int k = 4 * kk + q;  // label index
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float a = Pt[(mb * 16 + r) * LDP + k];
#pragma unroll
            for (int jb = 0; jb < JBW; ++jb)
              accY[mb][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bwo[jb][PREB ? kk : 0], accY[mb][jb], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 2
      for (int kk = 0; kk < CB * 4; ++kk) {
        const int k = 4 * kk + q;
        float b[JBW];
#pragma unroll
        for (int jb = 0; jb < JBW; ++jb) b[jb] = (k < Cp) ? Wout[(size_t)(c0 + k) * D + (wave * JBW + jb) * 16 + r] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float a = Pt[(mb * 16 + r) * LDP + k];
#pragma unroll
          for (int jb = 0; jb < JBW; ++jb) accY[mb][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[jb], accY[mb][jb], 0, 0, 0);
        }
      }
    }
#endif
    __syncthreads();  // all reads of Yt / Pt done
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) Yt[(mb * 16 + q * 4 + e) * LDY + (wave * JBW + jb) * 16 + r] = accY[mb][jb][e];
    __syncthreads();
    // ---- row pass: write dym, accumulate the BatchNorm-backward column sums (X rows still in registers)
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int rr = wave + t * NW;
      const int i = node0 + rr;
      float gv[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) gv[e] = i < n ? Yt[rr * LDY + lane * EPL + e] : 0.f;
      if (i < n && !first) {   // the earlier label passes' share of dym
        float prev[EPL];
        ld_row<EPL>(prev, &dym[(size_t)i * D + lane * EPL]);
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] += prev[e];
      }
      if (i < n) st_row<EPL>(&dym[(size_t)i * D + lane * EPL], gv);
      if (!last) continue;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int c = lane * EPL + e;
        const float g = gv[e];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < S) {
            float dy = g * invS;
            if (thresh) dy = dropout_keep(key, (uint32_t)(((size_t)s * n + i) * D + c), thresh) ? dy * keep_scale : 0.f;
            const float xh = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e];
            sdy[s][e] += dy;
            sdyx[s][e] += dy * xh;
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- partials
  float* P = part + (size_t)blockIdx.x * PS;
#ifndef HB_SKIP_PART
#pragma unroll
  for (int ib = 0; ib < CBMAX; ++ib)
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + ib * 16 < CPT) P[(size_t)(c0 + ib * 16 + q * 4 + e) * D + (wave * JBW + jb) * 16 + r] = accW[ib][jb][e];
#endif
  if (threadIdx.x < CP && c0 + (int)threadIdx.x < CPT) P[CPT * D + c0 + threadIdx.x] = dbo;
  if (last) head_stats_partial<D, NW, TR * LDY>(Yt, (double*)(P + CPT * D + CPT), sdy, sdyx, wave, lane);
}

// ------------------------------------------------------------------------------------------
// k_head_fused: training-mode head forward AND the tile-local part of its backward in one pass.
// d loss / d pred = (sigmoid(pred) - target) / (n C) is known as soon as a tile's logits are, and everything the
// head's backward does with it is tile-local (dym = dpred W_out, dW_out += dpred^T ym, db_out, the BatchNorm
// column sums) -- only the *consumers* of those sums need a global barrier.  So the forward kernel keeps going:
// no dpred round trip through memory, no second read of X, no recomputation of ym / dropout masks, one launch
// less.  Everything is computed for an upstream gradient of 1; cgcn_head_bwd / cgcn_layer_bwd scale by the
// actual d loss (all of it is linear in that scalar).
// Persistent over 32-node tiles like k_head_bwd; same partial layout, same label passes (labels [c0, c0 + Cp) per launch,
// Cp <= 128; a pass adds its share of dym to the earlier passes', the last one takes the BatchNorm column sums).
// ------------------------------------------------------------------------------------------
#ifdef HF_TIMING  // tuning build only (tools/khead.py --stamps): phase timestamps of a few workgroups
__device__ unsigned long long hf_stamps[8 * 16];
#define HF_STAMP(i)                                                                      \
  do {                                                                                   \
    __builtin_amdgcn_s_waitcnt(0);                                                       \
    if (threadIdx.x == 0 && (blockIdx.x & 31) == 0) hf_stamps[(blockIdx.x >> 5) * 16 + (i)] = wall_clock64(); \
  } while (0)
#else
#define HF_STAMP(i)
#endif

#ifndef HF_LOWREG
#define HF_LOWREG 0   // experiment: no register-resident W_out fragments / no tile prefetch, <= 128 VGPRs, two workgroups per CU
#endif
template <int D, int CBMAX>
__global__ __launch_bounds__(512, HF_LOWREG ? 4 : 1) void k_head_fused(int n, int S, int C, const float* __restrict__ X,
                                                    const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    const float* __restrict__ Wout, const float* __restrict__ bout,
                                                    const float* __restrict__ target, float keep_scale, uint32_t thresh,
                                                    const unsigned long long* __restrict__ rng_state, float inv_count,
                                                    float* __restrict__ probs, float* __restrict__ loss_part,
                                                    float* __restrict__ dym, float* __restrict__ part,
                                                    int c0, int Cp, int CPT, int first, int last, HeadStatAcc sa) {
  constexpr int TR = HEADB_TILE, NW = 8, EPL = D / 64, RPW = TR / NW, KQ = D / 4;
  constexpr int CP = CBMAX * 16, NCBW = CBMAX / 8;
  constexpr int LDP = CP + ((CP & 16) ? 2 : 18);
  constexpr int LDY = D + 16;
  constexpr int JBW = D / 128;
  constexpr bool PRE = (D == 128) && !HF_LOWREG;
  __shared__ __attribute__((aligned(16))) float Pt[TR * LDP];
  __shared__ __attribute__((aligned(16))) float Yt[TR * LDY];
  __shared__ float lsum[NW];
  __shared__ float wmax[NW];   // accumulate mode: per-wave max |W_out| (the binary points of the backward sums)

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, q = lane >> 4;
  const uint32_t key = thresh ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;
  const int CB = (Cp + 15) / 16;
  const float invS = 1.f / (float)S;
  const int PS = head_part_stride(CPT, D);
  if (sa.acc && last) {   // max |W_out| over ALL labels (the earlier label passes' share of dym is in the sums too)
    const float mw = head_wout_absmax(Wout, C * (D / 4), NW * 64);
    if (lane == 0) wmax[wave] = mw;
  }

  HF_STAMP(0);
  HF_STAMP(1);
  // operand fragments of W_out, fetched once per workgroup (D = 128): B of pred = ym W_out^T and B of dym = dpred W_out
  float bw[NCBW][PRE ? KQ : 1];
  float bwo[JBW][PRE ? CBMAX * 4 : 1];
  if (PRE) {
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi) {
      const int j = (wave + NW * cbi) * 16 + r;
#pragma unroll
      for (int t = 0; t < KQ / 4; ++t) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (j < Cp) v = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + 16 * t + 4 * q];
#pragma unroll
        for (int u = 0; u < 4; ++u) bw[cbi][4 * t + u] = v[u];
      }
    }
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
      for (int kk = 0; kk < CBMAX * 4; ++kk) {
        const int k = 4 * kk + q;
        bwo[jb][kk] = (k < Cp) ? Wout[(size_t)(c0 + k) * D + (wave * JBW + jb) * 16 + r] : 0.f;
      }
  }
  HF_STAMP(2);
  f32x4 accW[CBMAX][JBW];
#pragma unroll
  for (int ib = 0; ib < CBMAX; ++ib)
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb) accW[ib][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbo = 0.f, lacc = 0.f;
  float sdy[2][EPL], sdyx[2][EPL];   // per thread: <= a few dozen rows in fp32; float64 from the cross-wave merge on
  float mu[2][EPL], is[2][EPL], gw[EPL], gb[EPL];
  if (sa.acc) {   // accumulate mode: batch mean / invstd decoded from the integer totals, once per workgroup (head_stat_stage)
    static_assert(TR * LDY >= 6 * D, "the stash fits the Y tile");
    head_stat_stage<D>(sa, n, S, Yt, NW * 64);
    __syncthreads();
    // one wave of the first workgroup of the first label pass does k_head_bn_finalize's bookkeeping
    if (blockIdx.x == 0 && first && wave == NW - 1) head_stat_bookkeeping<D>(sa, Yt, S, lane);
  }
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int c = lane * EPL + e;
    gw[e] = bn_w[c];
    gb[e] = bn_b[c];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sdy[s][e] = sdyx[s][e] = 0.f;
      head_stat<D>(sa, Yt, mean, invstd, s < S ? s : 0, c, mu[s][e], is[s][e]);   // (accumulate mode: from the stash in Yt)
    }
  }
  if (sa.acc) __syncthreads();   // the stash has been read: Yt becomes the tile

  const int ntiles = (n + TR - 1) / TR;
  // All global loads of a tile -- this wave's X rows, the targets / bias of this lane's logits -- are issued together.
  // Workgroups that walk several tiles (large chromosomes) issue the NEXT tile's loads right after the current tile's
  // operands are in LDS, so they are in flight during the three MFMA phases; the current tile keeps its copy.
  constexpr bool PF = (CBMAX == 8) && !HF_LOWREG;  // the 256-label variant has no registers to spare for a second tile
  float xv[RPW][2][EPL], xv_n[PF ? RPW : 1][2][EPL];
  float tgv[NCBW][2][4], tgv_n[PF ? NCBW : 1][2][4], bjv[NCBW];
#pragma unroll
  for (int cbi = 0; cbi < NCBW; ++cbi) {
    const int j = (wave + NW * cbi) * 16 + r;
    bjv[cbi] = j < Cp ? bout[c0 + j] : 0.f;
  }
  auto load_tile = [&](int tile, auto& xd, auto& td) {
    const int node0 = tile * TR;
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int i = node0 + wave + t * NW;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (i < n && s < S) ld_row<EPL>(xd[t][s], &X[((size_t)s * n + i) * D + lane * EPL]);
        else zero_row<EPL>(xd[t][s]);
      }
    }
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi) {
      const int j = (wave + NW * cbi) * 16 + r;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = node0 + mb * 16 + q * 4 + e;
          td[cbi][mb][e] = (i < n && j < Cp) ? target[(size_t)i * C + c0 + j] : 0.f;
        }
    }
  };
  if (PF && (int)blockIdx.x < ntiles) load_tile(blockIdx.x, xv_n, tgv_n);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int node0 = tile * TR;
    if (PF) {
#pragma unroll
      for (int t = 0; t < RPW; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < EPL; ++e) xv[t][s][e] = xv_n[PF ? t : 0][s][e];
#pragma unroll
      for (int cbi = 0; cbi < NCBW; ++cbi)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) tgv[cbi][mb][e] = tgv_n[PF ? cbi : 0][mb][e];
    } else {
      load_tile(tile, xv, tgv);
    }
    HF_STAMP(3);
    // ---- ym rows -> Yt
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int rr = wave + t * NW;
      const int i = node0 + rr;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        float ym = 0.f;
        if (i < n) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (s < S) {
              float y = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e] * gw[e] + gb[e];
              if (thresh) y = dropout_keep(key, (uint32_t)(((size_t)s * n + i) * D + lane * EPL + e), thresh) ? y * keep_scale : 0.f;
              ym += y;
            }
          }
        }
        Yt[rr * LDY + lane * EPL + e] = ym * invS;
      }
    }
    if (PF && tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x, xv_n, tgv_n);  // prefetch (see above)
    __syncthreads();
    HF_STAMP(4);
    // ---- pred = ym W_out^T  (M = 32 rows, K = D permuted, N = this wave's label block(s))
    f32x4 acc[NCBW][2];
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[cbi][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KQ / 4; ++t) {
      f32x4 a[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) a[mb] = *(const f32x4*)&Yt[(mb * 16 + r) * LDY + 16 * t + 4 * q];
#pragma unroll
      for (int cbi = 0; cbi < NCBW; ++cbi) {
        if (wave + NW * cbi < CB) {
          f32x4 b;
          if (PRE) {
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = bw[cbi][PRE ? 4 * t + u : 0];
          } else {
            const int j = (wave + NW * cbi) * 16 + r;
            b = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (j < Cp) b = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + 16 * t + 4 * q];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[cbi][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], b[u], acc[cbi][mb], 0, 0, 0);
        }
      }
    }
    HF_STAMP(5);
    // ---- sigmoid / BCE; d loss / d pred goes straight into the LDS tile (zero outside the valid region)
#pragma unroll
    for (int cbi = 0; cbi < NCBW; ++cbi) {
      const int j = (wave + NW * cbi) * 16 + r;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = mb * 16 + q * 4 + e;
          const int i = node0 + row;
          float dp = 0.f;
          if (i < n && j < Cp) {
            const float pred = acc[cbi][mb][e] + bjv[cbi];
            const float en = __expf(-fabsf(pred));
            const float inv = 1.f / (1.f + en);
            const float p = pred >= 0.f ? inv : en * inv;
            lacc += fmaxf(pred, 0.f) - pred * tgv[cbi][mb][e] + __logf(1.f + en);
            probs[(size_t)i * C + c0 + j] = p;
            dp = (p - tgv[cbi][mb][e]) * inv_count;
          }
          Pt[row * LDP + j] = dp;
        }
    }
    __syncthreads();
    HF_STAMP(6);
    if (threadIdx.x < CP) {
      float sacc = 0.f;
#pragma unroll 8
      for (int row = 0; row < TR; ++row) sacc += Pt[row * LDP + threadIdx.x];
      dbo += sacc;
    }
    // ---- dW_out += Pt^T Yt   (K = TR rows)
#pragma unroll 2
    for (int kk = 0; kk < TR / 4; ++kk) {
      const int k = 4 * kk + q;
      float b[JBW];
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) b[jb] = Yt[k * LDY + (wave * JBW + jb) * 16 + r];
#pragma unroll
      for (int ib = 0; ib < CBMAX; ++ib) {
        if (ib < CB) {
          const float a = Pt[k * LDP + ib * 16 + r];
#pragma unroll
          for (int jb = 0; jb < JBW; ++jb) accW[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[jb], accW[ib][jb], 0, 0, 0);
        }
      }
    }
    HF_STAMP(7);
    // ---- dym tile = Pt W_out   (M = TR rows, K = labels, N = D)
    f32x4 accY[2][JBW];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb) accY[mb][jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PRE) {
#pragma unroll
      for (int kk = 0; kk < CBMAX * 4; ++kk) {
        if (kk < CB * 4) {
          const int k = 4 * kk + q;
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const float a = Pt[(mb * 16 + r) * LDP + k];
#pragma unroll
            for (int jb = 0; jb < JBW; ++jb)
              accY[mb][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bwo[jb][PRE ? kk : 0], accY[mb][jb], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 2
      for (int kk = 0; kk < CB * 4; ++kk) {
        const int k = 4 * kk + q;
        float b[JBW];
#pragma unroll
        for (int jb = 0; jb < JBW; ++jb) b[jb] = (k < Cp) ? Wout[(size_t)(c0 + k) * D + (wave * JBW + jb) * 16 + r] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const float a = Pt[(mb * 16 + r) * LDP + k];
#pragma unroll
          for (int jb = 0; jb < JBW; ++jb) accY[mb][jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[jb], accY[mb][jb], 0, 0, 0);
        }
      }
    }
    HF_STAMP(8);
    __syncthreads();  // all reads of Yt / Pt done
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) Yt[(mb * 16 + q * 4 + e) * LDY + (wave * JBW + jb) * 16 + r] = accY[mb][jb][e];
    __syncthreads();
    HF_STAMP(9);
    // ---- row pass: write dym, accumulate the BatchNorm-backward column sums (X rows still in registers)
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
      const int rr = wave + t * NW;
      const int i = node0 + rr;
      float gv[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) gv[e] = i < n ? Yt[rr * LDY + lane * EPL + e] : 0.f;
      if (i < n && !first) {   // the earlier label passes' share of dym
        float prev[EPL];
        ld_row<EPL>(prev, &dym[(size_t)i * D + lane * EPL]);
#pragma unroll
        for (int e = 0; e < EPL; ++e) gv[e] += prev[e];
      }
      if (i < n) st_row<EPL>(&dym[(size_t)i * D + lane * EPL], gv);
      if (!last) continue;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int c = lane * EPL + e;
        const float g = gv[e];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s < S) {
            float dy = g * invS;
            if (thresh) dy = dropout_keep(key, (uint32_t)(((size_t)s * n + i) * D + c), thresh) ? dy * keep_scale : 0.f;
            const float xh = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e];
            sdy[s][e] += dy;
            sdyx[s][e] += dy * xh;
          }
        }
      }
    }
    __syncthreads();
  }

  HF_STAMP(10);
  // ---- partials (same layout as k_head_bwd) + this workgroup's share of the loss
  float* P = part + (size_t)blockIdx.x * PS;
#pragma unroll
  for (int ib = 0; ib < CBMAX; ++ib)
#pragma unroll
    for (int jb = 0; jb < JBW; ++jb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c0 + ib * 16 < CPT) P[(size_t)(c0 + ib * 16 + q * 4 + e) * D + (wave * JBW + jb) * 16 + r] = accW[ib][jb][e];
  if (threadIdx.x < CP && c0 + (int)threadIdx.x < CPT) P[CPT * D + c0 + threadIdx.x] = dbo;
  HF_STAMP(11);
  lacc = wave_sum(lacc);
  if (lane == 0) lsum[wave] = lacc;
  if (last) {
    unsigned long long* bb = sa.acc ? bacc_base(sa.acc, S, D) : nullptr;
    int fa = 0, fb = 0;
    if (sa.acc) {   // (wmax: written in the prologue, every workgroup barrier of the tile loop in between)
      head_bacc_points(wmax, NW, keep_scale, n, fa, fb);
      if (blockIdx.x == 0 && threadIdx.x == 0)
        bb[(size_t)STAT_ACC_SLOTS * S * D * 2 + BACC_EXP] = ((unsigned long long)(unsigned)(fb + 1024) << 32) | (unsigned long long)(unsigned)(fa + 1024);
    }
    head_stats_partial<D, NW, TR * LDY>(Yt, (double*)(P + CPT * D + CPT), sdy, sdyx, wave, lane, bb, S, fa, fb);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += lsum[w];
    const float tl = first ? t : loss_part[blockIdx.x] + t;   // the label passes' shares add up
    loss_part[blockIdx.x] = tl;
    if (sa.acc && last) head_loss_ticket(sa, S, D, tl, inv_count);   // k_head_train_finish is not launched
  }
  HF_STAMP(12);
}

// ------------------------------------------------------------------------------------------
// k_head_fused_rs (D = 128): k_head_fused as ONE 16-wave workgroup per CU with the waves split into two ROLES that
// work on consecutive tiles at the same time, so that the vector work of one tile (BatchNorm rows, sigmoid / BCE
// epilogue) runs under the matrix products of the other instead of between them.  k_head_fused (8 waves, 222 VGPRs,
// one workgroup per CU) spent 13 us per 32-node tile of which 4.5 us are the three fp32 products' floor.
//   P waves 0-7  ("pred"): X rows + targets of tile t -> ym rows -> Yt[t & 1];  pred = ym W_out^T (wave w: label block w);
//                 sigmoid / BCE / probs;  d loss / d pred -> Pt[t & 1];  db_out.
//   Q waves 8-15 ("grad"): one tile behind:  dW_out += Pt^T Yt (wave w: feature columns 16 (w-8) .., all label blocks:
//                 28-32 accumulator registers);  dym = Pt W_out written straight from the accumulators;  the
//                 BatchNorm-backward column sums from the accumulator layout (a lane owns ONE column: no cross-wave
//                 merge), X re-read in that layout (L2 hits).
// Period k, three sub-phases, in each of which one team is on the matrix pipe and the other on the vector pipe:
//     S1:  P  ym(tile k) -> Yt            (vector)   |  Q  dym product of tile k-1              (matrix)
//     S2:  P  pred product of tile k      (matrix)   |  Q  dym out, BatchNorm sums of tile k-1  (vector)
//     S3:  P  sigmoid / BCE -> Pt, probs  (vector)   |  Q  dW_out product of tile k-1           (matrix)
// (with both teams' products in one phase and both epilogues in the next, the first version of this kernel ran at the
// 8-wave kernel's speed: the matrix phases contended, the vector phases did not overlap anything)
// W_out (this pass's <= 128 label rows) sits in LDS for the whole launch and feeds the B operands of both products
// that use it: 16 waves share ONE copy instead of holding 64 KB of fragments in registers, which is what lets both roles
// stay under 128 registers (the roles' persistent state -- P: X rows, targets, BatchNorm constants; Q: dW_out
// accumulators -- lives in the SAME 40 registers).  LDS: W_out 72 KB + two Yt + two Pt tiles 73 KB.
// Same partial records, label passes and results as k_head_fused.
// ------------------------------------------------------------------------------------------
#if (defined(HRSX_NOBACC) || defined(HRSX_NOSTAGE) || defined(HRSX_NOTICKET)) && !defined(CGCN_EXPERIMENT_BUILD)
#error "HRSX_* are decomposition switches (garbage results): build a variant with -DCGCN_EXPERIMENT_BUILD (tools/mkvariant.py), never the shipped library"
#endif
#ifndef HEAD_RS
#define HEAD_RS 1   // 0: the 8-wave k_head_fused at d = 128 as well (A/B builds)
#endif
#ifndef HEAD_NT_PROBS
#define HEAD_NT_PROBS 0   // non-temporal probs stores in k_head_fused_rs (A/B: profiles/r05_head_experiments.txt)
#endif
#ifndef HEAD_RS_PRIO
#define HEAD_RS_PRIO 0   // wave priority inside the matrix sub-phases (measured: 2 is neutral for the head, -2 % for the row-local kernel)
#endif
#ifdef RS_TIMING  // tuning build only (tools/khead_train.py --stamps): phase timestamps of period RS_STAMP_K of a few workgroups
__device__ unsigned long long rs_stamps[8 * 2 * 8];
#ifndef RS_STAMP_K
#define RS_STAMP_K 1
#endif
#define RS_STAMP(i)                                                                                        \
  do {                                                                                                     \
    __builtin_amdgcn_s_waitcnt(0);                                                                         \
    if (k == RS_STAMP_K && (threadIdx.x == 0 || threadIdx.x == 512) && (blockIdx.x & 31) == 0)             \
      rs_stamps[((blockIdx.x >> 5) * 2 + (threadIdx.x >> 9)) * 8 + (i)] = wall_clock64();                  \
  } while (0)
extern "C" int cgcn_debug_rs_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rs_stamps), sizeof(unsigned long long) * 8 * 2 * 8) == hipSuccess ? 0 : -1;
}
#else
#define RS_STAMP(i)
#endif
// MULTI: more than one label pass (C > 128): a pass adds the earlier passes' dym; only the last takes the column sums.
// NB: 16-label blocks the two backward products walk (7 when the pass has 97..112 labels -- C = 103 --, else 8; blocks
// past the pass's labels hold zeros in Pt / W_out's LDS copy, so walking them is only wasted work).  A compile-time
// count keeps the product loops free of branches: with `if (ib < CB)` inside them every MFMA sat behind its own
// ds_read + lgkmcnt(0).
template <bool MULTI, int NB, bool DROP, int NRB = 2>
__global__ __launch_bounds__(1024) void k_head_fused_rs(int n, int S, int C, const float* __restrict__ X,
                                                        const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ Wout, const float* __restrict__ bout,
                                                        const float* __restrict__ target, float keep_scale, uint32_t thresh,
                                                        const unsigned long long* __restrict__ rng_state, float inv_count,
                                                        float* __restrict__ probs, float* __restrict__ loss_part,
                                                        float* __restrict__ dym, float* __restrict__ part,
                                                        int c0, int Cp, int CPT, int first_, int last_, HeadStatAcc sa) {
  constexpr int D = 128, TR = 16 * NRB, EPL = 2, RPW = TR / 8, KQ = D / 4;   // NRB: 16-row MFMA blocks per tile (2 = HEADB_TILE)
  constexpr int CP = 128, CBMAX = 8;
  constexpr int LDP = CP + 18, LDY = D + 16, LDW = D + 16;
  __shared__ __attribute__((aligned(16))) float Wl[NB * 16 * LDW];   // W_out rows c0 .. c0 + 16 NB (zeros past Cp)
  __shared__ __attribute__((aligned(16))) float Pt[2][TR * LDP];
  __shared__ __attribute__((aligned(16))) float Yt[2][TR * LDY];
  __shared__ float lsum[8];
  __shared__ float wmax[16];   // accumulate mode: per-wave max |W_out| (the binary points of the backward sums)
  const bool first = MULTI ? first_ != 0 : true, last = MULTI ? last_ != 0 : true;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int own = wave & 7;            // P: label block;  Q: 16-column block of D
  const uint32_t key = DROP ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;
  const float invS = 1.f / (float)S;
  const bool S2 = S > 1;
  const int PS = head_part_stride(CPT, D);
  const int ntiles = (n + TR - 1) / TR;
  const int G = (int)gridDim.x;
  const int mt = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;   // tiles of this workgroup
  float* P = part + (size_t)blockIdx.x * PS;

  // ---- W_out -> LDS (all 16 waves, 16-byte pieces)
  float mwl = 0.f;   // max |W_out| over the rows this thread stages
  for (int idx = threadIdx.x; idx < NB * 16 * (D / 4); idx += 1024) {
    const int j = idx / (D / 4), c4 = idx % (D / 4);
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (j < Cp) v = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + c4 * 4];
    *(f32x4*)&Wl[j * LDW + c4 * 4] = v;
    mwl = fmaxf(fmaxf(mwl, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (sa.acc && last) {   // max |W_out| over ALL labels (the earlier label passes' share of dym is in the sums too): one pass
    float mw;             // (C <= 128) has just staged all of them; several passes read the whole matrix once more
    if (MULTI) mw = head_wout_absmax(Wout, C * (D / 4), 1024);
    else {
      mw = mwl;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, WAVE));
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mw;
  }
  // accumulate mode: batch mean / invstd decoded from the integer totals once per workgroup into Pt[1] (first written in S3 of
  // period 1, read by both roles right behind the "Wl complete" barrier)
  float* const stash = Pt[1];
  static_assert(TR * LDP >= 6 * D, "the stash fits a P tile");
#ifndef HRSX_NOSTAGE   // (decomposition build)
  if (sa.acc) head_stat_stage<D>(sa, n, S, stash, 1024);
#endif
  // The two roles run SEPARATE loops (the register allocator then sees each role's state on its own path) that execute
  // the same number of workgroup barriers: one before and one after the loop, three per period.
  // In every role phase the lane-derived indices are re-derived from an opaque copy of the lane id: addresses kept live
  // across the whole tile loop were what the allocator spilled, and every reload of one is a vmcnt(0) wait.
#define OPAQUE_LANE(r_, q_, l_)          \
  int l_ = lane;                         \
  asm volatile("" : "+v"(l_));           \
  const int r_ = l_ & 15, q_ = l_ >> 4

  if (wave < 8) {
    // =============================================================== P: pred team
    float xv[RPW][2][EPL], tgv[NRB][4], mu[2][EPL], is[2][EPL], gw[EPL], gb[EPL];
    float dbo = 0.f, lacc = 0.f;
    const float bj = own * 16 + (lane & 15) < Cp ? bout[c0 + own * 16 + (lane & 15)] : 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane * EPL + e;
      gw[e] = bn_w[c];
      gb[e] = bn_b[c];
    }
    // this wave's X rows of a tile are requested one tile ahead (at the top of phase B of the tile before: in flight
    // during its pred product and epilogue); the targets of this lane's logits at the end of that phase B
    auto load_rows = [&](int tile) {
      const int node0 = tile * TR;
#pragma unroll
      for (int t = 0; t < RPW; ++t) {
        const int i = node0 + own + t * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (i < n && s < S) ld_row<EPL>(xv[t][s], &X[(unsigned)((s * n + i) * D + lane * EPL)]);
          else zero_row<EPL>(xv[t][s]);
        }
      }
    };
    auto load_targets = [&](int tile) {
      const int node0 = tile * TR;
      const int j = own * 16 + (lane & 15), q = lane >> 4;
#pragma unroll
      for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = node0 + mb * 16 + q * 4 + e;
          tgv[mb][e] = (i < n && j < Cp) ? target[(unsigned)(i * C + c0 + j)] : 0.f;
        }
    };
    if (mt > 0) {
      load_rows(blockIdx.x);
      load_targets(blockIdx.x);
    }
    __syncthreads();   // Wl complete (and the statistics stash)
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
      for (int s = 0; s < 2; ++s) head_stat<D>(sa, stash, mean, invstd, s < S ? s : 0, lane * EPL + e, mu[s][e], is[s][e]);
    for (int k = 0; k <= mt; ++k) {
      const int tile = (int)blockIdx.x + k * G;
      const int node0 = tile * TR;
      // ---- S1: ym rows of tile k -> Yt[k & 1]                                  (vector work; Q: dym product)
      RS_STAMP(0);
      if (k < mt) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        float* __restrict__ Yb = Yt[k & 1];
        // (branch-free per element: with a branch per bound / strand / dropout test every element was its own serial
        // chain -- 2.5 us for 16 elements a lane; rows past n were loaded as zeros and are zeroed again below)
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
          const int rr = own + t * 8;
          const int i = node0 + rr;
          f32x2 ym;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            float y[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              y[s] = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e] * gw[e] + gb[e];
              if (DROP) y[s] = dropout_keep(key, (uint32_t)((s * n + i) * D + lane_ * EPL + e), thresh) ? y[s] * keep_scale : 0.f;
            }
            const float a = y[0] + (S2 ? y[1] : 0.f);
            ym[e] = i < n ? a * invS : 0.f;
          }
          *(f32x2*)&Yb[rr * LDY + lane_ * EPL] = ym;
        }
      }
      RS_STAMP(1);
      __syncthreads();
      // ---- S2: pred = ym W_out^T (M = 32 rows, K = D permuted, N = this wave's label block)   (matrix; Q: its epilogue)
      RS_STAMP(2);
      f32x4 acc[NRB];
#pragma unroll
      for (int mb = 0; mb < NRB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (k < mt) {
        if (k + 1 < mt) load_rows(tile + G);   // the rows were consumed in S1
        if (own < NB) {
          OPAQUE_LANE(r, q, lq);
          __builtin_amdgcn_s_setprio(HEAD_RS_PRIO);   // matrix sub-phase: issue ahead of the other team's vector work
          const float* __restrict__ Ya = Yt[k & 1] + r * LDY + 4 * q;
          const float* __restrict__ Wa = Wl + (own * 16 + r) * LDW + 4 * q;
#pragma unroll
          for (int t = 0; t < KQ / 4; ++t) {
            f32x4 a[NRB];
#pragma unroll
            for (int mb = 0; mb < NRB; ++mb) a[mb] = *(const f32x4*)&Ya[mb * 16 * LDY + 16 * t];
            const f32x4 b = *(const f32x4*)&Wa[16 * t];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int mb = 0; mb < NRB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][u], b[u], acc[mb], 0, 0, 0);
          }
          __builtin_amdgcn_s_setprio(0);
        }
      }
      RS_STAMP(3);
      __syncthreads();
      // ---- S3: sigmoid / BCE / probs; d loss / d pred -> Pt[k & 1] (zero outside the valid region)   (vector; Q: dW_out product)
      RS_STAMP(4);
      if (k < mt) {
        float* __restrict__ Pb = Pt[k & 1];
        OPAQUE_LANE(r, q, lq);
        const int j = own * 16 + r;
#pragma unroll
        for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // branch-free but for the predicated store (see S1)
            const int row = mb * 16 + q * 4 + e;
            const int i = node0 + row;
            const bool ok = i < n && j < Cp;
            const float pred = acc[mb][e] + bj;
            const float en = __expf(-fabsf(pred));
            const float inv = __builtin_amdgcn_rcpf(1.f + en);   // 1 ulp; the IEEE division is ten instructions
            const float p = pred >= 0.f ? inv : en * inv;
            const float l = fmaxf(pred, 0.f) - pred * tgv[mb][e] + __logf(1.f + en);
            lacc += ok ? l : 0.f;
            if (ok) {   // written once, read by nobody on the device before the epoch's end (HEAD_NT_PROBS: keep it out of the L2s)
              if (HEAD_NT_PROBS) __builtin_nontemporal_store(p, &probs[(unsigned)(i * C + c0 + j)]);
              else probs[(unsigned)(i * C + c0 + j)] = p;
            }
            const float dp = ok ? (p - tgv[mb][e]) * inv_count : 0.f;
            dbo += dp;
            Pb[row * LDP + j] = dp;
          }
        if (k + 1 < mt) load_targets(tile + G);
      }
      RS_STAMP(5);
      __syncthreads();
    }
    // ---- db_out share and the loss share of this workgroup
    dbo += __shfl_xor(dbo, 16);
    dbo += __shfl_xor(dbo, 32);
    if ((lane >> 4) == 0 && c0 + own * 16 + (lane & 15) < CPT) P[CPT * D + c0 + own * 16 + (lane & 15)] = dbo;
    lacc = wave_sum(lacc);
    if (lane == 0) lsum[own] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += lsum[w];
      const float tl = first ? t : loss_part[blockIdx.x] + t;   // the label passes' shares add up
      loss_part[blockIdx.x] = tl;
#ifndef HRSX_NOTICKET   // (decomposition build)
      if (sa.acc && last) head_loss_ticket(sa, S, D, tl, inv_count);   // (the Q team's integer adds: before the barrier above)
#endif
    }
  } else {
    // =============================================================== Q: grad team, one tile behind
    f32x4 accW[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) accW[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float sdy[2] = {0.f, 0.f}, sdyx[2] = {0.f, 0.f}, mu[2], is[2];
    __syncthreads();   // Wl complete (and the statistics stash)
    if (sa.acc && blockIdx.x == 0 && first && wave == 8) head_stat_bookkeeping<D>(sa, stash, S, lane);   // (see HeadStatAcc)
#pragma unroll
    for (int s = 0; s < 2; ++s) head_stat<D>(sa, stash, mean, invstd, s < S ? s : 0, own * 16 + (lane & 15), mu[s], is[s]);
    // X of this lane's (row, column) elements of a tile, for the BatchNorm sums of its epilogue: requested a whole matrix
    // product before they are used -- the rows have left the L2 since P read them (a 30 MB table), and their latency was
    // all of the epilogue's time when they were requested there
    float xq[2][NRB][4];
    auto load_xq = [&](int tile) {
      OPAQUE_LANE(r, q, lq);
      const int node0 = tile * TR, c = own * 16 + r;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = node0 + mb * 16 + q * 4 + e;
            xq[s][mb][e] = (last && i < n && s < S) ? X[(unsigned)((s * n + i) * D + c)] : 0.f;
          }
    };
    for (int k = 0; k <= mt; ++k) {
      const int node0 = ((int)blockIdx.x + (k - 1) * G) * TR;   // tile k-1
      f32x4 accY[NRB];
#pragma unroll
      for (int mb = 0; mb < NRB; ++mb) accY[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // ---- S1: dym tile = Pt W_out of tile k-1 (M = TR rows, K = labels, N = this wave's 16 columns)   (matrix; P: ym rows)
      RS_STAMP(0);
      if (k >= 1) {
        const float* __restrict__ Pb = Pt[(k - 1) & 1];
        OPAQUE_LANE(r, q, lq);
        if (MULTI && !first) {   // the earlier label passes' share of dym seeds the accumulators
          const int c = own * 16 + r;
#pragma unroll
          for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = node0 + mb * 16 + q * 4 + e;
              accY[mb][e] = i < n ? dym[(unsigned)(i * D + c)] : 0.f;
            }
        }
        // label groups of 16 (4 k-steps: 8 A reads from Pt, 4 B reads from W_out), the next group's reads under this group's MFMAs
        __builtin_amdgcn_s_setprio(HEAD_RS_PRIO);
        const float* __restrict__ Pa = Pb + r * LDP + q;
        const float* __restrict__ Wa = Wl + q * LDW + own * 16 + r;
        float pa[2][4 * NRB], wb[2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          wb[0][u] = Wa[4 * u * LDW];
#pragma unroll
          for (int mb = 0; mb < NRB; ++mb) pa[0][u * NRB + mb] = Pa[mb * 16 * LDP + 4 * u];
        }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
          if (g + 1 < NB) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              wb[(g + 1) & 1][u] = Wa[4 * (4 * (g + 1) + u) * LDW];
#pragma unroll
              for (int mb = 0; mb < NRB; ++mb) pa[(g + 1) & 1][u * NRB + mb] = Pa[mb * 16 * LDP + 4 * (4 * (g + 1) + u)];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int mb = 0; mb < NRB; ++mb) accY[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[g & 1][u * NRB + mb], wb[g & 1][u], accY[mb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      RS_STAMP(1);
      __syncthreads();
      // ---- S2: dym out, BatchNorm-backward column sums of tile k-1              (vector; P: pred product)
      RS_STAMP(2);
      if (k >= 1) {
        OPAQUE_LANE(r, q, lq);
        const int c = own * 16 + r;
#pragma unroll
        for (int mb = 0; mb < NRB; ++mb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // branch-free but for the predicated store: rows past n have d loss / d pred = 0,
            const int i = node0 + mb * 16 + q * 4 + e;   // hence dym = 0 and contribute nothing to the sums
            const float g = accY[mb][e];
            if (i < n) dym[(unsigned)(i * D + c)] = g;    // 64-byte row segments; the other column blocks come from the neighbouring waves
            if (last) {
#pragma unroll
              for (int s = 0; s < 2; ++s) {
                float dy = g * invS;
                if (DROP) dy = dropout_keep(key, (uint32_t)((s * n + i) * D + c), thresh) ? dy * keep_scale : 0.f;
                if (s == 1) dy = S2 ? dy : 0.f;
                const float xh = (fmaxf(xq[s][mb][e], 0.f) - mu[s]) * is[s];
                sdy[s] += dy;
                sdyx[s] += dy * xh;
              }
            }
          }
      }
      RS_STAMP(3);
      __syncthreads();
      // ---- S3: dW_out += Pt^T Yt of tile k-1 (K = TR rows); operands of step kk + 1 are read under step kk   (matrix; P: epilogue)
      RS_STAMP(4);
      if (k < mt) load_xq((int)blockIdx.x + k * G);   // tile k's values, used in S2 of the next period
      if (k >= 1) {
        OPAQUE_LANE(r, q, lq);
        __builtin_amdgcn_s_setprio(HEAD_RS_PRIO);
        const float* __restrict__ Pa = Pt[(k - 1) & 1] + q * LDP + r;
        const float* __restrict__ Ya = Yt[(k - 1) & 1] + q * LDY + own * 16 + r;
        float a0[NB], a1[NB], b0, b1;
        b0 = Ya[0];
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) a0[ib] = Pa[ib * 16];
#pragma unroll
        for (int kk = 0; kk < TR / 4; kk += 2) {
          b1 = Ya[(kk + 1) * 4 * LDY];
#pragma unroll
          for (int ib = 0; ib < NB; ++ib) a1[ib] = Pa[(kk + 1) * 4 * LDP + ib * 16];
          __builtin_amdgcn_sched_barrier(0);   // one step of operand reads ahead, no more (registers)
#pragma unroll
          for (int ib = 0; ib < NB; ++ib) accW[ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[ib], b0, accW[ib], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kk + 2 < TR / 4) {
            b0 = Ya[(kk + 2) * 4 * LDY];
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) a0[ib] = Pa[(kk + 2) * 4 * LDP + ib * 16];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ib = 0; ib < NB; ++ib) accW[ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ib], b1, accW[ib], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      RS_STAMP(5);
      __syncthreads();
    }
    // ---- dW_out share; BatchNorm-backward column sums (float64 from the cross-lane merge on)
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int ib = 0; ib < CBMAX; ++ib)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = ib < NB ? accW[ib < NB ? ib : 0][e] : 0.f;
        if (c0 + ib * 16 < CPT) P[(size_t)(c0 + ib * 16 + q * 4 + e) * D + own * 16 + r] = v;
      }
    if (last) {   // column sums over this lane's rows -> over the four row groups of the wave
      double* out = (double*)(P + CPT * D + CPT);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        double a = (double)sdy[s], b = (double)sdyx[s];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (q == 0) {
          out[s * D + own * 16 + r] = a;
          out[(2 + s) * D + own * 16 + r] = b;
          if (sa.acc && s < S) {   // accumulate mode: the same sums as integer totals for the row-local backward's prologue
            int fa, fb;
            head_bacc_points(wmax, 16, keep_scale, n, fa, fb);
            unsigned long long* bb = bacc_base(sa.acc, S, D);
#ifndef HRSX_NOBACC   // (decomposition build: profiles/r06_stat_acc_experiment.txt item 5)
            bacc_add(bb, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), s, own * 16 + r, a, b, fa, fb);
#endif
            if (blockIdx.x == 0 && s == 0 && own == 0 && r == 0)
              bb[(size_t)STAT_ACC_SLOTS * S * D * 2 + BACC_EXP] = ((unsigned long long)(unsigned)(fb + 1024) << 32) | (unsigned long long)(unsigned)(fa + 1024);
          }
        }
      }
    }
    __syncthreads();   // (the P team's loss merge)
  }
#undef OPAQUE_LANE
}

// ------------------------------------------------------------------------------------------
// k_head_fused_sp (D = 128, round 6): k_head_fused_rs at 16-row tiles with its three products as SPLIT PRODUCTS
// (cgcn_common.hpp: six bf16 MFMA partial products of an exact 3-way split of both fp32 operands, fp32 accumulators).  Same
// teams, same three sub-phases per period, same barriers, same partial records, label passes and statistics paths; what
// changes is what the tiles in LDS hold and which instruction reads them:
//   Yt, Pt, W_out: three bf16 LEVEL tiles each (16 rows x 128 columns x 2 B = 4 KB per level, swizzled image sp_sigma), written
//                  as levels by whoever produces them (the P team's S1 / S3; the launch's prologue);
//   pred = ym W_out^T   P wave w, label block w: v_mfma_f32_16x16x32_bf16; ym rows by ds_read_b128; the wave's 16 W_out rows
//                       are RESIDENT as levels (48 registers; a P wave's persistent state is 24);
//   dym  = Pt W_out     Q wave w, feature columns [16 w, +16): 16x16x32; Pt rows by ds_read_b128, W_out K-major by
//                       ds_read_b64_tr_b16 (the hardware transpose); label blocks past the pass hold zeros;
//   dW_out += Pt^T Yt   Q wave w: labels [64 (w & 1), +64) x columns [32 (w >> 1), +32) as two 32 x 32 accumulators,
//                       v_mfma_f32_32x32x16_bf16 with K = the tile's 16 rows, both operands transposed reads.
// LDS: W_out 96 KB + 2 x (Yt + Pt) 48 KB.  fp32 MFMA held the SIMD's vector ALUs, so the teams' "matrix" and "vector"
// sub-phases added up; the bf16 forms take a sixteenth of the cycles per flop and leave half their issue slots free.
// ------------------------------------------------------------------------------------------
// Barriers per period: TWO.  k_head_fused_rs has three, so that in every sub-phase one team is on the matrix pipe and the other
// on the vector pipe (fp32 MFMA holds the vector ALUs); with the bf16 forms that alternation buys nothing, and the data only asks
// for: P's ym rows complete before P's pred product (B1), and the period's tiles complete before the next period reuses their
// buffers (B2).  P: S1 | B1 | S2 S3 | B2;  Q (one tile behind): S1 S2 | B1 | S3 | B2.  (HEADSP_BARRIERS = 3: A/B builds.)
#ifndef HEADSP_BARRIERS
#define HEADSP_BARRIERS 2
#endif
template <bool MULTI, int NB, bool DROP>
__global__ __launch_bounds__(1024) void k_head_fused_sp(int n, int S, int C, const float* __restrict__ X,
                                                        const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ Wout, const float* __restrict__ bout,
                                                        const float* __restrict__ target, float keep_scale, uint32_t thresh,
                                                        const unsigned long long* __restrict__ rng_state, float inv_count,
                                                        float* __restrict__ probs, float* __restrict__ loss_part,
                                                        float* __restrict__ dym, float* __restrict__ part,
                                                        int c0, int Cp, int CPT, int first_, int last_, HeadStatAcc sa) {
  constexpr int D = 128, TR = 16, EPL = 2, RPW = TR / 8;
  constexpr int LVT = TR * 256;          // bytes of one level tile (16 rows)
  constexpr int WLV = 128 * 256;         // bytes of one level of the W_out image (128 label rows, zeros past the pass)
  __shared__ __attribute__((aligned(16))) unsigned char Wl[3 * WLV];
  __shared__ __attribute__((aligned(16))) unsigned char Ytb[2][3 * LVT];
  __shared__ __attribute__((aligned(16))) unsigned char Ptb[2][3 * LVT];
  __shared__ float lsum[8];
  __shared__ float wmax[16];   // accumulate mode: per-wave max |W_out| (the binary points of the backward sums)
  const bool first = MULTI ? first_ != 0 : true, last = MULTI ? last_ != 0 : true;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int own = wave & 7;            // P: label block;  Q: 16-column block of D
  const uint32_t key = DROP ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;
  const float invS = 1.f / (float)S;
  const bool S2 = S > 1;
  const int PS = head_part_stride(CPT, D);
  const int ntiles = (n + TR - 1) / TR;
  const int G = (int)gridDim.x;
  const int mt = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;   // tiles of this workgroup
  float* P = part + (size_t)blockIdx.x * PS;

  // ---- W_out -> LDS as three level images (all 16 waves, 4 columns per thread and step; rows past the pass: zeros)
  float mwl = 0.f;   // max |W_out| over the rows this thread stages
  for (int idx = threadIdx.x; idx < 128 * (D / 4); idx += 1024) {
    const int j = idx / (D / 4), c4 = idx % (D / 4);
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (j < Cp) v = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + c4 * 4];
    u32x2 h2, m2, l2;
    sp_split4(v, h2, m2, l2);
    unsigned char* w = Wl + j * 256 + (((c4 >> 1) ^ sp_sigma(j & 15)) << 4) + ((c4 & 1) << 3);
    *(u32x2*)w = h2;
    *(u32x2*)(w + WLV) = m2;
    *(u32x2*)(w + 2 * WLV) = l2;
    mwl = fmaxf(fmaxf(mwl, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (sa.acc && last) {   // max |W_out| over ALL labels (see k_head_fused_rs)
    float mw;
    if (MULTI) mw = head_wout_absmax(Wout, C * (D / 4), 1024);
    else {
      mw = mwl;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, WAVE));
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mw;
  }
  // accumulate mode: batch mean / invstd decoded once per workgroup into the second P tile (first written in S3 of period 1)
  float* const stash = (float*)Ptb[1];
  static_assert(3 * LVT >= 6 * D * (int)sizeof(float), "the stash fits a P tile");
#ifndef HRSX_NOSTAGE   // (decomposition build)
  if (sa.acc) head_stat_stage<D>(sa, n, S, stash, 1024);
#endif
#define OPAQUE_LANE(r_, q_, l_)          \
  int l_ = lane;                         \
  asm volatile("" : "+v"(l_));           \
  const int r_ = l_ & 15, q_ = l_ >> 4

  if (wave < 8) {
    // =============================================================== P: pred team
    float xv[RPW][2][EPL], tgv[4], mu[2][EPL], is[2][EPL], gw[EPL], gb[EPL];
    float dbo = 0.f, lacc = 0.f;
    const float bj = own * 16 + (lane & 15) < Cp ? bout[c0 + own * 16 + (lane & 15)] : 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      const int c = lane * EPL + e;
      gw[e] = bn_w[c];
      gb[e] = bn_b[c];
    }
    // this wave's 16 rows of W_out as B operands of the pred product: B(k, j) = W_out[c0 + 16 own + j][k], K-step s: k = 32 s + 8 q ..
    bf16x8 wh[4], wm[4], wl[4];
    {
      const int r = lane & 15, q = lane >> 4;
      const int j = own * 16 + r;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (j < Cp) {
          v0 = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + 32 * s4 + 8 * q];
          v1 = *(const f32x4*)&Wout[(size_t)(c0 + j) * D + 32 * s4 + 8 * q + 4];
        }
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        sp_split8(v, wh[s4], wm[s4], wl[s4]);
      }
    }
    auto load_rows = [&](int tile) {
      const int node0 = tile * TR;
#pragma unroll
      for (int t = 0; t < RPW; ++t) {
        const int i = node0 + own + t * 8;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (i < n && s < S) ld_row<EPL>(xv[t][s], &X[(unsigned)((s * n + i) * D + lane * EPL)]);
          else zero_row<EPL>(xv[t][s]);
        }
      }
    };
    auto load_targets = [&](int tile) {
      const int node0 = tile * TR;
      const int j = own * 16 + (lane & 15), q = lane >> 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = node0 + q * 4 + e;
        tgv[e] = (i < n && j < Cp) ? target[(unsigned)(i * C + c0 + j)] : 0.f;
      }
    };
    if (mt > 0) {
      load_rows(blockIdx.x);
      load_targets(blockIdx.x);
    }
    __syncthreads();   // W_out image complete (and the statistics stash)
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
      for (int s = 0; s < 2; ++s) head_stat<D>(sa, stash, mean, invstd, s < S ? s : 0, lane * EPL + e, mu[s][e], is[s][e]);
    for (int k = 0; k <= mt; ++k) {
      const int tile = (int)blockIdx.x + k * G;
      const int node0 = tile * TR;
      // ---- S1: ym rows of tile k -> the three levels of Yt[k & 1]                (vector work; Q: dym product)
      if (k < mt) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        unsigned char* __restrict__ Yb = Ytb[k & 1];
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
          const int rr = own + t * 8;
          const int i = node0 + rr;
          float ym[EPL];
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            float y[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              y[s] = (fmaxf(xv[t][s][e], 0.f) - mu[s][e]) * is[s][e] * gw[e] + gb[e];
              if (DROP) y[s] = dropout_keep(key, (uint32_t)((s * n + i) * D + lane_ * EPL + e), thresh) ? y[s] * keep_scale : 0.f;
            }
            const float a = y[0] + (S2 ? y[1] : 0.f);
            ym[e] = i < n ? a * invS : 0.f;
          }
          // columns 2 lane, 2 lane + 1 of row rr: one bf16 pair per level
          uint32_t yh, ymid, yl;
          sp_split2(ym[0], ym[1], yh, ymid, yl);
          unsigned char* w = Yb + rr * 256 + (((lane_ >> 2) ^ sp_sigma(rr)) << 4) + ((lane_ & 3) << 2);
          *(uint32_t*)w = yh;
          *(uint32_t*)(w + LVT) = ymid;
          *(uint32_t*)(w + 2 * LVT) = yl;
        }
      }
      __syncthreads();
      // ---- S2: pred = ym W_out^T (M = 16 rows, K = D, N = this wave's label block)   (matrix; Q: its epilogue)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (k < mt) {
        if (k + 1 < mt) load_rows(tile + G);   // the rows were consumed in S1
        if (own < NB) {
          OPAQUE_LANE(r, q, lq);
          const unsigned char* __restrict__ Ya = Ytb[k & 1] + r * 256 + ((q ^ (sp_sigma(r) & 3)) << 4);
          const int hi = sp_sigma(r) >> 2;
          SpAcc sacc;
          sacc.zero();
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const int o = (s4 ^ hi) << 6;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ya + o));
            const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ya + LVT + o));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(Ya + 2 * LVT + o));
            sacc.step(ah, am, al, wh[s4], wm[s4], wl[s4]);
          }
          acc = sacc.sum();
        }
      }
#if HEADSP_BARRIERS == 3
      __syncthreads();
#endif
      // ---- S3: sigmoid / BCE / probs; d loss / d pred -> the levels of Pt[k & 1] (zero outside the valid region)   (vector; Q: dW_out product)
      if (k < mt) {
        unsigned char* __restrict__ Pb = Ptb[k & 1];
        OPAQUE_LANE(r, q, lq);
        const int j = own * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = q * 4 + e;
          const int i = node0 + row;
          const bool ok = i < n && j < Cp;
          const float pred = acc[e] + bj;
          const float en = __expf(-fabsf(pred));
          const float inv = __builtin_amdgcn_rcpf(1.f + en);   // 1 ulp; the IEEE division is ten instructions
          const float p = pred >= 0.f ? inv : en * inv;
          const float l = fmaxf(pred, 0.f) - pred * tgv[e] + __logf(1.f + en);
          lacc += ok ? l : 0.f;
          if (ok) {
            if (HEAD_NT_PROBS) __builtin_nontemporal_store(p, &probs[(unsigned)(i * C + c0 + j)]);
            else probs[(unsigned)(i * C + c0 + j)] = p;
          }
          const float dp = ok ? (p - tgv[e]) * inv_count : 0.f;
          dbo += dp;
          uint16_t dh, dm, dl;
          sp_split1(dp, dh, dm, dl);
          unsigned char* w = Pb + row * 256 + (((j >> 3) ^ sp_sigma(row)) << 4) + ((j & 7) << 1);
          *(uint16_t*)w = dh;
          *(uint16_t*)(w + LVT) = dm;
          *(uint16_t*)(w + 2 * LVT) = dl;
        }
        if (k + 1 < mt) load_targets(tile + G);
      }
      __syncthreads();
    }
    // ---- db_out share and the loss share of this workgroup
    dbo += __shfl_xor(dbo, 16);
    dbo += __shfl_xor(dbo, 32);
    if ((lane >> 4) == 0 && c0 + own * 16 + (lane & 15) < CPT) P[CPT * D + c0 + own * 16 + (lane & 15)] = dbo;
    lacc = wave_sum(lacc);
    if (lane == 0) lsum[own] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += lsum[w];
      const float tl = first ? t : loss_part[blockIdx.x] + t;   // the label passes' shares add up
      loss_part[blockIdx.x] = tl;
#ifndef HRSX_NOTICKET   // (decomposition build)
      if (sa.acc && last) head_loss_ticket(sa, S, D, tl, inv_count);   // (the Q team's integer adds: before the barrier above)
#endif
    }
  } else {
    // =============================================================== Q: grad team, one tile behind
    f32x16 accW[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int e = 0; e < 16; ++e) accW[bb][e] = 0.f;
    float sdy[2] = {0.f, 0.f}, sdyx[2] = {0.f, 0.f}, mu[2], is[2];
    __syncthreads();   // W_out image complete (and the statistics stash)
    if (sa.acc && blockIdx.x == 0 && first && wave == 8) head_stat_bookkeeping<D>(sa, stash, S, lane);   // (see HeadStatAcc)
#pragma unroll
    for (int s = 0; s < 2; ++s) head_stat<D>(sa, stash, mean, invstd, s < S ? s : 0, own * 16 + (lane & 15), mu[s], is[s]);
    float xq[2][4];
    auto load_xq = [&](int tile) {
      OPAQUE_LANE(r, q, lq);
      const int node0 = tile * TR, c = own * 16 + r;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = node0 + q * 4 + e;
          xq[s][e] = (last && i < n && s < S) ? X[(unsigned)((s * n + i) * D + c)] : 0.f;
        }
    };
    // transposed-read lane constants (see k_bwd_rowlocal_ring): lane = 32 h + 16 c16 + i (32x32x16 operands) or 16 q + i (16x16x32)
    for (int k = 0; k <= mt; ++k) {
      const int node0 = ((int)blockIdx.x + (k - 1) * G) * TR;   // tile k-1
      f32x4 accY = {0.f, 0.f, 0.f, 0.f};
      // ---- S1: dym tile = Pt W_out of tile k-1 (M = 16 rows, K = 128 labels, N = this wave's 16 columns)   (matrix; P: ym rows)
      if (k >= 1) {
        const unsigned char* __restrict__ Pb = Ptb[(k - 1) & 1];
        OPAQUE_LANE(r, q, lq);
        // A: Pt rows, chunk 4 s + q of row r;  B: W_out K-major, the 16-lane group q fetches label rows 32 s + 8 q + 4 rd + (i >> 2),
        // piece (i & 3) of columns [16 own, +16)
        const unsigned char* __restrict__ Pa = Pb + r * 256 + ((q ^ (sp_sigma(r) & 3)) << 4);
        const int hi = sp_sigma(r) >> 2;
        const int qp = r >> 2, pp = r & 3;
        SpAcc2 sacc;
        sacc.zero();
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int o = (s4 ^ hi) << 6;
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const u32x4*)(Pa + o));
          const bf16x8 am = __builtin_bit_cast(bf16x8, *(const u32x4*)(Pa + LVT + o));
          const bf16x8 al = __builtin_bit_cast(bf16x8, *(const u32x4*)(Pa + 2 * LVT + o));
          int wo[2];
#pragma unroll
          for (int rd = 0; rd < 2; ++rd) {
            const int lrow = 32 * s4 + 8 * q + 4 * rd + qp;
            wo[rd] = lrow * 256 + (((2 * own + (pp >> 1)) ^ sp_sigma(lrow & 15)) << 4) + ((pp & 1) << 3);
          }
          const bf16x8 bh = tr_pair(Wl + wo[0], Wl + wo[1]);
          const bf16x8 bm = tr_pair(Wl + WLV + wo[0], Wl + WLV + wo[1]);
          const bf16x8 bl = tr_pair(Wl + 2 * WLV + wo[0], Wl + 2 * WLV + wo[1]);
          sacc.step(ah, am, al, bh, bm, bl);
        }
        accY = sacc.sum();
        if (MULTI && !first) {   // the earlier label passes' share of dym
          const int c = own * 16 + r;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = node0 + q * 4 + e;
            accY[e] += i < n ? dym[(unsigned)(i * D + c)] : 0.f;
          }
        }
      }
#if HEADSP_BARRIERS == 3
      __syncthreads();
#endif
      // ---- S2: dym out, BatchNorm-backward column sums of tile k-1              (vector; P: pred product)
      if (k >= 1) {
        OPAQUE_LANE(r, q, lq);
        const int c = own * 16 + r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = node0 + q * 4 + e;
          const float g = accY[e];
          if (i < n) dym[(unsigned)(i * D + c)] = g;
          if (last) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              float dy = g * invS;
              if (DROP) dy = dropout_keep(key, (uint32_t)((s * n + i) * D + c), thresh) ? dy * keep_scale : 0.f;
              if (s == 1) dy = S2 ? dy : 0.f;
              const float xh = (fmaxf(xq[s][e], 0.f) - mu[s]) * is[s];
              sdy[s] += dy;
              sdyx[s] += dy * xh;
            }
          }
        }
      }
      __syncthreads();
      // ---- S3: dW_out += Pt^T Yt of tile k-1 (K = the tile's 16 rows), two 32 x 32 blocks per wave   (matrix; P: epilogue)
      if (k < mt) load_xq((int)blockIdx.x + k * G);   // tile k's values, used in S2 of the next period
      if (k >= 1) {
        int lq = lane;
        asm volatile("" : "+v"(lq));
        const int hh = lq >> 5, c16 = (lq >> 4) & 1, qp = (lq & 15) >> 2, pp = lq & 3;
        const int cbk = own >> 1, lb0 = 2 * (own & 1);
        const unsigned char* __restrict__ Pb = Ptb[(k - 1) & 1];
        const unsigned char* __restrict__ Yb = Ytb[(k - 1) & 1];
        int ty[2], tp[2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
          const int row = 8 * hh + 4 * rd + qp, sg = sp_sigma(row);
          ty[rd] = row * 256 + (((4 * cbk + 2 * c16 + (pp >> 1)) ^ sg) << 4) + ((pp & 1) << 3);
          tp[rd] = row * 256 + (((4 * lb0 + 2 * c16 + (pp >> 1)) ^ sg) << 4) + ((pp & 1) << 3);   // label block bb: ^ (bb << 6)
        }
        bf16x8 yf[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) yf[v] = tr_pair(Yb + v * LVT + ty[0], Yb + v * LVT + ty[1]);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          bf16x8 pf[3];
#pragma unroll
          for (int v = 0; v < 3; ++v) pf[v] = tr_pair(Pb + v * LVT + (tp[0] ^ (bb << 6)), Pb + v * LVT + (tp[1] ^ (bb << 6)));
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[2], yf[0], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[0], yf[2], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[1], yf[1], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[1], yf[0], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[0], yf[1], accW[bb], 0, 0, 0);
          accW[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[0], yf[0], accW[bb], 0, 0, 0);
        }
      }
      __syncthreads();
    }
    // ---- dW_out share: accW[bb][reg] = dW_out[c0 + 32 (lb0 + bb) + (reg & 3) + 8 (reg >> 2) + 4 hh][32 cbk + (lane & 31)]
    {
      const int hh = lane >> 5, cbk = own >> 1, lb0 = 2 * (own & 1);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int lab = 32 * (lb0 + bb) + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (c0 + (lab & ~15) < CPT) P[(size_t)(c0 + lab) * D + 32 * cbk + (lane & 31)] = accW[bb][e];
        }
    }
    const int r = lane & 15, q = lane >> 4;
    if (last) {   // column sums over this lane's rows -> over the four row groups of the wave
      double* out = (double*)(P + CPT * D + CPT);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        double a = (double)sdy[s], b = (double)sdyx[s];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (q == 0) {
          out[s * D + own * 16 + r] = a;
          out[(2 + s) * D + own * 16 + r] = b;
          if (sa.acc && s < S) {   // accumulate mode: the same sums as integer totals for the row-local backward's prologue
            int fa, fb;
            head_bacc_points(wmax, 16, keep_scale, n, fa, fb);
            unsigned long long* bb = bacc_base(sa.acc, S, D);
#ifndef HRSX_NOBACC
            bacc_add(bb, S, D, (int)blockIdx.x & (STAT_ACC_SLOTS - 1), s, own * 16 + r, a, b, fa, fb);
#endif
            if (blockIdx.x == 0 && s == 0 && own == 0 && r == 0)
              bb[(size_t)STAT_ACC_SLOTS * S * D * 2 + BACC_EXP] = ((unsigned long long)(unsigned)(fb + 1024) << 32) | (unsigned long long)(unsigned)(fa + 1024);
          }
        }
      }
    }
    __syncthreads();   // (the P team's loss merge)
  }
#undef OPAQUE_LANE
}

// second stage: workgroups [0, wslabs) sum the dW_out / db_out slabs wslab0 + b (head_finalize_slab), the rest the
// BatchNorm columns in float64 (head_stats_finalize)
__global__ __launch_bounds__(512) void k_head_bwd_finalize(int wslab0, int wslabs, int P, int n, int S, int D, int C, int CP,
                                                           const float* __restrict__ part, float* __restrict__ dWout,
                                                           float* __restrict__ dbout, float* __restrict__ dbn_w,
                                                           float* __restrict__ dbn_b, float* __restrict__ bnc,
                                                           int accumulate, const float* __restrict__ dloss) {
  if ((int)blockIdx.x < wslabs)
    head_finalize_slab<512>(wslab0 + blockIdx.x, P, D, C, CP, part, dWout, dbout, accumulate, dloss);
  else
    head_stats_finalize<512>((int)blockIdx.x - wslabs, P, n, S, D, CP, part, dbn_w, dbn_b, bnc, accumulate, dloss);
}

// Last launch of cgcn_head_train: workgroup 0 adds up the loss shares (fixed-order tree); the others sum the
// BatchNorm-backward columns of the partials (float64) into bnc = (mean dy, mean dy*xhat) per strand, for an upstream
// d loss of 1 (the consumer, k_bwd_rowlocal's head prologue, scales by the real one).  One launch instead of a
// loss-sum launch in the forward plus a finalize launch in the backward.
__global__ __launch_bounds__(512) void k_head_train_finish(int P, int n, int S, int D, int CP,
                                                           const float* __restrict__ part, float* __restrict__ bnc,
                                                           const float* __restrict__ loss_part, float inv_count,
                                                           float* __restrict__ loss) {
  if (blockIdx.x == 0) {
    __shared__ float sm[512];
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 512) s += loss_part[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = sm[0] * inv_count;
    return;
  }
  head_stats_finalize<512>((int)blockIdx.x - 1, P, n, S, D, CP, part, nullptr, nullptr, bnc, 0, nullptr);
}

template <int D>
__global__ __launch_bounds__(256) void k_head_bn_bwd_apply(int n, int S, const float* __restrict__ X,
                                                           const float* __restrict__ bn_w, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ dym,
                                                           const float* __restrict__ bnc, float keep_scale, uint32_t thresh,
                                                           const unsigned long long* __restrict__ rng_state,
                                                           float* __restrict__ dX) {
  const uint32_t key = thresh ? dropout_key(rng_state, HEAD_STREAM_ID) : 0u;
  const size_t total4 = (size_t)S * n * D / 4;
  const float invS = 1.f / (float)S;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total4; v += (size_t)gridDim.x * blockDim.x) {
    const size_t el0 = v * 4;
    const int c0 = (int)(el0 % D);
    const size_t row = el0 / D;       // s*n + i
    const int s = (int)(row / n);
    const int i = (int)(row % n);
    const f32x4 x = *(const f32x4*)&X[el0];
    const f32x4 g = *(const f32x4*)&dym[(size_t)i * D + c0];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + e;
      float dy = g[e] * invS;
      if (thresh) dy = dropout_keep(key, (uint32_t)(el0 + e), thresh) ? dy * keep_scale : 0.f;
      const float is = invstd[s * D + c];
      const float xh = (fmaxf(x[e], 0.f) - mean[s * D + c]) * is;
      const float dr = bn_w[c] * is * (dy - bnc[(s * 2 + 0) * D + c] - xh * bnc[(s * 2 + 1) * D + c]);
      o[e] = x[e] > 0.f ? dr : 0.f;
    }
    *(f32x4*)&dX[el0] = o;
  }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static int head_check(int n, int S, int d, int C) {
  if (n < 0 || C < 1) return CGCN_ERR_BAD_ARG;
  if (!(S == 1 || S == 2) || !(d == 128 || d == 256) || C > 256) return CGCN_ERR_UNSUPPORTED;
  if ((double)n * S * d * 4.0 >= 4294967296.0 || (double)n * C * 4.0 >= 4294967296.0) return CGCN_ERR_UNSUPPORTED;
  return CGCN_OK;
}
static inline int head_cp(int C) { return C <= 128 ? 128 : 256; }
static inline int head_stat_blocks(int n, int* rows_per_blk) {
  int rpb = (n + HEAD_STAT_BLOCKS - 1) / HEAD_STAT_BLOCKS;
  if (rpb < 8) rpb = 8;
  *rows_per_blk = rpb;
  int nb = (n + rpb - 1) / rpb;
  return nb < 1 ? 1 : nb;
}
// Workgroups (= partials) of k_head_fused / k_head_bwd: one 32-node tile each up to HEAD_MAX_PARTIALS (one per CU),
// persistent beyond.  Measured alternatives: twice as many workgroups for large chromosomes (the kernel needs 206
// VGPRs, so they do not co-reside: 81 vs 75 us at chr1 size); 16-node tiles for small ones (more, shorter chains but
// twice the partials: chr21 step 0.234 vs 0.221 ms).  k_head_fused_rs keeps this workgroup count at either tile height
// (it picks 16- or 32-row tiles per launch, cgcn_head_train).
static inline int head_bwd_partials(int n) {
  int t = (n + HEADB_TILE - 1) / HEADB_TILE;
  if (t > HEAD_MAX_PARTIALS) t = HEAD_MAX_PARTIALS;
  return t < 1 ? 1 : t;
}
// workspace regions (floats), in this order
static inline size_t ws_stats(int S, int d) { return (size_t)HEAD_STAT_BLOCKS * S * d * 2; }
static inline size_t ws_loss(int n) { return (size_t)((n + HEAD_TILE - 1) / HEAD_TILE + 4); }
static inline size_t ws_dym(int n, int d) { return (size_t)n * d + 4; }
static inline size_t ws_bnc(int d) { return (size_t)2 * 2 * d; }
static inline size_t ws_part(int n, int d, int C) { return (size_t)head_bwd_partials(n) * (size_t)head_part_stride(head_cp(C), d); }
static inline size_t align4(size_t x) { return (x + 3) & ~(size_t)3; }

extern "C" {

#ifdef HF_TIMING
int cgcn_debug_hf_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hf_stamps), sizeof(unsigned long long) * 8 * 16) == hipSuccess ? 0 : -1;
}
#endif


int cgcn_head_workspace_layout(int n, int S, int d, int C, size_t* dym_offset, size_t* bnc_offset, size_t* part_offset) {
  int rc = head_check(n, S, d, C);
  if (rc) return rc;
  if (!dym_offset || !bnc_offset || !part_offset) return CGCN_ERR_BAD_ARG;
  *dym_offset = 4 * (align4(ws_stats(S, d)) + align4(ws_loss(n)));
  *bnc_offset = *dym_offset + 4 * align4(ws_dym(n, d));
  *part_offset = *bnc_offset + 4 * align4(ws_bnc(d));
  return CGCN_OK;
}

int cgcn_head_bwd_partials(int n) { return head_bwd_partials(n); }

size_t cgcn_head_workspace_bytes(int n, int S, int d, int C) {
  if (head_check(n, S, d, C) != CGCN_OK) return 0;
  return 4 * (align4(ws_stats(S, d)) + align4(ws_loss(n)) + align4(ws_dym(n, d)) + align4(ws_bnc(d)) + align4(ws_part(n, d, C)));
}

int cgcn_head_fwd(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                  const float* bn_b, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                  float eps, int training, const float* Wout, const float* bout, const float* target, float dropout_p,
                  const unsigned long long* rng_state, float* probs, float* loss, float* dpred, float* save_mean,
                  float* save_invstd, void* workspace, size_t workspace_bytes) {
  int rc = head_check(n, S, d, C);
  if (rc) return rc;
  if (!X || !bn_w || !bn_b || !Wout || !bout || !target || !probs || !loss || !workspace) return CGCN_ERR_BAD_ARG;
  if (!run_mean || !run_var) return CGCN_ERR_BAD_ARG;
  if (training && (!save_mean || !save_invstd || n < 2)) return CGCN_ERR_BAD_ARG;  // BatchNorm needs > 1 value per channel
  const bool drop = training && dropout_p > 0.f;
  if (drop && (!rng_state || dropout_p >= 1.f)) return CGCN_ERR_BAD_ARG;
  if (workspace_bytes < cgcn_head_workspace_bytes(n, S, d, C)) return CGCN_ERR_WORKSPACE;
  if (misaligned16(Wout) || misaligned16(workspace) || misaligned16(X)) return CGCN_ERR_BAD_ARG;  // vector row accesses
  hipStream_t st = (hipStream_t)stream;
  float* w_stats = (float*)workspace;
  float* w_loss = w_stats + align4(ws_stats(S, d));
  if (n == 0) {
    hipLaunchKernelGGL(k_sum_scale, dim3(1), dim3(256), 0, st, 0, w_loss, 0.f, loss);
    return launch_status();
  }
  if (training) {
    int rpb;
    const int nblk = head_stat_blocks(n, &rpb);
    if (d == 128) hipLaunchKernelGGL((k_head_colstats<128>), dim3(nblk), dim3(256), 0, st, n, S, rpb, X, w_stats);
    else hipLaunchKernelGGL((k_head_colstats<256>), dim3(nblk), dim3(256), 0, st, n, S, rpb, X, w_stats);
    if ((rc = launch_status())) return rc;
    launch_bn_finalize(st, n, S, d, nblk, rpb, w_stats, momentum, eps, run_mean, run_var, num_batches_tracked, save_mean, save_invstd);
    if ((rc = launch_status())) return rc;
  }
  const int blocks = (n + HEAD_TILE - 1) / HEAD_TILE;
  const float keep_scale = drop ? 1.f / (1.f - dropout_p) : 1.f;
  const uint32_t thresh = drop ? dropout_threshold(dropout_p) : 0u;
  const float inv_count = 1.f / ((float)n * (float)C);
  const float* mean = training ? save_mean : run_mean;
  const float* vr = training ? save_invstd : run_var;
#define HF(D_, NC_)                                                                                                   \
  hipLaunchKernelGGL((k_head_fwd<D_, NC_>), dim3(blocks), dim3(512), 0, st, n, S, C, X, bn_w, bn_b, mean, vr,         \
                     training ? 0 : 1, eps, Wout, bout, target, keep_scale, thresh, rng_state, inv_count, probs,      \
                     training ? dpred : nullptr, w_loss, nullptr)
  if (d == 128) { if (C <= 128) HF(128, 1); else HF(128, 2); }
  else { if (C <= 128) HF(256, 1); else HF(256, 2); }
#undef HF
  if ((rc = launch_status())) return rc;
  hipLaunchKernelGGL(k_sum_scale, dim3(1), dim3(256), 0, st, blocks, w_loss, inv_count, loss);
  return launch_status();
}

// The eval-mode head of ONE ChromeGCN.forward call per strand (models/ChromeModels.py:48-51 with the module in eval
// mode: running statistics, dropout off): logits[s] = BatchNorm1d(relu(X[s])) W_out^T + b_out.  No strand mean, no loss.
int cgcn_head_logits(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                     const float* bn_b, const float* run_mean, const float* run_var, float eps, const float* Wout,
                     const float* bout, float* logits) {
  int rc = head_check(n, S, d, C);
  if (rc) return rc;
  if (!X || !bn_w || !bn_b || !run_mean || !run_var || !Wout || !bout || !logits) return CGCN_ERR_BAD_ARG;
  if (misaligned16(Wout) || misaligned16(X)) return CGCN_ERR_BAD_ARG;  // vector row accesses
  if (n == 0) return CGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (n + HEAD_TILE - 1) / HEAD_TILE;
  for (int s = 0; s < S; ++s) {
    const float* Xs = X + (size_t)s * n * d;
    float* Ls = logits + (size_t)s * n * C;
#define HF(D_, NC_)                                                                                                   \
  hipLaunchKernelGGL((k_head_fwd<D_, NC_>), dim3(blocks), dim3(512), 0, st, n, 1, C, Xs, bn_w, bn_b, run_mean, run_var, \
                     1, eps, Wout, bout, nullptr, 1.f, 0u, nullptr, 0.f, nullptr, nullptr, nullptr, Ls)
    if (d == 128) { if (C <= 128) HF(128, 1); else HF(128, 2); }
    else { if (C <= 128) HF(256, 1); else HF(256, 2); }
#undef HF
    if ((rc = launch_status())) return rc;
  }
  return CGCN_OK;
}

}  // extern "C"

// phases: bit 0 = batch statistics (k_head_colstats when the producer did not supply them, k_head_bn_finalize),
// bit 1 = k_head_fused (every label pass), bit 2 = k_head_train_finish.  cgcn_head_train runs all three;
// cgcn_debug_head_train_phases lets a profiler time them one at a time (on the state an earlier full call left).
static int head_train_impl(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                    const float* bn_b, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                    float eps, const float* Wout, const float* bout, const float* target, float dropout_p,
                    const unsigned long long* rng_state, float* probs, float* loss, float* save_mean, float* save_invstd,
                    const float* col_stats, int col_stats_tiles, int col_stats_rows, void* workspace,
                    size_t workspace_bytes, int phases) {
  int rc = head_check(n, S, d, C);
  if (rc) return rc;
  if (!X || !bn_w || !bn_b || !Wout || !bout || !target || !probs || !loss || !workspace || !run_mean || !run_var ||
      !save_mean || !save_invstd || n < 2)
    return CGCN_ERR_BAD_ARG;
  // accumulate mode (cgcn_layer_fwd_colstats_plan reported rows = -1): the buffer holds integer totals, not records
  const size_t acc_tile_bytes = (size_t)S * d * 2 * sizeof(float);
  const bool stat_acc = col_stats && col_stats_rows == -1;
  if (stat_acc && ((size_t)col_stats_tiles * acc_tile_bytes < stat_acc_words(S, d) * 8 || ((uintptr_t)col_stats & 7)))
    return CGCN_ERR_BAD_ARG;
  if (col_stats && !stat_acc && (col_stats_rows < 1 || col_stats_tiles != (n + col_stats_rows - 1) / col_stats_rows)) return CGCN_ERR_BAD_ARG;
  const bool drop = dropout_p > 0.f;
  if (drop && (!rng_state || dropout_p >= 1.f)) return CGCN_ERR_BAD_ARG;
  if (workspace_bytes < cgcn_head_workspace_bytes(n, S, d, C)) return CGCN_ERR_WORKSPACE;
  if (misaligned16(Wout) || misaligned16(workspace) || misaligned16(X)) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* w_stats = (float*)workspace;
  float* w_loss = w_stats + align4(ws_stats(S, d));
  float* w_dym = w_loss + align4(ws_loss(n));
  float* w_bnc = w_dym + align4(ws_dym(n, d));
  float* w_part = w_bnc + align4(ws_bnc(d));
  int rpb;
  int nblk = head_stat_blocks(n, &rpb);
  const float* stats = w_stats;
  if (col_stats) {
    // first stage already done by the producer of X (cgcn_layer_fwd's colstats output)
    stats = col_stats;
    nblk = col_stats_tiles;
    rpb = col_stats_rows;
  } else if (phases & 1) {
    if (d == 128) hipLaunchKernelGGL((k_head_colstats<128>), dim3(nblk), dim3(256), 0, st, n, S, rpb, X, w_stats);
    else hipLaunchKernelGGL((k_head_colstats<256>), dim3(nblk), dim3(256), 0, st, n, S, rpb, X, w_stats);
    if ((rc = launch_status())) return rc;
  }
  if ((phases & 1) && !stat_acc) {
    launch_bn_finalize(st, n, S, d, nblk, rpb, stats, momentum, eps, run_mean, run_var, num_batches_tracked, save_mean, save_invstd);
    if ((rc = launch_status())) return rc;
  }
  // accumulate mode: no finalize launch; the main kernel reads the totals (and its first workgroup finishes the bookkeeping --
  // only when the whole call runs: a profiling call of phase 2 alone must not touch the running statistics again)
  const HeadStatAcc sa = {stat_acc ? (const unsigned long long*)col_stats : nullptr, eps, momentum,
                          run_mean, run_var, (phases & 1) ? num_batches_tracked : nullptr, save_mean, save_invstd, loss};
  const HeadStatAcc sa_prof = {sa.acc, eps, 0.f, run_mean, run_var, nullptr, save_mean, save_invstd, loss};
  const int P = head_bwd_partials(n);
  const int CP = head_cp(C);
  const float keep_scale = drop ? 1.f / (1.f - dropout_p) : 1.f;
  const uint32_t thresh = drop ? dropout_threshold(dropout_p) : 0u;
  const float inv_count = 1.f / ((float)n * (float)C);
  // label passes of at most 128 (k_head_fused): one launch per pass
  for (int c0 = 0; c0 < C && (phases & 2); c0 += 128) {
    const int Cp = C - c0 < 128 ? C - c0 : 128;
    const int first = c0 == 0, last = c0 + 128 >= C;
#define HFU(D_)                                                                                                        \
    hipLaunchKernelGGL((k_head_fused<D_, 8>), dim3(P), dim3(512), 0, st, n, S, C, X, bn_w, bn_b, save_mean, save_invstd, \
                       Wout, bout, target, keep_scale, thresh, rng_state, inv_count, probs, w_loss, w_dym, w_part, c0, Cp, \
                       CP, first, last, (phases & 1) ? sa : sa_prof)
    if (d == 128 && HEAD_RS) {
      const bool nb7 = Cp > 96 && Cp <= 112;
      // Tile height.  A launch lasts (tiles per workgroup + 1) periods -- the Q team runs one tile behind the P team --,
      // so 1.1 tiles per workgroup cost 3 periods of 32-row tiles; 16-row tiles cost HEAD_TR16_COST_PCT % of a 32-row
      // period each (their products reuse every W_out operand half as often) but quantise the work twice as finely.
#ifndef HEAD_TR16_COST_PCT
#define HEAD_TR16_COST_PCT 55   // measured 55 < 60 < 75 = off (0: 32-row tiles always); profiles/r04_head_tile_height.txt
#endif
      auto periods = [&](int tr) { return (((n + tr - 1) / tr) + P - 1) / P + 1; };
      const bool tr16 = HEAD_TR16_COST_PCT > 0 && periods(16) * HEAD_TR16_COST_PCT < periods(32) * 100;
#define HRS2(M_, NB_, DR_, NRB_)                                                                                       \
        hipLaunchKernelGGL((k_head_fused_rs<M_, NB_, DR_, NRB_>), dim3(P), dim3(1024), 0, st, n, S, C, X, bn_w, bn_b, save_mean, save_invstd, \
                         Wout, bout, target, keep_scale, thresh, rng_state, inv_count, probs, w_loss, w_dym, w_part, c0, Cp, \
                         CP, first, last, (phases & 1) ? sa : sa_prof)
#define HSP(M_, NB_, DR_)                                                                                              \
        hipLaunchKernelGGL((k_head_fused_sp<M_, NB_, DR_>), dim3(P), dim3(1024), 0, st, n, S, C, X, bn_w, bn_b, save_mean, save_invstd, \
                         Wout, bout, target, keep_scale, thresh, rng_state, inv_count, probs, w_loss, w_dym, w_part, c0, Cp, \
                         CP, first, last, (phases & 1) ? sa : sa_prof)
      // split products (cgcn_debug_set_products; cgcn_common.hpp): k_head_fused_sp, 16-row tiles at every size
      const bool sph = cgcn_debug_get_products() != CGCN_PRODUCTS_FP32_CHAIN;
#define HRS(M_, NB_)                                                                                                   \
      do {                                                                                                             \
        if (sph) { if (thresh) HSP(M_, NB_, true); else HSP(M_, NB_, false); }                                          \
        else if (thresh) { if (tr16) HRS2(M_, NB_, true, 1); else HRS2(M_, NB_, true, 2); }                             \
        else { if (tr16) HRS2(M_, NB_, false, 1); else HRS2(M_, NB_, false, 2); }                                       \
      } while (0)
      if (C <= 128) { if (nb7) HRS(false, 7); else HRS(false, 8); }
      else { if (nb7) HRS(true, 7); else HRS(true, 8); }
#undef HRS
#undef HRS2
#undef HSP
    }
    else if (d == 128) HFU(128);
    else HFU(256);
#undef HFU
    if ((rc = launch_status())) return rc;
  }
  if (!(phases & 4) || stat_acc) return CGCN_OK;   // accumulate mode: the main kernel left the backward sums as integer
                                                    // totals (cgcn_head_grad::stat_acc) and wrote the loss itself
  hipLaunchKernelGGL(k_head_train_finish, dim3(1 + d / HEAD_STAT_COLS), dim3(512), 0, st, P, n, S, d, CP, w_part, w_bnc,
                     w_loss, inv_count, loss);
  return launch_status();
}

extern "C" {

int cgcn_head_train(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                    const float* bn_b, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                    float eps, const float* Wout, const float* bout, const float* target, float dropout_p,
                    const unsigned long long* rng_state, float* probs, float* loss, float* save_mean, float* save_invstd,
                    const float* col_stats, int col_stats_tiles, int col_stats_rows, void* workspace,
                    size_t workspace_bytes) {
  return head_train_impl(stream, n, S, d, C, X, bn_w, bn_b, run_mean, run_var, num_batches_tracked, momentum, eps, Wout, bout,
                         target, dropout_p, rng_state, probs, loss, save_mean, save_invstd, col_stats, col_stats_tiles,
                         col_stats_rows, workspace, workspace_bytes, 7);
}

int cgcn_debug_head_train_phases(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                                 const float* bn_b, float* run_mean, float* run_var, long long* num_batches_tracked,
                                 float momentum, float eps, const float* Wout, const float* bout, const float* target,
                                 float dropout_p, const unsigned long long* rng_state, float* probs, float* loss,
                                 float* save_mean, float* save_invstd, const float* col_stats, int col_stats_tiles,
                                 int col_stats_rows, void* workspace, size_t workspace_bytes, int phases) {
  if (phases < 1 || phases > 7) return CGCN_ERR_BAD_ARG;
  return head_train_impl(stream, n, S, d, C, X, bn_w, bn_b, run_mean, run_var, num_batches_tracked, momentum, eps, Wout, bout,
                         target, dropout_p, rng_state, probs, loss, save_mean, save_invstd, col_stats, col_stats_tiles,
                         col_stats_rows, workspace, workspace_bytes, phases);
}

int cgcn_head_bwd(cgcn_stream_t stream, int n, int S, int d, int C, const float* X, const float* bn_w,
                  const float* bn_b, const float* save_mean, const float* save_invstd, const float* Wout,
                  const float* dpred, const float* dloss, float dropout_p, const unsigned long long* rng_state,
                  float* dX, float* dWout, float* dbout, float* dbn_w, float* dbn_b, int accumulate, void* workspace,
                  size_t workspace_bytes) {
  int rc = head_check(n, S, d, C);
  if (rc) return rc;
  if (!X || !bn_w || !bn_b || !save_mean || !save_invstd || !Wout || !dWout || !dbout || !dbn_w || !dbn_b || !workspace)
    return CGCN_ERR_BAD_ARG;
  const bool fused = dpred == nullptr;  // the workspace already holds dym + partials from cgcn_head_train
  if (fused && dX) return CGCN_ERR_UNSUPPORTED;  // the fused path exists only in deferred mode (dX == NULL)
  const bool drop = dropout_p > 0.f;
  if (drop && (!rng_state || dropout_p >= 1.f)) return CGCN_ERR_BAD_ARG;
  if (workspace_bytes < cgcn_head_workspace_bytes(n, S, d, C)) return CGCN_ERR_WORKSPACE;
  if (misaligned16(X) || (dX && misaligned16(dX)) || misaligned16(workspace) || n < 1) return CGCN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  float* w = (float*)workspace;
  float* w_dym = w + align4(ws_stats(S, d)) + align4(ws_loss(n));
  float* w_bnc = w_dym + align4(ws_dym(n, d));
  float* w_part = w_bnc + align4(ws_bnc(d));
  const int P = head_bwd_partials(n);
  const int CP = head_cp(C);
  const float keep_scale = drop ? 1.f / (1.f - dropout_p) : 1.f;
  const uint32_t thresh = drop ? dropout_threshold(dropout_p) : 0u;
  if (!fused) {
    for (int c0 = 0; c0 < C; c0 += 128) {   // label passes of at most 128 (k_head_bwd)
      const int Cp = C - c0 < 128 ? C - c0 : 128;
      const int first = c0 == 0, last = c0 + 128 >= C;
#define HB(D_)                                                                                                         \
      hipLaunchKernelGGL((k_head_bwd<D_, 8>), dim3(P), dim3(512), 0, st, n, S, C, X, bn_w, bn_b, save_mean, save_invstd, \
                         Wout, dpred, dloss, keep_scale, thresh, rng_state, w_dym, w_part, c0, Cp, CP, first, last)
      if (d == 128) HB(128); else HB(256);
#undef HB
      if ((rc = launch_status())) return rc;
    }
  }
  const float* fin_scale = fused ? dloss : nullptr;  // the unfused kernel already multiplied dpred by d loss
  const int wslabs = (CP * d + CP) / 64, sblocks = d / HEAD_STAT_COLS;  // CP is 128 or 256: slab aligned
  if (fused) {
    // cgcn_head_train left dym, bnc (for d loss = 1) and the partials; every remaining sum -- dW_out, db_out, d(bn
    // weight), d(bn bias) -- rides in k_bwd_rowlocal's extra workgroups (cgcn_head_grad.part / dW_out / db_out /
    // dbn_w / dbn_b), scaled there by d loss.  Nothing to launch.
    return CGCN_OK;
  }
  if (!dX) {
    // deferred mode: only the BatchNorm columns now (cgcn_layer_bwd needs bnc); the dW_out / db_out slabs ride at the
    // end of k_bwd_rowlocal's grid (cgcn_head_grad.part / dW_out / db_out)
    hipLaunchKernelGGL(k_head_bwd_finalize, dim3(sblocks), dim3(512), 0, st, 0, 0, P, n, S, d, C, CP, w_part, dWout,
                       dbout, dbn_w, dbn_b, w_bnc, accumulate, fin_scale);
    return launch_status();
  }
  hipLaunchKernelGGL(k_head_bwd_finalize, dim3(wslabs + sblocks), dim3(512), 0, st, 0, wslabs, P, n, S, d, C, CP, w_part,
                     dWout, dbout, dbn_w, dbn_b, w_bnc, accumulate, fin_scale);
  if ((rc = launch_status())) return rc;
  const size_t total4 = (size_t)S * n * d / 4;
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (d == 128)
    hipLaunchKernelGGL((k_head_bn_bwd_apply<128>), dim3(blocks), dim3(256), 0, st, n, S, X, bn_w, save_mean, save_invstd, w_dym,
                       w_bnc, keep_scale, thresh, rng_state, dX);
  else
    hipLaunchKernelGGL((k_head_bn_bwd_apply<256>), dim3(blocks), dim3(256), 0, st, n, S, X, bn_w, save_mean, save_invstd, w_dym,
                       w_bnc, keep_scale, thresh, rng_state, dX);
  return launch_status();
}

}  // extern "C"
