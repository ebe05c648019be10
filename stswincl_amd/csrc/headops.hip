// Decode-head and loss kernels on NHWC "token" matrices [M = frames*H*W rows][C channels] (HBM-bound work;
// 16-byte pieces per lane everywhere).  BatchNorm is grouped: the M rows are G equal groups with separate batch
// statistics (G = 1 for the head; G = frames lets the per-frame ResNet calls of base18.py:86-89 be batched without
// changing the per-frame statistics the reference computes).
#include "common.h"
#include <cstdlib>

// ----------------------------------------------------------------------------------------- column statistics
// sum[g][c] += sum_rows (x - pivot), sumsq[g][c] += sum_rows (x - pivot)^2, pivot = x[first row of group][c]
// (shifted sums keep E[x^2]-E[x]^2 well conditioned).  Block = 32 column pieces x 8 row lanes.
// Thread layout of the column-reduction / column-owner kernels: `cpb` column pieces (16 B each) x `256/cpb` row lanes,
// cpb = min(32, C/PACK) rounded down to a power of two, so narrow matrices (C = 64) still use every thread.
// rows per block for the column-reduction kernels: as many as possible (the LDS fold and the atomics are per block)
// while still launching >= ~1024 blocks
static int reduce_rows_per_chunk(int group_rows, int other_blocks, int nrl) {
  int rpc = 8 * nrl;
  while (rpc < 64 * nrl && (long)other_blocks * ((group_rows + 2 * rpc - 1) / (2 * rpc)) >= 1024) rpc *= 2;
  return rpc;
}
// Rows per thread of the BatchNorm backward passes: as many as leave about 512 (reduce pass: the LDS fold and the slab are per
// workgroup) or 2048 (dx pass) workgroups.  Until round 5 the dx pass ran 512, too: every thread loaded its 8 channels' constants
// from six arrays (48 loads + the arithmetic in front of the first row - a quarter of the pass, and the reason why smaller
// workgroups were slower); now one thread per channel computes them and LDS hands them out, and the dx pass is 25-30 % faster at
// 2048 workgroups (layer5: 63 -> 43 us, stem: 96 -> 79 us on cold operands; profiles/r05_bn_dx_prologue_experiment.txt,
// r05_bn_bwd_lds_constants_sweep.txt; step +1.1 % same box).
static int bn_bwd_rows(long rows_all_groups, int col_blocks, int nrl, int max_rows) {
  // (workgroup target: STSWIN_BN_RED_WGS for the reduce pass [max_rows 64], STSWIN_BN_DX_WGS for the dx pass [max_rows 32]; tuning knobs, read once)
  static const int tgt_red = getenv("STSWIN_BN_RED_WGS") ? atoi(getenv("STSWIN_BN_RED_WGS")) : 512;
  static const int tgt_dx = getenv("STSWIN_BN_DX_WGS") ? atoi(getenv("STSWIN_BN_DX_WGS")) : 2048;
  const int tgt = max_rows > 32 ? tgt_red : tgt_dx;
  int rows = 8;
  while (rows < max_rows && (long)col_blocks * (rows_all_groups / ((long)2 * rows * nrl)) >= tgt) rows *= 2;
  return rows;
}

// First physical row of the rows a block works on, minus its first row index inside the group: contiguous groups
// (unit == 0) start at g * group_rows; interleaved groups (unit > 0: group g owns the units g, g + G, g + 2G, ... of `unit`
// rows - frame t of every clip when the batch is stored clip-major) map the chunk's unit; a chunk never straddles a unit
// (the launchers make rows_per_chunk divide unit).
DEVI long group_row0(int g, int ch, int rows_per_chunk, int group_rows, int unit, int G) {
  if (unit <= 0) return (long)g * group_rows;
  return ((long)((ch * rows_per_chunk) / unit) * (G - 1) + g) * unit;
}
DEVI long group_first_row(int g, int group_rows, int unit) { return unit > 0 ? (long)g * unit : (long)g * group_rows; }
static int fit_chunk(int rpc, int unit) {            // largest power-of-two shrink of rpc that divides unit; 0: impossible
  if (unit <= 0) return rpc;
  while (rpc > 1 && unit % rpc) rpc >>= 1;
  return unit % rpc ? 0 : rpc;
}

static int pick_cpb(int pieces_per_row) {
  int c = 32;
  while (c > pieces_per_row) c >>= 1;
  return c < 1 ? 1 : c;
}

template <typename T, bool SQ>
__global__ __launch_bounds__(256) void colstats_kernel(const T* x, long ldx, float* part, int C, int group_rows,
                                                        int chunks_per_group, int rows_per_chunk, int cpb, int unit) {
  constexpr int PACK = TT<T>::PACK;
  __shared__ float lpart[2][256 * 8];
  const int cp = threadIdx.x % cpb, rl = threadIdx.x / cpb, nrl = 256 / cpb;
  const int c = (blockIdx.x * cpb + cp) * PACK;
  const int g = blockIdx.y / chunks_per_group, ch = blockIdx.y % chunks_per_group;
  const long gr0 = group_row0(g, ch, rows_per_chunk, group_rows, unit, gridDim.y / chunks_per_group);
  float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c < C) {
    if (SQ) {
      Vec16<T> p0;
      p0.v = *(const decltype(p0.v)*)(x + group_first_row(g, group_rows, unit) * ldx + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e) pv[e] = p0.get(e);
    }
    const int r_end = min(group_rows, (ch + 1) * rows_per_chunk);
    for (int r = ch * rows_per_chunk + rl; r < r_end; r += nrl) {
      Vec16<T> in;
      in.v = *(const decltype(in.v)*)(x + (gr0 + r) * ldx + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e) {
        const float d = in.get(e) - pv[e];
        a1[e] += d;
        if (SQ) a2[e] += d * d;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { lpart[0][(rl * cpb + cp) * 8 + e] = a1[e]; lpart[1][(rl * cpb + cp) * 8 + e] = a2[e]; }
  __syncthreads();
  {   // parallel fold: thread t < cpb*8 owns (column piece t/8, element t%8) and adds the nrl row-lane partials
    const int t = threadIdx.x, fc = t >> 3, fe = t & 7;
    const int cc = (blockIdx.x * cpb + fc) * PACK + fe;
    if (fc < cpb && fe < PACK && cc < C) {
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < nrl; ++k) { s1 += lpart[0][(k * cpb + fc) * 8 + fe]; s2 += lpart[1][(k * cpb + fc) * 8 + fe]; }
      // slab blockIdx.y = (group, row chunk) of the caller's scratch: [2][C] (plain stores; slab_fold_kernel adds the chunks of a
      // group in order: deterministic statistics, no same-address atomics)
      float* slab = part + (long)blockIdx.y * 2 * C;
      slab[cc] = s1;
      if (SQ) slab[C + cc] = s2;
    }
  }
}

// mean/rstd per (group, channel) from the shifted sums; running-stat update group by group in order
// (nn.BatchNorm2d momentum semantics: running = (1-m) running + m stat, unbiased variance).
template <typename T>
__global__ void bn_finalize_kernel(const T* x, long ldx, const float* sum, const float* sumsq, float* mean, float* rstd,
                                   float* running_mean, float* running_var, int C, int G, int group_rows, float eps,
                                   float momentum, int unit) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  for (int g = 0; g < G; ++g) {
    const float pivot = x ? to_f32<T>(x[group_first_row(g, group_rows, unit) * ldx + c]) : 0.f;   // (sums from a GEMM epilogue: raw)
    const float ms = sum[(long)g * C + c] / group_rows;
    const float var = fmaxf(sumsq[(long)g * C + c] / group_rows - ms * ms, 0.f);
    mean[(long)g * C + c] = pivot + ms;
    rstd[(long)g * C + c] = rsqrtf(var + eps);
    rm = (1.f - momentum) * rm + momentum * (pivot + ms);
    rv = (1.f - momentum) * rv + momentum * var * ((float)group_rows / (float)max(group_rows - 1, 1));
  }
  if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// y = act( (x-mean)*rstd*gamma + beta [+ resid] ).  A thread owns one 16-byte channel piece and walks rows (stride 8
// inside its row chunk), so the per-channel scale/shift live in registers instead of being re-loaded per piece.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* x, long ldx, const float* mean, const float* rstd,
                                                        const float* gamma, const float* beta, const T* resid, long ldr,
                                                        T* y, long ldy, int C, int group_rows, int chunks_per_group,
                                                        int rows_per_chunk, int relu, int cpb, int unit) {
  constexpr int PACK = TT<T>::PACK;
  __shared__ __attribute__((aligned(16))) float coef[2][256];
  const int cp = threadIdx.x % cpb, rl = threadIdx.x / cpb, nrl = 256 / cpb;
  const int c = (blockIdx.x * cpb + cp) * PACK;
  const int g = blockIdx.y / chunks_per_group, ch = blockIdx.y % chunks_per_group;
  const long gr0 = group_row0(g, ch, rows_per_chunk, group_rows, unit, gridDim.y / chunks_per_group);
  {   // scale / shift of the workgroup's cpb * PACK <= 256 channels: one thread per channel, handed out through LDS (bn_bwd_dx_kernel)
    const int t = threadIdx.x, cc = blockIdx.x * cpb * PACK + t;
    if (t < cpb * PACK && cc < C) {
      const float rs = rstd[(long)g * C + cc] * gamma[cc];
      coef[0][t] = rs;
      coef[1][t] = beta[cc] - mean[(long)g * C + cc] * rs;
    }
  }
  __syncthreads();
  if (c >= C) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < PACK; e += 4) {
    *(f32x4*)(sc + e) = *(const f32x4*)&coef[0][cp * PACK + e];
    *(f32x4*)(sh + e) = *(const f32x4*)&coef[1][cp * PACK + e];
  }
  const int r_end = min(group_rows, (ch + 1) * rows_per_chunk);
  for (int r = ch * rows_per_chunk + rl; r < r_end; r += nrl) {
    Vec16<T> in, rs, o;
    in.v = *(const decltype(in.v)*)(x + (gr0 + r) * ldx + c);
    if (resid) rs.v = *(const decltype(rs.v)*)(resid + (gr0 + r) * ldr + c);
#pragma unroll
    for (int e = 0; e < PACK; ++e) {
      float v = in.get(e) * sc[e] + sh[e];
      if (resid) v += rs.get(e);
      if (relu) v = fmaxf(v, 0.f);
      o.set(e, v);
    }
    *(decltype(o.v)*)(y + (gr0 + r) * ldy + c) = o.v;
  }
}

// pass 1 of the backward: s1[g][c] = sum dyr, s2[g][c] = sum dyr*xhat, dyr = dy * (y > 0 if relu)
// Both passes walk their rows four at a time with all loads of the four rows issued first (one row per iteration left a
// thread with 2-3 requests in flight: 2.0-3.6 TB/s against the 5.4-6.5 TB/s of bn_apply, tools/bench_bn.py).
// ReLU mask: from the stored output y, or - when there is no residual, y == NULL - recomputed from x with bn_apply's own
// expression x * (rstd*gamma) + (beta - mean*rstd*gamma) > 0 (the same operations in the same order: the same sign): one
// tensor less to read in both passes.
#ifndef BN_BWD_U
#define BN_BWD_U 4
#endif
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* dy, long lddy, const T* x, long ldx, const T* y, long ldy,
                                                             const float* mean, const float* rstd, float* spart,
                                                             int C, int group_rows, int chunks_per_group, int rows_per_chunk,
                                                             int relu, int cpb, const float* gamma, const float* beta, int unit) {
  constexpr int PACK = TT<T>::PACK;
  constexpr int U = BN_BWD_U;
  __shared__ __attribute__((aligned(16))) float part[2][256 * 8];
  const int cp = threadIdx.x % cpb, rl = threadIdx.x / cpb, nrl = 256 / cpb;
  const int c = (blockIdx.x * cpb + cp) * PACK;
  const int g = blockIdx.y / chunks_per_group, ch = blockIdx.y % chunks_per_group;
  const long gr0 = group_row0(g, ch, rows_per_chunk, group_rows, unit, gridDim.y / chunks_per_group);
  float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool remask = relu && y == nullptr, ymask = relu && y != nullptr;
  {   // channel constants: one thread per channel, handed out through LDS (see bn_bwd_dx_kernel); `part` is free until the fold
    const int t = threadIdx.x, cc = blockIdx.x * cpb * PACK + t;
    if (t < cpb * PACK && cc < C) {
      const float m = mean[(long)g * C + cc], r = rstd[(long)g * C + cc];
      const float pmv = remask ? r * gamma[cc] : 0.f;
      part[0][t] = m; part[0][256 + t] = r; part[0][512 + t] = pmv; part[0][768 + t] = remask ? beta[cc] - m * pmv : 0.f;
    }
  }
  __syncthreads();
  float mu[8], rs[8], pm[8], qm[8];
  if (c < C) {
#pragma unroll
    for (int e = 0; e < PACK; e += 4) {
      *(f32x4*)(mu + e) = *(const f32x4*)&part[0][cp * PACK + e];
      *(f32x4*)(rs + e) = *(const f32x4*)&part[0][256 + cp * PACK + e];
      *(f32x4*)(pm + e) = *(const f32x4*)&part[0][512 + cp * PACK + e];
      *(f32x4*)(qm + e) = *(const f32x4*)&part[0][768 + cp * PACK + e];
    }
  }
  __syncthreads();
  if (c < C) {
    const int r_end = min(group_rows, (ch + 1) * rows_per_chunk);
    for (int r = ch * rows_per_chunk + rl; r < r_end; r += U * nrl) {
      Vec16<T> d[U], xi[U], yo[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r + u * nrl < r_end) {
          const long row = gr0 + r + u * nrl;
          d[u].v = *(const decltype(d[u].v)*)(dy + row * lddy + c);
          xi[u].v = *(const decltype(xi[u].v)*)(x + row * ldx + c);
          if (ymask) yo[u].v = *(const decltype(yo[u].v)*)(y + row * ldy + c);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r + u * nrl < r_end) {
#pragma unroll
          for (int e = 0; e < PACK; ++e) {
            float dv = d[u].get(e);
            const float xv = xi[u].get(e);
            if (remask) { if (!(xv * pm[e] + qm[e] > 0.f)) dv = 0.f; }
            else if (ymask) { if (!(yo[u].get(e) > 0.f)) dv = 0.f; }
            a1[e] += dv;
            a2[e] += dv * ((xv - mu[e]) * rs[e]);
          }
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { part[0][(rl * cpb + cp) * 8 + e] = a1[e]; part[1][(rl * cpb + cp) * 8 + e] = a2[e]; }
  __syncthreads();
  {
    const int t = threadIdx.x, fc = t >> 3, fe = t & 7;
    const int cc = (blockIdx.x * cpb + fc) * PACK + fe;
    if (fc < cpb && fe < PACK && cc < C) {
      float t1 = 0.f, t2 = 0.f;
      for (int k = 0; k < nrl; ++k) { t1 += part[0][(k * cpb + fc) * 8 + fe]; t2 += part[1][(k * cpb + fc) * 8 + fe]; }
      float* slab = spart + (long)blockIdx.y * 2 * C;          // [(group, chunk)][2][C]: summed per group by slab_fold_kernel
      slab[cc] = t1;
      slab[C + cc] = t2;
    }
  }
}

// pass 2: dx = gamma*rstd*(dyr - [training] (s1 + xhat*s2)/n) ; dresid = dyr (optional).  Column-owner threads.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* dy, long lddy, const T* x, long ldx, const T* y, long ldy,
                                                         const float* mean, const float* rstd, const float* gamma,
                                                         const float* s1, const float* s2, T* dx, long lddx, T* dres,
                                                         long lddr, int C, int group_rows, int chunks_per_group,
                                                         int rows_per_chunk, int relu, int training, int cpb, float inv_n,
                                                         const float* beta, int unit, float* gsum) {
  constexpr int PACK = TT<T>::PACK;
  constexpr int U = BN_BWD_U;
  __shared__ __attribute__((aligned(16))) float coef[5][256];
  const int cp = threadIdx.x % cpb, rl = threadIdx.x / cpb, nrl = 256 / cpb;
  const int c = (blockIdx.x * cpb + cp) * PACK;
  const int g = blockIdx.y / chunks_per_group, ch = blockIdx.y % chunks_per_group;
  const long gr0 = group_row0(g, ch, rows_per_chunk, group_rows, unit, gridDim.y / chunks_per_group);
  const bool remask = relu && y == nullptr, ymask = relu && y != nullptr;
  // dx = A*dyr + B*x + D  with  A = gamma*rstd, B = -A*rstd*s2/n, D = -A*s1/n - B*mean      (training)
  // The workgroup's cpb * PACK <= 256 channels get their five constants from ONE thread each (coalesced loads) and LDS hands every
  // thread its 8: as 48 per-thread loads + the arithmetic in front of the first row the prologue was a quarter of the pass and
  // made more, smaller workgroups slower (profiles/r05_bn_dx_prologue_experiment.txt).
  {
    const int t = threadIdx.x, cc = blockIdx.x * cpb * PACK + t;
    if (t < cpb * PACK && cc < C) {
      const long gc = (long)g * C + cc;
      const float rsd = rstd[gc], mu = mean[gc], ga = gamma[cc];
      const float A = ga * rsd;
      const float kbv = training ? -A * rsd * s2[gc] * inv_n : 0.f;
      const float pmv = remask ? rsd * ga : 0.f;
      coef[0][t] = A;
      coef[1][t] = kbv;
      coef[2][t] = training ? -A * s1[gc] * inv_n - kbv * mu : 0.f;
      coef[3][t] = pmv;
      coef[4][t] = remask ? beta[cc] - mu * pmv : 0.f;
    }
  }
  __syncthreads();
  if (c >= C) return;
  if (gsum && blockIdx.y == 0 && rl == 0) {              // parameter gradients: dbeta | dgamma = s1 | s2 summed over the groups
    const int G = gridDim.y / chunks_per_group;
#pragma unroll
    for (int e = 0; e < PACK; ++e) {
      float a = 0.f, b = 0.f;
      for (int gg = 0; gg < G; ++gg) { a += s1[(long)gg * C + c + e]; b += s2[(long)gg * C + c + e]; }
      gsum[c + e] = a;
      gsum[C + c + e] = b;
    }
  }
  float ka[8], kb[8], kd[8], pm[8], qm[8];
#pragma unroll
  for (int e = 0; e < PACK; e += 4) {
    *(f32x4*)(ka + e) = *(const f32x4*)&coef[0][cp * PACK + e];
    *(f32x4*)(kb + e) = *(const f32x4*)&coef[1][cp * PACK + e];
    *(f32x4*)(kd + e) = *(const f32x4*)&coef[2][cp * PACK + e];
    *(f32x4*)(pm + e) = *(const f32x4*)&coef[3][cp * PACK + e];
    *(f32x4*)(qm + e) = *(const f32x4*)&coef[4][cp * PACK + e];
  }
  const int r_end = min(group_rows, (ch + 1) * rows_per_chunk);
  for (int r = ch * rows_per_chunk + rl; r < r_end; r += U * nrl) {
    Vec16<T> d[U], xi[U], yo[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r + u * nrl < r_end) {
        const long row = gr0 + r + u * nrl;
        d[u].v = *(const decltype(d[u].v)*)(dy + row * lddy + c);
        xi[u].v = *(const decltype(xi[u].v)*)(x + row * ldx + c);
        if (ymask) yo[u].v = *(const decltype(yo[u].v)*)(y + row * ldy + c);
      }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (r + u * nrl < r_end) {
        const long row = gr0 + r + u * nrl;
        Vec16<T> o, od;
#pragma unroll
        for (int e = 0; e < PACK; ++e) {
          float dv = d[u].get(e);
          const float xv = xi[u].get(e);
          if (remask) { if (!(xv * pm[e] + qm[e] > 0.f)) dv = 0.f; }
          else if (ymask) { if (!(yo[u].get(e) > 0.f)) dv = 0.f; }
          o.set(e, ka[e] * dv + kb[e] * xv + kd[e]);
          od.set(e, dv);
        }
        *(decltype(o.v)*)(dx + row * lddx + c) = o.v;
        if (dres) *(decltype(od.v)*)(dres + row * lddr + c) = od.v;
      }
  }
}

// out[r][c] (+)= v[r / group_rows][c] * scale      (image-pool broadcast and adaptive-avg-pool backward)
template <typename T>
__global__ __launch_bounds__(256) void rows_broadcast_kernel(const float* v, T* out, long ldo, int M, int C, int group_rows,
                                                              float scale, int accumulate) {
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)M * ppr) return;
  const int r = idx / ppr, c = (idx % ppr) * PACK;
  const float* src = v + (long)(r / group_rows) * C + c;
  T* dst = out + (long)r * ldo + c;
  Vec16<T> o;
  if (accumulate) o.v = *(const decltype(o.v)*)dst;
#pragma unroll
  for (int e = 0; e < PACK; ++e) o.set(e, (accumulate ? o.get(e) : 0.f) + src[e] * scale);
  *(decltype(o.v)*)dst = o.v;
}

// ----------------------------------------------------------------------------------------- bilinear (align_corners=False)
DEVI void bil_src(int dst, float scale, int n_in, int& i0, int& i1, float& w1) {
  float s = ((float)dst + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = s - (float)i0;
}

// NHWC tokens [F][h][w][C] -> [F][H][W][C]  (F.interpolate / F.upsample bilinear of base18.py:102-103)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* in, long ldi, T* out, long ldo, int F, int h, int w,
                                                            int H, int W, int C) {
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)F * H * W * ppr) return;
  const int c = (idx % ppr) * PACK;
  const long px = idx / ppr;
  const int x = px % W, y = (px / W) % H, f = px / ((long)W * H);
  int y0, y1, x0, x1; float wy, wx;
  bil_src(y, (float)h / H, h, y0, y1, wy);
  bil_src(x, (float)w / W, w, x0, x1, wx);
  const T* b = in + ((long)f * h * w) * ldi + c;
  Vec16<T> p00, p01, p10, p11, o;
  p00.v = *(const decltype(o.v)*)(b + ((long)y0 * w + x0) * ldi);
  p01.v = *(const decltype(o.v)*)(b + ((long)y0 * w + x1) * ldi);
  p10.v = *(const decltype(o.v)*)(b + ((long)y1 * w + x0) * ldi);
  p11.v = *(const decltype(o.v)*)(b + ((long)y1 * w + x1) * ldi);
#pragma unroll
  for (int e = 0; e < PACK; ++e)
    o.set(e, (1.f - wy) * ((1.f - wx) * p00.get(e) + wx * p01.get(e)) + wy * ((1.f - wx) * p10.get(e) + wx * p11.get(e)));
  *(decltype(o.v)*)(out + px * ldo + c) = o.v;
}

// gather-form backward: din[f][yi][xi][c] = sum over the output pixels that read (yi, xi)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* dout, long ldo, T* din, long ldi, int F, int h, int w,
                                                            int H, int W, int C) {
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)F * h * w * ppr) return;
  const int c = (idx % ppr) * PACK;
  const long px = idx / ppr;
  const int xi = px % w, yi = (px / w) % h, f = px / ((long)w * h);
  const float sy = (float)h / H, sx = (float)w / W;
  const int ylo = max(0, (int)floorf(((float)yi - 0.5f) / sy - 0.5f) - 1), yhi = min(H - 1, (int)ceilf(((float)yi + 1.5f) / sy - 0.5f) + 1);
  const int xlo = max(0, (int)floorf(((float)xi - 0.5f) / sx - 0.5f) - 1), xhi = min(W - 1, (int)ceilf(((float)xi + 1.5f) / sx - 0.5f) + 1);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int y = ylo; y <= yhi; ++y) {
    int y0, y1; float wy;
    bil_src(y, sy, h, y0, y1, wy);
    const float cy = (y0 == yi ? 1.f - wy : 0.f) + (y1 == yi ? wy : 0.f);
    if (cy == 0.f) continue;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1; float wx;
      bil_src(x, sx, w, x0, x1, wx);
      const float cx = (x0 == xi ? 1.f - wx : 0.f) + (x1 == xi ? wx : 0.f);
      if (cx == 0.f) continue;
      Vec16<T> d;
      d.v = *(const decltype(d.v)*)(dout + (((long)f * H + y) * W + x) * ldo + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e) acc[e] += cy * cx * d.get(e);
    }
  }
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < PACK; ++e) o.set(e, acc[e]);
  *(decltype(o.v)*)(din + px * ldi + c) = o.v;
}

// tokens [F][h][w][nc] (pitch ldi) -> NCHW logits [F][nc][H][W]  (nn.functional.interpolate of base18.py:106)
template <typename T, typename TO>
__global__ __launch_bounds__(256) void logits_up_fwd_kernel(const T* in, long ldi, TO* out, int F, int h, int w, int H, int W,
                                                             int nc) {
  const long px = (long)blockIdx.x * 256 + threadIdx.x;
  if (px >= (long)F * H * W) return;
  const int x = px % W, y = (px / W) % H, f = px / ((long)W * H);
  int y0, y1, x0, x1; float wy, wx;
  bil_src(y, (float)h / H, h, y0, y1, wy);
  bil_src(x, (float)w / W, w, x0, x1, wx);
  const T* b = in + ((long)f * h * w) * ldi;
  const T* p00 = b + ((long)y0 * w + x0) * ldi; const T* p01 = b + ((long)y0 * w + x1) * ldi;
  const T* p10 = b + ((long)y1 * w + x0) * ldi; const T* p11 = b + ((long)y1 * w + x1) * ldi;
  for (int c = 0; c < nc; ++c) {
    const float v = (1.f - wy) * ((1.f - wx) * to_f32<T>(p00[c]) + wx * to_f32<T>(p01[c])) +
                    wy * ((1.f - wx) * to_f32<T>(p10[c]) + wx * to_f32<T>(p11[c]));
    out[(((long)f * nc + c) * H + y) * W + x] = from_f32<TO>(v);
  }
}

// backward, separable gather: dtok[f][yi][xi][c] = sum_y cy(yi,y) sum_x cx(xi,x) dlogits[f][c][y][x]
// one thread per (f, yi, xi, c); nc is small so the tensor is tiny and L2-resident.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void logits_up_bwd_kernel(const TO* dout, T* din, long ldi, int F, int h, int w, int H, int W,
                                                             int nc) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long npx = (long)F * h * w;
  if (idx >= npx * nc) return;
  // neighbouring lanes = neighbouring low-resolution pixels of ONE class plane: their gather windows are adjacent stretches of
  // the same high-resolution rows (with the class as the fastest index every lane read a different plane: 82 us for 25 MB)
  const int c = (int)(idx / npx);
  const long px = idx - (long)c * npx;
  const int xi = px % w, yi = (px / w) % h, f = px / ((long)w * h);
  const float sy = (float)h / H, sx = (float)w / W;
  const int ylo = max(0, (int)floorf(((float)yi - 0.5f) / sy - 0.5f) - 1), yhi = min(H - 1, (int)ceilf(((float)yi + 1.5f) / sy - 0.5f) + 1);
  const int xlo = max(0, (int)floorf(((float)xi - 0.5f) / sx - 0.5f) - 1), xhi = min(W - 1, (int)ceilf(((float)xi + 1.5f) / sx - 0.5f) + 1);
  const TO* plane = dout + ((long)f * nc + c) * H * W;
  float acc = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    int y0, y1; float wy;
    bil_src(y, sy, h, y0, y1, wy);
    const float cy = (y0 == yi ? 1.f - wy : 0.f) + (y1 == yi ? wy : 0.f);
    if (cy == 0.f) continue;
    float row = 0.f;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1; float wx;
      bil_src(x, sx, w, x0, x1, wx);
      const float cx = (x0 == xi ? 1.f - wx : 0.f) + (x1 == xi ? wx : 0.f);
      if (cx != 0.f) row += cx * to_f32<TO>(plane[(long)y * W + x]);
    }
    acc += cy * row;
  }
  din[px * ldi + c] = from_f32<T>(acc);
}

// ----------------------------------------------------------------------------------------- OHEM cross entropy
// per-pixel CE (ignore_index -> 0) + the two scalars the OHEM rule needs: stats[0] = #(loss > thresh),
// stats[1] = sum of those losses.   losses.py:32-34.
DEVI unsigned long long loss_to_fix(float v) { return (unsigned long long)((double)fmaxf(v, 0.f) * 4294967296.0); }
DEVI float fix_to_loss(unsigned long long f) { return (float)((double)f * (1.0 / 4294967296.0)); }

template <typename TL>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const TL* logits, const long* labels, float* loss, float* stats, int F,
                                                      long HW, int nc, int ignore_index, float thresh) {
  // grid-stride over the pixels: at one pixel per thread the 4096 workgroups of a 4 x 512 x 512 batch ended in 8192 same-address
  // fp32 atomics (~40 ns each, serialised at the memory side): 114 us for a 62 MB pass
  float hc = 0.f, hs = 0.f;
  const long n = (long)F * HW;
  for (long px = (long)blockIdx.x * 256 + threadIdx.x; px < n; px += (long)gridDim.x * 256) {
    float l = 0.f;
    const long f = px / HW, p = px % HW;
    const long lab = labels[px];
    if (lab != ignore_index) {
      const TL* b = logits + f * nc * HW + p;
      float mx = -3.0e38f;
      for (int c = 0; c < nc; ++c) mx = fmaxf(mx, to_f32<TL>(b[(long)c * HW]));
      float s = 0.f;
      for (int c = 0; c < nc; ++c) s += expf(to_f32<TL>(b[(long)c * HW]) - mx);
      l = mx + logf(s) - to_f32<TL>(b[lab * HW]);
    }
    loss[px] = l;
    if (l > thresh) { hc += 1.f; hs += l; }
  }
  float cnt = wave_sum(hc), sm = wave_sum(hs);
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cnt; red[1][threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // stats[0]: a count (integers < 2^24: exact in fp32, so the atomic sum does not depend on the order); the SUM goes to the 64-bit
    // fixed-point accumulator at stats[2..3] (2^-32 units; losses are >= 0 and n * max loss < 2^32): integer adds are associative,
    // so two runs give the same bits whatever the order the workgroups finish in
    atomicAdd(stats + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd((unsigned long long*)(stats + 2), loss_to_fix(red[1][0] + red[1][1] + red[1][2] + red[1][3]));
  }
}

// dlogits = g * w(px) * (softmax - onehot),  w = sel[1] if loss(px) `>`/`>=` sel[0] else 0   (sel on device: no host sync)
template <typename TL>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const TL* logits, const long* labels, const float* loss, const float* sel,
                                                      const float* gscale, TL* dlogits, int F, long HW, int nc,
                                                      int ignore_index) {
  const long px = (long)blockIdx.x * 256 + threadIdx.x;
  if (px >= (long)F * HW) return;
  const long f = px / HW, p = px % HW;
  const long lab = labels[px];
  const float l = loss[px];
  const bool take = sel[2] != 0.f ? (l >= sel[0]) : (l > sel[0]);
  const float wgt = (take && lab != ignore_index) ? sel[1] * gscale[0] : 0.f;
  const TL* b = logits + f * nc * HW + p;
  TL* d = dlogits + f * nc * HW + p;
  if (wgt == 0.f) {
    for (int c = 0; c < nc; ++c) d[(long)c * HW] = from_f32<TL>(0.f);
    return;
  }
  float mx = -3.0e38f;
  for (int c = 0; c < nc; ++c) mx = fmaxf(mx, to_f32<TL>(b[(long)c * HW]));
  float s = 0.f;
  for (int c = 0; c < nc; ++c) s += expf(to_f32<TL>(b[(long)c * HW]) - mx);
  const float inv = 1.f / s;
  for (int c = 0; c < nc; ++c) {
    const float pr = expf(to_f32<TL>(b[(long)c * HW]) - mx) * inv;
    d[(long)c * HW] = from_f32<TL>(wgt * (pr - (c == lab ? 1.f : 0.f)));
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
#define PACK_OF(dt) ((dt) == 0 ? 8 : 4)
#define DISPATCH_T(dt, CALL_BF16, CALL_F32) do { if ((dt) == 0) { CALL_BF16; } else { CALL_F32; } } while (0)

static int colstats_geometry(int dtype, int M, int C, int groups, int unit_rows, int* cpb_o, int* rpc_o, int* cpg_o) {
  if (C % PACK_OF(dtype) || groups <= 0 || M % groups) return -1401;
  if (unit_rows > 0 && M % ((long)groups * unit_rows)) return -1405;
  const int ppr = C / PACK_OF(dtype), cpb = pick_cpb(ppr);
  const int gr = M / groups, rpc = fit_chunk(reduce_rows_per_chunk(gr, groups * ((ppr + cpb - 1) / cpb), 256 / cpb), unit_rows);
  if (rpc <= 0) return -1405;
  *cpb_o = cpb; *rpc_o = rpc; *cpg_o = (gr + rpc - 1) / rpc;
  return 0;
}
/* floats of scratch stswin_colstats needs for this geometry ([groups * chunks][2][C]); < 0: invalid geometry */
extern "C" long stswin_colstats_scratch(int dtype, int M, int C, int groups, int unit_rows) {
  int cpb, rpc, cpg;
  const int rc = colstats_geometry(dtype, M, C, groups, unit_rows, &cpb, &rpc, &cpg);
  return rc ? rc : (long)groups * cpg * 2 * C;
}

extern "C" int stswin_colstats(int dtype, const void* x, long ldx, float* sum, float* sumsq, int M, int C, int groups,
                               int unit_rows, float* scratch, void* stream) {
  int cpb, rpc, cpg;
  const int rc = colstats_geometry(dtype, M, C, groups, unit_rows, &cpb, &rpc, &cpg);
  if (rc) return rc;
  if (ldx % PACK_OF(dtype)) return -1401;
  if (!scratch) return -1406;
  const int ppr = C / PACK_OF(dtype), gr = M / groups;
  dim3 grid((ppr + cpb - 1) / cpb, groups * cpg);
  hipStream_t st = (hipStream_t)stream;
  if (sumsq)
    DISPATCH_T(dtype, hipLaunchKernelGGL((colstats_kernel<bf16, true>), grid, dim3(256), 0, st, (const bf16*)x, ldx, scratch, C, gr, cpg, rpc, cpb, unit_rows),
               hipLaunchKernelGGL((colstats_kernel<float, true>), grid, dim3(256), 0, st, (const float*)x, ldx, scratch, C, gr, cpg, rpc, cpb, unit_rows));
  else
    DISPATCH_T(dtype, hipLaunchKernelGGL((colstats_kernel<bf16, false>), grid, dim3(256), 0, st, (const bf16*)x, ldx, scratch, C, gr, cpg, rpc, cpb, unit_rows),
               hipLaunchKernelGGL((colstats_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, ldx, scratch, C, gr, cpg, rpc, cpb, unit_rows));
  // sum[g] | sumsq[g] += the group's chunk slabs, in chunk order
  const int rf = stswin_fold_launch(scratch, 2L * C, (long)cpg * 2 * C, cpg, C, sumsq ? 2 : 1, sum, sumsq, nullptr, C, groups, 1, st);
  if (rf) return rf;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bn_finalize(int dtype, const void* x, long ldx, const float* sum, const float* sumsq, float* mean,
                                  float* rstd, float* running_mean, float* running_var, int M, int C, int groups, float eps,
                                  float momentum, int unit_rows, void* stream) {
  if (groups <= 0 || M % groups) return -1401;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((C + 255) / 256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_finalize_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ldx, sum, sumsq, mean, rstd, running_mean, running_var, C, groups, M / groups, eps, momentum, unit_rows),
             hipLaunchKernelGGL(bn_finalize_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, sum, sumsq, mean, rstd, running_mean, running_var, C, groups, M / groups, eps, momentum, unit_rows));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bn_apply(int dtype, const void* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const void* resid, long ldr, void* y, long ldy, int M, int C, int groups,
                               int relu, int unit_rows, void* stream) {
  const int pk = PACK_OF(dtype);
  if (C % pk || ldx % pk || ldy % pk || (resid && ldr % pk) || groups <= 0 || M % groups) return -1402;
  const int ppr = C / pk, cpb = pick_cpb(ppr);
  if (unit_rows > 0 && M % ((long)groups * unit_rows)) return -1405;
  const int gr = M / groups, rpc = fit_chunk(8 * (256 / cpb), unit_rows);
  if (rpc <= 0) return -1405;
  const int cpg = (gr + rpc - 1) / rpc;
  dim3 grid((ppr + cpb - 1) / cpb, groups * cpg);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ldx, mean, rstd, gamma, beta, (const bf16*)resid, ldr, (bf16*)y, ldy, C, gr, cpg, rpc, relu, cpb, unit_rows),
             hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, mean, rstd, gamma, beta, (const float*)resid, ldr, (float*)y, ldy, C, gr, cpg, rpc, relu, cpb, unit_rows));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

static int bn_bwd_geometry(int dtype, int M, int C, int groups, int unit_rows, int* cpb_o, int* rpc_o, int* rpc2_o) {
  const int pk = PACK_OF(dtype);
  if (C % pk || groups <= 0 || M % groups) return -1403;
  if (unit_rows > 0 && M % ((long)groups * unit_rows)) return -1405;
  const int ppr = C / pk, cpb = pick_cpb(ppr), xb = (ppr + cpb - 1) / cpb, nrl = 256 / cpb;
  const int rpc = fit_chunk(bn_bwd_rows(M, xb, nrl, 64) * nrl, unit_rows);
  const int rpc2 = fit_chunk(bn_bwd_rows(M, xb, nrl, 32) * nrl, unit_rows);
  if (rpc <= 0 || rpc2 <= 0) return -1405;
  *cpb_o = cpb; *rpc_o = rpc; *rpc2_o = rpc2;
  return 0;
}
/* floats of scratch the reduce pass of stswin_bn_bwd needs ([groups * chunks][2][C]); < 0: invalid geometry */
extern "C" long stswin_bn_bwd_scratch(int dtype, int M, int C, int groups, int unit_rows) {
  int cpb, rpc, rpc2;
  const int rc = bn_bwd_geometry(dtype, M, C, groups, unit_rows, &cpb, &rpc, &rpc2);
  return rc ? rc : (long)groups * ((M / groups + rpc - 1) / rpc) * 2 * C;
}

extern "C" int stswin_bn_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, const void* y, long ldy,
                             const float* mean, const float* rstd, const float* gamma, const float* beta, float* s1, float* s2,
                             void* dx, long lddx, void* dresid, long lddr, int M, int C, int groups, int relu, int training,
                             int phase, long rows_total, int unit_rows, float* gsum, float* scratch, void* stream) {
  const int pk = PACK_OF(dtype);
  int cpb, rpc, rpc2;
  const int rc = bn_bwd_geometry(dtype, M, C, groups, unit_rows, &cpb, &rpc, &rpc2);
  if (rc) return rc;
  if (ldx % pk || lddy % pk || lddx % pk || (relu && y && ldy % pk)) return -1403;
  if (relu && !y && !beta) return -1404;               // no stored output: the mask is recomputed and needs beta
  if (phase != 2 && !scratch) return -1406;
  const int ppr = C / pk, gr = M / groups;
  const int cpg = (gr + rpc - 1) / rpc;
  dim3 g1((ppr + cpb - 1) / cpb, groups * cpg);
  const int cpg2 = (gr + rpc2 - 1) / rpc2;
  dim3 g2((ppr + cpb - 1) / cpb, groups * cpg2);
  hipStream_t st = (hipStream_t)stream;
  const float inv_n = 1.0f / (float)(rows_total > 0 ? rows_total : gr);
  if (phase != 2) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16>, g1, dim3(256), 0, st, (const bf16*)dy, lddy, (const bf16*)x, ldx, (const bf16*)y, ldy, mean, rstd, scratch, C, gr, cpg, rpc, relu, cpb, gamma, beta, unit_rows),
               hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, g1, dim3(256), 0, st, (const float*)dy, lddy, (const float*)x, ldx, (const float*)y, ldy, mean, rstd, scratch, C, gr, cpg, rpc, relu, cpb, gamma, beta, unit_rows));
    // s1[g] | s2[g] += the group's chunk slabs, in chunk order (deterministic; s1 / s2 arrive zeroed)
    const int rf = stswin_fold_launch(scratch, 2L * C, (long)cpg * 2 * C, cpg, C, 2, s1, s2, nullptr, C, groups, 1, st);
    if (rf) return rf;
  }
  if (phase != 1)
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_dx_kernel<bf16>, g2, dim3(256), 0, st, (const bf16*)dy, lddy, (const bf16*)x, ldx, (const bf16*)y, ldy, mean, rstd, gamma, s1, s2, (bf16*)dx, lddx, (bf16*)dresid, lddr, C, gr, cpg2, rpc2, relu, training, cpb, inv_n, beta, unit_rows, gsum),
             hipLaunchKernelGGL(bn_bwd_dx_kernel<float>, g2, dim3(256), 0, st, (const float*)dy, lddy, (const float*)x, ldx, (const float*)y, ldy, mean, rstd, gamma, s1, s2, (float*)dx, lddx, (float*)dresid, lddr, C, gr, cpg2, rpc2, relu, training, cpb, inv_n, beta, unit_rows, gsum));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// BatchNorm (batch or running statistics of `groups` statistic groups) + ReLU + nn.MaxPool2d(3, 2, 1) in one pass over the
// BatchNorm's input: out [frames*Hp*Wp][C], arg = winning tap per value (first maximum in (ky, kx) scan order, like torch and
// like maxpool_fwd_kernel).  Each candidate is bn_apply_kernel's expression rounded to T before it is compared, so output and
// taps are those of the two-pass form.  torchvision resnet18 conv1 -> bn1 -> relu -> maxpool, resnet.py:98-102.
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const T* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                                                            const float* beta, T* out, long ldo, unsigned char* arg, int frames, int H,
                                                            int W, int Hp, int Wp, int C, int G, int group_rows, int unit) {
  // A thread owns a 2 x 2 block of pooled pixels of one 16-byte channel chunk: their windows share a 5 x 5 patch of the input
  // (25 loads and BatchNorm evaluations instead of 36), scanned row by row so every output meets its taps in (ky, kx) order.
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK, Hb = (Hp + 1) >> 1, Wb = (Wp + 1) >> 1;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)frames * Hb * Wb * ppr) return;
  const int c = (int)(idx % ppr) * PACK;
  const long pb = idx / ppr;
  const int xb = (int)(pb % Wb), yb = (int)((pb / Wb) % Hb);
  const long f = pb / ((long)Wb * Hb);
  const long row0 = f * H * W;
  const int g = unit > 0 ? (int)((row0 / unit) % G) : (int)(row0 / group_rows);
  float sc[8], sh[8], best[4][8];
  unsigned char ba[4][8];
#pragma unroll
  for (int e = 0; e < PACK; ++e) {
    const float rs = rstd[(long)g * C + c + e] * gamma[c + e];
    sc[e] = rs;
    sh[e] = beta[c + e] - mean[(long)g * C + c + e] * rs;
#pragma unroll
    for (int o = 0; o < 4; ++o) { best[o][e] = -3.0e38f; ba[o][e] = 255; }
  }
  const int y0 = 4 * yb - 1, x0 = 4 * xb - 1;           // input coordinates of the patch origin (pooled (2 yb, 2 xb), tap (0, 0))
#pragma unroll
  for (int iy = 0; iy < 5; ++iy) {
    const int yy = y0 + iy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int ix = 0; ix < 5; ++ix) {
      const int xx = x0 + ix;
      if (xx < 0 || xx >= W) continue;
      Vec16<T> v, z;
      v.v = *(const decltype(v.v)*)(x + (row0 + (long)yy * W + xx) * ldx + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e) z.set(e, fmaxf(v.get(e) * sc[e] + sh[e], 0.f));
#pragma unroll
      for (int oy = 0; oy < 2; ++oy) {
        const int ky = iy - 2 * oy;                      // tap row of this input row in output row 2 yb + oy
        if (ky < 0 || ky > 2) continue;
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
          const int kx = ix - 2 * ox;
          if (kx < 0 || kx > 2) continue;
          const int o = oy * 2 + ox;
          const unsigned char t = (unsigned char)(ky * 3 + kx);
#pragma unroll
          for (int e = 0; e < PACK; ++e) {
            const float u = z.get(e);
            if (u > best[o][e] || ba[o][e] == 255) { best[o][e] = u; ba[o][e] = t; }
          }
        }
      }
    }
  }
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
      const int yo = 2 * yb + oy, xo = 2 * xb + ox, o = oy * 2 + ox;
      if (yo >= Hp || xo >= Wp) continue;
      const long px = (f * Hp + yo) * Wp + xo;
      Vec16<T> r;
      unsigned a[2] = {0, 0};
#pragma unroll
      for (int e = 0; e < PACK; ++e) { r.set(e, best[o][e]); a[e >> 2] |= (unsigned)ba[o][e] << (8 * (e & 3)); }
      if (PACK == 8) *(uint2*)(arg + px * C + c) = uint2{a[0], a[1]};
      else *(unsigned*)(arg + px * C + c) = a[0];
      *(decltype(r.v)*)(out + px * ldo + c) = r.v;
    }
}

extern "C" int stswin_bn_relu_pool(int dtype, const void* x, long ldx, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, void* out, long ldo, unsigned char* arg, int frames, int H, int W, int C, int groups,
                                   int unit_rows, void* stream) {
  const int pk = PACK_OF(dtype);
  const long M = (long)frames * H * W;
  if (frames <= 0 || H <= 0 || W <= 0 || C % pk || ldx % pk || ldo % pk || groups <= 0 || M % groups) return -1408;
  if (unit_rows > 0 ? (unit_rows % ((long)H * W) != 0 || M % ((long)groups * unit_rows)) : ((M / groups) % ((long)H * W) != 0)) return -1408;
  const int Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
  const long n = (long)frames * ((Hp + 1) / 2) * ((Wp + 1) / 2) * (C / pk);
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_relu_pool_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)x, ldx, mean, rstd, gamma, beta, (bf16*)out, ldo, arg, frames, H, W, Hp, Wp, C, groups, (int)(M / groups), unit_rows),
             hipLaunchKernelGGL(bn_relu_pool_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, mean, rstd, gamma, beta, (float*)out, ldo, arg, frames, H, W, Hp, Wp, C, groups, (int)(M / groups), unit_rows));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_rows_broadcast(int dtype, const float* v, void* out, long ldo, int M, int C, int groups, float scale,
                                     int accumulate, void* stream) {
  const int pk = PACK_OF(dtype);
  if (C % pk || ldo % pk || groups <= 0 || M % groups) return -1404;
  const long n = (long)M * (C / pk);
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(rows_broadcast_kernel<bf16>, grid, dim3(256), 0, st, v, (bf16*)out, ldo, M, C, M / groups, scale, accumulate),
             hipLaunchKernelGGL(rows_broadcast_kernel<float>, grid, dim3(256), 0, st, v, (float*)out, ldo, M, C, M / groups, scale, accumulate));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bilinear(int dtype, const void* in, long ldi, void* out, long ldo, int frames, int h, int w, int H, int W,
                               int C, int backward, void* stream) {
  /* forward: in [F][h][w][C] -> out [F][H][W][C];  backward: `in` = d(out) [F][H][W][C] (pitch ldi), `out` = d(in) (pitch ldo) */
  const int pk = PACK_OF(dtype);
  if (C % pk || ldi % pk || ldo % pk) return -1405;
  hipStream_t st = (hipStream_t)stream;
  if (!backward) {
    const long n = (long)frames * H * W * (C / pk);
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, ldi, (bf16*)out, ldo, frames, h, w, H, W, C),
               hipLaunchKernelGGL(bilinear_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, frames, h, w, H, W, C));
  } else {
    const long n = (long)frames * h * w * (C / pk);
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL(bilinear_bwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, ldi, (bf16*)out, ldo, frames, h, w, H, W, C),
               hipLaunchKernelGGL(bilinear_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, frames, h, w, H, W, C));
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_logits_upsample(int dtype, const void* tokens, long ldt, void* nchw, int frames, int h, int w, int H,
                                      int W, int nc, int backward, void* stream) {
  /* forward: tokens [F][h][w][nc] -> nchw [F][nc][H][W] (same dtype); backward: nchw = dlogits -> tokens = d(tokens) */
  hipStream_t st = (hipStream_t)stream;
  if (!backward) {
    const long n = (long)frames * H * W;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL((logits_up_fwd_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)tokens, ldt, (bf16*)nchw, frames, h, w, H, W, nc),
               hipLaunchKernelGGL((logits_up_fwd_kernel<float, float>), grid, dim3(256), 0, st, (const float*)tokens, ldt, (float*)nchw, frames, h, w, H, W, nc));
  } else {
    const long n = (long)frames * h * w * nc;
    dim3 grid((unsigned)((n + 255) / 256));
    DISPATCH_T(dtype, hipLaunchKernelGGL((logits_up_bwd_kernel<bf16, bf16>), grid, dim3(256), 0, st, (const bf16*)nchw, (bf16*)tokens, ldt, frames, h, w, H, W, nc),
               hipLaunchKernelGGL((logits_up_bwd_kernel<float, float>), grid, dim3(256), 0, st, (const float*)nchw, (float*)tokens, ldt, frames, h, w, H, W, nc));
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_ce_fwd(int dtype, const void* logits, const long* labels, float* loss, float* stats, int frames, long HW,
                             int nc, int ignore_index, float thresh, void* stream) {
  const long n = (long)frames * HW;
  const long want = (n + 255) / 256;
  dim3 grid((unsigned)(want < 1024 ? want : 1024));        // grid-stride: <= 2048 atomics on the two counters per call
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(ce_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, labels, loss, stats, frames, HW, nc, ignore_index, thresh),
             hipLaunchKernelGGL(ce_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, labels, loss, stats, frames, HW, nc, ignore_index, thresh));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_ce_bwd(int dtype, const void* logits, const long* labels, const float* loss, const float* sel,
                             const float* gscale, void* dlogits, int frames, long HW, int nc, int ignore_index, void* stream) {
  const long n = (long)frames * HW;
  dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_T(dtype, hipLaunchKernelGGL(ce_bwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, labels, loss, sel, gscale, (bf16*)dlogits, frames, HW, nc, ignore_index),
             hipLaunchKernelGGL(ce_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, labels, loss, sel, gscale, (float*)dlogits, frames, HW, nc, ignore_index));
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// ----------------------------------------------------------------------------------------- a15: OHEM selection
// OhemCELoss2D (seg18/utils/losses.py:32-40) sorts all B*H*W pixel losses to read loss[n_min] and, when that is not above
// the threshold, to average the n_min largest.  Here: #(loss > thresh) comes from ce_fwd (the branch test), and the mean
// of the n_min largest from an exact 3-level radix select over the float bits (losses are >= 0, so the bit patterns
// order like the values): per level a 2048-bin (11 + 11 + 10 bits) histogram of counts AND sums, restricted to the
// prefix found by the previous level, gives the k-th largest value and the sum of everything above it:
//   mean = (sum_above + k_rem * kth) / n_min.
// Three passes over the 4 MB loss vector (L2 resident) + one block of bookkeeping instead of a 1 M-element top-k; every
// pass returns at once when the threshold branch is taken (stats[0] > n_min).
struct OhemWork {                       // zeroed by the launcher
  unsigned cnt[3][2048];
  unsigned long long sum[3][2048];      // sums of the losses per bin in 2^-32 fixed point (see loss_to_fix): integer atomics are
                                        // associative, so the selected mean does not depend on the order the workgroups finish in
  unsigned long long sabove[3];         // per level: sum (fixed point) of everything above the level's prefix bin
  unsigned state[3][2];                 // per level: prefix, k (still to take at this level)
};

// k-th largest over `nbins` bins (from the top): bin with count(above) < k <= count(above) + count(bin); 256 threads
__device__ void ohem_scan(const unsigned* cnt, const unsigned long long* sum, int nbins, unsigned k, unsigned* out /* LDS [2]: bin, k_rem */,
                          unsigned long long* out_sum /* LDS [1]: sum above the bin */) {
  __shared__ unsigned pc[256];
  __shared__ unsigned long long ps[256];
  const int t = threadIdx.x, per = nbins / 256;
  unsigned c = 0;
  unsigned long long sm = 0;
  for (int b = 0; b < per; ++b) { c += cnt[nbins - 1 - (t * per + b)]; sm += sum[nbins - 1 - (t * per + b)]; }
  pc[t] = c; ps[t] = sm;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {                 // inclusive scan over threads (thread 0 = highest bins)
    const unsigned c2 = t >= d ? pc[t - d] : 0u;
    const unsigned long long s2 = t >= d ? ps[t - d] : 0ull;
    __syncthreads();
    pc[t] += c2; ps[t] += s2;
    __syncthreads();
  }
  const unsigned incl = pc[t], excl = incl - c;
  if (excl < k && k <= incl) {                         // exactly one thread
    unsigned above = excl;
    unsigned long long sabove = ps[t] - sm;
    for (int b = 0; b < per; ++b) {
      const int bin = nbins - 1 - (t * per + b);
      const unsigned cb = cnt[bin];
      if (above + cb >= k) { out[0] = (unsigned)bin; out[1] = k - above; out_sum[0] = sabove; break; }
      above += cb; sabove += sum[bin];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void ohem_hist_kernel(const float* loss, long n, long n_min, const float* stats, OhemWork* wk, int level) {
  if (stats[0] > (float)n_min) return;                 // threshold branch: the top-k mean is not needed
  __shared__ unsigned hc[2048];
  __shared__ unsigned long long hs[2048];
  __shared__ unsigned sc[2];
  __shared__ unsigned long long ssum[1];
  for (int i = threadIdx.x; i < 2048; i += 256) { hc[i] = 0u; hs[i] = 0ull; }
  unsigned prefix = 0;
  if (level > 0) {
    const unsigned k = level == 1 ? (unsigned)n_min : wk->state[1][1];
    ohem_scan(wk->cnt[level - 1], wk->sum[level - 1], 2048, k, sc, ssum);
    const unsigned up = level == 1 ? 0u : wk->state[1][0];
    prefix = level == 1 ? sc[0] : ((up << 11) | sc[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const unsigned long long sprev = level == 1 ? 0ull : wk->sabove[1];
      wk->state[level][0] = prefix; wk->state[level][1] = sc[1]; wk->sabove[level] = sprev + ssum[0];
    }
  }
  __syncthreads();
  const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
  const unsigned mask = level == 2 ? 1023u : 2047u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = loss[i];
    const unsigned key = __float_as_uint(v) & 0x7fffffffu;
    const bool in = level == 0 || (level == 1 ? (key >> 21) == prefix : (key >> 10) == prefix);
    if (in) { const unsigned b = (key >> shift) & mask; atomicAdd(&hc[b], 1u); atomicAdd(&hs[b], loss_to_fix(v)); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256)
    if (hc[i]) { atomicAdd(&wk->cnt[level][i], hc[i]); atomicAdd(&wk->sum[level][i], hs[i]); }
}

__global__ __launch_bounds__(256) void ohem_final_kernel(long n_min, float thresh, const float* stats, const OhemWork* wk, float* value,
                                                         float* sel) {
  const float n_hard = stats[0];
  if (n_hard > (float)n_min) {                         // loss[n_min] > thresh: mean over loss > thresh
    if (threadIdx.x == 0) {
      const float inv = 1.f / fmaxf(n_hard, 1.f);
      value[0] = fix_to_loss(*(const unsigned long long*)(stats + 2)) * inv; sel[0] = thresh; sel[1] = inv; sel[2] = 0.f;
    }
    return;
  }
  __shared__ unsigned sc[2];
  __shared__ unsigned long long ssum[1];
  ohem_scan(wk->cnt[2], wk->sum[2], 1024, wk->state[2][1], sc, ssum);
  if (threadIdx.x == 0) {
    const float kth = __uint_as_float((wk->state[2][0] << 10) | sc[0]);
    const float total = fix_to_loss(wk->sabove[2] + ssum[0]) + (float)sc[1] * kth;
    value[0] = total / (float)n_min; sel[0] = kth; sel[1] = 1.f / (float)n_min; sel[2] = 1.f;
  }
}

extern "C" int stswin_ohem_select(const float* loss, long n, long n_min, float thresh, const float* stats, void* work,
                                  long work_bytes, float* value, float* sel, void* stream) {
  if (n <= 0 || n_min <= 0 || n_min > n) return -1301;
  if (work_bytes < (long)sizeof(OhemWork)) return -1302;
  hipStream_t st = (hipStream_t)stream;
  stswin_zero_bytes(work, sizeof(OhemWork), st);         // (a kernel, not hipMemsetAsync: see common.h)
  const unsigned grid = (unsigned)max(1L, min(512L, (n + 4095) / 4096));
  for (int level = 0; level < 3; ++level)
    hipLaunchKernelGGL(ohem_hist_kernel, dim3(grid), dim3(256), 0, st, loss, n, n_min, stats, (OhemWork*)work, level);
  hipLaunchKernelGGL(ohem_final_kernel, dim3(1), dim3(256), 0, st, n_min, thresh, stats, (const OhemWork*)work, value, sel);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// ----------------------------------------------------------------------------------------- f4: inference post-processing
// seg18/test.py:153-158: F.interpolate(out, (H, W), bilinear, align_corners=True) -> softmax -> argmax, then Dice / IoU
// (utils/EndoMetric.py).  One pass: interpolate the nc logits of an output pixel in registers, take the arg-max (softmax
// is monotone, so it is skipped), write the label, and - when ground truth is given - count per frame and class
// |gt == c|, |pred == c| and |gt == c and pred == c| (LDS histogram per block, one atomic per non-zero bin).
template <typename TL>
__global__ __launch_bounds__(256) void upsample_argmax_kernel(const TL* logits, unsigned char* labels, const long* gt,
                                                              int* counts, int F, int nc, int h, int w, int H, int W) {
  __shared__ int hist[3 * 64];
  for (int i = threadIdx.x; i < 3 * 64; i += 256) hist[i] = 0;
  __syncthreads();
  const long per_frame = (long)H * W;
  const long blocks_per_frame = (per_frame + 255) / 256;
  const int f = blockIdx.x / blocks_per_frame;
  const long px = (blockIdx.x % blocks_per_frame) * 256 + threadIdx.x;
  if (px < per_frame) {
    const int x = px % W, y = px / W;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const float fy = y * sy, fx = x * sx;
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float wy = fy - y0, wx = fx - x0;
    const TL* b = logits + (long)f * nc * h * w;
    float best = -3.0e38f; int arg = 0;
    for (int c = 0; c < nc; ++c) {
      const TL* pl = b + (long)c * h * w;
      const float v = (1.f - wy) * ((1.f - wx) * to_f32<TL>(pl[y0 * w + x0]) + wx * to_f32<TL>(pl[y0 * w + x1])) +
                      wy * ((1.f - wx) * to_f32<TL>(pl[y1 * w + x0]) + wx * to_f32<TL>(pl[y1 * w + x1]));
      if (v > best) { best = v; arg = c; }
    }
    labels[(long)f * per_frame + px] = (unsigned char)arg;
    if (gt) {
      const int g = (int)gt[(long)f * per_frame + px];
      if (g >= 0 && g < 64) atomicAdd(&hist[g], 1);
      atomicAdd(&hist[64 + arg], 1);
      if (g == arg) atomicAdd(&hist[128 + arg], 1);
    }
  }
  if (gt) {
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * 64; i += 256) {
      const int k = i / 64, c = i % 64;
      if (c < nc && hist[i]) atomicAdd(counts + ((long)f * 3 + k) * nc + c, hist[i]);
    }
  }
}

extern "C" int stswin_upsample_argmax(int dtype, const void* logits, unsigned char* labels, const long* gt, int* counts,
                                      int frames, int nc, int h, int w, int H, int W, void* stream) {
  if (nc <= 0 || nc > 64) return -1406;
  const long bpf = ((long)H * W + 255) / 256;
  dim3 grid((unsigned)(bpf * frames));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(upsample_argmax_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)logits, labels, gt, counts, frames, nc, h, w, H, W);
  else hipLaunchKernelGGL(upsample_argmax_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, labels, gt, counts, frames, nc, h, w, H, W);
  STSWIN_CHECK_LAUNCH();
  return 0;
}


// ----------------------------------------------------------------------------------------- statistics from a GEMM epilogue
// A convolution GEMM launched with STSWIN_GF_CS_PARTIAL | STSWIN_GF_CS_SQ leaves per-128-row-block column sums and sums of
// squares of its output in a table [2 planes][nb = 2*ceil(M/256)][N]; this folds the blocks of every BatchNorm statistic group
// (contiguous groups, or interleaved units of unit_rows rows: group = unit index % G) into sum / sumsq [G][N] for
// stswin_bn_finalize (x = NULL: no pivot) - the colstats pass over the activation (one of the three HBM passes of a train-mode
// BatchNorm forward) disappears.  Grid (N / 64, G); 256 threads = 64 columns x 4 block lanes.
__global__ __launch_bounds__(256) void cs_group_reduce_kernel(const float* tab, int nb, int N, int G, int group_rows, int unit,
                                                               float* sum, float* sumsq) {
  __shared__ float red[2][4][64];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, g = blockIdx.y;
  float a = 0.f, b = 0.f;
  if (c < N) {
    const float* t1 = tab + c;
    const float* t2 = tab + (long)nb * N + c;
    if (unit <= 0) {
      const int b0 = (int)((long)g * group_rows / 128), b1 = (int)((long)(g + 1) * group_rows / 128);
      for (int k = b0 + bl; k < b1; k += 4) { a += t1[(long)k * N]; b += t2[(long)k * N]; }
    } else {
      const int bpu = unit / 128, units = (int)((long)G * group_rows / unit);
      for (int u = g; u < units; u += G)
        for (int k = u * bpu + bl; k < (u + 1) * bpu; k += 4) { a += t1[(long)k * N]; b += t2[(long)k * N]; }
    }
  }
  red[0][bl][cl] = a;
  red[1][bl][cl] = b;
  __syncthreads();
  if (bl == 0 && c < N) {
    sum[(long)g * N + c] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    sumsq[(long)g * N + c] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// BatchNorm statistics straight from a gemm_nt GF_CS_SQ table: group sums over the 128-row blocks, mean / rstd and the
// sequential running-statistic updates (group 0, 1, ... as the per-frame calls of base18.py:86-89 would make them) in ONE
// launch.  1024 threads = 16 columns x 64 block lanes (a 262144-row output has 2048 table rows: 32 independent loads per
// lane and plane); final combination in double.
// PER_GROUP (many groups: the 24 of the batched key views): one workgroup per (16 columns, group) - blockIdx.y = group - writes
// mean / rstd only; bn_running_update_kernel then applies the running-statistic updates in group order.
template <bool PER_GROUP>
__global__ __launch_bounds__(1024) void bn_table_finalize_kernel(const float* tab, int nb, int N, int G, int group_rows, int unit,
                                                                 float* mean, float* rstd, float* running_mean, float* running_var,
                                                                 float eps, float momentum) {
  extern __shared__ float red[];                       // [G][2][16 waves][16 columns]   (PER_GROUP: [1][2][16][16])
  const int cl = threadIdx.x & 15, bl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 16 + cl;
  const bool ok = c < N;
  const float* t1 = tab + (ok ? c : 0);
  const float* t2 = t1 + (long)nb * N;
  const int g_lo = PER_GROUP ? (int)blockIdx.y : 0, g_hi = PER_GROUP ? g_lo + 1 : G;
  for (int g = g_lo; g < g_hi; ++g) {
    float a = 0.f, b = 0.f;
    if (unit <= 0) {
      const int b0 = (int)((long)g * group_rows / 128), b1 = (int)((long)(g + 1) * group_rows / 128);
#pragma unroll 4
      for (int k = b0 + bl; k < b1; k += 64) { a += t1[(long)k * N]; b += t2[(long)k * N]; }
    } else {
      const int bpu = unit / 128, total = (int)((long)group_rows / 128);      // blocks of this group: unit u = g + G * (j / bpu)
#pragma unroll 4
      for (int j = bl; j < total; j += 64) {
        const long k = ((long)g + (long)G * (j / bpu)) * bpu + j % bpu;
        a += t1[k * N];
        b += t2[k * N];
      }
    }
    a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
    a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
    if ((threadIdx.x & 63) < 16) {
      red[(((g - g_lo) * 2 + 0) * 16 + wave) * 16 + cl] = a;
      red[(((g - g_lo) * 2 + 1) * 16 + wave) * 16 + cl] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x >= 16 || !ok) return;
  float rm = (!PER_GROUP && running_mean) ? running_mean[c] : 0.f, rv = (!PER_GROUP && running_var) ? running_var[c] : 0.f;
  for (int g = g_lo; g < g_hi; ++g) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      s += (double)red[(((g - g_lo) * 2 + 0) * 16 + w) * 16 + cl];
      q += (double)red[(((g - g_lo) * 2 + 1) * 16 + w) * 16 + cl];
    }
    const double m = s / group_rows;
    const float var = fmaxf((float)(q / group_rows - m * m), 0.f);
    mean[(long)g * N + c] = (float)m;
    rstd[(long)g * N + c] = rsqrtf(var + eps);
    rm = (1.f - momentum) * rm + momentum * (float)m;
    rv = (1.f - momentum) * rv + momentum * var * ((float)group_rows / (float)max(group_rows - 1, 1));
  }
  if (!PER_GROUP && running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// running statistics from per-group mean / rstd, group 0 first (var = 1 / rstd^2 - eps)
__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* mean, const float* rstd, float* running_mean, float* running_var,
                                                                int N, int G, int group_rows, float eps, float momentum) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float rm = running_mean[c], rv = running_var[c];
  const float unb = (float)group_rows / (float)max(group_rows - 1, 1);
  for (int g = 0; g < G; ++g) {
    const float r = rstd[(long)g * N + c];
    const float var = fmaxf(1.f / (r * r) - eps, 0.f);
    rm = (1.f - momentum) * rm + momentum * mean[(long)g * N + c];
    rv = (1.f - momentum) * rv + momentum * var * unb;
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}

extern "C" int stswin_bn_table_finalize(const float* table, int M, int N, int groups, int unit_rows, float* mean, float* rstd,
                                        float* running_mean, float* running_var, float eps, float momentum, void* stream) {
  if (M <= 0 || N <= 0 || groups <= 0 || groups > 32 || M % groups) return -1413;      // (32 groups = 64 KB of LDS partials)
  const int gr = M / groups;
  if (unit_rows > 0 ? (unit_rows % 256 || M % ((long)groups * unit_rows)) : (gr % 256)) return -1414;   // whole 256-row tiles per group
  const int nb = 2 * ((M + 255) / 256);
  if (groups > 8) {                                    // the one-launch kernel walks its groups serially: 19.8 us for 24 of them
    hipLaunchKernelGGL(bn_table_finalize_kernel<true>, dim3((unsigned)((N + 15) / 16), (unsigned)groups), dim3(1024), (size_t)2 * 16 * 16 * sizeof(float),
                       (hipStream_t)stream, table, nb, N, groups, gr, unit_rows, mean, rstd, running_mean, running_var, eps, momentum);
    if (running_mean && running_var)
      hipLaunchKernelGGL(bn_running_update_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mean, rstd, running_mean,
                         running_var, N, groups, gr, eps, momentum);
  } else {
    hipLaunchKernelGGL(bn_table_finalize_kernel<false>, dim3((unsigned)((N + 15) / 16)), dim3(1024), (size_t)groups * 2 * 16 * 16 * sizeof(float),
                       (hipStream_t)stream, table, nb, N, groups, gr, unit_rows, mean, rstd, running_mean, running_var, eps, momentum);
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_cs_group_reduce(const float* table, int M, int N, int groups, int unit_rows, float* sum, float* sumsq,
                                      void* stream) {
  if (M <= 0 || N <= 0 || groups <= 0 || M % groups) return -1411;
  const int gr = M / groups;
  if (unit_rows > 0 ? (unit_rows % 256 || M % ((long)groups * unit_rows)) : (gr % 256)) return -1412;   // whole 256-row tiles per group
  const int nb = 2 * ((M + 255) / 256);
  hipLaunchKernelGGL(cs_group_reduce_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)groups), dim3(256), 0, (hipStream_t)stream, table, nb,
                     N, groups, gr, unit_rows, sum, sumsq);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
