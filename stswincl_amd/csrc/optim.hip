// Multi-tensor optimizer / EMA kernels: one launch updates up to 48 parameter tensors.
// The reference steps Adam (seg18/train_swin.py:122), SGD with momentum (train_CL_ft_mswin_sgd_minput.py:147-162,
// main_pretrain_swinv5.py:37-45 under LARS) and a Python loop of ~370 EMA updates (PixPro_swin_v5.py:258-289).
// Pointer tables travel in the kernel arguments (no device-side table to rebuild when autograd hands out new gradient
// tensors every step); a block finds its (tensor, chunk) by scanning the <= 48 chunk counts.
#include "common.h"

#define MT_MAX 48
#define MT_CHUNK 8192          // elements per block

struct MTArgs {
  float* p[MT_MAX];            // parameter (fp32)            | EMA: key parameter
  const float* g[MT_MAX];      // gradient (fp32)             | EMA: query parameter
  float* m[MT_MAX];            // Adam exp_avg / SGD momentum buffer
  float* v[MT_MAX];            // Adam exp_avg_sq
  int n[MT_MAX];               // elements
  int count;
  float lr, b1, b2, eps, wd, c1, c2;   // c1 = 1 - b1^t, c2 = sqrt(1 - b2^t) (Adam); EMA: b1 = momentum
  int mode;                    // 0 Adam, 1 SGD (momentum b1, dampening 0, nesterov off; first = c1 != 0), 2 EMA,
                               // 3 LARS-scaled SGD: mode 1 with the gradient (after weight decay) times the tensor's trust ratio
  const float* norms;          // mode 3: fp32 [count][2] = sum p^2, sum (g + wd p)^2 per tensor (mt_norms_kernel), or NULL = ratio 1
  float trust;                 // mode 3: trust coefficient (b2 is unused there; eps = LARS eps)
  const float* hyper;          // device fp32 [4] = {lr, c1, c2, EMA momentum} or NULL.  When set, the step-dependent scalars are READ
                               // FROM MEMORY (written by stswin_optim_tick / a stream-ordered fill) instead of the kernel arguments:
                               // a hipGraph replay of the step then advances exactly like eager steps do.
};

// One thread: advance a device-resident step counter and derive the step-dependent scalars from it, in double precision like the
// host expressions they replace.
//   kind 0 (Adam, torch.optim.Adam's bias corrections): t = ++counter; hyper[1] = 1 - a^t; hyper[2] = sqrt(1 - b^t)   (a, b = betas)
//   kind 1 (PixPro key-encoder momentum, PixPro_swin_v5.py:258-262): k = counter++; hyper[3] = 1 - (1 - a) (cos(pi k / b) + 1) / 2
//          (a = base momentum, b = K total steps)
__global__ void optim_tick_kernel(int kind, int* __restrict__ counter, float* __restrict__ hyper, double a, double b) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (kind == 0) {
    const int t = counter[0] + 1;
    counter[0] = t;
    hyper[1] = (float)(1.0 - pow(a, (double)t));
    hyper[2] = (float)sqrt(1.0 - pow(b, (double)t));
  } else {
    const int k = counter[0];
    counter[0] = k + 1;
    hyper[3] = (float)(1.0 - (1.0 - a) * (cos(3.141592653589793 * (double)k / b) + 1.0) / 2.0);
  }
}

extern "C" int stswin_optim_tick(int kind, int* counter, float* hyper, double a, double b, void* stream) {
  if (kind < 0 || kind > 1 || !counter || !hyper) return -1604;
  hipLaunchKernelGGL(optim_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kind, counter, hyper, a, b);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// LARS needs ||p|| and ||g + wd p|| of every tensor before any element moves (contrast/lars.py:131-139): one pass over the
// same (tensor, chunk) grid as the update kernel, a wave-level fold and one fp32 atomic pair per block (<= a few hundred
// adds per address, spread over the launch).
__global__ __launch_bounds__(256) void mt_norms_kernel(MTArgs a, float* __restrict__ norms) {
  int b = blockIdx.x, t = 0;
  for (; t < a.count; ++t) {
    const int nb = (a.n[t] + MT_CHUNK - 1) / MT_CHUNK;
    if (b < nb) break;
    b -= nb;
  }
  if (t >= a.count) return;
  const int n = a.n[t];
  const float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  const int base = b * MT_CHUNK, end = min(n, base + MT_CHUNK);
  float sp = 0.f, sg = 0.f;
  for (int i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
    const int cnt = min(4, end - i);
    if (cnt == 4 && ((((uintptr_t)(p + i)) | ((uintptr_t)(g + i))) & 15) == 0) {
      const f32x4 pv = *(const f32x4*)(p + i), gv = *(const f32x4*)(g + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float gr = gv[e] + a.wd * pv[e]; sp += pv[e] * pv[e]; sg += gr * gr; }
    } else {
      for (int e = 0; e < cnt; ++e) { const float pv = p[i + e], gr = g[i + e] + a.wd * pv; sp += pv * pv; sg += gr * gr; }
    }
  }
  sp = wave_sum(sp);
  sg = wave_sum(sg);
  __shared__ float fold[4][2];
  if ((threadIdx.x & 63) == 0) { fold[threadIdx.x >> 6][0] = sp; fold[threadIdx.x >> 6][1] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {                      // one partial pair per block (no atomics: mt_norms_fold_kernel adds them in block order)
    norms[2 * (long)blockIdx.x] = fold[0][0] + fold[1][0] + fold[2][0] + fold[3][0];
    norms[2 * (long)blockIdx.x + 1] = fold[0][1] + fold[1][1] + fold[2][1] + fold[3][1];
  }
}

// norms[t] = sum of the block partials of tensor t, in block order (one workgroup per tensor; a fixed strided + tree order)
__global__ __launch_bounds__(256) void mt_norms_fold_kernel(MTArgs a, const float* __restrict__ part, float* __restrict__ norms) {
  const int t = blockIdx.x;
  long b0 = 0;
  for (int i = 0; i < t; ++i) b0 += (a.n[i] + MT_CHUNK - 1) / MT_CHUNK;
  const int nb = (a.n[t] + MT_CHUNK - 1) / MT_CHUNK;
  float sp = 0.f, sg = 0.f;
  for (int b = threadIdx.x; b < nb; b += 256) { sp += part[2 * (b0 + b)]; sg += part[2 * (b0 + b) + 1]; }
  sp = wave_sum(sp);
  sg = wave_sum(sg);
  __shared__ float fold[4][2];
  if ((threadIdx.x & 63) == 0) { fold[threadIdx.x >> 6][0] = sp; fold[threadIdx.x >> 6][1] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    norms[2 * t] = fold[0][0] + fold[1][0] + fold[2][0] + fold[3][0];
    norms[2 * t + 1] = fold[0][1] + fold[1][1] + fold[2][1] + fold[3][1];
  }
}

__global__ __launch_bounds__(256) void multi_tensor_kernel(MTArgs a) {
  int b = blockIdx.x, t = 0;
  for (; t < a.count; ++t) {
    const int nb = (a.n[t] + MT_CHUNK - 1) / MT_CHUNK;
    if (b < nb) break;
    b -= nb;
  }
  if (t >= a.count) return;
  const int n = a.n[t];
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const int base = b * MT_CHUNK;
  const int end = min(n, base + MT_CHUNK);
  float lr = a.lr, b1 = a.b1, c1 = a.c1, c2 = a.c2;
  if (a.hyper) {                               // step-dependent scalars from memory (uniform loads)
    if (a.mode == 2) b1 = a.hyper[3];
    else lr = a.hyper[0];
    if (a.mode == 0) { c1 = a.hyper[1]; c2 = a.hyper[2]; }
  }
  float ratio = 1.f;                           // LARS trust ratio of this tensor (lars.py:136-139)
  if (a.mode == 3 && a.norms) {
    const float pn = sqrtf(a.norms[2 * t]), gn = sqrtf(a.norms[2 * t + 1]);
    if (pn > 0.f && gn > 0.f) ratio = a.trust * pn / (gn + a.eps);
  }
  for (int i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
    const int cnt = min(4, end - i);
    float pv[4], gv[4], mv[4], vv[4];
    const bool vec = cnt == 4 && ((((uintptr_t)(p + i)) | ((uintptr_t)(g + i))) & 15) == 0 &&
                     (a.mode == 2 || ((((uintptr_t)(m + i)) & 15) == 0 && (a.mode != 0 || (((uintptr_t)(v + i)) & 15) == 0)));
    if (vec) {
      // nontemporal: 3.5 GB of optimizer state stream through once per step - nothing of it is reused before it has left every cache
      // (round 6: 0.741 -> 0.690 ms for the Adam step of the segmentation model, same bits; tools/adam_nt_ab.py)
      *(f32x4*)pv = __builtin_nontemporal_load((const f32x4*)(p + i));
      *(f32x4*)gv = __builtin_nontemporal_load((const f32x4*)(g + i));
      if (a.mode != 2) *(f32x4*)mv = __builtin_nontemporal_load((const f32x4*)(m + i));
      if (a.mode == 0) *(f32x4*)vv = __builtin_nontemporal_load((const f32x4*)(v + i));
    } else {
      for (int e = 0; e < cnt; ++e) {
        pv[e] = p[i + e]; gv[e] = g[i + e];
        if (a.mode != 2) mv[e] = m[i + e];
        if (a.mode == 0) vv[e] = v[i + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= cnt) break;
      if (a.mode == 0) {                       // torch.optim.Adam (no amsgrad; L2 weight decay added to the gradient)
        float gr = gv[e] + a.wd * pv[e];
        mv[e] = b1 * mv[e] + (1.f - b1) * gr;
        vv[e] = a.b2 * vv[e] + (1.f - a.b2) * gr * gr;
        const float denom = sqrtf(vv[e]) / c2 + a.eps;
        pv[e] -= (lr / c1) * (mv[e] / denom);
      } else if (a.mode == 1 || a.mode == 3) { // torch.optim.SGD with momentum (mode 3: gradient scaled by the trust ratio)
        float gr = (gv[e] + a.wd * pv[e]) * ratio;
        mv[e] = (c1 != 0.f) ? gr : b1 * mv[e] + gr;     // first step: buf = grad
        pv[e] -= lr * (b1 != 0.f ? mv[e] : gr);
      } else {                                 // EMA: key = key * m + query * (1 - m)
        pv[e] = pv[e] * b1 + gv[e] * (1.f - b1);
      }
    }
    if (vec) {
      __builtin_nontemporal_store(*(const f32x4*)pv, (f32x4*)(p + i));
      if (a.mode != 2) __builtin_nontemporal_store(*(const f32x4*)mv, (f32x4*)(m + i));
      if (a.mode == 0) __builtin_nontemporal_store(*(const f32x4*)vv, (f32x4*)(v + i));
    } else {
      for (int e = 0; e < cnt; ++e) {
        p[i + e] = pv[e];
        if (a.mode != 2) m[i + e] = mv[e];
        if (a.mode == 0) v[i + e] = vv[e];
      }
    }
  }
}

// ptrs: host arrays of `count` device pointers (count <= 48 per call; the caller chunks).
static int multi_tensor_impl(int mode, int count, void* const* p, const void* const* g, void* const* m, void* const* v,
                             const int* n, float lr, float b1, float b2, float eps, float wd, float c1, float c2, const float* hyper,
                             void* stream) {
  if (count <= 0) return 0;
  if (count > MT_MAX || mode < 0 || mode > 2) return -1601;
  MTArgs a;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    a.p[i] = (float*)p[i]; a.g[i] = (const float*)g[i];
    a.m[i] = m ? (float*)m[i] : nullptr; a.v[i] = v ? (float*)v[i] : nullptr;
    a.n[i] = n[i];
    blocks += (n[i] + MT_CHUNK - 1) / MT_CHUNK;
  }
  a.count = count; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.wd = wd; a.c1 = c1; a.c2 = c2; a.mode = mode;
  a.norms = nullptr; a.trust = 0.f; a.hyper = hyper;
  hipLaunchKernelGGL(multi_tensor_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_multi_tensor(int mode, int count, void* const* p, const void* const* g, void* const* m, void* const* v,
                                   const int* n, float lr, float b1, float b2, float eps, float wd, float c1, float c2,
                                   void* stream) {
  return multi_tensor_impl(mode, count, p, g, m, v, n, lr, b1, b2, eps, wd, c1, c2, nullptr, stream);
}

// The same update with its step-dependent scalars in device memory (`hyper`, fp32 [4] = {lr, c1, c2, EMA momentum}; see
// stswin_optim_tick): Adam reads lr, c1, c2; SGD reads lr (`first` as in stswin_multi_tensor's c1); EMA reads the momentum.
extern "C" int stswin_multi_tensor_dev(int mode, int count, void* const* p, const void* const* g, void* const* m, void* const* v,
                                       const int* n, const float* hyper, float b1, float b2, float eps, float wd, int first,
                                       void* stream) {
  if (!hyper) return -1604;
  return multi_tensor_impl(mode, count, p, g, m, v, n, 0.f, b1, b2, eps, wd, first ? 1.f : 0.f, 1.f, hyper, stream);
}

// LARS over SGD-momentum (contrast/lars.py:109-152 around torch.optim.SGD, main_pretrain_swinv5.py:37-47) for up to 48 tensors
// of ONE parameter group: g' = g + wd p; adaptive: g' *= trust_coef ||p|| / (||g'|| + eps) when both norms are > 0;
// buf = first ? g' : momentum buf + g'; p -= lr buf.  `norms` = caller-owned fp32 scratch of norms_floats >= 2 * count + 2 * blocks
// floats (blocks = sum_i ceil(n_i / 8192)): [count][2] norms, then one partial pair per block.
static int multi_tensor_lars_impl(int count, void* const* p, const void* const* g, void* const* m, const int* n,
                                  float* norms, long norms_floats, float lr, const float* hyper, float momentum, float wd, float trust_coef,
                                  float eps, int first, int adaptive, void* stream) {
  if (count <= 0) return 0;
  if (count > MT_MAX || (adaptive && !norms)) return -1602;
  MTArgs a;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    a.p[i] = (float*)p[i]; a.g[i] = (const float*)g[i]; a.m[i] = (float*)m[i]; a.v[i] = nullptr; a.n[i] = n[i];
    blocks += (n[i] + MT_CHUNK - 1) / MT_CHUNK;
  }
  a.count = count; a.lr = lr; a.b1 = momentum; a.b2 = 0.f; a.eps = eps; a.wd = wd; a.c1 = first ? 1.f : 0.f; a.c2 = 0.f; a.mode = 3;
  a.norms = adaptive ? norms : nullptr; a.trust = trust_coef; a.hyper = hyper;
  if (adaptive) {
    if (norms_floats < 2 * (long)count + 2 * blocks) return -1603;
    float* part = norms + 2 * count;
    hipLaunchKernelGGL(mt_norms_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, part);
    hipLaunchKernelGGL(mt_norms_fold_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, a, (const float*)part, norms);
  }
  hipLaunchKernelGGL(multi_tensor_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_multi_tensor_lars(int count, void* const* p, const void* const* g, void* const* m, const int* n,
                                        float* norms, long norms_floats, float lr, float momentum, float wd, float trust_coef, float eps,
                                        int first, int adaptive, void* stream) {
  return multi_tensor_lars_impl(count, p, g, m, n, norms, norms_floats, lr, nullptr, momentum, wd, trust_coef, eps, first, adaptive, stream);
}

// ... with the learning rate in device memory (hyper[0]; the scheduler's value reaches it by a stream-ordered fill outside a captured step)
extern "C" int stswin_multi_tensor_lars_dev(int count, void* const* p, const void* const* g, void* const* m, const int* n,
                                            float* norms, long norms_floats, const float* hyper, float momentum, float wd, float trust_coef,
                                            float eps, int first, int adaptive, void* stream) {
  if (!hyper) return -1604;
  return multi_tensor_lars_impl(count, p, g, m, n, norms, norms_floats, 0.f, hyper, momentum, wd, trust_coef, eps, first, adaptive, stream);
}
