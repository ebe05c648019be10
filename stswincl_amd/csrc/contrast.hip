// Label-guided pixel-contrastive similarity (reference PixPro_swin_v5.py:71-129, posMask/negMask :48-69).
//
// For one sample n and one key map j the reference materialises logit = q^T k (HW x HW), one_hot-bmm masks and
// their products (25 HW x HW fp32 tensors per loss call).  Here one workgroup owns 128 query pixels of (n, j):
//      S tile = Q[128 x C] . K_j[128 keys x C]^T      (MFMA 16x16x32 bf16 / exact f32 16x16x4, same staging as gemm_nt)
//      pos[i] += sum_p S[i][p] * (lq[i] == lk_j[p]) ;  all[i] += sum_p S[i][p]
// looping over all key tiles, so only 2 floats per (query, key map) ever reach HBM.  The masked means, exp/log and
// the class-count denominators are O(HW) work done by the caller.
#include "common.h"

struct ContrastArgs {
  const void* Q; long ldq;
  const void* K[5]; long ldk;
  const int* lq; const int* lk[5];
  float* pos; float* all;               // [N][HW][5]
  int N, HW, C;
};

template <typename T>
__global__ __launch_bounds__(256) void contrast_fwd_kernel(ContrastArgs p) {
  constexpr int PACK = TT<T>::PACK;
  constexpr int BK = 8 * PACK;
  constexpr int ROWB = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int wr = w >> 1, wc = w & 1;
  const int m0 = blockIdx.x << 7, j = blockIdx.y, n = blockIdx.z;
  const T* Qn = (const T*)p.Q + (long)n * p.HW * p.ldq;
  const T* Kn = (const T*)p.K[j] + (long)n * p.HW * p.ldk;
  const int* lqn = p.lq + (long)n * p.HW;
  const int* lkn = p.lk[j] + (long)n * p.HW;

  const char* zero = (const char*)g_stswin_zero;
  const int rsub = l >> 3, cphys = l & 7, csrc = cphys ^ rsub;
  const char* abase[4]; int astep[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + (w * 4 + i) * 8 + rsub;
    if (gm < p.HW) { abase[i] = (const char*)(Qn + (long)gm * p.ldq) + csrc * 16; astep[i] = BK * sizeof(T); }
    else { abase[i] = zero + cphys * 16; astep[i] = 0; }
  }
  const int kps = p.C / BK;                       // K tiles per key tile
  const int nkt = (p.HW + 127) >> 7;              // key tiles
  const int nst = nkt * kps;
  auto stage = [&](int s, int buf) {
    const int kt = s / kps, kk = s - kt * kps;
    char* Ab = smem + buf * 32768;
    char* Bb = Ab + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(abase[i] + (long)kk * astep[i], Ab + (w * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gk = kt * 128 + (w * 4 + i) * 8 + rsub;
      const char* src = gk < p.HW ? (const char*)(Kn + (long)gk * p.ldk) + csrc * 16 + (long)kk * BK * sizeof(T)
                                  : zero + cphys * 16;
      glds16(src, Bb + (w * 4 + i) * 1024);
    }
  };

  const int fr = l & 15, fq = l >> 4;
  int lrow[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wr * 64 + i * 16 + 4 * fq + r;
      lrow[i * 4 + r] = gm < p.HW ? lqn[gm] : -2147483647;
    }
  float ppos[16], pall[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) { ppos[e] = 0.f; pall[e] = 0.f; }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  for (int s = 0; s < nst; ++s) {
    wait_vm0();
    __syncthreads();
    if (s + 1 < nst) stage(s + 1, (s + 1) & 1);
    const char* Ab = smem + (s & 1) * 32768;
    const char* Bb = Ab + 16384;
    if constexpr (TT<T>::IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr * 64 + i * 16 + fr;
          a[i] = *(const bf16x8*)(Ab + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int row = wc * 64 + jj * 16 + fr;
          b[jj] = *(const bf16x8*)(Bb + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[jj], acc[i][jj], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr * 64 + i * 16 + fr;
          a[i] = *(const float*)(Ab + row * ROWB + ((kk ^ (row & 7)) << 4) + fq * 4);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int row = wc * 64 + jj * 16 + fr;
          b[jj] = *(const float*)(Bb + row * ROWB + ((kk ^ (row & 7)) << 4) + fq * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jj], acc[i][jj], 0, 0, 0);
      }
    }
    if ((s + 1) % kps == 0) {                     // key tile finished: masked row reductions, reset
      const int kt = s / kps;
      int lcol[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int gk = kt * 128 + wc * 64 + jj * 16 + fr;
        lcol[jj] = gk < p.HW ? lkn[gk] : -2147483646;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[i][jj][r];
            pall[i * 4 + r] += v;
            if (lrow[i * 4 + r] == lcol[jj]) ppos[i * 4 + r] += v;
          }
          acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  }
  // reduce over the 16 lanes that share a row, then over the two column waves
#pragma unroll
  for (int e = 0; e < 16; ++e) { ppos[e] = sum16(ppos[e]); pall[e] = sum16(pall[e]); }
  __syncthreads();
  float* red = (float*)smem;                      // [2 (pos/all)][2 (wc)][128 rows]
  if (fr == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 64 + i * 16 + 4 * fq + r;
        red[(0 * 2 + wc) * 128 + row] = ppos[i * 4 + r];
        red[(1 * 2 + wc) * 128 + row] = pall[i * 4 + r];
      }
  }
  __syncthreads();
  if (tid < 128 && m0 + tid < p.HW) {
    const long o = ((long)n * p.HW + m0 + tid) * 5 + j;
    p.pos[o] = red[tid] + red[128 + tid];
    p.all[o] = red[256 + tid] + red[384 + tid];
  }
}

extern "C" int stswin_contrast_fwd(int dtype, const void* Q, long ldq, const void* const* K5, long ldk, const int* lq,
                                   const int* const* lk5, float* pos, float* all, int N, int HW, int C, void* stream) {
  const int bk = dtype == 0 ? 64 : 32;
  if (C % bk || N <= 0 || HW <= 0) return -1501;
  ContrastArgs a;
  a.Q = Q; a.ldq = ldq; a.ldk = ldk; a.lq = lq; a.pos = pos; a.all = all; a.N = N; a.HW = HW; a.C = C;
  for (int j = 0; j < 5; ++j) { a.K[j] = K5[j]; a.lk[j] = lk5[j]; }
  dim3 grid((HW + 127) / 128, 5, N);
  static int once = (int)hipFuncSetAttribute((const void*)contrast_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) |
                    (int)hipFuncSetAttribute((const void*)contrast_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  (void)once;
  if (dtype == 0) hipLaunchKernelGGL(contrast_fwd_kernel<bf16>, grid, dim3(256), 65536, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(contrast_fwd_kernel<float>, grid, dim3(256), 65536, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
