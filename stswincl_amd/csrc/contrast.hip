// Label-guided pixel-contrastive similarity (reference PixPro_swin_v5.py:71-129, posMask/negMask :48-69).
//
// For one sample n and one key map j the reference materialises logit = q^T k (HW x HW), one_hot-bmm masks and
// their products (25 HW x HW fp32 tensors per loss call).  Here one workgroup owns 128 query pixels of (n, j):
//      S tile = Q[128 x C] . K_j[128 keys x C]^T      (MFMA 16x16x32 bf16 / exact f32 16x16x4, same staging as gemm_nt)
//      pos[i] += sum_p S[i][p] * (lq[i] == lk_j[p]) ;  all[i] += sum_p S[i][p]
// looping over all key tiles, so only 2 floats per (query, key map) ever reach HBM.  The masked means, exp/log and
// the class-count denominators are O(HW) work done by the caller.
#include "common.h"
#include <type_traits>

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

// =====================================================================================================================
// Bank mode: every query pixel against a bank of key embeddings (all-gathered over samples and ranks), as ONE launch for
// both loss directions and all key maps.
//
//   queries  Q  [M][C], labels lq[M]; the M rows are `q_sets` equal sets (the two loss directions) of `nblk` blocks of
//            q_block rows (a block = one sample in the reference's per-sample mode; nblk = 1 = every query sees the whole
//            bank segment in inter-video mode)
//   bank     Kb [maps][seg][C], labels lb[maps][seg]; query set s uses, as its group g (g < groups), map gmap[s][g]; a query
//            of block b sees rows [b*bank_block, (b+1)*bank_block) of that map's segment
//   outputs  pos[m][g] = sum_p S[m][p] [lq[m] == lb[p]],  all[m][g] = sum_p S[m][p]   over the visible rows p of group g,
//            and (optional) rowmax[m], lse[m] = max / log-sum-exp of inv_tau * S[m][p] over ALL visible rows of all groups.
//
// With q_sets = 2, gmap = {{1,2,3,4,5},{0,2,3,4,5}}, q_block = bank_block = HW this is exactly the reference's two
// regression_loss calls (PixPro_swin_v5.py:594-595, :71-129); with nblk = 1 it is the inter-video bank the reference
// sketches in its unused dist_collect (contrast/util.py:47-58).
//
// Kernel: one workgroup (8 waves as 4 row x 2 column waves of 32 x 64) = 128 query rows x one (group, bank split).  The query
// fragments stay in REGISTERS for the whole kernel (C <= 256: 2 row fragments x 8 k-steps x 4 VGPRs), so only bank rows
// stream through LDS: 128 rows x 64 columns
// (128 bytes per row, chunk ^ row&7 swizzle on the LDS-DMA source address) per stage, 3-stage ring, counted vmcnt + raw
// s_barrier (the copies of the next two stages stay in flight across the barrier).  The bank labels of a tile ride in the
// same ring as a 4-byte LDS-DMA per stage (an ordinary global load inside the loop would make hipcc drain the ring with
// vmcnt(0)).  After the C/64 stages of a bank tile the 64x64 accumulators of a wave are folded: label compare + masked add
// (pos), add (all) and an online max / sum-exp per row.  Partials per (row, group, split) go to a workspace; a small
// combine kernel adds the splits (fixed order: deterministic) and merges the max / sum-exp pairs.
// =====================================================================================================================
#define CB_MAX_GROUPS 8
#define CB_NST 8              // ring depth: 8 x 18 KB = 144 KB.  With 3 stages (one stage of 0.25 us of MFMA work in flight beyond the
                              // one being multiplied) every stage waited ~0.7 us for its copy: 0.95 us per stage, 468 TFLOP/s
struct BankArgs {
  const void* Q; long ldq; const int* lq;
  const void* Kb; long ldk; const int* lb;
  int M, C, q_sets, nblk, q_block, seg, bank_block, groups, splits, chunk;   // chunk = bank rows per split (multiple of 128)
  int gmap[2][CB_MAX_GROUPS];
  float inv_tau;
  float* part;      // [4][M][groups*splits]: pos, all, max, sumexp partials
  int want_lse;
};

template <typename T, int KCH>      // KCH = C / 64 (bf16) resp. C / 32 (f32): stages per bank tile
__global__ __launch_bounds__(512, 2) void contrast_bank_kernel(BankArgs p) {
  constexpr bool BF = TT<T>::IS_BF16;
  constexpr int PACK = TT<T>::PACK;
  constexpr int BK = 8 * PACK;                      // 64 (bf16) / 32 (f32) columns = 128 bytes per bank row and stage
  constexpr int KST = KCH;                          // stages per bank tile = C / BK
  constexpr int FI = 2;                             // 16-row fragments per wave: wave tile 32 x 64, workgroup tile 128 x 128
  constexpr int TM = 4 * 16 * FI;
  constexpr int SUB = BF ? 2 : 8;                   // MFMA k-steps per stage (32 resp. 4 columns each)
  constexpr int NST = CB_NST, STAGE = 16384 + 2048; // bank tile slice + per-wave label slots [8][64] ints
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int wr = w >> 1, wc = w & 1;
  const int fr = l & 15, fq = l >> 4;

  // ---- which rows / which bank range
  const int tiles_per_blk = (p.q_block + TM - 1) / TM;
  int rt = blockIdx.x;
  const int qset = rt / (p.nblk * tiles_per_blk);
  rt -= qset * p.nblk * tiles_per_blk;
  const int blk = rt / tiles_per_blk, t_in = rt - blk * tiles_per_blk;
  const int row_lo = (qset * p.nblk + blk) * p.q_block + t_in * TM;            // first query row of this workgroup
  const int row_hi = (qset * p.nblk + blk + 1) * p.q_block;                    // end of its block
  const int g = blockIdx.y / p.splits, sp = blockIdx.y - g * p.splits;
  const long bank0 = (long)p.gmap[qset][g] * p.seg + (long)blk * p.bank_block;  // first visible bank row of group g
  const int k_lo = sp * p.chunk, k_hi = min(p.bank_block, k_lo + p.chunk);     // this split's rows within the visible range
  const int nkt = k_hi > k_lo ? (k_hi - k_lo + 127) >> 7 : 0;
  const int nst = nkt * KST;

  // ---- query fragments (registers, loaded once) and their labels
  typedef typename std::conditional<BF, bf16x8, float>::type AFrag;
  AFrag a[FI][KST * SUB];
  int lrow[FI * 4];
#pragma unroll
  for (int i = 0; i < FI; ++i) {
    const int gm = row_lo + wr * 16 * FI + i * 16 + fr;
    const bool ok = gm < row_hi;
    const T* src = (const T*)p.Q + (long)(ok ? gm : row_lo) * p.ldq;
#pragma unroll
    for (int kk = 0; kk < KST * SUB; ++kk) {
      if constexpr (BF) {
        a[i][kk] = ok ? *(const bf16x8*)(src + kk * 32 + fq * 8) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      } else {
        a[i][kk] = ok ? src[kk * 4 + fq] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = row_lo + wr * 16 * FI + i * 16 + 4 * fq + r;
      lrow[i * 4 + r] = gr < row_hi ? p.lq[gr] : -2147483647;
    }
  }
  // every ordinary load above must be complete before the first LDS-DMA is issued: from here on vmcnt counts ring copies only
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  const char* zero = (const char*)g_stswin_zero;
  const int rsub = l >> 3, cphys = l & 7, csrc = cphys ^ rsub;
  auto issue = [&](int s) {
    const int kt = s / KST, sk = s - kt * KST;
    char* Bb = smem + (s % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gk = k_lo + kt * 128 + (w * 2 + i) * 8 + rsub;
      const char* src = gk < k_hi ? (const char*)((const T*)p.Kb + (bank0 + gk) * p.ldk) + csrc * 16 + (long)sk * BK * sizeof(T)
                                  : zero + cphys * 16;
      glds16(src, Bb + (w * 2 + i) * 1024);
    }
    // labels of the 64 bank rows this wave's column half covers (rows beyond the range: any valid address, masked later)
    const int gl = min(k_lo + kt * 128 + wc * 64 + l, k_hi - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.lb + bank0 + gl),
                                     (__attribute__((address_space(3))) void*)(Bb + 16384 + w * 256), 4, 0, 0);
  };

  float ppos[FI * 4], pall[FI * 4], pmax[FI * 4], pse[FI * 4];
#pragma unroll
  for (int e = 0; e < FI * 4; ++e) { ppos[e] = 0.f; pall[e] = 0.f; pmax[e] = -3.0e38f; pse[e] = 0.f; }
  f32x4 acc[FI][4];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < NST - 1 && s < nst; ++s) issue(s);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int sk = 0; sk < KST; ++sk) {
      const int s = kt * KST + sk;
      // 2 copies + 1 label copy per wave and stage; the NST - 2 younger stages (fewer at the tail) may stay in flight
      switch (min(NST - 2, nst - 1 - s)) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      }
      __builtin_amdgcn_s_barrier();
      if (s + NST - 1 < nst) issue(s + NST - 1);
      const char* Bb = smem + (s % NST) * STAGE;
#pragma unroll
      for (int kk = 0; kk < SUB; ++kk) {
        if constexpr (BF) {
          bf16x8 b[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int row = wc * 64 + jj * 16 + fr;
            b[jj] = *(const bf16x8*)(Bb + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][sk * SUB + kk], b[jj], acc[i][jj], 0, 0, 0);
        } else {
          float b[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int row = wc * 64 + jj * 16 + fr;
            b[jj] = *(const float*)(Bb + row * 128 + ((kk ^ (row & 7)) << 4) + fq * 4);
          }
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][sk * SUB + kk], b[jj], acc[i][jj], 0, 0, 0);
        }
      }
      if (sk == KST - 1) {                          // bank tile finished: fold the 16 FI x 64 scores of this wave
        // The fold is VALU work beside the MFMAs of the next tile (~8 instructions per score against a budget of 4-5 at C = 256).  A tile
        // that lies wholly inside the split's range - all but the last one - needs no per-column validity selects: FULL drops 8 of the
        // ~35 instructions per 4 scores of the log-sum-exp path (round 6).
        auto fold = [&](auto fullc) __attribute__((always_inline)) {
          constexpr bool FULL = decltype(fullc)::value;
          int lcol[4]; bool cok[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int c = wc * 64 + jj * 16 + fr;
            cok[jj] = FULL || k_lo + kt * 128 + c < k_hi;
            const int lab = *(const int*)(Bb + 16384 + w * 256 + (jj * 16 + fr) * 4);
            lcol[jj] = cok[jj] ? lab : -2147483646;
          }
#pragma unroll
          for (int i = 0; i < FI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = i * 4 + r;
              float v0 = acc[i][0][r], v1 = acc[i][1][r], v2 = acc[i][2][r], v3 = acc[i][3][r];
              pall[e] += (v0 + v1) + (v2 + v3);
              ppos[e] += ((lrow[e] == lcol[0] ? v0 : 0.f) + (lrow[e] == lcol[1] ? v1 : 0.f)) +
                         ((lrow[e] == lcol[2] ? v2 : 0.f) + (lrow[e] == lcol[3] ? v3 : 0.f));
              if (p.want_lse == 2) {
                // unit-norm rows (stswin_contrast_bank_fwd_unit): |score| <= 1, so inv_tau is an upper bound of every scaled score and
                // serves as the FIXED reference point of the sum of exponentials - no running maximum inside the sum, no rescaling
                // of the partial sum per tile: one fma + one v_exp_f32 per score (the online form: 35 VALU per 4 scores, after the
                // MFMAs of the tile and as long as them - 795 -> 609 TFLOP/s; profiles/r04_contrast_kernels.txt).  The row maximum
                // (an output of its own) is two v_max3.
                const float c2 = p.inv_tau * 1.4426950408889634f;
                float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(v0, c2, -c2)), e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(v1, c2, -c2));
                float e2 = __builtin_amdgcn_exp2f(__builtin_fmaf(v2, c2, -c2)), e3 = __builtin_amdgcn_exp2f(__builtin_fmaf(v3, c2, -c2));
                if constexpr (!FULL) {
                  e0 = cok[0] ? e0 : 0.f; e1 = cok[1] ? e1 : 0.f; e2 = cok[2] ? e2 : 0.f; e3 = cok[3] ? e3 : 0.f;
                  v0 = cok[0] ? v0 : -3.0e38f; v1 = cok[1] ? v1 : -3.0e38f; v2 = cok[2] ? v2 : -3.0e38f; v3 = cok[3] ? v3 : -3.0e38f;
                }
                pse[e] += (e0 + e1) + (e2 + e3);
                pmax[e] = fmaxf(fmaxf(pmax[e], fmaxf(v0, v1)), fmaxf(v2, v3));        // (unscaled: x inv_tau at the end)
              } else if (p.want_lse) {
                v0 = cok[0] ? v0 * p.inv_tau : -3.0e38f; v1 = cok[1] ? v1 * p.inv_tau : -3.0e38f;
                v2 = cok[2] ? v2 * p.inv_tau : -3.0e38f; v3 = cok[3] ? v3 * p.inv_tau : -3.0e38f;
                const float mx = fmaxf(fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)), pmax[e]);
                pse[e] = pse[e] * __expf(pmax[e] - mx) + ((__expf(v0 - mx) + __expf(v1 - mx)) + (__expf(v2 - mx) + __expf(v3 - mx)));
                pmax[e] = mx;
              }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
        };
        if (k_lo + kt * 128 + 128 <= k_hi) fold(std::true_type{});
        else fold(std::false_type{});
      }
    }
  }
  // ---- fold over the 16 lanes of a row (columns), then over the two column waves through LDS
  __syncthreads();
  float* red = (float*)smem;                        // [4][2 (wc)][TM]  (8 KB)
#pragma unroll
  for (int e = 0; e < FI * 4; ++e) {
    ppos[e] = sum16(ppos[e]);
    pall[e] = sum16(pall[e]);
    if (p.want_lse == 2) {                          // fixed reference point: plain sums and maxima over the 16 lanes
      pse[e] = sum16(pse[e]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) pmax[e] = fmaxf(pmax[e], __shfl_xor(pmax[e], o));
      pmax[e] = pmax[e] > -1.0e38f ? pmax[e] * p.inv_tau : pmax[e];
    } else if (p.want_lse) {                        // merge (max, sumexp) pairs over the 16 lanes: butterflies inside the row group
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float om = __shfl_xor(pmax[e], o), os = __shfl_xor(pse[e], o);
        const float mx = fmaxf(pmax[e], om);
        pse[e] = pse[e] * __expf(pmax[e] - mx) + os * __expf(om - mx);
        pmax[e] = mx;
      }
    }
  }
  if (fr == 0) {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * 16 * FI + i * 16 + 4 * fq + r, e = i * 4 + r;
        red[(0 * 2 + wc) * TM + row] = ppos[e];
        red[(1 * 2 + wc) * TM + row] = pall[e];
        red[(2 * 2 + wc) * TM + row] = pmax[e];
        red[(3 * 2 + wc) * TM + row] = pse[e];
      }
  }
  __syncthreads();
  if (tid < TM && row_lo + tid < row_hi) {
    const int gs = p.groups * p.splits;
    const long o = (long)(row_lo + tid) * gs + blockIdx.y, plane = (long)p.M * gs;
    p.part[o] = red[tid] + red[TM + tid];
    p.part[plane + o] = red[2 * TM + tid] + red[3 * TM + tid];
    if (p.want_lse == 2) {
      p.part[2 * plane + o] = fmaxf(red[4 * TM + tid], red[5 * TM + tid]);
      p.part[3 * plane + o] = red[6 * TM + tid] + red[7 * TM + tid];           // (both relative to inv_tau)
    } else if (p.want_lse) {
      const float m0 = red[4 * TM + tid], m1 = red[5 * TM + tid], mx = fmaxf(m0, m1);
      p.part[2 * plane + o] = mx;
      p.part[3 * plane + o] = red[6 * TM + tid] * __expf(m0 - mx) + red[7 * TM + tid] * __expf(m1 - mx);
    }
  }
}

// pos / all [M][groups] = sum over splits (fixed order); rowmax / lse [M] over all groups and splits.
__global__ __launch_bounds__(256) void contrast_bank_combine_kernel(const float* part, int M, int groups, int splits, float* pos,
                                                                    float* all, float* rowmax, float* lse, float lse_ref, int fixed_ref) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int gs = groups * splits;
  const long plane = (long)M * gs;
  const float* pp = part + (long)m * gs;
  for (int g = 0; g < groups; ++g) {
    float a = 0.f, b = 0.f;
    for (int s = 0; s < splits; ++s) { a += pp[g * splits + s]; b += pp[plane + g * splits + s]; }
    pos[(long)m * groups + g] = a;
    all[(long)m * groups + g] = b;
  }
  if (rowmax || lse) {
    float mx = -3.0e38f;
    for (int k = 0; k < gs; ++k) mx = fmaxf(mx, pp[2 * plane + k]);
    float se = 0.f;
    if (fixed_ref) { for (int k = 0; k < gs; ++k) se += pp[3 * plane + k]; }          // every partial is relative to lse_ref
    else { for (int k = 0; k < gs; ++k) se += pp[3 * plane + k] * __expf(pp[2 * plane + k] - mx); }
    if (rowmax) rowmax[m] = mx;
    if (lse) lse[m] = (fixed_ref ? lse_ref : mx) + __logf(se);
  }
}

static int contrast_bank_fwd_impl(int dtype, const void* Q, long ldq, const int* lq, int M, int C, int q_sets, int q_block,
                                  const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int groups,
                                  const int* gmap /* host, [q_sets][groups] */, float inv_tau, float* pos, float* all,
                                  float* rowmax, float* lse, float* workspace, long workspace_floats, void* stream, int unit_rows) {
  const int bk = dtype == 0 ? 64 : 32;
  if (M <= 0 || groups <= 0 || seg <= 0) return 0;
  if (C % bk || C > 256 || groups > CB_MAX_GROUPS || q_sets < 1 || q_sets > 2 || q_block <= 0 || bank_block <= 0) return -1511;
  if (M % (q_sets * q_block) || seg % bank_block) return -1512;
  const int nblk = M / (q_sets * q_block);
  if (nblk != seg / bank_block && !(nblk == 1 && bank_block == seg)) return -1513;
  for (int i = 0; i < q_sets * groups; ++i)
    if (gmap[i] < 0 || gmap[i] >= maps) return -1514;
  BankArgs a;
  a.Q = Q; a.ldq = ldq; a.lq = lq; a.Kb = bank; a.ldk = ldb; a.lb = lb;
  a.M = M; a.C = C; a.q_sets = q_sets; a.nblk = nblk; a.q_block = q_block; a.seg = seg; a.bank_block = bank_block; a.groups = groups;
  for (int s = 0; s < 2; ++s)
    for (int g = 0; g < CB_MAX_GROUPS; ++g) a.gmap[s][g] = (s < q_sets && g < groups) ? gmap[s * groups + g] : 0;
  a.inv_tau = inv_tau;
  // fixed-reference form exp(inv_tau (s - 1)) only while its smallest term, exp(-2 inv_tau), stays a normal fp32 number
  // (inv_tau <= 40: e^-80 = 1.8e-35); colder temperatures take the online (running-maximum) form
  a.want_lse = (rowmax || lse) ? (unit_rows && inv_tau > 0.f && inv_tau <= 40.f ? 2 : 1) : 0;
  const int TM = 128;
  const long row_tiles = (long)q_sets * nblk * ((q_block + TM - 1) / TM);
  // bank splits (one 8-wave workgroup per CU): the split count that minimises rounds x (bank tiles per workgroup + ~2 tiles of
  // fixed cost: query fragments, pipeline fill, partial stores), with at least 2 bank tiles of 128 rows per split
  const int tiles_total = (bank_block + 127) / 128;
  const int max_by_rows = tiles_total / 2 > 0 ? tiles_total / 2 : 1;
  int splits = 1;
  long best = -1;
  for (int sp = 1; sp <= max_by_rows && sp <= 64; ++sp) {
    const long wgs = row_tiles * groups * sp, rounds = (wgs + 255) / 256;
    const long cost = rounds * ((tiles_total + sp - 1) / sp + 2);
    if (best < 0 || cost < best) { best = cost; splits = sp; }
  }
  while (splits > 1 && 4L * M * groups * splits > workspace_floats) --splits;
  if (4L * M * groups * splits > workspace_floats || !workspace) return -1515;
  a.splits = splits;
  a.chunk = ((bank_block + splits - 1) / splits + 127) / 128 * 128;
  a.part = workspace;
  const dim3 grid((unsigned)row_tiles, (unsigned)(groups * splits));
  const int lds = CB_NST * (16384 + 2048);
#define CB_LAUNCH(TT_, KCH_)                                                                                                  \
  do {                                                                                                                        \
    static int once = (int)hipFuncSetAttribute((const void*)contrast_bank_kernel<TT_, KCH_>,                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);                              \
    (void)once;                                                                                                               \
    hipLaunchKernelGGL((contrast_bank_kernel<TT_, KCH_>), grid, dim3(512), lds, (hipStream_t)stream, a);                      \
  } while (0)
  const int kch = C / bk;
  if (dtype == 0) {
    if (kch == 1) CB_LAUNCH(bf16, 1); else if (kch == 2) CB_LAUNCH(bf16, 2); else if (kch == 3) CB_LAUNCH(bf16, 3); else CB_LAUNCH(bf16, 4);
  } else {
    if (kch == 1) CB_LAUNCH(float, 1); else if (kch == 2) CB_LAUNCH(float, 2); else if (kch == 4) CB_LAUNCH(float, 4);
    else if (kch == 8) CB_LAUNCH(float, 8); else return -1516;
  }
#undef CB_LAUNCH
  hipLaunchKernelGGL(contrast_bank_combine_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, M, groups, splits, pos, all, rowmax, lse, inv_tau, a.want_lse == 2 ? 1 : 0);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
extern "C" int stswin_contrast_bank_fwd(int dtype, const void* Q, long ldq, const int* lq, int M, int C, int q_sets, int q_block,
                                        const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int groups,
                                        const int* gmap /* host, [q_sets][groups] */, float inv_tau, float* pos, float* all,
                                        float* rowmax, float* lse, float* workspace, long workspace_floats, void* stream) {
  return contrast_bank_fwd_impl(dtype, Q, ldq, lq, M, C, q_sets, q_block, bank, ldb, lb, maps, seg, bank_block, groups, gmap, inv_tau, pos, all,
                                rowmax, lse, workspace, workspace_floats, stream, 0);
}
/* the same for L2-NORMALISED query and bank rows (what ConsistencyLoss feeds it: F.normalize'd embeddings, PixPro_swin_v5.py:_embed):
 * |score| <= 1, so the log-sum-exp uses the fixed reference point inv_tau instead of a running maximum */
extern "C" int stswin_contrast_bank_fwd_unit(int dtype, const void* Q, long ldq, const int* lq, int M, int C, int q_sets, int q_block,
                                             const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block, int groups,
                                             const int* gmap, float inv_tau, float* pos, float* all, float* rowmax, float* lse,
                                             float* workspace, long workspace_floats, void* stream) {
  return contrast_bank_fwd_impl(dtype, Q, ldq, lq, M, C, q_sets, q_block, bank, ldb, lb, maps, seg, bank_block, groups, gmap, inv_tau, pos, all,
                                rowmax, lse, workspace, workspace_floats, stream, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the masked sums with respect to the queries (the keys are no-grad: PixPro_swin_v5.py:366).  The sums are
// linear in the scores, so  d q_m = sum_g  dpos[m][g] * Kcls[g][l_m] + dneg[m][g] * (Ktot[g] - Kcls[g][l_m])   with the per-class
// key sums Kcls[c] = sum_{p visible, l_p = c} k_p and Ktot = sum_p k_p: O(rows x C) instead of a second dense GEMM.  Two
// kernels: class sums of the bank, then the per-row combination.  Empty sets must give EXACTLY zero like the reference's
// masked products (their denominators are 0 + 1e-6, so a 1e-7 residual would become a 0.1 gradient): an empty positive set
// has an untouched (zero) class row; an empty negative set (cnt == visible rows) is skipped by its count.
// ---------------------------------------------------------------------------------------------------------------------
// part [maps * nb][row chunks][ncls + 1][C] fp32: per-workgroup partial class sums (nb = seg / bank_block; slot ncls = all rows);
// the launcher's slab fold adds the row chunks in order into ksum.  No atomics: every (row lane, column pair) thread owns its cells
// of a per-row-lane LDS table [rlanes][ncls + 1][C], the lanes are added in lane order - reproducible bits.
template <typename T>
__global__ __launch_bounds__(256) void contrast_class_sums_kernel(const T* bank, long ldk, const int* lb, int seg, int bank_block,
                                                                  int C, int ncls, int rows_per_wg, float* part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = (float*)smem;                         // [rlanes][ncls + 1][C]
  const int nb = seg / bank_block;
  const int map = blockIdx.y / nb, b = blockIdx.y - map * nb;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(bank_block, r0 + rows_per_wg);
  const int cpairs = C >> 1, rlanes = max(1, 256 / cpairs), cells = (ncls + 1) * C;
  for (int i = threadIdx.x; i < rlanes * cells; i += 256) tab[i] = 0.f;
  __syncthreads();
  const long base = (long)map * seg + (long)b * bank_block;
  const int cp = threadIdx.x % cpairs, rl = threadIdx.x / cpairs;
  if (rl < rlanes) {
    float* mine = tab + (long)rl * cells;
    float t0 = 0.f, t1 = 0.f;
    for (int r = r0 + rl; r < r1; r += rlanes) {
      const int lab = lb[base + r];
      const T* src = bank + (base + r) * ldk + 2 * cp;
      const float v0 = to_f32<T>(src[0]), v1 = to_f32<T>(src[1]);
      t0 += v0; t1 += v1;
      if (lab >= 0 && lab < ncls) { mine[lab * C + 2 * cp] += v0; mine[lab * C + 2 * cp + 1] += v1; }
    }
    mine[ncls * C + 2 * cp] = t0;
    mine[ncls * C + 2 * cp + 1] = t1;
  }
  __syncthreads();
  float* out = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * cells;
  for (int i = threadIdx.x; i < cells; i += 256) {
    float v = 0.f;
    for (int k = 0; k < rlanes; ++k) v += tab[(long)k * cells + i];
    out[i] = v;
  }
}

struct BankDqArgs {
  const float* dpos; const float* dneg;      // [M][groups]
  const float* cnt; int visible;             // cnt[m][g] = visible rows of group g with label lq[m]; visible = rows per group
  const int* lq; const float* ksum; float* dq; long lddq;
  int M, C, q_sets, nblk, q_block, nb, ncls, groups;
  int gmap[2][CB_MAX_GROUPS];
};

__global__ __launch_bounds__(256) void contrast_bank_dq_kernel(BankDqArgs p) {
  const int c4 = p.C >> 2;                          // float4 column groups per row
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int m = (int)(idx / c4), c = (int)(idx - (long)m * c4) * 4;
  if (m >= p.M) return;
  const int qset = m / (p.nblk * p.q_block);
  const int blk = p.nb == 1 ? 0 : (m - qset * p.nblk * p.q_block) / p.q_block;
  const int lab = p.lq[m];
  const bool lab_ok = lab >= 0 && lab < p.ncls;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < p.groups; ++g) {
    const float* tabs = p.ksum + ((long)p.gmap[qset][g] * p.nb + blk) * (p.ncls + 1) * p.C;
    const float dp = p.dpos[(long)m * p.groups + g], dn = p.dneg[(long)m * p.groups + g];
    const f32x4 kc = lab_ok ? *(const f32x4*)(tabs + (long)lab * p.C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    acc += dp * kc;
    if (p.cnt[(long)m * p.groups + g] < (float)p.visible) acc += dn * (*(const f32x4*)(tabs + (long)p.ncls * p.C + c) - kc);
  }
  *(f32x4*)(p.dq + (long)m * p.lddq + c) = acc;
}

static void class_sums_geometry(int maps, int seg, int bank_block, int* rows_per_wg, int* chunks) {
  const int nb = seg / bank_block;
  int r = 512;
  while (r > 64 && (long)((bank_block + r - 1) / r) * maps * nb < 512) r >>= 1;
  *rows_per_wg = r; *chunks = (bank_block + r - 1) / r;
}
extern "C" long stswin_contrast_class_sums_scratch(int maps, int seg, int bank_block, int C, int ncls) {
  if (maps <= 0 || seg <= 0 || bank_block <= 0 || seg % bank_block) return -1521;
  int r, chunks;
  class_sums_geometry(maps, seg, bank_block, &r, &chunks);
  return (long)maps * (seg / bank_block) * chunks * (ncls + 1) * C;
}

extern "C" int stswin_contrast_class_sums(int dtype, const void* bank, long ldb, const int* lb, int maps, int seg, int bank_block,
                                          int C, int ncls, float* ksum, float* scratch, void* stream) {
  if (maps <= 0 || seg <= 0) return 0;
  if (C % 4 || C > 512 || ncls <= 0 || ncls > 63 || bank_block <= 0 || seg % bank_block || !scratch) return -1521;
  const int nb = seg / bank_block;
  int rows_per_wg, chunks;
  class_sums_geometry(maps, seg, bank_block, &rows_per_wg, &chunks);
  const dim3 grid((unsigned)chunks, (unsigned)(maps * nb));
  const int cells = (ncls + 1) * C, rlanes = 256 / (C / 2) > 0 ? 256 / (C / 2) : 1;
  const int lds = rlanes * cells * 4;
  if (lds > 160 * 1024) return -1521;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0) {
    static int once = (int)hipFuncSetAttribute((const void*)contrast_class_sums_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)once;
    hipLaunchKernelGGL(contrast_class_sums_kernel<bf16>, grid, dim3(256), lds, st, (const bf16*)bank, ldb, lb, seg,
                       bank_block, C, ncls, rows_per_wg, scratch);
  } else {
    static int once = (int)hipFuncSetAttribute((const void*)contrast_class_sums_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)once;
    hipLaunchKernelGGL(contrast_class_sums_kernel<float>, grid, dim3(256), lds, st, (const float*)bank, ldb, lb, seg,
                       bank_block, C, ncls, rows_per_wg, scratch);
  }
  // ksum[map][b] = sum of the row-chunk slabs in chunk order (overwrites: no zero fill needed)
  const int rf = stswin_fold_launch(scratch, cells, (long)chunks * cells, chunks, cells, 1, ksum, nullptr, nullptr, cells, maps * nb, 0, st);
  if (rf) return rf;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_contrast_bank_dq(const float* dpos, const float* dneg, const float* cnt, const int* lq, const float* ksum,
                                       float* dq, long lddq, int M, int C, int q_sets, int q_block, int seg, int bank_block, int ncls, int groups,
                                       const int* gmap, void* stream) {
  if (M <= 0) return 0;
  if (C % 4 || groups <= 0 || groups > CB_MAX_GROUPS || q_sets < 1 || q_sets > 2 || M % (q_sets * q_block) || seg % bank_block || lddq % 4)
    return -1531;
  BankDqArgs a;
  a.dpos = dpos; a.dneg = dneg; a.cnt = cnt; a.visible = bank_block; a.lq = lq; a.ksum = ksum; a.dq = dq; a.lddq = lddq;
  a.M = M; a.C = C; a.q_sets = q_sets; a.nblk = M / (q_sets * q_block); a.q_block = q_block; a.nb = seg / bank_block; a.ncls = ncls;
  a.groups = groups;
  if (a.nb != a.nblk && a.nb != 1) return -1532;
  for (int s = 0; s < 2; ++s)
    for (int g = 0; g < CB_MAX_GROUPS; ++g) a.gmap[s][g] = (s < q_sets && g < groups) ? gmap[s * groups + g] : 0;
  const long threads = (long)M * (C >> 2);
  hipLaunchKernelGGL(contrast_bank_dq_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// =====================================================================================================================
// Round 5: the glue of the contrastive step as kernels (it was ~125 torch elementwise launches per step: F.normalize, the
// NCHW <-> token permutes around it, view slices, stack / cat of the loss operands, six F.interpolate + casts, one_hot + sum of the
// label counts, exp / log / mean of the loss and their autograd adds - profiles/r04_contrast_steady_state_kernels.txt).
//   rownorm_scatter  : y = x / max(||x||_2, 1e-12) per token row in fp32 (F.normalize(proj.float(), dim=1), PixPro_swin_v5.py:_embed),
//                      stored in the compute dtype at row (view, sample, pixel) of a view-major matrix: the [maps][N*HW][C] key bank
//                      and the [2][N*HW][C] query matrix of the pair loss are written in place by the encoder passes' last kernel;
//   labels_resize    : the six label maps -> int32 [maps][N*h*w], nearest neighbour (F.interpolate(mode='nearest') + .int());
//   label_hist       : per (map, bank block) class histogram;  count_gather: cnt[m][g] = |{visible rows of group g with label lq[m]}|
//                      (the row sums of posMask, PixPro_swin_v5.py:116-118);
//   pair_loss fwd/bwd: P, N, -log(e^P / (e^P + e^N) + 1e-6), the per-set means and their derivatives (PixPro_swin_v5.py:119-129).
// =====================================================================================================================
struct RowNormArgs {
  const void* X; long ldx;        // [R][C] token rows, clip-major: row = (sample * V + view) * HW + pixel
  void* Y; long ldy;              // [V][b * HW][C] view-major result (compute dtype)
  float* inv;                     // [R] 1 / max(norm, 1e-12), kept for the backward (optional)
  const float* dY; long lddy;     // backward: fp32 gradient of Y (view-major rows)
  void* dX; long lddx;            // backward: gradient of X (X's dtype and row order)
  int R, C, V, HW, b;
};
DEVI long rn_out_row(const RowNormArgs& p, int r) {
  const int c = r / p.HW, px = r - c * p.HW, v = c % p.V, i = c / p.V;
  return (long)v * p.b * p.HW + (long)i * p.HW + px;
}
template <typename T>
__global__ __launch_bounds__(256) void rownorm_scatter_fwd_kernel(RowNormArgs p) {
  const int l = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.R) return;
  const T* x = (const T*)p.X + (long)r * p.ldx;
  T* y = (T*)p.Y + rn_out_row(p, r) * p.ldy;
  float ss = 0.f;
  const int per = p.C >> 6;                         // elements per lane (C % 64 == 0, C <= 1024); the row is re-read in the second sweep
  for (int e = 0; e < per; ++e) { const float v = to_f32<T>(x[l * per + e]); ss += v * v; }   // (from L1: no per-lane array, no scratch)
  ss = wave_sum(ss);
  const float den = fmaxf(sqrtf(ss), 1e-12f);          // x / max(||x||, eps): a division like ATen's, so the bf16 roundings agree
  if (p.inv && l == 0) p.inv[r] = 1.0f / den;
  for (int e = 0; e < per; ++e) y[l * per + e] = from_f32<T>(to_f32<T>(x[l * per + e]) / den);
}
// dx = inv * (dy - y (y . dy)),  y = x * inv in fp32 (autograd of F.normalize for norm > eps)
template <typename T>
__global__ __launch_bounds__(256) void rownorm_scatter_bwd_kernel(RowNormArgs p) {
  const int l = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.R) return;
  const T* x = (const T*)p.X + (long)r * p.ldx;
  const float* dy = p.dY + rn_out_row(p, r) * p.lddy;
  T* dx = (T*)p.dX + (long)r * p.lddx;
  const float inv = p.inv[r];
  const int per = p.C >> 6;
  float dot = 0.f;
  for (int e = 0; e < per; ++e) dot += to_f32<T>(x[l * per + e]) * inv * dy[l * per + e];
  dot = wave_sum(dot);
  for (int e = 0; e < per; ++e) dx[l * per + e] = from_f32<T>(inv * (dy[l * per + e] - to_f32<T>(x[l * per + e]) * inv * dot));
}
extern "C" int stswin_rownorm_scatter(int dtype, const void* X, long ldx, void* Y, long ldy, float* inv, int R, int C, int views, int HW,
                                      int samples, void* stream) {
  if (R <= 0) return 0;
  if (C % 64 || C > 1024 || views < 1 || HW < 1 || samples < 1 || R != views * samples * HW) return -1541;
  RowNormArgs a{X, ldx, Y, ldy, inv, nullptr, 0, nullptr, 0, R, C, views, HW, samples};
  const dim3 grid((unsigned)((R + 3) / 4));
  if (dtype == 0) hipLaunchKernelGGL(rownorm_scatter_fwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(rownorm_scatter_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
extern "C" int stswin_rownorm_scatter_bwd(int dtype, const void* X, long ldx, const float* inv, const float* dY, long lddy, void* dX,
                                          long lddx, int R, int C, int views, int HW, int samples, void* stream) {
  if (R <= 0) return 0;
  if (C % 64 || C > 1024 || views < 1 || HW < 1 || samples < 1 || R != views * samples * HW || !inv) return -1541;
  RowNormArgs a{X, ldx, nullptr, 0, const_cast<float*>(inv), dY, lddy, dX, lddx, R, C, views, HW, samples};
  const dim3 grid((unsigned)((R + 3) / 4));
  if (dtype == 0) hipLaunchKernelGGL(rownorm_scatter_bwd_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(rownorm_scatter_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// ---- labels: up to 8 float label maps [N][1][S][S'] -> int32 [maps][N * h * w], nearest neighbour as F.interpolate(mode='nearest')
// (source index = min(floor(dst * in / out), in - 1), fp32 scale like ATen), then truncation toward zero like .to(torch.int32)
struct LabelArgs { const float* m[8]; int maps, N, Hs, Ws, h, w; int* lb; };
__global__ __launch_bounds__(256) void labels_resize_kernel(LabelArgs p) {
  const long per = (long)p.N * p.h * p.w;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per * p.maps) return;
  const int mp = (int)(idx / per);
  const long r = idx - mp * per;
  const int n = (int)(r / (p.h * p.w)), y = (int)((r / p.w) % p.h), x = (int)(r % p.w);
  const float sy = (float)p.Hs / (float)p.h, sx = (float)p.Ws / (float)p.w;
  const int ys = min((int)floorf((float)y * sy), p.Hs - 1), xs = min((int)floorf((float)x * sx), p.Ws - 1);
  p.lb[idx] = (int)p.m[mp][((long)n * p.Hs + ys) * p.Ws + xs];
}
extern "C" int stswin_labels_resize(const float* const* masks /* host array of `maps` device pointers */, int maps, int N, int Hs, int Ws,
                                    int h, int w, int* lb, void* stream) {
  if (maps < 1 || maps > 8 || N < 1 || Hs < 1 || Ws < 1 || h < 1 || w < 1) return -1542;
  LabelArgs a;
  for (int i = 0; i < 8; ++i) a.m[i] = i < maps ? masks[i] : nullptr;
  a.maps = maps; a.N = N; a.Hs = Hs; a.Ws = Ws; a.h = h; a.w = w; a.lb = lb;
  const long n = (long)maps * N * h * w;
  hipLaunchKernelGGL(labels_resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// hist[map][block][cls] = rows of the block whose label (clamped to [0, ncls - 1] like the reference's one_hot operand) is cls: integer
// LDS atomics - exact, order-free
__global__ __launch_bounds__(256) void label_hist_kernel(const int* lb, int seg, int bank_block, int ncls, int* hist) {
  __shared__ int hs[64];
  const int blk = blockIdx.x, mp = blockIdx.y, nb = seg / bank_block;
  if (threadIdx.x < 64) hs[threadIdx.x] = 0;
  __syncthreads();
  const int* src = lb + (long)mp * seg + (long)blk * bank_block;
  for (int i = threadIdx.x; i < bank_block; i += 256) atomicAdd(&hs[min(max(src[i], 0), ncls - 1)], 1);
  __syncthreads();
  if (threadIdx.x < ncls) hist[((long)mp * nb + blk) * ncls + threadIdx.x] = hs[threadIdx.x];
}
struct CountArgs { const int* lq; const int* hist; float* cnt; int M, q_sets, q_block, nb, ncls, groups; int gmap[2][CB_MAX_GROUPS]; };
__global__ __launch_bounds__(256) void count_gather_kernel(CountArgs p) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const int per_set = p.M / p.q_sets, qs = m / per_set;
  const int blk = p.nb > 1 ? (m - qs * per_set) / p.q_block : 0;
  const int lab = min(max(p.lq[m], 0), p.ncls - 1);
  for (int g = 0; g < p.groups; ++g) p.cnt[(long)m * p.groups + g] = (float)p.hist[((long)p.gmap[qs][g] * p.nb + blk) * p.ncls + lab];
}
extern "C" int stswin_label_counts(const int* lq, const int* lb, int M, int maps, int seg, int q_sets, int q_block, int bank_block, int ncls,
                                   int groups, const int* gmap /* host [q_sets][groups] */, int* hist /* [maps][seg / bank_block][ncls] */,
                                   float* cnt /* [M][groups] */, void* stream) {
  if (M <= 0) return 0;
  if (ncls < 1 || ncls > 64 || groups < 1 || groups > CB_MAX_GROUPS || q_sets < 1 || q_sets > 2 || bank_block < 1 || seg % bank_block ||
      M % q_sets)
    return -1543;
  const int nb = seg / bank_block;
  hipLaunchKernelGGL(label_hist_kernel, dim3((unsigned)nb, (unsigned)maps), dim3(256), 0, (hipStream_t)stream, lb, seg, bank_block, ncls, hist);
  CountArgs a;
  a.lq = lq; a.hist = hist; a.cnt = cnt; a.M = M; a.q_sets = q_sets; a.q_block = q_block; a.nb = nb; a.ncls = ncls; a.groups = groups;
  for (int s = 0; s < 2; ++s)
    for (int g = 0; g < CB_MAX_GROUPS; ++g) {
      a.gmap[s][g] = (s < q_sets && g < groups) ? gmap[s * groups + g] : 0;
      if (a.gmap[s][g] < 0 || a.gmap[s][g] >= maps) return -1543;
    }
  hipLaunchKernelGGL(count_gather_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// ---- the loss of PixPro_swin_v5.py:119-129 on the masked sums: per query row
//      P = sum_g pos_g / (sum_g cnt_g + 1e-6),  N = sum_g (all_g - pos_g) / (visible - cnt_g + 1e-6),  term = -log(e^P / (e^P + e^N) + 1e-6);
//      loss = sum over the query sets of mean(term).  ONE workgroup: every thread adds its strided rows, the 1024 partial sums are added in
//      a fixed tree - deterministic.  The backward writes dpos / dneg (neg = all - pos is formed here, so "dneg" multiplies (all - pos)).
struct PairLossArgs { const float* pos; const float* all; const float* cnt; float* loss; const float* dloss; float* dpos; float* dneg; int M, groups, q_sets; float visible; };
DEVI void pair_terms(const PairLossArgs& p, int m, float& P, float& Nn, float& csum) {
  float ps = 0.f;
  csum = 0.f; Nn = 0.f;
  for (int g = 0; g < p.groups; ++g) {
    const float c = p.cnt[(long)m * p.groups + g], po = p.pos[(long)m * p.groups + g];
    ps += po; csum += c;
    Nn += (p.all[(long)m * p.groups + g] - po) / ((p.visible - c) + 1e-6f);
  }
  P = ps / (csum + 1e-6f);
}
__global__ __launch_bounds__(1024) void pair_loss_fwd_kernel(PairLossArgs p) {
  __shared__ float red[1024];
  const int per_set = p.M / p.q_sets;
  float total = 0.f;
  for (int s = 0; s < p.q_sets; ++s) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < per_set; i += 1024) {
      float P, Nn, cs;
      pair_terms(p, s * per_set + i, P, Nn, cs);
      const float pe = expf(P), ne = expf(Nn);
      acc += -logf(pe / (pe + ne) + 1e-6f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    total += red[0] / (float)per_set;
    __syncthreads();
  }
  if (threadIdx.x == 0) p.loss[0] = total;
}
__global__ __launch_bounds__(256) void pair_loss_bwd_kernel(PairLossArgs p) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const int per_set = p.M / p.q_sets;
  float P, Nn, cs;
  pair_terms(p, m, P, Nn, cs);
  const float pe = expf(P), ne = expf(Nn), s = pe / (pe + ne);
  // term = -log(s + 1e-6), s = pe / (pe + ne): ds/dP = s (1 - s), ds/dN = -s (1 - s)
  const float dterm = p.dloss[0] / (float)per_set;
  const float ds = -dterm / (s + 1e-6f), k = s * (1.0f - s);
  const float dP = ds * k, dN = -ds * k;
  for (int g = 0; g < p.groups; ++g) {
    const float c = p.cnt[(long)m * p.groups + g];
    p.dpos[(long)m * p.groups + g] = dP / (cs + 1e-6f);
    p.dneg[(long)m * p.groups + g] = dN / ((p.visible - c) + 1e-6f);
  }
}
extern "C" int stswin_pair_loss(const float* pos, const float* all, const float* cnt, int M, int groups, int q_sets, int visible, float* loss,
                                void* stream) {
  if (M <= 0 || groups < 1 || q_sets < 1 || M % q_sets) return -1544;
  PairLossArgs a{pos, all, cnt, loss, nullptr, nullptr, nullptr, M, groups, q_sets, (float)visible};
  hipLaunchKernelGGL(pair_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
extern "C" int stswin_pair_loss_bwd(const float* pos, const float* all, const float* cnt, const float* dloss, int M, int groups, int q_sets,
                                    int visible, float* dpos, float* dneg, void* stream) {
  if (M <= 0 || groups < 1 || q_sets < 1 || M % q_sets) return -1544;
  PairLossArgs a{pos, all, cnt, nullptr, dloss, dpos, dneg, M, groups, q_sets, (float)visible};
  hipLaunchKernelGGL(pair_loss_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
