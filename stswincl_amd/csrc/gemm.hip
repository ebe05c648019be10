// Segmented gather GEMMs for gfx950 (MFMA 16x16x32 bf16 / exact-f32 16x16x4).
//
//   gemm_nt : C[M,N] = epilogue( sum_s A[a_rows[s][m], 0:Kseg] . B[n, s*Kseg:(s+1)*Kseg] )
//             One template serves:  plain Linear (S=1, no maps);  the fused roll+window-partition+pair-regroup
//             feeding the QKV projection (a_rows = window gather map, reference swin_512.py:207-218 + :115);
//             the window_reverse+un-roll+residual behind proj (c_rows/r_rows, swin_512.py:224-234);
//             patch-merging's 2x2 gather (S=4, swin_512.py:265-272) and every 3x3 / dilated convolution of the
//             decode head as implicit GEMM (S=9 taps, a_rows = shifted pixel or -1 for padding; ASPP.py:13-20,
//             base18.py:73).
//   gemm_tn : C[Ni,Nj] += sum_m At[at_rows[m], i] * Bt[bt_rows[m], j]   (weight gradients; fp32 atomics, split over m)
//
// Tile 128x128, 256 threads = 2x2 waves of 64x64, BK = 64 (bf16) / 32 (f32): 128-byte LDS rows, written by
// LDS-DMA (global_load_lds_dwordx4) with the XOR swizzle on the source address, double buffered, one barrier
// per K tile.  Epilogue goes through an fp32 LDS image so that stores / residual reads are whole 16-byte
// row pieces (and so that scatter maps cost nothing).
#include "common.h"
#include "stswin_hip.h"
#include <cstdint>
#include <cstdlib>
#include <type_traits>

enum {
  GF_GELU = 1,       // out = gelu(v); C2 (if any) receives v (pre-activation)
  GF_RESID = 2,      // v += R[r_rows[m]][n]
  GF_MUL_DGELU = 4,  // v *= gelu'(R[r_rows[m]][n])           (R = saved pre-activation)
  GF_OUT_F32 = 8,    // C is float regardless of T
  GF_ACCUM = 16,     // C += v (only with GF_OUT_F32)
  GF_RELU = 32,
  GF_MUL_R = 8192,        // v *= R[r_rows[m]][n]                  (R = saved GELU' values, see GF_C2_DGELU)
  GF_C2_DGELU = 16384,    // with GF_GELU: C2 receives gelu'(v) instead of the pre-activation v, so that the backward
                          // epilogue is one multiply (the polynomial Phi is shared with the forward GELU)
  GF_BIG = 128,      // tuning: force the 256x256 4-stage kernel (bf16)
  GF_NOBIG = 256,    // tuning: forbid it
  GF_MID = 512,      // tuning: 256x128x32 tile, 3-stage ring, 2 workgroups per CU
  GF_NOPIPE = 1024,  // tuning: 256x256 ring without the ping-pong schedule
  GF_HALF = 2048,    // tuning: force the 256x128 ping-pong ring
  GF_ROT = 4096,     // tuning: the rotated ping-pong loop of the 256x256 ring (one barrier per stage)   [was GF_NOHALF, never tested anywhere]
  GF_NONARROW = 1 << 26,  // tuning: forbid the 256x64 tile for N <= 64
  GF_STREAM = 1 << 25,    // tuning: persistent streaming 256x256 variant (measured no faster: both wave rows idle through each other's epilogue)
  GF_DUO = 1 << 24,       // tuning: 128x256 tiles, 4 waves, two workgroups per CU
  GF_CS_PARTIAL = 1 << 15,  // colsum is fp32 [2*ceil(M/256)][N]: row b = column sums of output rows 128b..128b+127, stored (not added)
  GF_TAPSKIP = 1 << 29,     // tiled kernels: skip the segments (taps) whose row map has no row inside this tile (costs a 4 us map scan)
  GF_CS_SQ = 1 << 16,       // with GF_CS_PARTIAL: a second table plane (offset 2*ceil(M/256)*N floats) receives the column sums of
                            // SQUARES of the same values - the BatchNorm statistics of a convolution output without a pass over it
  GF_NOSTREAM = 1 << 23,  // tuning: 256x256 ring without the persistent streaming variant
  GF_NOREGEPI = 1 << 22,  // tuning: 256x256 ring with the LDS-staged fp32 epilogue instead of the register epilogue
  GF_M32PP = (int)(1u << 31),   // tuning: the 8-wave ping-pong 256x256 ring on 32x32x16 MFMA tiles
  GF_W4R = 1 << 30,       // tuning: 256x256 ring with the 4-wave register-pipelined main loop (one wave per SIMD, 128x128 wave tiles)
  GF_WAVES4 = 64,    // tuning: 4 waves of 64x64 per tile instead of the default 8 waves of 64x32 (4 waves/SIMD)
  GF_NODEEP = 1 << 27,    // tuning: the 128x64 few-tiles kernel with its double buffer instead of the 4-stage ring
  GF_DEEP = 1 << 28,      // tuning: 3-stage rings for the 128x128 / 256x64 kernels too (one workgroup per CU)
};

// Diagnosis switches of the ring kernels (flag bits 17 stagger, 18 late R, 19 timestamps through p.colsum, 20 no stores, 21 no epilogue;
// tools/gemm_timeline.py, tools/epi_decomp.py, tools/probes/*): compiled into STSWIN_TUNING builds only - the product library neither
// tests them nor lets them through the launcher (a stray bit 19 would turn the column-sum table into a stamp buffer).
#ifdef STSWIN_TUNING
#define STSWIN_DBG(flags, bit) (((flags) & (1 << (bit))) != 0)
#else
#define STSWIN_DBG(flags, bit) false
#endif
enum { GF_DEBUG_BITS = (1 << 17) | (1 << 18) | (1 << 19) | (1 << 20) | (1 << 21) };
#ifdef STSWIN_TUNING
static __device__ int g_dbg_stagger[2] = {8, 0};     // stagger experiment (flag bit 17): phases, 100 MHz ticks per phase (0: (nt + 4) / 8 us)
extern "C" int stswin_debug_set_stagger(int phases, int ticks) {
  const int v[2] = {phases, ticks};
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stagger), v, sizeof(v));
}
#endif

struct GemmNT {
  const void* A; long lda; const int* a_rows;
  const void* B; long ldb;
  void* C; long ldc; const int* c_rows;
  void* C2; long ldc2;
  const float* bias;
  const void* R; long ldr; const int* r_rows;
  int M, N, Kseg, S;
  float scale; int scale_cols;
  int flags;
  float* colsum;                 // optional fp32 [N]: += column sums of the values written to C
  int qsplit;                    // ring kernels, split-K launches (stswin_gemm_nt_splitk): 32-deep stages per blockIdx.y slice, 0 = whole K
  // fp8 (e4m3) q | k | v output (stswin_gemm_nt_qkv_fp8): C is uint8 [M][ldc], C2 the fp32 scale table [M / f8_rows][N / f8_cols];
  // one scale per (f8_rows consecutive rows = one window problem, f8_cols consecutive columns = one head of q, k or v)
  int f8_rows, f8_cols;
  // ring kernels: start-time stagger of the first round of workgroups (see stagger_wait): 100 MHz ticks per phase, 0 = off
  int stagger_ticks;
  int persist;                   // ring kernels: > 0 = persistent form, the grid's workgroups walk the tiles (experiment, STSWIN_NT_PERSIST)
};

// Column sum `v` of output rows [128*blk, 128*blk + 128*nblk) of column gn.  Default: one fp32 atomic per (tile, column) -
// M/128 same-address atomics serialise at ~40 ns each (38 us on a 65536 x 512 output).  GF_CS_PARTIAL: plain stores into
// per-block rows that cs_reduce_kernel sums afterwards (every (block, column) cell is written by exactly one tile).
__device__ __forceinline__ void cs_emit_sq(const GemmNT& p, int blk, int nblk, int gn, float v2) {
  float* t2 = p.colsum + (long)(2 * ((p.M + 255) / 256)) * p.N;
  t2[(long)blk * p.N + gn] = v2;
  if (nblk == 2) t2[(long)(blk + 1) * p.N + gn] = 0.f;
}
__device__ __forceinline__ void cs_emit(const GemmNT& p, int blk, int nblk, int gn, float v) {
  if (p.flags & GF_CS_PARTIAL) {
    p.colsum[(long)blk * p.N + gn] = v;
    if (nblk == 2) p.colsum[(long)(blk + 1) * p.N + gn] = 0.f;
  } else {
    atomicAdd(p.colsum + gn, v);
  }
}

// ---- epilogue of one 8-column piece of output row gm: bias, q-scale, pre-activation copy, GELU, residual / GELU',
// ReLU, store (T or fp32, optional accumulate), column-sum accumulation.  Shared by every gemm_nt variant.
// `rpre` (optional): the 8 residual / pre-activation values of this piece, fetched by epi_prefetch ahead of the LDS
// round trip so that the row loop does not pay one dependent global-load latency per pass.
template <typename T>
DEVI void epi_prefetch(const GemmNT& p, Vec16<T> (&dst)[8 / TT<T>::PACK], int gm, int gn0) {
  constexpr int PACK = TT<T>::PACK;
  const long rrow = p.r_rows ? (long)p.r_rows[gm] : (long)gm;
  const T* src = (const T*)p.R + rrow * p.ldr + gn0;
#pragma unroll
  for (int h = 0; h < 8 / PACK; ++h) dst[h].v = __builtin_nontemporal_load((const decltype(dst[h].v)*)(src + h * PACK));
}

template <typename T>
DEVI void epi_piece(const GemmNT& p, float (&v)[8], const float (&bv)[8], float (&cs)[8], int gm, int gn0, int ncols,
                    const Vec16<T> (&rpre)[8 / TT<T>::PACK], bool use_pre) {
  constexpr int PACK = TT<T>::PACK;
  const bool vec_ok = (ncols == 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] += bv[e];
    if (gn0 + e < p.scale_cols) v[e] *= p.scale;
  }
  const long orow = p.c_rows ? (long)p.c_rows[gm] : (long)gm;
  if (p.C2) {
    float c2v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c2v[e] = (p.flags & GF_C2_DGELU) ? dgelu_t<T>(v[e]) : v[e];
    T* dst = (T*)p.C2 + orow * p.ldc2 + gn0;
    if (vec_ok && (p.ldc2 % PACK) == 0) {
#pragma unroll
      for (int h = 0; h < 8 / PACK; ++h) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < PACK; ++e) o.set(e, c2v[h * PACK + e]);
        __builtin_nontemporal_store(o.v, (decltype(o.v)*)(dst + h * PACK));   // outputs stream past L2 (see readback)
      }
    } else {
      for (int e = 0; e < ncols; ++e) dst[e] = from_f32<T>(c2v[e]);
    }
  }
  if (p.flags & GF_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_t<T>(v[e]);
  }
  if (p.flags & (GF_RESID | GF_MUL_DGELU | GF_MUL_R)) {
    const long rrow = p.r_rows ? (long)p.r_rows[gm] : (long)gm;
    const T* src = (const T*)p.R + rrow * p.ldr + gn0;
    float rv[8];
    if (use_pre) {
#pragma unroll
      for (int h = 0; h < 8 / PACK; ++h)
#pragma unroll
        for (int e = 0; e < PACK; ++e) rv[h * PACK + e] = rpre[h].get(e);
    } else if (vec_ok && (p.ldr % PACK) == 0) {
#pragma unroll
      for (int h = 0; h < 8 / PACK; ++h) {
        Vec16<T> in;
        in.v = *(const decltype(in.v)*)(src + h * PACK);
#pragma unroll
        for (int e = 0; e < PACK; ++e) rv[h * PACK + e] = in.get(e);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) rv[e] = (e < ncols) ? to_f32<T>(src[e]) : 0.f;
    }
    if (p.flags & GF_RESID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    } else if (p.flags & GF_MUL_R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= rv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= dgelu_t<T>(rv[e]);
    }
  }
  if (p.flags & GF_RELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] += v[e];
  if (p.flags & GF_OUT_F32) {
    float* dst = (float*)p.C + orow * p.ldc + gn0;
    if (p.flags & GF_ACCUM) {
      for (int e = 0; e < ncols; ++e) dst[e] += v[e];
    } else if (vec_ok && (p.ldc % 4) == 0) {
      *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
      *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
      for (int e = 0; e < ncols; ++e) dst[e] = v[e];
    }
  } else {
    T* dst = (T*)p.C + orow * p.ldc + gn0;
    if (STSWIN_DBG(p.flags, 20)) { if (v[0] == 123.456f) dst[0] = from_f32<T>(v[1]); }   // DBG: no stores
    else if (vec_ok && (p.ldc % PACK) == 0) {
#pragma unroll
      for (int h = 0; h < 8 / PACK; ++h) {
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < PACK; ++e) o.set(e, v[h * PACK + e]);
        __builtin_nontemporal_store(o.v, (decltype(o.v)*)(dst + h * PACK));   // outputs stream past L2 (see readback)
      }
    } else {
      for (int e = 0; e < ncols; ++e) dst[e] = from_f32<T>(v[e]);
    }
  }
  }

// BM x BN tile (128x128 by default; 256x64 for N <= 64 so that narrow convolutions do not multiply a half-empty tile),
// NW waves as (BM/64) x WN, wave tile 64 x (BN/WN).
template <int N> DEVI void wait_vmcnt() {
#define STSWIN_VMCNT_CASE(k) else if constexpr (N == k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STSWIN_VMCNT_CASE(1) STSWIN_VMCNT_CASE(2) STSWIN_VMCNT_CASE(3) STSWIN_VMCNT_CASE(4) STSWIN_VMCNT_CASE(5) STSWIN_VMCNT_CASE(6)
  STSWIN_VMCNT_CASE(7) STSWIN_VMCNT_CASE(8) STSWIN_VMCNT_CASE(9) STSWIN_VMCNT_CASE(10) STSWIN_VMCNT_CASE(12) STSWIN_VMCNT_CASE(15)
  STSWIN_VMCNT_CASE(16)
  else static_assert(N == 0, "add the literal");
#undef STSWIN_VMCNT_CASE
}

// NST = 2: double buffer, the loads of tile t+1 fly while tile t multiplies (2+ workgroups per CU hide the rest).  NST > 2: a
// ring with prefetch distance NST-1 and counted vmcnt waits behind a raw s_barrier, for the shapes that leave one workgroup
// per CU alone with the memory latency (few tiles, long K: ASPP's dilated 3x3 convolutions at M = 4096).
template <typename T, int NW, int BM = 128, int BN = 128, int NST = 2>
__global__ __launch_bounds__(NW * 64) void gemm_nt_kernel(GemmNT p) {
  constexpr int WM = BM / 64, WN = NW / WM;  // wave rows / columns; wave tile = 64 x (BN / WN)
  constexpr int JN = BN / WN / 16;           // 16-wide fragments per wave along N: 4 (NW=4) or 2 (NW=8)
  constexpr int NIA = BM / 8 / NW, NIB = BN / 8 / NW;   // LDS-DMA instructions per wave per operand per K tile (8 rows each)
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  static_assert(WM * WN == NW && JN >= 1 && NIA >= 1 && NIB >= 1 && BM * BN * 4 <= NST * STAGE && NST >= 2, "tile shape");
  constexpr int PACK = TT<T>::PACK;
  constexpr int BK = 8 * PACK;               // 128-byte rows
  constexpr int ROWB = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int wr = w / WN, wc = w % WN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

  const char* zero = (const char*)g_stswin_zero;
  const int rsub = l >> 3, cphys = l & 7, csrc = cphys ^ rsub;
  const char* abase[NIA]; int astep[NIA];
  const char* bbase[NIB]; int bstep[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int gn = n0 + (w * NIB + i) * 8 + rsub;
    if (gn < p.N) { bbase[i] = (const char*)p.B + ((long)gn * p.ldb) * sizeof(T) + csrc * 16; bstep[i] = BK * sizeof(T); }
    else { bbase[i] = zero + cphys * 16; bstep[i] = 0; }
  }
  auto load_a_bases = [&](int seg) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int gm = m0 + (w * NIA + i) * 8 + rsub;
      long row = -1;
      if (gm < p.M) row = p.a_rows ? (long)p.a_rows[(long)seg * p.M + gm] : (long)gm;
      if (row >= 0) { abase[i] = (const char*)p.A + (row * p.lda) * sizeof(T) + csrc * 16; astep[i] = BK * sizeof(T); }
      else { abase[i] = zero + cphys * 16; astep[i] = 0; }
    }
  };
  const int kps = p.Kseg / BK;               // K tiles per segment
  // Tap skipping (flag GF_TAPSKIP; dilated 3x3 convolutions of ASPP.py:13-20 on small maps): a segment whose row map is -1 for EVERY row of this
  // tile multiplies nothing but padding - with dilation 12 / 18 on a 32 x 32 map the three taps above (below) the image are all
  // padding for 3/8 (1/2) of the row tiles.  The workgroup ORs a validity bit per segment over its rows (one pass over S x BM map
  // entries, through the first word of the still unused ring memory) and walks only the segments that have a row.
  unsigned segmask = 0xFFFFFFFFu;
  if (p.S > 1 && p.S <= 32 && p.a_rows && (p.flags & GF_TAPSKIP)) {
    unsigned* sm = (unsigned*)smem;
    if (tid == 0) *sm = 0u;
    __syncthreads();
    unsigned mine = 0u;
    if (tid < BM && m0 + tid < p.M)
      for (int sg = 0; sg < p.S; ++sg) mine |= (p.a_rows[(long)sg * p.M + m0 + tid] >= 0 ? 1u : 0u) << sg;
    if (mine) atomicOr(sm, mine);
    __syncthreads();
    segmask = *sm;
    if (segmask == 0u) segmask = 1u;        // an all-padding tile: one segment of zeros keeps the pipeline's shape
    __syncthreads();                         // (the word is ring memory: nobody reads it after the first copy lands)
  } else if (p.S < 32) {
    segmask = (1u << p.S) - 1u;
  }
  const int nact = p.S <= 32 ? __builtin_popcount(segmask) : p.S;
  const int nt = nact * kps;
  auto real_seg = [&](int v) -> int {        // v-th segment that has a row
    if (nact == p.S) return v;
    unsigned mk = segmask;
    for (int i = 0; i < v; ++i) mk &= mk - 1u;
    return __builtin_ctz(mk);
  };
  auto stage = [&](int ktg, int kt, int buf) {
    char* Ab = smem + buf * STAGE;
    char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int i = 0; i < NIA; ++i) glds16(abase[i] + (long)kt * astep[i], Ab + (w * NIA + i) * 1024);
#pragma unroll
    for (int i = 0; i < NIB; ++i) glds16(bbase[i] + (long)ktg * bstep[i], Bb + (w * NIB + i) * 1024);
  };

  f32x4 acc[4][JN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int seg = 0, rseg = real_seg(0);
  load_a_bases(rseg);
  auto issue = [&](int tile) {                // K tiles are staged in order: `seg` follows the staging front
    const int nseg = tile / kps;
    if (nseg != seg) { seg = nseg; rseg = real_seg(seg); load_a_bases(rseg); }
    stage(rseg * kps + (tile - nseg * kps), tile - nseg * kps, tile % NST);
  };
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nt) issue(s0);
  const int fr = l & 15, fq = l >> 4;
  for (int ktg = 0; ktg < nt; ++ktg) {
    if constexpr (NST == 2) {
      wait_vm0();
      __syncthreads();
    } else {                                  // tiles ktg+1 .. ktg+NST-2 stay in flight
      if (ktg + NST - 1 <= nt) wait_vmcnt<(NST - 2) * (NIA + NIB)>();
      else wait_vm0();
      __builtin_amdgcn_s_barrier();
    }
    if (ktg + NST - 1 < nt) issue(ktg + NST - 1);
    const char* Ab = smem + (ktg % NST) * STAGE;
    const char* Bb = Ab + A_BYTES;
    if constexpr (TT<T>::IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], b[JN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr * 64 + i * 16 + fr;
          a[i] = *(const bf16x8*)(Ab + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          const int row = wc * (16 * JN) + j * 16 + fr;
          b[j] = *(const bf16x8*)(Bb + row * ROWB + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        float a[4], b[JN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wr * 64 + i * 16 + fr;
          a[i] = *(const float*)(Ab + row * ROWB + ((kk ^ (row & 7)) << 4) + fq * 4);
        }
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          const int row = wc * (16 * JN) + j * 16 + fr;
          b[j] = *(const float*)(Bb + row * ROWB + ((kk ^ (row & 7)) << 4) + fq * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---------------- epilogue: accumulators -> fp32 LDS image -> row-wise 16-byte pieces ----------------
  constexpr int CG = BN / 8;                 // 8-column groups per row
  const int c8 = (tid % CG) * 8;
  const int gn0 = n0 + c8;
  const int ncols = max(0, min(8, p.N - gn0));
  constexpr int RG = NW * 64 / CG;           // rows per pass
  constexpr int NPASS = BM / RG;
  // residual / pre-activation pieces of all passes are requested now and land during the LDS round trip
  const bool pre_r = (p.flags & (GF_RESID | GF_MUL_DGELU | GF_MUL_R)) && ncols == 8 && (p.ldr % PACK) == 0;
  Vec16<T> rp[NPASS][8 / PACK];
  if (pre_r) {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int gm = m0 + pass * RG + tid / CG;
      if (gm < p.M) epi_prefetch<T>(p, rp[pass], gm, gn0);
    }
  }
  __syncthreads();
  float* ct = (float*)smem;                  // [BM][BN]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ct[(wr * 64 + i * 16 + 4 * fq + r) * BN + wc * (16 * JN) + j * 16 + fr] = acc[i][j][r];
  __syncthreads();

  float bv[8], cs[8], cs2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bv[e] = (p.bias && e < ncols) ? p.bias[gn0 + e] : 0.f; cs[e] = 0.f; cs2[e] = 0.f; }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int rr = pass * RG + tid / CG;
    const int gm = m0 + rr;
    if (gm >= p.M || ncols <= 0) break;
    float v[8];
    {
      const f32x4 lo = *(const f32x4*)(ct + rr * BN + c8);
      const f32x4 hi = *(const f32x4*)(ct + rr * BN + c8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    }
    epi_piece<T>(p, v, bv, cs, gm, gn0, ncols, rp[pass], pre_r);
    if (p.flags & GF_CS_SQ) {                            // (epi_piece leaves the stored values in v)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs2[e] += v[e] * v[e];
    }
  }
  if (p.colsum) {                 // fold the 16 row groups through LDS (the C image is no longer needed), 1 atomic / column
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) ct[(tid / CG) * BN + c8 + e] = cs[e];
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < RG; ++k) t += ct[k * BN + tid];
      cs_emit(p, m0 / 128, BM / 128, n0 + tid, t);
    }
    if (p.flags & GF_CS_SQ) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) ct[(tid / CG) * BN + c8 + e] = cs2[e];
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < RG; ++k) t += ct[k * BN + tid];
        cs_emit_sq(p, m0 / 128, BM / 128, n0 + tid, t);
      }
    }
  }
}


// =====================================================================================================
// gemm_nt "big": 256x256x32 tile, 8 waves as 2 (M) x 4 (N) of 128x64, FOUR-stage LDS ring (4 x 32 KB) filled by
// LDS-DMA with a prefetch distance of 3 K tiles: the loads of tiles t+1..t+2 stay in flight across the raw
// s_barrier (counted s_waitcnt vmcnt(8/4/0) instead of draining), so a wave never waits for the tile it has just
// requested.  128 FLOP per L2 byte (the 128x128 tile: 64), 1 workgroup per CU, 2 waves per SIMD.
// LDS rows are 64 bytes (BK = 32 bf16): chunk' = chunk ^ ((4 - ((row >> 2) & 3)) & 3) makes every ds_read_b128
// 16-lane group touch 16 distinct 16-byte slots.  bf16 only (the fp32 parity path uses the 128x128 kernel).
// =====================================================================================================
DEVI int swz64(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// BM x BN x 32 tile, 8 waves as WM x WN, NST-stage ring (prefetch distance NST-1), epilogue in EH row slabs.
//   <256,256,2,4,4>: 128 KB LDS, 1 workgroup / CU (128 FLOP per L2 byte)
//   <256,128,4,2,3>:  72 KB LDS, 2 workgroups / CU (87 FLOP per L2 byte; the neighbour's main loop hides the epilogue)
template <int BM, int BN, int WM, int WN, int NST, int MINW, int PIPE, bool SWAP = false, bool PERSIST = false>   // PIPE: 0 plain, 1 ping-pong (8 waves), 2 register-pipelined (4 waves)
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_nt_ring_kernel(GemmNT p) {
  using T = bf16;
  constexpr int BK = 32, ROWB = 64;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int TM = BM / WM, TN = BN / WN;          // wave tile
  // Fragment geometry.  16x16x32 MFMA (PIPE 0 / 1): fragment (i, j) = rows i*16 + fr, 4 columns j*16 + 4*fq + e per lane (fr = l & 15,
  // fq = l >> 4).  32x32x16 MFMA (PIPE 2): a 32x32 accumulator tile is FOUR such sub-fragments - rows i*32 + fr, columns j*8 + 4*fq + e
  // with fr = l & 31, fq = l >> 5 and j = 4 * tile + r / 4 - so the register epilogue below is written once on (FRH, FCW).
  constexpr bool M32 = PIPE >= 2 && PIPE <= 4;         // PIPE 3: the 8-wave ping-pong schedule on 32x32x16 tiles (tuning)
  constexpr int FRH = M32 ? 32 : 16, FCW = M32 ? 8 : 16;   // rows of a fragment, column step between fragments
  constexpr int FI = TM / FRH, FJ = TN / FCW;         // fragments per wave
  constexpr int NWV = WM * WN, NTHR = NWV * 64;     // waves / threads per workgroup (8 / 512, or 4 / 256)
  constexpr int NIA = BM / 16 / NWV, NIB = BN / 16 / NWV; // LDS-DMA instructions per wave per stage (16 rows each)
  constexpr int PER_STAGE = NIA + NIB;
  constexpr int EROWS = SWAP ? BM : ((NST * STAGE) / (BN * 4) >= BM ? BM : ((NST * STAGE) / (BN * 4) / TM) * TM);   // rows per epilogue slab
  static_assert(SWAP || (EROWS >= TM && BM % EROWS == 0), "epilogue slab");
  static_assert(!SWAP || BM * BN * 2 <= NST * STAGE, "register epilogue image");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  // Persistent form (p.persist, round-5 experiment: STSWIN_NT_PERSIST=1): a grid of <= 256 workgroups, workgroup b walks the tiles b,
  // b + grid, ... one after the other (no cross-tile overlap: epilogue, barrier, next prologue) - what a launch gains when the
  // per-tile workgroup dispatch disappears.  Default: one tile per workgroup (the loop runs once).
  // (a TEMPLATE switch, tuning builds only: as a run-time loop in the product kernel the back edge cost it 26 registers and 280 bytes of
  //  scratch - the per-tile constants become loop-carried)
  const int ntot_ = tiles_m * tiles_n;
  int tix_ = blockIdx.x;
  do {
  int tid_ = threadIdx.x;
  if constexpr (PERSIST) asm volatile("" : "+v"(tid_));   // (lane constants are re-derived per tile, not carried across the tile loop)
  const int tid = tid_, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int t = xcd_remap(tix_, ntot_);
  int tm = t / tiles_n, tn = t % tiles_n;
  // Wide outputs (>= 24 column tiles; none in the training step): row-major order hands the 32 CUs of an XCD one tile row at a
  // time = 1 A panel + 32 B panels through a 4 MB L2 (8192^3: all of B re-streamed per tile row, 4.3 GB, HBM-bound at 1.25
  // PFLOP/s).  Groups of 8 tile rows walked column-major give them 8 + 4 panels instead.
  if (tiles_n >= 24) {
    constexpr int GM = 8;
    const int per_group = GM * tiles_n, grp = t / per_group, first = grp * GM;
    const int rows = min(GM, tiles_m - first), in = t - grp * per_group;
    tm = first + in % rows;
    tn = in / rows;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // Copies are BUFFER-addressed LDS-DMA: a 128-bit descriptor in SGPRs per operand, one 32-bit byte offset per lane and
  // copy instruction (row * pitch + swizzled chunk), and the K position as the instruction's scalar offset.  A stage
  // request is then 4 instructions and a few SALU ops: no per-lane 64-bit pointer arithmetic in the read phase, whose VALU
  // instructions come straight out of the partner wave's MFMA issue slots (tools/probes/pingpong_probe.hip: 0.61 us per
  // stage with an empty read phase, 0.79 with 48 VALU in it; this kernel ran 0.85).  Rows that do not exist (tile edge,
  // convolution padding = map entry -1) get offset 0xFFFFFFFF, which the descriptor's range check turns into zeros.
  const int rsub = l >> 2, cphys = l & 3, csrc = cphys ^ swz64(rsub);
  unsigned aoff[NIA], boff[NIB];
  const long arow_lim = (long)(0xFFFFFF00u / (unsigned)(p.lda * sizeof(T))) - 1;   // rows a 32-bit byte offset can reach
  bool arow_bad = false;                            // a gathered row beyond that: trap at the end (fail loudly, not wrongly)
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int gn = n0 + (w * NIB + i) * 16 + rsub;
    boff[i] = gn < p.N ? (unsigned)gn * (unsigned)(p.ldb * sizeof(T)) + csrc * 16 : 0xFFFFFFFFu;
  }
  auto load_a_bases = [&](int seg) {
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int gm = m0 + (w * NIA + i) * 16 + rsub;
      long row = -1;
      if (gm < p.M) row = p.a_rows ? (long)p.a_rows[(long)seg * p.M + gm] : (long)gm;
      aoff[i] = (row >= 0 && row <= arow_lim) ? (unsigned)row * (unsigned)(p.lda * sizeof(T)) + csrc * 16 : 0xFFFFFFFFu;
      arow_bad |= row > arow_lim;
    }
  };
  const int kps = p.Kseg / BK;
  // split-K launch: workgroup row blockIdx.y owns stages [qbase, qbase + nt) of the K loop and the fp32 slab blockIdx.y of C
  const int qbase = p.qsplit > 0 ? (int)blockIdx.y * p.qsplit : 0;
  const int nt = p.qsplit > 0 ? min(p.qsplit, p.S * kps - qbase) : p.S * kps;
  if (p.qsplit > 0) p.C = (char*)p.C + (long)blockIdx.y * p.M * p.ldc * ((p.flags & GF_OUT_F32) ? 4 : 2);
  int seg = -1;
  auto issue = [&](int q) {
    const int Q = q + qbase;
    int sg = 0, kt = Q;
    if (p.S > 1) { sg = Q / kps; kt = Q - sg * kps; }
    if (sg != seg) { seg = sg; load_a_bases(sg); }
    char* Ab = smem + (q % NST) * STAGE;
    char* Bb = Ab + A_BYTES;
#pragma unroll
    for (int i = 0; i < NIA; ++i) glds16_buf(p.A, aoff[i], kt * (BK * (int)sizeof(T)), Ab + (w * NIA + i) * 1024);
#pragma unroll
    for (int i = 0; i < NIB; ++i) glds16_buf(p.B, boff[i], Q * (BK * (int)sizeof(T)), Bb + (w * NIB + i) * 1024);
  };

  f32x4 acc[M32 ? 1 : FI][M32 ? 1 : FJ];             // zeroed AFTER the prologue copies are requested (below)
  f32x16 acc32[M32 ? FI : 1][M32 ? FJ / 4 : 1];      // (PIPE 2: 32x32 tiles)
  auto accg = [&](int i, int j) __attribute__((always_inline)) -> f32x4 {
    if constexpr (M32) {
      const int o = (j & 3) * 4;
      return (f32x4){acc32[i][j >> 2][o], acc32[i][j >> 2][o + 1], acc32[i][j >> 2][o + 2], acc32[i][j >> 2][o + 3]};
    } else return acc[i][j];
  };
  auto accs = [&](int i, int j, f32x4 v) __attribute__((always_inline)) {
    if constexpr (M32) {
      const int o = (j & 3) * 4;
      acc32[i][j >> 2][o] = v[0]; acc32[i][j >> 2][o + 1] = v[1]; acc32[i][j >> 2][o + 2] = v[2]; acc32[i][j >> 2][o + 3] = v[3];
    } else acc[i][j] = v;
  };

  const int fr = M32 ? (l & 31) : (l & 15), fq = M32 ? (l >> 5) : (l >> 4), fr15 = fr & 15;
  // LDS fragment reads.  16x16x32: lane (fr, fq) reads 16-byte chunk fq of row fr.  32x32x16: chunk 2*kh + fq of row fr for the k half
  // kh; with the chunk ^ swz64(row) layout the second half's address is the first one's ^ 32 (rd_off1).
  const int rd_off = fr * ROWB + ((fq ^ swz64(fr)) << 4);
  const int rd_off1 = rd_off ^ 32;
  // DBG (STSWIN_TUNING builds): p.colsum is a u64 [blocks][16] timestamp buffer (100 MHz clock).  Slots: 0 start, 1 prologue requested,
  // 2 first stage landed, 3 main loop done, 4 ring free (barrier), 5 C image written (activation math), 6 C stores issued, 7 shader-clock
  // cycles between stamps 2 and 3 (with the 100 MHz stamps: the clock the CU really ran the loop at - tools/probes/gemm_clock.py),
  // 8 second-output image written, 9 its stores issued, 10 R tile landed.
  const bool dbg_ts = STSWIN_DBG(p.flags, 19);
#ifdef STSWIN_TUNING
  unsigned long long dbg_c0 = 0;
  auto stamp = [&](int slot) {
    if (dbg_ts && tid == 0) {
      ((unsigned long long*)p.colsum)[(long)blockIdx.x * 16 + slot] = wall_clock64();
      if (slot == 2) dbg_c0 = clock64();
      if (slot == 3) ((unsigned long long*)p.colsum)[(long)blockIdx.x * 16 + 7] = clock64() - dbg_c0;
    }
  };
#else
  auto stamp = [&](int) {};
#endif
  // R tile by LDS-DMA, one 64-row block (= one ring stage buffer of the image layout) at a time; see the main loop's tail
  constexpr bool R_EARLY_OK = SWAP && BM == 256 && BN == 256 && NST == 4 && PER_STAGE == 4 && NTHR == 512;
  bool r_early = false;
  auto issue_r_block = [&](int blk) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int row = (blk * 4 + qq) * 16 + w * 2 + (l >> 5);
      const int col = min(n0 + (((l & 31) ^ (row & 15)) << 3), p.N - 8);
      glds16_nt((const T*)p.R + (long)min(m0 + row, p.M - 1) * p.ldr + col, smem + ((blk * 4 + qq) * 16 + w * 2) * 512);
    }
  };
  // Stagger (flag bit 17): the first round's workgroups start up to 7/8 of a tile time apart (8 phases among the CUs of an
  // XCD).  All 256 CUs otherwise run in lock-step and their C stores (128 KB each: 32 MB per round, more than the L2s hold)
  // leave as one burst; the next tile's first loads queue behind the CU's own share of that HBM write drain.
#ifdef STSWIN_TUNING
  if (STSWIN_DBG(p.flags, 17) && blockIdx.x < 256) {   // (phases / ticks per phase: stswin_debug_set_stagger, tools/stagger_ab2.py)
    const int ph = g_dbg_stagger[0] > 0 ? g_dbg_stagger[0] : 8, tk = g_dbg_stagger[1] > 0 ? g_dbg_stagger[1] : (nt + 4) * 100 / 8;
    const unsigned long long t_go = wall_clock64() + (unsigned long long)(((blockIdx.x >> 3) % ph) * tk);
    while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(8);
  }
#endif
  stamp(0);
  auto epilogue = [&]() {
    if (STSWIN_DBG(p.flags, 21)) {    // DBG: no epilogue at all (keeps the accumulators alive)
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) { const f32x4 q = accg(i, j); t += q[0] + q[1] + q[2] + q[3]; }
      if (t == 123.456f) ((T*)p.C)[tid] = from_f32<T>(t);
      return;
    }
  // ---------------- epilogue: BM / EROWS slabs of [EROWS][BN] fp32 through the ring memory ----------------
    float* ct = (float*)smem;
    constexpr int CG = BN / 8;                        // column groups of 8
    constexpr int RGP = NTHR / CG;                    // rows per pass
    const int c8 = (tid % CG) * 8;
    const int gn0 = n0 + c8;
    const int ncols = max(0, min(8, p.N - gn0));
    float bv[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bv[e] = (p.bias && e < ncols) ? p.bias[gn0 + e] : 0.f; cs[e] = 0.f; }
    constexpr int NPASS = EROWS / RGP;
    const bool pre_r = (p.flags & (GF_RESID | GF_MUL_DGELU | GF_MUL_R)) && ncols == 8 && (p.ldr % 8) == 0;
    Vec16<T> rp[NPASS][1];
#pragma unroll 1
    for (int slab = 0; slab < BM / EROWS; ++slab) {
      if (pre_r) {      // this slab's residual / pre-activation pieces: in flight across the LDS round trip below
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          const int gm = m0 + slab * EROWS + pass * RGP + tid / CG;
          if (gm < p.M) epi_prefetch<T>(p, rp[pass], gm, gn0);
        }
      }
      __syncthreads();
      if (wr * TM >= slab * EROWS && wr * TM < (slab + 1) * EROWS) {
        const int rbase = wr * TM - slab * EROWS;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
          for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { if constexpr (!M32) ct[(rbase + i * 16 + 4 * fq + r) * BN + wc * TN + j * 16 + fr] = acc[i][j][r]; }
      }
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int rr = pass * RGP + tid / CG;
        const int gm = m0 + slab * EROWS + rr;
        if (gm >= p.M || ncols <= 0) break;
        float v[8];
        const f32x4 lo = *(const f32x4*)(ct + rr * BN + c8);
        const f32x4 hi = *(const f32x4*)(ct + rr * BN + c8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
        epi_piece<T>(p, v, bv, cs, gm, gn0, ncols, rp[pass], pre_r);
      }
    }
    if (p.colsum) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) ct[(tid / CG) * BN + c8 + e] = cs[e];
      __syncthreads();
      if (tid < BN && n0 + tid < p.N) {
        float tsum = 0.f;
#pragma unroll
        for (int k = 0; k < RGP; ++k) tsum += ct[k * BN + tid];
        cs_emit(p, m0 / 128, BM / 128, n0 + tid, tsum);
      }
    }

  };

  // ---------------- register epilogue (SWAP kernels): all per-element work happens on the accumulators ----------------
  // Lane (fr, fq) of fragment (i, j) holds C[m0 + wr*TM + i*16 + fr][n0 + wc*TN + j*16 + 4fq + 0..3]: bias, q-scale, GELU,
  // residual / GELU' (8-byte loads of R in the same layout), ReLU and the column sums are applied in registers, the
  // bf16 result goes through ONE [BM][BN] bf16 LDS image (ds_write_b64, chunk ^= row & 15: conflict-free both
  // ways) and leaves as whole 16-byte row pieces.  The LDS-staged fp32 epilogue above spent 7.4 us per 256x256 tile
  // (2 slabs x 128 ds_write_b32 per lane + ~90 VALU per 8-column piece) against 12 us of main loop at K = 512.
  // The per-tile flag combination is resolved ONCE (epi_dispatch below) into a straight-line instantiation of this body:
  // with the flags tested per fragment the 32 fragments of a wave spent 3.6 us of a 24 us tile in scalar branches
  // (tools/gemm_timeline.py).  MD >= 0: compile-time mode bits; MD < 0: generic fallback testing the runtime flags.
  enum { E_BIAS = 1, E_SCALE = 2, E_GELU = 4, E_C2 = 8, E_RESID = 16, E_DGELU = 32, E_COLSUM = 64, E_RELU = 128, E_MULR = 256, E_C2D = 512,
         E_COLSQ = 1024 };
  auto epilogue_body = [&](auto tag) __attribute__((always_inline)) {
    constexpr int MD = decltype(tag)::value;
#define EPI_HAS(bit, rt) (MD >= 0 ? ((MD & (bit)) != 0) : (rt))
    constexpr int PITCH = BN * 2, CPRW = BN / 8, RPP = NTHR / CPRW, NPASS = BM / RPP;
    char* img = smem;
    const bool has_bias = EPI_HAS(E_BIAS, p.bias != nullptr), has_scale = EPI_HAS(E_SCALE, p.scale_cols > 0);
    const bool do_gelu = EPI_HAS(E_GELU, (p.flags & GF_GELU) != 0), has_c2 = EPI_HAS(E_C2, p.C2 != nullptr);
    const bool do_resid = EPI_HAS(E_RESID, (p.flags & GF_RESID) != 0), do_dgelu = EPI_HAS(E_DGELU, (p.flags & GF_MUL_DGELU) != 0);
    const bool do_cs = EPI_HAS(E_COLSUM, p.colsum != nullptr) && !dbg_ts, do_relu = EPI_HAS(E_RELU, (p.flags & GF_RELU) != 0);
    const bool do_mulr = EPI_HAS(E_MULR, (p.flags & GF_MUL_R) != 0), c2_dgelu = EPI_HAS(E_C2D, (p.flags & GF_C2_DGELU) != 0);
    const bool do_sq = do_cs && EPI_HAS(E_COLSQ, (p.flags & GF_CS_SQ) != 0);
    const bool has_r = do_resid || do_dgelu || do_mulr;
    const int colb = n0 + wc * TN + 4 * fq;           // + j*16: first of this lane's 4 columns
    f32x4 bj[FJ];
    if (has_bias) {
#pragma unroll
      for (int j = 0; j < FJ; ++j) bj[j] = *(const f32x4*)(p.bias + min(colb + j * FCW, p.N - 4));   // columns >= N are never stored
    }
    auto pre_act = [&](int i, int j) -> f32x4 {
      f32x4 v = accg(i, j);
      if (has_bias) v += bj[j];
      if (has_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (colb + j * FCW + e < p.scale_cols) v[e] *= p.scale;
      }
      return v;
    };
    auto put = [&](int i, int j, f32x4 v) {
      const int row = wr * TM + i * FRH + fr;
      const int chunk = (wc * TN + j * FCW + 4 * fq) >> 3;
      bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      *(bf16x4*)(img + row * PITCH + (((chunk ^ fr15) & (CPRW - 1)) << 4) + (fq & 1) * 8) = o;
    };
    auto put_pk = [&](int i, int j, bf16x4 o) {
      const int row = wr * TM + i * FRH + fr;
      const int chunk = (wc * TN + j * FCW + 4 * fq) >> 3;
      *(bf16x4*)(img + row * PITCH + (((chunk ^ fr15) & (CPRW - 1)) << 4) + (fq & 1) * 8) = o;
    };
    // fc1 forward (GELU to C, GELU' to C2): one pass computes both from one polynomial, the GELU' tile waits as packed
    // bf16 in the registers the accumulators vacate and goes through the image after C has left
    constexpr bool FUSED_C2D = MD >= 0 && (MD & E_C2) && (MD & E_C2D) && (MD & E_GELU) && !(MD & (E_RESID | E_DGELU | E_MULR));
    bf16x4 dpk[FUSED_C2D ? FI : 1][FUSED_C2D ? FJ : 1];
    // output rows of this thread's readback pieces (c_rows is a scatter map): fetched before the LDS round trip
    const int rb_row = tid / CPRW, rb_chunk = tid % CPRW;
    const bool rb_col_ok = n0 + rb_chunk * 8 < p.N;
    // The stores are non-temporal: a round of tiles writes 32 MB of C, as much as all eight L2s hold, and with ordinary
    // stores that evicted the A / B panels the next tiles start on - their first 8 stages ran 4.3 us slower (7.2 vs
    // 4.1 + 4.2 vs 2.95 us, tools/probes/loop_stamps.py; -15..25 % on the K = 512 shapes, tools/epi_decomp.py).
    // Two bodies: with a scatter map the output rows are fetched (as one batch) before the LDS round trip; without one
    // there must be NO load in the path - hipcc puts the wait of a conditional load at the join, executed either way,
    // and behind the stores of a first output (stores count in vmcnt) that wait sits out their whole drain.
    auto readback = [&](void* Cout, long ldo) {
      if (p.c_rows) {
        int orow[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int gm = m0 + ps * RPP + rb_row;
          orow[ps] = (gm < p.M && rb_col_ok) ? p.c_rows[gm] : -1;
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int row = ps * RPP + rb_row;
          if (orow[ps] >= 0) {
            const bf16x8 val = *(const bf16x8*)(img + row * PITCH + (((rb_chunk ^ (row & 15)) & (CPRW - 1)) << 4));
            __builtin_nontemporal_store(val, (bf16x8*)((T*)Cout + (long)orow[ps] * ldo + n0 + rb_chunk * 8));
          }
        }
      } else {
        __syncthreads();
#ifdef STSWIN_DEBUG_LOOP_STAMPS                        // diagnosis build: bit 20 really drops the stores of this path
        if (STSWIN_DBG(p.flags, 20)) return;
#endif
        T* cbase = (T*)Cout + (long)(m0 + rb_row) * ldo + n0 + rb_chunk * 8;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int row = ps * RPP + rb_row;
          if (m0 + row < p.M && rb_col_ok) {
            const bf16x8 val = *(const bf16x8*)(img + row * PITCH + (((rb_chunk ^ (row & 15)) & (CPRW - 1)) << 4));
            __builtin_nontemporal_store(val, (bf16x8*)(cbase + (long)(ps * RPP) * ldo));
          }
        }
      }
    };
    stamp(3);
    __syncthreads();                                   // every wave is done with the ring stages
    stamp(4);
    if (has_c2 && !FUSED_C2D) {                        // pre-activation copy (fc1 forward keeps it for GELU')
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          f32x4 v = pre_act(i, j);
          if (c2_dgelu) {                              // keep gelu'(pre) for the backward multiply instead of pre itself
            const f32x2 lo = dgelu_fast2((f32x2){v[0], v[1]}), hi = dgelu_fast2((f32x2){v[2], v[3]});
            v = (f32x4){lo[0], lo[1], hi[0], hi[1]};
          }
          put(i, j, v);
          __builtin_amdgcn_sched_barrier(0);
        }
      stamp(8);
      readback(p.C2, p.ldc2);
      stamp(9);
      __syncthreads();
      // launder the bias registers: otherwise hipcc keeps all 128 bias-added values of the pass above for the pass
      // below (common subexpression) and spills them
      if (has_bias) {
#pragma unroll
        for (int j = 0; j < FJ; ++j) asm volatile("" : "+v"(bj[j]));
      }
    }
    // R (residual / GELU' factor): the whole [BM][BN] tile comes in by LDS-DMA, straight into the image positions that
    // the results will overwrite (same row-major layout, same chunk ^ row swizzle, so a lane reads its 8-byte piece and
    // later writes its 8-byte result to the very same address).  16 bytes per lane and whole 512-byte rows per request
    // instead of 8-byte pieces of 16 different rows (fragment layout: 270 -> 2xx us on the fc2 input gradient), no
    // registers, and the row map (if any) is fetched as one batch.  The launcher routes R with an unaligned pitch elsewhere.
    if (has_r && r_early) {                            // three blocks are under way since the last stages; the fourth:
      issue_r_block((nt - 1) % NST);
      wait_vmcnt<0>();
      __syncthreads();
      stamp(10);
    } else if (has_r) {
      constexpr int RPW = 64 / CPRW;                   // image rows per wave request (a row = CPRW 16-byte chunks)
      constexpr int RROWS = NTHR / 64 * RPW, RPASS = BM / RROWS;
      int rq[RPASS];
#pragma unroll
      for (int q = 0; q < RPASS; ++q) rq[q] = min(m0 + q * RROWS + w * RPW + l / CPRW, p.M - 1);   // rows >= M / columns >= N: clamped, never stored
      if (p.r_rows) {
#pragma unroll
        for (int q = 0; q < RPASS; ++q) rq[q] = p.r_rows[rq[q]];
      }
#pragma unroll
      for (int q = 0; q < RPASS; ++q) {
        const int row = q * RROWS + w * RPW + l / CPRW;
        const int col = min(n0 + ((((l % CPRW) ^ (row & 15)) & (CPRW - 1)) << 3), p.N - 8);
        glds16_nt((const T*)p.R + (long)rq[q] * p.ldr + col, img + (q * RROWS + w * RPW) * PITCH);
      }
      wait_vmcnt<0>();
      __syncthreads();
      stamp(10);
    }
    // one fragment: activation / R operand / ReLU on the accumulators, the result into the bf16 image; its column sums into csj / cs2j
    auto fragment = [&](int i, int j, f32x4& csj, f32x4& cs2j) __attribute__((always_inline)) {
      const bool row_ok = m0 + wr * TM + i * FRH + fr < p.M;
      f32x4 v = pre_act(i, j);
      if constexpr (FUSED_C2D) {
        // the two volatile asms pin this fragment between its neighbours: without them the polynomials of all 32
        // fragments are hoisted ahead of the first store and 340 registers spill
        asm volatile("" : "+v"(v));
        f32x2 glo, ghi, dlo, dhi;
        gelu_dgelu_fast2((f32x2){v[0], v[1]}, glo, dlo);
        gelu_dgelu_fast2((f32x2){v[2], v[3]}, ghi, dhi);
        bf16x4 dq = {(bf16)dlo[0], (bf16)dlo[1], (bf16)dhi[0], (bf16)dhi[1]};
        asm volatile("" : "+v"(dq));
        dpk[i][j] = dq;
        v = (f32x4){glo[0], glo[1], ghi[0], ghi[1]};
      } else if (do_gelu) {
        const f32x2 lo = gelu_fast2((f32x2){v[0], v[1]}), hi = gelu_fast2((f32x2){v[2], v[3]});
        v = (f32x4){lo[0], lo[1], hi[0], hi[1]};
      }
      if (has_r) {
        const bf16x4 rb = *(const bf16x4*)(img + (wr * TM + i * FRH + fr) * PITCH +
                                           (((((wc * TN + j * FCW + 4 * fq) >> 3) ^ fr15) & (CPRW - 1)) << 4) + (fq & 1) * 8);
        const f32x4 r = {(float)rb[0], (float)rb[1], (float)rb[2], (float)rb[3]};
        if (do_resid) v += r;
        else if (do_mulr) v *= r;
        else {
          const f32x2 lo = dgelu_fast2((f32x2){r[0], r[1]}), hi = dgelu_fast2((f32x2){r[2], r[3]});
          v *= (f32x4){lo[0], lo[1], hi[0], hi[1]};
        }
      }
      if (do_relu) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
      if (do_cs) { if (row_ok) { csj += v; if (do_sq) cs2j += v * v; } }
      put(i, j, v);
      if (do_gelu || do_dgelu) __builtin_amdgcn_sched_barrier(0);   // one fragment's polynomial temporaries at a time
    };
    constexpr bool CS_BODY = MD < 0 || (MD & E_COLSUM) != 0;
    if constexpr (CS_BODY) {
      // Bodies that sum columns walk the fragments COLUMN GROUP by column group (j outer, i inner) and emit each group's four sums
      // (+ four sums of squares) as soon as its eight fragments are through: one f32x4 (two) of accumulators live instead of
      // FJ = 4 (8) of them.  With all of them live beside the 128 accumulator registers hipcc spilled the sums to scratch and
      // brought them back one at a time, each reload behind an s_waitcnt vmcnt(0): 16 dependent scratch round trips per tile in
      // front of the C stores - on the fc2 input gradient (mul_r + column sums) a third of the epilogue (round 4, from the ISA).
      // The sums still leave BEFORE the C stores are issued (stores count in vmcnt: a later wait would sit out their drain).
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        f32x4 csj = {0.f, 0.f, 0.f, 0.f}, cs2j = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < FI; ++i) fragment(i, j, csj, cs2j);
        if (do_cs) {                                   // fold the 16 rows (fr) of each lane group, then one value per column
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = sum16(csj[e]);
            if constexpr (M32) t += __shfl_xor(t, 16);   // 32 rows per fragment: the two 16-lane rows of this column group
            const int gn = colb + j * FCW + e;
            if (fr == 0 && gn < p.N) {                 // (register epilogue = SWAP kernels only: TM is 128 there)
              if (TM % 128 == 0) cs_emit(p, (m0 + wr * TM) / 128, TM / 128, gn, t);
              else atomicAdd(p.colsum + gn, t);
            }
            if (do_sq) {
              float t2 = sum16(cs2j[e]);
              if constexpr (M32) t2 += __shfl_xor(t2, 16);
              if (fr == 0 && gn < p.N && TM % 128 == 0) cs_emit_sq(p, (m0 + wr * TM) / 128, TM / 128, gn, t2);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);            // keep the column groups apart
      }
    } else {
      f32x4 none = {0.f, 0.f, 0.f, 0.f}, none2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < FI; ++i) {
#pragma unroll
        for (int j = 0; j < FJ; ++j) fragment(i, j, none, none2);
        __builtin_amdgcn_sched_barrier(0);            // keep fragment rows apart: interleaved they spill
      }
    }
    stamp(5);
    readback(p.C, p.ldc);
    stamp(6);
    if constexpr (FUSED_C2D) {
      __syncthreads();                                 // every wave has read its pieces of the C image
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) put_pk(i, j, dpk[i][j]);
      stamp(8);
      readback(p.C2, p.ldc2);
      stamp(9);
    }
#undef EPI_HAS
  };
  // ---------------- fp8 epilogue (SWAP kernels, BASELINE configs[4]): the QKV projection leaves as e4m3 bytes + per-(window, head)
  // scales, so that the attention forward AND backward read half the bytes (swin_512.py:115-121 with q | k | v stored in fp8).
  // bias, q scaling on the accumulators; |v| maximum per (row group, column group): in-lane over the group's fragments, across the
  // wave by DPP, across the 2 (head dim 128) or 4 (256) waves that share the head through 32 floats of LDS; scale = amax / 448;
  // 4 values -> one dword of e4m3 (v_cvt_pk_fp8_f32), a [256][256 B] LDS image (16-byte chunk ^= row & 15), whole 16-byte row
  // pieces out.  One lane per (wave, row group) writes the scale.
  auto epilogue_fp8 = [&]() __attribute__((always_inline)) {
    constexpr int NG = 4;                              // row groups per wave tile at 32 rows per problem (1 group at 128)
    char* img = smem;
    float* ex = (float*)(smem + BM * BN);              // [WM][WN][NG] wave maxima
    const int rg = p.f8_rows, cgw = p.f8_cols / TN;    // rows per scale, waves per column group (2 or 4)
    const int ngroups = TM / rg;                       // 1 (128 rows) or 4 (32 rows)
    const int colb = n0 + wc * TN + 4 * fq;
    f32x4 bj[FJ];
#pragma unroll
    for (int j = 0; j < FJ; ++j) bj[j] = p.bias ? *(const f32x4*)(p.bias + colb + j * FCW) : (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool qcols = n0 + wc * TN < p.scale_cols;    // (scale_cols is a multiple of the head dim: a wave tile is all-q or not)
    float gm[NG] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        f32x4 v = accg(i, j) + bj[j];
        if (qcols) v *= p.scale;
        accs(i, j, v);
        const float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        const int g = ngroups == 1 ? 0 : i * FRH / 32; // the fragments of rows 32g .. 32g + 31
        gm[g] = fmaxf(gm[g], m4);
      }
    // (|values| are non-negative: their maximum is order-free, so the wave reduction runs BEFORE the barrier that frees the ring)
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (g < ngroups) gm[g] = wave_max_dpp(gm[g]);
    __syncthreads();                                   // every wave is done with the ring stages (ex / img live there)
    if (l == 0) {
#pragma unroll
      for (int g = 0; g < NG; ++g) ex[(wr * WN + wc) * NG + g] = gm[g];
    }
    __syncthreads();
    float inv[NG], sc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float m = 0.f;
      const int w0 = (wc / cgw) * cgw;
      for (int u = 0; u < cgw; ++u) m = fmaxf(m, ex[(wr * WN + w0 + u) * NG + g]);
      sc[g] = m > 0.f ? m * (1.0f / 448.0f) : 1.0f;
      inv[g] = 1.0f / sc[g];
    }
    // (the image [0, 64 KB) and the exchange area behind it do not overlap: no barrier between reading one and writing the other)
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        const float k = inv[ngroups == 1 ? 0 : i * FRH / 32];
        const f32x4 q = accg(i, j);
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[0] * k, q[1] * k, pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(q[2] * k, q[3] * k, pk, true);
        const int cl = wc * TN + j * FCW + 4 * fq;         // tile column of this lane's 4 values: 16-byte chunk cl >> 4, byte cl & 15
        const int row = wr * TM + i * FRH + fr, chunk = cl >> 4;
        *(int*)(img + row * BN + (((chunk ^ fr15) & (BN / 16 - 1)) << 4) + (cl & 15)) = pk;
      }
    if (l == 0 && (wc % cgw) == 0) {
      float* st = (float*)p.C2 + (long)((m0 + wr * TM) / rg) * p.ldc2 + (n0 + wc * TN) / p.f8_cols;
      for (int g = 0; g < ngroups; ++g) st[(long)g * p.ldc2] = sc[g];
    }
    __syncthreads();
    constexpr int CPRW8 = BN / 16, RPP8 = NTHR / CPRW8;   // 16-byte pieces per image row, rows per pass
    const int rb_row = tid / CPRW8, rb_chunk = tid % CPRW8;
    typedef int v4i32e __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int ps = 0; ps < BM / RPP8; ++ps) {
      const int row = ps * RPP8 + rb_row;
      const v4i32e val = *(const v4i32e*)(img + row * BN + (((rb_chunk ^ (row & 15)) & (CPRW8 - 1)) << 4));
      __builtin_nontemporal_store(val, (v4i32e*)((char*)p.C + (long)(m0 + row) * p.ldc + n0 + rb_chunk * 16));
    }
  };
  auto epilogue_reg = [&]() __attribute__((always_inline)) {
    if (p.f8_rows > 0) { epilogue_fp8(); return; }
    if (STSWIN_DBG(p.flags, 21)) {    // DBG: no epilogue at all (keeps the accumulators alive)
      stamp(3); stamp(4); stamp(5);
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) { const f32x4 q = accg(i, j); t += q[0] + q[1] + q[2] + q[3]; }
      if (t == 123.456f) ((T*)p.C)[tid] = from_f32<T>(t);
      stamp(6);
      return;
    }
    const int mode = (p.bias ? E_BIAS : 0) | (p.scale_cols > 0 ? E_SCALE : 0) | ((p.flags & GF_GELU) ? E_GELU : 0) |
                     (p.C2 ? E_C2 : 0) | ((p.flags & GF_RESID) ? E_RESID : 0) | ((p.flags & GF_MUL_DGELU) ? E_DGELU : 0) |
                     ((p.colsum && !dbg_ts) ? E_COLSUM : 0) | ((p.flags & GF_RELU) ? E_RELU : 0) | ((p.flags & GF_MUL_R) ? E_MULR : 0) |
                     ((p.flags & GF_C2_DGELU) ? E_C2D : 0) | ((p.colsum && !dbg_ts && (p.flags & GF_CS_SQ)) ? E_COLSQ : 0);
    switch (mode) {                                   // the combinations the Swin / conv paths issue; anything else: generic
      case 0: epilogue_body(std::integral_constant<int, 0>{}); break;
#ifndef STSWIN_DEBUG_ONLY_PLAIN_EPI                   // diagnosis build: a kernel with ONE epilogue body (code size experiment)
      case E_BIAS: epilogue_body(std::integral_constant<int, E_BIAS>{}); break;
      case E_BIAS | E_SCALE: epilogue_body(std::integral_constant<int, E_BIAS | E_SCALE>{}); break;
      case E_BIAS | E_GELU | E_C2: epilogue_body(std::integral_constant<int, E_BIAS | E_GELU | E_C2>{}); break;
      case E_BIAS | E_GELU: epilogue_body(std::integral_constant<int, E_BIAS | E_GELU>{}); break;      // fc1 of a no-grad forward
      case E_BIAS | E_RESID: epilogue_body(std::integral_constant<int, E_BIAS | E_RESID>{}); break;
      case E_RESID: epilogue_body(std::integral_constant<int, E_RESID>{}); break;
      case E_BIAS | E_GELU | E_C2 | E_C2D: epilogue_body(std::integral_constant<int, E_BIAS | E_GELU | E_C2 | E_C2D>{}); break;
      case E_MULR | E_COLSUM: epilogue_body(std::integral_constant<int, E_MULR | E_COLSUM>{}); break;
      case E_DGELU | E_COLSUM: epilogue_body(std::integral_constant<int, E_DGELU | E_COLSUM>{}); break;
      case E_COLSUM: epilogue_body(std::integral_constant<int, E_COLSUM>{}); break;
      case E_COLSUM | E_COLSQ: epilogue_body(std::integral_constant<int, E_COLSUM | E_COLSQ>{}); break;          // conv + BatchNorm statistics
      case E_BIAS | E_COLSUM | E_COLSQ: epilogue_body(std::integral_constant<int, E_BIAS | E_COLSUM | E_COLSQ>{}); break;
      default: epilogue_body(std::integral_constant<int, -1>{}); break;
#else
      default: epilogue_body(std::integral_constant<int, E_COLSUM>{}); break;   // (the timeline runs pass the stamp buffer as colsum)
#endif
    }
  };
  if constexpr (PIPE != 4) {                          // (PIPE 4 stages through registers: its own prologue)
    // (PIPE 7 keeps a stage's fragments in registers one step ahead, so all NST slots can be in flight: prefetch distance NST)
    for (int q = 0; q < (PIPE == 7 ? NST : NST - 1) && q < nt; ++q) issue(q);
  }
  // Start-time stagger (round 5, product): without it the 256 CUs run their tiles in LOCK-STEP - every CU multiplies while HBM idles,
  // then every CU stores its 128-256 KB at once: 32-64 MB per round arrive as one burst, the store instructions back up behind the
  // memory side (C readback + stores 2.1 us on a plain tile, 4.1 + 2.7 us with the second output of fc1: profiles/r05_gemm_tile_timeline.txt)
  // and the matrix pipes wait.  The first-round workgroups of the launch therefore start in 8 phases, `stagger_ticks` apart (4 CUs of
  // every XCD per phase; the launcher spreads them over ~0.7 tile times): the phase pattern persists through the launch (a CU's
  // next workgroup starts when its predecessor ends), the chip's store stream becomes smooth, and the start delay - paid once - is
  // smaller than what every tile gains BEHIND AN HBM-BOUND SPACER KERNEL: -11 .. -19 % on the K = 512 launches with 6-8 tiles per CU
  // (profiles/r05_stagger_behind_spacer.txt, r05_stagger_sweep.txt), nothing or a loss at K >= 1024 / <= 4 tiles per CU.  Inside the
  // training step the same launches gain 0.2-0.7 % and the step nothing (profiles/r05_stagger_in_step_ab.txt,
  // r05_gemm_shapes_in_step_stagger{0,1}.txt): the launcher therefore leaves stagger_ticks at 0 unless STSWIN_NT_STAGGER asks for it.
  // The copies of the first stages are already in flight while a workgroup sleeps.
  if (p.stagger_ticks > 0 && blockIdx.x < 256 && gridDim.y == 1) {
    const unsigned long long t_go = wall_clock64() + (unsigned long long)(((blockIdx.x >> 3) & 7) * p.stagger_ticks);
    while (wall_clock64() < t_go) __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_sched_barrier(0);                 // first get the copies going, then spend 128 v_mov on the accumulators
  if constexpr (M32) {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ / 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#ifdef STSWIN_TUNING
#include "gemm_nt_ring_tuning.inc"   // `if constexpr (PIPE == 0) {..} else if constexpr (PIPE == 2) {..} .. else`: the measured losers, A/B builds only
#else
  static_assert(PIPE == 1 && !PERSIST, "the product library instantiates the ping-pong loop only (other loops: STSWIN_TUNING builds)");
#endif
  {
    // Ping-pong: waves w and w+4 share a SIMD (wave rows wr = 0 / 1).  With ONE barrier per stage both read their
    // fragments at the same time (matrix pipe idle) and then serialise their MFMAs.  Here every stage has two barriers
    // and the two wave rows run half a stage apart: while row 0 reads the fragments of tile kt, row 1 issues the MFMAs
    // of tile kt-1; then they swap.  No extra registers (one fragment set per wave), the matrix pipe always has a
    // wave with operands ready.  Ring safety: tile kt-1's last reader (row 1) finishes before b0(kt), after which both
    // rows may request tile kt+3 into that slot.
    static_assert(NST >= 4 && WM == 2, "ping-pong variant: 4-stage ring, two wave rows");
    const bool lag = (wr == 1);                        // wave-uniform
    constexpr int NA = M32 ? 2 * FI : FI, NB = M32 ? FJ / 2 : FJ;   // fragment registers (32x32x16: [tile * 2 + k half])
    bf16x8 a[NA], b[NB];
    auto read_frags = [&](int q) {
      const char* Ab = smem + (q % NST) * STAGE;
      const char* Bb = Ab + A_BYTES;
      if constexpr (M32) {
#pragma unroll
        for (int g = 0; g < NB; ++g) b[g] = *(const bf16x8*)(Bb + (wc * TN + (g >> 1) * 32) * ROWB + ((g & 1) ? rd_off1 : rd_off));
#pragma unroll
        for (int g = 0; g < NA; ++g) a[g] = *(const bf16x8*)(Ab + (wr * TM + (g >> 1) * 32) * ROWB + ((g & 1) ? rd_off1 : rd_off));
      } else {
#ifdef STSWIN_DEBUG_HALF_READS                         // diagnosis build (tools/probes/half_reads.sh): HALF the fragment reads, the other fragments
#pragma unroll                                         // are copies (wrong results, same MFMA stream on data) - what do the LDS reads cost in clock?
      for (int j = 0; j < FJ; j += 2) { b[j] = *(const bf16x8*)(Bb + (wc * TN + j * 16) * ROWB + rd_off); b[j + 1] = b[j]; }
#pragma unroll
      for (int i = 0; i < FI; i += 2) { a[i] = *(const bf16x8*)(Ab + (wr * TM + i * 16) * ROWB + rd_off); a[i + 1] = a[i]; }
#else
#pragma unroll
      for (int j = 0; j < FJ; ++j) b[j] = *(const bf16x8*)(Bb + (wc * TN + j * 16) * ROWB + rd_off);
#pragma unroll
      for (int i = 0; i < FI; ++i) a[i] = *(const bf16x8*)(Ab + (wr * TM + i * 16) * ROWB + rd_off);
#endif
      }
    };
    auto mma_all = [&]() {
      __builtin_amdgcn_s_setprio(1);
      if constexpr (M32) {
        static_assert(!M32 || SWAP, "32x32x16 tiles: register epilogue only");
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ / 4; ++j)
              acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j * 2 + kh], a[i * 2 + kh], acc32[i][j], 0, 0, 0);
      } else {
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) {
          // SWAP: the weight fragment is the MFMA row operand, so a lane ends up with 4 consecutive COLUMNS of one
          // output row (C[fr][4fq..4fq+3]) instead of 4 consecutive rows of one column - what epilogue_reg wants
          if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    };
    // Both rows run the SAME loop body {barrier; request + read tile kt; barrier; MFMAs of tile kt}; the lag row enters
    // through one extra leading barrier and the lead row leaves through one trailing barrier, so in barrier interval
    // 2kt the lead row reads tile kt while the lag row multiplies tile kt-1, and in interval 2kt+1 the lead row
    // multiplies tile kt while the lag row reads it.  Every wave has waited for ITS share of tile kt before barrier 2kt
    // (lead: after its MFMAs of kt-1; lag: after its reads of kt-1), which is all that differs between the rows.
    auto wait_tile = [&](int kt) {
      const int newer = min(NST - 2, nt - 1 - kt);
      if (newer >= 2) wait_vmcnt<2 * PER_STAGE>();
      else if (newer == 1) wait_vmcnt<PER_STAGE>();
      else wait_vmcnt<0>();
    };
    stamp(1);
    wait_tile(0);
    stamp(2);
    if (lag) __builtin_amdgcn_s_barrier();
    // Steady state without a single data-dependent branch in the body: both rows wait for THEIR share of stage kt+1
    // right after their fragment reads (it was requested two iterations earlier, so the counted wait does not stall),
    // every iteration requests a stage, the wait count is the constant "two younger stages".  The last NST-1
    // iterations (nothing left to request, shrinking wait counts) run in the general form below.  The in-order wave
    // pays ~20 cycles per scalar branch; the general body has ten of them per stage (measured 0.85 us per stage against
    // 0.62 for the same loop without them, tools/probes/pingpong_probe.hip).
    int kt = 0;
    if (p.S == 1) {
      for (; kt + NST - 1 < nt; ++kt) {
#ifdef STSWIN_DEBUG_LOOP_STAMPS                        // diagnosis build only (tools/gemm_timeline.py loop): stage 4 / 8 / 12 times
        if (kt == 4) stamp(1);
        if (kt == 8) stamp(2);
        if (kt == 12) stamp(7);
#endif
        __builtin_amdgcn_s_barrier();
        issue(kt + NST - 1);
        read_frags(kt);
        wait_vmcnt<2 * PER_STAGE>();
        __builtin_amdgcn_s_barrier();
        mma_all();
      }
      // The last NST-1 iterations have no stage left to request - but the epilogue's R tile (residual / GELU' factor) lives
      // in the same LDS as the ring, one 64-row block per stage buffer, and a block costs exactly PER_STAGE copies per
      // wave.  So the loop keeps its steady-state form (same wait constant, no branches) and requests R block (kt-1) % NST
      // into the buffer that stage kt-1 just vacated: three quarters of R are under way 1-3 stages before the epilogue.
      if constexpr (R_EARLY_OK) {
        if ((p.flags & (GF_RESID | GF_MUL_R | GF_MUL_DGELU)) && !p.r_rows && kt + NST - 1 == nt && !STSWIN_DBG(p.flags, 18)) {
          r_early = true;
          for (; kt < nt; ++kt) {
            __builtin_amdgcn_s_barrier();
            issue_r_block((kt + NST - 1) % NST);
            read_frags(kt);
            wait_vmcnt<2 * PER_STAGE>();
            __builtin_amdgcn_s_barrier();
            mma_all();
          }
        }
      }
      // Without R: the three draining iterations, straight-line with their own wait counts (one stage in flight, none, none)
      // instead of the general form's ten scalar branches per stage.
      if (kt + NST - 1 == nt && NST == 4) {
        __builtin_amdgcn_s_barrier();
        read_frags(kt);
        wait_vmcnt<PER_STAGE>();
        __builtin_amdgcn_s_barrier();
        mma_all();
        ++kt;
        __builtin_amdgcn_s_barrier();
        read_frags(kt);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        mma_all();
        ++kt;
        __builtin_amdgcn_s_barrier();
        read_frags(kt);
        __builtin_amdgcn_s_barrier();
        mma_all();
        ++kt;
      }
    }
    for (; kt < nt; ++kt) {                            // general form (tap-segmented A, and the last NST-1 stages)
      __builtin_amdgcn_s_barrier();
      if (kt + NST - 1 < nt) issue(kt + NST - 1);
      read_frags(kt);
      if (lag && kt + 1 < nt) wait_tile(kt + 1);
      __builtin_amdgcn_s_barrier();
      mma_all();
      if (!lag && kt + 1 < nt) wait_tile(kt + 1);       // (the lead row waits after its MFMAs: at a tap switch the fresh
    }                                                   //  index loads of issue() would otherwise be waited for at once)
    if (!lag) __builtin_amdgcn_s_barrier();
    if (arow_bad) __builtin_trap();
    if constexpr (SWAP) epilogue_reg();
    else epilogue();
  }
  if constexpr (PERSIST) {
    tix_ += (int)gridDim.x;
    if (tix_ < ntot_) __syncthreads();               // every wave has read its pieces of the image before the next tile's copies land
  }
  } while (PERSIST && tix_ < ntot_);
}


#ifdef STSWIN_TUNING   // measured losers kept for A/B runs only: not compiled into the product library (build with STSWIN_TUNING=1)
#include "gemm_nt_tuning_kernels.inc"
#endif  // STSWIN_TUNING

// =====================================================================================================
// TN: C[i][j] += sum_m At[m][i] * Bt[m][j]      (both operands have the contraction index as their ROW)
// LDS tiles [BM rows m][128 cols] (bf16: 64 rows x 256 B, f32: 32 rows x 512 B), swz256, read transposed.
// =====================================================================================================
struct GemmTN {
  const void* At; long lda; const int* at_rows;   // rows m (after optional gather) x Ni columns
  const void* Bt; long ldb; const int* bt_rows;   // rows m x Nj columns
  float* C; long ldc;                             // [Ni][Nj] fp32, atomically accumulated
  int Mk, Ni, Nj;
  int splits;                                     // grid.y
  int bseg;                                       // >0: Bt column j reads source column j % bseg of row bt_rows[(j / bseg)*Mk + m]
  float* slabs;                                   // optional [splits][Ni][Nj] partial results (plain stores) instead of atomics
  int slab_bf16;                                  // the partials are stored as bf16 (half the slab traffic; see stswin_gemm_tn)
  // fused split-K combine (gemm_tn_ring_kernel<MODE, true>): arrival / departure counters of the output tiles [2][tiles], zero
  // before and after every launch; overwrite / perm as tn_reduce_kernel's
  unsigned* tile_cnt; int overwrite; int perm;
};

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_tn_kernel(GemmTN p) {
  constexpr int WN = NW / 2, JN = 128 / WN / 16;
  constexpr int PACK = TT<T>::PACK;
  constexpr int BM = TT<T>::IS_BF16 ? 64 : 32;      // contraction rows per tile
  constexpr int ROWB = 128 * sizeof(T);             // 256 / 512
  constexpr int RPI = 1024 / ROWB;                  // rows per wave-instruction: 4 / 2
  constexpr int NI = BM / (NW * RPI);               // DMA instructions per wave per operand: 4 (NW=4) / 2 (NW=8)
  constexpr int CPR = ROWB / 16;                    // chunks per row: 16 / 32
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int wr = w / WN, wc = w % WN;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so `split = bid % splits` (splits is a multiple
  // of 8 or a power of two <= 8) puts every output tile of one contraction range on ONE XCD: its L2 then serves the
  // tiles_i + tiles_j fold re-use of the At / Bt row panel instead of all 8 L2s fetching everything from HBM.
  const int tiles_j = (p.Nj + 127) >> 7;
  const int split_id = blockIdx.x % p.splits, tile_id = blockIdx.x / p.splits;
  const int i0 = (tile_id / tiles_j) << 7, j0 = (tile_id % tiles_j) << 7;
  // split the contraction range in multiples of BM
  const int ntile_all = (p.Mk + BM - 1) / BM;
  const int per = (ntile_all + p.splits - 1) / p.splits;
  const int t_begin = split_id * per, t_end = min(ntile_all, t_begin + per);
  if (t_begin >= t_end) return;

  const char* zero = (const char*)g_stswin_zero;
  const int rsub = l / CPR, cphys = l % CPR;        // row within the instruction's RPI rows, physical chunk
  // per-lane row of instruction i: r = (w*NI + i)*RPI + rsub ; source chunk = cphys ^ swz256(r) (low 4 bits)
  // Row indices come through SCALAR loads (wave-uniform address, lgkmcnt): an ordinary vector load here makes hipcc
  // drain vmcnt(0) right behind the LDS-DMA it was issued after, which serialises copy and MFMA.  A wave stages
  // NI*RPI consecutive contraction rows per tile; each lane then picks the one of its instruction's RPI rows it
  // copies.  With bseg a 128-column tile spans at most two taps (the launcher checks it), so two index rows cover it.
  long arow[NI], brow[NI];                          // gathered source rows of the NEXT tile to stage
  int bcol[NI];                                     // source column of this lane's Bt chunk (constant over tiles)
  bool btap1[NI];                                   // this lane's Bt chunk belongs to the tile's second tap
  const int tap0 = p.bseg > 0 ? j0 / p.bseg : 0;
  const int tap1 = p.bseg > 0 ? min(tap0 + 1, p.Nj / p.bseg - 1) : 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (w * NI + i) * RPI + rsub;
    const int cj = j0 + ((cphys & ~15) | ((cphys ^ swz256(r)) & 15)) * PACK;
    bcol[i] = p.bseg > 0 ? cj % p.bseg : cj;
    btap1[i] = p.bseg > 0 && cj / p.bseg != tap0;
  }
  typedef int ivec __attribute__((ext_vector_type(RPI)));
  typedef const __attribute__((address_space(4))) ivec* civec;
  const bool vec_maps = (p.Mk % RPI) == 0;          // RPI-int scalar loads stay aligned and in range
  auto fetch_rows = [&](int tile) {
    const int mw = tile * BM + w * (NI * RPI);      // wave-uniform
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int mb = mw + i * RPI;
      int va[RPI], vb0[RPI], vb1[RPI];
      if (vec_maps && mb + RPI <= p.Mk) {
        ivec ta, t0, t1;
#pragma unroll
        for (int r = 0; r < RPI; ++r) { ta[r] = mb + r; t0[r] = mb + r; t1[r] = -1; }
        if (p.at_rows) ta = *(civec)(p.at_rows + mb);
        if (p.bseg > 0) {
          t0 = *(civec)(p.bt_rows + (long)tap0 * p.Mk + mb);
          t1 = *(civec)(p.bt_rows + (long)tap1 * p.Mk + mb);
        } else if (p.bt_rows) {
          t0 = *(civec)(p.bt_rows + mb);
        }
#pragma unroll
        for (int r = 0; r < RPI; ++r) { va[r] = ta[r]; vb0[r] = t0[r]; vb1[r] = t1[r]; }
      } else {
#pragma unroll
        for (int r = 0; r < RPI; ++r) {
          const int m = mb + r, mc = min(m, p.Mk - 1);
          va[r] = p.at_rows ? sload(p.at_rows, mc) : mc;
          vb1[r] = -1;
          if (p.bseg > 0) {
            vb0[r] = sload(p.bt_rows, (long)tap0 * p.Mk + mc);
            vb1[r] = sload(p.bt_rows, (long)tap1 * p.Mk + mc);
          } else {
            vb0[r] = p.bt_rows ? sload(p.bt_rows, mc) : mc;
          }
          if (m >= p.Mk) { va[r] = -1; vb0[r] = -1; vb1[r] = -1; }
        }
      }
      int ra = -1, rb = -1;
#pragma unroll
      for (int r = 0; r < RPI; ++r)
        if (rsub == r) { ra = va[r]; rb = btap1[i] ? vb1[r] : vb0[r]; }
      arow[i] = ra; brow[i] = rb;
    }
  };
  auto stage = [&](int buf) {
    char* Ab = smem + buf * 32768;
    char* Bb = Ab + 16384;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = (w * NI + i) * RPI + rsub;
      const int cs = (cphys & ~15) | ((cphys ^ swz256(r)) & 15);
      const int ci = i0 + cs * PACK, cj = j0 + cs * PACK;
      const char* sa = (arow[i] >= 0 && ci < p.Ni) ? (const char*)p.At + (arow[i] * p.lda + ci) * sizeof(T) : zero;
      const char* sb = (brow[i] >= 0 && cj < p.Nj) ? (const char*)p.Bt + (brow[i] * p.ldb + bcol[i]) * sizeof(T) : zero;
      glds16_raw(sa, Ab + (w * NI + i) * 1024);     // raw: behind the builtin hipcc drains vmcnt(0) before every
      glds16_raw(sb, Bb + (w * NI + i) * 1024);     // ds_read_b64_tr_b16, i.e. waits for the tile it has just requested
    }
  };

  f32x4 acc[4][JN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = l & 15, fq = l >> 4;
  const bool row_active = i0 + wr * 64 < p.Ni;
  fetch_rows(t_begin);
  stage(0);
  if (t_begin + 1 < t_end) fetch_rows(t_begin + 1);
  for (int tk = t_begin; tk < t_end; ++tk) {
    const int buf = (tk - t_begin) & 1;
    wait_vm0();
    __syncthreads();
    if (tk + 1 < t_end) stage(buf ^ 1);
    const char* Ab = smem + buf * 32768;
    const char* Bb = Ab + 16384;
    // (wave rows whose 64 output rows lie beyond Ni - the second row of a tile when Ni <= 64, ResNet layer 1 - only copy)
    if (!row_active) {
    } else if constexpr (TT<T>::IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], b[JN];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          a[i] = cat4(lds_tr4(Ab, ROWB, kk * 32 + 8 * fq, wr * 64 + i * 16),
                      lds_tr4(Ab, ROWB, kk * 32 + 8 * fq + 4, wr * 64 + i * 16));
#pragma unroll
        for (int j = 0; j < JN; ++j)
          b[j] = cat4(lds_tr4(Bb, ROWB, kk * 32 + 8 * fq, wc * (16 * JN) + j * 16),
                      lds_tr4(Bb, ROWB, kk * 32 + 8 * fq + 4, wc * (16 * JN) + j * 16));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int row = kk * 4 + fq;
        float a[4], b[JN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int col = wr * 64 + i * 16 + fr;
          a[i] = *(const float*)(Ab + row * ROWB + ((((col >> 2) & ~15) | (((col >> 2) ^ swz256(row)) & 15)) << 4) + (col & 3) * 4);
        }
#pragma unroll
        for (int j = 0; j < JN; ++j) {
          const int col = wc * (16 * JN) + j * 16 + fr;
          b[j] = *(const float*)(Bb + row * ROWB + ((((col >> 2) & ~15) | (((col >> 2) ^ swz256(row)) & 15)) << 4) + (col & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < JN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    // the scalar map loads of the tile after next go out BEHIND the fragment reads: requested in front of them they turned
    // the staged lgkmcnt waits of the reads into one wait for everything (see gemm_tn_ring_kernel)
    if (tk + 2 < t_end) fetch_rows(tk + 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wr * 64 + i * 16 + 4 * fq + r;
        const int gj = j0 + wc * (16 * JN) + j * 16 + fr;
        if (gi < p.Ni && gj < p.Nj) {
          if (p.slabs) {
            const long so = ((long)split_id * p.Ni + gi) * p.Nj + gj;
            if (p.slab_bf16) ((bf16*)p.slabs)[so] = (bf16)acc[i][j][r];
            else p.slabs[so] = acc[i][j][r];
          }
          else atomicAdd(p.C + (long)gi * p.ldc + gj, acc[i][j][r]);
        }
      }
}


// =====================================================================================================
// gemm_tn "ring" (bf16): 256x256 output tile, 8 waves as 2 x 4 of 128x64, contraction in stages of 32 rows.  At / Bt
// stage tiles [32][256] (512-byte rows, swz256 on the low 4 chunk bits) live in a 4-stage LDS ring filled by raw
// LDS-DMA (glds16_raw: three stages stay in flight, counted vmcnt waits), operands come back through
// ds_read_b64_tr_b16, and the two wave rows run half a stage apart (ping-pong, as gemm_nt_ring_kernel) so one
// row's MFMAs cover the other row's LDS reads.  Gather maps are read with scalar loads one stage ahead.
// One workgroup per CU: the launcher picks splits so that tiles x splits ~ 256.
// =====================================================================================================
// MODE: 0 no gather maps, 1 at_rows only, 2 bt_rows only, 3 bt_rows with per-tap maps (bseg).  The launcher sends everything
// else (both maps, Mk not a multiple of 32 with maps) to the 128x128 kernel: with the map handling resolved at compile time a
// stage costs one or two scalar index loads, a dozen scalar multiplies / selects and 2 VALU per copy; written with runtime
// tests it was 18 scalar branches and ~150 SALU instructions per stage, longer than the partner row's MFMA phase.
// FUSE (round 4): the split-K combine runs INSIDE this launch.  The MFMA operands are swapped (a lane then holds 4 consecutive output
// COLUMNS of one row), the bf16 partial tile goes through one [256][256] LDS image and leaves as whole 16-byte row pieces - written
// THROUGH the caches (sc1) - every wave drains its stores, one lane adds 1 to the tile's arrival counter (agent scope); once all
// `splits` partial tiles of the output tile are complete the workgroups of the tile that are present add its row slices
// [slice * 256 / splits, ...) in split order (sc1 loads: no acquire fence needed, MI355X_MICROARCH.md "Valid forms", row 1) and store
// the fp32 result.  Same values bit for bit as tn_reduce_kernel (same fp32 additions in the same order).  It replaces a 9.6 us
// all-chip pass and a kernel boundary per weight gradient (69 of them per training step), and the 128 two-byte stores per lane
// of the unfused epilogue.  Progress (round 5): slices are handed out by ticket and an early workgroup waits a bounded time, then
// leaves (see the hand-over below) - no workgroup needs another one to be resident, so a busy second stream, a second process or an
// RCCL kernel holding compute units costs time, never correctness; the launcher still fuses only grids of <= one workgroup per CU,
// because that is where the distributed combine is faster than the separate pass.
#ifdef STSWIN_TUNING
// tuning builds: stamps of a NORMAL run (results written, combine included) into a caller-set buffer [workgroups][8] u64
// (stswin_debug_set_tn_stamps; tools/tn_group_timeline.py): 0 start, 1 prologue issued, 2 first stage landed, 3 main loop done,
// 4 the tile's partials complete, 5 combine done, 6 Ni of the problem, 7 split * 1000 + tile
__device__ unsigned long long* g_tn_stamp_buf = nullptr;
extern "C" int stswin_debug_set_tn_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tn_stamp_buf), &buf, sizeof(buf)); }
#endif
// The kernel body as a device function of (problem, position `seq` in the problem's split-major (split, tile) sequence, workgroups of
// the problem, stamp row): gemm_tn_ring_kernel runs one problem per launch, gemm_tn_ring_group_kernel (round 5) several.
template <int MODE, bool FUSE>
DEVI void gemm_tn_ring_body(const GemmTN& p, const int seq, const int nwg, const int stamp_row) {
  constexpr bool MAPS = MODE != 0, MAP_A = MODE == 1, MAP_B = MODE >= 2, TAPS = MODE == 3;
  using T = bf16;
  constexpr int BMK = 32, ROWB = 512, NST = 4, OP_BYTES = BMK * ROWB, STAGE = 2 * OP_BYTES, PER_STAGE = 4;
  constexpr int FI = 8, FJ = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int wr = w >> 2, wc = w & 3;
  const int tiles_j = (p.Nj + 255) >> 8;
  // XCD-aware mapping (round 5): the workgroups of one SPLIT (same contraction rows, different output tiles) re-use each other's operand
  // panels - the At panel of a tile row is read by tiles_j of them, the Bt panel of a tile column by tiles_i - so they should share an L2.
  // Workgroups are dealt round-robin over the 8 XCDs; xcd_remap hands every XCD a CONTIGUOUS range of the split-major sequence
  // (split, tile), so a split's tiles sit on one XCD (or straddle two) WHATEVER the split count.  The round-1 mapping
  // (split = blockIdx.x % splits) did that only for split counts that are multiples of 8: with 21 splits (the 1536 x 512 qkv weight
  // gradient: 12 tiles) or 7 (the 512 x 4608 convolution weight gradients: 36 tiles) every XCD fetched every panel - 807 TFLOP/s
  // against 1079 for the 16-split 2048 x 512 shape with the same 256 workgroups (profiles/r05_tn_gather_cost.txt).
  const int ntiles_ = nwg / p.splits;
  const int split_id = seq / ntiles_, tile_id = seq - split_id * ntiles_;
  const int i0 = (tile_id / tiles_j) << 8, j0 = (tile_id % tiles_j) << 8;
  const int nst_all = (p.Mk + BMK - 1) / BMK;
  const int per = (nst_all + p.splits - 1) / p.splits;
  const int s_begin = split_id * per, s_end = min(nst_all, s_begin + per);
  const int nt = max(0, s_end - s_begin);
  const bool dbg_ts = p.ldc == -1;                       // DBG (tools/gemm_timeline.py tn): C is a u64 [workgroups][8] timestamp buffer
  auto stamp = [&](int slot) {
    if (dbg_ts && threadIdx.x == 0) ((unsigned long long*)p.C)[(long)stamp_row * 8 + slot] = wall_clock64();
#ifdef STSWIN_TUNING
    else if (g_tn_stamp_buf && threadIdx.x == 0) {
      g_tn_stamp_buf[(long)stamp_row * 8 + slot] = wall_clock64();
      if (slot == 0) { g_tn_stamp_buf[(long)stamp_row * 8 + 6] = (unsigned long long)p.Ni; g_tn_stamp_buf[(long)stamp_row * 8 + 7] = (unsigned long long)(split_id * 1000 + tile_id); }
    }
#endif
  };
  stamp(0);

  // one LDS-DMA instruction = 2 rows x 512 B; wave w issues instructions w*2 + {0,1} of each operand = stage rows 4w..4w+3.
  // Buffer-addressed (raw ISA, see glds16_buf_raw): per lane a 32-bit offset = source row * pitch + column bytes, or
  // 0xFFFFFFFF (out of range -> zeros) for missing rows / columns; without gather maps the row part is the scalar offset.
  const int rsub = l >> 5, cphys = l & 31;
  const buf_rsrc_t rsA = make_buf_rsrc(p.At), rsB = make_buf_rsrc(p.Bt);
  const unsigned pitchA = (unsigned)(p.lda * sizeof(T)), pitchB = (unsigned)(p.ldb * sizeof(T));
  const int limA = (int)(0xFFFF0000u / pitchA), limB = (int)(0xFFFF0000u / pitchB);   // rows a 32-bit offset can reach
  if ((!MAP_A && p.Mk > limA) || (!MAP_B && p.Mk > limB)) __builtin_trap();
  const bool ragged_cols = (p.Ni & 255) != 0 || (p.Nj & 255) != 0;
  unsigned offA[2], offB[2];                        // byte offset of this lane's chunk inside its source row, or 0xFFFFFFFF
  bool tap1[2];
  const int tapA0 = TAPS ? j0 / p.bseg : 0;
  const int tapA1 = TAPS ? min(tapA0 + 1, p.Nj / p.bseg - 1) : 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (w * 2 + i) * 2 + rsub;
    const int cs = (cphys & ~15) | ((cphys ^ swz256(r)) & 15);
    const int ci = i0 + cs * 8, cj = j0 + cs * 8;
    offA[i] = ci < p.Ni ? (unsigned)(ci * sizeof(T)) : 0xFFFFFFFFu;
    int sc = cj;
    tap1[i] = false;
    if (TAPS) { sc = cj % p.bseg; tap1[i] = (cj / p.bseg) != tapA0; }
    offB[i] = cj < p.Nj ? (unsigned)(sc * sizeof(T)) : 0xFFFFFFFFu;
    // an operand without a map: the lane's row-in-stage is folded into the offset once, the stage's first row is the
    // copy's scalar offset
    if (!MAP_A && offA[i] != 0xFFFFFFFFu) offA[i] += (unsigned)r * pitchA;
    if (!MAP_B && offB[i] != 0xFFFFFFFFu) offB[i] += (unsigned)r * pitchB;
  }
  typedef int int4v __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(4))) int4v* cint4;
  // Gather indices live in SGPR sets (source rows of the wave's stage rows 4w..4w+3; second set = second tap).  A scalar
  // load from the map takes longer than one loop iteration to come back (measured: +0.45 us per stage when the set is
  // requested one iteration ahead) and SMEM returns out of order, so any wait is a wait for ALL requests: the steady loop
  // below runs two iterations per trip and requests the NEXT trip's pair of sets at its top, i.e. two / three iterations
  // before their use and with no other scalar request in between.
  struct IdxSet { int4v x0, x1; };
  IdxSet SA0, SA1, SB0, SB1, SC0, SC1;
  int ixmax = 0;                                    // largest index seen: range-checked once, after the loop
  auto load_idx = [&](IdxSet& d, int q) {            // (maps: Mk % 32 == 0, so every stage is whole and 16-byte aligned)
    d.x0 = (int4v){0, 0, 0, 0}; d.x1 = (int4v){-1, -1, -1, -1};
    if constexpr (MAPS) {
      if (q >= nt) return;
      const int mb = (s_begin + q) * BMK + w * 4;   // wave-uniform
      const int* map = MAP_A ? p.at_rows : p.bt_rows;
#ifdef STSWIN_DEBUG_TN_NOSMEM                          // diagnosis build: identity indices computed in SALU, no scalar loads in the loop
      d.x0 = (int4v){mb, mb + 1, mb + 2, mb + 3};
      if constexpr (TAPS) d.x1 = d.x0;
      (void)map;
#else
      d.x0 = *(cint4)(map + (TAPS ? (long)tapA0 * p.Mk : 0L) + mb);
      if constexpr (TAPS) d.x1 = *(cint4)(map + (long)tapA1 * p.Mk + mb);
#endif
    }
  };
  auto issue = [&](int q, const IdxSet& ix) {        // stage q of this split -> ring slot q & 3
    const int4v ix0 = ix.x0, ix1 = ix.x1;
    char* Ab = smem + (q & 3) * STAGE;
    char* Bb = Ab + OP_BYTES;
    const int mb = (s_begin + q) * BMK;              // first contraction row of the stage (wave-uniform)
    const bool whole = MAPS || mb + BMK <= p.Mk;     // (maps: Mk % 32 == 0, the launcher checks it - no test in the loop)
    unsigned s0[4], s1[4];
    if constexpr (MAPS) {
      const unsigned pitch = MAP_A ? pitchA : pitchB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ixmax = max(ixmax, max(ix0[r], ix1[r]));
        s0[r] = ix0[r] >= 0 ? (unsigned)ix0[r] * pitch : 0xFFFF0000u;         // -1 (padding) -> out of range -> zeros
        s1[r] = ix1[r] >= 0 ? (unsigned)ix1[r] * pitch : 0xFFFF0000u;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned va = offA[i], vb = offB[i], sa = (unsigned)mb * pitchA, sb = (unsigned)mb * pitchB;
      if constexpr (MAP_A) { va = (rsub ? s0[2 * i + 1] : s0[2 * i]) + offA[i]; sa = 0u; }
      if constexpr (MAP_B) {
        const unsigned v0 = rsub ? s0[2 * i + 1] : s0[2 * i];
        if constexpr (TAPS) { const unsigned v1 = rsub ? s1[2 * i + 1] : s1[2 * i]; vb = (tap1[i] ? v1 : v0) + offB[i]; }
        else vb = v0 + offB[i];
        sb = 0u;
      }
      if (ragged_cols) {                              // lanes whose column does not exist (uniform test, rare)
        if (offA[i] == 0xFFFFFFFFu) va = 0xFFFFFFFFu;
        if (offB[i] == 0xFFFFFFFFu) vb = 0xFFFFFFFFu;
      }
      if (!whole) {                                   // ragged last stage (no maps): rows >= Mk read zeros
        const bool in = mb + (w * 2 + i) * 2 + rsub < p.Mk;
        if (!in) { va = 0xFFFFFFFFu; vb = 0xFFFFFFFFu; }
      }
      glds16_buf_raw(rsA, va, sa, Ab + (w * 2 + i) * 1024);
      glds16_buf_raw(rsB, vb, sb, Bb + (w * 2 + i) * 1024);
    }
  };

  f32x4 acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = l & 15, fq = l >> 4;
  const bool lag = (wr == 1);
  bf16x8 a[FI], b[FJ];
  auto read_frags = [&](int q) {
    const char* Ab = smem + (q & 3) * STAGE;
    const char* Bb = Ab + OP_BYTES;
#pragma unroll
    for (int j = 0; j < FJ; ++j) b[j] = cat4(lds_tr4(Bb, ROWB, 8 * fq, wc * 64 + j * 16), lds_tr4(Bb, ROWB, 8 * fq + 4, wc * 64 + j * 16));
#pragma unroll
    for (int i = 0; i < FI; ++i) a[i] = cat4(lds_tr4(Ab, ROWB, 8 * fq, wr * 128 + i * 16), lds_tr4(Ab, ROWB, 8 * fq + 4, wr * 128 + i * 16));
  };
  auto mma_all = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        if constexpr (FUSE) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);   // C^T fragment: 4 columns per lane
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_tile = [&](int kt) {
    const int newer = min(NST - 2, nt - 1 - kt);
    if (newer >= 2) wait_vmcnt<2 * PER_STAGE>();
    else if (newer == 1) wait_vmcnt<PER_STAGE>();
    else wait_vmcnt<0>();
  };
  if (nt > 0) {
    for (int q = 0; q < NST - 1 && q < nt; ++q) { load_idx(SA0, q); issue(q, SA0); }   // prologue: blocking index loads
    load_idx(SA0, NST - 1);                            // sets for iterations 0 and 1 of the steady loop ...
    load_idx(SA1, NST);
    load_idx(SB0, NST + 1);                            // ... and for iterations 2 and 3 (maps only; no-ops otherwise)
    load_idx(SB1, NST + 2);
    stamp(1);
    wait_tile(0);
    stamp(2);
    if (lag) __builtin_amdgcn_s_barrier();
    // steady state without data-dependent branches (see gemm_nt_ring_kernel): both rows wait for their share of stage
    // kt+1 right behind their fragment reads, with the constant "two younger stages" count.
    // Scalar map loads: SMEM returns out of order, so hipcc turns the first use of ANY loaded set into lgkmcnt(0), a wait
    // for everything in flight.  Requested at the top of an iteration (as before) the fresh pair was waited for at once
    // (+0.12 us per stage).  Now a pair is requested behind an even iteration's MFMAs, four iterations before its use: the
    // next lgkmcnt(0) - the odd iteration's fragment reads - finds it (nearly) landed, and the sets used meanwhile were
    // covered by earlier waits.  Three pairs rotate: A in use, B next trip, C in flight.
    auto body = [&](int kt, const IdxSet& cur, IdxSet* n0, IdxSet* n1) {
      __builtin_amdgcn_s_barrier();
      issue(kt + NST - 1, cur);
      read_frags(kt);
      wait_vmcnt<2 * PER_STAGE>();
      __builtin_amdgcn_s_barrier();
      mma_all();
      if (MAPS && n0) {                                // (behind the last fragment wait: a pending scalar load turns the
        __builtin_amdgcn_sched_barrier(0);             //  staged lgkmcnt(N) waits of the fragment reads into lgkmcnt(0))
        load_idx(*n0, kt + 4 + NST - 1);               // for the iterations kt+4 and kt+5
        load_idx(*n1, kt + 5 + NST - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    int kt = 0;
    for (; kt + 2 + NST - 1 <= nt; kt += 2) {          // 2 iterations per trip
      body(kt, SA0, &SC0, &SC1);
      body(kt + 1, SA1, nullptr, nullptr);
      SA0 = SB0; SA1 = SB1; SB0 = SC0; SB1 = SC1;
    }
    // remainder (< 2 requesting iterations): SA0 / SA1 hold the sets of kt and kt+1
    for (int r = 0; kt + NST - 1 < nt; ++kt, ++r) {
      __builtin_amdgcn_s_barrier();
      if (r == 0) issue(kt + NST - 1, SA0);
      else if (r == 1) issue(kt + NST - 1, SA1);
      else { load_idx(SB0, kt + NST - 1); issue(kt + NST - 1, SB0); }
      read_frags(kt);
      wait_vmcnt<2 * PER_STAGE>();
      __builtin_amdgcn_s_barrier();
      mma_all();
    }
    for (; kt < nt; ++kt) {
      __builtin_amdgcn_s_barrier();
      read_frags(kt);
      if (kt + 1 < nt) wait_tile(kt + 1);
      __builtin_amdgcn_s_barrier();
      mma_all();
    }
    if (!lag) __builtin_amdgcn_s_barrier();
  }
  if (MAPS && ixmax > (MAP_A ? limA : limB)) __builtin_trap();   // gathered operand beyond the 4 GB a 32-bit offset reaches
  stamp(3);
  if (p.ldc < 0) return;                               // debug / timing runs: no result
  if constexpr (FUSE) {
    typedef int v4i32 __attribute__((ext_vector_type(4)));
    char* img = smem;                                  // [256 rows][512 B] bf16, 16-byte chunk ^= row & 15 (conflict-free both ways)
    __syncthreads();                                   // every wave is done with the ring stages
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FJ; ++j) {
        const int row = wr * 128 + i * 16 + fr, chunk = (wc * 64 + j * 16 + 4 * fq) >> 3;
        const bf16x4 o = {(bf16)acc[i][j][0], (bf16)acc[i][j][1], (bf16)acc[i][j][2], (bf16)acc[i][j][3]};
        *(bf16x4*)(img + row * 512 + (((chunk ^ fr) & 31) << 4) + (fq & 1) * 8) = o;
      }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)p.slabs, (short)0, (int)0xFFFFFFFE, 0x00020000);
    const int rb_row = tid >> 5, rb_chunk = tid & 31;
    const int gj = j0 + rb_chunk * 8;                  // this thread's 8 columns (Nj % 8 == 0: all or none exist)
    const long plane = (long)p.Ni * p.Nj;
    if (gj < p.Nj) {
#pragma unroll
      for (int ps = 0; ps < 16; ++ps) {
        const int row = ps * 16 + rb_row, gi = i0 + row;
        if (gi < p.Ni) {
          const v4i32 val = *(const v4i32*)(img + row * 512 + (((rb_chunk ^ (row & 15)) & 31) << 4));
          __builtin_amdgcn_raw_buffer_store_b128(val, rsS, (int)((((long)split_id * plane) + (long)gi * p.Nj + gj) * 2), 0, 16);   // sc1
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                   // ... before the one lane that signals for all of them (the image is dead now)
    typedef __attribute__((address_space(1))) unsigned gu32;
    const int ntiles = nwg / p.splits;
    gu32* arrive = (gu32*)p.tile_cnt + tile_id;
    gu32* depart = (gu32*)p.tile_cnt + ntiles + tile_id;
    gu32* ticket = (gu32*)p.tile_cnt + 2 * ntiles + tile_id;
    // Round 5: NO workgroup depends on another one being resident.  The combine of a tile is cut into `splits` row slices that are
    // handed out by TICKET to whichever workgroups of the tile are present once all `splits` partial tiles are complete: the workgroup
    // whose arrival was the last one (it never waits) and every earlier one that is still polling.  An early workgroup polls for at most
    // TN_WAIT_TICKS and then simply leaves - its CU goes to workgroups that have not started yet (a grid larger than the free CUs: another
    // stream or process, an RCCL kernel of the communication stream) - and the slices it would have taken fall to those who stay; in the
    // worst case the last arriver adds the whole tile.  Every slice is summed in split order by exactly one workgroup, so the result is
    // bit for bit that of tn_reduce_kernel whoever does it.  (Round 4 made every workgroup wait for its tile's other splits: correct only
    // with the whole grid resident, and a trap after 2^26 polls otherwise.)
    constexpr unsigned long long TN_WAIT_TICKS = 20000ull;   // 200 us of the 100 MHz clock: arrival skew of a resident tile is < 20 us
    int* bc = (int*)smem;                              // [0] combine? [1] slice ticket - broadcast words
    if (tid == 0) {
      const unsigned n = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      int ok = 1;
      if (n < (unsigned)p.splits) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.splits) {
          __builtin_amdgcn_s_sleep(8);
          if (wall_clock64() - t0 > TN_WAIT_TICKS) { ok = 0; break; }
        }
      }
      bc[0] = ok;
    }
    __syncthreads();                                   // the other waves load behind the barrier the polling wave joins
    stamp(4);
    const bool go = bc[0] != 0;
    const int per = (256 + p.splits - 1) / p.splits;   // rows of a slice
    while (go) {
      __syncthreads();                                 // (everybody has read the previous ticket)
      if (tid == 0) bc[1] = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int slice = bc[1];
      if (slice >= p.splits) break;
      const int r_end = min(256, (slice + 1) * per);
      if (gj < p.Nj) {
        for (int row = slice * per + rb_row; row < r_end; row += 16) {
          const int gi = i0 + row;
          if (gi >= p.Ni) break;
          const int off0 = (int)(((long)gi * p.Nj + gj) * 2);
          float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          int sp = 0;
          for (; sp + 8 <= p.splits; sp += 8) {          // eight partial rows in flight, added in split order
            v4i32 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsS, off0, (int)((long)(sp + u) * plane * 2), 16);   // sc1
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const bf16x8 h = __builtin_bit_cast(bf16x8, v[u]);
#pragma unroll
              for (int e = 0; e < 8; ++e) s8[e] += (float)h[e];
            }
          }
          for (; sp < p.splits; ++sp) {
            const bf16x8 h = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsS, off0, (int)((long)sp * plane * 2), 16));
#pragma unroll
            for (int e = 0; e < 8; ++e) s8[e] += (float)h[e];
          }
          if (p.perm > 0) {                              // tap-minor output (see tn_reduce_kernel)
            const int S = p.Nj / p.perm, sg = gj / p.perm, c = gj - sg * p.perm;
            float* d = p.C + (long)gi * p.ldc + (long)c * S + sg;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[(long)e * S] = p.overwrite ? s8[e] : d[(long)e * S] + s8[e];
          } else {
            float* dst = p.C + (long)gi * p.ldc + gj;
            if ((p.ldc & 3) == 0) {
              f32x4 lo = {s8[0], s8[1], s8[2], s8[3]}, hi = {s8[4], s8[5], s8[6], s8[7]};
              if (!p.overwrite) { lo += *(const f32x4*)dst; hi += *(const f32x4*)(dst + 4); }
              *(f32x4*)dst = lo;
              *(f32x4*)(dst + 4) = hi;
            } else {
              for (int e = 0; e < 8; ++e) dst[e] = p.overwrite ? s8[e] : dst[e] + s8[e];
            }
          }
        }
      }
    }
    // departure: the last workgroup of the tile to leave (its own slices done, every other one gone) zeroes the three counters for the
    // next launch on this stream
    __syncthreads();
    if (tid == 0) {
      const unsigned d = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == (unsigned)p.splits - 1) {
        __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    stamp(5);
    return;
  }
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wr * 128 + i * 16 + 4 * fq + r;
        const int gj = j0 + wc * 64 + j * 16 + fr;
        if (gi < p.Ni && gj < p.Nj) {
          if (p.slabs) {
            const long so = ((long)split_id * p.Ni + gi) * p.Nj + gj;
            if (p.slab_bf16) ((bf16*)p.slabs)[so] = (bf16)acc[i][j][r];
            else p.slabs[so] = acc[i][j][r];
          }
          else atomicAdd(p.C + (long)gi * p.ldc + gj, acc[i][j][r]);
        }
      }
}

template <int MODE, bool FUSE = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_ring_kernel(GemmTN p) {
  gemm_tn_ring_body<MODE, FUSE>(p, xcd_remap((int)blockIdx.x, (int)gridDim.x), (int)gridDim.x, (int)blockIdx.x);
}

// Several weight-gradient problems in ONE launch (round 5, stswin_gemm_tn_group): the three late weight gradients of a Swin block
// (fc1, proj, qkv) are 16 + 4 + 12 output tiles at stage 1 - launched one by one each needs 16 / 64 / 21 splits to fill the chip and pays
// its own ramp, prologue, partial-tile round trip and combine (~30 us of a 58-129 us launch); together they are 32 tiles x 7-9 splits =
// 256 workgroups with twice to eight times the contraction rows per workgroup.  Measured (profiles/r05_tn_group_timeline_static.txt,
// r05_tn_group_static_in_step_ab.txt): 237 us for the three of stage 1 against 134 + 54 + 116 one by one; step +1.9 %.
#define TN_GROUP_MAX 4
struct GemmTNGroup { GemmTN q[TN_GROUP_MAX]; int first[TN_GROUP_MAX + 1]; int mode[TN_GROUP_MAX]; int count; };
// The same with the problems' slots and gather modes fixed at compile time (M0..M3, -1 = no such slot): every field of a problem then
// sits at a constant kernel-argument offset like the single-problem kernel's.  Indexed dynamically (the generic kernel below) the
// selected problem's fields are held in SGPRs for the whole body and the row-map bodies - six index sets = 48 SGPRs - spill 19 of
// them into VGPR lanes: 1.07 us per stage against 0.78 us of the plain workgroups beside them (profiles/r05_tn_group_timeline.txt).
template <int M0, int M1, int M2, int M3>
__global__ __launch_bounds__(512, 2) void gemm_tn_ring_group_static_kernel(GemmTNGroup g) {
  const int seq = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  if constexpr (M3 >= 0) {
    if (seq >= g.first[3]) { gemm_tn_ring_body<M3 < 0 ? 0 : M3, true>(g.q[3], seq - g.first[3], g.first[4] - g.first[3], (int)blockIdx.x); return; }
  }
  if constexpr (M2 >= 0) {
    if (seq >= g.first[2]) { gemm_tn_ring_body<M2 < 0 ? 0 : M2, true>(g.q[2], seq - g.first[2], g.first[3] - g.first[2], (int)blockIdx.x); return; }
  }
  if constexpr (M1 >= 0) {
    if (seq >= g.first[1]) { gemm_tn_ring_body<M1 < 0 ? 0 : M1, true>(g.q[1], seq - g.first[1], g.first[2] - g.first[1], (int)blockIdx.x); return; }
  }
  gemm_tn_ring_body<M0, true>(g.q[0], seq, g.first[1], (int)blockIdx.x);
}
__global__ __launch_bounds__(512, 2) void gemm_tn_ring_group_kernel(GemmTNGroup g) {
  const int seq = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  int i = 0;
  while (i + 1 < g.count && seq >= g.first[i + 1]) ++i;
  const int local = seq - g.first[i], nwg = g.first[i + 1] - g.first[i];
  switch (g.mode[i]) {
    case 0: gemm_tn_ring_body<0, true>(g.q[i], local, nwg, (int)blockIdx.x); break;
    case 1: gemm_tn_ring_body<1, true>(g.q[i], local, nwg, (int)blockIdx.x); break;
    case 2: gemm_tn_ring_body<2, true>(g.q[i], local, nwg, (int)blockIdx.x); break;
    default: gemm_tn_ring_body<3, true>(g.q[i], local, nwg, (int)blockIdx.x); break;
  }
}

// C[i][j] += sum_s slabs[s][i][j]   (the split-K combine: plain stores + this pass cost ~half of 32 MB of fp32 atomics).
// bf16 partials: 8 outputs per thread (16-byte loads) and eight splits requested before the first is consumed - the pass is a
// pure HBM stream (splits x Ni x Nj x 2 bytes in, 4 bytes per output out); with 8-byte loads issued one split at a time it
// ran at 2.5 TB/s (14.5 us for the 16-split 512 x 2048 weight gradients, 1.4 ms per training step over its 80 launches).
// perm > 0 (convolution weight gradients, STSWIN_TN_OUT_TAPMINOR): column j = (tap s, channel c) = (j / perm, j % perm) of the GEMM result is stored at
// column c * S + s (S = Nj / perm taps): row i then IS conv.weight.grad[i] in its own [cin][k][k] layout - no permuting copy afterwards.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* slabs, float* C, long ldc, int Ni, int Nj, int splits,
                                                        int overwrite, int slab_bf16, int perm) {
  if (slab_bf16) {
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    const long total = (long)Ni * Nj;
    if (idx >= total) return;
    const int i = idx / Nj, j = idx % Nj;         // Nj % 8 == 0 (checked by the launcher)
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16* src = (const bf16*)slabs + idx;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const bf16x8*)(src + (long)(s + u) * total);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += (float)v[u][e];
    }
    for (; s < splits; ++s) {
      const bf16x8 v = *(const bf16x8*)(src + (long)s * total);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
    }
    if (perm > 0) {
      const int S = Nj / perm, sg = j / perm, c = j - sg * perm;
      float* d = C + (long)i * ldc + (long)c * S + sg;
#pragma unroll
      for (int e = 0; e < 8; ++e) d[(long)e * S] = overwrite ? a[e] : d[(long)e * S] + a[e];
      return;
    }
    float* dst = C + (long)i * ldc + j;
    if ((ldc & 3) == 0) {
      f32x4 lo = {a[0], a[1], a[2], a[3]}, hi = {a[4], a[5], a[6], a[7]};
      if (!overwrite) { lo += *(const f32x4*)dst; hi += *(const f32x4*)(dst + 4); }
      *(f32x4*)dst = lo;
      *(f32x4*)(dst + 4) = hi;
    } else {
      for (int e = 0; e < 8; ++e) dst[e] = overwrite ? a[e] : dst[e] + a[e];
    }
    return;
  }
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= (long)Ni * Nj) return;
  const int i = idx / Nj, j = idx % Nj;           // Nj % 4 == 0
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) a += *(const f32x4*)(slabs + (long)s * Ni * Nj + idx);
  if (perm > 0) {
    const int S = Nj / perm, sg = j / perm, c = j - sg * perm;
    float* d = C + (long)i * ldc + (long)c * S + sg;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[(long)e * S] = overwrite ? a[e] : d[(long)e * S] + a[e];
    return;
  }
  float* dst = C + (long)i * ldc + j;
  if (overwrite) {
    if ((ldc & 3) == 0) *(f32x4*)dst = a;
    else { for (int e = 0; e < 4; ++e) dst[e] = a[e]; }
  } else if ((ldc & 3) == 0) { *(f32x4*)dst = *(const f32x4*)dst + a; }
  else { for (int e = 0; e < 4; ++e) dst[e] += a[e]; }
}

// ------------------------------------------------------------------------------------------------ C ABI
// Which kernel the launcher picked for the calling thread's most recent stswin_gemm_nt / stswin_gemm_tn call (codes in
// include/stswin_hip.h): the parity tests assert that the production shapes really run the production kernels.
static thread_local int g_last_variant[2] = {0, 0};
extern "C" int stswin_tuning_build() {
#ifdef STSWIN_TUNING
  return 1;
#else
  return 0;
#endif
}
extern "C" int stswin_last_variant(int family) { return family >= 0 && family < 2 ? g_last_variant[family] : -1; }

static int set_lds_once(const void* fn) {
  return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
}

// out[n] += sum_b part[b][n] over the `rows` 128-row blocks written by a GF_CS_PARTIAL gemm_nt, in a fixed order (one workgroup per
// 64 columns: 16 float4 lanes x 16 row lanes, the 16 lane sums added in lane order; one writer per address): deterministic bias
// gradients.  (= slab_fold_kernel of rowops.hip with slab stride N.)
extern "C" int stswin_cs_reduce(const float* partials, int M, int N, float* out, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (N % 4) return -1004;
  const int rows = 2 * ((M + 255) / 256);
  const int rc = stswin_fold_launch(partials, N, 0, rows, N, 1, out, nullptr, nullptr, 0, 1, 1, (hipStream_t)stream);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// ---- a handful of rows (nn.Conv2d 1x1 behind AdaptiveAvgPool2d(1), ASPP.py:43-46: M = frames of the batch) -------------------
// The tiled kernels run a 4-row product as one workgroup walking K alone (24 us for 4 x 512 x 1024).  Here a wave owns one output
// column: lanes split K in 16-byte pieces, up to 8 row accumulators, one wave reduction per row.  bf16, plain epilogue (bias, ReLU).
__global__ __launch_bounds__(256) void gemm_nt_rows_kernel(GemmNT p) {
  const int l = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= p.N) return;
  const bf16* A = (const bf16*)p.A;
  const bf16* Bn = (const bf16*)p.B + (long)n * p.ldb;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = l * 8; k < p.Kseg; k += 512) {
    const bf16x8 b = *(const bf16x8*)(Bn + k);
#pragma unroll
    for (int m = 0; m < 8; ++m)
      if (m < p.M) {
        const bf16x8 a = *(const bf16x8*)(A + (long)m * p.lda + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[m] += (float)a[e] * (float)b[e];
      }
  }
#pragma unroll
  for (int m = 0; m < 8; ++m)
    if (m < p.M) {
      float v = wave_sum(acc[m]);
      if (l == 0) {
        if (p.bias) v += p.bias[n];
        if (p.flags & GF_RELU) v = fmaxf(v, 0.f);
        ((bf16*)p.C)[(long)m * p.ldc + n] = (bf16)v;
      }
    }
}
// the matching weight gradient: C[i][j] (+)= sum over <= 8 rows m of At[m][i] * Bt[m][j] - a sum of outer products, one 16-byte
// piece of a C row per thread
__global__ __launch_bounds__(256) void gemm_tn_rows_kernel(const bf16* At, long lda, const bf16* Bt, long ldb, float* C, long ldc, int Mk, int Ni,
                                                           int Nj, int overwrite) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int pj = Nj / 4;
  if (idx >= (long)Ni * pj) return;
  const int i = (int)(idx / pj), j = (int)(idx % pj) * 4;
  f32x4 v = overwrite ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(C + (long)i * ldc + j);
  for (int m = 0; m < Mk; ++m) {
    const float a = (float)At[(long)m * lda + i];
    const bf16x4 b = *(const bf16x4*)(Bt + (long)m * ldb + j);
    v += f32x4{a * (float)b[0], a * (float)b[1], a * (float)b[2], a * (float)b[3]};
  }
  *(f32x4*)(C + (long)i * ldc + j) = v;
}

// ---- split-K form of gemm_nt for FEW output tiles and a long K (ASPP's dilated 3x3 convolutions: M = 4096, N = 512, K = 9 x 1024 -
// 32 tiles of 256 x 256 on 256 CUs; the 128x64-tile kernel that filled the CUs ran them at 500 TFLOP/s): the 256x256 ring kernel on
// a grid of tiles x splits, every split writing its fp32 partial tile into its own slab of the caller's workspace, and one pass
// that adds the slabs in split order (+ bias, ReLU) and stores the compute dtype.  Deterministic; same result as the unsplit
// kernel up to the association of the fp32 sum.
__global__ __launch_bounds__(256) void nt_splitk_combine_kernel(const float* ws, int splits, long slab, int M, int N, bf16* C, long ldc, const float* bias,
                                                                int relu) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int pn = N / 8;
  if (idx >= (long)M * pn) return;
  const int m = (int)(idx / pn), n = (int)(idx % pn) * 8;
  const float* src = ws + (long)m * N + n;
  f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
  for (int j = 1; j < splits; ++j) {
    a += *(const f32x4*)(src + j * slab);
    b += *(const f32x4*)(src + j * slab + 4);
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = (e < 4 ? a[e] : b[e - 4]) + (bias ? bias[n + e] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    o[e] = (bf16)v;
  }
  *(bf16x8*)(C + (long)m * ldc + n) = o;
}

__global__ __launch_bounds__(256) void nt_splitk_combine16_kernel(const bf16* ws, int splits, long slab, int M, int N, bf16* C, long ldc, const float* bias,
                                                                  int relu) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int pn = N / 8;
  if (idx >= (long)M * pn) return;
  const int m = (int)(idx / pn), n = (int)(idx % pn) * 8;
  const bf16* src = ws + (long)m * N + n;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < splits; ++j) {
    const bf16x8 v = *(const bf16x8*)(src + j * slab);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = a[e] + (bias ? bias[n + e] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    o[e] = (bf16)v;
  }
  *(bf16x8*)(C + (long)m * ldc + n) = o;
}

static int nt_splitk_plan(int M, int N, int Kseg, int S, int* qsplit_o) {
  if (M < 256 || N < 256 || N % 8 || Kseg % 32) return 0;
  const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  const int nt = S * (Kseg / 32);
  if (tiles > 32 || nt < 64) return 0;                   // (at 64 tiles = 4 splits the tiled kernels measured faster: 57 vs 62 us)
  int splits = (int)(256 / tiles);
  if (splits > nt / 16) splits = nt / 16;                // >= 16 stages per split
  if (splits < 2) return 0;
  const int qs = (nt + splits - 1) / splits;
  *qsplit_o = qs;
  return (nt + qs - 1) / qs;
}
/* floats of workspace stswin_gemm_nt_splitk needs, 0 = the shape is not a split-K candidate (use stswin_gemm_nt) */
extern "C" long stswin_gemm_nt_splitk_scratch(int M, int N, int Kseg, int S) {
  int qs = 0;
  const int splits = nt_splitk_plan(M, N, Kseg, S, &qs);
  return splits ? (long)splits * M * N : 0;
}
extern "C" int stswin_gemm_nt_splitk(const void* A, long lda, const int* a_rows, const void* B, long ldb, void* C, long ldc, const float* bias, int M,
                                     int N, int Kseg, int S, int relu, float* workspace, long workspace_floats, void* stream) {
  int qs = 0;
  const int splits = nt_splitk_plan(M, N, Kseg, S, &qs);
  if (!splits) return -1008;
  if (M <= 0 || N <= 0 || Kseg <= 0 || S <= 0 || lda % 8 || ldb % 8) return -1001;
  // the combine kernels store whole 16-byte row pieces: a column slice that starts off an 8-column boundary (or a row pitch that
  // is not a multiple of 8) is not a split-K candidate - the caller falls back to stswin_gemm_nt, whose epilogues handle it
  if (ldc % 8 || ((uintptr_t)C & 15)) return -1008;
  if (!workspace || workspace_floats < (long)splits * M * N) return -1009;
  if ((a_rows ? false : (long)M * lda * 2 > 0xFFFF0000L) || (long)N * ldb * 2 > 0xFFFF0000L) return -1008;
  const char* e16 = getenv("STSWIN_SPLITK_BF16");
  const int b16 = e16 && atoi(e16) ? 1 : 0;
  GemmNT p{A, lda, a_rows, B, ldb, workspace, N, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, M, N, Kseg, S, 1.0f, 0, b16 ? 0 : GF_OUT_F32, nullptr, qs};
  static const int attr = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  if (attr != 0) return -attr;
  const long tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  g_last_variant[0] = STSWIN_VAR_NT_SPLITK | (splits << 16);
  if (b16) hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>), dim3((unsigned)tiles, (unsigned)splits), dim3(512), 131072, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true>), dim3((unsigned)tiles, (unsigned)splits), dim3(512), 131072, (hipStream_t)stream, p);
  STSWIN_CHECK_LAUNCH();
  const long n = (long)M * (N / 8);
  if (b16) hipLaunchKernelGGL(nt_splitk_combine16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)workspace, splits, (long)M * N, M, N,
                     (bf16*)C, ldc, bias, relu);
  else hipLaunchKernelGGL(nt_splitk_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, splits, (long)M * N, M, N,
                     (bf16*)C, ldc, bias, relu);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

/* The QKV projection of a Swin block with an fp8 (OCP e4m3) result (BASELINE.json configs[4]; swin_512.py:115-121):
 * out8[m][n] = e4m3((sum_k A[a_rows[m]][k] * B[n][k] + bias[n]) * (n < scale_cols ? scale : 1) / s), s = scales[m / rows_per_problem][n / head_dim]
 * = (largest |value| of that rows_per_problem x head_dim block) / 448 (1 for an all-zero block).  bf16 operands; M % 256 == 0,
 * N % 256 == 0, Kseg % 32 == 0, rows_per_problem in {32, 128}, head_dim in {128, 256}, scale_cols % head_dim == 0, ld8 % 16 == 0.
 * 256x256 ring kernel (as stswin_gemm_nt picks for these shapes), register epilogue. */
extern "C" int stswin_gemm_nt_qkv_fp8(const void* A, long lda, const int* a_rows, const void* B, long ldb, void* out8, long ld8,
                                      float* scales, long ld_scales, const float* bias, int M, int N, int Kseg, float scale, int scale_cols,
                                      int rows_per_problem, int head_dim, void* stream) {
  if (M <= 0 || N <= 0 || M % 256 || N % 256 || Kseg <= 0 || Kseg % 32 || lda % 8 || ldb % 8) return -1001;
  if ((rows_per_problem != 32 && rows_per_problem != 128) || (head_dim != 128 && head_dim != 256) || scale_cols % head_dim || N % head_dim)
    return -1010;
  if (ld8 % 16 || ((uintptr_t)out8 & 15) || !scales || ld_scales < N / head_dim || (bias && ((uintptr_t)bias & 15))) return -1011;
  if ((!a_rows && (long)M * lda * 2 > 0xFFFF0000L) || (long)N * ldb * 2 > 0xFFFF0000L) return -1008;
  GemmNT p{A, lda, a_rows, B, ldb, out8, ld8, nullptr, scales, ld_scales, bias, nullptr, 0, nullptr, M, N, Kseg, 1, scale, scale_cols, 0, nullptr, 0,
           rows_per_problem, head_dim};
  static const int attr = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  if (attr != 0) return -attr;
  const long tiles = (long)(M / 256) * (N / 256);
  g_last_variant[0] = STSWIN_VAR_NT_RING256_REGEPI;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>), dim3((unsigned)tiles), dim3(512), 131072, (hipStream_t)stream, p);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_gemm_nt(int dtype, const void* A, long lda, const int* a_rows, const void* B, long ldb,
                              void* C, long ldc, const int* c_rows, void* C2, long ldc2, const float* bias,
                              const void* R, long ldr, const int* r_rows, int M, int N, int Kseg, int S,
                              float scale, int scale_cols, int flags, float* colsum, void* stream) {
  if (M <= 0 || N <= 0) return 0;
#ifndef STSWIN_TUNING
  // the stream / duo / mid / half / nopipe kernels (measured slower everywhere, kept for A/B runs) exist in STSWIN_TUNING builds
  // only: the product library ignores their flags (stswin_tuning_build() tells a caller which library it has)
  flags &= ~(GF_MID | GF_HALF | GF_DUO | GF_STREAM | GF_NOPIPE | GF_W4R | GF_M32PP | GF_ROT | GF_DEBUG_BITS);
#endif
  const int bk = dtype == 0 ? 64 : 32;
  if (Kseg <= 0 || Kseg % bk || S <= 0) return -1001;
  if ((flags & GF_ACCUM) && !(flags & GF_OUT_F32)) return -1002;
  if ((flags & GF_CS_SQ) && (!(flags & GF_CS_PARTIAL) || !colsum || (flags & (GF_MID | GF_HALF | GF_DUO | GF_STREAM | GF_NOREGEPI))))
    return -1006;                                            // squares need the per-block table and a kernel that writes them
  if ((flags & GF_CS_PARTIAL) && colsum && ((M + 127) / 128) % 2) { // 128-row tiles never touch the last row of the [2*ceil(M/256)][N] table
    stswin_zero_bytes(colsum + (long)(2 * ((M + 255) / 256) - 1) * N, sizeof(float) * N, (hipStream_t)stream);   // (a kernel: see common.h)
    if (flags & GF_CS_SQ)
      stswin_zero_bytes(colsum + (long)(4 * ((M + 255) / 256) - 1) * N, sizeof(float) * N, (hipStream_t)stream);
  }
  GemmNT p{A, lda, a_rows, B, ldb, C, ldc, c_rows, C2, ldc2, bias, R, ldr, r_rows, M, N, Kseg, S, scale, scale_cols, flags, colsum};
  if (dtype == 0 && M <= 8 && S == 1 && !a_rows && !c_rows && !C2 && !R && !colsum && !(flags & ~GF_RELU) && scale_cols == 0 && Kseg % 8 == 0 &&
      lda % 8 == 0 && ldb % 8 == 0) {
    g_last_variant[0] = STSWIN_VAR_NT_ROWS;
    hipLaunchKernelGGL(gemm_nt_rows_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  const int nblk = ((M + 127) / 128) * ((N + 127) / 128);
  static int once = set_lds_once((const void*)gemm_nt_kernel<bf16, 4>) | set_lds_once((const void*)gemm_nt_kernel<float, 4>) |
                    set_lds_once((const void*)gemm_nt_kernel<bf16, 8>) | set_lds_once((const void*)gemm_nt_kernel<float, 8>);
  (void)once;
  const bool w8 = (flags & GF_WAVES4) == 0;
  // 256x256 ping-pong ring kernel (1 workgroup per CU): faster than the 128x128 kernel wherever its tile count fills
  // whole rounds of the 256 CUs (measured +4..24 %, profiles/r01_v7_gemm_shapes.txt); with a ragged last round
  // (e.g. 128 or 384 tiles) the 2-workgroups-per-CU 128x128 kernel wins.
  const long big_tiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  const long rounds = (big_tiles + 255) / 256;
  const bool fills = big_tiles * 10 >= rounds * 256 * 9;          // >= 90 % of the last round's slots used
  // The ring kernels address their operands with 32-bit byte offsets (buffer descriptors): a plain operand beyond 4 GB goes
  // to the 64-bit-addressed 128x128 family instead (gathered rows are only known on the device: those trap in the kernel).
  const bool fits32 = (a_rows || (long)M * lda * 2 <= 0xFFFF0000L) && (long)N * ldb * 2 <= 0xFFFF0000L;
  bool big = dtype == 0 && Kseg % 32 == 0 && N >= 256 && M >= 256 && fills && fits32 && !(flags & (GF_NOBIG | GF_WAVES4));
  if ((flags & GF_BIG) && dtype == 0 && Kseg % 32 == 0 && fits32) big = true;
  // sums of squares exist in the ring kernel's register epilogue only; an output that cannot take it (a pointer or pitch that is
  // not 16-byte aligned: a column slice of a wider buffer) goes to the 128x128 family, whose epilogue has them for any alignment
  const bool regepi_ok = !(flags & (GF_OUT_F32 | GF_ACCUM | GF_NOREGEPI)) && N % 8 == 0 && ldc % 8 == 0 && (!C2 || ldc2 % 8 == 0) &&
                         (!R || ldr % 8 == 0) && ((uintptr_t)C % 16 == 0) && (!C2 || (uintptr_t)C2 % 16 == 0) &&
                         (!R || (uintptr_t)R % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
  if ((flags & GF_CS_SQ) && !regepi_ok && !(flags & GF_BIG)) big = false;
  const bool mid_ok = dtype == 0 && Kseg % 32 == 0 && fits32 && !(flags & (GF_NOBIG | GF_WAVES4));
  const bool mid = (flags & GF_MID) && mid_ok;
  // 256x128 ping-pong ring (twice the tiles of the 256x256 one): for shapes whose 256x256 grid leaves CUs idle
  const long half_tiles = (long)((M + 255) / 256) * ((N + 127) / 128);
  const long hrounds = (half_tiles + 255) / 256;
  const bool hfills = half_tiles * 10 >= hrounds * 256 * 9;
  // measured slower than the 8-wave 128x128 kernel on every shape tried (128x32 wave tiles: 10 fragment reads per 16
  // MFMAs) - kept as a forced tuning option only
  (void)hfills;
#ifdef STSWIN_TUNING
  const bool half = (flags & GF_HALF) && mid_ok && !mid;
  if (half && !(flags & GF_BIG)) {
    static int once_half = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 128, 2, 4, 4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)once_half;
    g_last_variant[0] = STSWIN_VAR_NT_RING256x128_PP;
    hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 128, 2, 4, 4, 2, true>), dim3((unsigned)half_tiles), dim3(512), 98304, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
#endif
  if (big && !mid) {
    static int once_big = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)once_big;
    // register epilogue (operand-swapped MFMA): bf16 output, every row piece 16-byte aligned, R rows copied by 16-byte LDS-DMA
    const bool regepi = !(flags & (GF_OUT_F32 | GF_ACCUM | GF_NOREGEPI)) && N % 8 == 0 && ldc % 8 == 0 && (!C2 || ldc2 % 8 == 0) &&
                        (!R || ldr % 8 == 0) && ((uintptr_t)C % 16 == 0) && (!C2 || (uintptr_t)C2 % 16 == 0) &&
                        (!R || (uintptr_t)R % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
    static int once_swap = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    (void)once_swap;
    if ((flags & GF_CS_SQ) && (!regepi || (flags & GF_NOPIPE))) return -1006;   // (only the register epilogue sums squares)
#ifdef STSWIN_TUNING
    // "duo" (round 5, gemm_nt_ring_kernel<128, 256, 1, 4, 3, 2, 7, true>): 128x256 tiles, 4 self-pipelined waves, 72 KB of LDS - two
    // independent workgroups per CU, one multiplying while the other runs its epilogue; see the kernel's PIPE 7 branch.  Measured
    // (profiles/r05_epi_sweep_duo.txt, r05_duo_in_step_ab.txt): 5 % faster than the 256x256 ring on the fc1 + GELU + GELU' launch in a
    // back-to-back loop, 7-13 % slower with an R operand or a plain epilogue (two 128x256 tiles stream 1.5x the operand bytes per flop
    // through L2: ~20 TB/s at full MFMA rate), and 0.3-1.3 % SLOWER inside the training step - a tuning build variant, not a product path.
    if (regepi && (flags & GF_DUO) && !(flags & GF_NOPIPE)) {
      static int once_duo7 = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<128, 256, 1, 4, 3, 2, 7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
      (void)once_duo7;
      const long duo_tiles = (long)((M + 127) / 128) * ((N + 255) / 256);
      g_last_variant[0] = STSWIN_VAR_NT_DUO;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<128, 256, 1, 4, 3, 2, 7, true>), dim3((unsigned)duo_tiles), dim3(256), 73728, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
#endif
#ifdef STSWIN_TUNING
    // persistent streaming variant: plain Linear GEMMs (S = 1, no A gather) with the register epilogue
    const bool streamk = regepi && S == 1 && !a_rows && (flags & GF_STREAM) && !(flags & (GF_NOPIPE | GF_NOSTREAM));
    if (streamk) {
      static int once_stream = (int)hipFuncSetAttribute((const void*)gemm_nt_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)once_stream;
      const unsigned grid = (unsigned)(big_tiles < 256 ? big_tiles : 256);
      g_last_variant[0] = STSWIN_VAR_NT_STREAM;
      hipLaunchKernelGGL(gemm_nt_stream_kernel, dim3(grid), dim3(512), 163840, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    // round-2 "duo" (PIPE 0 main loop; GF_DUO | GF_NOPIPE): kept for A/B against the self-pipelined one below
    if (regepi && (flags & GF_DUO) && (flags & GF_NOPIPE)) {
      static int once_duo = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<128, 256, 1, 4, 3, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
      (void)once_duo;
      const long duo_tiles = (long)((M + 127) / 128) * ((N + 255) / 256);
      g_last_variant[0] = STSWIN_VAR_NT_DUO;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<128, 256, 1, 4, 3, 2, false, true>), dim3((unsigned)duo_tiles), dim3(256), 73728, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    if (flags & GF_NOPIPE) {
      static int once_np = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_np;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_NOPIPE;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, false>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
#endif
#ifdef STSWIN_TUNING
    // 4 waves of 128x128 (one per SIMD, 32x32x16 MFMA, 256 accumulator registers), register-pipelined: 16 fragment reads per stage
    // instead of the 8-wave kernel's 24 - and measured SLOWER (profiles/r04_gemm_w4_experiment.txt): with nobody else on the SIMD
    // every LDS-DMA piece costs the wave ~57 cycles of issue (8 per stage = 0.26 us of a 0.87 us stage, whatever their placement),
    // which the ping-pong partner otherwise hides; and half the threads run the same epilogue.
    static const int env_w4 = getenv("STSWIN_NT_W4") ? atoi(getenv("STSWIN_NT_W4")) : 0;
    if (regepi && (env_w4 == 6 || ((flags & GF_ROT) && (flags & GF_W4R)))) {   // 8 waves, software-pipelined, no phases
      static int once_swp = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_swp;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_W4 + 4;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 6, true>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    if (regepi && (env_w4 == 5 || (flags & GF_ROT))) {     // rotated ping-pong: one barrier per stage
      static int once_rot = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_rot;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_W4 + 3;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 5, true>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    if (regepi && (env_w4 == 3 || ((flags & GF_M32PP) && (flags & GF_W4R)))) {   // 4 waves, register-staged (no LDS-DMA)
      static int once_rs = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 2, 4, 1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_rs;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_W4 + 2;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 2, 4, 1, 4, true>), dim3((unsigned)big_tiles), dim3(256), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    if (regepi && (env_w4 == 2 || (flags & GF_M32PP))) {   // the 8-wave ping-pong schedule on 32x32x16 MFMA tiles (same wave tile, same LDS traffic)
      static int once_m32 = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_m32;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_W4 + 1;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 3, true>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
    if (regepi && ((flags & GF_W4R) || env_w4)) {
      static int once_w4 = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 2, 4, 1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_w4;
      g_last_variant[0] = STSWIN_VAR_NT_RING256_W4;
      hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 2, 4, 1, 2, true>), dim3((unsigned)big_tiles), dim3(256), 131072, (hipStream_t)stream, p);
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
#endif
    // start-time stagger of the first round (see the kernel), OPT-IN (STSWIN_NT_STAGGER=1; = N > 1 forces N ticks of 10 ns per phase
    // whatever the shape; read per call): launches of >= 6 tiles per CU whose tiles are short (<= 24 stages of 32) and end in a register
    // epilogue, spread = 0.7 x the estimated tile time over 8 phases.  Measured (profiles/r05_stagger_*.txt): behind an HBM-bound
    // spacer kernel -11 .. -19 % on the fc1 / fc2-gradient launches of stage 1; INSIDE the training step 216.0 -> 214.5 us and
    // 180.8 -> 180.4 us per launch, 604.9 -> 604.5 frames/s over three alternating runs - the start delay it pays is what it
    // gains.  Off by default.
    if (regepi) {
      const int nt_ = S * (Kseg / 32);
      const char* es = getenv("STSWIN_NT_STAGGER");
      const int ev = es ? atoi(es) : 0;
      static const int min_rounds = getenv("STSWIN_NT_STAGGER_MIN_ROUNDS") ? atoi(getenv("STSWIN_NT_STAGGER_MIN_ROUNDS")) : 6;
      static const int max_nt = getenv("STSWIN_NT_STAGGER_MAX_NT") ? atoi(getenv("STSWIN_NT_STAGGER_MAX_NT")) : 24;
      if (ev > 1) p.stagger_ticks = ev;
      else if (ev == 1 && big_tiles >= (long)min_rounds * 256 && nt_ <= max_nt) {
        const float epi_us = C2 ? 12.f : ((flags & (GF_GELU | GF_RESID | GF_MUL_R | GF_MUL_DGELU)) ? 6.f : 3.f);
        p.stagger_ticks = (int)(0.7f * (0.75f * nt_ + epi_us) * 100.f / 8.f);
      }
    }
    g_last_variant[0] = regepi ? STSWIN_VAR_NT_RING256_REGEPI : STSWIN_VAR_NT_RING256_LDSEPI;
#ifdef STSWIN_TUNING
    {
      const char* ep = getenv("STSWIN_NT_PERSIST");       // experiment (read per call): walk the tiles with a grid of 256 workgroups
      if (regepi && ep && atoi(ep) > 0 && big_tiles > 256 && S == 1) {
        static int once_ps = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)once_ps;
        p.persist = 1;
        hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, 1, true, true>), dim3(256), dim3(512), 131072, (hipStream_t)stream, p);
        STSWIN_CHECK_LAUNCH();
        return 0;
      }
    }
#endif
    if (regepi) hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true, true>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 256, 2, 4, 4, 2, true>), dim3((unsigned)big_tiles), dim3(512), 131072, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
#ifdef STSWIN_TUNING
  if (mid) {
    static int once_mid = (int)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<256, 128, 4, 2, 3, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    (void)once_mid;
    const long mid_tiles = (long)((M + 255) / 256) * ((N + 127) / 128);
    g_last_variant[0] = STSWIN_VAR_NT_MID;
    hipLaunchKernelGGL((gemm_nt_ring_kernel<256, 128, 4, 2, 3, 4, false>), dim3((unsigned)mid_tiles), dim3(512), 73728, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
#endif
  // narrow AND short (the decode head's 1x1 convolutions to 48 / 64 channels: M = 4096 .. 16384): 256-row tiles would be 16 .. 64
  // workgroups on 256 CUs - 128x64 tiles double them and two fit a CU
  if (dtype == 0 && N <= 64 && M >= 128 && w8 && !(flags & GF_NONARROW) && (M + 255) / 256 <= 128) {
    static int once_ns = (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152) |
                         (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 128, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)once_ns;
    const int nb = ((M + 127) / 128) * ((N + 63) / 64);
    g_last_variant[0] = STSWIN_VAR_NT_128x64;
    if (!(flags & GF_NODEEP) && (long)S * Kseg >= 1024)
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 128, 64, 4>), dim3(nb), dim3(512), 98304, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 128, 64>), dim3(nb), dim3(512), 49152, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  // narrow outputs (ResNet stem / layer1: N = 64): 256x64 tiles instead of 128x128 ones whose second half multiplies zeros
  if (N <= 64 && M >= 256 && w8 && !(flags & GF_NONARROW)) {
    static int once_n = (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920) |
                        (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<float, 8, 256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    (void)once_n;
    const int nb = ((M + 255) / 256) * ((N + 63) / 64);
    g_last_variant[0] = STSWIN_VAR_NT_256x64 + (dtype ? STSWIN_VAR_F32 : 0);
    if (dtype == 0 && (flags & GF_DEEP)) {
      static int once_nd = (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 256, 64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
      (void)once_nd;
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 256, 64, 3>), dim3(nb), dim3(512), 122880, (hipStream_t)stream, p);
    } else if (dtype == 0) hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 256, 64>), dim3(nb), dim3(512), 81920, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, 8, 256, 64>), dim3(nb), dim3(512), 81920, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  // few 128x128 tiles (ASPP's dilated 3x3 convolutions: M = 4096 -> 128 tiles on 256 CUs, K = 9216): 128x64 tiles double
  // the workgroups; each CU can hold two of them
  if (dtype == 0 && w8 && nblk <= 160 && N >= 128 && M >= 128 && !(flags & GF_NONARROW)) {
    static int once_s = (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 128, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152) |
                        (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 128, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)once_s;
    const int nb = ((M + 127) / 128) * ((N + 63) / 64);
    g_last_variant[0] = STSWIN_VAR_NT_128x64;
    // <= 1 workgroup per CU and a long K: nothing but its own prefetch depth hides the memory latency -> 4-stage ring
    if (!(flags & GF_NODEEP) && (long)S * Kseg >= 1024)
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 128, 64, 4>), dim3(nb), dim3(512), 98304, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 128, 64>), dim3(nb), dim3(512), 49152, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == 0 && w8 && (flags & GF_DEEP)) {
    static int once_d = (int)hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16, 8, 128, 128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)once_d;
    g_last_variant[0] = STSWIN_VAR_NT_128x128;
    hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8, 128, 128, 3>), dim3(nblk), dim3(512), 98304, (hipStream_t)stream, p);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  g_last_variant[0] = (w8 ? STSWIN_VAR_NT_128x128 : STSWIN_VAR_NT_128x128_W4) + (dtype ? STSWIN_VAR_F32 : 0);
  if (dtype == 0) {
    if (w8) hipLaunchKernelGGL((gemm_nt_kernel<bf16, 8>), dim3(nblk), dim3(512), 65536, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<bf16, 4>), dim3(nblk), dim3(256), 65536, (hipStream_t)stream, p);
  } else {
    if (w8) hipLaunchKernelGGL((gemm_nt_kernel<float, 8>), dim3(nblk), dim3(512), 65536, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<float, 4>), dim3(nblk), dim3(256), 65536, (hipStream_t)stream, p);
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// Workgroups of gemm_tn_ring_kernel (128 KB of LDS: one per CU) that are resident at the same time = compute units of the device.
static long tn_resident_workgroups() {
  static thread_local int dev_cached = -1;
  static thread_local long cus = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != dev_cached) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
    dev_cached = dev;
  }
  return cus;
}
// Arrival / ticket / departure counters of the fused split-K combine: [3][tiles] unsigned per launch, zero when a launch starts and
// zeroed again by the last workgroup of each tile to leave.  One region per (device, stream): launches that share a region are
// ordered by their stream, two host threads or two streams never share one (round-4 advisor: the round-robin hand-out could give two
// concurrent launches the same counters).  Up to 16 streams per device; a 17th takes the unfused path.  Allocated and cleared once
// per device, outside any stream capture (the first fused launch of a process is a warm-up launch; if it is not, the unfused path
// is taken).  A captured hipGraph node keeps the region of the stream it was captured on: replay it on that stream.
#include <atomic>
// Zero fill of an [Ni][Nj] fp32 result with row pitch ldc, as a KERNEL.  Round 6: this used to be hipMemset2DAsync; inside a hipGraph
// (ROCm 7.0 / 7.2) the captured 2-D memset node does not reproduce what the eager call does - replays of a weight-gradient GEMM that
// accumulates into the filled buffer (overwrite without split-K slabs: ASPP's dilated convolutions at 16x16 / 32x32 maps) left a quarter
// to most of the result on stale or garbage values (tools/probes/tn_graph_repro.py: 78 of 120 replayed launches differed from the eager
// ones, by up to 1e25), which is what sent a graph-replayed training run to NaN after ~130 steps while 20-step comparisons passed.
__global__ __launch_bounds__(256) void tn_zero_fill_kernel(float* __restrict__ C, long ldc, int Ni, int Nj) {
  const long n4 = (long)Ni * ((Nj + 3) / 4);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long r = i / ((Nj + 3) / 4);
    const int c = (int)(i - r * ((Nj + 3) / 4)) * 4;
    float* q = C + r * ldc + c;
    if (c + 3 < Nj && ((((uintptr_t)q) & 15) == 0)) *(f32x4*)q = (f32x4){0.f, 0.f, 0.f, 0.f};
    else
      for (int e = 0; e < 4 && c + e < Nj; ++e) q[e] = 0.f;
  }
}
static void tn_zero_fill(float* C, long ldc, int Ni, int Nj, hipStream_t stream) {
  const long n4 = (long)Ni * ((Nj + 3) / 4);
  long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(tn_zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, C, ldc, Ni, Nj);
}

#include <mutex>
static std::mutex g_tn_mutex;
static std::atomic<int> g_tn_fused_holds{0};
static unsigned* tn_tile_counters(int tiles, hipStream_t stream) {
  constexpr int REGIONS = 16, PER = 3 * 1024;
  static unsigned* bufs[64] = {nullptr};
  static hipStream_t owners[64][REGIONS];
  static int nown[64] = {0};
  int dev = 0;
  if (tiles > 1024 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_tn_mutex);
  if (!bufs[dev]) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    unsigned* b = nullptr;
    if (hipMalloc((void**)&b, sizeof(unsigned) * REGIONS * PER) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemset(b, 0, sizeof(unsigned) * REGIONS * PER) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(b); return nullptr; }
    bufs[dev] = b;
  }
  int r = -1;
  for (int i = 0; i < nown[dev]; ++i)
    if (owners[dev][i] == stream) { r = i; break; }
  if (r < 0) {
    if (nown[dev] >= REGIONS) return nullptr;
    r = nown[dev]++;
    owners[dev][r] = stream;
  }
  return bufs[dev] + (size_t)r * PER;
}
// Compute units the persistent / one-workgroup-per-CU launches plan for (gemm_tn ring: tiles x splits; attention backward: persistent
// workgroups): all of the device by default; stswin_set_cu_budget(n) lowers it to n while a communication kernel is known to hold the
// rest (a grid of 256 on 256 - k free units runs as one round + a tail round of k workgroups).  0 restores the default.
static std::atomic<int> g_cu_budget{0};
extern "C" int stswin_set_cu_budget(int cus) {
  const int old = g_cu_budget.load();
  g_cu_budget.store(cus > 0 ? cus : 0);
  return old;
}
int stswin_cu_budget() {
  const int b = g_cu_budget.load();
  return b > 0 ? b : 256;
}

// Process-wide switch of the fused combine: every hold (+1) turns it off until released (-1).  GradBucketReducer holds it while its
// collectives overlap backward (stswincl_amd/dp.py).  STSWIN_TN_FUSED=0 / 1 in the environment overrides the holds (read per call).
extern "C" int stswin_tn_fused_hold(int delta) {
  if (delta > 0) return g_tn_fused_holds.fetch_add(1) + 1;
  if (delta < 0) {
    int cur = g_tn_fused_holds.load();
    while (cur > 0 && !g_tn_fused_holds.compare_exchange_weak(cur, cur - 1)) {}
    return cur > 0 ? cur - 1 : 0;
  }
  return g_tn_fused_holds.load();
}

// Several bf16 weight-gradient problems in one launch of gemm_tn_ring_group_kernel (see there).  Every problem must be one the ring
// kernel with the fused combine takes on its own (both output dims >= 256, at most one gather map, 32-bit operand offsets, bf16
// partials); the split counts are chosen so that every workgroup of the launch gets about the same number of 32-row stages and the
// whole grid is at most one workgroup per compute unit.  Returns 0 when launched, STSWIN_TN_GROUP_DECLINED (-1050) when the set is not
// eligible or would fill less than 7/8 of the device (the caller then launches the problems one by one - nothing has been written),
// another negative code for malformed arguments.  `splits_out` (optional, [count]): the split counts used.
typedef stswin_tn_problem StswinTnProblem;       // include/stswin_hip.h
extern "C" int stswin_gemm_tn_group(int dtype, int count, const StswinTnProblem* pr, float* workspace, long workspace_floats,
                                    int* splits_out, void* stream) {
  if (count <= 0) return 0;
  if (count > TN_GROUP_MAX || !pr) return -1052;
  const int declined = -1050;
  if (dtype != 0 || !workspace) return declined;
  { const char* f32_env = getenv("STSWIN_TN_F32_SLABS"); if (f32_env && atoi(f32_env)) return declined; }
  { const char* ef = getenv("STSWIN_TN_FUSED"); if (ef ? atoi(ef) == 0 : g_tn_fused_holds.load() != 0) return declined; }
  { const char* eg = getenv("STSWIN_TN_GROUP"); if (eg && atoi(eg) == 0) return declined; }          // A/B switch, read per call
  int tiles[TN_GROUP_MAX], nst[TN_GROUP_MAX], rs[TN_GROUP_MAX];
  long tile_stages = 0, total_tiles = 0;
  for (int i = 0; i < count; ++i) {
    const StswinTnProblem& q = pr[i];
    if (q.Mk <= 0 || q.Ni <= 0 || q.Nj <= 0) return -1052;
    if (q.Ni % 8 || q.Nj % 8 || q.lda % 8 || q.ldb % 8) return -1003;
    if (q.bseg < 0 || (q.bseg > 0 && (q.bseg % 8 || !q.bt_rows || q.Nj % q.bseg))) return -1004;
    if (q.Ni < 256 || q.Nj < 256 || q.ldc < 0 || (q.at_rows && q.bt_rows) || ((q.at_rows || q.bt_rows) && q.Mk % 32 != 0)) return declined;
    if ((!q.at_rows && (long)q.Mk * q.lda * 2 > 0xFFFF0000L) || (!q.bt_rows && (long)q.Mk * q.ldb * 2 > 0xFFFF0000L)) return declined;
    if (q.bseg > 0)
      for (int j0 = 0; j0 < q.Nj; j0 += 256)
        if ((j0 + 255 < q.Nj ? j0 + 255 : q.Nj - 1) / q.bseg - j0 / q.bseg > 1) return declined;
    if (q.tapminor && q.bseg <= 0) return -1052;
    tiles[i] = ((q.Ni + 255) / 256) * ((q.Nj + 255) / 256);
    nst[i] = (q.Mk + 31) / 32;
    tile_stages += (long)tiles[i] * nst[i];
    total_tiles += tiles[i];
  }
  if (total_tiles > 1024) return declined;
  const int cus = stswin_cu_budget();
  if (cus > tn_resident_workgroups()) return declined;
  // stages per workgroup: the smallest count with which the grid fits the budget (>= 16: below that the ring never reaches steady state)
  // A workgroup of a problem with a row map runs its stages ~8 % slower than a plain one beside it (0.83 against 0.77 us;
  // profiles/r05_tn_group_timeline_static.txt) and the launch ends with its slowest workgroup: such a problem's stage count is weighted
  // by 1.05 in the split choice (STSWIN_TN_GROUP_W, percent, tuning; 100-110 measure the same in the step, 120 is slower)
  long target = (tile_stages + cus - 1) / cus;
  if (target < 16) target = 16;
  const char* ew = getenv("STSWIN_TN_GROUP_W");
  const long wmap = ew && atoi(ew) >= 100 && atoi(ew) <= 300 ? atoi(ew) : 105;
  long total = 0;
  for (int guard = 0; guard < 4096; ++guard, ++target) {
    total = 0;
    for (int i = 0; i < count; ++i) {
      const long wi = (pr[i].at_rows || pr[i].bt_rows) ? wmap : 100;
      int r = (int)((nst[i] * wi + target * 100 - 1) / (target * 100));
      if (r < 1) r = 1;
      const int per = (nst[i] + r - 1) / r;
      rs[i] = (nst[i] + per - 1) / per;              // no empty split (its slab would never be written)
      total += (long)tiles[i] * rs[i];
    }
    if (total <= cus) break;
  }
  if (total > cus || total * 8 < (long)cus * 7) return declined;
  {   // balance: the launch ends with its longest workgroup - decline a plan whose longest (weighted) stage count is more than 15 %
      // above the mean over the device (stage 2 of the Swin stack, all four weight gradients: 192 tiles -> splits 1 / 1 / 2 / 2,
      // 389 us against 117 + 230 for fc2 alone + the other three; profiles/r05_tn_group4_in_step_ab.txt)
    long wsum = 0, wmax = 0;
    for (int i = 0; i < count; ++i) {
      const long wi = (pr[i].at_rows || pr[i].bt_rows) ? wmap : 100;
      const long per = (nst[i] + rs[i] - 1) / rs[i];
      wsum += (long)tiles[i] * nst[i] * wi;
      if (per * wi > wmax) wmax = per * wi;
    }
    if (wmax * cus * 100 > wsum * 115) return declined;
  }
  long ws_need = 0;
  for (int i = 0; i < count; ++i) {
    if ((long)rs[i] * pr[i].Ni * pr[i].Nj * 2 >= 0x7FFFFFF0L) return declined;
    ws_need += ((long)rs[i] * pr[i].Ni * pr[i].Nj + 1) / 2 + 64;       // bf16 partials, in floats, each problem's block 256-byte aligned
    ws_need = (ws_need + 63) / 64 * 64;
  }
  if (ws_need > workspace_floats) return declined;
  unsigned* cnt = tn_tile_counters((int)total_tiles, (hipStream_t)stream);
  if (!cnt) return declined;
  GemmTNGroup g;
  g.count = count;
  g.first[0] = 0;
  long wo = 0, to = 0;
  for (int i = 0; i < count; ++i) {
    const StswinTnProblem& q = pr[i];
    const int perm = (q.tapminor && q.bseg > 0) ? q.bseg : 0;
    g.q[i] = GemmTN{q.At, q.lda, q.at_rows, q.Bt, q.ldb, q.bt_rows, q.C, q.ldc, q.Mk, q.Ni, q.Nj, rs[i], q.bseg, workspace + wo, 1,
                    cnt + 3 * to, q.overwrite ? 1 : 0, perm};
    g.mode[i] = q.at_rows ? 1 : (q.bt_rows ? (q.bseg > 0 ? 3 : 2) : 0);
    g.first[i + 1] = g.first[i] + tiles[i] * rs[i];
    wo += ((long)rs[i] * q.Ni * q.Nj + 1) / 2 + 64;
    wo = (wo + 63) / 64 * 64;
    to += tiles[i];
    if (splits_out) splits_out[i] = rs[i];
  }
  for (int i = count; i < TN_GROUP_MAX; ++i) { g.q[i] = g.q[0]; g.mode[i] = 0; g.first[i + 1] = g.first[count]; }
  static int once_g = (int)hipFuncSetAttribute((const void*)gemm_tn_ring_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                      (int)hipFuncSetAttribute((const void*)gemm_tn_ring_group_static_kernel<0, 1, 2, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                      (int)hipFuncSetAttribute((const void*)gemm_tn_ring_group_static_kernel<0, 0, -1, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                      (int)hipFuncSetAttribute((const void*)gemm_tn_ring_group_static_kernel<0, 0, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  (void)once_g;
  g_last_variant[1] = STSWIN_VAR_TN_RING_PLAIN | STSWIN_VAR_TN_SLABS_BF16 | STSWIN_VAR_TN_FUSED | (rs[0] << 16);
  // slot / mode combinations of the training step get the kernel with constant argument offsets (the problems keep their order:
  // the caller lists them plain, A-gathered, B-gathered)
  const bool m012 = count == 3 && g.mode[0] == 0 && g.mode[1] == 1 && g.mode[2] == 2;
  const bool m00 = count == 2 && g.mode[0] == 0 && g.mode[1] == 0;
  const bool m0012 = count == 4 && g.mode[0] == 0 && g.mode[1] == 0 && g.mode[2] == 1 && g.mode[3] == 2;
  static const char* egd = getenv("STSWIN_TN_GROUP_DYNAMIC");          // A/B switch: the dynamically indexed kernel for every set
  if (m012 && !(egd && atoi(egd))) hipLaunchKernelGGL((gemm_tn_ring_group_static_kernel<0, 1, 2, -1>), dim3((unsigned)total), dim3(512), 131072, (hipStream_t)stream, g);
  else if (m00 && !(egd && atoi(egd))) hipLaunchKernelGGL((gemm_tn_ring_group_static_kernel<0, 0, -1, -1>), dim3((unsigned)total), dim3(512), 131072, (hipStream_t)stream, g);
  else if (m0012 && !(egd && atoi(egd))) hipLaunchKernelGGL((gemm_tn_ring_group_static_kernel<0, 0, 1, 2>), dim3((unsigned)total), dim3(512), 131072, (hipStream_t)stream, g);
  else
  hipLaunchKernelGGL(gemm_tn_ring_group_kernel, dim3((unsigned)total), dim3(512), 131072, (hipStream_t)stream, g);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_gemm_tn(int dtype, const void* At, long lda, const int* at_rows, const void* Bt, long ldb,
                              const int* bt_rows, float* C, long ldc, int Mk, int Ni, int Nj, int splits, int bseg,
                              float* workspace, long workspace_floats, void* stream) {
  if (Mk <= 0 || Ni <= 0 || Nj <= 0) return 0;
  const int overwrite = (splits > 0 && (splits & (1 << 27))) ? 1 : 0;         // C = result instead of C += result
  // bit 26: leave the split-K partials in the workspace - the caller runs stswin_tn_combine itself (on another stream, so that
  // the combine overlaps the next GEMM instead of standing between two launches); stswin_last_variant(1) tells it how many
  const int no_combine = (splits > 0 && (splits & (1 << 26))) ? 1 : 0;
  if (splits > 0) splits &= ~(1 << 26);
  // bit 25 (STSWIN_TN_OUT_TAPMINOR, with bseg > 0): the split-K combine stores the result in [row][channel][tap] order (see
  // tn_reduce_kernel); honoured only where partial slabs are combined - stswin_last_variant(1) says whether it was (0x4000)
  const int perm = (splits > 0 && (splits & (1 << 25)) && bseg > 0 && !no_combine) ? bseg : 0;
  if (splits > 0) splits &= ~(1 << 25);
  // bf16 operands: the split-K partials are stored as bf16 (each the fp32 sum of Mk / splits products, rounded once; the
  // combine pass adds them in fp32) - half the slab traffic of the weight gradients, the relative rounding error of a
  // gradient is 2^-9 / sqrt(splits) (a bf16 autocast GEMM rounds its whole result once, 2^-9).  STSWIN_TN_F32_SLABS=1: fp32.
  const char* f32_env = getenv("STSWIN_TN_F32_SLABS");        // read per call: the parity tests flip it inside one process
  const int f32_slabs = f32_env && atoi(f32_env) ? 1 : 0;
  const int slab_bf16 = (dtype == 0 && !f32_slabs) ? 1 : 0;
  if (splits > 0) splits &= ~(1 << 27);
  const int splits_flags_w4 = (splits > 0 && (splits & (1 << 30))) ? 1 : 0;   // tuning: bit 30 selects the 4-wave variant
  if (splits > 0) splits &= ~(1 << 30);
  const int pack = dtype == 0 ? 8 : 4;
  if (Ni % pack || Nj % pack || lda % pack || ldb % pack) return -1003;
  if (bseg < 0 || (bseg > 0 && (bseg % pack || !bt_rows || Nj % bseg))) return -1004;
  if (dtype == 0 && Mk <= 8 && !at_rows && !bt_rows && bseg == 0 && ldc % 4 == 0) {            // a sum of <= 8 outer products
    g_last_variant[1] = STSWIN_VAR_TN_ROWS;
    const long nthr = (long)Ni * (Nj / 4);
    hipLaunchKernelGGL(gemm_tn_rows_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)At, lda, (const bf16*)Bt,
                       ldb, C, ldc, Mk, Ni, Nj, overwrite);
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  if (bseg > 0)                                            // the kernel keeps two tap index rows per 128-column tile
    for (int j0 = 0; j0 < Nj; j0 += 128)
      if ((j0 + 127 < Nj ? j0 + 127 : Nj - 1) / bseg - j0 / bseg > 1) return -1005;
  const int bm = dtype == 0 ? 64 : 32;
  const int ntile = (Mk + bm - 1) / bm;
  // 256x256 ping-pong ring (bf16): 1 workgroup per CU, so tiles x splits should be ~256, with >= 16 stages of 32 rows per
  // workgroup.  Chosen when both output dims are >= 256 (else the 128x128 kernel wastes less of the tile).
  // Tuning bits of `splits`: bit 29 forces the ring with the given split count, bit 28 forbids it.
  const bool force_ring = splits > 0 && (splits & (1 << 29));
  const bool no_ring = splits > 0 && (splits & (1 << 28));
  if (splits > 0) splits &= ~((1 << 29) | (1 << 28));
  bool ring = dtype == 0 && !splits_flags_w4 && !no_ring && (force_ring || (splits <= 0 && Ni >= 256 && Nj >= 256));
  if (ring && (at_rows || bt_rows) && ((at_rows && bt_rows) || Mk % 32 != 0)) ring = false;   // map modes of the ring kernel
  if (ring && ((!at_rows && (long)Mk * lda * 2 > 0xFFFF0000L) || (!bt_rows && (long)Mk * ldb * 2 > 0xFFFF0000L)))
    ring = false;                                          // 32-bit buffer offsets: plain operands beyond 4 GB take the 128x128 kernel
  if (ring && bseg > 0)
    for (int j0 = 0; j0 < Nj; j0 += 256)
      if ((j0 + 255 < Nj ? j0 + 255 : Nj - 1) / bseg - j0 / bseg > 1) ring = false;
  if (ring) {
    const int t256 = ((Ni + 255) / 256) * ((Nj + 255) / 256);
    const int nst = (Mk + 31) / 32;
    const int cus = stswin_cu_budget();               // 256 unless the caller lowered it (stswin_set_cu_budget)
    int rs = splits > 0 ? splits : cus / t256;
    if (splits <= 0) {
      const int max_by_k = nst / 16 > 0 ? nst / 16 : 1;
      if (rs > max_by_k) rs = max_by_k;
      if (rs < 1) rs = 1;
      while (rs > 1 && workspace && (long)rs * Ni * Nj > workspace_floats) --rs;
      const long blocks = (long)t256 * rs;
      if (blocks < cus * 11 / 16 || (blocks > cus && blocks % cus != 0 && blocks % cus < cus * 25 / 32)) ring = false;
    }
    if (rs > nst) rs = nst;
    { const int per = (nst + rs - 1) / rs; rs = (nst + per - 1) / per; }   // no empty split (its slab would never be written)
    if (ring) {
      const bool slabs = workspace && rs > 1 && (long)rs * Ni * Nj <= workspace_floats;
      if (overwrite && !slabs) tn_zero_fill(C, ldc, Ni, Nj, (hipStream_t)stream);
      // Fused split-K combine (gemm_tn_ring_kernel<MODE, true>): bf16 partials, a grid of at most one workgroup per CU (where the
      // distributed combine beats the separate pass; residency is NOT needed for correctness any more), byte offsets of the partials
      // within 31 bits.  Off while stswin_tn_fused_hold() holds are outstanding; STSWIN_TN_FUSED=0 / 1 overrides (read per call).
      unsigned* tile_cnt = nullptr;
      if (slabs && slab_bf16 && !no_combine && ldc >= 0 && (long)t256 * rs <= tn_resident_workgroups() && (long)rs * Ni * Nj * 2 < 0x7FFFFFF0L) {
        const char* ef = getenv("STSWIN_TN_FUSED");
        const bool on = ef ? atoi(ef) != 0 : g_tn_fused_holds.load() == 0;
        if (on) tile_cnt = tn_tile_counters(t256, (hipStream_t)stream);
      }
      GemmTN q{At, lda, at_rows, Bt, ldb, bt_rows, C, ldc, Mk, Ni, Nj, rs, bseg, slabs ? workspace : nullptr, slab_bf16, tile_cnt, overwrite, perm};
      static int once_r = (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072) |
                          (int)hipFuncSetAttribute((const void*)gemm_tn_ring_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      (void)once_r;
      const dim3 grid((unsigned)(t256 * rs));
      g_last_variant[1] = (at_rows ? STSWIN_VAR_TN_RING_ATROWS : (bt_rows && bseg > 0) ? STSWIN_VAR_TN_RING_BSEG : bt_rows ? STSWIN_VAR_TN_RING_BTROWS : STSWIN_VAR_TN_RING_PLAIN) |
                          (slabs ? (slab_bf16 ? STSWIN_VAR_TN_SLABS_BF16 : STSWIN_VAR_TN_SLABS_F32) : 0) | ((slabs && perm) ? 0x4000 : 0) | (rs << 16) |
                          (tile_cnt ? STSWIN_VAR_TN_FUSED : 0);
      if (tile_cnt) {
        if (at_rows) hipLaunchKernelGGL((gemm_tn_ring_kernel<1, true>), grid, dim3(512), 131072, (hipStream_t)stream, q);
        else if (bt_rows && bseg > 0) hipLaunchKernelGGL((gemm_tn_ring_kernel<3, true>), grid, dim3(512), 131072, (hipStream_t)stream, q);
        else if (bt_rows) hipLaunchKernelGGL((gemm_tn_ring_kernel<2, true>), grid, dim3(512), 131072, (hipStream_t)stream, q);
        else hipLaunchKernelGGL((gemm_tn_ring_kernel<0, true>), grid, dim3(512), 131072, (hipStream_t)stream, q);
        STSWIN_CHECK_LAUNCH();
        return 0;
      }
      if (at_rows) hipLaunchKernelGGL(gemm_tn_ring_kernel<1>, grid, dim3(512), 131072, (hipStream_t)stream, q);
      else if (bt_rows && bseg > 0) hipLaunchKernelGGL(gemm_tn_ring_kernel<3>, grid, dim3(512), 131072, (hipStream_t)stream, q);
      else if (bt_rows) hipLaunchKernelGGL(gemm_tn_ring_kernel<2>, grid, dim3(512), 131072, (hipStream_t)stream, q);
      else hipLaunchKernelGGL(gemm_tn_ring_kernel<0>, grid, dim3(512), 131072, (hipStream_t)stream, q);
      if (slabs && !no_combine) {
        const long n4 = slab_bf16 ? ((long)Ni * Nj + 7) / 8 : ((long)Ni * Nj + 3) / 4;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, C, ldc,
                           Ni, Nj, rs, overwrite, slab_bf16, perm);
      }
      STSWIN_CHECK_LAUNCH();
      return 0;
    }
  }
  if (splits <= 0) {   // 128x128 kernel, 2 resident workgroups per CU on 256 CUs: one round of <= 512 workgroups (a 513th
                       // costs a whole extra round) with >= 8 K tiles each; gathered operands hide their latency better
                       // with 2-4 rounds as long as a workgroup keeps >= 32 K tiles (tools/tn_sweep.py).
    const int tiles = ((Ni + 127) / 128) * ((Nj + 127) / 128);
    int rounds = 1;
    if (at_rows || bt_rows)
      for (int r = 4; r >= 2; --r)
        if (ntile / (r * 512 / tiles > 0 ? r * 512 / tiles : 1) >= 32) { rounds = r; break; }
    splits = rounds * 512 / tiles;
    if (splits < 1) splits = 1;
    const int max_by_k = ntile / 8 > 0 ? ntile / 8 : 1;
    if (splits > max_by_k) splits = max_by_k;
    if (splits >= 8) splits &= ~7;                        // whole splits per XCD
    else { int p2 = 1; while (p2 * 2 <= splits) p2 *= 2; splits = p2; }
  }
  if (splits > ntile) splits = ntile;
  // ceil(ntile / splits) tiles per split can leave the last splits WITHOUT a tile (259 tiles over 64 splits: 5 each, 12 splits
  // empty): such a workgroup returns before it writes its slab and the combine would add whatever the workspace held
  { const int per = (ntile + splits - 1) / splits; splits = (ntile + per - 1) / per; }
  const bool use_slabs = workspace && splits > 1 && (long)splits * Ni * Nj <= workspace_floats;
  if (overwrite && !use_slabs) tn_zero_fill(C, ldc, Ni, Nj, (hipStream_t)stream);
  GemmTN p{At, lda, at_rows, Bt, ldb, bt_rows, C, ldc, Mk, Ni, Nj, splits, bseg, use_slabs ? workspace : nullptr, slab_bf16, nullptr, 0, 0};
  dim3 grid(((Ni + 127) / 128) * ((Nj + 127) / 128) * splits);
  static int once = set_lds_once((const void*)gemm_tn_kernel<bf16, 4>) | set_lds_once((const void*)gemm_tn_kernel<float, 4>) |
                    set_lds_once((const void*)gemm_tn_kernel<bf16, 8>) | set_lds_once((const void*)gemm_tn_kernel<float, 8>);
  (void)once;
  const bool w8 = splits_flags_w4 == 0;
  g_last_variant[1] = ((w8 ? STSWIN_VAR_TN_128x128 : STSWIN_VAR_TN_128x128_W4) + (dtype ? STSWIN_VAR_F32 : 0)) |
                      (use_slabs ? (slab_bf16 ? STSWIN_VAR_TN_SLABS_BF16 : STSWIN_VAR_TN_SLABS_F32) : 0) | ((use_slabs && perm) ? 0x4000 : 0) |
                      (splits << 16);
  if (dtype == 0) {
    if (w8) hipLaunchKernelGGL((gemm_tn_kernel<bf16, 8>), grid, dim3(512), 65536, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<bf16, 4>), grid, dim3(256), 65536, (hipStream_t)stream, p);
  } else {
    if (w8) hipLaunchKernelGGL((gemm_tn_kernel<float, 8>), grid, dim3(512), 65536, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_tn_kernel<float, 4>), grid, dim3(256), 65536, (hipStream_t)stream, p);
  }
  if (use_slabs && !no_combine) {
    const long n4 = slab_bf16 ? ((long)Ni * Nj + 7) / 8 : ((long)Ni * Nj + 3) / 4;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, C, ldc,
                       Ni, Nj, splits, overwrite, slab_bf16, perm);
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// Second half of a stswin_gemm_tn launched with STSWIN_TN_NO_COMBINE: C (+)= sum of the `splits` partial results in `workspace`
// (bf16 or fp32 slabs, as stswin_last_variant(1) reported for that launch).
extern "C" int stswin_tn_combine(const float* workspace, float* C, long ldc, int Ni, int Nj, int splits, int overwrite, int slab_bf16,
                                 void* stream) {
  if (!workspace || !C || Ni <= 0 || Nj <= 0 || splits < 2) return -1007;
  const long n4 = slab_bf16 ? ((long)Ni * Nj + 7) / 8 : ((long)Ni * Nj + 3) / 4;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, workspace, C, ldc, Ni, Nj,
                     splits, overwrite, slab_bf16, 0);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
