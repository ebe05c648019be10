// Temporal window attention core for gfx950: per (window, head) problem of NTOK = T*ws*ws tokens,
//     S = q_s k^T + bias[(i mod N),(j mod N)] + mask[w][(i mod N),(j mod N)] ; P = softmax(S) ; O = P v
// (reference swin_512.py:117-138; q arrives pre-scaled from the QKV GEMM epilogue, bias/mask arrive
// expanded and TRANSPOSED as [.., key n, query n] fp32 so a wave reads them coalesced.)
//
// MFMA 32x32x16 bf16 (32x32x2 exact f32).  The scores are computed TRANSPOSED, S^T = K Q^T, so that a
// lane owns one query column and the softmax reduction over keys is in-lane (+ one exchange with lane^32).
// Work split: a wave owns 32 queries; a workgroup of 4 waves owns 128/NTOK problems (stage 1: one
// 128-token problem; stage 2: the 4 heads of one 32-token window).
//
// bf16: K and V tiles are staged in LDS by LDS-DMA (swz256 image); Q / dO operands come straight from
// global as 16-byte row fragments; P (and dS) go through LDS so the second product can read them either
// row-wise (ds_read_b128) or transposed (ds_read_b64_tr_b16).
// f32 : fragments are single elements, so K/V/Q/dO are read from global directly; only P/dS use LDS.
#include "common.h"
#include <cstdlib>

struct AttnArgs {
  const void* qkv; long ld;       // [rows][3C]: q (pre-scaled) | k | v, window-ordered rows
  void* out; long ldo;            // fwd: O [rows][C] ; bwd: dqkv [rows][3C]
  const void* dout; long lddo;    // bwd only: dO [rows][C]
  const float* biasT;             // [heads][N][N]   biasT[h][key n][query n]
  const float* maskT;             // [nW][N][N] or null
  float* dbiasT;                  // bwd: [heads][N][N] fp32, += the folded per-slot partial sums
  float* dqkv_colsum;             // bwd, optional: [C] fp32 += column sums of the dq third of dqkv (see include/stswin_hip.h)
  float* slabs;                   // bwd: caller-owned scratch [problem slots][N*N + HD]: per-slot partial sums of the two above
  int nB_, nW, heads, C, N;       // nB_ = number of (clip, window) problems = B*nW
  float scale;                    // bwd: dq = scale * (dS k)
  int bias_windows;               // 1: biasT is [heads][N][N]; > 1: biasT is [bias_windows][heads][N][N] with the mask already added
  const int* bias_index;          // optional [nW]: table slot of each window (null: slot = window, bias_windows == nW)
  const float* qscale; long ld_scale;   // fp8 q | k | v (the *_f8 kernels): qkv is e4m3 bytes (ld in BYTES), value = byte * qscale[problem][t * heads + head], t = 0 q, 1 k, 2 v
};

template <int CPR> DEVI int swz_cpr(int row) {
  constexpr int m = (CPR >= 16 ? 16 : CPR) - 1;
  return swz256(row) & m;
}
template <int ROWB> DEVI int tile_off(int row, int chunk) {   // byte offset of 16-byte chunk `chunk` of row `row`
  constexpr int CPR = ROWB / 16;
  return row * ROWB + (((chunk & ~15) | ((chunk ^ swz_cpr<CPR>(row)) & 15)) << 4);
}

// LDS-DMA a [ROWS][ROWB bytes] tile (rows `row0..` of a row-major global matrix with pitch ld_bytes) using
// `nwaves` waves (this wave is number `wv` of them).  Destination image: tile_off().
// lane: the caller's lane id when it wants the per-lane address arithmetic re-derived at the call (an `asm volatile("" : "+v"(lane))`
// inside its loop): hipcc otherwise hoists the 64-bit lane offsets of every call site out of a persistent loop and keeps them live.
// RAW: the copies are issued as raw ISA (glds16_raw), invisible to hipcc's waitcnt pass - behind a BUILTIN LDS-DMA it drains vmcnt(0) in
// front of every ds_read_b64_tr_b16 (and every LDS store), i.e. the transposed products of the backward waited for the prefetch of
// the next problem's tiles and for their own output stores; the caller then owns every wait (wait_vm0 + barrier before a tile is read).
template <int ROWS, int ROWB, bool RAW = false>
DEVI void stage_tile(char* lds, const char* g, long ld_bytes, int wv, int nwaves, int lane = -1) {
  constexpr int CPR = ROWB / 16;
  constexpr int RPI = (1024 / ROWB) > 0 ? (1024 / ROWB) : 1;       // rows per wave instruction
  constexpr int IPR = ROWB > 1024 ? ROWB / 1024 : 1;               // instructions per row (rows > 1 KiB)
  constexpr int NINST = ROWS * ROWB / 1024;
  const int l = lane >= 0 ? lane : (int)(threadIdx.x & 63);
  for (int ins = wv; ins < NINST; ins += nwaves) {
    int row, cphys;
    if constexpr (IPR == 1) { row = ins * RPI + l / CPR; cphys = l % CPR; }
    else { row = ins / IPR; cphys = (ins % IPR) * 64 + l; }
    const int csrc = (cphys & ~15) | ((cphys ^ swz_cpr<CPR>(row)) & 15);
    if constexpr (RAW) glds16_raw(g + row * ld_bytes + csrc * 16, lds + ins * 1024);
    else glds16(g + row * ld_bytes + csrc * 16, lds + ins * 1024);
  }
}

// row fragment (k contiguous) of a tile whose rows are the MFMA row/col index: 8 bf16 at k = 16*ks + 8*half
template <int ROWB> DEVI bf16x8 frag_row(const char* tile, int row, int ks, int half) {
  return *(const bf16x8*)(tile + tile_off<ROWB>(row, 2 * ks + half));
}
// transposed fragment of a tile whose ROWS are the contraction index: operand element (idx = tile col, k)
template <int ROWB> DEVI bf16x8 frag_tr(const char* tile, int ks, int col_tile) {
  const int l = threadIdx.x & 63;
  const int k0 = 16 * ks + 8 * (l >> 5), col0 = 32 * col_tile + 16 * ((l >> 4) & 1);
  const int lam = l & 15, q = lam >> 2, p = lam & 3;
  bf16x4 r[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = k0 + 4 * e + q, col = col0 + 4 * p;
    const char* addr = tile + tile_off<ROWB>(row, col >> 3) + ((col & 7) << 1);
    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
    r[e] = __builtin_bit_cast(bf16x4, t);
  }
  return cat4(r[0], r[1]);
}

// Same transposed read with the contraction rows in the order a lane HOLDS a 32x32 MFMA result: k-slot e of lane half h
// is row kbase + 4h + (e & 3) + 8 (e >> 2) (= crow32 of accumulator registers 8m..8m+7 with kbase = 32 kt + 16 m).  With
// it the probabilities go from the score accumulators straight into the next MFMA as its other operand - any k order
// is fine as long as both operands use the same one - instead of through an LDS tile.
template <int ROWB> DEVI bf16x8 frag_tr_perm(const char* tile, int kbase, int col_tile) {
  const int l = threadIdx.x & 63;
  const int k0 = kbase + 4 * (l >> 5), col0 = 32 * col_tile + 16 * ((l >> 4) & 1);
  const int lam = l & 15, q = lam >> 2, p = lam & 3;
  bf16x4 r[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = k0 + 8 * e + q, col = col0 + 4 * p;
    const char* addr = tile + tile_off<ROWB>(row, col >> 3) + ((col & 7) << 1);
    short4v t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
    r[e] = __builtin_bit_cast(bf16x4, t);
  }
  return cat4(r[0], r[1]);
}

DEVI int crow32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }   // C-map row of register r

template <typename T, int NTOK, int HD>
struct AttnCfg {
  static constexpr int QW = NTOK / 32;            // waves (query tiles) per problem
  static constexpr int PPB = 4 / QW;              // problems per workgroup
  static constexpr int KT = NTOK / 32;            // key tiles
  static constexpr int DT = HD / 32;              // head-dim tiles
  static constexpr int RB = HD * sizeof(T);       // K/V/Q/dO tile row bytes
  static constexpr int PRB = NTOK * sizeof(T);    // P/dS tile row bytes
  static constexpr int KV_BYTES = TT<T>::IS_BF16 ? NTOK * RB : 0;
  static constexpr int P_BYTES = NTOK * PRB;      // per problem, all query tiles
  static constexpr int FWD_LDS = TT<T>::IS_BF16 ? PPB * 2 * KV_BYTES : PPB * P_BYTES;   // bf16 forward keeps P in registers
  // backward tile prefetch (bf16, one problem per workgroup): a third K/V-sized buffer when it still fits 160 KB
  static constexpr bool BWD_PF = TT<T>::IS_BF16 && PPB == 1 && 3 * KV_BYTES + 2 * P_BYTES <= 160 * 1024;
  static constexpr int BWD_LDS = BWD_PF ? 3 * KV_BYTES + 2 * P_BYTES : PPB * (2 * KV_BYTES + 2 * P_BYTES);
  static_assert(NTOK % 32 == 0 && HD % 32 == 0 && (QW == 1 || QW == 2 || QW == 4), "shape");
};

// ---- buffer-addressed stores / loads: a wave-uniform base (descriptor in SGPRs), ONE per-lane 32-bit byte offset and a scalar byte
// offset per instruction.  The row-strided output stores of the attention kernels (16 rows per lane and tile) otherwise cost a 64-bit
// VGPR address pair per row, which hipcc hoists out of the persistent loops and - at the register cap - spills to scratch.
DEVI const void* uniform_ptr(const void* p) {        // the same pointer, provably wave-uniform (two v_readfirstlane)
  const unsigned long a = (unsigned long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
  return (const void*)(((unsigned long)hi << 32) | lo);
}
DEVI void buf_store_b16(void* base, bf16 v, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)0xFFFFFFFE, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, v), rs, voff, soff, 0);
}
typedef int v4i32_t __attribute__((ext_vector_type(4)));
DEVI void buf_store_b128(void* base, bf16x8 v, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)0xFFFFFFFE, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i32_t, v), rs, voff, soff, 0);
}
DEVI float buf_load_f32(const void* base, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)0xFFFFFFFE, 0x00020000);
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
DEVI void buf_store_b32(void* base, float v, int voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)0xFFFFFFFE, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, voff, soff, 0);
}
template <typename T> DEVI void buf_store_elem(void* base, float v, int voff, int soff) {
  if constexpr (TT<T>::IS_BF16) buf_store_b16(base, (bf16)v, voff, soff);
  else buf_store_b32(base, v, voff, soff);
}
// ---- S^T (+bias, +mask) and softmax for this wave's 32 queries: returns normalised P^T in p[KT] -------------
// NC: ws*ws when known at compile time (0 = read a.N): with it the compiler sees which table addresses repeat over the
// T frames of a window (kn = key mod N) and loads each bias value once instead of T times.
// Row fragments (16-byte pieces of this lane's query / dO row) straight from global memory: requested BEFORE the wait for
// the LDS-DMA tiles so that their (first-touch, HBM) latency overlaps the tile copies.
template <typename T, int HD>
DEVI void load_row_frags(bf16x8 (&f)[HD / 16], const T* base, long ld, int row) {
  const int half = (threadIdx.x & 63) >> 5;
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) f[ks] = *(const bf16x8*)(base + (long)row * ld + 16 * ks + 8 * half);
  }
}

template <typename T, int NTOK, int HD, int NC>
DEVI void scores_softmax(f32x16 (&p)[NTOK / 32], const AttnArgs& a, const char* Kt, const T* qbase, const T* kbase,
                         int q0, int head, int widx, const bf16x8 (&qf)[HD / 16], float sscale = 1.0f) {
  using Cfg = AttnCfg<T, NTOK, HD>;
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[kt][r] = 0.f;
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks)
        p[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<Cfg::RB>(Kt, kt * 32 + lr, ks, half), qf[ks], p[kt], 0, 0, 0);
  } else {
#pragma unroll 4
    for (int kk = 0; kk < HD / 2; ++kk) {
      const float qv = qbase[(long)(q0 + lr) * a.ld + 2 * kk + half];
#pragma unroll
      for (int kt = 0; kt < Cfg::KT; ++kt)
        p[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kbase[(long)(kt * 32 + lr) * a.ld + 2 * kk + half], qv, p[kt], 0, 0, 0);
    }
  }
  // + bias + mask (transposed tables: [key n][query n], lanes contiguous in query)
  const int N = NC ? NC : a.N;
  const int qn = (q0 + lr) % N;
  // (scalar load: the slot index is wave-uniform - it becomes part of a scalar offset below, and a vector load of it would be followed by
  //  s_waitcnt vmcnt(0))
  const int slot = (a.bias_windows & 0xffffff) > 1 ? (a.bias_index ? sload(a.bias_index, widx) : widx) : 0;
  const float* bt = a.biasT + ((long)slot * a.heads + head) * N * N + qn;
  const float* mt = a.maskT ? a.maskT + (long)widx * N * N + qn : nullptr;
  // All table values are requested first and consumed afterwards.  (With the optional mask tested per element the loop
  // was load - branch - wait - add, 32 dependent L2 round trips: 11.7 of the backward kernel's 24 us per problem.)
  float tb[Cfg::KT][16];
  if constexpr (NC != 0 && NC % 8 == 0) {
    // window size known: key n = (32 kt + crow32(r, 0)) % NC + 4 half (no wrap: NC % 8 == 0), so a table value sits at
    // lane part (4 half NC + qn) + scalar part (slot base + key row) - buffer loads with ONE lane offset instead of a 64-bit lane
    // pointer per table (round 5: those pointer pairs were spilled by the kernels at their register cap)
    const int tvoff = (4 * half * NC + qn) * 4;
    const int tsoff = (slot * a.heads + head) * NC * NC * 4;
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[kt][r] = buf_load_f32(a.biasT, tvoff, tsoff + ((kt * 32 + crow32(r, 0)) % NC) * NC * 4);
    if (a.maskT) {
      const int msoff = widx * NC * NC * 4;
#pragma unroll
      for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[kt][r] += buf_load_f32(a.maskT, tvoff, msoff + ((kt * 32 + crow32(r, 0)) % NC) * NC * 4);
    }
  } else {
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[kt][r] = bt[((kt * 32 + crow32(r, half)) % N) * N];
  if (mt) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[kt][r] += mt[((kt * 32 + crow32(r, half)) % N) * N];
  }
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = p[kt][r] * sscale + tb[kt][r];     // (sscale: fp8-stored q | k | v, the product of their scales; x 1.0f is exact)
      p[kt][r] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = TT<T>::IS_BF16 ? __expf(p[kt][r] - mx) : expf(p[kt][r] - mx);
      p[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) p[kt][r] *= inv;
}

// write this wave's X^T registers (lane = query column, regs = keys) into tile[query row][key] (T elements)
template <typename T, int NTOK>
DEVI void store_qk_tile(char* tile, const f32x16 (&x)[NTOK / 32], int q0) {
  constexpr int PRB = NTOK * sizeof(T);
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5;
#pragma unroll
  for (int kt = 0; kt < NTOK / 32; ++kt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int key = kt * 32 + 8 * g + 4 * half;      // 4 consecutive keys: registers 4g..4g+3
      if constexpr (TT<T>::IS_BF16) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (bf16)x[kt][4 * g + e];
        *(bf16x4*)(tile + tile_off<PRB>(q0 + lr, key >> 3) + ((key & 7) << 1)) = v;
      } else {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = x[kt][4 * g + e];
        *(f32x4*)(tile + tile_off<PRB>(q0 + lr, key >> 2)) = v;
      }
    }
}
template <typename T, int ROWB> DEVI float tile_elem_f32(const char* tile, int row, int col) {   // f32 tiles only
  return *(const float*)(tile + tile_off<ROWB>(row, col >> 2) + ((col & 3) << 2));
}

// ====================================================================================================
template <typename T, int NTOK, int HD, int NC>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
  using Cfg = AttnCfg<T, NTOK, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5, w = wave_id();
  const int sp = w / Cfg::QW, qt = w % Cfg::QW;
  const long prob = (long)blockIdx.x * Cfg::PPB + sp;
  const int b_ = prob / a.heads, head = prob % a.heads;
  const bool live = b_ < a.nB_;
  const long rowbase = (long)b_ * NTOK;
  const T* qbase = (const T*)a.qkv + rowbase * a.ld + head * HD;
  const T* kbase = qbase + a.C;
  const T* vbase = qbase + 2 * a.C;
  char* Kt = smem + sp * (TT<T>::IS_BF16 ? 2 * Cfg::KV_BYTES : Cfg::P_BYTES);
  char* Vt = Kt + Cfg::KV_BYTES;
  char* Pt = Kt;                                   // f32 only (no K / V tiles there)
  const int q0 = qt * 32;
  bf16x8 qf[HD / 16];
  if constexpr (TT<T>::IS_BF16) {
    if (live) {
      stage_tile<NTOK, Cfg::RB>(Kt, (const char*)kbase, a.ld * sizeof(T), qt, Cfg::QW);
      stage_tile<NTOK, Cfg::RB>(Vt, (const char*)vbase, a.ld * sizeof(T), qt, Cfg::QW);
      load_row_frags<T, HD>(qf, qbase, a.ld, q0 + lr);
    }
    wait_vm0();
    __syncthreads();
  }
  if (!live) return;   // (no barrier below is reached by a subset of a PROBLEM's waves only when QW == 1)

  f32x16 p[Cfg::KT];
  scores_softmax<T, NTOK, HD, NC>(p, a, Kt, qbase, kbase, q0, head, b_ % a.nW, qf);

  f32x16 o[Cfg::DT];
#pragma unroll
  for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  if constexpr (TT<T>::IS_BF16) {
    // O = P V with P taken from the score accumulators: this lane's query row, k-slots in accumulator order
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 pa;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[e] = (bf16)p[kt][8 * m + e];
#pragma unroll
        for (int dt = 0; dt < Cfg::DT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr_perm<Cfg::RB>(Vt, kt * 32 + 16 * m, dt), o[dt], 0, 0, 0);
      }
  } else {
    store_qk_tile<T, NTOK>(Pt, p, q0);   // rows q0..q0+31 are private to this wave
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): own LDS writes visible to own reads
#pragma unroll 4
    for (int kk = 0; kk < NTOK / 2; ++kk) {
      const float pa = tile_elem_f32<T, Cfg::PRB>(Pt, q0 + lr, 2 * kk + half);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, vbase[(long)(2 * kk + half) * a.ld + dt * 32 + lr], o[dt], 0, 0, 0);
    }
  }
  // O[query = crow32(r)][d = 32 dt + lr] -> out[rowbase + q0 + query][head*HD + d].  (The transposed product V^T P^T
  // would give every lane 8-byte pieces, but a store instruction then touches 32 rows x 16 B instead of 2 rows x 64 B and
  // measured 9 % SLOWER on the stage-1 shape: bytes per touched cache line matter more than the instruction count.)
  T* ob = (T*)a.out + (rowbase + q0) * a.ldo + head * HD;
#pragma unroll
  for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ob[(long)crow32(r, half) * a.ldo + dt * 32 + lr] = from_f32<T>(o[dt][r]);
}

// ---- fp8 (e4m3) tile -> bf16 tile, in registers: one 16-byte chunk (16 values) -> two bf16x8.  e4m3 values are exact in bf16, so the
// expanded tile holds the stored bytes' values; the per-(window, head) scales are applied to accumulators, not to the tile.
DEVI void f8_chunk_to_bf16(const int (&w4)[4], bf16x8& lo, bf16x8& hi) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(w4[d], false), b = __builtin_amdgcn_cvt_pk_f32_fp8(w4[d], true);
    bf16x8& t = d < 2 ? lo : hi;
    const int o = (d & 1) * 4;
    t[o] = (bf16)a[0]; t[o + 1] = (bf16)a[1]; t[o + 2] = (bf16)b[0]; t[o + 3] = (bf16)b[1];
  }
}
// A [ROWS][HD-byte] fp8 tile image (tile_off<HD>, at `src`) -> the [ROWS][2 HD-byte] bf16 image (tile_off<2 HD>, at `dst`), by NTHR
// threads (this one is number `t`).  load() then - after a barrier of those threads when dst overlaps src - store().
template <int ROWS, int HD, int NTHR>
struct F8Expand {
  static constexpr int CPR = HD / 16, NCH = ROWS * CPR / NTHR;     // 16-byte fp8 chunks per row / per thread
  static_assert(ROWS * CPR % NTHR == 0 && NCH >= 1, "F8Expand geometry");
  int v[NCH][4];
  DEVI void load(const char* src, int t) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = t * NCH + c, row = idx / CPR, ch = idx % CPR;
      const int4 q = *(const int4*)(src + tile_off<HD>(row, ch));
      v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
    }
  }
  DEVI void store(char* dst, int t) const {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int idx = t * NCH + c, row = idx / CPR, ch = idx % CPR;
      bf16x8 lo, hi;
      f8_chunk_to_bf16(v[c], lo, hi);
      *(bf16x8*)(dst + tile_off<2 * HD>(row, 2 * ch)) = lo;
      *(bf16x8*)(dst + tile_off<2 * HD>(row, 2 * ch + 1)) = hi;
    }
  }
};
DEVI float sloadf(const float* p, long idx) { return ((const __attribute__((address_space(4))) float*)p)[idx]; }

// ====================================================================================================
// QKV-fused forward for the stage-1 shape (bf16, 128 tokens per window pair, head dim 128): the window gather, the QKV
// projection, the attention core and (when a backward will follow) the q | k | v rows for it, in ONE kernel
// (swin_512.py:115-141 without the (B_, N, 3C) tensor between nn.Linear and the attention).  north_star: "window partition +
// cyclic shift fused with Q/K/V packing ... windowed attention as an MFMA bf16 batched-GEMM pair".
//
// One (window, head) problem at a time per 8-wave workgroup (persistent over problems; the four heads of a window run on
// neighbouring workgroups, so its token rows come out of L2):
//   A  projection  [128 tokens x C] . W_h^T -> [128 x 384]  (W_h = the head's 128 q, k and v rows of qkv.weight):
//      token rows gathered by the window row map and the weight rows (L2 resident: 1.5 MB) in 32-column chunks through a
//      4-stage LDS ring (LDS-DMA as raw ISA, requested 3 chunks ahead, counted vmcnt waits, one raw s_barrier per chunk);
//      every wave owns 48 of the 384 output columns for all 128 rows; MFMA 16x16x32 with the operands swapped (a lane then
//      holds 4 consecutive output columns of one token row);
//   A' epilogue: + bias, q * scale, bf16, 8-byte stores into the Q | K | V tiles in LDS (the images the attention core reads);
//   B  waves 0-3: the forward core of attn_fwd_kernel on the three LDS tiles (S^T = K Q^T + bias + mask, in-register softmax,
//      O = P V); waves 4-7 meanwhile copy the tiles to qkv_out (only when the backward needs them: a no-grad pass - the six
//      momentum-key encoder passes of the contrastive step, evaluation - never writes q, k, v to HBM at all).
struct AttnQkvArgs {
  const bf16* X; long ldx; const int* rmap;      // tokens [.][C]; rmap[window row] = token row or -1 (null: identity)
  const bf16* W; long ldw; const float* bqkv;    // qkv.weight [3C][C], qkv.bias [3C] or null
  bf16* qkv_out; long ldq;                       // optional [nB_*128][3C] (q pre-scaled | k | v), window order
  AttnArgs a;                                    // out / ldo, biasT, maskT, nB_, nW, heads, C, N, scale, bias_windows, bias_index
};

template <int NC, int CDIM>
__global__ __launch_bounds__(512) void attn_qkv_fwd_kernel(AttnQkvArgs q) {
  using T = bf16;
  constexpr int NTOK = 128, HD = 128, ROWB = 256, KV = NTOK * ROWB, NSTG = 4;
  constexpr int XB = 128 * 64, WB = 384 * 64, STAGE = XB + WB;               // ring stage: tokens [128][32 bf16] | weights [384][32 bf16]
  using Cfg = AttnCfg<T, NTOK, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const AttnArgs& a = q.a;
  char* Qt = smem;                                   // the three tiles REUSE the ring's memory: they are written after the last chunk
  char* Kt = smem + KV;                              // has been multiplied and are dead before the next problem's first copy
  char* Vt = smem + 2 * KV;
  char* ring = smem;
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  constexpr int C = CDIM, nks = C / 32;
  const long nprob = (long)a.nB_ * a.heads;
  const int arow = tid >> 2, apiece = (tid & 3) ^ ((arow >> 2) & 3);         // token copy: physical piece tid & 3 of chunk row arow
  const int fr = l & 15, fq = l >> 4;
  // fragment read offsets inside a stage (64-byte rows, piece ^= (row >> 2) & 3): tokens 16 i + fr, weight rows 48 w + 16 j + fr
  int xoff[8], woff[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int row = 16 * i + fr; xoff[i] = row * 64 + ((fq ^ ((row >> 2) & 3)) << 4); }
#pragma unroll
  for (int j = 0; j < 3; ++j) { const int row = 48 * w + 16 * j + fr; woff[j] = XB + row * 64 + ((fq ^ ((row >> 2) & 3)) << 4); }
  // DBG (tools/attn_qkv_timeline.py): a.dqkv_colsum = u64 [workgroups][8 waves][8] stamps (100 MHz) of every wave's SECOND problem
  unsigned long long* ts = (unsigned long long*)a.dqkv_colsum;
  int pcount = 0;
  auto stamp = [&](int slot) {
    if (ts && pcount == 1 && l == 0) ts[((long)blockIdx.x * 8 + w) * 8 + slot] = wall_clock64();
  };
  auto map_row = [&](long prob) -> int {             // token row of this thread's chunk row in problem `prob`
    const long rb = (prob / a.heads) * NTOK;
    return q.rmap ? q.rmap[rb + arow] : (int)(rb + arow);
  };
  int srow_next = blockIdx.x < nprob ? map_row(blockIdx.x) : -1;
  const buf_rsrc_t rsX = make_buf_rsrc(q.X), rsW = make_buf_rsrc(q.W);
  const int xrow_lim = (int)(0xFFFF0000u / (unsigned)(q.ldx * 2)) - 1;       // token rows a 32-bit byte offset reaches
  bool xrow_bad = false;                                                     // a gathered row beyond that: trap at the end (fail loudly)
  for (long prob = blockIdx.x; prob < nprob; prob += gridDim.x, ++pcount) {
    const int b_ = (int)(prob / a.heads), head = (int)(prob - (long)b_ * a.heads);
    const long rowbase = (long)b_ * NTOK;
    stamp(0);
    const int srow = srow_next;
    // (the next problem's map entry is requested now: at its start the first copies no longer wait for a dependent load)
    if (prob + gridDim.x < nprob) srow_next = map_row(prob + gridDim.x);
    // Copies are buffer-addressed (round 5): ONE 32-bit byte offset per lane and copy (token piece + three weight pieces) against two
    // descriptors in SGPRs, the chunk position as the instruction's scalar offset - four 64-bit pointer pairs less per lane in a kernel
    // that sits at its 256-register cap (they were part of what it spilled).  Padding rows (map entry -1) get an out-of-range offset:
    // the copy delivers zeros.
    const unsigned aoff = (srow >= 0 && srow <= xrow_lim) ? (unsigned)srow * (unsigned)(q.ldx * 2) + apiece * 16 : 0xFFFFFFFFu;
    xrow_bad |= srow > xrow_lim;
    unsigned woffs[3];                                                     // weight copy: pieces tid, tid + 512, tid + 1024 of the chunk
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pc = tid + 512 * j, n = pc >> 2, part = n >> 7, within = n & 127;
      woffs[j] = (unsigned)(part * C + head * HD + within) * (unsigned)(q.ldw * 2) + (((pc & 3) ^ ((n >> 2) & 3)) << 4);
    }
    f32x4 acc[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Every global -> LDS copy is raw ISA (glds16_raw: invisible to hipcc's waitcnt pass, which otherwise drains all outstanding
    // copies in front of LDS reads and barriers) and the waits are counted by hand: a chunk = 4 copy instructions per thread
    // (1 token piece + 3 weight pieces), chunks are requested 3 ahead, so chunk ks has landed when <= 8 copies are outstanding.
    auto request = [&](int ks) {
      char* st = ring + (ks % NSTG) * STAGE;
      glds16_buf_raw(rsX, aoff, (unsigned)(ks * 64), st + w * 1024);
#pragma unroll
      for (int j = 0; j < 3; ++j) glds16_buf_raw(rsW, woffs[j], (unsigned)(ks * 64), st + XB + j * 8192 + w * 1024);
    };
    // Software pipeline: while the MFMAs of chunk ks run, the fragments of chunk ks + 1 are read from LDS into the other register
    // set (22 KB of fragment reads per wave and chunk take as long as its 24 MFMAs: back to back they halved the rate), and chunk
    // ks + 4 is requested into the stage chunk ks has just left.
    request(0); request(1); request(2); request(3);
    bf16x8 wf[2][3], xa[2][8];
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(1);
#pragma unroll
    for (int j = 0; j < 3; ++j) wf[0][j] = *(const bf16x8*)(ring + woff[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[0][i] = *(const bf16x8*)(ring + xoff[i]);
#pragma unroll
    for (int ks = 0; ks < nks; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < nks) {
        // chunk ks + 1 has landed when only the chunks requested after it (up to ks + 3) are outstanding
        const int later = (ks + 3 < nks ? ks + 3 : nks - 1) - (ks + 1);
        // (lgkmcnt(0): this wave's own fragment reads of chunk ks are complete - the barrier then frees that stage for a copy)
        if (later >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // chunk ks + 1 visible; every wave has finished READING chunk ks (last step)
        if (ks + 4 < nks) request(ks + 4);           // into the stage of chunk ks
        const char* st = ring + ((ks + 1) % NSTG) * STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j) wf[nxt][j] = *(const bf16x8*)(st + woff[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[nxt][i] = *(const bf16x8*)(st + xoff[i]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][j], xa[cur][i], acc[i][j], 0, 0, 0);
    }
    stamp(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // every wave is done with the ring: its memory becomes the Q | K | V tiles
    stamp(3);
    // A': acc[i][j][r] = (token 16 i + fr, column 16 (3 w + j) + 4 fq + r of the head's q | k | v) -> LDS tiles
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = w * 3 + j, part = t >> 3, c0 = (t & 7) * 16 + 4 * fq;
      char* tile = part == 0 ? Qt : (part == 1 ? Kt : Vt);
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (q.bqkv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = q.bqkv[(long)part * C + head * HD + c0 + r];
      }
      const float mul = part == 0 ? a.scale : 1.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 16 * i + fr;
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (bf16)((acc[i][j][r] + bv[r]) * mul);
        *(bf16x4*)(tile + tile_off<ROWB>(row, c0 >> 3) + ((c0 & 7) << 1)) = v;
      }
    }
    __syncthreads();
    stamp(4);
    if (w < 4) {
      // B: attention core (attn_fwd_kernel) on the LDS tiles
      const int lr = l & 31, half = l >> 5, q0 = w * 32;
      bf16x8 qf[HD / 16];
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = frag_row<ROWB>(Qt, q0 + lr, ks, half);
      f32x16 p[Cfg::KT];
      scores_softmax<T, NTOK, HD, NC>(p, a, Kt, (const T*)nullptr, (const T*)nullptr, q0, head, b_ % a.nW, qf);
      f32x16 o[Cfg::DT];
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          bf16x8 pa;
#pragma unroll
          for (int e = 0; e < 8; ++e) pa[e] = (bf16)p[kt][8 * m + e];
#pragma unroll
          for (int dt = 0; dt < Cfg::DT; ++dt)
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr_perm<ROWB>(Vt, kt * 32 + 16 * m, dt), o[dt], 0, 0, 0);
        }
      // (buffer stores: wave-uniform base = this wave's first output row, per lane (4 half ldo + lr) elements, scalar row offsets)
      void* ob = (void*)uniform_ptr((T*)a.out + (rowbase + q0) * a.ldo + head * HD);
      const int ldo_b = (int)a.ldo * 2, vo = (4 * half) * ldo_b + lr * 2;
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) buf_store_b16(ob, (bf16)o[dt][r], vo + dt * 64, crow32(r, 0) * ldo_b);
    } else if (q.qkv_out) {
      // the rows the backward will read: 3 tiles x 128 rows x 16 pieces of 16 bytes, 24 per thread of waves 4-7
      const int t4 = tid - 256;
#pragma unroll 4
      for (int pc = t4; pc < 3 * NTOK * 16; pc += 256) {
        const int part = pc / (NTOK * 16), rem = pc - part * (NTOK * 16), row = rem >> 4, ch = rem & 15;
        const char* tile = smem + part * KV;
        const bf16x8 v = *(const bf16x8*)(tile + tile_off<ROWB>(row, ch));
        *(bf16x8*)(q.qkv_out + (rowbase + row) * q.ldq + (long)part * C + head * HD + ch * 8) = v;
      }
    }
    stamp(5);
    __syncthreads();                                   // tiles and ring are reused by the next problem
    stamp(6);
  }
#ifdef STSWIN_TUNING
  if (xrow_bad) __builtin_trap();                      // debug assertion only: the launcher refuses an x beyond the 32-bit offset range
#else
  (void)xrow_bad;
#endif
}

// ====================================================================================================
// fp8 (OCP e4m3) forward for BASELINE.json configs[4] ("fp8 MFMA attention"): the same kernel with q, k, v and the
// probabilities quantised in registers and both products on v_mfma_f32_32x32x16_fp8_fp8.  q / k / v stay bf16 in HBM (the
// QKV GEMM writes them, the backward reads them), so this mode changes the arithmetic, not the traffic: per (window, head)
// problem amax scales (amax / 448) for q, k and v, P scaled by 128 (softmax outputs below 2^-6 would otherwise fall into the
// e4m3 subnormals), fp32 accumulation, fp32 softmax.  The non-scaled fp8 MFMA has the bf16 rate on gfx950
// (MI355X_MICROARCH.md, Matrix cores), so this is a numerics mode, not a speed-up; the backward is the bf16 kernel on the
// unquantised q / k / v (straight-through).
DEVI long pack_fp8x8(const bf16x8 v, float mul) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[0] * mul, (float)v[1] * mul, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[2] * mul, (float)v[3] * mul, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[4] * mul, (float)v[5] * mul, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[6] * mul, (float)v[7] * mul, hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}
DEVI float amax8(const bf16x8 v, float m) {
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
  return m;
}

template <int NTOK, int HD, int NC>
__global__ __launch_bounds__(256) void attn_fwd_fp8_kernel(AttnArgs a) {
  using T = bf16;
  using Cfg = AttnCfg<T, NTOK, HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5, w = wave_id();
  const int sp = w / Cfg::QW, qt = w % Cfg::QW;
  const long prob = (long)blockIdx.x * Cfg::PPB + sp;
  const int b_ = min((int)(prob / a.heads), a.nB_ - 1), head = prob % a.heads;
  const bool live = prob / a.heads < a.nB_;
  const long rowbase = (long)b_ * NTOK;
  const T* qbase = (const T*)a.qkv + rowbase * a.ld + head * HD;
  char* Kt = smem + sp * 2 * Cfg::KV_BYTES;
  char* Vt = Kt + Cfg::KV_BYTES;
  float* ex = (float*)(smem + Cfg::PPB * 2 * Cfg::KV_BYTES);       // [4 waves][3]: |q|, |k|, |v| maxima of each wave's share
  const int q0 = qt * 32;
  bf16x8 qf[HD / 16];
  stage_tile<NTOK, Cfg::RB>(Kt, (const char*)(qbase + a.C), a.ld * sizeof(T), qt, Cfg::QW);
  stage_tile<NTOK, Cfg::RB>(Vt, (const char*)(qbase + 2 * a.C), a.ld * sizeof(T), qt, Cfg::QW);
  load_row_frags<T, HD>(qf, qbase, a.ld, q0 + lr);
  wait_vm0();
  __syncthreads();
  // ---- per-problem amax: this wave's 32 query rows, and its 32 rows of the K and V tiles
  float mq = 0.f, mk = 0.f, mv = 0.f;
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    mq = amax8(qf[ks], mq);
    mk = amax8(frag_row<Cfg::RB>(Kt, q0 + lr, ks, half), mk);
    mv = amax8(frag_row<Cfg::RB>(Vt, q0 + lr, ks, half), mv);
  }
  mq = wave_max(mq); mk = wave_max(mk); mv = wave_max(mv);
  if (l == 0) { ex[w * 3 + 0] = mq; ex[w * 3 + 1] = mk; ex[w * 3 + 2] = mv; }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < Cfg::QW; ++o) {
    mq = fmaxf(mq, ex[(sp * Cfg::QW + o) * 3 + 0]);
    mk = fmaxf(mk, ex[(sp * Cfg::QW + o) * 3 + 1]);
    mv = fmaxf(mv, ex[(sp * Cfg::QW + o) * 3 + 2]);
  }
  const float sq = mq > 0.f ? mq * (1.0f / 448.0f) : 1.0f, sk = mk > 0.f ? mk * (1.0f / 448.0f) : 1.0f,
              sv = mv > 0.f ? mv * (1.0f / 448.0f) : 1.0f;
  long q8[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) q8[ks] = pack_fp8x8(qf[ks], 1.0f / sq);
  // ---- S^T = K Q^T on the fp8 MFMA, rescaled; + bias (+ mask); softmax in fp32
  f32x16 p[Cfg::KT];
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) p[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
      p[kt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pack_fp8x8(frag_row<Cfg::RB>(Kt, kt * 32 + lr, ks, half), 1.0f / sk), q8[ks],
                                                        p[kt], 0, 0, 0);
  }
  const int N = NC ? NC : a.N;
  const int qn = (q0 + lr) % N, widx = b_ % a.nW;
  const int slot = (a.bias_windows & 0xffffff) > 1 ? (a.bias_index ? a.bias_index[widx] : widx) : 0;
  const float* bt = a.biasT + ((long)slot * a.heads + head) * N * N + qn;
  const float* mt = a.maskT ? a.maskT + (long)widx * N * N + qn : nullptr;
  const float ssc = sq * sk;
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kn = ((kt * 32 + crow32(r, half)) % N) * N;
      float sc = p[kt][r] * ssc + bt[kn];
      if (mt) sc += mt[kn];
      p[kt][r] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(p[kt][r] - mx);
      p[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float pscale = 128.0f / sum;               // P * 128 in e4m3
  // ---- O = P V: P from the score accumulators (k-slots in accumulator order), V read transposed in the same order
  f32x16 o[Cfg::DT];
#pragma unroll
  for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 0] * pscale, p[kt][8 * m + 1] * pscale, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 2] * pscale, p[kt][8 * m + 3] * pscale, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 4] * pscale, p[kt][8 * m + 5] * pscale, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 6] * pscale, p[kt][8 * m + 7] * pscale, hi, true);
      const long pa = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pa, pack_fp8x8(frag_tr_perm<Cfg::RB>(Vt, kt * 32 + 16 * m, dt), 1.0f / sv), o[dt],
                                                          0, 0, 0);
    }
  if (!live) return;
  const float osc = sv * (1.0f / 128.0f);
  T* ob = (T*)a.out + (rowbase + q0) * a.ldo + head * HD;
#pragma unroll
  for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ob[(long)crow32(r, half) * a.ldo + dt * 32 + lr] = (bf16)(o[dt][r] * osc);
}

// ====================================================================================================
// fp8-STORAGE forward (round 4; BASELINE configs[4] as a training path): q | k | v arrive as e4m3 bytes with one fp32 scale per
// (window problem, head, q / k / v) - written by stswin_gemm_nt_qkv_fp8 - so the kernel reads HALF the bytes of the bf16 forward
// and feeds the fp8 MFMA without a conversion: K rows are 8-byte LDS reads, the Q row pieces 8-byte global loads.  V has to reach
// the MFMA with the contraction (key) index contiguous, i.e. transposed; ds_read_b64_tr_b16 on the fp8 tile viewed as 16-bit
// elements transposes BYTE PAIRS (columns 2c, 2c + 1), and two v_perm_b32 per dword pair split them into the operand of the even and
// of the odd column - two MFMAs per transposed read instead of one, and the lane ends up with two NEIGHBOURING output columns
// (one 4-byte store).  P is quantised as in attn_fwd_fp8_kernel (x 128, e4m3).  scores = sq sk (q^ . k^), out = sv / 128 (P^ v^).
template <int NTOK, int HD, int NC>
__global__ __launch_bounds__(256) void attn_fwd_f8_kernel(AttnArgs a) {
  using Cfg = AttnCfg<bf16, NTOK, HD>;
  constexpr int KV8 = NTOK * HD, CT = HD / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5, w = wave_id();
  const int sp = w / Cfg::QW, qt = w % Cfg::QW;
  const long prob = (long)blockIdx.x * Cfg::PPB + sp;
  const int b_ = min((int)(prob / a.heads), a.nB_ - 1), head = prob % a.heads;
  const bool live = prob / a.heads < a.nB_;
  const long rowbase = (long)b_ * NTOK;
  const char* qb = (const char*)a.qkv + rowbase * a.ld + head * HD;      // e4m3 bytes; k at + C, v at + 2C
  char* Kt = smem + sp * 2 * KV8;
  char* Vt = Kt + KV8;
  const int q0 = qt * 32;
  stage_tile<NTOK, HD>(Kt, qb + a.C, a.ld, qt, Cfg::QW);
  stage_tile<NTOK, HD>(Vt, qb + 2 * a.C, a.ld, qt, Cfg::QW);
  long q8[HD / 16];
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) q8[ks] = *(const long*)(qb + (long)(q0 + lr) * a.ld + 16 * ks + 8 * half);
  const float* scp = a.qscale + (long)b_ * a.ld_scale;
  const float sq = scp[head], sk = scp[a.heads + head], sv = scp[2 * a.heads + head];
  wait_vm0();
  __syncthreads();
  f32x16 p[Cfg::KT];
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) p[kt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks)
      p[kt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(*(const long*)(Kt + tile_off<HD>(kt * 32 + lr, ks) + 8 * half), q8[ks], p[kt], 0, 0, 0);
  }
  const int N = NC ? NC : a.N;
  const int qn = (q0 + lr) % N, widx = b_ % a.nW;
  const int slot = (a.bias_windows & 0xffffff) > 1 ? (a.bias_index ? a.bias_index[widx] : widx) : 0;
  const float* bt = a.biasT + ((long)slot * a.heads + head) * N * N + qn;
  const float* mt = a.maskT ? a.maskT + (long)widx * N * N + qn : nullptr;
  float tb[Cfg::KT][16];
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) tb[kt][r] = bt[((kt * 32 + crow32(r, half)) % N) * N];
  if (mt) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[kt][r] += mt[((kt * 32 + crow32(r, half)) % N) * N];
  }
  const float ssc = sq * sk;
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sc = p[kt][r] * ssc + tb[kt][r];
      p[kt][r] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(p[kt][r] - mx);
      p[kt][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float pscale = 128.0f / sum;               // P * 128 in e4m3 (softmax outputs below 2^-6 would fall into the subnormals)
  f32x16 oe[CT], oo[CT];                           // even / odd output columns 64 ct + 2 lr (+ 1)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oe[ct][r] = 0.f; oo[ct][r] = 0.f; }
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      int lo = 0, hi = 0;
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 0] * pscale, p[kt][8 * m + 1] * pscale, lo, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 2] * pscale, p[kt][8 * m + 3] * pscale, lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 4] * pscale, p[kt][8 * m + 5] * pscale, hi, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(p[kt][8 * m + 6] * pscale, p[kt][8 * m + 7] * pscale, hi, true);
      const long pa = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        // eight (key) k-slots in accumulator order x one byte PAIR per slot: dword m2 holds slots 2 m2, 2 m2 + 1
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const u4v wv = __builtin_bit_cast(u4v, frag_tr_perm<HD>(Vt, kt * 32 + 16 * m, ct));
        const unsigned e0 = __builtin_amdgcn_perm(wv[1], wv[0], 0x06040200u), e1 = __builtin_amdgcn_perm(wv[3], wv[2], 0x06040200u);
        const unsigned d0 = __builtin_amdgcn_perm(wv[1], wv[0], 0x07050301u), d1 = __builtin_amdgcn_perm(wv[3], wv[2], 0x07050301u);
        oe[ct] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pa, (long)(((unsigned long)e1 << 32) | e0), oe[ct], 0, 0, 0);
        oo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pa, (long)(((unsigned long)d1 << 32) | d0), oo[ct], 0, 0, 0);
      }
    }
  if (!live) return;
  const float osc = sv * (1.0f / 128.0f);
  bf16* ob = (bf16*)a.out + (rowbase + q0) * a.ldo + head * HD;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bf16x2 o2 = {(bf16)(oe[ct][r] * osc), (bf16)(oo[ct][r] * osc)};
      *(bf16x2*)(ob + (long)crow32(r, half) * a.ldo + ct * 64 + 2 * lr) = o2;
    }
}

// ====================================================================================================
// Backward: dq = scale * dS k ; dk = dS^T q_s ; dv = P^T dO ; dbias += fold(dS) ;  dS = P o (dP - rowsum(P o dP)).
// Persistent over problems: a workgroup walks problem groups blockIdx.x, blockIdx.x + gridDim.x, ... and the launcher
// makes gridDim.x * PPB a multiple of `heads`, so a lane meets the same (head, query n, key n) and the same dqkv columns
// in every iteration.  The relative-position-bias gradient and the qkv-bias column sums are therefore summed in
// registers across iterations and leave as ONE atomic per lane-entry per workgroup: issued per problem they were 33 M
// (stage 1) / 2 M (stage 2) fp32 atomics onto 16 K / 1 K addresses and cost 25 % / 65 % of the kernel.
// F8 (bf16 instantiations with one problem per wave, i.e. the stage-2 shape 32 x 256; BASELINE configs[4]): q | k | v arrive as e4m3
// bytes + per-(window, head) scales.  Every fp8 tile is copied into the UPPER half of the bf16 tile's buffer and expanded in place by the
// wave that owns the problem (F8Expand: all loads of the wave precede its stores); the Q row pieces are 8-byte global loads converted in
// registers; the scales ride on the accumulators as in attn_bwd8_kernel<.., F8>.
template <typename T, int NTOK, int HD, int NC, bool F8 = false>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnArgs a) {
  using Cfg = AttnCfg<T, NTOK, HD>;
  static_assert(!F8 || (TT<T>::IS_BF16 && Cfg::QW == 1 && !Cfg::BWD_PF), "fp8 variant: bf16 kernels with one problem per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5, w = wave_id();
  const int sp = w / Cfg::QW, qt = w % Cfg::QW;
  const long ngroups = ((long)a.nB_ * a.heads + Cfg::PPB - 1) / Cfg::PPB;
  const int head = (int)(((long)blockIdx.x * Cfg::PPB + sp) % a.heads);
  f32x16 dbacc[Cfg::KT];
  float csacc[Cfg::DT];
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dbacc[kt][r] = 0.f;
#pragma unroll
  for (int dt = 0; dt < Cfg::DT; ++dt) csacc[dt] = 0.f;
  // Tile prefetch (Cfg::BWD_PF): three K/V-sized LDS buffers rotate through the roles  K | V -> Q | dO -> next K  so that
  // every LDS-DMA copy is requested one phase before it is needed: dO while the scores are computed, Q while dV is
  // multiplied, the next problem's K during dK / dQ and its V at the very end (it is first used after the next S^T).
  // Without it each of the three re-stagings of a problem exposed a full copy latency with one wave per SIMD.
  constexpr bool PF = Cfg::BWD_PF;
  const bool dbg_ts = (a.bias_windows & (1 << 30)) != 0;
  auto stamp = [&](int slot) {
    if (dbg_ts && threadIdx.x == 0) ((unsigned long long*)a.dqkv_colsum)[(long)blockIdx.x * 16 + slot] = wall_clock64();
  };
  int kbuf = 0, xbuf = 2;                        // PF: buffer holding K / the spare one (dO, then the next K); V and Q use buffer 1
  auto problem_rowbase = [&](long g) -> long {
    const long pr = g * Cfg::PPB + sp;
    return (long)min((int)(pr / a.heads), a.nB_ - 1) * NTOK;
  };
  if constexpr (PF) {
    const long rb0 = problem_rowbase(blockIdx.x);
    const T* q0b = (const T*)a.qkv + rb0 * a.ld + head * HD;
    stage_tile<NTOK, Cfg::RB>(smem, (const char*)(q0b + a.C), a.ld * sizeof(T), qt, Cfg::QW);
    stage_tile<NTOK, Cfg::RB>(smem + Cfg::KV_BYTES, (const char*)(q0b + 2 * a.C), a.ld * sizeof(T), qt, Cfg::QW);
  }
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
  const long prob = grp * Cfg::PPB + sp;
  const int b_ = min((int)(prob / a.heads), a.nB_ - 1);
  const bool live = prob / a.heads < a.nB_;      // dead problems recompute a live one and skip every store
  const long rowbase = (long)b_ * NTOK;
  const T* qbase = (const T*)a.qkv + rowbase * a.ld + head * HD;
  const T* kbase = qbase + a.C;
  const T* vbase = qbase + 2 * a.C;
  const T* dobase = (const T*)a.dout + rowbase * a.lddo + head * HD;
  char* Kt = PF ? smem + kbuf * Cfg::KV_BYTES : smem + sp * (2 * Cfg::KV_BYTES + 2 * Cfg::P_BYTES);
  char* Vt = PF ? smem + Cfg::KV_BYTES : Kt + Cfg::KV_BYTES;            // V, later Q (not PF: V, dO, Q)
  char* Xt = PF ? smem + xbuf * Cfg::KV_BYTES : Vt;                       // dO
  char* Pt = PF ? smem + 3 * Cfg::KV_BYTES : Vt + Cfg::KV_BYTES;
  char* St = Pt + Cfg::P_BYTES;                  // dS
  const int q0 = qt * 32;
  stamp(0);
  bf16x8 qf[HD / 16], df[HD / 16];
  const char* q8b = (const char*)a.qkv + rowbase * a.ld + head * HD;     // F8: e4m3 bytes (a.ld in bytes); k at + C, v at + 2C
  float sq = 1.f, sk = 1.f, sv = 1.f;
  if constexpr (F8) {
    const float* scp = a.qscale + (long)b_ * a.ld_scale;
    sq = scp[head]; sk = scp[a.heads + head]; sv = scp[2 * a.heads + head];
    long q8[HD / 16];
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) q8[ks] = *(const long*)(q8b + (long)(q0 + lr) * a.ld + 16 * ks + 8 * half);
    stage_tile<NTOK, HD>(Kt + Cfg::KV_BYTES / 2, q8b + a.C, a.ld, qt, Cfg::QW);
    stage_tile<NTOK, HD>(Vt + Cfg::KV_BYTES / 2, q8b + 2 * a.C, a.ld, qt, Cfg::QW);
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) {           // 8 e4m3 bytes -> this lane's bf16x8 row piece
      const int lo = (int)(q8[ks] & 0xffffffffL), hi = (int)(q8[ks] >> 32);
      const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(lo, true);
      const f32x2 b0 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, false), b1 = __builtin_amdgcn_cvt_pk_f32_fp8(hi, true);
      qf[ks] = (bf16x8){(bf16)a0[0], (bf16)a0[1], (bf16)a1[0], (bf16)a1[1], (bf16)b0[0], (bf16)b0[1], (bf16)b1[0], (bf16)b1[1]};
    }
    wait_vm0();
    __syncthreads();
    {                                               // the wave's own K and V: fp8 upper halves -> bf16 images (loads, then stores)
      F8Expand<NTOK, HD, 64> ex;                    // (one tile at a time: 32 registers of bytes, not 64)
      ex.load(Kt + Cfg::KV_BYTES / 2, l);
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): every lane's bytes are in registers
      __builtin_amdgcn_wave_barrier();
      ex.store(Kt, l);
      ex.load(Vt + Cfg::KV_BYTES / 2, l);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      ex.store(Vt, l);
    }
    load_row_frags<T, HD>(df, dobase, a.lddo, q0 + lr);   // (behind the expansion: 64 registers that would sit beside its 32 + the 64 of qf;
    __syncthreads();                                       //  the scores + softmax below cover the loads)
  } else {
  load_row_frags<T, HD>(qf, qbase, a.ld, q0 + lr);        // Q (and dO) row pieces: in flight during the tile wait
  if constexpr (!PF) load_row_frags<T, HD>(df, dobase, a.lddo, q0 + lr);   // (PF: dO pieces come from the LDS tile below)
  if constexpr (TT<T>::IS_BF16) {
    if constexpr (!PF) {
      stage_tile<NTOK, Cfg::RB>(Kt, (const char*)kbase, a.ld * sizeof(T), qt, Cfg::QW);
      stage_tile<NTOK, Cfg::RB>(Vt, (const char*)vbase, a.ld * sizeof(T), qt, Cfg::QW);
    }
    wait_vm0();
    __syncthreads();
    if constexpr (PF) stage_tile<NTOK, Cfg::RB>(Xt, (const char*)dobase, a.lddo * sizeof(T), qt, Cfg::QW);
  }
  }
  stamp(1);
  f32x16 p[Cfg::KT], dp[Cfg::KT];
  scores_softmax<T, NTOK, HD, NC>(p, a, Kt, qbase, kbase, q0, head, b_ % a.nW, qf, F8 ? sq * sk : 1.0f);
  stamp(2);
  if constexpr (PF) {                            // the dO tile requested before the scores has landed: row pieces from LDS
    wait_vm0();
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < HD / 16; ++ks) df[ks] = frag_row<Cfg::RB>(Xt, q0 + lr, ks, half);
  }
  // dP^T = V dO^T
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[kt][r] = 0.f;
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks)
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_row<Cfg::RB>(Vt, kt * 32 + lr, ks, half), df[ks], dp[kt], 0, 0, 0);
  } else {
#pragma unroll 4
    for (int kk = 0; kk < HD / 2; ++kk) {
      const float dv = dobase[(long)(q0 + lr) * a.lddo + 2 * kk + half];
#pragma unroll
      for (int kt = 0; kt < Cfg::KT; ++kt)
        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vbase[(long)(kt * 32 + lr) * a.ld + 2 * kk + half], dv, dp[kt], 0, 0, 0);
    }
  }
  stamp(3);
  float delta = 0.f;
  if constexpr (F8) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[kt][r] *= sv;                              // dP = sv (v^ dO)
  }
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) delta += p[kt][r] * dp[kt][r];
  delta += __shfl_xor(delta, 32);
#pragma unroll
  for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[kt][r] = p[kt][r] * (dp[kt][r] - delta);     // dp now holds dS^T
  // relative-position-bias gradient (expanded, transposed table): summed over this workgroup's problems in registers
  if (live) {
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dbacc[kt][r] += dp[kt][r];
  }
  store_qk_tile<T, NTOK>(Pt, p, q0);
  store_qk_tile<T, NTOK>(St, dp, q0);
  stamp(4);
  __syncthreads();                               // P, dS complete; every wave is done with V
  // dqkv rows of this problem through a buffer descriptor on (dqkv + the problem's first element): wave-uniform base, per lane
  // (4 half ldo + lr) elements, per store a scalar (row, column tile) offset - no 64-bit address pair per output row (round 5: the 16
  // hoisted row addresses were what this kernel spilled at its 512-register cap; a problem spans < 2 MB, so 32-bit offsets always reach)
  void* dqb = (void*)uniform_ptr((T*)a.out + rowbase * a.ldo + head * HD);
  const int esz = (int)sizeof(T);
  const int so_k = a.C * esz, so_v = 2 * a.C * esz;
  const int ldo_b = (int)a.ldo * esz;
  const int k0 = qt * 32;                        // this wave's KEY tile for dV / dK

  f32x16 acc[Cfg::DT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  };
  auto out_voff = [&]() -> int {                 // (formed at its use: two lane constants, not kept live across the problem loop)
    int lv = l;
    asm volatile("" : "+v"(lv));
    return (4 * (lv >> 5)) * ldo_b + (lv & 31) * esz;
  };
  // dQ keeps the untransposed product (registers = query rows): its column sums - the q third of the qkv bias gradient -
  // are then in-lane.  `csacc` carries them across this workgroup's problems.
  // (the scalar row offsets are formed from an opaque copy of the pitch at every store group: as loop invariants of the persistent
  //  problem loop hipcc kept all 16 x 3 of them in SGPRs, ran out, and spilled SGPRs into VGPR lanes at the VGPR cap)
  auto store_acc = [&](int sbase, int row0, float mul) {
    const int vo = out_voff();
    int ld_s = ldo_b;
    asm volatile("" : "+s"(ld_s));
#pragma unroll
    for (int dt = 0; dt < Cfg::DT; ++dt) {
      float csum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const T o = from_f32<T>(acc[dt][r] * mul);
        if (live) buf_store_elem<T>(dqb, to_f32<T>(o), vo + dt * 32 * esz, sbase + (row0 + crow32(r, 0)) * ld_s);
        csum += to_f32<T>(o);
      }
      if (live) csacc[dt] += csum;
    }
  };
  // dV / dK need no column sums: sum_key dK = 0 (rows of dS sum to zero) and sum_key dV = column sums of dO (softmax
  // rows sum to one), which the caller takes from the GEMM that produced dO.
  auto store_acc_plain = [&](int sbase, int row0, float mul = 1.0f) {
    if (!live) return;
    const int vo = out_voff();
    int ld_s = ldo_b;
    asm volatile("" : "+s"(ld_s));
#pragma unroll
    for (int dt = 0; dt < Cfg::DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        buf_store_elem<T>(dqb, F8 ? acc[dt][r] * mul : acc[dt][r], vo + dt * 32 * esz, sbase + (row0 + crow32(r, 0)) * ld_s);
  };

  // ---- dV[key][d] = sum_q P[q][key] dO[q][d]
  if constexpr (TT<T>::IS_BF16) {
    if constexpr (PF) {
      wait_vm0();                                // dO (requested before the scores) has landed
      __syncthreads();
      stage_tile<NTOK, Cfg::RB>(Vt, (const char*)qbase, a.ld * sizeof(T), qt, Cfg::QW);   // V is dead: Q for dK
    } else {
      stage_tile<NTOK, Cfg::RB>(Vt, (const char*)dobase, a.lddo * sizeof(T), qt, Cfg::QW);
      wait_vm0();
      __syncthreads();
    }
  }
  stamp(5);
  zero_acc();
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int ks = 0; ks < NTOK / 16; ++ks) {
      const bf16x8 pa = frag_tr<Cfg::PRB>(Pt, ks, qt);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_tr<Cfg::RB>(Xt, ks, dt), acc[dt], 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int kk = 0; kk < NTOK / 2; ++kk) {
      const int qq = 2 * kk + half;
      const float pa = tile_elem_f32<T, Cfg::PRB>(Pt, qq, k0 + lr);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa, dobase[(long)qq * a.lddo + dt * 32 + lr], acc[dt], 0, 0, 0);
    }
  }
  stamp(6);
  if constexpr (PF) {
    wait_vm0();                                  // Q has landed (waited for BEFORE the dV stores are issued)
    __syncthreads();                             // ... and every wave is done with the dO tile
    store_acc_plain(so_v, k0);
    if (grp + gridDim.x < ngroups) {             // the next problem's K into the buffer dO just left
      const T* nq = (const T*)a.qkv + problem_rowbase(grp + gridDim.x) * a.ld + head * HD;
      stage_tile<NTOK, Cfg::RB>(Xt, (const char*)(nq + a.C), a.ld * sizeof(T), qt, Cfg::QW);
    }
  } else {
    store_acc_plain(so_v, k0);
  }

  stamp(7);
  // ---- dK[key][d] = sum_q dS[q][key] q_s[q][d]
  if constexpr (TT<T>::IS_BF16 && !PF) {
    __syncthreads();                             // all waves finished reading the dO tile
    if constexpr (F8) {
      stage_tile<NTOK, HD>(Vt + Cfg::KV_BYTES / 2, q8b, a.ld, qt, Cfg::QW);
      wait_vm0();
      __syncthreads();
      F8Expand<NTOK, HD, 64> eq;
      eq.load(Vt + Cfg::KV_BYTES / 2, l);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      eq.store(Vt, l);
    } else {
      stage_tile<NTOK, Cfg::RB>(Vt, (const char*)qbase, a.ld * sizeof(T), qt, Cfg::QW);
      wait_vm0();
    }
    __syncthreads();
  }
  zero_acc();
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int ks = 0; ks < NTOK / 16; ++ks) {
      const bf16x8 sa = frag_tr<Cfg::PRB>(St, ks, qt);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, frag_tr<Cfg::RB>(Vt, ks, dt), acc[dt], 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int kk = 0; kk < NTOK / 2; ++kk) {
      const int qq = 2 * kk + half;
      const float sa = tile_elem_f32<T, Cfg::PRB>(St, qq, k0 + lr);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa, qbase[(long)qq * a.ld + dt * 32 + lr], acc[dt], 0, 0, 0);
    }
  }
  store_acc_plain(so_k, k0, sq);                 // (F8: dK = sq (dS^T q^))
  stamp(8);

  // ---- dQ[q][d] = scale * sum_key dS[q][key] K[key][d]
  zero_acc();
  if constexpr (TT<T>::IS_BF16) {
#pragma unroll
    for (int ks = 0; ks < NTOK / 16; ++ks) {
      const bf16x8 sa = frag_row<Cfg::PRB>(St, q0 + lr, ks, half);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, frag_tr<Cfg::RB>(Kt, ks, dt), acc[dt], 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int kk = 0; kk < NTOK / 2; ++kk) {
      const int key = 2 * kk + half;
      const float sa = tile_elem_f32<T, Cfg::PRB>(St, q0 + lr, key);
#pragma unroll
      for (int dt = 0; dt < Cfg::DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa, kbase[(long)key * a.ld + dt * 32 + lr], acc[dt], 0, 0, 0);
    }
  }
  store_acc(0, q0, F8 ? a.scale * sk : a.scale);
  stamp(9);
  __syncthreads();                               // the next problem's tiles overwrite K / dS
  stamp(10);
  if constexpr (PF) {
    if (grp + gridDim.x < ngroups) {             // next V into buffer 1 (Q is dead); first needed after the next S^T
      const T* nq = (const T*)a.qkv + problem_rowbase(grp + gridDim.x) * a.ld + head * HD;
      stage_tile<NTOK, Cfg::RB>(smem + Cfg::KV_BYTES, (const char*)(nq + 2 * a.C), a.ld * sizeof(T), qt, Cfg::QW);
    }
    const int tb = kbuf; kbuf = xbuf; xbuf = tb;   // the spare buffer now holds the next K
  }
  }
  // ---- hand-over of the register-accumulated sums, WITHOUT atomics (bitwise reproducible): every wave parks its entries in a
  // private LDS plane ([NTOK key rows][32 queries] + [HD]); the workgroup then adds, in a fixed order, the T x T entries that share
  // a bias-table cell (key n = row mod N, query n = column mod N) and the waves of a problem that share a dq column, and stores ONE
  // partial slab per problem slot (blockIdx.x * PPB + sp) of a.slabs = [slots][N*N + HD]; slab_fold_kernel adds the slots of a head
  // in slot order behind the launch (launch_attn).
  if (a.slabs && !dbg_ts) {
    constexpr int PLANE = NTOK * 32 + HD;
    const int N = NC ? NC : a.N;
    __syncthreads();                               // every tile is dead
    float* plane = (float*)smem + (long)w * PLANE;
#pragma unroll
    for (int kt = 0; kt < Cfg::KT; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) plane[(kt * 32 + crow32(r, half)) * 32 + lr] = dbacc[kt][r];
#pragma unroll
    for (int dt = 0; dt < Cfg::DT; ++dt) {
      float csum = csacc[dt];
      csum += __shfl_xor(csum, 32);
      if (half == 0) plane[NTOK * 32 + dt * 32 + lr] = csum;
    }
    __syncthreads();
    const float* pl0 = (const float*)smem + (long)sp * Cfg::QW * PLANE;
    float* slab = a.slabs + ((long)blockIdx.x * Cfg::PPB + sp) * ((long)N * N + HD);
    const int tl = threadIdx.x - sp * Cfg::QW * 64;
    for (int e = tl; e < N * N; e += Cfg::QW * 64) {
      const int kn = e / N, qn = e - kn * N;
      float t = 0.f;
      for (int x = qn; x < NTOK; x += N)
        for (int y = kn; y < NTOK; y += N) t += pl0[(long)(x >> 5) * PLANE + y * 32 + (x & 31)];
      slab[e] = t;
    }
    for (int c = tl; c < HD; c += Cfg::QW * 64) {
      float t = 0.f;
#pragma unroll
      for (int q2 = 0; q2 < Cfg::QW; ++q2) t += pl0[(long)q2 * PLANE + NTOK * 32 + c];
      slab[(long)N * N + c] = t;
    }
  }
}

// ====================================================================================================
// Backward for the stage-1 shape (bf16, 128 tokens, head dim 128) with EIGHT waves per workgroup.
// The 160 KB of LDS a problem needs (K, V/Q, dO tiles + P + dS) allow one workgroup per CU, and with four waves (one per
// SIMD) every phase of the kernel above exposes its full latency: 2.2 us of MFMA work took 15 us per problem
// (profiles/r01_v9_attn_bwd_timeline.txt), 3.2 TB/s = 40 % of the HBM rate.  Here two waves share every query tile:
//   scores / softmax / dP / dS : wave (qt, hw) owns the keys 64 hw .. 64 hw + 63 of query tile qt; the softmax statistics
//     (max, sum) and the row sums of P o dP are exchanged with the partner wave through 3 KB of LDS
//     (online-softmax merge: one exchange for max + sum);
//   dV / dK / dQ : waves 0-3 multiply dV (4 key tiles) and the first half of dQ's columns, waves 4-7 the second half of dQ
//     and dK - 1.5 tile products per wave instead of 3;
//   every LDS-DMA tile copy is issued by all eight waves.
// Same persistent schedule, buffer rotation (K | V -> Q | dO -> next K) and register-accumulated bias / q-bias gradients.
// Two waves per SIMD leave 256 registers per lane: LDS addresses are lane constants + instruction immediates (16 registers
// instead of one per unrolled read) and the bias-table reads are buffer instructions with ONE per-lane offset and scalar row
// offsets.
// Output stores stay two-byte row stores (64 per lane and product, as buffer instructions with scalar row offsets): routing them
// through a private 4 KB of the dead P tile as whole 16-byte row pieces (4 store instructions of 1 KB instead of 32 of 128 B)
// measured only 3 % faster (145.7 -> 141.1 us) - the kernel is bound by exposed load latency, not by its stores - and the P
// tile's buffer is better spent on the next problem's V (below).
DEVI bf16x8 lds_tr_pair(const char* tile, int off0, int off1, int imm) {
  const short4v t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + off0 + imm));
  const short4v t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(tile + off1 + imm));
  return cat4(__builtin_bit_cast(bf16x4, t0), __builtin_bit_cast(bf16x4, t1));
}

// QPF (round 4, the default): ALL of the next problem's q | k | v tiles are prefetched half a problem ahead, and its bias-table values
// at the end of the current one.
//  * The scores start right after the top-of-problem barrier with the Q row pieces read from LDS.  Before, V was prefetched instead of
//    Q and the Q row pieces came from global memory at the top of the problem: a first-touch HBM round trip of ~3 us per problem in
//    front of the first MFMA (profiles/r03_attn_bwd8_timeline.txt: 3.4 of 12.9 us).  Only dO is requested at the top; it lands
//    behind the scores + softmax.  P overwrites V once dP is done; Q is never re-staged.
//  * product 1 = dV (waves 0-3) and dK (waves 4-7), product 2 = the two halves of dQ: after product 1 the buffers of dO, P and Q are all
//    dead and take the next K, Q and V; every wave multiplies 4 + 2 tiles and issues 64 + 32 stores (before: 4 + 2 against 2 + 4 with
//    the stores of one group queueing behind the other's, 1.6 us of end-of-problem wait on waves 0-3).
//  * Buffer roles (five 32 KB buffers): K | Q | V -> P | dO | dS; rotation at the end: K <- dO's, Q <- P's, V <- Q's, dO <- K's buffer.
//  * The copies are raw ISA and the lane-constant LDS offsets are formed at their use (see stage_tile, ld_row, tro): at the 256-register
//    cap every spilled constant and every scratch-resident table came back through s_waitcnt vmcnt(0), i.e. behind the output stores
//    and the prefetch in flight - that, not the arithmetic, was most of the "latency-bound" phases of the round-2/3 timelines.
// F8 (with QPF; BASELINE configs[4]): q | k | v arrive as e4m3 bytes + per-(window, head) scales (stswin_gemm_nt_qkv_fp8).  The next
// problem's K, Q and V are prefetched as fp8 tiles (16 KB each) into the UPPER halves of the three buffers they will live in; at the top
// of a problem every thread takes its 2 x 16 bytes of each tile into registers, and after a barrier writes the bf16 image over the whole
// buffer - from there on the kernel is the bf16 kernel.  e4m3 values are exact in bf16, so the scales ride on the accumulators:
// S = sq sk (q^ k^), dP = sv (v^ dO), dQ = scale sk (dS k^), dK = sq (dS^T q^): the probabilities are those of the fp8 forward
// up to its own e4m3 rounding of P.  The backward then reads 131 KB per problem instead of 229 KB.
template <int NC, bool QPF = false, bool F8 = false>
__global__ __launch_bounds__(512) void attn_bwd8_kernel(AttnArgs a) {
  using T = bf16;
  static_assert(!F8 || QPF, "the fp8 variant uses the round-4 schedule");
  constexpr int NTOK = 128, HD = 128, ROWB = 256;          // K / V / Q / dO tiles and the P / dS tiles all have 256-byte rows
  using Cfg = AttnCfg<T, NTOK, HD>;
  constexpr int KV = Cfg::KV_BYTES;
  static_assert(Cfg::RB == ROWB && Cfg::PRB == ROWB, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int l = threadIdx.x & 63, lr = l & 31, half = l >> 5, w = wave_id();
  const int qt = w & 3, hw = w >> 2;
  const long ngroups = (long)a.nB_ * a.heads;
  const int head = (int)(blockIdx.x % a.heads);
  f32x16 dbacc[2];
  float csacc[2] = {0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int r = 0; r < 16; ++r) dbacc[kk][r] = 0.f;
  // five 32 KB buffers; roles of a problem: K | V, later Q | dO, later the NEXT K | P, later the NEXT V | dS (fixed).
  // Both tiles of the next problem are requested right after the barrier that ends dV, half a problem ahead.
  int kb = 0, vb = 1, xb = 2, pb = 3;
  char* St = smem + 4 * KV;
  float2* ex1 = (float2*)St;                       // [2 key halves][128 queries]: (max, sum of exp) of the wave's 64 keys
  float* ex2 = (float*)(St + 2048);                // [2][128]: sum over the wave's keys of P o dP
  const int N = NC ? NC : a.N;
  const int q0 = qt * 32, qn = (q0 + lr) % N;
  // ---- lane constants of the LDS images (tile_off<256>: byte = row * 256 + ((chunk ^ swz256(row)) & 15) * 16)
  // row fragments of rows X + lr (X a multiple of 32: swz256 sees lr only): offset of k-step ks = rbase + (((2 ks) ^ rsx) & 15) * 16 + X * 256
  // (two lane constants and two VALU instructions per read instead of eight constants: the kernel sits at its 256-register cap and
  // every spilled constant comes back through a scratch load + s_waitcnt vmcnt(0), i.e. behind the output stores in flight)
  const int rbase = lr * ROWB, rsx = half ^ swz256(lr);
  // transposed fragments (frag_tr): rows 16 ks + 8 (l >> 5) + 4 e + q, columns 32 ct + 16 ((l >> 4) & 1) + 4 p.  The byte offset of column
  // tile ct is trb[e] + (((4 ct) ^ trx[e]) << 4): two lane constants per row pair instead of a [4][2] table - indexed with a run-time
  // column tile (qt, hw) hipcc put that table into SCRATCH memory, and every scratch read is followed by s_waitcnt vmcnt(0): the
  // products waited for their own output stores and for the tile prefetch of the next problem (profiles/r04_attn_bwd8_timeline.txt).
  int trb[2], trx[2];
  {
    const int q = (l & 15) >> 2, pp = l & 3;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = 8 * half + 4 * e + q;
      trb[e] = row * ROWB + (pp & 1) * 8;
      trx[e] = (2 * ((l >> 4) & 1) + (pp >> 1)) ^ swz256(row);
    }
  }
  auto tro = [&](int ct, int e) -> int {
    int tx = trx[e];
    asm("" : "+v"(tx));
    return trb[e] + ((((4 * ct) ^ tx) & 15) << 4);
  };
  auto ld_row = [&](const char* tile, int X, int ks) -> bf16x8 {
    int sx = rsx;
    asm("" : "+v"(sx));                            // (opaque: the offset is formed here, not hoisted out of the persistent loop)
    return *(const bf16x8*)(tile + X * ROWB + rbase + ((((2 * ks) ^ sx) & 15) << 4));
  };
  auto ld_tr = [&](const char* tile, int ks, int ct) -> bf16x8 { return lds_tr_pair(tile, tro(ct, 0), tro(ct, 1), ks * 16 * ROWB); };
  {
    const long rb0 = (long)(blockIdx.x / a.heads) * NTOK;
    if constexpr (F8) {                              // (a.ld counts bytes; fp8 tiles have 128-byte rows and sit in the upper buffer halves)
      const char* q0b = (const char*)a.qkv + rb0 * a.ld + head * HD;
      stage_tile<NTOK, HD, true>(smem + KV / 2, q0b + a.C, a.ld, w, 8);
      stage_tile<NTOK, HD, true>(smem + KV + KV / 2, q0b, a.ld, w, 8);
      stage_tile<NTOK, HD, true>(smem + 2 * KV + KV / 2, q0b + 2 * a.C, a.ld, w, 8);
    } else {
      const T* q0b = (const T*)a.qkv + rb0 * a.ld + head * HD;
      stage_tile<NTOK, ROWB, QPF>(smem, (const char*)(q0b + a.C), a.ld * sizeof(T), w, 8);
      stage_tile<NTOK, ROWB, QPF>(smem + KV, (const char*)(QPF ? q0b : q0b + 2 * a.C), a.ld * sizeof(T), w, 8);
      if constexpr (QPF) stage_tile<NTOK, ROWB, QPF>(smem + 2 * KV, (const char*)(q0b + 2 * a.C), a.ld * sizeof(T), w, 8);   // V (buffer xb)
    }
  }
  // bias (+ mask) values of a problem's 2 x 16 (key, query) entries per lane
  float tb[2][16];
  auto load_tb = [&](int b_) {
    const int widx = b_ % a.nW;
    // (scalar load: a vector load of the wave-uniform slot index would be followed by s_waitcnt vmcnt(0) - at the end of a problem
    //  that is a wait for the 64 output stores and the next problem's tile copies)
    const int slot = (a.bias_windows & 0xffffff) > 1 ? (a.bias_index ? sload(a.bias_index, widx) : widx) : 0;
    if constexpr (NC == 64) {                      // key n = (kt & 1) * 32 + crow32(r, half): no wrap inside a 32-key tile
      const int tvoff = (4 * half * NC + qn) * 4;
      const int tsoff = (slot * a.heads + head) * NC * NC * 4;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[kk][r] = buf_load_f32(a.biasT, tvoff, tsoff + (kk * 32 + crow32(r, 0)) * NC * 4);
    } else {
      const float* bt = a.biasT + ((long)slot * a.heads + head) * N * N + qn;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[kk][r] = bt[(((2 * hw + kk) * 32 + crow32(r, half)) % N) * N];
    }
    if (a.maskT) {
      const float* mt = a.maskT + (long)widx * N * N + qn;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[kk][r] += mt[(((2 * hw + kk) * 32 + crow32(r, half)) % N) * N];
    }
  };
  if constexpr (QPF) load_tb((int)(blockIdx.x / a.heads));
  // Output stores are two-byte stores.  Round 4 re-measured the alternatives now that no other stall hides them: a 4 x 4 register
  // transpose (DPP + v_perm, 10 VALU per 4 registers) to 8-byte stores - a quarter of the instructions, each touching 8 rows x 64 B -
  // ran 7 % SLOWER (134.8 vs 126.1 us): the cost follows the row pieces touched, not the instruction count.  See store_pair below.
  const int out_voff = l * 2;                      // per-lane part of every dqkv store address: 64 consecutive columns of one row
  const bool dbg_ts = (a.bias_windows & (1 << 30)) != 0;   // DBG (tools/attn_timeline.py): dqkv_colsum = u64 [workgroups][32] stamps
  bool stamp_on = true;                            // (the timeline shows a STEADY-STATE problem: the last one, which requests nothing, is skipped)
  auto stamp = [&](int slot) {
    if (dbg_ts && stamp_on && (threadIdx.x & 255) == 0)
      ((unsigned long long*)a.dqkv_colsum)[(long)blockIdx.x * 32 + hw * 16 + slot] = wall_clock64();
  };
  stamp(12);                                       // (workgroup start: slot 12 is written once)
  for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b_ = (int)(grp / a.heads);
    const long rowbase = (long)b_ * NTOK;
    const T* qbase = (const T*)a.qkv + rowbase * a.ld + head * HD;        // (bf16 storage only)
    const T* dobase = (const T*)a.dout + rowbase * a.lddo + head * HD;
    char* Kt = smem + kb * KV;
    char* Vt = smem + (QPF ? xb : vb) * KV;        // V, later Q                  (QPF: V, later P, later the next problem's Q)
    char* Xt = smem + (QPF ? pb : xb) * KV;        // dO, later the next problem's K
    char* Pt = QPF ? Vt : smem + pb * KV;          // P, later the next problem's V
    char* Qt = smem + vb * KV;                     // QPF: Q (prefetched)
    const bool has_next = grp + gridDim.x < ngroups;
    const T* nqbase = (const T*)a.qkv + (long)((grp + gridDim.x) / a.heads) * NTOK * a.ld + head * HD;
    stamp_on = has_next || ngroups <= (long)gridDim.x;
    stamp(0);
    float sq = 1.f, sk = 1.f, sv = 1.f;            // F8: value = e4m3 byte * scale of (window, head, q | k | v)
    if constexpr (F8) {
      sq = sloadf(a.qscale, (long)b_ * a.ld_scale + head);
      sk = sloadf(a.qscale, (long)b_ * a.ld_scale + a.heads + head);
      sv = sloadf(a.qscale, (long)b_ * a.ld_scale + 2 * a.heads + head);
    }
    bf16x8 qf[HD / 16];
    if constexpr (!QPF) {
      load_row_frags<T, HD>(qf, qbase, a.ld, q0 + lr);
      // ---- bias (+ mask) values: requested BEFORE the tile wait
      load_tb(b_);
    }
    wait_vm0();
    __syncthreads();                               // K and V | Q (requested half a problem ago) have landed (+ the Q row pieces)
    stamp(1);
    if constexpr (QPF) {
      // the table values were requested one problem ago and are complete behind the wait above; as asm outputs hipcc no longer treats
      // them as pending loads (it would otherwise wait vmcnt(0) at their first use - with the V / dO copies below in flight)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(tb[kk][r]));
      int lv = l;                                  // (re-derive the copies' lane offsets here: see stage_tile)
      asm volatile("" : "+v"(lv));
      stage_tile<NTOK, ROWB, QPF>(Xt, (const char*)dobase, a.lddo * sizeof(T), w, 8, lv);   // (V was prefetched with K and Q)
      if constexpr (F8) {                          // fp8 tiles (upper buffer halves) -> bf16 images over the whole buffers
        int tv = threadIdx.x;
        asm volatile("" : "+v"(tv));
        F8Expand<NTOK, HD, 512> ek, eq, ev;
        ek.load(Kt + KV / 2, tv); eq.load(Qt + KV / 2, tv); ev.load(Vt + KV / 2, tv);
        __syncthreads();                           // every thread holds its bytes: the images may overwrite the fp8 halves
        ek.store(Kt, tv); eq.store(Qt, tv); ev.store(Vt, tv);
        __syncthreads();
      }
    } else {
      stage_tile<NTOK, ROWB, QPF>(Xt, (const char*)dobase, a.lddo * sizeof(T), w, 8);
    }
    // ---- S^T for this wave's two key tiles: lane = query column, registers = keys
    f32x16 p[2], dp[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[kk][r] = 0.f;
    if constexpr (QPF) {                           // the Q row pieces come from LDS one k-step at a time: no 32-register fragment set
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const bf16x8 qk = ld_row(Qt, q0, ks);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          p[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Kt, (2 * hw + kk) * 32, ks), qk, p[kk], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks)
          p[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Kt, (2 * hw + kk) * 32, ks), qf[ks], p[kk], 0, 0, 0);
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (F8) p[kk][r] = p[kk][r] * (sq * sk) + tb[kk][r];
        else p[kk][r] += tb[kk][r];
        mx = fmaxf(mx, p[kk][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(p[kk][r] - mx);
        p[kk][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    int le = l;                                    // (QPF: exchange indices re-derived, not kept live over the loop)
    if constexpr (QPF) asm volatile("" : "+v"(le));
    const int exi = q0 + (le & 31);
    if ((le >> 5) == 0) ex1[hw * NTOK + exi] = make_float2(mx, sum);
    stamp(2);
    wait_vm0();                                    // this wave's pieces of the dO tile
    __syncthreads();                               // partner statistics + dO tile visible
    stamp(3);
    {
      const float2 o = ex1[(hw ^ 1) * NTOK + exi];
      const float m = fmaxf(mx, o.x);
      const float mine = __expf(mx - m);
      const float inv = mine / (sum * mine + o.y * __expf(o.x - m));
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[kk][r] *= inv;
    }
    // ---- dP^T = V dO^T on the same key tiles (the dO row pieces replace Q's in qf)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[kk][r] = 0.f;
    if constexpr (QPF) {
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) {
        const bf16x8 ok = ld_row(Xt, q0, ks);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          dp[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Vt, (2 * hw + kk) * 32, ks), ok, dp[kk], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < HD / 16; ++ks) qf[ks] = ld_row(Xt, q0, ks);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ks = 0; ks < HD / 16; ++ks)
          dp[kk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_row(Vt, (2 * hw + kk) * 32, ks), qf[ks], dp[kk], 0, 0, 0);
    }
    float delta = 0.f;
    if constexpr (F8) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[kk][r] *= sv;                            // dP = sv (v^ dO)
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) delta += p[kk][r] * dp[kk][r];
    delta += __shfl_xor(delta, 32);
    if ((le >> 5) == 0) ex2[hw * NTOK + exi] = delta;
    stamp(4);
    __syncthreads();
    delta += ex2[(hw ^ 1) * NTOK + exi];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dp[kk][r] = p[kk][r] * (dp[kk][r] - delta);          // dS^T
        dbacc[kk][r] += dp[kk][r];
      }
    stamp(5);
    __syncthreads();                               // every wave has read the exchange area: P / dS may overwrite it
    // P and dS tiles [query row][key] (bf16): this wave's 32 query rows x 64 keys
    // (QPF: the eight store offsets are re-derived here instead of living in registers across the whole persistent loop)
    int lq = l;
    if constexpr (QPF) asm volatile("" : "+v"(lq));
    const int lrq = lq & 31, halfq = lq >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 vp, vs;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vp[e] = (bf16)p[kk][4 * g + e]; vs[e] = (bf16)dp[kk][4 * g + e]; }
        const int off = (q0 + lrq) * ROWB + (((((2 * hw + kk) * 4 + g) ^ swz256(lrq)) & 15) << 4) + halfq * 8;
        *(bf16x4*)(Pt + off) = vp;
        *(bf16x4*)(St + off) = vs;
      }
    __syncthreads();                               // P, dS complete; every wave is done with V
    stamp(6);
    if constexpr (!QPF) stage_tile<NTOK, ROWB, QPF>(Vt, (const char*)qbase, a.ld * sizeof(T), w, 8);      // V is dead: Q (for dK)
    // dqkv row rowbase + .., columns head * HD + ..: scalar byte offsets of the three thirds
    const int so_q = (int)((rowbase * a.ldo + head * HD) * 2), so_k = so_q + a.C * 2, so_v = so_k + a.C * 2;
    const int k0 = qt * 32;
    f32x16 acc[4];
    auto zero4 = [&]() {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    };
    // Two neighbouring 32-column tiles hold, per register, rows r' (lanes 0-31) and r' + 4 (lanes 32-63) of their columns.  ONE
    // v_permlane32_swap per register pair (upper half of tile A's register <-> lower half of tile B's) leaves register A with row r'
    // and register B with row r' + 4, each across all 64 lanes = 64 consecutive columns: every store instruction then writes ONE whole
    // 128-byte line instead of two 64-byte halves of two different lines.
    auto store_pair = [&](float va, float vb, int soff) {       // soff: byte offset of (row r', first column of tile A)
      const bf16 oa = (bf16)va, ob = (bf16)vb;
      const unsigned ua = (unsigned)__builtin_bit_cast(unsigned short, oa), ub = (unsigned)__builtin_bit_cast(unsigned short, ob);
      const auto sw = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
      buf_store_b16(a.out, __builtin_bit_cast(bf16, (unsigned short)sw[0]), out_voff, soff);
      buf_store_b16(a.out, __builtin_bit_cast(bf16, (unsigned short)sw[1]), out_voff, soff + 4 * (int)a.ldo * 2);
    };
    auto store4 = [&](int sbase, int row0, float mul = 1.0f) {
#pragma unroll
      for (int dt = 0; dt < 4; dt += 2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (F8) store_pair(acc[dt][r] * mul, acc[dt + 1][r] * mul, sbase + ((row0 + crow32(r, 0)) * (int)a.ldo + dt * 32) * 2);
          else store_pair(acc[dt][r], acc[dt + 1][r], sbase + ((row0 + crow32(r, 0)) * (int)a.ldo + dt * 32) * 2);
        }
    };
    // acc[0..3] += A^T(tile at, column tile qt) x B(tile bt) over the 128 contraction rows: dV (P, dO) and dK (dS, Q).
    // (the operands of step ks + 1 are requested before the MFMAs of step ks; the scheduling barrier keeps hipcc from
    // hoisting every step's LDS reads to the top)
    auto tr_product = [&](const char* at, const char* btile) {
      bf16x8 pa = ld_tr(at, 0, qt), xb[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) xb[dt] = ld_tr(btile, 0, dt);
#pragma unroll
      for (int ks = 0; ks < NTOK / 16; ++ks) {
        bf16x8 pa_n = pa, xb_n[4] = {xb[0], xb[1], xb[2], xb[3]};
        if (ks + 1 < NTOK / 16) {
          pa_n = ld_tr(at, ks + 1, qt);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) xb_n[dt] = ld_tr(btile, ks + 1, dt);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, xb[dt], acc[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        pa = pa_n;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) xb[dt] = xb_n[dt];
      }
    };
    // dQ[q][d] = scale * sum_key dS[q][key] K[key][d] for the two head-dim tiles 2 hw, 2 hw + 1 (acc[0], acc[1])
    auto dq_half = [&]() {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dd][r] = 0.f;
      bf16x8 sa = ld_row(St, q0, 0), kb[2];
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) kb[dd] = ld_tr(Kt, 0, 2 * hw + dd);
#pragma unroll
      for (int ks = 0; ks < NTOK / 16; ++ks) {
        bf16x8 sa_n = sa, kb_n[2] = {kb[0], kb[1]};
        if (ks + 1 < NTOK / 16) {
          sa_n = ld_row(St, q0, ks + 1);
#pragma unroll
          for (int dd = 0; dd < 2; ++dd) kb_n[dd] = ld_tr(Kt, ks + 1, 2 * hw + dd);
        }
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) acc[dd] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, kb[dd], acc[dd], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        sa = sa_n; kb[0] = kb_n[0]; kb[1] = kb_n[1];
      }
    };
    const float qmul = F8 ? a.scale * sk : a.scale;   // dQ = scale (dS k) = scale sk (dS k^)
    auto dq_store = [&]() {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        float csum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) csum += (float)(bf16)(acc[dd][r] * qmul);        // (sums of the ROUNDED values, own column)
        csacc[dd] += csum;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        store_pair(acc[0][r] * qmul, acc[1][r] * qmul, so_q + ((q0 + crow32(r, 0)) * (int)a.ldo + 2 * hw * 32) * 2);
    };
    if constexpr (QPF) {
      // product 1: dV (waves 0-3: P, dO) and dK (waves 4-7: dS, Q) - four tile products each; after it P, dO AND Q are dead, so the
      // next problem's K, Q and V all go out here (into dO's, P's and Q's buffers): only dO is left to request at the next top.
      // product 2: both wave groups multiply their half of dQ (dS, K).  Same work and the same number of stores in every wave.
      zero4();
      if (hw == 0) tr_product(Pt, Xt);             // dV[key][d] = sum_q P[q][key] dO[q][d]
      else tr_product(St, Qt);                     // dK[key][d] = sum_q dS[q][key] q_s[q][d]
      stamp(7);
      __syncthreads();                             // every wave is done with dO, P and Q
      stamp(8);
      store4(hw == 0 ? so_v : so_k, k0, hw == 0 ? 1.0f : sq);        // (F8: dK = sq (dS^T q^))
      if (has_next) {
        int lv = l;
        asm volatile("" : "+v"(lv));
        if constexpr (F8) {                          // fp8 tiles into the upper halves of the buffers they will be expanded in
          const char* nq8 = (const char*)a.qkv + (long)((grp + gridDim.x) / a.heads) * NTOK * a.ld + head * HD;
          stage_tile<NTOK, HD, true>(Xt + KV / 2, nq8 + a.C, a.ld, w, 8, lv);
          stage_tile<NTOK, HD, true>(Pt + KV / 2, nq8, a.ld, w, 8, lv);
          stage_tile<NTOK, HD, true>(Qt + KV / 2, nq8 + 2 * a.C, a.ld, w, 8, lv);
        } else {
          stage_tile<NTOK, ROWB, true>(Xt, (const char*)(nqbase + a.C), a.ld * sizeof(T), w, 8, lv);       // next K
          stage_tile<NTOK, ROWB, true>(Pt, (const char*)nqbase, a.ld * sizeof(T), w, 8, lv);               // next Q
          stage_tile<NTOK, ROWB, true>(Qt, (const char*)(nqbase + 2 * a.C), a.ld * sizeof(T), w, 8, lv);   // next V
        }
      }
      stamp(9);
      dq_half();
      dq_store();
      __builtin_amdgcn_sched_barrier(0);           // (not hoisted above the products: 32 more live registers there would spill)
      if (has_next) load_tb((int)((grp + gridDim.x) / a.heads));   // next problem's table values: complete long before its top
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (hw == 0) {                               // dV[key][d] = sum_q P[q][key] dO[q][d]
        zero4();
        tr_product(Pt, Xt);
      } else {
        dq_half();
      }
      stamp(7);
      wait_vm0();                                  // Q has landed (waited for BEFORE this phase's stores are issued)
      __syncthreads();                             // ... and every wave is done with the dO tile
      stamp(8);
      if (hw == 0) store4(so_v, k0);
      else dq_store();
      if (has_next) {                              // the next problem's K and V into the buffers dO and P just left
        stage_tile<NTOK, ROWB, false>(Xt, (const char*)(nqbase + a.C), a.ld * sizeof(T), w, 8);
        stage_tile<NTOK, ROWB, false>(Pt, (const char*)(nqbase + 2 * a.C), a.ld * sizeof(T), w, 8);
      }
      stamp(9);
      if (hw == 0) {
        dq_half();
        dq_store();
      } else {                                     // dK[key][d] = sum_q dS[q][key] q_s[q][d]
        zero4();
        tr_product(St, Vt);
        store4(so_k, k0);
      }
    }
    stamp(10);
    __syncthreads();                               // the next problem's tiles overwrite Q / K / dS
    stamp(11);
    stamp_on = true;
    stamp(14);                                     // (end of the workgroup's latest problem, whichever it is)
    if constexpr (QPF) { const int t0 = kb, t1 = vb; kb = pb; vb = xb; xb = t1; pb = t0; }   // K <- dO's, Q <- P's, V <- Q's, dO <- K's buffer
    else { const int t0 = kb, t1 = vb; kb = xb; vb = pb; xb = t0; pb = t1; }     // next K sits in xb, next V in pb
  }
  // ---- deterministic hand-over (see attn_bwd_kernel): wave (qt, hw) parks its 64 key rows x 32 queries + its 64 dq columns in a
  // private LDS plane; fixed-order sums of the T x T entries per table cell and of the four query tiles per dq column; one slab per
  // workgroup
  if (a.slabs && !dbg_ts) {
    constexpr int PLANE = 64 * 32 + 64;
    __syncthreads();
    float* plane = (float*)smem + (long)w * PLANE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int r = 0; r < 16; ++r) plane[(kk * 32 + crow32(r, half)) * 32 + lr] = dbacc[kk][r];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
      float csum = csacc[dd];
      csum += __shfl_xor(csum, 32);
      if (half == 0) plane[64 * 32 + dd * 32 + lr] = csum;
    }
    __syncthreads();
    const float* pl = (const float*)smem;
    float* slab = a.slabs + (long)blockIdx.x * ((long)N * N + HD);
    for (int e = threadIdx.x; e < N * N; e += 512) {
      const int kn = e / N, qn2 = e - kn * N;
      float t = 0.f;
      for (int x = qn2; x < NTOK; x += N)            // query tile x / 32, column x % 32
        for (int y = kn; y < NTOK; y += N)           // key half y / 64 (= hw), local row y % 64
          t += pl[(long)((y >> 6) * 4 + (x >> 5)) * PLANE + (y & 63) * 32 + (x & 31)];
      slab[e] = t;
    }
    for (int c = threadIdx.x; c < HD; c += 512) {    // dq column c: wave half c / 64, local column c % 64
      float t = 0.f;
#pragma unroll
      for (int q2 = 0; q2 < 4; ++q2) t += pl[(long)((c >> 6) * 4 + q2) * PLANE + 64 * 32 + (c & 63)];
      slab[(long)N * N + c] = t;
    }
  }
  stamp_on = true;
  stamp(13);                                       // (workgroup end, after the hand-over)
}

// ------------------------------------------------------------------------------------------------ C ABI
// dbiasT[h] += sum of the slots h, h + heads, ... of a.slabs (first N*N floats of a slot), dqkv_colsum[h * HD ..] += their last HD
// floats: fixed slot order (rowops.hip slab_fold_kernel) - the backward's parameter gradients are bitwise reproducible.
static thread_local long a_scratch_floats = 0;
static int attn_bwd_fold(const AttnArgs& a, int slots, int HD, hipStream_t st) {
  if (!a.slabs || (a.bias_windows & (1 << 30))) return 0;
  const int NN = a.N * a.N;
  const int len[3] = {NN, HD, 0};
  float* const out[3] = {a.dbiasT, a.dqkv_colsum, nullptr};
  const long obs[3] = {NN, HD, 0};
  return stswin_fold3_launch(a.slabs, (long)a.heads * (NN + HD), NN + HD, slots / a.heads, len, out, obs, a.heads, 1, st);
}

template <typename T, int NTOK, int HD, int NC>
static int launch_attn(const AttnArgs& a, bool bwd, hipStream_t st) {
  using Cfg = AttnCfg<T, NTOK, HD>;
  const long probs = (long)a.nB_ * a.heads;
  const int grid = (int)((probs + Cfg::PPB - 1) / Cfg::PPB);
  const int lds = bwd ? Cfg::BWD_LDS : Cfg::FWD_LDS;
  static_assert(Cfg::BWD_LDS <= 160 * 1024, "LDS budget");
  // raise the dynamic-LDS limit once per instantiation (not a stream operation: keep it out of graph capture)
  static const int attr_fwd = (int)hipFuncSetAttribute((const void*)attn_fwd_kernel<T, NTOK, HD, NC>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::FWD_LDS);
  static const int attr_bwd = (int)hipFuncSetAttribute((const void*)attn_bwd_kernel<T, NTOK, HD, NC>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::BWD_LDS);
  if (attr_fwd != 0 || attr_bwd != 0) return -(attr_fwd ? attr_fwd : attr_bwd);
  if constexpr (TT<T>::IS_BF16 && NTOK == 128 && HD == 128 && NC == 64) {      // (window size known: table offsets are scalar)
    const char* e4 = getenv("STSWIN_ATTN_BWD4");                             // A/B switch (read per call): the 4-wave kernel
    const int no8 = e4 && atoi(e4) ? 1 : 0;
    const bool fits = (long)a.nB_ * NTOK * a.ldo * 2 < 0x7FFF0000L;          // 32-bit buffer offsets of the dqkv stores
    if (bwd && !no8 && fits) {
      static const int attr8 = (int)hipFuncSetAttribute((const void*)attn_bwd8_kernel<NC, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        Cfg::BWD_LDS) |
                               (int)hipFuncSetAttribute((const void*)attn_bwd8_kernel<NC, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        Cfg::BWD_LDS);
      if (attr8 != 0) return -attr8;
      const char* eq = getenv("STSWIN_ATTN_BWD_QPF");                        // A/B switch (read per call): 0 = the round-3 schedule
      const bool qpf = !(eq && atoi(eq) == 0);
      int g = stswin_cu_budget();
      if (g > grid) g = grid;
      g = (g / a.heads) * a.heads;
      if (g < a.heads) g = a.heads;
      if (grid % a.heads == 0 && g >= a.heads) {
        if (a.slabs && (long)g * (a.N * a.N + HD) > a_scratch_floats) return -1205;
        if (qpf) hipLaunchKernelGGL((attn_bwd8_kernel<NC, true>), dim3(g), dim3(512), Cfg::BWD_LDS, st, a);
        else hipLaunchKernelGGL((attn_bwd8_kernel<NC, false>), dim3(g), dim3(512), Cfg::BWD_LDS, st, a);
        const int rf = attn_bwd_fold(a, g, HD, st);
        if (rf) return rf;
        STSWIN_CHECK_LAUNCH();
        return 0;
      }
    }
  }
  if (bwd) {
    // persistent: as many workgroups as fit the chip at once (LDS bound), rounded so that grid * PPB is a multiple of heads
    const int per_cu = (160 * 1024) / Cfg::BWD_LDS > 0 ? (160 * 1024) / Cfg::BWD_LDS : 1;
    int g = stswin_cu_budget() * per_cu;
    int step = a.heads;                          // smallest g granularity with (g * PPB) % heads == 0
    for (int d = Cfg::PPB; d > 1; --d)
      if (Cfg::PPB % d == 0 && a.heads % d == 0) { step = a.heads / d; break; }
    if (g > grid) g = grid;
    g = (g / step) * step;
    if (g < step) g = step;
    static_assert(4 * (NTOK * 32 + HD) * 4 <= Cfg::BWD_LDS, "LDS planes of the deterministic hand-over");
    if (a.slabs && (long)g * Cfg::PPB * (a.N * a.N + HD) > a_scratch_floats) return -1205;
    hipLaunchKernelGGL((attn_bwd_kernel<T, NTOK, HD, NC>), dim3(g), dim3(256), lds, st, a);
    const int rf = attn_bwd_fold(a, g * Cfg::PPB, HD, st);
    if (rf) return rf;
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<T, NTOK, HD, NC>), dim3(grid), dim3(256), lds, st, a);
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int dispatch_attn(const AttnArgs& a, int ntok, int hd, bool bwd, hipStream_t st) {
  // the reference model's two stages with ws*ws known at compile time (pair attention: 2 frames per window)
  if (ntok == 128 && hd == 128 && a.N == 64) return launch_attn<T, 128, 128, 64>(a, bwd, st);
  if (ntok == 32 && hd == 256 && a.N == 16) return launch_attn<T, 32, 256, 16>(a, bwd, st);
#define CASE(NT, D) if (ntok == NT && hd == D) return launch_attn<T, NT, D, 0>(a, bwd, st)
  CASE(128, 128); CASE(32, 256);      // the reference model (stage 1 / stage 2)
  CASE(128, 32);  CASE(32, 64);       // reduced-width test configuration (dim 128)
  CASE(128, 64);  CASE(32, 128);
  CASE(32, 32);
#undef CASE
  return -1201;
}

static int attn_common(int dtype, AttnArgs& a, int T_frames, int ws, bool bwd, void* stream) {
  const int ntok = T_frames * ws * ws;
  if (a.C % a.heads) return -1202;
  const int hd = a.C / a.heads;
  a.N = ws * ws;
  if (a.nW <= 0 || a.nB_ % a.nW) return -1203;
  return dtype == 0 ? dispatch_attn<bf16>(a, ntok, hd, bwd, (hipStream_t)stream)
                    : dispatch_attn<float>(a, ntok, hd, bwd, (hipStream_t)stream);
}

extern "C" int stswin_win_attn_fwd(int dtype, const void* qkv, long ld, void* out, long ldo, const float* biasT,
                                   const float* maskT, int nB_, int nW, int T_frames, int ws, int heads, int C,
                                   int bias_windows, const int* bias_index, void* stream) {
  if (bias_windows != 1 && ((!bias_index && bias_windows != nW) || maskT)) return -1204;
  AttnArgs a{qkv, ld, out, ldo, nullptr, 0, biasT, maskT, nullptr, nullptr, nullptr, nB_, nW, heads, C, 0, 1.0f, bias_windows, bias_index};
  return attn_common(dtype, a, T_frames, ws, false, stream);
}

/* floats of scratch stswin_win_attn_bwd needs: one [N*N + C/heads] slab per resident problem slot (an upper bound) */
extern "C" long stswin_win_attn_bwd_scratch(int nB_, int ws, int heads, int C) {
  if (heads <= 0 || C % heads || nB_ <= 0) return -1203;
  long slots = (long)nB_ * heads + 3;
  if (slots > 8192) slots = 8192;
  return slots * ((long)ws * ws * ws * ws + C / heads);
}

extern "C" int stswin_win_attn_bwd(int dtype, const void* qkv, long ld, const void* dout, long lddo, void* dqkv, long lddq,
                                   const float* biasT, const float* maskT, float* dbiasT, float* dqkv_colsum, int nB_,
                                   int nW, int T_frames, int ws, int heads, int C, float scale, int bias_windows,
                                   const int* bias_index, float* scratch, long scratch_floats, void* stream) {
  const int dbg = bias_windows & (1 << 30);    // DBG (tools/attn_timeline.py): dqkv_colsum is a u64 [workgroups][16] timestamp buffer
  bias_windows &= ~(1 << 30);
  if (bias_windows != 1 && ((!bias_index && bias_windows != nW) || maskT)) return -1204;
  if (!dbg && (dbiasT || dqkv_colsum) && !scratch) return -1205;
  AttnArgs a{qkv, ld, dqkv, lddq, dout, lddo, biasT, maskT, dbiasT, dqkv_colsum, dbg ? nullptr : scratch, nB_, nW, heads, C, 0, scale,
             bias_windows | dbg, bias_index};
  a_scratch_floats = scratch_floats;
  return attn_common(dtype, a, T_frames, ws, true, stream);
}

extern "C" int stswin_win_attn_qkv_fwd(const void* x, long ldx, long x_rows, const int* rmap, const void* w, long ldw, const float* bqkv,
                                       void* qkv_out, long ldq, void* out, long ldo, const float* biasT, int nB_, int nW, int T_frames, int ws,
                                       int heads, int C, float scale, int bias_windows, const int* bias_index, void* stream) {
  // the gathered token rows are addressed with 32-bit byte offsets from x: refuse (the caller takes the GEMM + attention pair) an x
  // whose last row a buffer offset cannot reach, instead of faulting on the device
  if (x_rows <= 0) x_rows = (long)nB_ * T_frames * ws * ws;
  if (ldx <= 0 || (unsigned long long)x_rows * (unsigned long long)ldx * 2ull > 0xFFFF0000ull) return -1208;
  unsigned long long* dbg_ts = nullptr;               // DBG: bit 30 of bias_windows = qkv_out is a u64 [256][8][8] stamp buffer instead
  if (bias_windows & (1 << 30)) { bias_windows &= ~(1 << 30); dbg_ts = (unsigned long long*)qkv_out; qkv_out = nullptr; }
  if (T_frames * ws * ws != 128 || heads <= 0 || C != heads * 128 || C % 32) return -1206;     // the stage-1 shape only
  if (nW <= 0 || nB_ % nW || (bias_windows != 1 && !bias_index && bias_windows != nW)) return -1204;
  if (ldx % 8 || ldw % 8 || ldo % 8 || (qkv_out && ldq % 8)) return -1207;
  AttnQkvArgs q;
  q.X = (const bf16*)x; q.ldx = ldx; q.rmap = rmap; q.W = (const bf16*)w; q.ldw = ldw; q.bqkv = bqkv;
  q.qkv_out = (bf16*)qkv_out; q.ldq = ldq;
  q.a = AttnArgs{nullptr, 0, out, ldo, nullptr, 0, biasT, nullptr, nullptr, (float*)dbg_ts, nullptr, nB_, nW, heads, C, ws * ws, scale,
                 bias_windows, bias_index};
  constexpr int LDS = 4 * (128 * 64 + 384 * 64);     // 4-stage ring of token + weight chunks (128 KB); the Q | K | V tiles reuse it
  hipStream_t st = (hipStream_t)stream;
  const long probs = (long)nB_ * heads;
  const int grid = (int)(probs < 256 ? probs : 256);
#define QKV_LAUNCH(NCV, CV)                                                                                                            \
  do {                                                                                                                                 \
    static const int attr = (int)hipFuncSetAttribute((const void*)attn_qkv_fwd_kernel<NCV, CV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
    if (attr != 0) return -attr;                                                                                                       \
    hipLaunchKernelGGL((attn_qkv_fwd_kernel<NCV, CV>), dim3(grid), dim3(512), LDS, st, q);                                             \
  } while (0)
  if (ws * ws == 64 && C == 512) QKV_LAUNCH(64, 512);
  else if (C == 512) QKV_LAUNCH(0, 512);
  else if (C == 256) QKV_LAUNCH(0, 256);
  else if (C == 1024) QKV_LAUNCH(0, 1024);
  else return -1206;
#undef QKV_LAUNCH
  STSWIN_CHECK_LAUNCH();
  return 0;
}

template <int NTOK, int HD, int NC>
static int launch_attn_fp8(const AttnArgs& a, hipStream_t st) {
  using Cfg = AttnCfg<bf16, NTOK, HD>;
  const long probs = (long)a.nB_ * a.heads;
  const int grid = (int)((probs + Cfg::PPB - 1) / Cfg::PPB);
  const int lds = Cfg::FWD_LDS + 64;
  static const int attr = (int)hipFuncSetAttribute((const void*)attn_fwd_fp8_kernel<NTOK, HD, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != 0) return -attr;
  hipLaunchKernelGGL((attn_fwd_fp8_kernel<NTOK, HD, NC>), dim3(grid), dim3(256), lds, st, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_win_attn_fwd_fp8(const void* qkv, long ld, void* out, long ldo, const float* biasT, const float* maskT, int nB_,
                                       int nW, int T_frames, int ws, int heads, int C, int bias_windows, const int* bias_index,
                                       void* stream) {
  if (bias_windows != 1 && ((!bias_index && bias_windows != nW) || maskT)) return -1204;
  AttnArgs a{qkv, ld, out, ldo, nullptr, 0, biasT, maskT, nullptr, nullptr, nullptr, nB_, nW, heads, C, ws * ws, 1.0f, bias_windows, bias_index};
  const int ntok = T_frames * ws * ws;
  if (C % heads || nW <= 0 || nB_ % nW) return -1203;
  const int hd = C / heads;
  hipStream_t st = (hipStream_t)stream;
  if (ntok == 128 && hd == 128 && a.N == 64) return launch_attn_fp8<128, 128, 64>(a, st);
  if (ntok == 32 && hd == 256 && a.N == 16) return launch_attn_fp8<32, 256, 16>(a, st);
  if (ntok == 128 && hd == 128) return launch_attn_fp8<128, 128, 0>(a, st);
  if (ntok == 32 && hd == 256) return launch_attn_fp8<32, 256, 0>(a, st);
  if (ntok == 128 && hd == 32) return launch_attn_fp8<128, 32, 0>(a, st);
  if (ntok == 32 && hd == 64) return launch_attn_fp8<32, 64, 0>(a, st);
  return -1201;
}

// ------------------------------------------------------------------------------------------------ fp8-storage attention (configs[4])
template <int NTOK, int HD, int NC>
static int launch_attn_f8_fwd(const AttnArgs& a, hipStream_t st) {
  using Cfg = AttnCfg<bf16, NTOK, HD>;
  const long probs = (long)a.nB_ * a.heads;
  const int grid = (int)((probs + Cfg::PPB - 1) / Cfg::PPB);
  constexpr int lds = Cfg::PPB * 2 * NTOK * HD;
  static const int attr = (int)hipFuncSetAttribute((const void*)attn_fwd_f8_kernel<NTOK, HD, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != 0) return -attr;
  hipLaunchKernelGGL((attn_fwd_f8_kernel<NTOK, HD, NC>), dim3(grid), dim3(256), lds, st, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_win_attn_fwd_f8(const void* qkv8, long ld8, const float* scales, long ld_scales, void* out, long ldo, const float* biasT,
                                      const float* maskT, int nB_, int nW, int T_frames, int ws, int heads, int C, int bias_windows,
                                      const int* bias_index, void* stream) {
  AttnArgs a{qkv8, ld8, out, ldo, nullptr, 0, biasT, maskT, nullptr, nullptr, nullptr, nB_, nW, heads, C, ws * ws, 1.0f, bias_windows, bias_index,
             scales, ld_scales};
  const int ntok = T_frames * ws * ws;
  if (C % heads || nW <= 0 || nB_ % nW || !scales || ld_scales < 3 * heads) return -1202;
  const int hd = C / heads;
  hipStream_t st = (hipStream_t)stream;
  if (ntok == 128 && hd == 128) return a.N == 64 ? launch_attn_f8_fwd<128, 128, 64>(a, st) : launch_attn_f8_fwd<128, 128, 0>(a, st);
  if (ntok == 32 && hd == 256) return a.N == 16 ? launch_attn_f8_fwd<32, 256, 16>(a, st) : launch_attn_f8_fwd<32, 256, 0>(a, st);
  return -1201;                                            // (other geometries keep the bf16 kernels)
}

// backward on fp8-stored q | k | v: stage 1 = attn_bwd8_kernel<64, true, true> (8x8 windows over a frame pair, head dim 128), stage 2 =
// attn_bwd_kernel<bf16, 32, 256, NC, true>.  dqkv is bf16 (the gradient of the DEQUANTISED q | k | v: straight-through).
extern "C" int stswin_win_attn_bwd_f8(const void* qkv8, long ld8, const float* scales, long ld_scales, const void* dout, long lddo, void* dqkv,
                                      long lddq, const float* biasT, const float* maskT, float* dbiasT, float* dqkv_colsum, int nB_, int nW,
                                      int T_frames, int ws, int heads, int C, float scale, int bias_windows, const int* bias_index,
                                      float* scratch, long scratch_floats, void* stream) {
  if (C % heads || nW <= 0 || nB_ % nW || !scales || ld_scales < 3 * heads) return -1202;
  const int ntok = T_frames * ws * ws, hd = C / heads, N = ws * ws;
  AttnArgs a{qkv8, ld8, dqkv, lddq, dout, lddo, biasT, maskT, dbiasT, dqkv_colsum, scratch, nB_, nW, heads, C, N, scale, bias_windows, bias_index,
             scales, ld_scales};
  a_scratch_floats = scratch_floats;
  hipStream_t st = (hipStream_t)stream;
  const long probs = (long)nB_ * heads;
  if (ntok == 128 && hd == 128 && N == 64) {
    using Cfg = AttnCfg<bf16, 128, 128>;
    static const int attr8 = (int)hipFuncSetAttribute((const void*)attn_bwd8_kernel<64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::BWD_LDS);
    if (attr8 != 0) return -attr8;
    if ((long)nB_ * 128 * lddq * 2 >= 0x7FFF0000L || probs % heads) return -1201;
    int g = stswin_cu_budget();
    if (g > probs) g = (int)probs;
    g = (g / heads) * heads;
    if (g < heads) return -1201;
    if (a.slabs && (long)g * (N * N + 128) > a_scratch_floats) return -1205;
    hipLaunchKernelGGL((attn_bwd8_kernel<64, true, true>), dim3(g), dim3(512), Cfg::BWD_LDS, st, a);
    const int rf = attn_bwd_fold(a, g, 128, st);
    if (rf) return rf;
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  if (ntok == 32 && hd == 256 && N == 16) {
    using Cfg = AttnCfg<bf16, 32, 256>;
    static const int attr = (int)hipFuncSetAttribute((const void*)attn_bwd_kernel<bf16, 32, 256, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::BWD_LDS);
    if (attr != 0) return -attr;
    const int grid = (int)((probs + Cfg::PPB - 1) / Cfg::PPB);
    const int per_cu = (160 * 1024) / Cfg::BWD_LDS > 0 ? (160 * 1024) / Cfg::BWD_LDS : 1;
    int g = stswin_cu_budget() * per_cu, step = heads;
    for (int d = Cfg::PPB; d > 1; --d)
      if (Cfg::PPB % d == 0 && heads % d == 0) { step = heads / d; break; }
    if (g > grid) g = grid;
    g = (g / step) * step;
    if (g < step) g = step;
    if (a.slabs && (long)g * Cfg::PPB * (N * N + 256) > a_scratch_floats) return -1205;
    hipLaunchKernelGGL((attn_bwd_kernel<bf16, 32, 256, 16, true>), dim3(g), dim3(256), Cfg::BWD_LDS, st, a);
    const int rf = attn_bwd_fold(a, g * Cfg::PPB, 256, st);
    if (rf) return rf;
    STSWIN_CHECK_LAUNCH();
    return 0;
  }
  return -1201;
}
