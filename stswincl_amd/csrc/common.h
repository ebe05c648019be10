// Device-side helpers shared by every kernel of libstswin_hip (gfx950 / CDNA4 only).
//
// Conventions
//   * wave = 64 lanes; workgroups are 256 threads (4 waves, one per SIMD) unless stated.
//   * storage type T is __bf16 (fast path) or float (exact-fp32 parity path); accumulation is fp32.
//   * MFMA lane maps (cdna_hip_programming.md section 3):
//       16x16x32 bf16 : A[row l&15][k 8(l>>4)+j]  B[k 8(l>>4)+j][col l&15]  C[row 4(l>>4)+r][col l&15]
//       32x32x16 bf16 : A[row l&31][k 8(l>>5)+j]  B[k 8(l>>5)+j][col l&31]  C[row (r&3)+8(r>>2)+4(l>>5)][col l&31]
//       16x16x4  f32  : A[row l&15][k l>>4]       B[k l>>4][col l&15]       C as 16x16
//       32x32x2  f32  : A[row l&31][k l>>5]       B[k l>>5][col l&31]       C as 32x32
//   * LDS images are written by LDS-DMA (global_load_lds_dwordx4: destination = wave base + lane*16,
//     so the XOR swizzle is applied to the per-lane SOURCE address and again on the read).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define DEVI __device__ __forceinline__

// rowops.hip: deterministic sum of per-workgroup partial vectors ("slabs") behind a launch boundary (see slab_fold_kernel)
int stswin_fold_launch(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, int seg_len, int nseg, float* o0, float* o1,
                       float* o2, long out_batch_stride, int batch, int accumulate, hipStream_t st);
int stswin_fold3_launch(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, const int* len /* [3] */, float* const* out /* [3] */,
                        const long* obs /* [3] batch strides of the outputs */, int batch, int accumulate, hipStream_t st);

// gemm.hip: compute units the persistent / one-workgroup-per-CU launches plan for (stswin_set_cu_budget; 256 by default)
int stswin_cu_budget();

// 256 B of zeros in device memory: the source of every padded / out-of-range 16-byte chunk.
static __device__ uint4 g_stswin_zero[16];  // per-TU copy (no -fgpu-rdc); zero-initialised

// Zero fill of a 4-byte-aligned region as a KERNEL on the launch stream.  The library does not use hipMemsetAsync / hipMemset2DAsync on its
// launch streams (round 6): inside a hipGraph (ROCm 7.0 / 7.2) a captured memset node did not reproduce the eager call - a 2-D fill left
// parts of a weight-gradient buffer on garbage, and the OHEM histogram block was not cleared between replays - so that a graph-replayed
// training run went to NaN where the eager run did not (tools/probes/tn_graph_repro.py, tools/probes/graph_vs_eager.py).
static __global__ __launch_bounds__(256) void stswin_zero_words_kernel(unsigned* __restrict__ p, long words) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long)gridDim.x * 256) p[i] = 0u;
}
static inline void stswin_zero_bytes(void* p, size_t bytes, hipStream_t stream) {
  const long words = (long)((bytes + 3) / 4);
  if (words <= 0) return;
  long blocks = (words + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stswin_zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned*)p, words);
}

DEVI int lane_id() { return threadIdx.x & 63; }
DEVI int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// Same copy issued as raw ISA, invisible to hipcc's waitcnt pass.  Needed wherever the tile is read back with
// ds_read_b64_tr_b16: behind the builtin the compiler puts s_waitcnt vmcnt(0) in front of every transposed read
// (it does not for plain ds_read), which drains the copies that were meant to stay in flight.  The caller owns all
// vmcnt waits for these copies; M0 carries the LDS base exactly as the builtin would set it.
DEVI void glds16_raw(const void* gsrc, void* lds_wave_base) {
  const unsigned lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds) : "memory", "m0");
}

// An empty volatile asm inside a wave-uniform `if` body keeps hipcc from if-converting it into always-executed VALU
// selects (the epilogue's rarely-set flag branches cost 12 VALU per fragment that way, taken or not).
#define NO_IFCVT asm volatile("" ::: "memory")

// ---- scalar (SMEM) load of a wave-uniform int: counted by lgkmcnt, so it never stalls the LDS-DMA vmcnt pipeline ----
DEVI int sload(const int* p, long idx) { return ((const __attribute__((address_space(4))) int*)p)[idx]; }

// ---- LDS-DMA: 16 B per lane, LDS destination = wave-uniform base + lane*16 -----------------
DEVI void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
DEVI void glds16_nt(const void* gsrc, void* lds_wave_base) {        // same copy, non-temporal (read-once data)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}
// ---- buffer-addressed LDS-DMA: descriptor base (wave-uniform) + per-lane 32-bit byte offset + scalar byte offset.  The
// descriptor claims 4 GB - 2 bytes, so an offset of 0xFFFFFFFF is out of range and the copy delivers zeros (edge rows,
// convolution padding).  Kept inside __device__ helpers: the resource type does not exist in the host pass.
DEVI void glds16_buf(const void* base, unsigned lane_off, int scalar_off, void* lds_wave_base) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)0xFFFFFFFE, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, lane_off, scalar_off, 0, 0);
}
// The same copy as raw ISA (invisible to hipcc's waitcnt pass: needed where ds_read_b64_tr_b16 follows, see glds16_raw).
typedef int buf_rsrc_t __attribute__((ext_vector_type(4)));
DEVI buf_rsrc_t make_buf_rsrc(const void* base) {      // raw buffer, stride 0, offsets < 0xFFFF0000 in range, above: zeros
  const unsigned long a = (unsigned long)base;
  buf_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = (int)0xFFFF0000;
  r[3] = 0x00020000;
  return r;
}
DEVI void glds16_buf_raw(const buf_rsrc_t& rs, unsigned lane_off, unsigned scalar_off, void* lds_wave_base) {
  const unsigned lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(lane_off), "s"(rs), "s"(scalar_off), "s"(lds)
               : "memory", "m0");
}
DEVI void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- swizzles ---------------------------------------------------------------------------------
// 128-byte rows (8 chunks of 16 B): chunk ^= row & 7   -> ds_read_b128 fragment reads conflict-free.
DEVI int swz128(int row) { return row & 7; }
// 256-byte (or longer) rows: chunk ^= ((row&3)<<2)|((row>>2)&3) on the low 4 chunk bits -> both the
// row-wise ds_read_b128 and the transposed ds_read_b64_tr_b16 reads are conflict-free (guide T10 (b)).
DEVI int swz256(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// ---- scalar conversions -----------------------------------------------------------------------
template <typename T> DEVI float to_f32(T v);
template <> DEVI float to_f32<float>(float v) { return v; }
template <> DEVI float to_f32<bf16>(bf16 v) { return (float)v; }
template <typename T> DEVI T from_f32(float v);
template <> DEVI float from_f32<float>(float v) { return v; }
template <> DEVI bf16 from_f32<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-safe

template <typename T> constexpr bool TT_is_bf16() { return sizeof(T) == 2; }
DEVI float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
DEVI float dgelu_erf(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// bf16-path GELU: Phi(x) = 0.5 + z*Q(z^2), z = clamp(x / (2.8*sqrt2), -1, 1), Q = degree-7 minimax fit of
// 0.5*erf(2.8 z)/z constrained to Q(1) = 0.5 (tails exactly 0 / 1).  |Phi error| <= 3.9e-5, |x*Phi error| <= 1.6e-4:
// below bf16 rounding of the activations it feeds.  No transcendental and only FMAs, so the f32x2 form compiles to
// v_pk_fma_f32 (two activations per instruction); erff() costs ~47 VALU instructions + a divergent branch and the
// Abramowitz-Stegun form used before still needed a quarter-rate rcp and exp per element -- at 256 activations per
// lane per 256x256 tile that was as long as the tile's whole main loop.  GELU' adds one v_exp_f32 for the density.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define STSWIN_PHI_COEFFS 1.579097463e+00f, -4.098180595e+00f, 9.285439338e+00f, -1.537328732e+01f, 1.769340906e+01f, -1.318582850e+01f, 5.639688737e+00f, -1.040338190e+00f
DEVI f32x2 phi_poly2(f32x2 x) {
  constexpr float C[8] = {STSWIN_PHI_COEFFS};
  f32x2 z = x * 0.2525381361380527f;
  z = __builtin_elementwise_min(__builtin_elementwise_max(z, (f32x2){-1.f, -1.f}), (f32x2){1.f, 1.f});
  const f32x2 s = z * z;
  f32x2 q = {C[7], C[7]};
#pragma unroll
  for (int i = 6; i >= 0; --i) q = __builtin_elementwise_fma(q, s, (f32x2){C[i], C[i]});
  return __builtin_elementwise_fma(z, q, (f32x2){0.5f, 0.5f});
}
DEVI f32x2 gelu_fast2(f32x2 x) { return x * phi_poly2(x); }
DEVI f32x2 dgelu_fast2(f32x2 x) {
  const f32x2 h = x * x * -0.72134752044448170368f;          // -x^2/2 * log2(e)
  const f32x2 dens = {__builtin_amdgcn_exp2f(h[0]), __builtin_amdgcn_exp2f(h[1])};
  return __builtin_elementwise_fma(x * 0.39894228040143267794f, dens, phi_poly2(x));
}
// gelu and gelu' of the same argument share the polynomial (the fc1 forward epilogue produces both)
DEVI void gelu_dgelu_fast2(f32x2 x, f32x2& g, f32x2& d) {
  const f32x2 phi = phi_poly2(x);
  const f32x2 h = x * x * -0.72134752044448170368f;
  const f32x2 dens = {__builtin_amdgcn_exp2f(h[0]), __builtin_amdgcn_exp2f(h[1])};
  g = x * phi;
  d = __builtin_elementwise_fma(x * 0.39894228040143267794f, dens, phi);
}
DEVI float gelu_fast(float x) { return gelu_fast2((f32x2){x, x})[0]; }
DEVI float dgelu_fast(float x) { return dgelu_fast2((f32x2){x, x})[0]; }
template <typename T> DEVI float gelu_t(float x) { return TT_is_bf16<T>() ? gelu_fast(x) : gelu_erf(x); }
template <typename T> DEVI float dgelu_t(float x) { return TT_is_bf16<T>() ? dgelu_fast(x) : dgelu_erf(x); }

// ---- per-type traits ----------------------------------------------------------------------------
template <typename T> struct TT;
template <> struct TT<bf16> {
  static constexpr int PACK = 8;    // elements per 16-byte chunk
  static constexpr bool IS_BF16 = true;
};
template <> struct TT<float> {
  static constexpr int PACK = 4;
  static constexpr bool IS_BF16 = false;
};

// 8 (bf16) or 4 (f32) consecutive elements <-> 16 bytes
template <typename T> struct Vec16;
template <> struct Vec16<bf16> {
  bf16x8 v;
  DEVI float get(int i) const { return (float)v[i]; }
  DEVI void set(int i, float f) { v[i] = (bf16)f; }
};
template <> struct Vec16<float> {
  f32x4 v;
  DEVI float get(int i) const { return v[i]; }
  DEVI void set(int i, float f) { v[i] = f; }
};

// ---- transposed fragment read from an LDS tile whose ROW index is the contraction index -------
// Returns, for the calling lane, the 4 elements tile[k0 + 0..3][col0 + (lane&15)] of a 16-lane group's
// 4x16 block (ds_read_b64_tr_b16; EXEC must be full).  `row_bytes` = LDS row pitch, swizzle = swz256.
DEVI bf16x4 lds_tr4(const char* tile, int row_bytes, int k0, int col0) {
  const int lam = threadIdx.x & 15;
  const int q = lam >> 2, p = lam & 3;
  const int row = k0 + q;
  const int col = col0 + 4 * p;              // element column; 8 elements per chunk
  const int chunk = (col >> 3) ^ swz256(row);
  const char* addr = tile + row * row_bytes + (chunk << 4) + ((col & 7) << 1);
  short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
  return __builtin_bit_cast(bf16x4, r);
}

DEVI bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// ---- XCD-aware, bijective block-id remap: blocks that share an XCD (bid % 8) get contiguous tiles -
DEVI int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// Sum over each aligned group of 16 lanes, result in every lane of the group: four DPP adds (xor 1, xor 2 inside the
// quads, then the half-row and row mirrors) instead of four ds_bpermute round trips through the LDS crossbar.
template <int CTRL> DEVI float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
DEVI float sum16(float v) {
  v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);       // row_half_mirror
  v += dpp_f32<0x140>(v);       // row_mirror
  return v;
}
DEVI float wave_sum(float v) {
  v = sum16(v);
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
DEVI float wave_max_dpp(float v) {                 // (16-lane rows by DPP - no LDS crossbar round trips - then two exchanges)
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
DEVI float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

#define STSWIN_CHECK_LAUNCH()                                  \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return -(int)e__;                   \
  } while (0)
