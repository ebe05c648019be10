// Stand-alone probe: what do LDS-DMA pieces / global loads / LDS fragment reads cost ONE wave per SIMD that is issuing MFMAs back to back?
// 256 threads per workgroup (one wave per SIMD), one workgroup per CU (128 KB of LDS), 256 workgroups.  Per iteration ("stage"): 1024 cycles
// of MFMA (64 x 16x16x32 or 32 x 32x32x16), plus - by variant - 8 copy pieces and / or 16 ds_read_b128, spread evenly between the MFMAs.
// Prints shader-clock cycles per iteration (s_memtime), mean over the workgroups.  Timing only: operands are whatever is in LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o w4_issue_probe w4_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define LDS3 __attribute__((address_space(3)))
// COPY: 0 none, 1 LDS-DMA (builtin: s_mov m0 in front of each piece), 2 LDS-DMA raw ISA with ONE voffset and m0 stepped by s_add AFTER the piece,
//       3 buffer_load_dwordx4 to registers + ds_write_b128 of the previous iteration's registers
// PAT: source pattern of a 1 KB piece: 0 contiguous, 1 = 16 rows x 64 B (a 32-deep K slice of bf16 rows, pitch 8 KB), 2 = 8 rows x 128 B (64-deep)
template <int MF, int COPY, int READS, int PAT = 0>
__global__ __launch_bounds__(256, 1) void probe(const char* src, unsigned long long* out, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 15) * 2097152), (short)0, (int)0x7FFFFFFE, 0x00020000);
  f32x4 acc4[16];
  f32x16 acc16[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc4[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  bf16x8 fa[2][8], fb[2][8];      // two fragment sets: the reads of an iteration fill the set the NEXT iteration multiplies
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa[0][i] = fa[1][i] = *(const bf16x8*)(smem + i * 1024 + l * 16); fb[0][i] = fb[1][i] = *(const bf16x8*)(smem + 16384 + i * 1024 + l * 16); }
  v4i stg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) stg[i] = (v4i){0, 0, 0, 0};
  // lane offset inside a piece; pieces of a wave: rows w*32 + pc*(16 or 8) ...
  const unsigned voff = PAT == 0 ? (unsigned)(tid * 16) : PAT == 1 ? (unsigned)((w * 64 + (l >> 2)) * 8192 + (l & 3) * 16) : (unsigned)((w * 64 + (l >> 3)) * 8192 + (l & 7) * 16);
  constexpr unsigned PSTEP = PAT == 0 ? 4096u : PAT == 1 ? 16u * 8192u : 8u * 8192u;   // next piece of this wave
  __syncthreads();
  const unsigned long long t0 = clock64();
  auto body = [&](auto setc, int it) __attribute__((always_inline)) {
    constexpr int SET = decltype(setc)::value;
    char* slot = smem + 32768 + (it & 1) * 32768 + w * 8192;      // this wave's 8 pieces of the stage being filled
    const int soff = PAT == 0 ? (it & 7) * 32768 : PAT == 1 ? (it & 127) * 64 : (it & 63) * 128;
    if (COPY == 1 || COPY == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    unsigned m0v = (unsigned)(unsigned long)(LDS3 char*)slot;
    if (COPY == 2) asm volatile("s_mov_b32 m0, %0" ::"s"(m0v) : "m0");
    constexpr int NM = MF == 16 ? 64 : 32;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      constexpr int PER_PIECE = NM / 8, PER_READ = NM / 16;
      if (READS && m % PER_READ == 0) {
        const int r = m / PER_READ;
        if (r < 8) fb[SET ^ 1][r] = *(const bf16x8*)(smem + 16384 + r * 1024 + ((l * 16 + it * 64) & 1023));
        else fa[SET ^ 1][r - 8] = *(const bf16x8*)(smem + (r - 8) * 1024 + ((l * 16 + it * 64) & 1023));
      }
      if (COPY && m % PER_PIECE == PER_PIECE / 2) {
        const int pc = m / PER_PIECE;
        if (COPY == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS3 void*)(slot + pc * 1024), 16, voff + pc * PSTEP, soff, 0, 0);
        if (COPY == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_add_u32 m0, m0, 0x400" ::"v"(voff), "s"(rs), "s"(soff + pc * PSTEP) : "memory", "m0");
        if (COPY == 3) {
          *(v4i*)(slot + pc * 1024 + l * 16) = stg[pc];
          stg[pc] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + pc * PSTEP), soff, 0);
        }
      }
      if (MF == 16) acc4[m & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[SET][m & 7], fa[SET][(m >> 3) & 7], acc4[m & 15], 0, 0, 0);
      else acc16[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[SET][m & 7], fa[SET][(m >> 3) & 7], acc16[m & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int it = 0; it < nt; it += 2) {
    body(std::integral_constant<int, 0>{}, it);
    body(std::integral_constant<int, 1>{}, it + 1);
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc4[i][0] + acc4[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc16[i][0] + acc16[i][15];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)stg[i][0];
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (s == 123.456f) out[blockIdx.x] = 0;
}

template <int MF, int COPY, int READS, int PAT = 0> void run(const char* name, const char* src, unsigned long long* out) {
  const int nt = 400;
  hipFuncSetAttribute((const void*)probe<MF, COPY, READS, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<MF, COPY, READS, PAT>), dim3(256), dim3(256), 131072, 0, src, out, nt);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-64s %8.1f cycles per stage (MFMA alone: 1024)\n", name, s / 256 / nt);
}

int main() {
  char* src; unsigned long long* out;
  hipMalloc(&src, 48u << 20); hipMemset(src, 0x11, 48u << 20);
  hipMalloc(&out, 256 * 8);
  run<16, 0, 0>("16x16x32  MFMAs only", src, out);
  run<16, 0, 1>("16x16x32  + 16 ds_read_b128 (into the other fragment set)", src, out);
  run<16, 1, 0, 0>("16x16x32  + 8 LDS-DMA pieces, contiguous KB", src, out);
  run<16, 1, 0, 1>("16x16x32  + 8 LDS-DMA pieces, 16 rows x 64 B", src, out);
  run<16, 1, 0, 2>("16x16x32  + 8 LDS-DMA pieces, 8 rows x 128 B", src, out);
  run<16, 2, 0, 1>("16x16x32  + 8 LDS-DMA pieces, 16 x 64 B, raw ISA (s_add m0 behind)", src, out);
  run<16, 3, 0, 1>("16x16x32  + 8 global_load_dwordx4 (16 x 64 B) + 8 ds_write_b128", src, out);
  run<16, 3, 0, 2>("16x16x32  + 8 global_load_dwordx4 (8 x 128 B) + 8 ds_write_b128", src, out);
  run<16, 1, 1, 1>("16x16x32  + 8 pieces (16 x 64 B) + 16 reads", src, out);
  run<16, 1, 1, 2>("16x16x32  + 8 pieces (8 x 128 B) + 16 reads", src, out);
  run<16, 3, 1, 2>("16x16x32  + 8 loads (8 x 128 B) + 8 writes + 16 reads", src, out);
  run<32, 0, 0>("32x32x16  MFMAs only", src, out);
  run<32, 0, 1>("32x32x16  + 16 ds_read_b128", src, out);
  run<32, 1, 0, 1>("32x32x16  + 8 LDS-DMA pieces, 16 rows x 64 B", src, out);
  run<32, 1, 0, 2>("32x32x16  + 8 LDS-DMA pieces, 8 rows x 128 B", src, out);
  run<32, 1, 1, 1>("32x32x16  + 8 pieces (16 x 64 B) + 16 reads", src, out);
  run<32, 1, 1, 2>("32x32x16  + 8 pieces (8 x 128 B) + 16 reads", src, out);
  return 0;
}
