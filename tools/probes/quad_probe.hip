// Stand-alone probe: per-stage time of a ONE-wave-per-SIMD GEMM main loop (4 waves of 128x128 on a 256x256x32 stage: 16 fragment
// reads + 8 LDS-DMA copies + 64 MFMA 16x16x32 per wave and stage) in several schedules.  Timing only - operands are whatever
// is in LDS.  Build: hipcc --offload-arch=gfx950 -O3 quad_probe.hip -o quad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS3 __attribute__((address_space(3)))
__device__ inline void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (LDS3 void*)l, 16, 0, 0);
}
template <int N> __device__ inline void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
// MODE bits: 1 = LDS-DMA copies, 2 = fragment reads, 4 = barrier, 8 = interleave reads with the MFMAs (sched_group_barrier),
//            16 = MFMA 32x32x16 instead of 16x16x32
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const char* src, float* out, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool DMA = MODE & 1, RD = MODE & 2, BAR = MODE & 4, ILV = MODE & 8, M32 = MODE & 16;
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)(blockIdx.x & 15) * 32768 + tid * 16;
  auto issue = [&](int q) {
    char* st = smem + (q & 3) * 32768 + w * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(base + ((size_t)(q & 7) * 524288 + i * 4096), st + i * 1024);
  };
  f32x4 acc[64];
  f32x16 acc32[16];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
  bf16x8 fr[2][16];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) fr[s][i][e] = (__bf16)(float)(l + i);
  if (DMA) { for (int q = 0; q < 3; ++q) issue(q); wait_vmcnt<16>(); }
  __builtin_amdgcn_s_barrier();
  auto rd = [&](int q, int set) {
    const char* st = smem + (q & 3) * 32768 + (l & 15) * 64 + ((l >> 4) << 4) + (w >> 1) * 8192;
    const char* sb = smem + (q & 3) * 32768 + 16384 + (l & 15) * 64 + ((l >> 4) << 4) + (w & 1) * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[set][i] = *(const bf16x8*)(st + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[set][8 + i] = *(const bf16x8*)(sb + i * 1024);
  };
  auto mma = [&](int set) {
    if constexpr (!M32) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[set][8 + j], fr[set][i], acc[i * 8 + j], 0, 0, 0);
    } else {                                  // 128x128x32 as 4x4 tiles of 32x32 x 2 K halves: 32 MFMA 32x32x16
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc32[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[set][8 + j * 2 + kk], fr[set][i * 2 + kk], acc32[i * 4 + j], 0, 0, 0);
    }
  };
  auto step = [&](int kt, int set) {
    if (DMA) wait_vmcnt<8>();
    if (BAR) __builtin_amdgcn_s_barrier();
    if (DMA && kt + 3 < nt) issue(kt + 3);
    if (RD) {
      __builtin_amdgcn_s_waitcnt(0xC07F);
      rd(kt + 1, set ^ 1);
    }
    if (!ILV) __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    mma(set);
    __builtin_amdgcn_s_setprio(0);
    if (ILV) {                                // 16 x {1 DS read, 4 MFMA} (or 2 MFMA 32x32)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
        __builtin_amdgcn_sched_group_barrier(0x008, M32 ? 2 : 4, 0);   // MFMA
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  long t0 = wall_clock64();
  int kt = 0;
  for (; kt + 1 < nt; kt += 2) { step(kt, 0); step(kt + 1, 1); }
  long t1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc32[i][e];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += (float)fr[0][i][0] + (float)fr[1][i][0];
  if (s == 123.456f) out[tid] = s;
  if (tid == 0) out[1024 + blockIdx.x] = (float)(t1 - t0);
}

template <int MODE> void run(const char* name, const char* src, float* out, int nt) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 131072, 0, src, out, nt);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  float h[256]; hipMemcpy(h, out + 1024, sizeof(h), hipMemcpyDeviceToHost);
  double ticks = 0; for (int i = 0; i < 256; ++i) ticks += h[i];
  const double flop = 256.0 * nt * 256 * 256 * 32 * 2;
  printf("%-44s %7.3f us/stage (event)  %7.3f us/stage (in-kernel, 100 MHz clock)  %7.0f TFLOP/s\n", name, ms * 1e3 / nt, ticks / 256 / 100.0 / nt,
         flop / (ms * 1e-3) / 1e12);
}

int main() {
  char* src; float* out;
  hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
  hipMalloc(&out, 1 << 16);
  const int nt = 4096;
  run<0>("mfma 16x16x32 only", src, out, nt);
  run<16>("mfma 32x32x16 only", src, out, nt);
  run<4>("mfma16 + barrier", src, out, nt);
  run<2 | 4>("mfma16 + reads (front) + barrier", src, out, nt);
  run<2 | 4 | 8>("mfma16 + reads (interleaved) + barrier", src, out, nt);
  run<1 | 2 | 4>("mfma16 + reads (front) + DMA + barrier", src, out, nt);
  run<1 | 2 | 4 | 8>("mfma16 + reads (interleaved) + DMA + barrier", src, out, nt);
  run<16 | 1 | 2 | 4>("mfma32 + reads (front) + DMA + barrier", src, out, nt);
  run<16 | 1 | 2 | 4 | 8>("mfma32 + reads (interleaved) + DMA + barrier", src, out, nt);
  return 0;
}
