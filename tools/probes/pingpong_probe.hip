// Stand-alone probe: per-stage time of the ping-pong ring loop (2 wave rows, 4-stage LDS ring of 32 KB stages, 12 fragment
// reads + 4 LDS-DMA copies per wave and stage) with the two bf16 MFMA shapes and a variable number of filler VALU
// instructions in the read phase.  Timing only - operands are whatever is in LDS.  Build: hipcc --offload-arch=gfx950 -O3
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
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}
template <int SHAPE, int FILL, int STREAM = 0>
__global__ __launch_bounds__(512, 2) void probe(const char* src, float* out, int nt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool lag = (w >> 2) == 1;
  const char* base = src + (size_t)(blockIdx.x & 15) * 32768 + tid * 16;   // 16 x 8 stage images: L2-resident, like reused GEMM tiles
  auto issue = [&](int q) {
    char* st = smem + (q & 3) * 32768 + w * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // STREAM: the two "A" copies of a stage come from a region streamed once per pair of workgroups (HBM, like the A
      // operand of an N = 512 GEMM); the two "B" copies and everything in the non-streaming variants stay L2-resident
      if (STREAM && i < 2) glds16(src + 16777216 + ((size_t)(blockIdx.x >> 1) * 256 + (q & 255)) * 16384 + i * 8192 + tid * 16, st + i * 1024);
      else glds16(base + ((size_t)(q & 7) * 524288 + i * 8192), st + i * 1024);
    }
  };
  f32x4 acc4[32];
  f32x16 acc16[8];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc4[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  bf16x8 fr[12];
  float fill = (float)l;
  for (int q = 0; q < 3; ++q) issue(q);
  wait_vmcnt<8>();
  if (lag) __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nt; ++kt) {
    __builtin_amdgcn_s_barrier();
    if (kt + 3 < nt) issue(kt + 3);
    const char* st = smem + (kt & 3) * 32768 + (l & 15) * 64 + ((l >> 4) << 4);
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[i] = *(const bf16x8*)(st + i * 2048);
#pragma unroll
    for (int i = 0; i < FILL; ++i) fill = fill * 1.0001f + 0.5f;
    if (lag) wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    if constexpr (SHAPE == 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[8 + j], fr[i], acc4[i * 4 + j], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc16[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[8 + j * 2 + ks], fr[i * 2 + ks], acc16[i * 2 + j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!lag) wait_vmcnt<8>();
  }
  if (!lag) __builtin_amdgcn_s_barrier();
  float s = fill;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc4[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc16[i][0];
  if (s == 123.456f) out[tid] = s;
}
template <int SHAPE, int FILL, int STREAM = 0> static void run(const char* src, float* out, const char* name) {
  hipFuncSetAttribute((const void*)probe<SHAPE, FILL, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int nt = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<SHAPE, FILL, STREAM>), dim3(256), dim3(512), 131072, 0, src, out, nt);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<SHAPE, FILL, STREAM>), dim3(256), dim3(512), 131072, 0, src, out, nt);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us_stage = ms * 1e3 / 10 / nt;
  printf("%-34s %.3f us per stage  (%.0f TFLOP/s-equivalent at 4.19 MFLOP per CU-stage)\n", name, us_stage, 256 * 4.194304 / us_stage);
}
int main() {
  char* src; float* out;
  hipMalloc(&src, (size_t)16777216 + (size_t)128 * 256 * 16384 + 65536); hipMalloc(&out, 4096);
  hipMemset(src, 0, (size_t)16777216 + (size_t)128 * 256 * 16384 + 65536);
  run<16, 0>(src, out, "16x16x32, no filler VALU");
  run<16, 16>(src, out, "16x16x32, 16 filler VALU");
  run<16, 48>(src, out, "16x16x32, 48 filler VALU");
  run<16, 0, 1>(src, out, "16x16x32, A streamed from HBM");
  run<16, 16, 1>(src, out, "16x16x32, 16 VALU, A streamed");
  run<32, 0>(src, out, "32x32x16, no filler VALU");
  run<32, 16>(src, out, "32x32x16, 16 filler VALU");
  run<32, 48>(src, out, "32x32x16, 48 filler VALU");
  return 0;
}
