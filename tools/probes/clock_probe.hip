// Shader clock under load: s_memtime (core clock ticks) against s_memrealtime (100 MHz) around an MFMA-only, a VALU-only
// and a mixed loop, with 1..8 waves per CU on all CUs.  Tells whether sustained MFMA work runs at a lower clock (power
// management) than the short kernels of a micro-benchmark.   hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(int mode, int iters, unsigned long long* out, float* sink) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.f + i * 0.01f); }
  float v = threadIdx.x * 1e-3f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (mode != 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (mode != 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = v;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[0] = s;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 4096 * 16); hipMalloc(&sink, 4);
  unsigned long long h[2 * 4096];
  const char* names[3] = {"mfma", "valu", "mfma+valu"};
  for (int mode = 0; mode < 3; ++mode)
    for (int waves = 1; waves <= 8; waves *= 2)
      for (int ms = 0; ms < 2; ++ms) {
        const int iters = ms ? 400000 : 20000;          // ~ms-scale and ~50 ms-scale runs
        const int blocks = 256;
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * waves), 0, 0, mode, iters, out, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost);
        double mhz = 0, us = 0;
        for (int i = 0; i < blocks; ++i) { mhz += (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0); us += h[2 * i + 1] / 100.0; }
        printf("%-10s waves/CU %d  iters %6d: %7.0f MHz  (%.1f ms)\n", names[mode], waves, iters, mhz / blocks, us / blocks / 1e3);
      }
  return 0;
}
