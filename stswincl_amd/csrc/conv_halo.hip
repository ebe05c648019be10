// 3x3 / stride 1 / pad 1 convolution of 64 -> 64 channels on NHWC token matrices as an implicit GEMM over an LDS-resident
// halo (torchvision resnet18.layer1 inside seg18/net/Ours/resnet.py:104-105: four such convolutions per frame batch, forward
// and input gradient; the 16-frame batch of a 4-clip step is 262144 pixels of 128x128 maps).
//
// The generic gather GEMM (gemm.hip) stages the A operand tap by tap: nine row-map-driven copies of the same 256 x 64 input
// tile per 256 x 64 output tile, 288 KB through L2 for 33 KB of distinct data - 300 MB per launch - and a wave tile of
// 64 x 64 reads as many LDS bytes per MFMA as the LDS can deliver at the full matrix rate.  Both bound the N = 64
// convolutions at 350-390 TFLOP/s (~50 us per launch).  Here
//   * a workgroup owns 256 consecutive pixels of one frame (256 / W image rows) and loads their HALO once: (rows + 2) x
//     (W + 2) pixel records of 128 bytes, zeros outside the image; every tap reads its pixel fragments straight from the
//     halo at a shifted address - no global traffic inside the multiply loop, input and output cross HBM once (plus the
//     two shared halo rows, which consecutive tiles of one workgroup find in its XCD's L2);
//   * the whole 64 x 576 weight matrix lives in REGISTERS (72 fragments of 4 VGPRs; one wave per SIMD owns all 512 of them):
//     the LDS serves pixel fragments only, 4 reads per 16 MFMAs;
//   * the halo of the next tile is copied (LDS-DMA) while the current one is multiplied: two halo buffers, one barrier per tile.
//
//   y[p][co] = sum_{tap, ci} x[p + off(tap)][ci] * w[co][tap * 64 + ci]      off(tap) = sign * (tap / 3 - 1, tap % 3 - 1)
// sign = +1: forward (w = the tap-major forward matrix of stswin_conv_pack); sign = -1: input gradient (w = its dgrad matrix,
// taps in forward order: the inverse row map negates the offsets, stswin_conv_rowmap).  Optional: + R (the gradient another
// consumer of the input produced: headops.GradLink), per-128-row-block column sums and sums of squares of the stored values
// (the BatchNorm statistics table of STSWIN_GF_CS_SQ: [2][2 * ceil(M / 256)][64]).
#include "common.h"

struct ConvHaloArgs {
  const bf16* X; const bf16* Wm; bf16* Y; const bf16* R; float* stats;
  int frames, H, sign; long M;
  unsigned long long* ts;          // tools/conv_halo_timeline.py: [workgroup][8 tiles][8 slots] wall-clock stamps of wave 0 (or NULL)
};

constexpr int CH_C = 64, CH_ROWB = 128;                 // channels, bytes per pixel record
constexpr int CH_EX = 2048;                             // statistics exchange [4 waves][2][64] floats
constexpr int ch_hp(int W) { return ((W + 2) + 7) / 8 * 8; }            // halo row pitch in pixels (whole 8-pixel copy groups)
constexpr int ch_halo_bytes(int W) { return (256 / W + 2) * ch_hp(W) * CH_ROWB; }

constexpr int CH_WPITCH = 1168;                         // weight rows staged in LDS: 1152 + 16 bytes (16 rows -> 16 distinct 16-byte bank groups)
constexpr int CH_WBYTES = 64 * CH_WPITCH;               // 74752 = 73 KB of copy instructions
constexpr int ch_lds_bytes(int W) { return CH_EX + ch_halo_bytes(W) + (ch_halo_bytes(W) > CH_WBYTES ? ch_halo_bytes(W) : CH_WBYTES); }

// The halo of a tile as copy groups of 8 pixel records (1 KB, one wave instruction); record chunk c of halo pixel hx lands at
// chunk c ^ (hx & 7) (the 128-byte-row swizzle: the fragment reads are conflict-free).  Requests are issued from inside the
// multiply loop, one per (tap, k half) step, so they must cost next to nothing: wave w copies halo rows w, w + 4, ..; its
// request number k is group k % GPR of its row number k / GPR (compile-time), the address is a per-row scalar base (set once per
// tile; rows outside the image point at a block of zeros) + k % GPR KB + one lane offset that is the same for every request;
// the pad columns (x = -1, x >= W) are zeroed once per buffer and their lanes masked out of the copy by EXEC.
DEVI unsigned long ch_uniform(unsigned long v) {      // a value the compiler must keep in scalar registers
  return (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
         ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}
static __device__ uint4 ch_zero[64 * 18];           // 18 KB of zeros: a whole halo row of copy groups

template <int LW>
struct ChHalo {
  static constexpr int W = 1 << LW, TY = 256 >> LW, HP = ch_hp(W), GPR = HP / 8, NR = (TY + 2 + 3) / 4, STEPS = NR * GPR;
  static constexpr int NPAD = HP - W;              // pad pixels per halo row: column 0 and columns W + 1 .. HP - 1
  static constexpr int FIRST = -256;               // group 0 of a row: pixel x = -1 (lanes 0-7) is padding
  static constexpr int LAST = 255;                 // last group: pixel x = W - 1 (lanes 0-7), then padding
  static_assert(W + 1 - (GPR - 1) * 8 == 1, "the last copy group of a halo row holds one image pixel");
  static_assert(STEPS <= 18, "one copy per (tap, k half) step must cover the halo");
  const char* base[NR]; unsigned lds[NR];
  DEVI void set(const ConvHaloArgs& a, long t, long tiles_per_frame, char* buf, int w, bool enable) {
    const int f = (int)(t / tiles_per_frame), y0 = (int)(t - (long)f * tiles_per_frame) * TY;
    const bf16* frame = a.X + (long)f * a.H * W * CH_C;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)buf);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      // (a row index past the halo repeats the last row: the same bytes to the same place; a disabled request - no next tile -
      //  copies zeros into the buffer nobody reads)
      const int hy = w + 4 * r < TY + 2 ? w + 4 * r : TY + 1, y = y0 - 1 + hy;
      const bool real = enable && (unsigned)y < (unsigned)a.H;
      base[r] = (const char*)ch_uniform((unsigned long)(real ? (const char*)(frame + ((long)y * W - 1) * CH_C) : (const char*)ch_zero));
      lds[r] = __builtin_amdgcn_readfirstlane(lds0 + hy * GPR * 1024);
    }
  }
  static DEVI void zero_pads(char* buf, int tid) {
    for (int idx = tid; idx < (TY + 2) * NPAD * 8; idx += 256) {
      const int c = idx & 7, k = (idx >> 3) % NPAD, row = (idx >> 3) / NPAD, hx = k == 0 ? 0 : W + k;
      *(uint4*)(buf + (row * HP + hx) * CH_ROWB + c * 16) = uint4{0, 0, 0, 0};
    }
  }
  // request k of the calling wave; voff: the lane's byte offset inside a group, (l >> 3) * 128 + (((l & 7) ^ (l >> 3)) << 4)
  template <int K>
  DEVI void issue(unsigned voff) const {
    constexpr int r = K / GPR, gx = K % GPR;
    constexpr int col = gx == 0 ? FIRST : gx == GPR - 1 ? LAST : -1;       // EXEC of the copy = sign-extended 32-bit literal
    const char* src = base[r] + gx * 1024;
    if constexpr (col == -1) {
      asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(lds[r]), "n"(gx * 1024)
                   : "memory", "m0", "scc");
    } else {
      // (EXEC from inline constants: lanes 8-63 = -1 << 8, lanes 0-7 = -1 >> 56; a 32-bit literal would not extend to 64 bits)
      if constexpr (col == FIRST)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_lshl_b64 exec, -1, 8\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                     ::"v"(voff), "s"(src), "s"(lds[r]), "n"(gx * 1024) : "memory", "m0", "scc");
      else
        asm volatile("s_add_u32 m0, %2, %3\n\ts_lshr_b64 exec, -1, 56\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                     ::"v"(voff), "s"(src), "s"(lds[r]), "n"(gx * 1024) : "memory", "m0", "scc");
    }
  }
};
template <int LW, int K>
DEVI void ch_issue_all(const ChHalo<LW>& h, unsigned voff) {
  if constexpr (K < ChHalo<LW>::STEPS) {
    h.template issue<K>(voff);
    ch_issue_all<LW, K + 1>(h, voff);
  }
}
// (s is a compile-time constant after unrolling: the switch folds to one call)
template <int LW>
DEVI void ch_issue_step(const ChHalo<LW>& h, unsigned voff, int s) {
#define CH_CASE(K) case K: if constexpr (K < ChHalo<LW>::STEPS) h.template issue<K>(voff); break;
  switch (s) {
    CH_CASE(0) CH_CASE(1) CH_CASE(2) CH_CASE(3) CH_CASE(4) CH_CASE(5) CH_CASE(6) CH_CASE(7) CH_CASE(8) CH_CASE(9)
    CH_CASE(10) CH_CASE(11) CH_CASE(12) CH_CASE(13) CH_CASE(14) CH_CASE(15) CH_CASE(16) CH_CASE(17)
  }
#undef CH_CASE
}

// byte offset of (tap row g / 3, fragment i of the wave's 64-pixel strip) from the lane base of tap column g % 3
template <int LW>
DEVI constexpr int ch_frag_off(int g, int i) {
  return ((g / 3) * ch_hp(1 << LW) + ((16 * i) >> LW) * ch_hp(1 << LW) + ((16 * i) & ((1 << LW) - 1))) * CH_ROWB;
}

template <int LW>
__global__ __launch_bounds__(256) void conv3x3_c64_halo_kernel(ConvHaloArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W = 1 << LW, TY = 256 >> LW, HP = ch_hp(W), HALO = ch_halo_bytes(W);
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int fr = l & 15, fq = l >> 4;
  float* ex = (float*)smem;
  char* hbuf = smem + CH_EX;
  const long tiles_per_frame = ((long)a.H * W) >> 8, ntiles = tiles_per_frame * a.frames;
  // consecutive tiles per workgroup: the two halo rows a tile shares with the next were just read through this XCD's L2
  const long per = (ntiles + gridDim.x - 1) / gridDim.x, t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  if (t0 >= t1) return;
#define CH_STAMP(tile, slot) \
  if (a.ts && tid == 0 && (tile) < 8) a.ts[((long)blockIdx.x * 8 + (tile)) * 8 + (slot)] = wall_clock64()
  CH_STAMP(0, 6);
  ChHalo<LW> halo;
  halo.set(a, t0, tiles_per_frame, hbuf, w, true);
  ChHalo<LW>::zero_pads(hbuf, tid);
  const unsigned voff = (l >> 3) * CH_ROWB + (((l & 7) ^ (l >> 3)) << 4);
  ch_issue_all<LW, 0>(halo, voff);
  // Weights: global -> LDS once per workgroup (second halo buffer: free until the first tile's multiply loop), then -> registers,
  // indexed by GEOMETRIC tap g (offset (g / 3 - 1, g % 3 - 1)): matrix tap g forward, 8 - g for the input gradient.  MFMA row m of
  // channel tile j carries output channel 16 (m / 4) + 4 j + (m % 4): lane (fr, fq) then ends up with the 16 CONSECUTIVE channels
  // 16 fq .. 16 fq + 15 of its pixel (two 16-byte stores per pixel).
  {
    char* wl = hbuf + HALO;
    for (int d0 = w * 1024; d0 < CH_WBYTES; d0 += 4 * 1024) {
      const int d = d0 + l * 16, row = d / CH_WPITCH, col = d - row * CH_WPITCH;
      const char* src = col < 1152 ? (const char*)a.Wm + row * 1152 + col : (const char*)g_stswin_zero;
      const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)wl + d0);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    }
  }
  wait_vm0();
  __syncthreads();
  bf16x8 wf[9][2][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const char* wr = hbuf + HALO + (16 * (fr >> 2) + 4 * j + (fr & 3)) * CH_WPITCH + fq * 16;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      const char* wt = wr + (a.sign > 0 ? g : 8 - g) * 128;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wf[g][kh][j] = *(const bf16x8*)(wt + kh * 64);
    }
  }
  // Fragment addresses inside a halo buffer: lane base per (tap column, k half) - the swizzle term depends on (fr + column) & 7
  // only, everything else (tap row, the wave's four 16-pixel fragments) is a compile-time byte offset of the ds_read.
  const int p0 = 64 * w, sy0 = p0 >> LW, sx0 = p0 & (W - 1);
  int abase[3][2];
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
      abase[dxi][kh] = (sy0 * HP + sx0 + fr + dxi) * CH_ROWB + ((((kh * 4 + fq) ^ (fr + dxi)) & 7) << 4);
  int buf = 0;
  CH_STAMP(0, 7);
  for (long t = t0; t < t1; ++t, buf ^= 1) {
    CH_STAMP(t - t0, 0);
    wait_vm0();
    __syncthreads();                               // halo t is in place; every wave is done reading the other buffer (weights: too)
    CH_STAMP(t - t0, 1);
    const char* hl = hbuf + buf * HALO;
    const bool more = t + 1 < t1;                  // the next halo is requested inside the multiply loop, one copy group per step
    halo.set(a, more ? t + 1 : t, tiles_per_frame, hbuf + (buf ^ 1) * HALO, w, more);
    if (t == t0 && more) ChHalo<LW>::zero_pads(hbuf + HALO, tid);       // (the second buffer held the weights until the barrier above)
    CH_STAMP(t - t0, 2);
    const int f = (int)(t / tiles_per_frame), y0 = (int)(t - (long)f * tiles_per_frame) * TY;
    const long pix0 = ((long)f * a.H + y0) * W;                           // first output pixel (token row) of the tile
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the residual operand is read in the epilogue (no registers to hold it across the multiply loop): touch its lines now, so
    // that those reads find them in L2 instead of paying an HBM round trip at the end of every tile
    int touch[4] = {0, 0, 0, 0};
    if (a.R) {
      NO_IFCVT;
#pragma unroll
      for (int i = 0; i < 4; ++i) touch[i] = *(const int*)(a.R + (pix0 + 64 * w + 16 * i + fr) * CH_C + 16 * fq);
    }
    // 18 (tap, k half) steps; the pixel fragments of step s + 1 are in flight while the 16 MFMAs of step s are issued (one wave
    // per SIMD: nobody else hides the LDS latency); the copy request of the step and the register shuffling fill MFMA shadows
    bf16x8 xf[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[s][i] = *(const bf16x8*)(hl + abase[(s >> 1) % 3][s & 1] + ch_frag_off<LW>(s >> 1, i));
#pragma unroll
    for (int s = 0; s < 18; ++s) {
      const int g = s >> 1, kh = s & 1;
      __builtin_amdgcn_sched_barrier(0);           // (the fragments of step s + 1 were requested above this line)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[g][kh][j], xf[s & 1][i], acc[i][j], 0, 0, 0);
      if (s + 2 < 18) {
        const int gn = (s + 2) >> 1, khn = s & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[s & 1][i] = *(const bf16x8*)(hl + abase[gn % 3][khn] + ch_frag_off<LW>(gn, i));
      }
      ch_issue_step<LW>(halo, voff, s);
    }
    CH_STAMP(t - t0, 3);
    // ---- epilogue: acc[i][j][r] = (pixel 64 w + 16 i + fr, channel 16 fq + 4 j + r)
    float s1[16], s2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long row = pix0 + 64 * w + 16 * i + fr;
      float v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[i][j][r];
      if (a.R) {
        asm volatile("" ::"v"(touch[i]));
        const bf16x8 r0 = *(const bf16x8*)(a.R + row * CH_C + 16 * fq), r1 = *(const bf16x8*)(a.R + row * CH_C + 16 * fq + 8);
#pragma unroll
        for (int c = 0; c < 8; ++c) { v[c] += (float)r0[c]; v[8 + c] += (float)r1[c]; }
      }
      bf16x8 o0, o1;
#pragma unroll
      for (int c = 0; c < 8; ++c) { o0[c] = (bf16)v[c]; o1[c] = (bf16)v[8 + c]; }
      *(bf16x8*)(a.Y + row * CH_C + 16 * fq) = o0;
      *(bf16x8*)(a.Y + row * CH_C + 16 * fq + 8) = o1;
      if (a.stats) {
        NO_IFCVT;
#pragma unroll
        for (int c = 0; c < 16; ++c) { s1[c] += v[c]; s2[c] += v[c] * v[c]; }
      }
    }
    CH_STAMP(t - t0, 4);
    if (a.stats) {
      // per-128-row-block column sums: the 16 pixel lanes by DPP, then the two waves of a block through LDS, in fixed order
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float u1 = sum16(s1[c]), u2 = sum16(s2[c]);
        if (fr == 0) { ex[(w * 2 + 0) * 64 + 16 * fq + c] = u1; ex[(w * 2 + 1) * 64 + 16 * fq + c] = u2; }
      }
      __syncthreads();
      const int blk = tid >> 7, which = (tid >> 6) & 1, c = tid & 63;      // block = waves (2 blk, 2 blk + 1); sums / sums of squares; channel
      const float v = ex[((2 * blk) * 2 + which) * 64 + c] + ex[((2 * blk + 1) * 2 + which) * 64 + c];
      const long nblk = 2 * ((a.M + 255) >> 8);
      a.stats[((long)which * nblk + ((pix0 >> 7) + blk)) * 64 + c] = v;
      // (the next write of ex comes after the barrier at the top of the next tile)
    }
    CH_STAMP(t - t0, 5);
  }
#undef CH_STAMP
}

template <int LW>
static int ch_launch(const ConvHaloArgs& a, hipStream_t st) {
  constexpr int W = 1 << LW, lds = ch_lds_bytes(W);
  static_assert(lds <= 160 * 1024, "two halo buffers must fit the LDS");
  static const int attr = (int)hipFuncSetAttribute((const void*)conv3x3_c64_halo_kernel<LW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (attr != 0) return -attr;
  const long ntiles = a.M / 256;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipLaunchKernelGGL(conv3x3_c64_halo_kernel<LW>, dim3(grid), dim3(256), lds, st, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_conv3x3_c64(const void* x, const void* wmat, void* y, const void* resid, float* stats, int frames, int H, int W, int sign,
                                  void* stream) {
  // (sign = +-2: a tools run - `stats` is a stamp buffer of 256 * 8 * 8 64-bit words, no statistics)
  if (frames <= 0 || H <= 0 || ((long)H * W) % 256 || sign == 0 || sign < -2 || sign > 2) return -1701;
  const bool dbg = sign == 2 || sign == -2;
  ConvHaloArgs a{(const bf16*)x, (const bf16*)wmat, (bf16*)y, (const bf16*)resid, dbg ? nullptr : stats, frames, H, sign > 0 ? 1 : -1,
                 (long)frames * H * W, dbg ? (unsigned long long*)stats : nullptr};
  switch (W) {
    case 16: return ch_launch<4>(a, (hipStream_t)stream);
    case 32: return ch_launch<5>(a, (hipStream_t)stream);
    case 64: return ch_launch<6>(a, (hipStream_t)stream);
    case 128: return ch_launch<7>(a, (hipStream_t)stream);
  }
  return -1702;                                   // (the caller keeps the gather GEMM for other widths)
}
