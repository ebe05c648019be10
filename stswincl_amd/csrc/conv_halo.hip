// 3x3 / stride 1 / pad 1 convolution of 64 -> 64 channels on NHWC token matrices as an implicit GEMM over an LDS-resident
// halo (torchvision resnet18.layer1 inside seg18/net/Ours/resnet.py:104-105: four such convolutions per frame batch, forward
// and input gradient; the 16-frame batch of a 4-clip step is 262144 pixels of 128x128 maps).
//
// The generic gather GEMM (gemm.hip) stages the A operand tap by tap: nine row-map-driven copies of the same 256 x 64 input
// tile per 256 x 64 output tile, 288 KB through L2 for 33 KB of distinct data - 300 MB per launch, with a row-map lookup and
// a copy request per 1 KB of it - which holds the N = 64 convolutions at 350-420 TFLOP/s (~50 us per launch).  Here
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

// =====================================================================================================================
// Weight gradient of the same convolution:  dW[co][tap][ci] = sum_p dy[p][co] * x[p + off(tap)][ci]   (resnet.py:31-51 backward).
// The generic path (gemm_tn with a tap-segmented, row-map-gathered B operand) reads x nine times and runs at 235 TFLOP/s on
// Mk = 262144, Ni = 64, Nj = 576.  Here a workgroup walks down consecutive 128-pixel units of a frame with the image rows in
// an LDS RING (each row of x is copied once; rows y - 1 .. y + TYU of the current unit stay while the rows of the next unit
// arrive) next to a double-buffered dy tile; both MFMA operands are read TRANSPOSED from their pixel-major tiles
// (ds_read_b64_tr_b16: the contraction index is the pixel), the nine taps are nine address shifts of the x fragments.  Wave w
// owns input-channel tile w (16 channels) of every tap: 9 x 4 accumulator tiles of 16 x 16 stay in registers for the whole run;
// at the end the workgroup's 64 x 576 partial goes into its slab of the caller's workspace, and a fixed-order fold
// (chw_fold_kernel) adds the slabs while it re-orders them: deterministic, no atomics.
// =====================================================================================================================
struct ConvHaloWgradArgs {
  const bf16* X; const bf16* DY; float* ws;
  int frames, H, tapminor; long per;            // units per workgroup
  unsigned long long* ts;                       // tools/conv_halo_timeline.py --wgrad: [workgroup][16 units][4 slots] stamps of wave 0, or NULL
};

template <int LW>
struct ChWg {
  static constexpr int W = 1 << LW, TYU = 128 >> LW, HP = ch_hp(W), GPR = HP / 8;
  static constexpr int RING = TYU == 1 ? 4 : TYU == 2 ? 8 : 16;             // image-row slots: >= 2 TYU + 2, power of two
  static constexpr int ROWB = HP * CH_ROWB, XB = RING * ROWB, DYB = 128 * CH_ROWB, LDS = XB + 2 * DYB;
  static constexpr int NPAD = HP - W;
  static_assert(RING >= 2 * TYU + 2 && LW >= 5 && LW <= 7, "unit = 128 pixels of whole image rows, 32 <= W <= 128");
  static_assert(LDS <= 160 * 1024, "ring + dy tiles must fit the LDS");
};

DEVI bf16x4 ch_tr4(const char* addr) {
  short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
  return __builtin_bit_cast(bf16x4, r);
}

// One copy group (8 pixel records, 1 KB) of image row y of frame f into its ring slot, or - n past the x groups - of the dy tile
// of `unit`.  n, y0, nrows are wave-uniform.  Rows outside the image copy zeros; the pad lanes of the first / last group of a
// row are masked out (their LDS positions were zeroed once).
template <int LW>
DEVI void chw_issue(const ConvHaloWgradArgs& a, int n, int f, int ybase, int nrows, long unit, char* xr, char* dyt, unsigned voff) {
  using C = ChWg<LW>;
  const int NX = nrows * C::GPR;
  if (n < NX) {
    const int r = n / C::GPR, gx = n - r * C::GPR, y = ybase + r, slot = (y + 1) & (C::RING - 1);
    const bool real = (unsigned)y < (unsigned)a.H;
    const char* src = (const char*)ch_uniform((unsigned long)(real ? (const char*)(a.X + (((long)f * a.H + y) * C::W + gx * 8 - 1) * CH_C)
                                                                    : (const char*)ch_zero));
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(xr + slot * C::ROWB + gx * 1024));
    if (gx == 0)
      asm volatile("s_mov_b32 m0, %2\n\ts_lshl_b64 exec, -1, 8\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(src), "s"(dst)
                   : "memory", "m0", "scc");
    else if (gx == C::GPR - 1)
      asm volatile("s_mov_b32 m0, %2\n\ts_lshr_b64 exec, -1, 56\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(src), "s"(dst)
                   : "memory", "m0", "scc");
    else
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0");
  } else if (n < NX + 16) {
    const int gd = n - NX;
    const char* src = (const char*)ch_uniform((unsigned long)(a.DY + (unit * 128 + gd * 8) * CH_C));
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(dyt + gd * 1024));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0");
  }
}

template <int LW>
__global__ __launch_bounds__(512) void conv3x3_c64_wgrad_kernel(ConvHaloWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = ChWg<LW>;
  constexpr int W = C::W, TYU = C::TYU, HP = C::HP;
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int fr = l & 15, fq = l >> 4, q = fr >> 2, pq = fr & 3;
  char* xr = smem;
  char* dyb = smem + C::XB;
  const long upf = ((long)a.H * W) >> 7, nunits = upf * a.frames;
  const long u0 = blockIdx.x * a.per, u1 = u0 + a.per < nunits ? u0 + a.per : nunits;
#define CHW_STAMP(unit, slot) \
  if (a.ts && tid == 0 && (unit) < 16) a.ts[((long)blockIdx.x * 16 + (unit)) * 4 + (slot)] = wall_clock64()
  CHW_STAMP(15, 0);
  // Eight waves, two per SIMD: 0-3 multiply (wave w = input-channel tile w), 4-7 request the copies - an LDS-DMA request costs its
  // wave 60-180 cycles of issue (plus its scalar address arithmetic), which a multiplying wave would pay in idle matrix cycles.
  if (w >= 4) {
    const int lw = w - 4;
    const unsigned voff = (l >> 3) * CH_ROWB + (((l & 7) ^ (l >> 3)) << 4);
    bool full = true;
    for (long u = u0; u < u1; ++u) {
      const int f = (int)(u / upf), y0 = (int)(u - (long)f * upf) * TYU;
      if (full) {                                   // first unit of the run / of a frame: rows y0 - 1 .. y0 + TYU and the dy tile, not overlapped
        if (u > u0) __syncthreads();                // (the previous frame's last unit is still being read)
        for (int n = lw; n < (TYU + 2) * C::GPR + 16; n += 4) chw_issue<LW>(a, n, f, y0 - 1, TYU + 2, u, xr, dyb + (int)(u & 1) * C::DYB, voff);
      }
      wait_vm0();
      __syncthreads();                              // unit u is in place; the multiplying waves are done with unit u - 1
      const bool next = u + 1 < u1 && y0 + TYU < a.H;   // the next unit continues this frame: its TYU new rows + dy tile arrive during this one
      full = u + 1 < u1 && !next;
      if (next)
        for (int n = lw; n < TYU * C::GPR + 16; n += 4) chw_issue<LW>(a, n, f, y0 + TYU + 1, TYU, u + 1, xr, dyb + (int)((u + 1) & 1) * C::DYB, voff);
    }
    return;
  }
  // pad columns of every ring slot: zero, once (nothing is copied there)
  for (int idx = tid; idx < C::RING * C::NPAD * 8; idx += 256) {
    const int c = idx & 7, k = (idx >> 3) % C::NPAD, row = (idx >> 3) / C::NPAD, hx = k == 0 ? 0 : W + k;
    *(uint4*)(xr + row * C::ROWB + hx * CH_ROWB + c * 16) = uint4{0, 0, 0, 0};
  }
  // lane parts of the transposed fragment addresses (a 16-lane group reads 4 pixels x 16 channels; lane = pixel q, channels 4 pq ..):
  // the 128-byte-row swizzle term depends on (pixel & 7) = (4 h + q [+ tap column]) & 7 only
  int lb[3][2], la[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi)
      lb[dxi][h] = (8 * fq + 4 * h + q + dxi) * CH_ROWB + ((((2 * w + (pq >> 1)) ^ ((4 * h + q + dxi) & 7)) & 7) << 4) + (pq & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      la[i][h] = (8 * fq + 4 * h + q) * CH_ROWB + ((((2 * i + (pq >> 1)) ^ ((4 * h + q) & 7)) & 7) << 4) + (pq & 1) * 8;
  }
  f32x4 acc[9][4];
#pragma unroll
  for (int g = 0; g < 9; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool full = false;
  for (long u = u0; u < u1; ++u) {
    const int y0 = (int)(u % upf) * TYU;
    const char* dyt = dyb + (int)(u & 1) * C::DYB;
    CHW_STAMP(u - u0, 0);
    if (full) __syncthreads();
    __syncthreads();                                // unit u is in place
    CHW_STAMP(u - u0, 1);
    full = u + 1 < u1 && !(y0 + TYU < a.H);
    // ring offsets of the unit's image rows y0 - 1 + r
    int rowoff[TYU + 2];
#pragma unroll
    for (int r = 0; r < TYU + 2; ++r) rowoff[r] = ((y0 + r) & (C::RING - 1)) * C::ROWB;
    // 12 steps n = (pixel step ks = n / 3, tap row dyi = n % 3): the three x fragments of the step (tap columns) against the four dy
    // fragments of ks (output-channel tiles): 12 MFMAs.  x fragments are requested two steps ahead, dy fragments one pixel step ahead.
    bf16x8 af[2][4], bf[3][3];
#define CHW_LOAD_A(ks)                                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                  \
    af[(ks) & 1][i] = cat4(ch_tr4(dyt + la[i][0] + (ks) * 32 * CH_ROWB), ch_tr4(dyt + la[i][1] + (ks) * 32 * CH_ROWB));
#define CHW_LOAD_B(n)                                                                                                           \
  {                                                                                                                              \
    const char* base = xr + rowoff[((32 * ((n) / 3)) >> LW) + (n) % 3] + ((32 * ((n) / 3)) & (W - 1)) * CH_ROWB;                 \
    _Pragma("unroll") for (int dxi = 0; dxi < 3; ++dxi) bf[(n) % 3][dxi] = cat4(ch_tr4(base + lb[dxi][0]), ch_tr4(base + lb[dxi][1])); \
  }
    CHW_LOAD_A(0)
    CHW_LOAD_B(0)
    CHW_LOAD_B(1)
#pragma unroll
    for (int n = 0; n < 12; ++n) {
      const int ks = n / 3, dyi = n % 3;
      if (n + 2 < 12) CHW_LOAD_B(n + 2)
      if (dyi == 0 && ks + 1 < 4) { CHW_LOAD_A(ks + 1) }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[3 * dyi + dxi][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1][i], bf[n % 3][dxi], acc[3 * dyi + dxi][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef CHW_LOAD_A
#undef CHW_LOAD_B
    CHW_STAMP(u - u0, 2);
  }
  CHW_STAMP(15, 1);
  // the workgroup's partial -> its slab, straight from the registers, in the order [tap][ci][co]: a lane holds 4 consecutive
  // output channels per (tap, tile), the four lanes of a column write one 64-byte segment; chw_fold_kernel re-orders while it adds
  float* slab = a.ws + (long)blockIdx.x * (64 * 576);
#pragma unroll
  for (int g = 0; g < 9; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(slab + (g * 64 + 16 * w + fr) * 64 + 16 * i + 4 * fq) = acc[g][i];
  CHW_STAMP(15, 3);
#undef CHW_STAMP
}

// out = [+] sum over slabs, re-ordered from the kernels' [column][co] (column = tap * 64 + ci; ncol of them) to [co][column] or (tapminor) [co][ci][tap].  A workgroup owns
// 16 float4 columns (256 contiguous bytes of every slab) and reads them with 16 slab lanes; lane sums meet in LDS and are added in
// lane order: the association of every output's sum is fixed by the slab count alone.
__global__ __launch_bounds__(256) void chw_fold_kernel(const float* ws, int nslabs, int ncol, float* out, int tapminor, int accumulate) {
  const long slab = 64L * ncol;
  __shared__ f32x4 part[256];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int j = (blockIdx.x * 16 + cl) * 4;         // slab offset of the column: (tap * 64 + ci) * 64 + co, co % 4 == 0
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int p = sl;
  for (; p + 48 < nslabs; p += 64) {
    const f32x4 v0 = *(const f32x4*)(ws + (long)p * slab + j), v1 = *(const f32x4*)(ws + (long)(p + 16) * slab + j);
    const f32x4 v2 = *(const f32x4*)(ws + (long)(p + 32) * slab + j), v3 = *(const f32x4*)(ws + (long)(p + 48) * slab + j);
    v += v0; v += v1; v += v2; v += v3;
  }
  for (; p < nslabs; p += 16) v += *(const f32x4*)(ws + (long)p * slab + j);
  part[sl * 16 + cl] = v;
  __syncthreads();
  if (sl == 0) {
    for (int k = 1; k < 16; ++k) v += part[k * 16 + cl];
    const int co = j & 63, col = j >> 6, ci = col & 63, g = col >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* dst = out + (tapminor ? ((co + r) * 64 + ci) * 9 + g : (co + r) * ncol + col);
      *dst = accumulate ? *dst + v[r] : v[r];
    }
  }
}

template <int LW>
static int chw_launch(const ConvHaloWgradArgs& a, int grid, hipStream_t st) {
  static const int attr = (int)hipFuncSetAttribute((const void*)conv3x3_c64_wgrad_kernel<LW>, hipFuncAttributeMaxDynamicSharedMemorySize, ChWg<LW>::LDS);
  if (attr != 0) return -attr;
  hipLaunchKernelGGL(conv3x3_c64_wgrad_kernel<LW>, dim3(grid), dim3(512), ChWg<LW>::LDS, st, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" long stswin_conv3x3_c64_wgrad_scratch(int frames, int H, int W) {
  const long nunits = (long)frames * H * W / 128;
  return (nunits < 256 ? nunits : 256) * 64 * 576;
}

extern "C" int stswin_conv3x3_c64_wgrad(const void* dy, const void* x, float* dw, int tapminor, int accumulate, float* scratch, long scratch_floats,
                                        int frames, int H, int W, void* stream) {
  if (frames <= 0 || H <= 0 || ((long)H * W) % 128 || !dw) return -1711;
  if (W != 32 && W != 64 && W != 128) return -1712;
  const long nunits = (long)frames * H * W / 128;
  const long g0 = nunits < 256 ? nunits : 256, per = (nunits + g0 - 1) / g0;
  const int grid = (int)((nunits + per - 1) / per);             // every workgroup has at least one unit: every slab is written
  if (!scratch || scratch_floats < (long)grid * 64 * 576) return -1713;
  // (accumulate == 2: a tools run - dw is a stamp buffer of grid * 16 * 4 64-bit words, nothing is folded)
  ConvHaloWgradArgs a{(const bf16*)x, (const bf16*)dy, scratch, frames, H, tapminor ? 1 : 0, per, accumulate == 2 ? (unsigned long long*)dw : nullptr};
  int rc = W == 128 ? chw_launch<7>(a, grid, (hipStream_t)stream) : W == 64 ? chw_launch<6>(a, grid, (hipStream_t)stream) : chw_launch<5>(a, grid, (hipStream_t)stream);
  if (rc || accumulate == 2) return rc;
  hipLaunchKernelGGL(chw_fold_kernel, dim3(576), dim3(256), 0, (hipStream_t)stream, scratch, grid, 576, dw, tapminor ? 1 : 0, accumulate ? 1 : 0);
  STSWIN_CHECK_LAUNCH();
  return 0;
}


// =====================================================================================================================
// Weight gradient of the stem convolution over the space-to-depth image (stem_s2d_kernel of rowops.hip; resnet.py:98-102 backward):
//   dW[co][s][t][ch] = sum_p dy[p][co] * rec[(oy + s, ox + t)][ch]        s, t = 0..3 tap row / record, ch = 0..15
// The gemm_tn path gathers 4 x 128 bytes per output pixel (537 MB for 16 frames of 512 x 512; 150 us).  Here a workgroup walks
// down a 128-pixel-wide strip of output rows: per unit ONE new row segment of records (131 x 32 bytes) enters an 8-slot LDS ring
// and one dy tile (128 x 128 bytes) a double buffer - 168 MB in all.  Wave w owns tap row s = w: 4 records x 4 output-channel
// tiles of accumulators; both operands are read transposed (the contraction index is the pixel; lane group fq takes pixels
// 4 fq .. 4 fq + 3 and 16 + 4 fq .. of a 32-pixel step - any assignment is legal if both operands agree, and this one makes the
// 32-lane halves of a record read contiguous); waves 4-7 request the copies.  Partials go to slabs [s][t][ch][co], chw_fold_kernel
// adds them in a fixed order into [co][s][t][ch].
// =====================================================================================================================
struct StemWgradArgs {
  const bf16* REC; const bf16* DY; float* ws;
  int frames, Ho, Wo, Hs, Ws; long per;
};

constexpr int SW_ROWP = 136 * 32;                       // ring slot pitch: 131 records used
constexpr int SW_XB = 8 * SW_ROWP, SW_DYB = 128 * CH_ROWB, SW_LDS = SW_XB + 2 * SW_DYB;

// copy n of a unit: n < 5 nrows: part n % 5 of padded record row row0 + n / 5 (columns x0 .. x0 + 130); then the 16 groups of the dy tile
DEVI void sw_issue(const StemWgradArgs& a, int n, int nrows, int f, int row0, int x0, long pix0, char* xr, char* dyt, int l) {
  if (n < 5 * nrows) {
    const int r = n / 5, part = n - 5 * r, y = row0 + r, slot = y & 7;
    const char* src = (const char*)ch_uniform((unsigned long)(a.REC + (((long)f * a.Hs + y) * a.Ws + x0) * 16 + part * 512));
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(xr + slot * SW_ROWP + part * 1024));
    const unsigned voff = l * 16;
    if (part < 4)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0");
    else                                            // records 128 .. 130: 6 lanes
      asm volatile("s_mov_b32 m0, %2\n\ts_lshr_b64 exec, -1, 58\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(voff), "s"(src), "s"(dst)
                   : "memory", "m0", "scc");
  } else if (n < 5 * nrows + 16) {
    const int gd = n - 5 * nrows;
    const char* src = (const char*)ch_uniform((unsigned long)(a.DY + (pix0 + gd * 8) * CH_C));
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)(dyt + gd * 1024));
    const unsigned voff = (l >> 3) * CH_ROWB + (((l & 7) ^ (l >> 3)) << 4);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(src), "s"(dst) : "memory", "m0");
  }
}

__global__ __launch_bounds__(512) void stem_wgrad_kernel(StemWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int fr = l & 15, fq = l >> 4, q = fr >> 2, pq = fr & 3;
  char* xr = smem;
  char* dyb = smem + SW_XB;
  const int nxh = a.Wo >> 7;
  const long nunits = (long)a.frames * nxh * a.Ho;
  const long u0 = blockIdx.x * a.per, u1 = u0 + a.per < nunits ? u0 + a.per : nunits;
  // unit u = (strip = u / Ho -> frame strip / nxh, column block strip % nxh; output row oy = u % Ho)
  if (w >= 4) {
    const int lw = w - 4;
    bool full = true;
    for (long u = u0; u < u1; ++u) {
      const int strip = (int)(u / a.Ho), oy = (int)(u - (long)strip * a.Ho), f = strip / nxh, x0 = (strip - f * nxh) << 7;
      const long pix0 = ((long)f * a.Ho + oy) * a.Wo + x0;
      if (full) {
        if (u > u0) __syncthreads();
        for (int n = lw; n < 5 * 4 + 16; n += 4) sw_issue(a, n, 4, f, oy, x0, pix0, xr, dyb + (int)(u & 1) * SW_DYB, l);
      }
      wait_vm0();
      __syncthreads();
      const bool next = u + 1 < u1 && oy + 1 < a.Ho;
      full = u + 1 < u1 && !next;
      if (next)
        for (int n = lw; n < 5 + 16; n += 4) sw_issue(a, n, 1, f, oy + 4, x0, pix0 + a.Wo, xr, dyb + (int)((u + 1) & 1) * SW_DYB, l);
    }
    return;
  }
  // lane parts of the transposed fragment addresses: pixel 4 fq + q (+ 16 h + 32 ks), channels 4 pq ..
  int la[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) la[i] = (4 * fq + q) * CH_ROWB + ((((2 * i + (pq >> 1)) ^ ((4 * (fq & 1) + q) & 7)) & 7) << 4) + (pq & 1) * 8;
  const int lbB = (4 * fq + q) * 32 + pq * 8;
  f32x4 acc[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool full = false;
  for (long u = u0; u < u1; ++u) {
    const int oy = (int)(u % a.Ho);
    const char* dyt = dyb + (int)(u & 1) * SW_DYB;
    if (full) __syncthreads();
    __syncthreads();                                // unit u is in place
    full = u + 1 < u1 && !(oy + 1 < a.Ho);
    const char* xrow = xr + ((oy + w) & 7) * SW_ROWP + lbB;      // this wave's tap row
    bf16x8 af[2][4], bf[2][4];
#define SW_LOAD(ks)                                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                   \
    af[(ks) & 1][i] = cat4(ch_tr4(dyt + la[i] + (ks) * 32 * CH_ROWB), ch_tr4(dyt + la[i] + ((ks) * 32 + 16) * CH_ROWB));          \
  _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                                   \
    bf[(ks) & 1][t] = cat4(ch_tr4(xrow + ((ks) * 32 + t) * 32), ch_tr4(xrow + ((ks) * 32 + 16 + t) * 32));
    SW_LOAD(0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) { SW_LOAD(ks + 1) }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks & 1][i], bf[ks & 1][t], acc[t][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SW_LOAD
  }
  float* slab = a.ws + (long)blockIdx.x * (64 * 256);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(slab + ((w * 4 + t) * 16 + fr) * 64 + 16 * i + 4 * fq) = acc[t][i];
}

extern "C" long stswin_stem_wgrad_scratch(int frames, int Ho, int Wo) {
  const long nunits = (long)frames * (Wo / 128) * Ho;
  return (nunits < 256 ? nunits : 256) * 64 * 256;
}

extern "C" int stswin_stem_wgrad(const void* dy, const void* rec, float* dw, int accumulate, float* scratch, long scratch_floats, int frames,
                                 int H, int W, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (frames <= 0 || H <= 0 || W <= 0 || !dw) return -1721;
  if (Wo % 128) return -1722;                                   // (the caller keeps stswin_gemm_tn over the row map)
  const long nunits = (long)frames * (Wo / 128) * Ho;
  const long g0 = nunits < 256 ? nunits : 256, per = (nunits + g0 - 1) / g0;
  const int grid = (int)((nunits + per - 1) / per);
  if (!scratch || scratch_floats < (long)grid * 64 * 256) return -1723;
  StemWgradArgs a{(const bf16*)rec, (const bf16*)dy, scratch, frames, Ho, Wo, Ho + 3, Wo + 3, per};
  static const int attr = (int)hipFuncSetAttribute((const void*)stem_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS);
  if (attr != 0) return -attr;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(grid), dim3(512), SW_LDS, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  hipLaunchKernelGGL(chw_fold_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, scratch, grid, 256, dw, 0, accumulate ? 1 : 0);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// =====================================================================================================================
// The stem convolution itself over the space-to-depth image (resnet.py:98-102 forward):
//   y[p][co] = sum_{s, t, ch} rec[(oy + s, ox + t)][ch] * W[co][s][t][ch]          K = 4 * 4 * 16 = 256 (147 of them non-zero)
// The gather GEMM reads 4 x 128 bytes per output pixel through its row map and runs this M = 1 M, N = 64 shape at 230-315 TFLOP/s.
// Same skeleton as stem_wgrad_kernel: a workgroup walks down a 128-pixel-wide strip, one new row of records per unit through the
// 8-slot LDS ring (requested by waves 4-7); waves 0-3 own 32 pixels each, hold the whole 64 x 256 weight matrix in 128 registers
// and read pixel fragments as plain 16-byte LDS loads (a record is 32 bytes: the k group picks the record and its half).  Weights
// are the first MFMA operand with permuted rows, so a lane ends up with 16 consecutive channels of its pixel.  Column sums and
// sums of squares go to the BatchNorm statistics table once per run of units (reducing them over lanes for every unit cost 26 of
// 70 us): the table row of a run's first 128-row block holds the run's sums, the rows of its other blocks are zero.
// =====================================================================================================================
struct StemConvArgs {
  const bf16* REC; const bf16* Wm; bf16* Y; float* stats;
  int frames, Ho, Wo, Hs, Ws; long per, M;
};

constexpr int SC_EX = 4 * 2 * 64 * 4;                   // statistics exchange: [wave][sum | squares][64] floats
constexpr int SC_LDS = SW_XB + SC_EX;

__global__ __launch_bounds__(512) void stem_conv_kernel(StemConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = wave_id();
  const int fr = l & 15, fq = l >> 4;
  char* xr = smem;
  float* ex = (float*)(smem + SW_XB);
  const int nxh = a.Wo >> 7;
  const long nunits = (long)a.frames * nxh * a.Ho;
  const long u0 = blockIdx.x * a.per, u1 = u0 + a.per < nunits ? u0 + a.per : nunits;
  if (w >= 4) {                                        // loader waves: the record rows
    const int lw = w - 4;
    bool full = true;
    StemWgradArgs ia{a.REC, nullptr, nullptr, a.frames, a.Ho, a.Wo, a.Hs, a.Ws, a.per};
    for (long u = u0; u < u1; ++u) {
      const int strip = (int)(u / a.Ho), oy = (int)(u - (long)strip * a.Ho), f = strip / nxh, x0 = (strip - f * nxh) << 7;
      if (full) {
        if (u > u0) __syncthreads();
        for (int n = lw; n < 5 * 4; n += 4) sw_issue(ia, n, 4, f, oy, x0, 0, xr, nullptr, l);
      }
      wait_vm0();
      __syncthreads();
      const bool next = u + 1 < u1 && oy + 1 < a.Ho;
      full = u + 1 < u1 && !next;
      if (next)
        for (int n = lw; n < 5; n += 4) sw_issue(ia, n, 1, f, oy + 4, x0, 0, xr, nullptr, l);
      if (a.stats && !next) __syncthreads();           // (end of a run inside one strip: the multiplying waves exchange their sums)
    }
    return;
  }
  // weights -> registers: MFMA row m of channel tile j = channel 16 (m / 4) + 4 j + (m % 4), k group fq of step ks
  bf16x8 wf[8][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16* wr = a.Wm + (long)(16 * (fr >> 2) + 4 * j + (fr & 3)) * 256 + fq * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wf[ks][j] = *(const bf16x8*)(wr + ks * 32);
  }
  // pixel fragment address: record (tap row s = ks / 2 -> ring slot; column pixel + t, t = 2 (ks & 1) + fq / 2), half fq & 1
  const int lane_off = (32 * w + fr + (fq >> 1)) * 32 + (fq & 1) * 16;
  // BatchNorm statistics: a lane keeps the column sums | sums of squares of its pixels over a whole RUN (consecutive units of one
  // strip of one frame); they are reduced over lanes and waves once per run and stored in the table row of the run's first
  // 128-row block, the rows of its other blocks are zero - the table's readers add the rows of whole frames.
  float s1[16], s2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  const long nblk = 2 * ((a.M + 255) >> 8);
  long run_blk = -1;
  bool full = false;
  for (long u = u0; u < u1; ++u) {
    const int strip = (int)(u / a.Ho), oy = (int)(u - (long)strip * a.Ho), f = strip / nxh, x0 = (strip - f * nxh) << 7;
    const long pix0 = ((long)f * a.Ho + oy) * a.Wo + x0;
    if (full) __syncthreads();
    __syncthreads();                                   // unit u is in place
    const bool next = u + 1 < u1 && oy + 1 < a.Ho;
    full = u + 1 < u1 && !next;
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[2][2];
#define SC_LOAD(ks)                                                                                                               \
  {                                                                                                                                \
    const char* rb = xr + ((oy + ((ks) >> 1)) & 7) * SW_ROWP + lane_off + ((ks) & 1) * 2 * 32;                                       \
    xf[(ks) & 1][0] = *(const bf16x8*)rb;                                                                                          \
    xf[(ks) & 1][1] = *(const bf16x8*)(rb + 16 * 32);                                                                              \
  }
    SC_LOAD(0)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) SC_LOAD(ks + 1)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], xf[ks & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SC_LOAD
    // epilogue: acc[i][j][r] = (pixel 32 w + 16 i + fr, channel 16 fq + 4 j + r)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long row = pix0 + 32 * w + 16 * i + fr;
      bf16x8 o0, o1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[i][j][r];
          const int c = 4 * j + r;
          if (c < 8) o0[c] = (bf16)v; else o1[c - 8] = (bf16)v;
          s1[c] += v;
          s2[c] += v * v;
        }
      *(bf16x8*)(a.Y + row * CH_C + 16 * fq) = o0;
      *(bf16x8*)(a.Y + row * CH_C + 16 * fq + 8) = o1;
    }
    if (a.stats) {
      NO_IFCVT;
      const long blk = pix0 >> 7;
      if (run_blk < 0) run_blk = blk;
      else if ((int)(u & 3) == w) {                    // not the run's first block: a row of zeros
        a.stats[(0 * nblk + blk) * 64 + l] = 0.f;
        a.stats[(1 * nblk + blk) * 64 + l] = 0.f;
      }
      if (!next) {                                     // end of the run: lanes -> waves -> the table row of its first block
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float t1 = sum16(s1[c]), t2 = sum16(s2[c]);
          if (fr == 0) { ex[(w * 2 + 0) * 64 + 16 * fq + c] = t1; ex[(w * 2 + 1) * 64 + 16 * fq + c] = t2; }
          s1[c] = 0.f;
          s2[c] = 0.f;
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float t = ex[(0 * 2 + h) * 64 + l] + ex[(1 * 2 + h) * 64 + l] + ex[(2 * 2 + h) * 64 + l] + ex[(3 * 2 + h) * 64 + l];
            a.stats[(h * nblk + run_blk) * 64 + l] = t;
          }
        }
        run_blk = -1;
      }
    }
  }
}

extern "C" int stswin_stem_conv(const void* rec, const void* wmat, void* y, float* stats, int frames, int H, int W, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (frames <= 0 || H <= 0 || W <= 0 || !y) return -1731;
  if (Wo % 128) return -1732;                                   // (the caller keeps stswin_gemm_nt over the row map)
  const long nunits = (long)frames * (Wo / 128) * Ho;
  const long g0 = nunits < 256 ? nunits : 256, per = (nunits + g0 - 1) / g0;
  const int grid = (int)((nunits + per - 1) / per);
  StemConvArgs a{(const bf16*)rec, (const bf16*)wmat, (bf16*)y, stats, frames, Ho, Wo, Ho + 3, Wo + 3, per, (long)frames * Ho * Wo};
  static const int attr = (int)hipFuncSetAttribute((const void*)stem_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LDS);
  if (attr != 0) return -attr;
  hipLaunchKernelGGL(stem_conv_kernel, dim3(grid), dim3(512), SC_LDS, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

