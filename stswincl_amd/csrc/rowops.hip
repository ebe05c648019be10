// HBM-bound row kernels: window gather / scatter (bit-exact indexing), row-map builders, LayerNorm, column sums.
// Every kernel moves whole 16-byte pieces per lane (8 bf16 / 4 f32), one wave per token row.
#include "common.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------
// Source token of row `r` of the pair-regrouped window layout  (B*nW, T, ws*ws)  <-  (B, T, H*W).
// Restates swin_512.py:210-218: roll(-s,-s) -> window_partition -> view(B,T,nW,N) -> permute(0,2,1,3).
// Returned index is into a token list of `frames_total` frames per clip starting at frame `f0`
// (so a layer that only touches frames 1..2 of a 4-frame clip needs no slice copy).
// ---------------------------------------------------------------------------------------------------
DEVI int win_src_token(int r, int T, int H, int W, int ws, int shift, int f0, int frames_total) {
  const int N = ws * ws, nwx = W / ws, nW = (H / ws) * nwx;
  const int n = r % N; int q = r / N;
  const int t = q % T; q /= T;
  const int wi = q % nW; const int b = q / nW;
  const int ys = (wi / nwx) * ws + n / ws, xs = (wi % nwx) * ws + n % ws;   // position in the rolled image
  int y = ys + shift, x = xs + shift;                                        // rolled[i] = orig[(i+s) mod n]
  if (y >= H) y -= H;
  if (x >= W) x -= W;
  return ((b * frames_total + f0 + t) * H + y) * W + x;
}

__global__ void win_rowmap_kernel(int* map, int rows, int T, int H, int W, int ws, int shift, int f0, int ftot) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) map[r] = win_src_token(r, T, H, W, ws, shift, f0, ftot);
}

// dir = 0: out[r] = in[src(r)]   (gather: a1+a3+a4)      dir = 1: out[src(r)] = in[r]   (scatter: a2+a3^-1)
template <typename T>
__global__ __launch_bounds__(256) void win_move_kernel(const T* in, T* out, int rows, int C, int Tt, int H, int W,
                                                        int ws, int shift, int f0, int ftot, int dir) {
  constexpr int PACK = TT<T>::PACK;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int s = win_src_token(r, Tt, H, W, ws, shift, f0, ftot);
  const long from = dir == 0 ? (long)s : (long)r, to = dir == 0 ? (long)r : (long)s;
  const uint4* src = (const uint4*)(in + from * C);
  uint4* dst = (uint4*)(out + to * C);
  for (int c = threadIdx.x & 63; c < C / PACK; c += 64) dst[c] = src[c];
}

// Patch-merging 2x2 gather map (swin_512.py:267-271): segment order [(0,0),(1,0),(0,1),(1,1)] = (dy,dx).
__global__ void merge_rowmap_kernel(int* map, int frames, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, M = frames * Ho * Wo;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  const int xo = r % Wo, yo = (r / Wo) % Ho, f = r / (Wo * Ho);
  const int dy[4] = {0, 1, 0, 1}, dx[4] = {0, 0, 1, 1};
#pragma unroll
  for (int s = 0; s < 4; ++s) map[s * M + r] = (f * H + 2 * yo + dy[s]) * W + 2 * xo + dx[s];
}

// 3x3 (dilated) convolution tap map: segment s = ky*3+kx -> source pixel or -1 (zero padding = dilation).
__global__ void conv3x3_rowmap_kernel(int* map, int frames, int H, int W, int dil) {
  const int M = frames * H * W;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  const int x = r % W, y = (r / W) % H, f = r / (W * H);
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const int yy = y + (s / 3 - 1) * dil, xx = x + (s % 3 - 1) * dil;
    map[s * M + r] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (f * H + yy) * W + xx : -1;
  }
}

// general k x k tap maps (forward gather and its inverse for dgrad)
__global__ void conv_rowmap_kernel(int* map, int frames, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad,
                                   int dil, int inverse) {
  const int Mo = frames * Hout * Wout, Mi = frames * Hin * Win;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (!inverse) {
    if (r >= Mo) return;
    const int x = r % Wout, y = (r / Wout) % Hout, f = r / (Wout * Hout);
    for (int t = 0; t < k * k; ++t) {
      const int yy = y * stride - pad + (t / k) * dil, xx = x * stride - pad + (t % k) * dil;
      map[(long)t * Mo + r] = (yy >= 0 && yy < Hin && xx >= 0 && xx < Win) ? (f * Hin + yy) * Win + xx : -1;
    }
  } else {
    if (r >= Mi) return;
    const int x = r % Win, y = (r / Win) % Hin, f = r / (Win * Hin);
    for (int t = 0; t < k * k; ++t) {
      const int ny = y + pad - (t / k) * dil, nx = x + pad - (t % k) * dil;
      int v = -1;
      if (ny >= 0 && nx >= 0 && ny % stride == 0 && nx % stride == 0) {
        const int yo = ny / stride, xo = nx / stride;
        if (yo < Hout && xo < Wout) v = (f * Hout + yo) * Wout + xo;
      }
      map[(long)t * Mi + r] = v;
    }
  }
}

// stem im2col: a thread owns ONE 16-byte chunk position of the patch row (PACK consecutive columns; column = tap*3 +
// channel, columns >= 147 are the zero padding up to ld) and walks over pixels, so the column -> (dy, dx, channel)
// decode is done once per thread and the row leaves in whole 16-byte stores.  Measured 290-300 us for 403 MB of patches
// in all three variants tried (per-tap 2-byte stores, per-chunk, this one): the bound is the gather itself - a wave load
// touches up to 64 distinct cache lines (taps x channels x pixels) - so the next step would be staging the 7 input rows
// of a pixel strip in LDS.  Runs once per step (the patches are kept for the weight gradient).
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* img, T* patches, long ld, int frames, int H, int W,
                                                           int Ho, int Wo) {
  // One workgroup = TP consecutive output pixels of one output row.  The 7 input rows x (2 TP + 5) columns x 3 channels they
  // read are staged in LDS with coalesced loads (zeros outside the image), then every thread assembles 16-byte pieces of
  // patch rows from LDS.  (Reading the taps straight from the NCHW image - eight scattered, bounds-tested 4-byte loads per
  // 16-byte store - ran at 1.4 TB/s of patch writes: 290 us for the 403 MB of a 16-frame 512x512 batch.)
  constexpr int PACK = TT<T>::PACK;
  constexpr int TPMAX = 40, LW = 2 * TPMAX + 8;  // LDS row pitch (floats)
  __shared__ float win[3][7][LW];
  const int cpr = (int)(ld / PACK);              // chunks per row (ld is a multiple of PACK)
  const int ppb = 256 / cpr;                     // pixels per pass
  const int TP = min(TPMAX, 4 * ppb);
  const int tiles_x = (Wo + TP - 1) / TP;
  const int tx = blockIdx.x % tiles_x;
  const int y = (blockIdx.x / tiles_x) % Ho, f = blockIdx.x / (tiles_x * Ho);
  const int x0 = tx * TP, ncols = 2 * TP + 5;
  const float* base = img + (long)f * 3 * H * W;
  for (int i = threadIdx.x; i < 3 * 7 * ncols; i += 256) {
    const int c = i / (7 * ncols), r = (i / ncols) % 7, q = i % ncols;
    const int yy = 2 * y - 3 + r, xx = 2 * x0 - 3 + q;
    win[c][r][q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? base[((long)c * H + yy) * W + xx] : 0.f;
  }
  __syncthreads();
  const int ch = threadIdx.x % cpr, pl0 = threadIdx.x / cpr;
  if (pl0 >= ppb) return;
  int off[PACK];                                 // LDS offset of tap element e relative to the pixel's window origin, or -1
#pragma unroll
  for (int e = 0; e < PACK; ++e) {
    const int col = ch * PACK + e, t = col / 3, c = col - 3 * t;
    off[e] = t < 49 ? (c * 7 + t / 7) * LW + t % 7 : -1;
  }
  const long row0 = ((long)f * Ho + y) * Wo;
  for (int pl = pl0; pl < TP && x0 + pl < Wo; pl += ppb) {
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < PACK; ++e) o.set(e, off[e] >= 0 ? (&win[0][0][0])[off[e] + 2 * pl] : 0.f);
    *(decltype(o.v)*)(patches + (row0 + x0 + pl) * ld + ch * PACK) = o.v;
  }
}

// Stem input as a 2 x 2 SPACE-TO-DEPTH image (torchvision conv1 7x7 / 2 / 3 of resnet.py:98-102 as a 4 x 4 convolution of stride 1
// over 12-channel superpixels): record (f, sy, sx) of the padded [F][Ho + 3][Wo + 3] grid (2 records of padding before, 1 after)
// holds the 2 x 2 x 3 input values img[f][c][2 (sy - 2) + dy][2 (sx - 2) + dx] at position (dy * 2 + dx) * 3 + c, then 4 zeros.
// Output pixel (oy, ox) reads the FOUR records (oy + s, ox .. ox + 3) of each tap row s = 0..3: 64 contiguous values - a segment
// the row-map gather of gemm_nt / gemm_tn takes as it is (row pitch 16 < segment length 64: overlapping rows), so neither the
// forward GEMM nor the weight gradient needs the 147-wide patch matrix (403 MB at 16 frames of 512 x 512; this image: 34 MB).
template <typename T>
__global__ __launch_bounds__(256) void stem_s2d_kernel(const float* img, T* out, int frames, int H, int W, int Hs, int Ws) {
  const long n = (long)blockIdx.x * 256 + threadIdx.x, total = (long)frames * Hs * Ws;
  if (n >= total) return;
  const int sx = (int)(n % Ws), sy = (int)((n / Ws) % Hs), f = (int)(n / ((long)Ws * Hs));
  const int iy = 2 * (sy - 2), ix = 2 * (sx - 2);
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = 0.f;
  const float* base = img + (long)f * 3 * H * W;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int y = iy + dy, x = ix + dx;
      if (y >= 0 && y < H && x >= 0 && x < W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[(dy * 2 + dx) * 3 + c] = base[((long)c * H + y) * W + x];
      }
    }
  constexpr int PACK = TT<T>::PACK;
#pragma unroll
  for (int h = 0; h < 16 / PACK; ++h) {
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < PACK; ++e) o.set(e, v[h * PACK + e]);
    *(decltype(o.v)*)(out + n * 16 + h * PACK) = o.v;
  }
}

// conv weight packing: (Cout, Cin, k, k) fp32 parameter -> the two GEMM operand matrices of the token convolutions in one
// launch: fwd [Cout_p][S][Cin_p] (B operand of y = X W^T, tap-major K) and dgrad [Cin_p][S][Cout_p].  omap / imap give the
// source channel of every padded channel position (-1 = zero padding; concatenated layouts have gaps).  Replaces, per
// convolution and step, a zero fill + slice copies + a cast for each of the two matrices (6 launches of a few us).
// One workgroup = a 64 (out) x 64 (in) channel tile of one tap: the reads run along the input channels, the forward matrix
// is written along them too, and the transposed (dgrad) matrix goes through an LDS tile so that its writes run along the
// output channels.  (One element per thread with a cop-strided 2-byte store for the transposed copy took 1.1 ms per
// training step for the 110 M parameters of TswinPlus - the re-cast happens after every optimizer step.)
template <typename T>
DEVI void conv_pack_tile(T (&tile)[64][66], const float* w, T* fwd, T* dg, const int* omap, const int* imap, int ci, int S, int cop,
                         int cip, int i0, int o0, int s) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  {
    const int ip = i0 + tx;
    const int si = ip < cip ? imap[ip] : -1;
    // three phases, each with its 16 requests in flight together (map entries, weights, stores): as one loop it was 16
    // dependent map -> weight round trips per thread
    int so[16];
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int op = o0 + k * 4 + ty; so[k] = op < cop ? omap[op] : -1; }
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (so[k] >= 0 && si >= 0) ? w[((long)so[k] * ci + si) * S + s] : 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int ol = k * 4 + ty, op = o0 + ol;
      const T o = from_f32<T>(v[k]);
      if (op < cop && ip < cip) fwd[((long)op * S + s) * cip + ip] = o;
      tile[ol][tx] = o;
    }
  }
  if (!dg) return;
  __syncthreads();
  const int op = o0 + tx;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int il = k * 4 + ty, ip = i0 + il;
    if (op < cop && ip < cip) dg[((long)ip * S + s) * cop + op] = tile[tx][il];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* w, T* fwd, T* dg, const int* omap, const int* imap,
                                                         int co, int ci, int S, int cop, int cip) {
  __shared__ T tile[64][66];
  conv_pack_tile<T>(tile, w, fwd, dg, omap, imap, ci, S, cop, cip, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z);
}

// Every convolution weight of a model in ONE launch: the re-cast after an optimizer step was 36 launches of ~15 us for the 36
// convolutions of TswinPlus (and 55 of ~6 us for its Linear weights, below) - kernels far too small to fill the chip, plus a
// launch boundary each.  A workgroup finds its (weight, tile) by scanning the <= 32 tile counts.
#define CPM_MAX 32
struct ConvPackMulti {
  const float* w[CPM_MAX]; void* fwd[CPM_MAX]; void* dg[CPM_MAX]; const int* omap[CPM_MAX]; const int* imap[CPM_MAX];
  int ci[CPM_MAX], S[CPM_MAX], cop[CPM_MAX], cip[CPM_MAX], tiles[CPM_MAX];
  int count;
};
template <typename T>
__global__ __launch_bounds__(256) void conv_pack_multi_kernel(ConvPackMulti a) {
  __shared__ T tile[64][66];
  int b = blockIdx.x, e = 0;
  for (; e < a.count; ++e) {
    if (b < a.tiles[e]) break;
    b -= a.tiles[e];
  }
  if (e >= a.count) return;
  const int ti = (a.cip[e] + 63) / 64, to = (a.cop[e] + 63) / 64;
  const int s = b / (ti * to), r = b - s * ti * to;
  conv_pack_tile<T>(tile, a.w[e], (T*)a.fwd[e], (T*)a.dg[e], a.omap[e], a.imap[e], a.ci[e], a.S[e], a.cop[e], a.cip[e], (r % ti) * 64,
                    (r / ti) * 64, s);
}

// nn.Linear weight [n][k] fp32 -> W (cast) and W^T (cast + transpose) in one launch, 16-byte reads, 8-byte writes: a
// 64 x 64 tile per workgroup, four consecutive elements per thread and pass, the transposed copy through an LDS tile.
template <typename T>
DEVI void linear_pack_tile(float (&tile)[64][65], const float* w, T* fwd, T* tr, int n, int k, int k0, int n0) {
  const int g = threadIdx.x & 15, r = threadIdx.x >> 4;          // 16 groups of 4 columns x 16 rows per pass
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 16 + r, gn = n0 + row, gk = k0 + g * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (gn < n && gk < k) {                                        // (k % 4 == 0: whole pieces)
      v = __builtin_nontemporal_load((const f32x4*)(w + (long)gn * k + gk));   // (the fp32 master weights stream through once: repack 0.275 -> 0.263 ms)
      typedef T tx4 __attribute__((ext_vector_type(4)));
      *(tx4*)(fwd + (long)gn * k + gk) = (tx4){from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row][g * 4 + e] = v[e];
  }
  if (!tr) return;
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 16 + r, gk = k0 + row, gn = n0 + g * 4;   // output row = input column
    if (gk < k && gn < n) {                                         // (n % 4 == 0)
      typedef T tx4 __attribute__((ext_vector_type(4)));
      *(tx4*)(tr + (long)gk * n + gn) = (tx4){from_f32<T>(tile[g * 4 + 0][row]), from_f32<T>(tile[g * 4 + 1][row]),
                                             from_f32<T>(tile[g * 4 + 2][row]), from_f32<T>(tile[g * 4 + 3][row])};
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void linear_pack_kernel(const float* w, T* fwd, T* tr, int n, int k) {
  __shared__ float tile[64][65];
  linear_pack_tile<T>(tile, w, fwd, tr, n, k, blockIdx.x * 64, blockIdx.y * 64);
}

#define LPM_MAX 64
struct LinearPackMulti {
  const float* w[LPM_MAX]; void* fwd[LPM_MAX]; void* tr[LPM_MAX];
  int n[LPM_MAX], k[LPM_MAX], tiles[LPM_MAX];
  int count;
};
template <typename T>
__global__ __launch_bounds__(256) void linear_pack_multi_kernel(LinearPackMulti a) {
  __shared__ float tile[64][65];
  int b = blockIdx.x, e = 0;
  for (; e < a.count; ++e) {
    if (b < a.tiles[e]) break;
    b -= a.tiles[e];
  }
  if (e >= a.count) return;
  const int tk = (a.k[e] + 63) / 64;
  linear_pack_tile<T>(tile, a.w[e], (T*)a.fwd[e], (T*)a.tr[e], a.n[e], a.k[e], (b % tk) * 64, (b / tk) * 64);
}

// maxpool 3x3 stride 2 pad 1 on tokens
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* in, long ldi, T* out, long ldo, unsigned char* arg,
                                                           int frames, int H, int W, int Ho, int Wo, int C) {
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)frames * Ho * Wo * ppr) return;
  const int c = (idx % ppr) * PACK;
  const long px = idx / ppr;
  const int x = px % Wo, y = (px / Wo) % Ho, f = px / ((long)Wo * Ho);
  float best[8]; unsigned char ba[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { best[e] = -3.0e38f; ba[e] = 255; }
  for (int t = 0; t < 9; ++t) {
    const int yy = 2 * y - 1 + t / 3, xx = 2 * x - 1 + t % 3;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    Vec16<T> v;
    v.v = *(const decltype(v.v)*)(in + (((long)f * H + yy) * W + xx) * ldi + c);
#pragma unroll
    for (int e = 0; e < PACK; ++e) {
      const float u = v.get(e);
      if (u > best[e] || ba[e] == 255) { best[e] = u; ba[e] = (unsigned char)t; }
    }
  }
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < PACK; ++e) { o.set(e, best[e]); arg[px * C + c + e] = ba[e]; }
  *(decltype(o.v)*)(out + px * ldo + c) = o.v;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* dout, long ldo, T* din, long ldi, const unsigned char* arg,
                                                           int frames, int H, int W, int Ho, int Wo, int C) {
  constexpr int PACK = TT<T>::PACK;
  const int ppr = C / PACK;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)frames * H * W * ppr) return;
  const int c = (idx % ppr) * PACK;
  const long px = idx / ppr;
  const int x = px % W, y = (px / W) % H, f = px / ((long)W * H);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // output pixel (yo, xo) reads (y, x) through tap (ty, tx) iff 2 yo - 1 + ty == y: an odd coordinate has the candidates ty = 0 and
  // ty = 2, an even one only ty = 1 - at most four output pixels, visited in tap order; their winning-tap bytes come in one load
  const int ny = (y & 1) ? 2 : 1, nx = (x & 1) ? 2 : 1;
  for (int iy = 0; iy < ny; ++iy) {
    const int ty = (y & 1) ? 2 * iy : 1, yo = (y + 1 - ty) >> 1;
    if (yo >= Ho) continue;
    for (int ix = 0; ix < nx; ++ix) {
      const int tx = (x & 1) ? 2 * ix : 1, xo = (x + 1 - tx) >> 1;
      if (xo >= Wo) continue;
      const unsigned t = (unsigned)(ty * 3 + tx);
      const long po = ((long)f * Ho + yo) * Wo + xo;
      Vec16<T> d;
      d.v = *(const decltype(d.v)*)(dout + po * ldo + c);
      unsigned a[2] = {0, 0};
      if (PACK == 8) { const uint2 w = *(const uint2*)(arg + po * C + c); a[0] = w.x; a[1] = w.y; }
      else a[0] = *(const unsigned*)(arg + po * C + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e)
        if (((a[e >> 2] >> (8 * (e & 3))) & 0xFFu) == t) acc[e] += d.get(e);
    }
  }
  Vec16<T> o;
#pragma unroll
  for (int e = 0; e < PACK; ++e) o.set(e, acc[e]);
  *(decltype(o.v)*)(din + px * ldi + c) = o.v;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm over rows that may be the concatenation of S gathered segments (patch merging: S = 4).
// One wave per row; the row lives in registers (C <= 4096); statistics two-pass in fp32 like torch.
// ---------------------------------------------------------------------------------------------------
template <typename T, int NP>   // NP = pieces of 16 B per lane
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* x, long ldx, const int* rows, int S, int Cseg, T* y,
                                                      long ldy, const float* gamma, const float* beta, float* mean,
                                                      float* rstd, int M, float eps) {
  constexpr int PACK = TT<T>::PACK;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (r >= M) return;
  const int C = S * Cseg;
  float v[NP][PACK];
  float sum = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = (p * 64 + l) * PACK;
    if (c < C) {
      const int s = c / Cseg, cc = c - s * Cseg;
      const long row = rows ? (long)rows[(long)s * M + r] : (long)r;
      Vec16<T> in;
      in.v = *(const decltype(in.v)*)(x + row * ldx + cc);
#pragma unroll
      for (int e = 0; e < PACK; ++e) { v[p][e] = in.get(e); sum += v[p][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < PACK; ++e) v[p][e] = 0.f;
    }
  }
  const float mu = wave_sum(sum) / C;
  float sq = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = (p * 64 + l) * PACK;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < PACK; ++e) { const float d = v[p][e] - mu; sq += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(sq) / C + eps);
  if (l == 0 && mean) { mean[r] = mu; rstd[r] = rs; }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int c = (p * 64 + l) * PACK;
    if (c < C) {
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < PACK; ++e) o.set(e, (v[p][e] - mu) * rs * gamma[c + e] + beta[c + e]);
      *(decltype(o.v)*)(y + (long)r * ldy + c) = o.v;
    }
  }
}

// ---- deterministic cross-workgroup sums ---------------------------------------------------------------------------------
// Every kernel of this library that sums over workgroups (bias / LayerNorm / BatchNorm parameter gradients, BatchNorm
// statistics, the relative-position-bias gradient) writes ONE partial vector per workgroup with plain stores ("slab" p of a
// caller-owned scratch buffer) and this kernel adds the slabs in a fixed order behind the launch boundary: no fp32 atomics,
// so the results do not depend on the order in which workgroups happen to finish - two runs of a training step give the same
// bits.  (Same-address fp32 atomics also serialise at ~40 ns each.)
//   out_s[b * obs_s + j] (+)= sum_{p < nslabs} ws[b * ws_batch_stride + p * slab_stride + off_s + j],  j < len_s, off_s = len_0 + .. + len_{s-1}
// for up to three output vectors s (lengths multiples of 4).  Block: 16 float4 columns x 16 slab lanes; lane q adds the slabs
// q, q + 16, ... and the 16 lane sums are added in lane order.
struct FoldArgs {
  const float* ws; long slab_stride, ws_batch_stride; int nslabs;
  int len[3]; float* out[3]; long obs[3]; int accumulate, cl;
};
// cl = float4 columns per workgroup (16: few slabs, 64 floats per workgroup; 4: many slabs - 64 slab lanes per column, four times
// the workgroups); the association of the sum is fixed by (nslabs, cl) alone.
DEVI void slab_fold_body(const FoldArgs& f, int block_x, int batch) {
  __shared__ f32x4 fold[256];
  const int CL = f.cl, QL = 256 / CL;
  const int cl = threadIdx.x % CL, q = threadIdx.x / CL;
  const int j = (block_x * CL + cl) * 4;
  const bool in = j < f.len[0] + f.len[1] + f.len[2];
  const float* src = f.ws + (long)batch * f.ws_batch_stride + j;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (in) {
    int p = q;
    for (; p + 3 * QL < f.nslabs; p += 4 * QL) {             // four independent loads in flight per lane
      const f32x4 v0 = *(const f32x4*)(src + (long)p * f.slab_stride), v1 = *(const f32x4*)(src + (long)(p + QL) * f.slab_stride);
      const f32x4 v2 = *(const f32x4*)(src + (long)(p + 2 * QL) * f.slab_stride), v3 = *(const f32x4*)(src + (long)(p + 3 * QL) * f.slab_stride);
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; p < f.nslabs; p += QL) a += *(const f32x4*)(src + (long)p * f.slab_stride);
  }
  fold[q * CL + cl] = a;
  __syncthreads();
  if (q == 0 && in) {
    for (int k = 1; k < QL; ++k) a += fold[k * CL + cl];
    const int sgm = j < f.len[0] ? 0 : (j < f.len[0] + f.len[1] ? 1 : 2);
    const int jj = j - (sgm > 0 ? f.len[0] : 0) - (sgm > 1 ? f.len[1] : 0);
    float* out = f.out[sgm];
    if (out) {
      f32x4* dst = (f32x4*)(out + (long)batch * f.obs[sgm] + jj);
      *dst = f.accumulate ? *dst + a : a;
    }
  }
}
__global__ __launch_bounds__(256) void slab_fold_kernel(FoldArgs f) { slab_fold_body(f, blockIdx.x, blockIdx.y); }

// Several folds in ONE launch (stswin_fold_defer / stswin_fold_flush): a backward pass of a Swin block queues the folds of its two
// LayerNorm backward passes, its bias-gradient tables and the attention kernel's slabs - none of their results is read before the
// pass ends - and launches them together (4 launches of ~5 us less per block and step).
#define FOLD_QUEUE_MAX 12
struct FoldBatch { FoldArgs f[FOLD_QUEUE_MAX]; int first_block[FOLD_QUEUE_MAX + 1]; int blocks_x[FOLD_QUEUE_MAX]; int count; };
__global__ __launch_bounds__(256) void slab_fold_multi_kernel(FoldBatch b) {
  int i = 0;
  while (i + 1 < b.count && (int)blockIdx.x >= b.first_block[i + 1]) ++i;
  const int local = blockIdx.x - b.first_block[i];
  slab_fold_body(b.f[i], local % b.blocks_x[i], local / b.blocks_x[i]);
}
static thread_local int g_fold_defer = 0;
static thread_local FoldBatch g_fold_batch;

static int fold_flush(hipStream_t st) {
  FoldBatch& b = g_fold_batch;
  if (b.count > 0) {
    hipLaunchKernelGGL(slab_fold_multi_kernel, dim3((unsigned)b.first_block[b.count]), dim3(256), 0, st, b);
    b.count = 0;
  }
  return 0;
}
extern "C" int stswin_fold_defer(int on, void* stream) {          // on = 0 flushes what is queued
  if (!on) { fold_flush((hipStream_t)stream); g_fold_defer = 0; STSWIN_CHECK_LAUNCH(); return 0; }
  g_fold_defer = 1;
  g_fold_batch.count = 0;
  return 0;
}
extern "C" int stswin_fold_flush(void* stream) {
  fold_flush((hipStream_t)stream);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

int stswin_fold3_launch(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, const int* len, float* const* out,
                        const long* obs, int batch, int accumulate, hipStream_t st) {
  FoldArgs f;
  f.ws = ws; f.slab_stride = slab_stride; f.ws_batch_stride = ws_batch_stride; f.nslabs = nslabs; f.accumulate = accumulate;
  int tot = 0;
  for (int i = 0; i < 3; ++i) {
    f.len[i] = len[i]; f.out[i] = out[i]; f.obs[i] = obs[i];
    if (len[i] < 0 || len[i] % 4 || obs[i] % 4) return -1110;
    tot += len[i];
  }
  if (nslabs <= 0 || tot <= 0 || batch <= 0) return 0;
  if (slab_stride % 4 || ws_batch_stride % 4) return -1110;
  f.cl = nslabs >= 96 ? 4 : 16;
  const int bx = (tot + 4 * f.cl - 1) / (4 * f.cl);
  if (g_fold_defer) {
    FoldBatch& b = g_fold_batch;
    if (b.count == FOLD_QUEUE_MAX) fold_flush(st);
    if (b.count == 0) b.first_block[0] = 0;
    b.f[b.count] = f; b.blocks_x[b.count] = bx;
    b.first_block[b.count + 1] = b.first_block[b.count] + bx * batch;
    ++b.count;
    return 0;
  }
  hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)bx, (unsigned)batch), dim3(256), 0, st, f);
  return 0;
}

int stswin_fold_launch(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, int seg_len, int nseg, float* o0, float* o1,
                       float* o2, long out_batch_stride, int batch, int accumulate, hipStream_t st) {
  if (nseg <= 0 || nseg > 3 || seg_len <= 0) return nseg > 3 ? -1110 : 0;
  const int len[3] = {seg_len, nseg > 1 ? seg_len : 0, nseg > 2 ? seg_len : 0};
  float* const out[3] = {o0, o1, o2};
  const long obs[3] = {out_batch_stride, out_batch_stride, out_batch_stride};
  return stswin_fold3_launch(ws, slab_stride, ws_batch_stride, nslabs, len, out, obs, batch, accumulate, st);
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma += sum dy*xhat ; dbeta += sum dy.
// Each wave walks `rows_per_wave` consecutive rows keeping its dgamma/dbeta columns in registers, the block
// folds its NW waves through LDS and stores one partial row per workgroup (summed by slab_fold_kernel).
// Round 5 (profiles/r05_layernorm_kernels.txt): ALL pieces of a row - x, dy, the old dx when accumulating, mean, rstd - are
// requested before the first one is used (inside the per-piece `if (c < C)` bodies hipcc kept each piece's loads behind the
// previous piece's arithmetic: one round trip per piece), gamma is read from LDS (as global loads its pieces were re-requested for
// every row), "accumulate" and "column sums of dx" are template switches (C = 2048 had spilled), and workgroups are 8 waves where
// four would leave the grid at <= 512 workgroups.  C = 1024: 35.5 -> 30.9 us, C = 2048 gathered: 113 -> 70 us; C = 512 unchanged
// (one piece per row; ~200 VALU instructions per row and wave make that launch as much VALU- as HBM-bound).  Measured and NOT
// kept: the next row prefetched behind the current one, 2 or 4 rows of a wave in flight together (138-186 registers, C = 512
// 50 -> 68 us), g / xhat forcibly re-derived in the second pass (more VALU work), fused multiply-adds (no change).
template <typename T, int NP, int NW, bool ACC, bool DXS>
__global__ __launch_bounds__(NW * 64) void ln_bwd_kernel(const T* dy, long lddy, const T* x, long ldx, const int* rows,
                                                          int S, int Cseg, const float* gamma, const float* mean,
                                                          const float* rstd, T* dx, long lddx, float* dgamma,
                                                          float* dbeta, int M, int rows_per_wave, float* ws,
                                                          const T* add, long ldadd) {      // ACC: dx = add + LN'(dy) (add may be dx itself)
  constexpr int PACK = TT<T>::PACK;
  typedef decltype(Vec16<T>().v) vec_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int C = S * Cseg;
  float dg[NP][PACK], db[NP][PACK], ds[DXS ? NP : 1][PACK];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < PACK; ++e) { dg[p][e] = 0.f; db[p][e] = 0.f; if (DXS) ds[p][e] = 0.f; }
  float* sgam = (float*)smem + NW * C;
  for (int c = threadIdx.x; c < C; c += NW * 64) sgam[c] = gamma[c];
  __syncthreads();
  // workgroups walk row groups blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher caps the grid at 1024: the parameter-gradient
  // partial sums stay in registers across a workgroup's groups, so there is one slab per workgroup to fold afterwards)
  typedef f32x4 raw_t;                 // 16 bytes of a row as a register quad
  constexpr bool EARLY = ACC && NP <= 4;      // (NP = 8: the row's own pieces fill the register file)
  struct Row { raw_t xi[NP], di[NP], old[EARLY ? NP : 1]; float mu, rs; };
  for (int rg = blockIdx.x; rg * NW * rows_per_wave < M; rg += gridDim.x) {
  const int r_begin = (rg * NW + w) * rows_per_wave;
  for (int r = r_begin; r < min(M, r_begin + rows_per_wave); ++r) {
    // every piece of the row is requested up front - the old dx, too, when accumulating (requested where it is used it cost a
    // second round trip per row)
    Row cur;
    cur.mu = mean[r]; cur.rs = rstd[r];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * 64 + l) * PACK;
      if (c < C) {
        const int s = c / Cseg, cc = c - s * Cseg;
        const long row = rows ? (long)rows[(long)s * M + r] : (long)r;
        cur.xi[p] = *(const raw_t*)(x + row * ldx + cc);
        cur.di[p] = *(const raw_t*)(dy + (long)r * lddy + c);
        if constexpr (EARLY) cur.old[p] = *(const raw_t*)(add + row * ldadd + cc);
      }
    }
    const float mu = cur.mu, rs = cur.rs;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * 64 + l) * PACK;
      if (c < C) {
        Vec16<T> xi, di;
        xi.v = __builtin_bit_cast(vec_t, cur.xi[p]); di.v = __builtin_bit_cast(vec_t, cur.di[p]);
        float gm[PACK];
#pragma unroll
        for (int e = 0; e < PACK; e += 4) *(f32x4*)(gm + e) = *(const f32x4*)(sgam + c + e);
#pragma unroll
        for (int e = 0; e < PACK; ++e) {
          const float d = di.get(e);
          const float xh = (xi.get(e) - mu) * rs;
          const float g = d * gm[e];
          s1 += g;
          s2 += g * xh;
          dg[p][e] += d * xh;
          db[p][e] += d;
        }
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
    // (g and xhat are written out again from the row's pieces; hipcc decides per instantiation whether to keep the first pass's
    // floats or to re-derive them)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * 64 + l) * PACK;
      if (c < C) {
        const int s = c / Cseg, cc = c - s * Cseg;
        const long row = rows ? (long)rows[(long)s * M + r] : (long)r;
        T* dst = dx + row * lddx + cc;
        Vec16<T> xi, di, o;
        xi.v = __builtin_bit_cast(vec_t, cur.xi[p]); di.v = __builtin_bit_cast(vec_t, cur.di[p]);
        if constexpr (EARLY) o.v = __builtin_bit_cast(vec_t, cur.old[p]);
        else if constexpr (ACC) o.v = *(const vec_t*)(add + row * ldadd + cc);
        float gm[PACK];
#pragma unroll
        for (int e = 0; e < PACK; e += 4) *(f32x4*)(gm + e) = *(const f32x4*)(sgam + c + e);
#pragma unroll
        for (int e = 0; e < PACK; ++e) {
          const float xh = (xi.get(e) - mu) * rs;
          const float g = di.get(e) * gm[e];
          const float val = rs * (g - s1 - xh * s2);
          const float fin = ACC ? o.get(e) + val : val;
          o.set(e, fin);
          if constexpr (DXS) ds[p][e] += fin;
        }
        *(vec_t*)dst = o.v;
      }
    }
  }
  }
  // fold the NW waves through ONE [NW][C] LDS table, a vector at a time; slab `blockIdx.x` of the caller's scratch:
  // [3][C] = dgamma | dbeta | dxsum partial sums of this workgroup's rows (plain stores; slab_fold_kernel adds the slabs in order
  // afterwards: deterministic, and no same-address atomics)
  float* sg = (float*)smem;
  float* rep = ws + (long)blockIdx.x * 3 * C;
  auto fold = [&](const float (&acc)[NP][PACK], float* out) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int c = (p * 64 + l) * PACK;
      if (c < C) {
#pragma unroll
        for (int e = 0; e < PACK; ++e) sg[w * C + c + e] = acc[p][e];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NW * 64) {
      float t = sg[c];
#pragma unroll
      for (int k = 1; k < NW; ++k) t += sg[k * C + c];
      out[c] = t;
    }
    __syncthreads();
  };
  fold(dg, rep);
  fold(db, rep + C);
  if constexpr (DXS) fold(ds, rep + 2 * C);     // column sums of the written dx (= the bias gradient of the Linear that produced x)
}

// part[blockIdx.y][n] = sum of rows [blockIdx.y * rows_per_block, ...) of Y[:, n]   (bias gradients; slab_fold_kernel adds the
// row-block partials into out).  Block = 256 threads: 32 column-pieces x 8 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* y, long ldy, float* part, int M, int N, int rows_per_block) {
  constexpr int PACK = TT<T>::PACK;
  __shared__ float red[8][32 * 8];
  const int cp = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + cp) * PACK;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int r0 = blockIdx.y * rows_per_block;
  if (c < N) {
    for (int r = r0 + rl; r < min(M, r0 + rows_per_block); r += 8) {
      Vec16<T> in;
      in.v = *(const decltype(in.v)*)(y + (long)r * ldy + c);
#pragma unroll
      for (int e = 0; e < PACK; ++e) acc[e] += in.get(e);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][cp * 8 + e] = acc[e];
  __syncthreads();
  if (rl == 0 && c < N) {
#pragma unroll
    for (int e = 0; e < PACK; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k][cp * 8 + e];
      part[(long)blockIdx.y * N + c + e] = t;
    }
  }
}

// out[i] = map[i] >= 0 ? v[map[i]] : fill   (channel-padding / un-padding of per-channel vectors: one launch instead of a
// fill plus one copy per layout segment)
__global__ __launch_bounds__(256) void vec_gather_kernel(const float* v, const int* map, float* out, int n, float fill) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const int j = map[i]; out[i] = j >= 0 ? v[j] : fill; }
}

// the same for up to four vectors that share the map (a BatchNorm's weight | bias | running mean | running variance, or its two
// parameter gradients): one launch instead of four, blockIdx.y = vector
struct VecGatherMulti { const float* v[4]; float* out[4]; float fill[4]; const int* map; int n; };
__global__ __launch_bounds__(256) void vec_gather_multi_kernel(VecGatherMulti p) {
  const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (i < p.n) { const int j = p.map[i]; p.out[k][i] = j >= 0 ? p.v[k][j] : p.fill[k]; }
}

// ---- relative position bias (swin_512.py:122-131): table[index] -> the attention kernels' [key][query] layout, with the
// SW-MSA mask folded in per window, and the transposed scatter of its gradient.  One launch each instead of the
// index / permute / contiguous / add chain (6 tiny kernels per block forward, index_add_ + permute per backward).
__global__ __launch_bounds__(256) void bias_expand_kernel(const float* table, const long* index, const float* mask, float* out,
                                                          int N, int heads, int nW) {
  const long n = (long)nW * heads * N * N;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int i = (int)(t % N), j = (int)((t / N) % N);          // out[w][h][key j][query i]
    const int h = (int)((t / ((long)N * N)) % heads), w = (int)(t / ((long)N * N * heads));
    float v = table[index[(long)i * N + j] * heads + h];
    if (mask) v += mask[((long)w * N + i) * N + j];
    out[t] = v;
  }
}
// the same for up to 16 tables at once (every Swin block's table after an optimizer step: ops.repack), blockIdx.y = table
#define BEM_MAX 16
struct BiasExpandMulti { const float* table[BEM_MAX]; const long* index[BEM_MAX]; const float* mask[BEM_MAX]; float* out[BEM_MAX];
                         int N[BEM_MAX], heads[BEM_MAX], nW[BEM_MAX]; };
__global__ __launch_bounds__(256) void bias_expand_multi_kernel(BiasExpandMulti p) {
  const int e = blockIdx.y, N = p.N[e], heads = p.heads[e];
  const float* table = p.table[e]; const long* index = p.index[e]; const float* mask = p.mask[e]; float* out = p.out[e];
  const long n = (long)p.nW[e] * heads * N * N;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
    const int i = (int)(t % N), j = (int)((t / N) % N);
    const int h = (int)((t / ((long)N * N)) % heads), w = (int)(t / ((long)N * N * heads));
    float v = table[index[(long)i * N + j] * heads + h];
    if (mask) v += mask[((long)w * N + i) * N + j];
    out[t] = v;
  }
}
// Gradient of the gather, in gather form (deterministic: no atomics).  `order` lists the (query i, key j) pairs (as i * N + j) sorted
// by their table row, `offs[e] .. offs[e + 1]` is row e's range (both built once per relative_position_index buffer by the
// caller): thread (e, h) adds its <= N pairs in list order.  nslabs > 1: dbiasT is [nslabs][heads][N][N] and the slabs are added on
// the way.
__global__ __launch_bounds__(256) void bias_scatter_kernel(const float* dbiasT, const int* order, const int* offs, float* dtable, int N,
                                                            int heads, int table_rows, int nslabs) {
  // one WAVE per (table row e, head h): lane q adds the pairs q, q + 64, ... of the row (<= N of them for a real index), then a fixed
  // butterfly over the lanes - the per-thread loop of dependent (order -> value) loads was latency-bound (17 us for 900 threads)
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (t >= table_rows * heads) return;
  const int e = t / heads, h = t - e * heads;
  const long NN = (long)N * N;
  float acc = 0.f;
  for (int q = offs[e] + l; q < offs[e + 1]; q += 64) {
    const int pr = order[q], i = pr / N, j = pr - i * N;
    const float* src = dbiasT + ((long)h * N + j) * N + i;        // dbiasT[h][key j][query i]
    for (int sl = 0; sl < nslabs; ++sl) acc += src[(long)sl * heads * NN];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if (l == 0) dtable[(long)e * heads + h] += acc;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int stswin_win_rowmap(int* map, int B, int T, int H, int W, int ws, int shift, int f0, int frames_total,
                                 void* stream) {
  if (H % ws || W % ws || shift < 0 || shift >= ws) return -1101;
  const int rows = B * T * H * W;
  hipLaunchKernelGGL(win_rowmap_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, map, rows, T, H, W,
                     ws, shift, f0, frames_total);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_win_move(int dtype, const void* in, void* out, int B, int T, int H, int W, int C, int ws, int shift,
                               int f0, int frames_total, int dir, void* stream) {
  if (H % ws || W % ws || shift < 0 || shift >= ws) return -1101;
  if (C % (dtype == 0 ? 8 : 4)) return -1102;
  const int rows = B * T * H * W;
  if (dtype == 0)
    hipLaunchKernelGGL(win_move_kernel<bf16>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16*)in,
                       (bf16*)out, rows, C, T, H, W, ws, shift, f0, frames_total, dir);
  else
    hipLaunchKernelGGL(win_move_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)in,
                       (float*)out, rows, C, T, H, W, ws, shift, f0, frames_total, dir);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_merge_rowmap(int* map, int frames, int H, int W, void* stream) {
  if ((H | W) & 1) return -1103;
  const int M = frames * (H / 2) * (W / 2);
  hipLaunchKernelGGL(merge_rowmap_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, map, frames, H, W);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_conv3x3_rowmap(int* map, int frames, int H, int W, int dilation, void* stream) {
  const int M = frames * H * W;
  hipLaunchKernelGGL(conv3x3_rowmap_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, map, frames, H, W,
                     dilation);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_conv_rowmap(int* map, int frames, int Hin, int Win, int Hout, int Wout, int k, int stride, int pad,
                                  int dilation, int inverse, void* stream) {
  if (k <= 0 || stride <= 0) return -1108;
  const int n = inverse ? frames * Hin * Win : frames * Hout * Wout;
  hipLaunchKernelGGL(conv_rowmap_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, map, frames, Hin, Win,
                     Hout, Wout, k, stride, pad, dilation, inverse);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_stem_im2col(int dtype, const float* img, void* patches, long ld, int frames, int H, int W, int Ho,
                                  int Wo, void* stream) {
  if (ld < 147 || ld > 192) return -1109;
  const int pk_ = dtype == 0 ? 8 : 4;
  if (ld % pk_) return -1109;
  const int ppb = 256 / (int)(ld / pk_), TP = 4 * ppb < 40 ? 4 * ppb : 40;        // must match the kernel
  const long want = (long)frames * Ho * ((Wo + TP - 1) / TP);
  if (want > 0x7fffffffL) return -1109;
  dim3 grid((unsigned)want);
  if (dtype == 0) hipLaunchKernelGGL(stem_im2col_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, img, (bf16*)patches, ld, frames, H, W, Ho, Wo);
  else hipLaunchKernelGGL(stem_im2col_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, (float*)patches, ld, frames, H, W, Ho, Wo);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_stem_s2d(int dtype, const float* img, void* out, int frames, int H, int W, void* stream) {
  if (frames <= 0 || H <= 0 || W <= 0) return -1112;
  const int Hs = (H - 1) / 2 + 1 + 3, Ws = (W - 1) / 2 + 1 + 3;
  const long total = (long)frames * Hs * Ws;
  if (total > 0x7fffffffL * 64) return -1112;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == 0) hipLaunchKernelGGL(stem_s2d_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, img, (bf16*)out, frames, H, W, Hs, Ws);
  else hipLaunchKernelGGL(stem_s2d_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img, (float*)out, frames, H, W, Hs, Ws);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_conv_pack(int dtype, const float* w, void* fwd, void* dgrad, const int* omap, const int* imap, int co,
                                int ci, int S, int cop, int cip, void* stream) {
  if (co <= 0 || ci <= 0 || S <= 0 || cop <= 0 || cip <= 0) return -1111;
  if (S > 65535) return -1111;
  dim3 grid((unsigned)((cip + 63) / 64), (unsigned)((cop + 63) / 64), (unsigned)S);
  if (dtype == 0) hipLaunchKernelGGL(conv_pack_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16*)fwd, (bf16*)dgrad, omap, imap, co, ci, S, cop, cip);
  else hipLaunchKernelGGL(conv_pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, (float*)fwd, (float*)dgrad, omap, imap, co, ci, S, cop, cip);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_maxpool3x3s2(int dtype, const void* in, long ldi, void* out, long ldo, unsigned char* arg, int frames,
                                   int H, int W, int Ho, int Wo, int C, int backward, void* stream) {
  const int pk = dtype == 0 ? 8 : 4;
  if (C % pk || ldi % pk || ldo % pk) return -1110;
  hipStream_t st = (hipStream_t)stream;
  if (!backward) {
    const long n = (long)frames * Ho * Wo * (C / pk);
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, ldi, (bf16*)out, ldo, arg, frames, H, W, Ho, Wo, C);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, arg, frames, H, W, Ho, Wo, C);
  } else {   /* in = d(out) [F*Ho*Wo][C] (pitch ldi), out = d(in) [F*H*W][C] (pitch ldo) */
    const long n = (long)frames * H * W * (C / pk);
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == 0) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)in, ldi, (bf16*)out, ldo, arg, frames, H, W, Ho, Wo, C);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, arg, frames, H, W, Ho, Wo, C);
  }
  STSWIN_CHECK_LAUNCH();
  return 0;
}

template <typename T>
static int ln_fwd_launch(const void* x, long ldx, const int* rows, int S, int Cseg, void* y, long ldy, const float* g,
                         const float* b, float* mean, float* rstd, int M, float eps, hipStream_t st) {
  constexpr int PACK = TT<T>::PACK;
  const int C = S * Cseg, np = (C / PACK + 63) / 64;
  dim3 grid((M + 3) / 4), blk(256);
#define LN_F(NP) hipLaunchKernelGGL((ln_fwd_kernel<T, NP>), grid, blk, 0, st, (const T*)x, ldx, rows, S, Cseg, (T*)y, ldy, g, b, mean, rstd, M, eps)
  switch (np) {
    case 1: LN_F(1); break;
    case 2: LN_F(2); break;
    case 3: case 4: LN_F(4); break;
    case 5: case 6: case 7: case 8: LN_F(8); break;
    default:
      if (np <= 16) { LN_F(16); break; }
      return -1104;
  }
#undef LN_F
  return 0;
}

extern "C" int stswin_layernorm_fwd(int dtype, const void* x, long ldx, const int* rows, int S, int Cseg, void* y,
                                    long ldy, const float* gamma, const float* beta, float* mean, float* rstd, int M,
                                    float eps, void* stream) {
  const int pack = dtype == 0 ? 8 : 4;
  if (Cseg % pack || ldx % pack || ldy % pack) return -1105;
  int rc = dtype == 0 ? ln_fwd_launch<bf16>(x, ldx, rows, S, Cseg, y, ldy, gamma, beta, mean, rstd, M, eps, (hipStream_t)stream)
                      : ln_fwd_launch<float>(x, ldx, rows, S, Cseg, y, ldy, gamma, beta, mean, rstd, M, eps, (hipStream_t)stream);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

// Geometry: 8 rows per wave (M >= 16384); workgroups of 8 waves when four would leave the grid at <= 512 workgroups (C >= 1024 in
// the step: M = 16384), so that 4 waves per SIMD are resident without more slabs to fold; grid capped at 1024.
static int ln_bwd_rows_per_wave(int M) { return M >= 16384 ? 8 : (M >= 4096 ? 4 : 2); }
static int ln_bwd_waves(int M, int C, int pack) {
  const int np = (C / pack + 63) / 64;
  static const char* e = getenv("STSWIN_LN_BWD_WAVES");     // A/B switch (tools/bench_ln.py): 4 or 8
  if (e && (atoi(e) == 4 || atoi(e) == 8)) return np <= 2 ? atoi(e) : 4;
  const int rpw = ln_bwd_rows_per_wave(M);
  return (M + 4 * rpw - 1) / (4 * rpw) <= 512 && M >= 8 * rpw * 128 && np <= 2 ? 8 : 4;
}
static int ln_bwd_grid(int M, int C, int pack) {
  const int rpw = ln_bwd_rows_per_wave(M), nw = ln_bwd_waves(M, C, pack), groups = (M + nw * rpw - 1) / (nw * rpw);
  return groups < 1024 ? groups : 1024;
}
extern "C" long stswin_layernorm_bwd_scratch(int M, int C) {      // (sized for the 4-wave geometry: the larger of the two)
  const int rpw = ln_bwd_rows_per_wave(M), groups = (M + 4 * rpw - 1) / (4 * rpw);
  return (long)(groups < 1024 ? groups : 1024) * 3 * C;
}

template <typename T>
static int ln_bwd_launch(const void* dy, long lddy, const void* x, long ldx, const int* rows, int S, int Cseg,
                         const float* g, const float* mean, const float* rstd, void* dx, long lddx, float* dg, float* db,
                         int M, int acc, float* dxsum, float* ws, hipStream_t st, const void* add = nullptr, long ldadd = 0) {
  if (acc && !add) { add = dx; ldadd = lddx; }
  constexpr int PACK = TT<T>::PACK;
  const int C = S * Cseg, np = (C / PACK + 63) / 64;
  const int rpw = ln_bwd_rows_per_wave(M), nw = ln_bwd_waves(M, C, PACK);
  dim3 grid((unsigned)ln_bwd_grid(M, C, PACK)), blk(nw * 64);
  const size_t lds = (size_t)(nw + 1) * C * sizeof(float);
  if (!ws) return -1111;
#define LN_B4(NP, NW, A, D) hipLaunchKernelGGL((ln_bwd_kernel<T, NP, NW, A, D>), grid, blk, lds, st, (const T*)dy, lddy, (const T*)x, ldx, rows, S, Cseg, g, mean, rstd, (T*)dx, lddx, dg, db, M, rpw, ws, (const T*)add, ldadd)
#define LN_B2(NP, NW) do { if (acc) { if (dxsum) LN_B4(NP, NW, true, true); else LN_B4(NP, NW, true, false); } \
                           else { if (dxsum) LN_B4(NP, NW, false, true); else LN_B4(NP, NW, false, false); } } while (0)
#define LN_B(NP) do { if (nw == 8) LN_B2(NP, 8); else LN_B2(NP, 4); } while (0)
#define LN_BW4(NP) LN_B2(NP, 4)
  switch (np) {
    case 1: LN_B(1); break;
    case 2: LN_B(2); break;
    case 3: case 4: LN_BW4(4); break;
    case 5: case 6: case 7: case 8: LN_BW4(8); break;
    default: return -1104;
  }
#undef LN_B
#undef LN_BW4
#undef LN_B2
#undef LN_B4
  return stswin_fold_launch(ws, 3L * C, 0, (int)grid.x, C, dxsum ? 3 : 2, dg, db, dxsum, 0, 1, 1, st);
}

extern "C" int stswin_layernorm_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, const int* rows, int S,
                                    int Cseg, const float* gamma, const float* mean, const float* rstd, void* dx,
                                    long lddx, float* dgamma, float* dbeta, int M, int accumulate_dx, float* dxsum,
                                    float* workspace, void* stream) {
  const int pack = dtype == 0 ? 8 : 4;
  if (Cseg % pack || ldx % pack || lddy % pack || lddx % pack) return -1105;
  if ((long)S * Cseg * 5 * 4 > 65536) return -1106;
  int rc = dtype == 0 ? ln_bwd_launch<bf16>(dy, lddy, x, ldx, rows, S, Cseg, gamma, mean, rstd, dx, lddx, dgamma, dbeta, M, accumulate_dx, dxsum, workspace, (hipStream_t)stream)
                      : ln_bwd_launch<float>(dy, lddy, x, ldx, rows, S, Cseg, gamma, mean, rstd, dx, lddx, dgamma, dbeta, M, accumulate_dx, dxsum, workspace, (hipStream_t)stream);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

/* dx = add + LN'(dy): the residual sum of a pre-norm block's backward written to a NEW buffer, so that `add` (the gradient of the block's
 * output branch, an operand of a weight-gradient GEMM that is launched later: stswin_gemm_tn_group) stays intact */
extern "C" int stswin_layernorm_bwd_add(int dtype, const void* dy, long lddy, const void* x, long ldx, const int* rows, int S,
                                        int Cseg, const float* gamma, const float* mean, const float* rstd, const void* add, long ldadd,
                                        void* dx, long lddx, float* dgamma, float* dbeta, int M, float* dxsum, float* workspace,
                                        void* stream) {
  const int pack = dtype == 0 ? 8 : 4;
  if (!add || Cseg % pack || ldx % pack || lddy % pack || lddx % pack || ldadd % pack) return -1105;
  if ((long)S * Cseg * 5 * 4 > 65536) return -1106;
  int rc = dtype == 0 ? ln_bwd_launch<bf16>(dy, lddy, x, ldx, rows, S, Cseg, gamma, mean, rstd, dx, lddx, dgamma, dbeta, M, 1, dxsum, workspace, (hipStream_t)stream, add, ldadd)
                      : ln_bwd_launch<float>(dy, lddy, x, ldx, rows, S, Cseg, gamma, mean, rstd, dx, lddx, dgamma, dbeta, M, 1, dxsum, workspace, (hipStream_t)stream, add, ldadd);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" long stswin_colsum_scratch(int M, int N) { return (long)((M + 511) / 512) * N; }
extern "C" int stswin_colsum(int dtype, const void* y, long ldy, float* out, int M, int N, float* scratch, void* stream) {
  const int pack = dtype == 0 ? 8 : 4;
  if (N % pack || ldy % pack) return -1107;
  if (!scratch) return -1111;
  if (M <= 0) return 0;
  const int rpb = 512;
  dim3 grid((N / pack + 31) / 32, (M + rpb - 1) / rpb);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)y, ldy, scratch, M, N, rpb);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)y, ldy, scratch, M, N, rpb);
  const int rc = stswin_fold_launch(scratch, N, 0, (int)grid.y, N, 1, out, nullptr, nullptr, 0, 1, 1, st);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_slab_fold(const float* ws, long slab_stride, long ws_batch_stride, int nslabs, int seg_len, int nseg, float* out0,
                                float* out1, float* out2, long out_batch_stride, int batch, int accumulate, void* stream) {
  const int rc = stswin_fold_launch(ws, slab_stride, ws_batch_stride, nslabs, seg_len, nseg, out0, out1, out2, out_batch_stride, batch,
                                    accumulate, (hipStream_t)stream);
  if (rc) return rc;
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bias_expand(const float* table, const long* index, const float* mask, float* out, int N, int heads, int nW,
                                  void* stream) {
  if (N <= 0 || heads <= 0 || nW <= 0) return -1108;
  const long n = (long)nW * heads * N * N;
  hipLaunchKernelGGL(bias_expand_kernel, dim3((unsigned)min(2048L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table,
                     index, mask, out, N, heads, nW);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bias_expand_multi(int count, const float* const* table, const long* const* index, const float* const* mask,
                                        float* const* out, const int* N, const int* heads, const int* nW, void* stream) {
  if (count <= 0) return 0;
  if (count > BEM_MAX) return -1108;
  BiasExpandMulti a;
  long most = 0;
  for (int e = 0; e < BEM_MAX; ++e) {
    const int s = e < count ? e : 0;
    if (N[s] <= 0 || heads[s] <= 0 || nW[s] <= 0) return -1108;
    a.table[e] = table[s]; a.index[e] = index[s]; a.mask[e] = mask[s]; a.out[e] = out[s]; a.N[e] = N[s]; a.heads[e] = heads[s]; a.nW[e] = nW[s];
    const long n = (long)nW[s] * heads[s] * N[s] * N[s];
    if (n > most) most = n;
  }
  hipLaunchKernelGGL(bias_expand_multi_kernel, dim3((unsigned)min(256L, (most + 255) / 256), (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_bias_scatter(const float* dbiasT, const int* order, const int* offs, float* dtable, int N, int heads, int table_rows,
                                   int nslabs, void* stream) {
  if (N <= 0 || heads <= 0 || table_rows <= 0 || nslabs <= 0 || !order || !offs) return -1109;
  hipLaunchKernelGGL(bias_scatter_kernel, dim3((unsigned)((table_rows * heads + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dbiasT,
                     order, offs, dtable, N, heads, table_rows, nslabs);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_vec_gather(const float* v, const int* map, float* out, int n, float fill, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(vec_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, v, map, out, n, fill);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_vec_gather_multi(int count, const float* const* v, const int* map, float* const* out, int n, const float* fill,
                                       void* stream) {
  if (n <= 0 || count <= 0) return 0;
  if (count > 4) return -1114;
  VecGatherMulti a;
  for (int k = 0; k < 4; ++k) { a.v[k] = k < count ? v[k] : nullptr; a.out[k] = k < count ? out[k] : nullptr; a.fill[k] = k < count ? fill[k] : 0.f; }
  a.map = map; a.n = n;
  hipLaunchKernelGGL(vec_gather_multi_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)count), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_linear_pack_multi(int dtype, int count, const float* const* w, void* const* fwd, void* const* tr, const int* n,
                                        const int* k, void* stream) {
  if (count <= 0) return 0;
  if (count > LPM_MAX) return -1113;
  LinearPackMulti a;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (n[i] <= 0 || k[i] <= 0 || n[i] % 4 || k[i] % 4) return -1112;
    a.w[i] = w[i]; a.fwd[i] = fwd[i]; a.tr[i] = tr ? tr[i] : nullptr; a.n[i] = n[i]; a.k[i] = k[i];
    a.tiles[i] = ((k[i] + 63) / 64) * ((n[i] + 63) / 64);
    blocks += a.tiles[i];
  }
  a.count = count;
  if (dtype == 0) hipLaunchKernelGGL(linear_pack_multi_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(linear_pack_multi_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_conv_pack_multi(int dtype, int count, const float* const* w, void* const* fwd, void* const* dgrad,
                                      const int* const* omap, const int* const* imap, const int* ci, const int* S, const int* cop,
                                      const int* cip, void* stream) {
  if (count <= 0) return 0;
  if (count > CPM_MAX) return -1113;
  ConvPackMulti a;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (ci[i] <= 0 || S[i] <= 0 || cop[i] <= 0 || cip[i] <= 0) return -1111;
    a.w[i] = w[i]; a.fwd[i] = fwd[i]; a.dg[i] = dgrad ? dgrad[i] : nullptr; a.omap[i] = omap[i]; a.imap[i] = imap[i];
    a.ci[i] = ci[i]; a.S[i] = S[i]; a.cop[i] = cop[i]; a.cip[i] = cip[i];
    a.tiles[i] = ((cip[i] + 63) / 64) * ((cop[i] + 63) / 64) * S[i];
    blocks += a.tiles[i];
  }
  a.count = count;
  if (dtype == 0) hipLaunchKernelGGL(conv_pack_multi_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_pack_multi_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  STSWIN_CHECK_LAUNCH();
  return 0;
}

extern "C" int stswin_linear_pack(int dtype, const float* w, void* fwd, void* tr, int n, int k, void* stream) {
  if (n <= 0 || k <= 0 || n % 4 || k % 4) return -1112;
  dim3 grid((unsigned)((k + 63) / 64), (unsigned)((n + 63) / 64));
  if (dtype == 0) hipLaunchKernelGGL(linear_pack_kernel<bf16>, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16*)fwd, (bf16*)tr, n, k);
  else hipLaunchKernelGGL(linear_pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, w, (float*)fwd, (float*)tr, n, k);
  STSWIN_CHECK_LAUNCH();
  return 0;
}
